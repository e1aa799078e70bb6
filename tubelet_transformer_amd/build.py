"""Build libtuber_hip.so (gfx950) in-tree with hipcc.  No cmake, no JIT cache:
the .so lands next to the sources so it travels with the repo snapshot to the GPU box."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib")
LIB = os.path.join(OUT, "libtuber_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast", "-Wno-unused-result"]
if os.environ.get("TUBER_AB_VARIANTS"):      # also build the measured-and-rejected GEMM tile variants (tuning runs only; not the product library)
    FLAGS.append("-DTUBER_AB_VARIANTS")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))


def _stamp():
    """compile flags + compiler path: part of every object's staleness key (an object built with other flags is stale
    even when its source is older)"""
    return " ".join([HIPCC] + FLAGS)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OUT, exist_ok=True)
    stamp_file = os.path.join(OUT, "FLAGS.stamp")
    if not os.path.exists(stamp_file) or open(stamp_file).read() != _stamp():
        force = True                                  # objects of another flag set (or of unknown provenance): rebuild all of them
    hdrs = glob.glob(os.path.join(CSRC, "*.h"))
    objs, jobs = [], []
    for src in sources():
        obj = os.path.join(OUT, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            lang = ["-x", "hip"] if src.endswith(".hip") else []
            jobs.append([HIPCC] + FLAGS + lang + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-4000:]))
    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    with open(stamp_file, "w") as f:
        f.write(_stamp())
    if jobs or force or _stale(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
