#!/usr/bin/env python
"""Price a software grid barrier on the MI355X (VERDICT r03 "next round" item 1): compiles scripts/microbench/grid_barrier_bench.hip with
hipcc for gfx950 and runs it on cuda:0.  Output: microseconds per publish -> barrier -> consume round for device-wide and XCD-local
barriers, with and without fence instructions, next to a dependent kernel launch.  The decision it feeds is in DESIGN.md section 7."""
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
src = os.path.join(HERE, "microbench", "grid_barrier_bench.hip")
exe = os.path.join(tempfile.gettempdir(), "grid_barrier_bench")
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-o", exe, src])
sys.exit(subprocess.call(["timeout", "120", exe]))
