"""Instruction budget of one kernel of an ISA listing (hipcc -S --cuda-device-only): VALU instructions by purpose, split into the k-loop
(basic blocks that contain MFMAs and sit inside a backward branch) and everything outside it (prologue + epilogue), weighted by trip
count.  VERDICT r05 item 6: where do the 16 VALU per MFMA of the join GEMM / the BatchNorm-prologue conv4 go?
usage: python scripts/isa_budget.py /tmp/gemm.s <mangled-name-substring> <k-loop trip count> [<second loop trip count> ...]"""
import collections
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
trips = [int(x) for x in sys.argv[3:]] or [1]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and pat in l and ":" in l)
end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
body = lines[start + 1:end]


def cls(op):
    if op.startswith("v_mfma"):
        return "MFMA"
    if op.startswith("v_cvt_pk_bf16") or op.startswith("v_cvt"):
        return "valu: convert (bf16 <-> f32)"
    if op.startswith(("v_lshlrev_b32", "v_and_b32", "v_lshrrev_b32", "v_and_or_b32", "v_lshl_or_b32", "v_bfe", "v_perm_b32", "v_or_b32", "v_lshl_add_u32", "v_alignbit")):
        return "valu: shift / mask / pack (bf16 unpack, address bits)"
    if op.startswith(("v_add_u32", "v_add_co", "v_addc", "v_mad_u64", "v_mad_i64", "v_mul_lo", "v_mul_hi", "v_sub_u32", "v_add3_u32", "v_mad_u32", "v_lshl_add_u64", "v_add_nc", "v_sub_co", "v_subb", "v_mul_u32", "v_mad_i32", "v_ashr", "v_xor", "v_xad")):
        return "valu: integer / address arithmetic"
    if op.startswith(("v_cndmask", "v_cmp", "v_cmpx")):
        return "valu: compare / select (masks, bounds)"
    if "dpp" in op or op.startswith(("v_permlane", "v_readlane", "v_readfirstlane", "v_writelane", "ds_bpermute", "ds_swizzle")):
        return "valu: cross-lane (DPP / permlane: statistics reductions)"
    if op.startswith(("v_pk_fma", "v_pk_mul", "v_pk_add", "v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_max", "v_min", "v_fma_mix", "v_mad", "v_rcp", "v_rsq", "v_exp", "v_log", "v_pk_max", "v_med3", "v_fma_f64", "v_add_f64", "v_mul_f64")):
        return "valu: float arithmetic (BN apply, residual, ReLU, statistics)"
    if op.startswith(("v_mov", "v_accvgpr", "v_swap", "v_pk_mov", "v_nop")):
        return "valu: moves"
    if op.startswith("v_"):
        return "valu: other (%s)" % op
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_barrier"):
        return "wait / barrier"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds " + ("read" if "read" in op or "load" in op else "write")
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "global load"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic")):
        return "global store"
    return "other (%s)" % op


# basic blocks
blocks, cur, label = [], [], "entry"
for l in body:
    t = l.strip()
    if re.match(r"^\.?LBB[\w]*:", t):
        if cur:
            blocks.append((label, cur))
        label, cur = t.split(":")[0], []
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    op = t.split()[0]
    cur.append((op, t))
    if op.startswith(("s_cbranch", "s_branch")):
        blocks.append((label, cur))
        label, cur = label + "'", []
if cur:
    blocks.append((label, cur))
names = [b[0] for b in blocks]
# loops: a branch to an earlier label
loops = []
for i, (lab, ins) in enumerate(blocks):
    for op, t in ins:
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = t.split()[-1]
            if tgt in names and names.index(tgt) <= i:
                loops.append((names.index(tgt), i))
mf = lambda a, b: sum(1 for k in range(a, b + 1) for op, _ in blocks[k][1] if op.startswith("v_mfma"))
loops = sorted(set(loops), key=lambda ab: -mf(*ab))
loops = [ab for ab in loops if mf(*ab) > 0]
weight = [1] * len(blocks)
print("kernel %s: %d instructions, %d basic blocks; loops with MFMAs (block range, MFMAs): %s" % (lines[start].rstrip(":")[:70], sum(len(b[1]) for b in blocks), len(blocks), [(ab, mf(*ab)) for ab in loops]))
for (a, b), trip in zip(loops, trips):
    for k in range(a, b + 1):
        weight[k] = max(weight[k], trip)
tot, inloop, outloop = collections.Counter(), collections.Counter(), collections.Counter()
for k, (lab, ins) in enumerate(blocks):
    for op, t in ins:
        c = cls(op)
        tot[c] += weight[k]
        (inloop if weight[k] > 1 else outloop)[c] += weight[k]
nm = tot["MFMA"]
valu = sum(v for k, v in tot.items() if k.startswith("valu"))
print("dynamic count per wave with trip counts %s: %d MFMA, %d VALU = %.1f VALU per MFMA" % (trips, nm, valu, valu / max(nm, 1)))
print("%-64s %8s %8s %8s %10s" % ("class", "k-loop", "outside", "total", "per MFMA"))
for k in sorted(tot, key=lambda k: -tot[k]):
    print("%-64s %8d %8d %8d %10.2f" % (k, inloop[k], outloop[k], tot[k], tot[k] / max(nm, 1)))
