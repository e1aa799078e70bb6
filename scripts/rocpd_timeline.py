"""Timeline of the LAST replayed step in a rocprofv3 rocpd database (graph-mode bench): per-queue kernel time by kernel,
and the main-queue chain split into phases at marker kernels."""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
names = {r[0]: r[1] for r in cur.execute("select id, kernel_name from %s" % ks)}
rows = list(cur.execute("select start,end,queue_id,kernel_id from %s order by start" % kd))
# one step = from a stem_conv_fwd launch to the next one; the SHORTEST one is a hipGraph replay (the eager warm-up / per-kernel timing
# passes of bench.py are host-bound and much longer)
idx = [i for i, r in enumerate(rows) if "stem_conv_fwd" in names[r[3]]]
spans = [(rows[(idx[j + 1] if j + 1 < len(idx) else len(rows)) - 1][1] - rows[idx[j]][0], j) for j in range(len(idx) - 1)]
best = min(spans)[1] if spans else len(idx) - 1
first, last = idx[best], (idx[best + 1] if best + 1 < len(idx) else len(rows))
# the weight refresh kernels right before the stem (multi_transpose_bf16, cast) belong to the step; the ones before the NEXT stem do not
lo = first
while first > 0 and rows[first][0] - rows[first - 1][1] < 50_000 and first > lo - 12:
    first -= 1
while last > first and last - 1 > idx[best] and any(k in names[rows[last - 1][3]] for k in ("multi_transpose_bf16", "multi_cast_transpose", "cast_f32_bf16", "stem_pack_weight")):
    last -= 1
step = rows[first:last]
t0 = step[0][0]
print("step span %.3f ms, %d kernels" % ((step[-1][1] - t0) / 1e6, len(step)))
perq = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
for s, e, q, k in step:
    n = names[k].split("(")[0][:60]
    perq[q][n][0] += 1
    perq[q][n][1] += e - s
for q, d in perq.items():
    tot = sum(v[1] for v in d.values())
    print("== queue %s: %d kernels, %.3f ms kernel time" % (q, sum(v[0] for v in d.values()), tot / 1e6))
    for n, v in sorted(d.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
        print("   %-62s n=%-4d %8.3f ms" % (n, v[0], v[1] / 1e6))
# phases on the timeline
marks = [("stem_conv_fwd", "backbone fwd start"), ("posenc", "transformer start"), ("criterion_cost", "cost"),
         ("criterion_loss", "criterion"), ("stem_pool_bwd", "stem pool bwd"), ("sumsq_partial", "clip+adamw")]
for key, label in marks:
    hit = [r for r in step if key in names[r[3]]]
    if hit:
        print("%-22s first at %8.3f ms, last ends %8.3f ms" % (label, (hit[0][0] - t0) / 1e6, (hit[-1][1] - t0) / 1e6))
# idle gaps on the union of queues
last = step[0][0]
gap = 0
for s, e, _, _ in step:
    if s > last:
        gap += s - last
    last = max(last, e)
print("idle (no kernel on any queue) %.3f ms" % (gap / 1e6))
# per-bottleneck-block wall time: deltas between consecutive block_out_fwd / block_out_bwd launches
for key in ("block_out_fwd", "block_out_bwd"):
    ts = [r[0] for r in step if key in names[r[3]]]
    d = [(b - a) / 1e3 for a, b in zip(ts, ts[1:])]
    print(key, "deltas (us):", " ".join("%.0f" % x for x in d))
for key in ("criterion_loss", "block_out_bwd", "stem_pool_bwd", "stem_conv_bwd_w", "sumsq_partial", "adamw"):
    hit = [r for r in step if key in names[r[3]]]
    if hit:
        print("%-18s first start %8.3f last end %8.3f (queues %s)" % (key, (hit[0][0] - t0) / 1e6, (hit[-1][1] - t0) / 1e6, sorted({r[2] for r in hit})))
if len(sys.argv) > 4:
    a, b = float(sys.argv[3]) * 1e6 + t0, float(sys.argv[4]) * 1e6 + t0
    win = [r for r in step if a <= r[0] < b]
    agg = collections.defaultdict(lambda: [0, 0])
    for s, e, q, k in win:
        agg[names[k].split("(")[0][:70]][0] += 1
        agg[names[k].split("(")[0][:70]][1] += e - s
    print("window %.3f..%.3f ms: %d kernels, kernel time %.3f ms" % (float(sys.argv[3]), float(sys.argv[4]), len(win), sum(v[1] for v in agg.values()) / 1e6))
    for n, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print("   %-72s n=%-4d %8.3f ms" % (n, v[0], v[1] / 1e6))
    last = win[0][0]; gap = 0; big = []
    for s, e, q, k in win:
        if s > last:
            gap += s - last
            if s - last > 20000: big.append(((s - t0) / 1e6, (s - last) / 1e3, names[k][:40]))
        last = max(last, e)
    print("   idle in window %.3f ms; gaps > 20us:" % (gap / 1e6), big[:30])
if len(sys.argv) > 5 and sys.argv[5] == "block":
    # kernels of the N-th bottleneck block in backward (between consecutive block_out_bwd launches) and forward
    for key, nth in [(k.split(":")[0], int(k.split(":")[1])) for k in (sys.argv[6].split(",") if len(sys.argv) > 6 else ["block_out_bwd:20", "block_out_fwd:25"])]:
        idx = [i for i, r in enumerate(step) if key in names[r[3]]]
        a, b = idx[nth], idx[nth + 1]
        print("---- kernels from %s #%d to #%d" % (key, nth, nth + 1))
        for s, e, q, k in step[a:b]:
            print("   %8.1f us  +%6.1f  %s" % ((e - s) / 1e3, (s - step[a][0]) / 1e3, names[k][:90]))
