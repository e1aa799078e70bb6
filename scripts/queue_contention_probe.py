"""Round 6: what does a SECOND HIP stream cost a launch-bound captured graph on this runtime?

The one-rank forced-DDP line is +0.55 - 0.8 ms although the one-rank in-place ncclAllReduce does nothing on the device (5 us of stream
time) -- and the same with the call skipped, with device-scope events, with fence-free events (scripts/r06_ddp_probe.sh, r06_ddp_edge.sh).
What is left is the pair of stream edges itself.  This probe replays a captured graph of N tiny dependent kernels on stream `main` and
times the replay (mean over rounds, GPU fully queued) for:
  alone            nothing else
  edge_after       after every replay: side waits for main (event), main waits for side     <- the reducer's pattern at one cut
  side_wait_only   after every replay: side waits for main; nothing joins back
  record_only      after every replay: an event is recorded on main, nobody waits for it
  side_busy        a side stream keeps ONE long-running 1-workgroup kernel in flight next to every replay (no dependency at all)
  host_edge        after every replay the HOST waits for main (event synchronize), then launches on side, main waits for side
usage: python scripts/queue_contention_probe.py [n_kernels]"""
import sys
import time

import torch

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 900
x = torch.zeros(4096, device=dev)
y = torch.zeros(1, device=dev)
main = torch.cuda.Stream()
side = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(main):
    for _ in range(3):
        x.add_(1.0)
    main.synchronize()
    with torch.cuda.graph(g, stream=main):
        for _ in range(N):
            x.add_(1.0)
big = torch.zeros(64 << 20, device=dev)


def long_kernel():          # ~ a few ms on a fraction of the chip
    big.add_(1.0)


def run(mode, rounds=60):
    torch.cuda.synchronize()
    t0 = None
    with torch.cuda.stream(main):
        for r in range(rounds + 5):
            if r == 5:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            if mode == "side_busy":
                with torch.cuda.stream(side):
                    long_kernel()
            g.replay()
            if mode == "edge_after":
                side.wait_stream(main)
                main.wait_stream(side)
            elif mode == "side_wait_only":
                side.wait_stream(main)
            elif mode == "record_only":
                e = torch.cuda.Event()
                e.record(main)
            elif mode == "host_edge":
                e = torch.cuda.Event()
                e.record(main)
                e.synchronize()
                with torch.cuda.stream(side):
                    y.add_(1.0)
                main.wait_stream(side)
        torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / rounds


for rep in range(2):
    for mode in ("alone", "edge_after", "side_wait_only", "record_only", "side_busy", "host_edge"):
        print("%-16s %8.3f ms per replay of %d kernels" % (mode, run(mode), N), flush=True)
