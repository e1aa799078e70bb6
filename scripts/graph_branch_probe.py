"""Does a captured hipGraph with two parallel branches overlap them on this ROCm?  Branch A = a chain of tiny dependent kernels
(latency-bound, 1 workgroup each), branch B = a chain of large GEMMs (throughput-bound).  Prints the replay time of A alone, B alone,
A then B on one stream, and A || B forked inside the capture."""
import time
import torch

dev = torch.device("cuda:0")
small = torch.zeros(256, device=dev)
big = torch.randn(4096, 4096, device=dev, dtype=torch.bfloat16)
out = torch.empty_like(big)
side = torch.cuda.Stream()


def chain_a(n=300):
    for _ in range(n):
        small.add_(1.0)


def chain_b(n=20):
    for _ in range(n):
        torch.mm(big, big, out=out)


def capture(fn):
    g = torch.cuda.CUDAGraph()
    fn()
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        fn()
    return g


def both_serial():
    chain_a(); chain_b()


def both_forked():
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        chain_b()
    chain_a()
    main.wait_stream(side)


def t(g, it=20):
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(it):
        g.replay()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / it


for name, fn in (("A: 300 tiny dependent kernels", chain_a), ("B: 20 bf16 GEMMs 4096^3", chain_b), ("A then B, one stream", both_serial),
                 ("A || B, forked in the capture", both_forked)):
    print("%-34s %8.3f ms per replay" % (name, t(capture(fn))), flush=True)
