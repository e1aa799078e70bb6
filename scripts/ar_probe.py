import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group(os.environ.get("TUBER_DIST_BACKEND", "gloo"), rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
torch.cuda.set_device(0)
x = torch.ones(50_000_000, device="cuda")
def t(fn, name):
    torch.cuda.synchronize(); dist.barrier(); t0 = time.time(); fn(); torch.cuda.synchronize(); dt = time.time() - t0
    if dist.get_rank() == 0: print("%-40s %.1f ms" % (name, dt * 1e3), flush=True)
for _ in range(2):
    t(lambda: dist.all_reduce(x), "full sync")
    t(lambda: dist.all_reduce(x[2_000_000:]), "slice sync")
    def a():
        h1 = dist.all_reduce(x[2_000_000:], async_op=True); h2 = dist.all_reduce(x[:2_000_000], async_op=True); h1.wait(); h2.wait()
    t(a, "two async slices")
