"""One process per GPU (reference ``pipelines/launch.py:8-50``).

``spawn_workers(main, cfg)`` keeps the reference's call form (train_tuber_ava.py:103, eval_tuber_ava.py:64, ...):

* launched plainly (``python train_tuber_ava.py``): forks one worker per visible GPU with ``torch.multiprocessing.spawn``; worker
  ``g`` pins GPU ``g`` (``torch.cuda.set_device``), joins the process group described by ``cfg.DDP_CONFIG`` (``DIST_BACKEND`` --
  "nccl" is RCCL on ROCm -- ``DIST_URL``, rank ``WORLD_RANK * gpus + g`` of ``WORLD_SIZE * gpus``) and calls ``main(cfg)``;
* launched by ``python -m torch.distributed.run`` (RANK / LOCAL_RANK / WORLD_SIZE in the environment): this process IS the
  worker -- the rendez-vous comes from the environment (``env://``), nothing is forked;
* ``DDP_CONFIG.DISTRIBUTED: false``: runs ``main(cfg)`` in-process on ``DDP_CONFIG.GPU``.

``HSA_ENABLE_IPC_MODE_LEGACY=0`` is exported for the children: the host driver only supports dmabuf IPC and RCCL's
intra-node transport fails without it.
"""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def get_local_ip_and_match(ip_list):
    """index of this host in ``ip_list`` (pipelines/launch.py:8-17), -1 if absent.  The address is taken from the route towards
    the first peer (a UDP connect sends nothing), falling back to the host name's addresses on an isolated node."""
    # the reference matches ONE address: the source of the route to a public host (pipelines/launch.py:9-13).  Here: the source
    # address of the route towards the first OTHER entry of the list (no packet is sent by a UDP connect), then the public route,
    # then the host name's addresses; the loopback address only counts when nothing else was found (single isolated node).  More
    # than one matching entry -- a multi-homed host listed twice, a list that contains 127.0.0.1 next to real addresses -- would hand
    # the same WORLD_RANK to two nodes: that is an error, not index 0.
    def route_source(peer):
        try:
            s = socket.socket(socket.AF_INET, socket.SOCK_DGRAM)
            s.connect((peer, 80))
            ip = s.getsockname()[0]
            s.close()
            return ip
        except OSError:
            return None
    mine = []
    for peer in list(ip_list) + ["8.8.8.8"]:
        ip = route_source(peer)
        if ip and not ip.startswith("127.") and ip not in mine:
            mine.append(ip)
    if not mine:
        try:
            mine = [ip for ip in socket.gethostbyname_ex(socket.gethostname())[2] if not ip.startswith("127.")]
        except OSError:
            mine = []
    hits = [i for i, ip in enumerate(ip_list) if ip in mine]
    if not hits:                         # no real address of this host is listed: a loopback entry means "this node"
        hits = [i for i, ip in enumerate(ip_list) if ip.startswith("127.")]
    if len(hits) > 1:
        raise RuntimeError("AUTO_RANK_MATCH: this host's addresses %s match %d entries of WOLRD_URLS %s" % (mine, len(hits), list(ip_list)))
    return hits[0] if hits else -1


def main_worker(gpu, ngpus_per_node, main, cfg, from_env=False):
    """the per-GPU process body (pipelines/launch.py:37-50)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cfg.DDP_CONFIG.GPU = gpu
    print("Use GPU: {}".format(gpu))
    if torch.cuda.is_available() and gpu is not None:
        torch.cuda.set_device(gpu)
    if cfg.DDP_CONFIG.DISTRIBUTED:
        if from_env:
            cfg.DDP_CONFIG.GPU_WORLD_RANK = int(os.environ["RANK"])
            cfg.DDP_CONFIG.GPU_WORLD_SIZE = int(os.environ["WORLD_SIZE"])
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend=cfg.DDP_CONFIG.DIST_BACKEND, init_method="env://")
        else:
            cfg.DDP_CONFIG.GPU_WORLD_RANK = cfg.DDP_CONFIG.WORLD_RANK * ngpus_per_node + gpu
            dist.init_process_group(backend=cfg.DDP_CONFIG.DIST_BACKEND, init_method=cfg.DDP_CONFIG.DIST_URL,
                                    world_size=cfg.DDP_CONFIG.GPU_WORLD_SIZE, rank=cfg.DDP_CONFIG.GPU_WORLD_RANK)
    main(cfg)


def spawn_workers(main, cfg, nprocs=None):
    """pipelines/launch.py:20-34.  ``nprocs`` overrides the per-node worker count (default: the number of visible GPUs)."""
    if cfg.DDP_CONFIG.AUTO_RANK_MATCH:
        assert len(cfg.DDP_CONFIG.WOLRD_URLS) > 0
        assert cfg.DDP_CONFIG.WOLRD_URLS[0] in cfg.DDP_CONFIG.DIST_URL
        assert len(cfg.DDP_CONFIG.WOLRD_URLS) == cfg.DDP_CONFIG.WORLD_SIZE
        cfg.DDP_CONFIG.WORLD_RANK = get_local_ip_and_match(cfg.DDP_CONFIG.WOLRD_URLS)
        assert cfg.DDP_CONFIG.WORLD_RANK != -1
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if cfg.DDP_CONFIG.DISTRIBUTED and "RANK" in os.environ and "WORLD_SIZE" in os.environ and nprocs is None:
        return main_worker(int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("LOCAL_WORLD_SIZE", "1")), main, cfg, from_env=True)
    ngpus_per_node = nprocs if nprocs is not None else torch.cuda.device_count()
    if cfg.DDP_CONFIG.DISTRIBUTED:
        if ngpus_per_node < 1:
            raise RuntimeError("spawn_workers: no GPU visible (DDP_CONFIG.DISTRIBUTED is true)")
        cfg.DDP_CONFIG.GPU_WORLD_SIZE = ngpus_per_node * cfg.DDP_CONFIG.WORLD_SIZE
        mp.spawn(main_worker, nprocs=ngpus_per_node, args=(ngpus_per_node, main, cfg))
    else:
        main_worker(cfg.DDP_CONFIG.GPU, ngpus_per_node, main, cfg)
