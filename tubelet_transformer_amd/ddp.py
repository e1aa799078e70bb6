"""Data-parallel training over the GPUs of one node: one process per GPU, RCCL all-reduce over xGMI.

The reference wraps the model in ``DistributedDataParallel(find_unused_parameters=True)`` (utils/model_utils.py:39-58,
pipelines/launch.py:20-50).  Here every parameter gradient already lives in ONE flat fp32 buffer in ``named_parameters()``
order -- transformer, embeddings, projections, class-branch encoder, heads, THEN the CSN body (stem, layer1..4) and the pool
decoder -- so the reducer is a handful of large all-reduces on contiguous slices, issued from the backward pass as soon as a
slice is final and overlapped with the remaining backward kernels on RCCL's own stream:

    backward reaches ...            slice that is final              -> dist.all_reduce(slice, async_op=True)
    body backward entry             everything laid out behind the body (pool decoder)
    end of layer4 / 3 / 2 / 1       that stage's parameters (offsets >= the stage's first block)
    end of backward                 the rest: stem, and the transformer / head slice laid out before the body

(The hipGraph step, training.GraphedTrainStep, does not use these hooks: it cuts its graph once, where layer3's backward ends,
and reduces [layer3 .. end) and [0 .. body) under the layer2 / layer1 / stem backward.)

No bucket copies, no unused-parameter bitmap (C2), no per-step buffer broadcast (C3): BatchNorm statistics stay local
like the reference's non-synchronised BatchNorm3d; rank 0's running statistics are what a checkpoint holds.
xGMI is point-to-point (7 links x ~153 GB/s per GPU): a few 25-100 MB messages keep RCCL in its bandwidth regime.
"""
import torch
import torch.distributed as dist


def trainable_ranges(store):
    """merged [begin, end) windows of the flat buffers that belong to parameters with ``requires_grad``."""
    out = []
    for n, p in zip(store.names, store.params):
        if not p.requires_grad:
            continue
        o = store.offsets[n]
        e = o + (p.numel() + 63) // 64 * 64
        if out and out[-1][1] == o:
            out[-1][1] = e
        else:
            out.append([o, e])
    return [(a, b) for a, b in out]


class FlatGradReducer:
    def __init__(self, store, world_size=None, min_bucket=8 << 20):
        self.store = store
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.min_bucket = min_bucket          # elements
        self.handles = []
        self.done_from = store.total          # everything at offsets >= done_from has been handed to RCCL
        self.late = []                        # [(begin, end)] slices that must wait for the end of backward
        for n in store.names:
            if n.endswith("query_embed.weight") or n.endswith("query_pool.weight"):
                o = store.offsets[n]
                self.late.append((o, o + (store.module.get_parameter(n).numel() + 63) // 64 * 64))
        self.late.sort()

    def begin(self):
        self.handles = []
        self.done_from = self.store.total
        self.ranges = trainable_ranges(self.store)       # windows of frozen parameters hold no gradient: never sent

    def _reduce(self, lo, hi):
        if hi <= lo or self.world <= 1:
            return
        for a, b in self.ranges:
            a, b = max(a, lo), min(b, hi)
            if b > a:
                self.handles.append(dist.all_reduce(self.store.gflat[a:b], op=dist.ReduceOp.SUM, async_op=True))

    def _reduce_excluding_late(self, lo, hi):
        cur = lo
        for a, b in self.late:
            if b <= lo or a >= hi:
                continue
            self._reduce(cur, max(cur, a))
            cur = max(cur, b)
        self._reduce(cur, hi)

    def notify(self, offset, force=False):
        """Backward has finished every parameter at flat offsets >= ``offset``."""
        if offset >= self.done_from:
            return
        if not force and self.done_from - offset < self.min_bucket:
            return
        self._reduce_excluding_late(offset, self.done_from)
        self.done_from = offset

    def finish(self):
        """End of backward: reduce what is left (stem + late slices), wait, and average."""
        self.notify(0, force=True)
        for a, b in self.late:
            self._reduce(a, b)
        for h in self.handles:
            h.wait()
        self.handles = []
        if self.world > 1:
            from . import lib
            if self.store.gflat.is_cuda:
                lib.call("tuber_scale_f32", self.store.gflat, self.store.total, None, 1.0 / self.world)
            else:
                self.store.gflat.mul_(1.0 / self.world)


def broadcast_parameters(store, src=0):
    """Initial parameter + buffer broadcast (DDP constructor, collective C4)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    dist.broadcast(store.flat, src)
    for b in store.module.buffers():
        dist.broadcast(b, src)
