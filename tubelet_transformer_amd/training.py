"""One optimisation step of TubeR on the MI355X path -- the body of the reference's hot loop
(``train_tuber_detection``, utils/video_action_recognition.py:96-157) without its per-step host syncs:

    outputs = model(samples) ; loss_dict = criterion(outputs, targets) ; losses = sum_k w_k * loss_k
    optimizer.zero_grad() ; losses.backward() ; clip_grad_norm_(0.1) ; optimizer.step()

The reference additionally copies ``pred_logits`` to the host every step as a NaN probe (:140) and calls ``.item()`` five
times on rank 0 (:182-193); here the only device->host transfer of a step is the single matcher cost copy, and logging
reads the (still on-device) loss tensors only when asked to.
"""
import os

import torch

from .ddp import FlatGradReducer, broadcast_parameters
from .optim import FusedClipAdamW, build_param_groups


def deploy_model(model, cfg, device=None):
    """Counterpart of utils/model_utils.py:39-58 for the flat-gradient data-parallel path: moves the model to this rank's
    GPU, broadcasts rank 0's parameters and attaches the gradient reducer.  Returns the (unwrapped) model."""
    import torch.distributed as dist
    dev = torch.device(device if device is not None else "cuda:%d" % cfg.DDP_CONFIG.GPU)
    model.to(dev)
    store, _ = model.engine()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        broadcast_parameters(store)
        store.reducer = FlatGradReducer(store)
    return model


def build_optimizer(model, cfg):
    T = cfg.CONFIG.TRAIN
    return FusedClipAdamW(build_param_groups(model, cfg), lr=T.LR, weight_decay=T.W_DECAY, model=model)


def train_step(model, criterion, optimizer, samples, targets, max_norm, epoch=0, cfg=None):
    """Returns (total loss tensor, loss dict) -- all on the device, nothing synchronised."""
    store, _ = model.engine()
    reducer = getattr(store, "reducer", None)
    outputs = model(samples)
    loss_dict = criterion(outputs, targets)
    weight_dict = criterion.weight_dict
    if cfg is not None and epoch > cfg.CONFIG.LOSS_COFS.WEIGHT_CHANGE:          # video_action_recognition.py:145-146
        weight_dict["loss_ce"] = cfg.CONFIG.LOSS_COFS.LOSS_CHANGE_COF
    losses = criterion.weighted_total(loss_dict, weight_dict)
    optimizer.zero_grad()
    if reducer is not None:
        reducer.begin()
    losses.backward()
    store.side_join()
    if reducer is not None:
        reducer.finish()
    optimizer.step(max_norm=max_norm if max_norm and max_norm > 0 else None)
    return losses.detach(), loss_dict


class GraphedTrainStep:
    """The same optimisation step replayed from ONE captured hipGraph (HIP graphs instead of a tracing compiler): bf16 weight
    refresh, forward of the whole model, matching-cost kernel, Hungarian assignment on the device (tuber_lsap_device), fused
    criterion (losses + output gradients), backward of the whole model, global-norm clip + AdamW.  With >1 rank the optimizer is
    a second graph behind an eager RCCL all-reduce of the flat gradient buffer; assignment problems beyond the device solver's
    128 x 128 bound split the graph around the host tuber_lsap.

    ~2000 kernel launches per step are issued by the GPU front-end instead of Python, so the step is GPU-bound.
    Inputs are copied into static device buffers; targets use the padded [B, Tmax] layout, so clips with a different number
    of boxes replay the same graphs (shape changes of the clip batch re-capture).  Dropout masks differ every replay (the seed
    lives in device memory), AdamW reads its step count from device memory.
    """

    def __init__(self, model, criterion, optimizer, max_norm, tmax=16):
        self.model, self.criterion, self.optimizer = model, criterion, optimizer
        self.max_norm, self.tmax = max_norm, tmax
        self.graphs = {}

    def _capture(self, clips, targets):
        from .criterion import PaddedTargets
        from .misc import NestedTensor
        model, crit, opt = self.model, self.criterion, self.optimizer
        store, _ = model.engine()
        dev = store.device
        # the hooks of the eager reducer must not fire inside a stream capture: in graph mode the gradient all-reduce is
        # issued eagerly between graph B1 (backward) and B2 (optimizer), so detach the reducer for good
        red = getattr(store, "reducer", None)
        world = getattr(red, "world", 1) if red is not None else getattr(self, "world", 1)
        self.world = world
        g = type("Captured", (), {})()
        g.clips = clips.clone()
        g.mask = torch.zeros(clips.shape[0], clips.shape[-2], clips.shape[-1], dtype=torch.bool, device=dev)
        g.pt = PaddedTargets(targets, crit.ava, crit.num_classes if crit.ava else crit.num_classes + 1, dev, tmax=self.tmax)
        g.targets = targets
        # eager warm-up: sizes every persistent workspace before capture
        for _ in range(2):
            train_step(model, crit, opt, NestedTensor(g.clips, g.mask), targets, self.max_norm)
        torch.cuda.synchronize()
        store.reducer = None
        if red is not None:
            # the eager reducer made the backward flush its deferred reductions once per bottleneck; the captured backward flushes
            # twice.  One forward + backward WITHOUT the reducer (no optimizer step: ranks stay in sync) builds those reduce tables now,
            # because nothing can be uploaded during the capture.
            ld = crit(model(NestedTensor(g.clips, g.mask)), targets)
            opt.zero_grad()
            crit.weighted_total(ld).backward()
            store.side_join()
            torch.cuda.synchronize()
        # With the assignment on the device (tuber_lsap_device) the whole step is ONE graph: refresh, forward, matching cost,
        # assignment, fused criterion, backward, clip + AdamW -- no device->host round trip.  (DDP: the optimizer is a second
        # graph behind the eager RCCL all-reduce.  Problems beyond the device solver's 128 x 128 bound use graph A / host / B.)
        Q = model.query_embed.num_embeddings
        g.on_device = Q <= 128 and g.pt.tmax <= 128
        g.A = torch.cuda.CUDAGraph()
        g.B1 = None

        def head():
            outputs = model(NestedTensor(g.clips, g.mask))
            logits, logits_b, boxes = crit.stacked(outputs)
            g.logits_s, g.boxes_s = crit.select(logits, boxes, targets)
            g.logits_b = logits_b
            with torch.no_grad():
                g.cost = crit.matcher.cost(g.logits_s.detach().contiguous(),
                                           (logits_b if crit.ava else g.logits_s).detach().contiguous(),
                                           g.boxes_s.detach().contiguous(), g.pt)

        def tail():
            g.loss_dict = crit.losses_from_match(g.logits_s, g.logits_b, g.boxes_s, g.pt, g.match, targets)
            g.loss_dict["class_error"] = crit.class_error(g.logits_s[-1], g.pt, g.match[-1])
            g.loss = crit.weighted_total(g.loss_dict)
            opt.zero_grad()
            g.loss.backward()
            store.side_join()
            if world == 1:
                opt.step(max_norm=self.max_norm if self.max_norm and self.max_norm > 0 else None)

        g.A2, g.split = None, None
        # Opt-in (TUBER_SPLIT_GRAPH=1): verified bit-exact on one GPU and run end-to-end with 2 gloo ranks, but this build has never
        # had a multi-GPU box to validate it under RCCL, so the default DDP step keeps the plain form (graph, all-reduce, graph).
        split = g.on_device and bool(os.environ.get("TUBER_SPLIT_GRAPH") or os.environ.get("TUBER_FORCE_SPLIT_GRAPH")) \
            and not os.environ.get("TUBER_NO_SPLIT_GRAPH")
        if split:
            # DDP: graph A is cut where layer3's backward ends (~95 % of the gradient bytes are final there), so the RCCL
            # all-reduce of that slice runs under the layer2 / layer1 / stem backward (graph A2) instead of after it.
            _, runner = model.engine()
            cut = {}
            a2 = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            cs = torch.cuda.Stream()
            cs.wait_stream(torch.cuda.current_stream())
            try:
                with torch.cuda.stream(cs):
                    g.A.capture_begin(capture_error_mode="relaxed")      # the cut happens on autograd's worker thread

                    def hook(off):
                        if "off" not in cut:
                            g.A.capture_end()
                            cut["off"] = off
                            a2.capture_begin(pool=g.A.pool(), capture_error_mode="relaxed")
                    runner.split_hook = hook
                    try:
                        head()
                        g.match = crit.assign(g.cost, g.pt)
                        tail()
                    finally:
                        runner.split_hook = None
                    if "off" in cut:
                        a2.capture_end()
                        g.A2, g.split, g.body_begin = a2, int(cut["off"]), int(runner.body_begin)
                    else:
                        g.A.capture_end()
                torch.cuda.current_stream().wait_stream(cs)
            except Exception:
                raise
        elif g.on_device:
            with torch.cuda.graph(g.A):
                head()
                g.match = crit.assign(g.cost, g.pt)
                tail()
        else:
            with torch.cuda.graph(g.A):
                head()
            L, B = g.cost.shape[:2]
            g.match = torch.full((L, B, g.pt.tmax), -1, dtype=torch.int32, device=dev)
            g.match_host = torch.full((L, B, g.pt.tmax), -1, dtype=torch.int32).pin_memory()
            g.cost_host = torch.empty(g.cost.shape, dtype=torch.float32).pin_memory()
            g.B1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g.B1, pool=g.A.pool()):
                tail()
        g.B2 = None
        if world > 1:
            g.B2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g.B2, pool=g.A.pool()):
                opt.step(max_norm=self.max_norm if self.max_norm and self.max_norm > 0 else None)
        return g

    def __call__(self, clips, targets):
        key = tuple(clips.shape)
        g = self.graphs.get(key)
        if g is None:
            g = self.graphs[key] = self._capture(clips, targets)
        store, _ = self.model.engine()
        g.clips.copy_(clips, non_blocking=True)
        sizes = [int(t["boxes"].shape[0]) for t in targets]
        if targets is not g.targets:
            g.pt.sizes = sizes
            g.pt.tboxes.zero_()
            g.pt.tlabels.zero_()
            g.pt.tcount.copy_(torch.tensor(sizes, dtype=torch.int32), non_blocking=True)
            g.pt.fill(targets)
        g.A.replay()
        if g.on_device:
            self.criterion._indices, self.criterion._match_dev = None, (g.match, sizes)
        else:
            g.cost_host.copy_(g.cost, non_blocking=True)
            torch.cuda.current_stream().synchronize()             # host assignment: the step's one host sync
            match, indices = self.criterion.matcher.solve(g.cost_host.numpy(), sizes)
            g.match_host.copy_(torch.from_numpy(match))
            g.match.copy_(g.match_host, non_blocking=True)
            L = len(indices)
            self.criterion.last_indices = [indices[L - 1]] + indices[:L - 1]
            g.B1.replay()
        if g.A2 is not None:
            import torch.distributed as dist
            ddp = g.B2 is not None and dist.is_initialized()
            dbg = os.environ.get("TUBER_DEBUG_TIMING")
            if dbg:
                import time
                torch.cuda.synchronize(); t0 = time.time()
            # final at the cut: layer3, layer4 and everything laid out behind the body [split, total) and everything laid out
            # before it (transformer, heads: [0, body_begin)); pending: stem, layer1, layer2 [body_begin, split)
            bb = g.body_begin
            h1 = [dist.all_reduce(store.gflat[g.split:], op=dist.ReduceOp.SUM, async_op=True),
                  dist.all_reduce(store.gflat[:bb], op=dist.ReduceOp.SUM, async_op=True)] if ddp else None
            if dbg:
                t1 = time.time()
            g.A2.replay()                                     # layer2 / layer1 / stem backward, under the all-reduce
            if dbg:
                t2 = time.time()
            if ddp:
                h2 = dist.all_reduce(store.gflat[bb:g.split], op=dist.ReduceOp.SUM, async_op=True)
                for h in h1:
                    h.wait()
                if dbg:
                    t3 = time.time()
                h2.wait()
                if dbg:
                    torch.cuda.synchronize()
                    print("split step: issue ar1 %.1f ms, replay A2 %.1f ms, wait ar1 %.1f ms, ar2+sync %.1f ms (phase 1: [%d,%d) + [0,%d) of %d floats)"
                          % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (time.time() - t3) * 1e3, g.split, store.total, bb, store.total), flush=True)
                from . import lib
                lib.call("tuber_scale_f32", store.gflat, store.total, None, 1.0 / dist.get_world_size())
            if g.B2 is not None:
                g.B2.replay()
            return g.loss, g.loss_dict
        if g.B2 is not None:
            import torch.distributed as dist
            dist.all_reduce(store.gflat, op=dist.ReduceOp.SUM)
            from . import lib
            lib.call("tuber_scale_f32", store.gflat, store.total, None, 1.0 / dist.get_world_size())
            g.B2.replay()
        return g.loss, g.loss_dict


def train_tuber_detection(cfg, model, criterion, data_loader, optimizer, epoch, max_norm, lr_scheduler=None, writer=None,
                          graphed=None, print_freq=10):
    """One training epoch -- the reference's ``train_tuber_detection`` (utils/video_action_recognition.py:64-220) on the HIP path:
    every batch is one ``train_step`` (or one replay of ``graphed``, a ``GraphedTrainStep``); losses are read back only every
    ``print_freq`` iterations so the GPU queue stays full.  ``epoch > LOSS_COFS.WEIGHT_CHANGE`` switches ``loss_ce``'s weight
    like the reference (:145-146)."""
    import time
    model.train()
    criterion.train()
    dev = next(model.parameters()).device
    rank = getattr(cfg.DDP_CONFIG, "GPU_WORLD_RANK", 0)
    if epoch > cfg.CONFIG.LOSS_COFS.WEIGHT_CHANGE:
        criterion.weight_dict["loss_ce"] = cfg.CONFIG.LOSS_COFS.LOSS_CHANGE_COF
    end = time.time()
    loss = None
    for idx, data in enumerate(data_loader):
        samples, targets = data[0], data[1]
        samples = samples.to(dev)            # reference :121; for an input_pipeline.ClipBatch this IS the HIP pre-pass (uint8 frames -> fp32 batch)
        targets = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in t.items() if k != "image_id"} for t in targets]
        if graphed is not None:
            clips = (samples.tensors if hasattr(samples, "tensors") else samples).to(dev, torch.float32)
            loss, loss_dict = graphed(clips, targets)
        else:
            loss, loss_dict = train_step(model, criterion, optimizer, samples, targets, max_norm, epoch=epoch, cfg=cfg)
        if lr_scheduler is not None and cfg.CONFIG.TRAIN.LR_POLICY == "cosine":
            lr_scheduler.step_update(epoch * len(data_loader) + idx)
        if rank == 0 and (idx % print_freq == 0 or idx + 1 == len(data_loader)):
            lv = float(loss.detach())                          # the only host sync, every print_freq iterations
            if lv != lv or lv in (float("inf"), float("-inf")):
                raise FloatingPointError("loss is %r at epoch %d iteration %d" % (lv, epoch, idx))
            print("Epoch: [%d][%d/%d]  %.3f s/iter  loss %.4f  " % (epoch, idx + 1, len(data_loader), (time.time() - end), lv) +
                  ", ".join("%s %.4f" % (k, float(v.detach() if torch.is_tensor(v) else v)) for k, v in loss_dict.items() if k in ("loss_ce", "loss_bbox", "loss_giou", "loss_ce_b", "class_error")))
            if writer is not None:
                writer.add_scalar("train/loss", lv, idx + epoch * len(data_loader))
        end = time.time()
    return loss
