"""One optimisation step of TubeR on the MI355X path -- the body of the reference's hot loop
(``train_tuber_detection``, utils/video_action_recognition.py:96-157) without its per-step host syncs:

    outputs = model(samples) ; loss_dict = criterion(outputs, targets) ; losses = sum_k w_k * loss_k
    optimizer.zero_grad() ; losses.backward() ; clip_grad_norm_(0.1) ; optimizer.step()

The reference additionally copies ``pred_logits`` to the host every step as a NaN probe (:140) and calls ``.item()`` five
times on rank 0 (:182-193); here a step has no device->host transfer at all (the Hungarian assignment runs on the device) and
logging reads the (still on-device) loss tensors only when asked to.
"""
import collections
import os

import torch

from . import ab
from .ddp import attach_reducer, broadcast_parameters
from .optim import FusedClipAdamW, adopt, build_param_groups


def deploy_model(model, cfg, is_tuber=True, device=None):
    """Counterpart of utils/model_utils.py:39-63, same call form (``deploy_model(model, cfg, is_tuber=True)``): selects this
    rank's GPU (``torch.cuda.set_device(cfg.DDP_CONFIG.GPU)``), moves the model there, and -- instead of wrapping it in
    ``DistributedDataParallel`` -- broadcasts rank 0's parameters / buffers and attaches the flat-gradient reducer (ddp.py)
    when a process group with more than one rank exists.  Then initialises the transformer from the DETR checkpoint
    ``CONFIG.MODEL.PRETRAIN_TRANSFORMER_DIR`` like the reference (:60-61); a blank path skips that step (the reference would
    fail in ``torch.load``; there are no checkpoints in an offline image).  Returns the (unwrapped) model."""
    import torch.distributed as dist
    gpu = getattr(cfg.DDP_CONFIG, "GPU", None)
    if device is not None:
        dev = torch.device(device)
    elif gpu is not None:
        dev = torch.device("cuda", int(gpu))
    else:
        dev = torch.device("cuda", torch.cuda.current_device())
    if dev.type == "cuda":
        torch.cuda.set_device(dev)        # every launch goes to the CURRENT device's stream with raw pointers: pin it (model_utils.py:45)
    model.to(dev)
    store, _ = model.engine()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        broadcast_parameters(store)
    attach_reducer(store, force=bool(os.environ.get("TUBER_FORCE_DDP")))      # None with a single rank (unless forced)
    path = getattr(cfg.CONFIG.MODEL, "PRETRAIN_TRANSFORMER_DIR", "")
    if is_tuber and path:
        from .checkpoint import load_detr_weights
        print("loading detr")
        load_detr_weights(model, path, cfg)
    return model


def build_optimizer(model, cfg):
    T = cfg.CONFIG.TRAIN
    return FusedClipAdamW(build_param_groups(model, cfg), lr=T.LR, weight_decay=T.W_DECAY, model=model)


def train_step(model, criterion, optimizer, samples, targets, max_norm, epoch=0, cfg=None):
    """One eager optimisation step.  ``optimizer``: a ``FusedClipAdamW``, or the object the reference script builds --
    ``torch.optim.AdamW(param_dicts, lr=..., weight_decay=...)`` (train_tuber_ava.py:58), adopted transparently (optim.adopt: shared
    ``param_groups``, state exposed as views) -- or any other ``torch.optim.Optimizer``, driven by the reference's literal sequence
    ``clip_grad_norm_`` + ``optimizer.step()`` on the gradient views (video_action_recognition.py:152-154).
    Returns (total loss tensor, loss dict) -- all on the device, nothing synchronised."""
    store, _ = model.engine()
    reducer = getattr(store, "reducer", None)
    fused = adopt(optimizer, model)
    outputs = model(samples)
    loss_dict = criterion(outputs, targets)
    weight_dict = criterion.weight_dict
    if cfg is not None and epoch > cfg.CONFIG.LOSS_COFS.WEIGHT_CHANGE:          # video_action_recognition.py:145-146
        weight_dict["loss_ce"] = cfg.CONFIG.LOSS_COFS.LOSS_CHANGE_COF
    losses = criterion.weighted_total(loss_dict, weight_dict)
    store.zero_grad()                      # optimizer.zero_grad() of a stock optimizer would drop the views into the flat buffer
    if reducer is not None:
        reducer.begin()
    losses.backward()
    if reducer is not None:
        reducer.finish()
    if fused is not None:
        fused.step(max_norm=max_norm if max_norm and max_norm > 0 else None)
    else:
        if max_norm and max_norm > 0:
            torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)
        optimizer.step()
    return losses.detach(), loss_dict


class _Snapshot:
    """Everything a training step mutates: parameters, BatchNorm buffers, Adam moments + step count, dropout seed.  The captured
    step needs eager warm-up passes (workspace sizing, reduce tables); they must not count as optimisation steps."""

    def __init__(self, model, store, opt):
        self.items = [(store.flat, store.flat.clone()), (store.seed, store.seed.clone()), (opt.exp_avg, opt.exp_avg.clone()),
                      (opt.exp_avg_sq, opt.exp_avg_sq.clone()), (opt.t_dev, opt.t_dev.clone())]
        self.items += [(b, b.clone()) for b in model.buffers()]

    def restore(self):
        with torch.no_grad():
            for live, saved in self.items:
                live.copy_(saved)


class CaptureFailed(RuntimeError):
    """the hipGraph capture of a training step did not complete (the eager step remains available)"""


class GraphedTrainStep:
    """The same optimisation step replayed from ONE captured hipGraph (HIP graphs instead of a tracing compiler): bf16 weight
    refresh, forward of the whole model, matching-cost kernel, Hungarian assignment on the device (tuber_lsap_device), fused
    criterion (losses + output gradients), backward of the whole model, global-norm clip + AdamW.

    ~1600 kernel launches per step are issued by the GPU front-end instead of Python, so the step is GPU-bound.

    Everything that changes from batch to batch or epoch to epoch is read from DEVICE memory by the captured kernels and
    refreshed in ``__call__`` before the replay: the clips and their padding mask, the padded [B, Tmax] targets (boxes, labels,
    counts and -- JHMDB/UCF -- key-frame positions and visibility labels), the per-group learning rate / weight decay
    (``lr_scheduler.step()`` keeps working), the loss weights (the ``epoch > WEIGHT_CHANGE`` switch), the dropout seed and the
    AdamW step count.  A new clip shape or a changed set of frozen parameters captures a new graph (the eager warm-up passes of
    a capture are rolled back, so they are not optimisation steps); at most ``max_graphs`` graphs are kept (LRU).

    With > 1 rank (ddp.py) the step is graph A (refresh .. layer4's backward) -> RCCL all-reduce, on the reducer's own stream, of the
    ~58 % of the gradient buffer that is final at that point (transformer, heads, class branch, layer4, pool decoder) -> graph A1
    (layer3's backward, a latency-bound chain that leaves HBM and most CUs to the collective) -> all-reduce of layer3's 38 % ->
    graph A2 (layer2 / layer1 / stem backward) -> all-reduce of the remainder -> graph B2 (clip + AdamW), which waits for the side
    stream by event.  ``TUBER_DDP_CUTS=3`` restores the single cut of rounds 2-5 (97 % under layer2 / layer1 / stem); with
    ``TUBER_RCCL_IN_GRAPH=1`` the collectives are captured into ONE graph as a forked branch instead.
    """

    def __init__(self, model, criterion, optimizer, max_norm, tmax=16, max_graphs=4):
        fused = adopt(optimizer, model)
        if fused is None:
            raise TypeError("GraphedTrainStep needs AdamW (FusedClipAdamW or a stock torch.optim.AdamW); got %s" % type(optimizer).__name__)
        self.model, self.criterion, self.optimizer = model, criterion, fused
        self.max_norm, self.tmax, self.max_graphs = max_norm, tmax, max_graphs
        self.graphs = collections.OrderedDict()
        self.part_marks = None

    # -- capture -------------------------------------------------------------------------------------------------------------
    def _capture(self, clips, mask, targets, tmax):
        """capture with the cyclic garbage collector parked: a collection that runs INSIDE a stream capture may destroy the hipGraph /
        device tensors of an earlier step object (reference cycles through autograd nodes keep them until the collector runs), and the
        runtime aborts the process on such a call while a capture is open (seen in the GPU suite: "Fatal Python error: Aborted",
        Garbage-collecting, under test_deferred_weight_gradient_reductions_are_bit_identical).  Collected once up front instead."""
        import gc
        gc.collect()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            return self._capture_impl(clips, mask, targets, tmax)
        finally:
            if was_enabled:
                gc.enable()

    def _capture_impl(self, clips, mask, targets, tmax):
        from .criterion import PaddedTargets
        from .misc import NestedTensor
        model, crit, opt = self.model, self.criterion, self.optimizer
        store, _ = model.engine()
        dev = store.device
        red = getattr(store, "reducer", None)
        ddp = red is not None                        # N > 1 ranks (or a forced one-rank communicator): gradients are all-reduced
        in_graph = ddp and red.comm is not None and bool(os.environ.get("TUBER_RCCL_IN_GRAPH"))
        g = type("Captured", (), {})()
        g.red, g.in_graph = red, in_graph
        g.clips = clips.clone()
        g.mask = mask.clone()
        g.pt = PaddedTargets(targets, crit.ava, crit.num_classes if crit.ava else crit.num_classes + 1, dev, tmax=tmax)
        max_norm = self.max_norm if self.max_norm and self.max_norm > 0 else None

        def head():
            outputs = model(NestedTensor(g.clips, g.mask))
            logits, logits_b, boxes = crit.stacked(outputs)
            g.logits_s, g.boxes_s = crit.select(logits, boxes, g.pt)
            g.logits_b = logits_b
            with torch.no_grad():
                g.cost = crit.matcher.cost(g.logits_s.detach().contiguous(),
                                           (logits_b if crit.ava else g.logits_s).detach().contiguous(),
                                           g.boxes_s.detach().contiguous(), g.pt)

        def tail(step):
            g.loss_dict = crit.losses_from_match(g.logits_s, g.logits_b, g.boxes_s, g.pt, g.match)
            g.loss_dict["class_error"] = crit.class_error(g.logits_s[-1], g.pt, g.match[-1])
            g.loss = crit.weighted_total(g.loss_dict)
            opt.zero_grad()
            g.loss.backward()
            if step:
                opt.step(max_norm=max_norm)

        # eager warm-up with the reducer detached (its hooks must not fire inside a stream capture, and it flushes the deferred
        # reductions per bottleneck where the captured backward flushes twice): sizes every workspace, builds the reduce tables and
        # the loss-weight / hyper-parameter device tables.  Rolled back afterwards -- a capture is not an optimisation step.
        snap = _Snapshot(model, store, opt)
        store.reducer = red if in_graph else None
        split = ddp and not in_graph and not os.environ.get("TUBER_NO_SPLIT_GRAPH")
        split = split or bool(os.environ.get("TUBER_FORCE_SPLIT_GRAPH"))
        _, runner = model.engine()
        # where the backward makes its gradient windows final (backbone.CSNRunner._backward_blocks): fixed BEFORE the warm-up passes so
        # that they and the capture issue the same weight-gradient groups and deferred-reduce tables
        runner.cut_stages = tuple(int(x) for x in os.environ.get("TUBER_DDP_CUTS", "4,3").split(",") if x) if split else (3,)
        if in_graph:
            red.dry = True                           # hooks fire (same deferred-reduce flush points as the capture), nothing is sent
        try:
            for _ in range(2):
                head()
                g.on_device = g.cost.shape[2] <= 128 and g.cost.shape[3] <= 128      # the bound of tuber_lsap_device (criterion.assign)
                g.match = crit.assign(g.cost, g.pt)
                if in_graph:
                    red.begin()
                tail(not in_graph)
                if in_graph:
                    red.finish()
                    opt.step(max_norm=max_norm)
            torch.cuda.synchronize()
        finally:
            snap.restore()
            if in_graph:
                red.dry = False
        opt.sync_hyper()
        crit.sync_weights(dev)
        g.A, g.A2, g.B1, g.B2, g.split, g.parts, g.cuts, g.flag_edges = torch.cuda.CUDAGraph(), None, None, None, None, [], [], frozenset()
        split = split and g.on_device
        own_step = not ddp or in_graph               # clip + AdamW inside the main graph (else graph B2, behind the all-reduce)
        if in_graph:
            # ONE graph: the reducer's hooks fork its side stream off the capture stream inside the backward pass, so the RCCL
            # all-reduces (+ averaging) are captured as a parallel branch that joins before clip + AdamW
            with torch.cuda.graph(g.A, capture_error_mode="relaxed"):      # RCCL may touch the runtime while enqueuing
                head()
                g.match = crit.assign(g.cost, g.pt)
                red.begin()
                tail(False)
                red.finish()
                opt.step(max_norm=max_norm)
        elif split and g.on_device:
            # DDP: the graph is CUT inside the backward pass where a gradient window becomes final, so that window's RCCL all-reduce runs
            # on the reducer's stream under the backward of the stages below it.  Default cuts (TUBER_DDP_CUTS=4,3): (1) where layer4's
            # backward ends -- transformer + heads + class branch + layer4 + pool decoder, ~58 % of the bytes, go under layer3's
            # latency-bound chain, which leaves HBM and most CUs idle; (2) where layer3's backward ends -- layer3's 19 M parameters go
            # under layer2 / layer1 / stem; the ~1.5 M tail after the last graph.  (Until round 5 the only cut was (2): the whole 97 % ran
            # beside the bandwidth-bound layer2 / layer1 backward and cost a rank 0.77 ms of contention with nothing on the wire.)
            cut = []
            graphs = [g.A]
            # issue points ordered by a device-memory counter instead of an event (ddp.FlatGradReducer.flag_points): by default the FIRST
            flag_edges = g.flag_edges = red.flag_points() if red is not None else frozenset()
            torch.cuda.synchronize()
            cs = torch.cuda.Stream()
            cs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cs):
                g.A.capture_begin(capture_error_mode="relaxed")      # the cuts happen on autograd's worker thread

                def hook(off):
                    if len(cut) in flag_edges:
                        red.signal(len(cut))                         # last node of this part: the windows above are final
                    graphs[-1].capture_end()
                    cut.append(int(off))
                    nxt = torch.cuda.CUDAGraph()
                    nxt.capture_begin(pool=g.A.pool(), capture_error_mode="relaxed")
                    graphs.append(nxt)
                runner.split_hook = hook
                try:
                    head()
                    g.match = crit.assign(g.cost, g.pt)
                    tail(own_step)
                finally:
                    runner.split_hook = None
                if len(cut) in flag_edges and not own_step and cut:  # (no cut -- a frozen body: one event-ordered window after the graph)
                    red.signal(len(cut))                             # end of backward: the remainder is final
                graphs[-1].capture_end()
                g.parts, g.cuts, g.body_begin = graphs[1:], cut, int(runner.body_begin)
                g.A2 = graphs[-1] if cut else None
                g.split = cut[-1] if cut else None
            torch.cuda.current_stream().wait_stream(cs)
        elif g.on_device:
            with torch.cuda.graph(g.A):
                head()
                g.match = crit.assign(g.cost, g.pt)
                tail(own_step)
        else:
            # assignment problems beyond the device solver's 128 x 128 bound: graph A / host tuber_lsap / graph B1
            with torch.cuda.graph(g.A):
                head()
            L, B = g.cost.shape[:2]
            g.match = torch.full((L, B, g.pt.tmax), -1, dtype=torch.int32, device=dev)
            g.match_host = torch.full((L, B, g.pt.tmax), -1, dtype=torch.int32).pin_memory()
            g.cost_host = torch.empty(g.cost.shape, dtype=torch.float32).pin_memory()
            g.B1 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g.B1, pool=g.A.pool()):
                tail(own_step)
        store.reducer = red
        if not own_step:
            g.B2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g.B2, pool=g.A.pool()):
                opt.step(max_norm=max_norm)
        g.ranges = store.trainable_ranges()
        return g

    def input_buffers(self, clips_shape):
        """(clips, mask) buffers the captured step for this clip shape reads, or None before its first call: a producer that fills them in
        place and passes them back to ``__call__`` saves the step its 67 MB device-to-device copy."""
        store, _ = self.model.engine()
        for k, g in self.graphs.items():
            if k[:3] == (tuple(clips_shape), store.trainable_signature(), self.criterion.training) and k[4] == store.coop_off:
                return g.clips, g.mask
        return None

    # -- replay --------------------------------------------------------------------------------------------------------------
    def __call__(self, samples, targets):
        """``samples``: NestedTensor (clips + padding mask) or a plain [B,3,T,H,W] tensor (no padding)."""
        store, _ = self.model.engine()
        if hasattr(samples, "tensors"):
            clips, mask = samples.tensors, samples.mask
        else:
            clips, mask = samples, None
        if mask is None:
            mask = torch.zeros(clips.shape[0], clips.shape[-2], clips.shape[-1], dtype=torch.bool, device=store.device)
        # the padded [B, Tmax] target layout is baked into the capture: a batch with more boxes per clip than any before (dense
        # AVA key frames) gets a graph with a wider layout instead of an error -- the reference has no such limit
        need = max([int(t["boxes"].shape[0]) for t in targets] + [1])
        tmax = max(self.tmax, (need + 15) // 16 * 16)
        for k in self.graphs:                                     # a captured wider layout serves narrower batches too
            if k[:3] == (tuple(clips.shape), store.trainable_signature(), self.criterion.training) and k[3] >= tmax and k[4] == store.coop_off:
                tmax = k[3]
                break
        # (a captured step bakes the decoder's launch form in: after a timed-out cooperative launch -- engine.coop_failed -- a new one is captured)
        key = (tuple(clips.shape), store.trainable_signature(), self.criterion.training, tmax, store.coop_off)
        g = self.graphs.get(key)
        if g is None:
            while len(self.graphs) >= self.max_graphs:            # LRU: a graph holds its own memory pool
                self.graphs.popitem(last=False)
            try:
                g = self._capture(clips.to(store.device, torch.float32), mask.to(store.device), targets, tmax)
            except (RuntimeError, ValueError) as e:
                raise CaptureFailed("%s: %s" % (type(e).__name__, e)) from e
            self.graphs[key] = g
        else:
            self.graphs.move_to_end(key)
        # a producer that writes straight into the captured buffers (``input_buffers``: the input pre-pass, bench.py's resident synthetic
        # clips) hands those very tensors back: nothing to copy then
        if clips.data_ptr() != g.clips.data_ptr():
            g.clips.copy_(clips, non_blocking=True)
        if mask.data_ptr() != g.mask.data_ptr():
            g.mask.copy_(mask, non_blocking=True)
        g.pt.refill(targets)
        self.optimizer.sync_hyper()
        self.optimizer.mark_stepped()
        self.criterion.sync_weights(store.device)
        sizes = g.pt.sizes
        red = g.red
        if red is not None and not g.in_graph:
            red.begin()
        marks = self.part_marks                               # bench.py: HIP-event stamps between the graph parts of the measured steps
        if marks is not None:
            marks.append([])
            self._mark()
        g.A.replay()
        if marks is not None:
            self._mark()
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
        if g.parts:
            # final at cut i: the stage that just ended, the stages above it, everything laid out behind the body [cuts[i], previous cut)
            # and -- at the first cut -- everything laid out before the body (transformer, heads: [0, body_begin)); pending after the
            # last cut: [body_begin, cuts[-1]) (stem, layer1, layer2)
            hi = store.total
            for i, part in enumerate(g.parts):
                if red is not None:
                    soft = i in g.flag_edges                  # ordered by the counter the graph part bumps, not by an event (csrc/stream_flag.hip)
                    if soft:
                        red.wait_for(i)
                    red.reduce(g.cuts[i], hi, edge=not soft)
                    if i == 0:
                        red.reduce(0, g.body_begin, edge=False)
                hi = g.cuts[i]
                part.replay()                                 # the backward of the stages below, under the all-reduce
                if marks is not None:
                    self._mark()
            if red is not None:
                soft = len(g.parts) in g.flag_edges
                if soft:
                    red.wait_for(len(g.parts))
                red.reduce(g.body_begin, hi, edge=not soft)
        elif red is not None and not g.in_graph:
            red.reduce(0, store.total)
        if red is not None and not g.in_graph:
            red.finish(rest=False)                            # the optimizer graph waits for the side stream (events only)
        if g.B2 is not None:
            g.B2.replay()
            if marks is not None:
                self._mark()
        return g.loss, g.loss_dict

    def _mark(self):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.part_marks[-1].append(e)

    def part_ms(self):
        """mean duration (ms) of each graph part over the steps stamped since ``part_marks = []`` (syncs)"""
        torch.cuda.synchronize()
        rows = [[a.elapsed_time(b) for a, b in zip(m[:-1], m[1:])] for m in (self.part_marks or []) if len(m) > 1]
        self.part_marks = None
        if not rows:
            return None
        n = min(len(r) for r in rows)
        return [round(sum(r[i] for r in rows) / len(rows), 3) for i in range(n)]


class _DeviceMeters:
    """The reference's AverageMeters (class_err, losses_avg, losses_box, losses_giou, losses_ce, losses_ce_b:
    video_action_recognition.py:97-104,182-193), accumulated ON THE DEVICE -- every iteration contributes (weight = len(targets),
    like ``meter.update(value, len(targets))``), the host reads the six running averages only when it prints."""
    KEYS = ("class_error", "loss", "loss_bbox", "loss_giou", "loss_ce", "loss_ce_b")

    def __init__(self, device):
        self.sum = torch.zeros(len(self.KEYS), dtype=torch.float64, device=device)
        self.count = 0

    def update(self, loss, loss_dict, n):
        zero = loss.new_zeros(())
        vals = [loss if k == "loss" else loss_dict.get(k, zero) for k in self.KEYS]
        vals = [v.detach().reshape(()).to(loss.dtype) if torch.is_tensor(v) else loss.new_tensor(float(v)) for v in vals]
        self.sum.add_(torch.stack(vals).double(), alpha=float(n))
        self.count += n

    def averages(self):
        """{key: running average} -- ONE device->host read"""
        host = (self.sum / max(self.count, 1)).tolist()
        return dict(zip(self.KEYS, host))


def _graphed_for(model, criterion, optimizer, max_norm):
    """the cached GraphedTrainStep of (model, criterion, optimizer, max_norm), or None when the optimizer is not AdamW"""
    fused = adopt(optimizer, model)
    if fused is None:
        return None
    cache = model.__dict__.setdefault("_tuber_graphed", {})
    key = (id(criterion), id(fused), float(max_norm or 0.0))
    g = cache.get(key)
    if g is None:
        cache.clear()                       # one live training configuration per model: a graph owns its memory pool
        g = cache[key] = GraphedTrainStep(model, criterion, fused, max_norm)
    return g


def train_tuber_detection(cfg, model, criterion, data_loader, optimizer, epoch, max_norm, lr_scheduler=None, writer=None,
                          graphed=None, print_freq=10):
    """One training epoch -- the reference's ``train_tuber_detection`` (utils/video_action_recognition.py:64-220), same argument
    list, on the HIP path.  ``optimizer`` is whatever the script built: the stock ``torch.optim.AdamW(param_dicts, ...)`` of
    train_tuber_ava.py:58 is adopted (optim.adopt), ``lr_scheduler`` stays bound to that object.

    Every batch is one replay of the captured hipGraph step (``GraphedTrainStep``, created on first use and cached on the model;
    ``graphed=False`` or ``TUBER_AB=eager_step`` forces the eager ``train_step``, which is also the fallback when a capture fails or
    the optimizer is not AdamW).  The six scalars the reference logs -- ``train/{class_error,totall_loss,loss_bbox,loss_giou,
    loss_ce,loss_ce_b}`` (:215-220), running averages weighted by ``len(targets)`` -- are accumulated on the device for EVERY
    iteration and read back every ``print_freq`` iterations, so the GPU queue stays full; a non-finite loss stops training like
    the reference (:195-198).  ``epoch > LOSS_COFS.WEIGHT_CHANGE`` switches ``loss_ce``'s weight (:145-146)."""
    import time
    model.train()
    criterion.train()
    dev = next(model.parameters()).device
    rank = getattr(cfg.DDP_CONFIG, "GPU_WORLD_RANK", 0)
    if epoch > cfg.CONFIG.LOSS_COFS.WEIGHT_CHANGE:
        criterion.weight_dict["loss_ce"] = cfg.CONFIG.LOSS_COFS.LOSS_CHANGE_COF
    if graphed is None and not ab.on("eager_step"):
        graphed = _graphed_for(model, criterion, optimizer, max_norm)
    elif graphed is False or graphed is None:
        graphed = None
    meters = _DeviceMeters(dev)
    end = time.time()
    loss = None
    store = model.engine()[0]
    tracker = _StepTracker(dev)
    import torch.distributed as _dist
    world = _dist.get_world_size() if _dist.is_available() and _dist.is_initialized() else 1
    n_iter = len(data_loader)
    for idx, data in enumerate(data_loader):
        samples, targets = data[0], data[1]
        samples = samples.to(dev)            # reference :121; for an input_pipeline.ClipBatch this IS the HIP pre-pass (uint8 frames -> fp32 batch)
        targets = [{k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in t.items() if k != "image_id"} for t in targets]
        if graphed is not None:
            try:
                loss, loss_dict = graphed(samples, targets)       # NestedTensor: clips AND padding mask go to the captured buffers
            except CaptureFailed as e:                            # a failed capture must not end the epoch
                if getattr(model.engine()[0], "reducer", None) is not None:
                    raise                                         # N > 1: every rank must stay on the same collective sequence
                import sys
                print("[tuber] hipGraph step unavailable (%s: %s); continuing with the eager step" % (type(e).__name__, e), file=sys.stderr, flush=True)
                graphed = None
                model.__dict__.pop("_tuber_graphed", None)
                loss, loss_dict = train_step(model, criterion, optimizer, samples, targets, max_norm, epoch=epoch, cfg=cfg)
        else:
            loss, loss_dict = train_step(model, criterion, optimizer, samples, targets, max_norm, epoch=epoch, cfg=cfg)
        if lr_scheduler is not None and cfg.CONFIG.TRAIN.LR_POLICY == "cosine":
            lr_scheduler.step_update(epoch * n_iter + idx)
        if rank == 0:
            meters.update(loss, loss_dict, len(targets))
        # non-finite loss (reference :195-198 checks the rank-reduced loss on every rank, every iteration, BEFORE optimizer.step()): here the
        # optimizer's device-side guard skips the update of any step whose gradient norm is not finite (csrc/optim.hip), so nothing invalid is
        # ever applied, and the host learns about it from a device-side tracker updated on EVERY rank every step (no host-to-device copy, no
        # sync), MAX-reduced over the ranks and read on every rank, so that all ranks stop together instead of rank 0 raising while the
        # others wait in the next gradient all-reduce.  The collective is issued at iterations every rank reaches by
        # construction (idx % print_freq == 0 -- the same idx on every rank as long as the loaders have one length, which the gradient
        # all-reduce needs anyway) and once more after the loop; NOT at ``idx + 1 == n_iter``, which would pair with nothing on a rank whose
        # loader is longer.  The same tracker carries the cooperative decoder launch's error word: a timed-out
        # launch poisons its step with NaN (skipped like any other non-finite step); when ANY rank reports one, EVERY rank switches to
        # the launch chain at this same iteration and training continues.
        tracker.update(loss, store.coop_sync)
        if idx % print_freq == 0:
            _check_tracker(tracker, store, loss_dict, epoch, idx, rank, world, _dist)      # (a captured step is keyed on store.coop_off: re-captured by itself)
        if rank == 0 and (idx % print_freq == 0 or idx + 1 == n_iter):
            avg = meters.averages()
            lr = optimizer.param_groups[-1]["lr"]
            print("Epoch: [%d][%d/%d]" % (epoch, idx + 1, n_iter))
            print("lr: ", lr)
            print("batch time: %.3f" % (time.time() - end))
            print("class_error: {class_error:.3f}, loss: {loss:.3f}, loss_bbox: {loss_bbox:.3f}, loss_giou: {loss_giou:.3f}, "
                  "loss_ce: {loss_ce:.3f}, loss_ce_b: {loss_ce_b:.3f}".format(**avg))
            if writer is not None:
                it = idx + epoch * n_iter
                writer.add_scalar("train/class_error", avg["class_error"], it)
                writer.add_scalar("train/totall_loss", avg["loss"], it)          # (sic) the reference's tag, :216
                writer.add_scalar("train/loss_bbox", avg["loss_bbox"], it)
                writer.add_scalar("train/loss_giou", avg["loss_giou"], it)
                writer.add_scalar("train/loss_ce", avg["loss_ce"], it)
                writer.add_scalar("train/loss_ce_b", avg["loss_ce_b"], it)
        end = time.time()
    if loss is not None:
        _check_tracker(tracker, store, loss_dict, epoch, n_iter - 1, rank, world, _dist)
    return loss


class _StepTracker:
    """Device-side record of what went wrong since the last check: [a non-finite loss was seen, -(first such iteration + 1), the cooperative
    decoder launch's error word].  The iteration is stored NEGATED (initially -inf) so that the MAX all-reduce over the ranks yields the
    FIRST offending iteration of any rank; the counter lives on the device (no per-step host-to-device copy)."""

    def __init__(self, device):
        self.device = device
        self.reset()

    def reset(self):
        self.state = torch.tensor([0.0, float("-inf"), 0.0], dtype=torch.float32, device=self.device)
        self.it = torch.zeros((), dtype=torch.float32, device=self.device)

    def update(self, loss, coop_sync):
        bad = (~torch.isfinite(loss.detach())).reshape(()).to(torch.float32)
        self.it = self.it + 1.0
        first = torch.where((self.state[0] == 0) & (bad > 0), -self.it, self.state[1])
        self.state = torch.stack([torch.maximum(self.state[0], bad), first, torch.maximum(self.state[2], coop_sync[2].to(torch.float32))])


def _check_tracker(tracker, store, loss_dict, epoch, idx, rank, world, _dist):
    """MAX-reduce the tracker over the ranks (the only host sync of the loop).  A timed-out cooperative decoder launch on any rank: every
    rank clears its words, switches to the launch chain and goes on (returns True; the poisoned steps were skipped by the optimizer on
    every rank -- the NaN gradients reached the others through the all-reduce).  Otherwise a non-finite loss stops every rank together."""
    state = tracker.state
    if world > 1:
        _dist.all_reduce(state, op=_dist.ReduceOp.MAX)
    flag, negfirst, coop = state.tolist()
    if coop:
        if not store.coop_failed():          # another rank's launch failed: same switch here, so that all ranks run the same launch sequence
            store.coop_sync.zero_()
            store.coop_off = True
        tracker.reset()
        return True
    if flag:
        losses = {k: float(v.detach()) if torch.is_tensor(v) else v for k, v in loss_dict.items()}
        print("Loss is non-finite on some rank, stopping training")
        print(losses)
        raise FloatingPointError("non-finite loss first seen at epoch %d, iteration %d (checked at iteration %d, rank %d of %d; this rank's "
                                 "last loss terms: %s); steps with a non-finite gradient norm were skipped by the optimizer, the weights hold "
                                 "the last finite update" % (epoch, int(-negfirst) - 1, idx, rank, world, losses))
    return False
