"""Fused clip_grad_norm_ + AdamW over the ParamStore's flat buffers.

Mirrors what the reference's step does (utils/video_action_recognition.py:150-154 with the parameter groups of
train_tuber_ava.py:41-58): ``clip_grad_norm_(model.parameters(), max_norm)`` then ``AdamW.step()`` -- as three
kernel launches per learning-rate segment instead of ~5 passes over 684 tensors.
Constructed from the SAME ``param_dicts`` list the reference builds, so scripts only swap the class name.
"""
import math

import torch

from . import lib


class FusedClipAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, model=None):
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        if model is None:
            raise ValueError("FusedClipAdamW needs model= (the tubelet_transformer_amd DETR, bare or wrapped)")
        self.model = model.module if hasattr(model, "module") else model
        self.store, _ = self.model.engine()
        st = self.store
        dev = st.device
        self.exp_avg = torch.zeros_like(st.flat)
        self.exp_avg_sq = torch.zeros_like(st.flat)
        self.partial = torch.zeros(1024, dtype=torch.float32, device=dev)
        self.norm_out = torch.zeros(2, dtype=torch.float32, device=dev)
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=dev)      # step count read by the kernel (graph-replay safe)
        # per-group (lr, weight_decay) in device memory, refreshed from param_groups before every step / graph replay, so
        # lr_scheduler.step() / step_update() act on a captured hipGraph too
        self.hyper = torch.zeros(len(self.param_groups), 2, dtype=torch.float32, device=dev)
        self._hyper_host = None
        self._hyper_pin = None          # pinned staging of the table: per-iteration schedulers change lr every step, no blocking copy
        # flat segments with uniform hyper-parameters: walk the store in layout order
        by_ptr = {}
        for gi, group in enumerate(self.param_groups):
            for p in group["params"]:
                by_ptr[p.data_ptr()] = gi
        segs = []
        for n, p in zip(st.names, st.params):
            gi = by_ptr.get(p.data_ptr())
            used = gi is not None and p.requires_grad and not n.startswith("backbone.body.out_fc")
            o = st.offsets[n]
            end = o + (p.numel() + 63) // 64 * 64
            key = gi if used else None
            if segs and segs[-1][2] == key and segs[-1][1] == o:
                segs[-1][1] = end
            else:
                segs.append([o, end, key])
        self.segments = [s for s in segs if s[2] is not None]

    @property
    def t(self):
        """number of steps taken (host copy of the device counter; syncs)."""
        return int(self.t_dev.item())

    def sync_hyper(self):
        """push param_groups' lr / weight_decay to the device table if they changed (one tiny async copy)."""
        host = [(float(g["lr"]), float(g["weight_decay"])) for g in self.param_groups]
        if host != self._hyper_host:
            if self.hyper.is_cuda:
                if self._hyper_pin is None:
                    self._hyper_pin = [torch.empty(len(host), 2, dtype=torch.float32).pin_memory() for _ in range(4)]
                    self._hyper_ev = [None] * len(self._hyper_pin)
                    self._hyper_k = 0
                # a ring of pinned tables: the async copy of step k may still be in flight when step k+1 fills the next one; an
                # entry is reused only after its copy has executed (the host waits only if it is > 4 lr changes ahead of the GPU)
                self._hyper_k = (self._hyper_k + 1) % len(self._hyper_pin)
                pin, ev = self._hyper_pin[self._hyper_k], self._hyper_ev[self._hyper_k]
                if ev is not None:
                    ev.synchronize()
                pin.copy_(torch.tensor(host, dtype=torch.float32).view(-1, 2))
                self.hyper.copy_(pin, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._hyper_ev[self._hyper_k] = ev
            else:
                self.hyper.copy_(torch.tensor(host, dtype=torch.float32).view(-1, 2))
            self._hyper_host = host

    # -- checkpointing: the moments live in flat buffers outside Optimizer.state -----------------------------------------------
    def state_dict(self):
        sd = super().state_dict()
        sd["flat"] = {"exp_avg": self.exp_avg.detach().cpu(), "exp_avg_sq": self.exp_avg_sq.detach().cpu(), "step": self.t,
                      "names": list(self.store.names), "offsets": [self.store.offsets[n] for n in self.store.names]}
        return sd

    def load_state_dict(self, sd):
        sd = dict(sd)
        flat = sd.pop("flat", None)
        super().load_state_dict(sd)
        self._hyper_host = None
        if flat is not None:
            if list(flat["names"]) != list(self.store.names) or tuple(flat["exp_avg"].shape) != tuple(self.exp_avg.shape):
                raise ValueError("optimizer state was saved for a different parameter layout")
            self.exp_avg.copy_(flat["exp_avg"])
            self.exp_avg_sq.copy_(flat["exp_avg_sq"])
            self.t_dev.fill_(int(flat["step"]))

    @torch.no_grad()
    def zero_grad(self, set_to_none=False):
        self.store.zero_grad()

    def mark_stepped(self):
        """an adopted stock optimizer never has its own step() called: tell torch's lr_scheduler bookkeeping that a step happened
        (it warns about 'lr_scheduler.step() before optimizer.step()' otherwise)"""
        stock = getattr(self, "_stock", None)
        if stock is not None:
            stock._opt_called = True

    @torch.no_grad()
    def grad_norm(self, max_norm=0.0, advance=False):
        """device-side total gradient norm (float tensor [2] = norm, clip coefficient; coefficient -1 = non-finite norm, the AdamW launches
        skip their update); no host sync.  ``advance``: the device step counter moves on by one iff the norm is finite."""
        lib.call("tuber_grad_norm_clip_coef", self.store.gflat, self.store.total, float(max_norm), self.partial, self.norm_out,
                 self.t_dev if advance else None)
        return self.norm_out

    @torch.no_grad()
    def step(self, closure=None, max_norm=None):
        """AdamW step; with ``max_norm`` the global-norm clipping is fused in (else call clip_grad_norm_ yourself).  A step whose
        gradient buffer holds a NaN / Inf is SKIPPED on the device (parameters, moments and the step count stay as they were).
        The norm runs over the whole flat gradient buffer: windows of frozen parameters are never written and stay zero, so it equals
        ``clip_grad_norm_`` over the parameters that have a gradient (video_action_recognition.py:153)."""
        st = self.store
        self.mark_stepped()
        # the norm pass runs with or without clipping: it is also the guard that keeps a non-finite gradient (NaN loss, a poisoned
        # cooperative-decoder launch, a peer's NaN through the all-reduce) from ever being applied -- coefficient -1, step count unchanged
        clip = self.grad_norm(max_norm if max_norm is not None and max_norm > 0 else 0.0, advance=True)
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        for o, end, gi in self.segments:
            g = self.param_groups[gi]
            b1, b2 = g["betas"]
            f = st.flat.data_ptr() + 4 * o
            lib.call("tuber_adamw_segment", f, st.gflat.data_ptr() + 4 * o, self.exp_avg.data_ptr() + 4 * o,
                     self.exp_avg_sq.data_ptr() + 4 * o, end - o, clip, float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                     float(g["weight_decay"]), self.t_dev, 0, self.hyper.data_ptr() + 8 * gi)
        return None


def build_param_groups(model, cfg):
    """The four groups of train_tuber_ava.py:41-55 (selected by substring of the parameter name)."""
    named = list(model.named_parameters())
    T = cfg.CONFIG.TRAIN
    return [
        {"params": [p for n, p in named if "backbone" not in n and "class_embed" not in n and "query_embed" not in n and p.requires_grad]},
        {"params": [p for n, p in named if "backbone" in n and p.requires_grad], "lr": T.LR_BACKBONE},
        {"params": [p for n, p in named if "class_embed" in n and p.requires_grad], "lr": T.LR},
        {"params": [p for n, p in named if "query_embed" in n and p.requires_grad], "lr": T.LR},
    ]


# ------------------------------------------------------------------------------------------------------------------------------
# the reference's own optimizer object: torch.optim.AdamW(param_dicts, lr=..., weight_decay=...) (train_tuber_ava.py:41-58)
# ------------------------------------------------------------------------------------------------------------------------------
def _is_plain_adamw(opt):
    if type(opt).__name__ != "AdamW" or not isinstance(opt, torch.optim.Optimizer):
        return False
    for g in opt.param_groups:
        if g.get("amsgrad") or g.get("maximize") or g.get("decoupled_weight_decay") is False:
            return False
    return True


def adopt(optimizer, model):
    """The optimizer the training loop should drive, given what the SCRIPT built.

    * a ``FusedClipAdamW``: itself;
    * a stock ``torch.optim.AdamW`` over (a subset of) the model's parameters -- what ``train_tuber_ava.py:58`` constructs: a
      ``FusedClipAdamW`` that SHARES its ``param_groups`` (the same list of the same dicts, so ``MultiStepLR`` /
      ``build_scheduler`` objects bound to the user's optimizer keep steering the fused step, on a captured hipGraph too) and whose
      flat moment buffers are exposed as ``optimizer.state[p]['exp_avg' | 'exp_avg_sq' | 'step']`` VIEWS, so
      ``optimizer.state_dict()`` (``save_checkpoint``, utils/model_utils.py:118-134) stores the live AdamW state and
      ``optimizer.load_state_dict()`` (resume) flows back into the flat buffers.  Cached on the optimizer object;
    * anything else (SGD, amsgrad ...): ``None`` -- the caller falls back to ``clip_grad_norm_`` + ``optimizer.step()`` on the
      gradient views (the reference's literal sequence, video_action_recognition.py:152-154).
    """
    if isinstance(optimizer, FusedClipAdamW):
        return optimizer
    fused = getattr(optimizer, "_tuber_fused", None)
    if fused is not None:
        return fused
    if not _is_plain_adamw(optimizer):
        return None
    model = model.module if hasattr(model, "module") else model
    groups = optimizer.param_groups
    g0 = groups[0]
    fused = FusedClipAdamW([{k: v for k, v in g.items()} for g in groups], lr=g0["lr"], betas=g0["betas"], eps=g0["eps"],
                           weight_decay=g0["weight_decay"], model=model)
    fused.param_groups = groups                    # shared, not copied: lr_scheduler.step() on the user's optimizer acts here
    fused._hyper_host = None
    st = fused.store
    import_state = bool(optimizer.state)

    by_ptr = {q.data_ptr(): n for n, q in zip(st.names, st.params)}

    def bind(load):
        """(re)bind optimizer.state to views of the flat moments; ``load``: first copy what the state holds (resume) into them"""
        step = None
        for g in optimizer.param_groups:           # (looked up at call time: load_state_dict() REPLACES optimizer.param_groups)
            for p in g["params"]:
                n = by_ptr.get(p.data_ptr())
                if n is None:
                    raise ValueError("optimizer holds a parameter that is not part of the model's ParamStore")
                o = st.offsets[n]
                ea = fused.exp_avg[o:o + p.numel()].view(p.shape)
                es = fused.exp_avg_sq[o:o + p.numel()].view(p.shape)
                old = optimizer.state.get(p)
                if load and old and "exp_avg" in old:
                    with torch.no_grad():
                        ea.copy_(old["exp_avg"])
                        es.copy_(old["exp_avg_sq"])
                    step = int(float(old["step"])) if step is None else step
                optimizer.state[p] = {"step": torch.tensor(float(fused._host_t)), "exp_avg": ea, "exp_avg_sq": es}
        if load and step is not None:
            fused.t_dev.fill_(step)
            fused._host_t = step
            for s in optimizer.state.values():
                s["step"] = torch.tensor(float(step))

    fused._host_t = 0
    bind(import_state)

    def pre_state_dict(opt):                       # optimizer.state_dict(): one device->host read of the step counter, only here
        t = float(fused.t)
        for s in opt.state.values():
            s["step"] = torch.tensor(t)

    def post_load(opt):                            # optimizer.load_state_dict(): loaded tensors replaced the views -> copy in, re-bind
        # torch's load_state_dict() installs NEW group dicts in optimizer.param_groups: share those again, otherwise
        # lr_scheduler.step() after a resume would write dicts the fused step no longer reads
        fused.param_groups = opt.param_groups
        bind(True)
        fused._hyper_host = None

    optimizer.register_state_dict_pre_hook(pre_state_dict)
    optimizer.register_load_state_dict_post_hook(post_load)
    optimizer._tuber_fused = fused
    fused._stock = optimizer          # (the stock object outlives the fused one: it owns it)
    return fused
