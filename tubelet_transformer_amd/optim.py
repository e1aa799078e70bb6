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
            self.hyper.copy_(torch.tensor(host, dtype=torch.float32).view(-1, 2), non_blocking=False)
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

    @torch.no_grad()
    def grad_norm(self, max_norm=0.0):
        """device-side total gradient norm (float tensor [2] = norm, clip coefficient); no host sync."""
        lib.call("tuber_grad_norm_clip_coef", self.store.gflat, self.store.total, float(max_norm), self.partial, self.norm_out)
        return self.norm_out

    @torch.no_grad()
    def step(self, closure=None, max_norm=None):
        """AdamW step; with ``max_norm`` the global-norm clipping is fused in (else call clip_grad_norm_ yourself).
        The norm runs over the whole flat gradient buffer: windows of frozen parameters are never written and stay zero, so it equals
        ``clip_grad_norm_`` over the parameters that have a gradient (video_action_recognition.py:153)."""
        st = self.store
        clip = None
        if max_norm is not None and max_norm > 0:
            clip = self.grad_norm(max_norm)
        if not torch.cuda.is_current_stream_capturing():
            self.sync_hyper()
        self.t_dev.add_(1)
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
