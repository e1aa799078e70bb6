"""Device-side parameter store of the TubeR MI355X path.

Layout in HBM (per model, per GPU):
  * ``flat``   fp32 master parameters, one contiguous buffer; every ``nn.Parameter`` of the model is a
               view into it, so ``state_dict()`` keeps the reference's names/shapes;
  * ``gflat``  fp32 gradients, same offsets; ``p.grad`` are views.  The backward kernels accumulate
               straight into these slices, the optimizer / gradient all-reduce work on the flat buffer;
  * ``shadow`` bf16 copy of ``flat`` (one cast launch per step) -- the B operands of the forward GEMMs;
  * ``tshadow`` bf16 TRANSPOSED copies of every GEMM weight (one batched launch per step) -- the B
               operands of the data-gradient GEMMs; leading dimension padded to a multiple of 64.
Offsets are multiples of 64 elements so every bf16 row is 16-byte aligned.
"""
import ctypes
import os

import numpy as np
import torch

from . import ab, lib

ALIGN = 64


def _ceil(x, m):
    return (x + m - 1) // m * m


class TnArgs(ctypes.Structure):
    """one entry of tuber_gemm_tn_group (struct TuberGemmTNArgs in csrc/gemm.hip)"""
    _fields_ = [("G", ctypes.c_void_p), ("ldg", ctypes.c_long), ("A", ctypes.c_void_p), ("lda", ctypes.c_long),
                ("partial", ctypes.c_void_p), ("out", ctypes.c_void_p)] + \
               [(k, ctypes.c_int) for k in ("accumulate", "M", "N", "K", "amode", "gather", "To", "Ho", "Wo", "Ti", "Hi", "Wi", "st", "ss")] + \
               [("a_scale", ctypes.c_void_p), ("a_shift", ctypes.c_void_p), ("bias_grad", ctypes.c_void_p), ("A2", ctypes.c_void_p),
                ("lda2", ctypes.c_long)]


class WgradQueue:
    """Weight-gradient GEMMs feed nothing until the optimizer: the backbone and the tape queue them here (operand tensors kept
    alive) and ``flush`` launches up to tuber_gemm_tn_group_max() = 16 of them in ONE tuber_gemm_tn_group launch -- fewer launch gaps, and for the short-M
    layer3 / layer4 / transformer shapes enough independent workgroups in flight to fill 256 CUs.  Flushed before every deferred
    second-stage reduction (DeferredReduce.flush), so gradient windows are complete wherever the old per-call launches had them."""

    def __init__(self, store):
        self.store, self.q = store, []
        # (measured and rejected: running the grouped launches on a second HIP stream next to the data-gradient chain -- ~35 fork /
        #  join points per step inside the hipGraph cost +1.9 ms/step, 18.65 -> 20.53: cross-queue edges serialise the replay)
        self.max = lib.query("tuber_gemm_tn_group_max")
        if lib.query("tuber_gemm_tn_args_bytes") != ctypes.sizeof(TnArgs):
            raise RuntimeError("TuberGemmTNArgs layout drift between engine.py and libtuber_hip.so")

    @property
    def enabled(self):
        return not ab.on("no_wgrad_groups")      # A/B switch: one tuber_gemm_tn launch per weight gradient

    @staticmethod
    def eligible(M, N, K, ldg, lda):
        """shapes the transpose-read kernel takes (64 x 64 output tiles, 8-element aligned)"""
        return not ((N | K | ldg | lda) & 7) and lib.query("tuber_gemm_tn_fuses_bias", M, N, K, ldg, lda) != 0

    def add(self, args, keep, defers):
        """args: TnArgs; keep: tensors that must outlive the launch; defers: DeferredReduce.add argument tuples registered at flush"""
        self.q.append((args, keep, defers))
        if len(self.q) >= self.max:
            self.flush()

    def flush(self):
        q, self.q = self.q, []
        if not q:
            return
        arr = (TnArgs * len(q))(*[e[0] for e in q])
        lib.call("tuber_gemm_tn_group", arr, len(q))
        for _, _, defers in q:
            for d in defers:
                self.store.defer.add(*d)


class DeferredReduce:
    """Second-stage reductions of the weight-gradient kernels, batched into ONE launch per backward pass.

    The dW GEMMs, depthwise weight gradients, LayerNorm and bias gradients leave per-workgroup fp32 partials; reducing each right
    away costs ~240 five-microsecond launches per step.  With ``accumulate = 2`` their launchers skip that stage; the partials stay
    in this arena (bump-allocated, same addresses every step, so it is hipGraph-safe) and ``flush()`` reduces all of them with
    ``tuber_multi_reduce`` -- same summation order as the immediate kernels, bit-identical gradients.  ``TUBER_AB=immediate_reduce``
    restores the per-call reductions."""
    CHUNK = 64 << 20             # floats per arena chunk (256 MB)
    _ENTRY = np.dtype([("P", "<u8"), ("out", "<u8"), ("n", "<i8"), ("stride", "<i8"), ("S", "<i4"), ("mode", "<i4"), ("C", "<i4"), ("next", "<i4")])

    def __init__(self, device):
        self.device = device
        self.chunks, self.ci, self.off = [], 0, 0
        self.entries, self.outs, self.heads, self.cache = [], {}, [], {}
        self.pre_flush = None
        # gradient windows that are known to be ZERO: filled by ParamStore.zero_grad (every window), emptied as windows are reduced into.  The first
        # reduction into such a window does not read it (mode bit 3 of tuber_multi_reduce: out = sum instead of out += sum; round 6)
        self.fresh = None                # None: unknown (something else may have written the buffer) -> always accumulate; else the set of outs written since zero_grad
        self.resolve = None              # (address, n) -> the tensor slice behind it, for the debug check
        self.check_zero = bool(os.environ.get("TUBER_CHECK_DEFER_ZERO"))     # debug: assert that a window taken as zero IS zero (eager steps only)

    @property
    def enabled(self):
        return not ab.on("immediate_reduce")

    def reset(self):
        """start of a step: the arena is reused from its first byte (nothing may be pending)."""
        if self.entries:
            self.flush()
        self.ci, self.off = 0, 0

    def alloc(self, n):
        """device address of n fp32 of scratch that stays untouched until the next ``reset``."""
        n = _ceil(int(n), ALIGN)
        while True:
            if self.ci < len(self.chunks):
                c = self.chunks[self.ci]
                if self.off + n <= c.numel():
                    ptr = c.data_ptr() + 4 * self.off
                    self.off += n
                    return ptr
                self.ci, self.off = self.ci + 1, 0
                continue
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("deferred-reduce arena would grow during hipGraph capture (run an eager warm-up step first)")
            self.chunks.append(torch.empty(max(self.CHUNK, n), dtype=torch.float32, device=self.device))

    def add(self, part, out, n, stride, S, mode, C=0):
        """out[j] += sum_{s<S} part[s*stride + j], j < n  (mode 0: element-wise, S ascending; mode 1: 32-way tree; C: depthwise layout)."""
        part, out, n, stride = int(part), int(out), int(n), int(stride)
        if mode == 0 and not ((n | stride) & 3) and not ((part | out) & 15):
            mode = 2                     # same sums, float4 per thread
        ent = [part, out, n, stride, int(S), int(mode), int(C), -1]
        head = self.outs.get(out)
        if head is not None:             # a further contribution to the same gradient (shared parameter): chain it, order preserved
            h = self.entries[head]
            if (h[2], h[5] & 7, h[6]) != (n, mode, int(C)):
                self.flush()
                head = None
            else:
                tail = head
                while self.entries[tail][7] >= 0:
                    tail = self.entries[tail][7]
                self.entries[tail][7] = len(self.entries)
        if head is None:
            self.outs[out] = len(self.entries)
            self.heads.append(len(self.entries))
            if self.fresh is not None and not ab.on("no_fresh_reduce"):
                if out not in self.fresh:
                    ent[5] |= 8          # nothing has been reduced into this window since zero_grad: overwrite
                self.fresh.add(out)
        self.entries.append(ent)

    def flush(self):
        if self.pre_flush is not None:
            self.pre_flush()             # queued weight-gradient GEMMs register their slab reductions when they are launched
        if not self.entries:
            return
        key = tuple(tuple(e) for e in self.entries)
        hit = self.cache.get(key)
        if hit is None:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("deferred-reduce table changed during hipGraph capture (the warm-up step ran a different sequence)")
            if lib.query("tuber_multi_reduce_entry_bytes") != self._ENTRY.itemsize:
                raise RuntimeError("MultiReduceEntry layout drift between engine.py and libtuber_hip.so")
            tab = np.array(list(key), dtype=self._ENTRY)
            heads = np.array(self.heads, dtype=np.int32)
            ht = tab[heads]
            hm = ht["mode"] & 7
            per = np.where(hm == 0, (ht["n"] + 1023) // 1024, np.where(hm == 2, (ht["n"] + 4095) // 4096, (ht["n"] + 31) // 32)).astype(np.int64)
            blk = np.empty((int(per.sum()), 2), np.int32)
            blk[:, 0] = np.repeat(heads, per)
            starts = np.concatenate([[0], np.cumsum(per)[:-1]])
            blk[:, 1] = np.arange(len(blk), dtype=np.int64) - np.repeat(starts, per)
            if len(self.cache) >= 512:       # eager DDP flushes once per bottleneck: ~60 tables per step, all reused
                self.cache.clear()
            hit = (torch.from_numpy(tab.view(np.uint8).copy()).to(self.device), torch.from_numpy(blk).to(self.device), len(blk))
            self.cache[key] = hit
        if self.check_zero and not torch.cuda.is_current_stream_capturing():
            for h in self.heads:
                e = self.entries[h]
                if e[5] & 8:
                    buf = self.resolve(e[1], e[2]) if self.resolve is not None else None      # (depthwise layout: the same n elements, permuted)
                    assert buf is None or not bool(buf.any()), "deferred reduce: window at %#x taken as zero is not" % e[1]
        lib.call("tuber_multi_reduce", hit[0], hit[1], hit[2])
        self.entries, self.outs, self.heads = [], {}, []


class ParamStore:
    def __init__(self, module, device):
        self.module = module
        self.device = torch.device(device)
        names, params = [], []
        seen = set()
        for n, p in module.named_parameters():
            if id(p) in seen:
                continue
            seen.add(id(p))
            names.append(n)
            params.append(p)
        self.names, self.params = names, params
        self.offsets = {}
        off = 0
        for n, p in zip(names, params):
            self.offsets[n] = off
            off += _ceil(p.numel(), ALIGN)
        self.total = off
        self.flat = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.gflat = torch.zeros(off, dtype=torch.float32, device=self.device)
        self.shadow = torch.zeros(off, dtype=torch.bfloat16, device=self.device)
        with torch.no_grad():
            for n, p in zip(names, params):
                o = self.offsets[n]
                view = self.flat[o:o + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = self.gflat[o:o + p.numel()].view(p.shape)
        self._ptrs = [p.data_ptr() for p in params]
        # transposed GEMM weights: 2-D view [N, K] of 1x1x1 convs, linears and packed in-projections
        self.tinfo = {}
        toff, entries, tiles = 0, [], 0
        for n, p in zip(names, params):
            if p.dim() >= 2 and self._is_gemm_weight(n, p):
                N = p.shape[0]
                K = p.numel() // N
                ldt = _ceil(N, 64)
                self.tinfo[n] = (toff, N, K, ldt)
                tx, ty = (K + 63) // 64, (N + 63) // 64          # 64 x 64 tiles of tuber_multi_transpose_bf16
                entries.append((self.offsets[n], toff, N, K, ldt, tiles, tx, 0))
                tiles += tx * ty
                toff += _ceil(K * ldt, ALIGN)
        self.tshadow = torch.zeros(max(toff, ALIGN), dtype=torch.bfloat16, device=self.device)
        self.ttiles = tiles
        tab = np.zeros(len(entries), dtype=[("src", "<i8"), ("dst", "<i8"), ("R", "<i4"), ("C", "<i4"), ("ldt", "<i4"),
                                             ("tb", "<i4"), ("tx", "<i4"), ("pad", "<i4")])
        for i, e in enumerate(entries):
            tab[i] = e
        self.ttable = torch.from_numpy(tab.view(np.uint8).copy()).to(self.device)
        self.nmat = len(entries)
        self.step_seed = 0
        self._by_name = dict(zip(names, params))
        self.reducer = None              # ddp.FlatGradReducer when gradients are all-reduced (attach_reducer)
        # dropout seed lives in DEVICE memory (read by the kernels), so a captured hipGraph draws new masks every replay
        self.seed = torch.zeros(1, dtype=torch.int64, device=self.device)
        # tuber_decoder_coop_fwd's synchronisation words (arrival counter, XCC census, error word, departure counter): zero between launches
        self.coop_sync = torch.zeros(4, dtype=torch.int32, device=self.device)
        self.coop_off = False            # set by coop_failed(): a timed-out cooperative launch switches this engine to the launch chain
        self.defer = DeferredReduce(self.device)

        def _window(ptr, n, g=self.gflat):
            o = (ptr - g.data_ptr()) // 4
            return g[o:o + n] if 0 <= o and o + n <= g.numel() else None
        self.defer.resolve = _window
        self.wq = WgradQueue(self)
        self.defer.pre_flush = self.wq.flush

    @staticmethod
    def _is_gemm_weight(name, p):
        if name.endswith("conv3.weight") or name.endswith("query_embed.weight") or name.endswith("query_pool.weight"):
            return False
        if name == "backbone.body.conv1.weight":
            return False
        return name.endswith("weight")

    # -- consistency -------------------------------------------------------------------------
    def valid(self):
        """False if someone re-allocated the parameters (``model.to()``, ``.half()`` ...)."""
        return all(p.data_ptr() == q for p, q in zip(self.params, self._ptrs))

    def ensure_grad_views(self):
        """``p.grad`` of every trainable parameter is its window of the flat gradient buffer; frozen parameters
        (``requires_grad = False``) have ``p.grad is None`` like under autograd, and their windows stay zero."""
        for n, p in zip(self.names, self.params):
            if not p.requires_grad:
                p.grad = None
                continue
            o = self.offsets[n]
            want = self.gflat[o:o + p.numel()].view(p.shape)
            if p.grad is None or p.grad.data_ptr() != want.data_ptr():
                if p.grad is not None:
                    want.copy_(p.grad)
                p.grad = want

    def trainable(self, name):
        return self._by_name[name].requires_grad

    def trainable_signature(self):
        """hashable summary of which parameters are trainable (a captured hipGraph bakes the launch sequence in)."""
        return tuple(i for i, p in enumerate(self.params) if not p.requires_grad)

    def trainable_ranges(self):
        """merged [begin, end) ranges of the flat buffers that hold trainable parameters (gradient all-reduce)."""
        from .ddp import trainable_ranges
        return trainable_ranges(self)

    def partial(self, key, numel, fallback):
        """scratch for a weight-gradient kernel's partials -> (pointer or tensor, accumulate flag): arena + 2 when the second stage is
        deferred to ``defer.flush()``, else ``fallback(key, numel)`` (the shared workspace) + 1."""
        if self.defer.enabled:
            return self.defer.alloc(numel), 2
        return fallback(key, numel), 1

    def zero_grad(self):
        self.gflat.zero_()
        self.ensure_grad_views()
        self.defer.fresh = set()         # every gradient window is zero from here on, until something is reduced into it

    def begin_step(self, train):
        """new forward: call-site salts restart at 0; in training the device seed advances (captured in graphs)."""
        self.defer.reset()
        self.step_seed = 0
        if train:
            self.seed.add_(1)

    def coop_failed(self):
        """True when a barrier of tuber_decoder_coop_fwd timed out since the last call (one 16-byte device read: call it where the loop
        synchronises anyway).  The launch needs its 16 workgroups co-resident on one XCD; another kernel holding those CUs (a collective
        on the reducer's stream, another process) starves the barrier, which gives up after its spin bound.  The kernel is FAIL-SAFE:
        it overwrites its output with NaN, so the step's loss and gradient norm are NaN and the optimizer's device-side guard skips the
        update (csrc/optim.hip).  This call clears the words and switches the engine to the launch chain (``coop_off``; captured steps
        are keyed on it and re-captured), so the caller only has to repeat / continue."""
        w = self.coop_sync.cpu().tolist()
        if not w[2]:
            return False
        import sys
        self.coop_sync.zero_()
        self.coop_off = True
        print("[tuber] tuber_decoder_coop_fwd: a workgroup barrier timed out (sync words %s): the affected steps produced NaN and were skipped by the "
              "optimizer; the decoder runs as separate launches from here on" % w, file=sys.stderr, flush=True)
        return True

    def check_coop(self):
        """for callers that cannot repeat the affected work (bench.py's timed region, smoke): raise if a cooperative decoder launch failed."""
        if self.coop_failed():
            raise RuntimeError("tuber_decoder_coop_fwd: a workgroup barrier timed out; the affected steps are invalid (their outputs are NaN, their "
                               "optimizer updates were skipped).  The engine now runs the decoder as separate launches (same as TUBER_AB=no_decoder_coop)")

    def manual_seed(self, seed):
        self.seed.fill_(int(seed))

    # -- per-step refresh --------------------------------------------------------------------
    def refresh(self, backward=True):
        """bf16 shadow of the fp32 masters; ``backward``: also the transposed copies the data-gradient GEMMs read (not needed by a forward nobody
        differentiates)."""
        lib.call("tuber_cast_f32_bf16", self.flat, self.shadow, self.total)
        if self.nmat and backward:
            lib.call("tuber_multi_transpose_bf16", self.shadow, self.tshadow, self.ttable, self.nmat, self.ttiles)

    # -- accessors ---------------------------------------------------------------------------
    def w(self, name):
        """fp32 master view (flat 1-D)."""
        o = self.offsets[name]
        return self.flat[o:]

    def g(self, name):
        o = self.offsets[name]
        return self.gflat[o:]

    def wb(self, name):
        """bf16 shadow (flat 1-D view starting at the parameter)."""
        o = self.offsets[name]
        return self.shadow[o:]

    def wt(self, name):
        """(bf16 transposed weight flat view, ldt)."""
        toff, N, K, ldt = self.tinfo[name]
        return self.tshadow[toff:], ldt
