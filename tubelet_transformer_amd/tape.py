"""Hand-scheduled forward/backward of the transformer / head part of TubeR over the C ABI.

The transformer half of the training step is ~250 small launches forward and, under torch.autograd, ~950 backward (gradient
accumulation adds, zero fills, contiguous copies, one kernel per dropout ...).  On MI355X every launch costs 4-8 us of GPU
time even inside a hipGraph, so this module keeps its own tape instead: each op launches its forward kernels and records
one backward closure; ``Tape.backward`` replays them in reverse.  What that buys over autograd:

* gradient accumulation is fused into the producing GEMM (``R`` residual input of tuber_gemm_nt) -- a tensor consumed by
  several ops never costs an add kernel unless two non-GEMM producers meet;
* Dropout is fused into the consumer: LayerNorm(Dropout(x) + res) is one kernel forward and one backward (the mask is
  regenerated from the seed), FFN ReLU+Dropout is the epilogue of linear1, and its backward mask is the epilogue of linear2's
  data-gradient GEMM;
* ``with_pos_embed`` adds whose second operand has no gradient are aliases in the backward (no kernel);
* LayerNorms write straight into row / column slices of the consumer's buffer (decoder layer stack ``hs``, the class branch's
  concatenated [t | s] features), so there are no cat / stack copies in either direction.

Activations are 2-D token-major bf16 tensors [rows, E]; parameter gradients are accumulated by the kernels straight into the
ParamStore's flat fp32 gradient buffer.  Reference modules: models/transformer/transformer.py, transformer_layers.py,
models/tuber_ava.py:97-157.
"""
import numpy as np
import torch

from . import ab, lib
from .engine import TnArgs

BF = torch.bfloat16
_WS = {}


def workspace(dev, key, numel):
    t = _WS.get((dev, key))
    if t is None or t.numel() < numel:
        t = torch.empty(int(numel * 1.25) + 64, dtype=torch.float32, device=dev)
        _WS[(dev, key)] = t
    return t


def _ceil(x, m):
    return (x + m - 1) // m * m


def _map(ld, sL, s1=0, s2=0, B2=1):
    return np.array([ld, sL, s1, s2, B2], dtype=np.int64)


class Tape:
    def __init__(self, store, train):
        self.store, self.train, self.dev = store, train, store.device
        self.ops = []          # backward closures, forward order
        self.g = {}            # id(tensor) -> gradient tensor
        self.alias = {}        # id(tensor) -> tensor that receives its gradient (x + const)
        self.mask = {}         # id(h) -> 1/(1-p): h = Dropout(ReLU(.)) whose backward mask the consumer's dgrad GEMM applies
        self.premasked = set()
        self.stack = {}        # id(tensor) -> list of gradient tensors summed lazily by the producer's backward
        self.dry = False       # dry run: the ops allocate their outputs, draw their dropout salts and record their backward closures, but do
        self.dry_log = []      # not launch the forward kernel -- (kind, dict) per op, for a fused launch that fills those outputs (tuber.py)
        self.pending = {}      # id(gradient tensor) -> _PendingLN: a LayerNorm backward not launched yet (it may fuse into its consumer)
        self.pair_ok = set()   # id(y) of LayerNorm outputs whose backward can take two unsummed gradient contributions (fusable ones)
        self.lin_out = set()   # id(y) of linear() outputs a LayerNorm backward may fuse into (plain linear with 256 outputs, no ReLU / Dropout)
        self.f32 = {}          # eval precision mode: id(bf16 LayerNorm output) -> (that tensor, its fp32 twin): the residual stream between LayerNorms
        self.req = set()       # ids of tensors whose gradient is needed (they depend on a trainable parameter): the backward pass
        #                        skips weight gradients of frozen parameters and data gradients nobody consumes, like autograd does

    # -- gradient bookkeeping -------------------------------------------------------------------
    def rec(self, fn):
        if self.train:
            self.ops.append(fn)

    def target(self, t):
        while id(t) in self.alias:
            t = self.alias[id(t)]
        return t

    def needs(self, t):
        return id(self.target(t)) in self.req

    def mark(self, t):
        self.req.add(id(t))

    def force(self, g):
        """make ``g`` a materialised tensor: launch the LayerNorm backward that produces it if that is still pending, form the sum of an
        unsummed pair (someone is about to read or re-deposit g)"""
        if isinstance(g, _Pair):
            a, b = self.force(g.a), self.force(g.b)
            out = torch.empty_like(a)
            lib.call("tuber_axpby", a, b, out, a.numel(), 1.0, 1.0)
            return out
        rec = self.pending.get(id(g)) if g is not None else None
        if rec is not None:
            rec.force()
        return g

    def take(self, t, force=True, pair=False):
        """the gradient of t (removed).  force=False: a pending LayerNorm backward stays pending; pair=True: an unsummed _Pair is returned
        as it is (its two members materialised)"""
        g = self.g.pop(id(self.target(t)), None)
        if isinstance(g, _Pair):
            if not pair:
                return self.force(g)
            g.a = self.force(g.a)                        # (b may stay a pending LayerNorm backward: _PendingLN.fuse forms it on load)
            return g
        return self.force(g) if force else g

    def peek(self, t):
        return self.force(self.g.get(id(self.target(t))))

    def set(self, t, g):
        self.g[id(self.target(t))] = g

    def put(self, t, g):
        """deposit g for t; adds to an existing gradient (one axpby launch) when a GEMM could not fuse the accumulation."""
        t = self.target(t)
        if id(t) in self.stack:
            self.stack[id(t)].append(self.force(g))
            return
        cur = self.g.get(id(t))
        if cur is None:
            self.g[id(t)] = g
        elif (id(t) in self.pair_ok and not isinstance(cur, _Pair) and cur.dtype == BF and g.dtype == BF and cur.shape == g.shape
              and cur.is_contiguous() and g.is_contiguous()):
            self.g[id(t)] = _Pair(cur, g)                # its consumer (a fusable LayerNorm backward) adds on load
        else:
            cur = self.force(cur)
            out = torch.empty_like(cur)
            lib.call("tuber_axpby", cur, self.force(g), out, cur.numel(), 1.0, 1.0)
            self.g[id(t)] = out

    def backward(self, seeds):
        for t, g in seeds:
            if g is not None:
                self.put(t, g)
        for fn in reversed(self.ops):
            fn()
        for rec in list(self.pending.values()):         # (nothing consumed them: cannot happen for a gradient somebody asked for; be safe)
            rec.force()
        self.store.wq.flush()
        self.clear()

    def clear(self):
        self.ops.clear(); self.g.clear(); self.alias.clear(); self.mask.clear(); self.premasked.clear(); self.stack.clear()
        self.req.clear(); self.pending.clear(); self.lin_out.clear(); self.pair_ok.clear()

    def salt(self):
        self.store.step_seed += 1
        return self.store.step_seed


class _Pair:
    """two gradient contributions of one tensor whose sum has not been formed: a consumer that can add on load takes them as they are
    (layer_norm.bwd -> tuber_ln_bwd_dx's dy / dy2), anyone else gets the tuber_axpby launch (Tape.force)"""

    def __init__(self, a, b):
        self.a, self.b = a, b


class _PendingLN:
    """A LayerNorm backward that layer_norm.bwd did not launch: its consumer -- the backward of the linear whose output the LayerNorm
    normalised (attention out_proj, the FFN's linear2) -- runs next and launches LayerNorm backward + its own data-gradient GEMM as ONE
    kernel (tuber_ln_bwd_dx).  Anyone else who touches the two gradient tensors first (Tape.force) gets the stand-alone launch."""

    def __init__(self, tp, g, gptr, ldg, xhat, rstd, gamma, dx, dxd, p, salt, M, E, register, g2=None):
        self.tp, self.g, self.gptr, self.ldg, self.xhat, self.rstd, self.gamma = tp, g, gptr, ldg, xhat, rstd, gamma
        self.g2 = g2                                     # a second, unsummed contribution to the LayerNorm output's gradient (or None)
        self.dx, self.dxd, self.p, self.salt, self.M, self.E, self.register = dx, dxd, p, salt, M, E, register
        self.keys = [id(t) for t in (dx, dxd) if t is not None]
        for k in self.keys:
            tp.pending[k] = self

    def _done(self):
        for k in self.keys:
            self.tp.pending.pop(k, None)

    def force(self):
        st = self.tp.store
        self._done()
        if self.g2 is not None:                          # the stand-alone kernel takes one gradient tensor
            self.g = self.tp.force(_Pair(self.g, self.g2))
            self.gptr, self.ldg, self.g2 = self.g.data_ptr(), self.E, None
        nb = lib.query("tuber_layernorm_bwd_blocks", self.M)
        part, dgamma, dbeta, acc = self.register(nb)
        lib.call("tuber_layernorm_bwd", self.gptr, self.ldg, self.xhat, self.rstd, self.gamma, self.dx, self.dxd, part, dgamma, dbeta, acc,
                 self.M, self.E, self.p, st.seed, self.salt)

    def fuse(self, wt, ldt, Kin, out, res, cm, alpha):
        """LayerNorm backward + out = (gradient of the linear's output) . W [+ res | masked by cm > 0, x alpha] in one launch"""
        st = self.tp.store
        self._done()
        nb = lib.query("tuber_ln_bwd_dx_blocks", self.M)
        part, _, _, acc = self.register(nb)
        assert acc == 2
        # the second contribution: a tensor (added on load), or the still-pending backward of ANOTHER LayerNorm over the same rows without
        # Dropout / residual (decoder.norm on this layer's output) -- formed on load from its own saved tensors, never materialised
        ra = self.tp.pending.get(id(self.g2)) if self.g2 is not None else None
        if ra is not None and not (ra.p == 0.0 and ra.dxd is None and ra.g2 is None and ra.M == self.M and ra.E == self.E and ra.dx is self.g2):
            ra = None
        if ra is not None:
            ra._done()
            part_a, _, _, acc_a = ra.register(nb)
            assert acc_a == 2
            g2, other = None, (ra.gptr, ra.ldg, ra.xhat, ra.rstd, ra.gamma, part_a)
            self.keep = (ra.g, ra.dx)
        else:
            g2, other = self.tp.force(self.g2), (None, 0, None, None, None, None)
        lib.call("tuber_ln_bwd_dx", self.gptr, self.ldg, g2, self.E if g2 is not None else 0, self.xhat, self.rstd, self.gamma, self.dx, self.dxd, part, self.M, self.E,
                 self.p, st.seed, self.salt, wt, ldt, Kin, out, res, cm, alpha, *other)


# --------------------------------------------------------------------------------------------------
# ops
# --------------------------------------------------------------------------------------------------
def linear(tp, x, wname, bname=None, rows=None, relu=False, out_f32=False, drop=0.0):
    """y = [Dropout]([relu](x @ W[r0:r1]^T + b[r0:r1])) -- nn.Linear / packed in-projection slices / 1x1x1 Conv3d."""
    st = tp.store
    M, K = x.shape
    if rows is None:
        rows = (0, st.module.get_parameter(wname).shape[0])
    r0, r1 = rows
    N = r1 - r0
    dev = x.device
    wb = st.shadow.data_ptr() + 2 * (st.offsets[wname] + r0 * K)
    bias = st.flat.data_ptr() + 4 * (st.offsets[bname] + r0) if bname else None
    y = torch.empty(M, N, dtype=torch.float32 if out_f32 else BF, device=dev)
    p = float(drop)
    salt = tp.salt() if p > 0.0 else 0
    if tp.dry:
        tp.dry_log.append(("linear", dict(x=x, y=y, w=wb, b=bias, relu=relu, p=p, salt=salt, out_f32=out_f32)))
    else:
        lib.call("tuber_gemm_nt", x, K, wb, K, y, N, M, N, K, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                 0, bias, None, 0, 1 if relu else 0, 1 if out_f32 else 0, None, None, None, 0, None, None, 1.0, p, st.seed, salt, None, 0, None)
    if not tp.train:
        return y
    wreq = st.trainable(wname)
    breq = bool(bname) and st.trainable(bname)
    xreq = tp.needs(x)
    if not (wreq or breq or xreq):
        return y
    tp.mark(y)
    t16 = int(p * 4294967296.0) >> 16                    # the quantised drop probability the kernels realise (common.h: dropout_inv_keep)
    inv_keep = 65536.0 / (65536 - t16) if t16 else 1.0
    if relu:
        tp.mask[id(y)] = inv_keep
    toff, _, _, ldt = st.tinfo[wname]
    if not relu and not out_f32 and p == 0.0 and N == 256 and r0 == 0 and xreq and ldt >= N and lib.query("tuber_ln_bwd_dx_pays", M, N, K) == 1:
        tp.lin_out.add(id(y))                            # a LayerNorm over y may leave its backward to this linear's (layer_norm.bwd)

    def fused_dgrad(rec, g):
        """the data gradient of x together with the pending LayerNorm backward that produces g (one launch); mirrors the epilogue choice of
        the stand-alone GEMM below.  -> True when launched."""
        tx = tp.target(x)
        if id(tx) in tp.stack:
            return False
        wt = st.tshadow.data_ptr() + 2 * (toff + r0)
        r = tp.g.pop(id(tx), None)
        if isinstance(r, _Pair) or (r is not None and (r.dtype != BF or tuple(r.shape) != (M, K) or not r.is_contiguous() or tp.pending.get(id(r)) is rec)):
            tp.g[id(tx)] = r                             # (an unsummed pair, Tape.put on a pair_ok tensor: left to the stand-alone GEMM, which forces it)
            return False
        tp.force(r)
        dx = torch.empty(M, K, dtype=BF, device=dev)
        if id(tx) in tp.mask and r is None and tx is x:
            rec.fuse(wt, ldt, K, dx, None, x, tp.mask[id(tx)])
            tp.premasked.add(id(tx))
        else:
            rec.fuse(wt, ldt, K, dx, r, None, 1.0)
        tp.put(tx, dx)
        return True

    def bwd():
        g = tp.take(y, force=False)
        if g is None:
            return
        rec = tp.pending.get(id(g))
        dgrad_done = False
        if rec is not None:
            if id(y) in tp.lin_out and g is (rec.dxd if rec.p > 0.0 else rec.dx) and g.dtype == BF and tuple(g.shape) == (M, N):
                dgrad_done = fused_dgrad(rec, g)         # (before the weight gradient below: a full queue launches at once and reads g)
            if not dgrad_done:
                rec.force()
        Np = _ceil(N, 64)
        if g.dtype != BF:
            gb = torch.empty(M, Np, dtype=BF, device=dev)
            lib.call("tuber_cast_pad_rows", g if g.dtype == torch.float32 else g.float(), gb, M, N, Np)
            ldg = Np
        elif Np != N:
            gb = torch.zeros(M, Np, dtype=BF, device=dev)
            gb[:, :N] = g
            ldg = Np
        else:
            gb, ldg = g, N
        if relu and id(y) not in tp.premasked:
            gm = torch.empty_like(gb)
            lib.call("tuber_relu_mask", gb, y, gm, M * N, inv_keep)       # dropped elements have y == 0 too
            gb = gm
        elif p > 0.0 and not relu:
            assert ldg == N
            gm = torch.empty_like(gb)
            lib.call("tuber_dropout", gb, gm, M * N, p, st.seed, salt)    # same (seed, salt, m*N+n) stream as the epilogue
            gb = gm
        gw = st.gflat.data_ptr() + 4 * (st.offsets[wname] + r0 * K)
        ws = lambda k, n: workspace(dev, k, n)
        gbias = st.gflat.data_ptr() + 4 * (st.offsets[bname] + r0) if breq else None
        fuse_b = 0
        N8 = _ceil(N, 8)
        if (wreq and N8 != N and st.wq.enabled and st.defer.enabled and ldg >= N8 and st.wq.eligible(M, N8, K, ldg, K)
                and lib.query("tuber_gemm_tn_slabs", M, N8, K) == 1):
            # a head with 3 or 4 outputs (class_embed_b, the last box layer): the transpose-read kernel wants multiples of 8, so the product runs
            # over the zero-padded columns of gb into arena scratch [N8][K] and the deferred reduction adds its first N rows to the gradient --
            # queued with its neighbours instead of a tuber_gemm_tn launch of its own (8 - 12 us each on the critical path)
            scratch = st.defer.alloc(N8 * K)
            st.wq.add(TnArgs(gb.data_ptr(), ldg, x.data_ptr(), K, None, scratch, 0, M, N8, K, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, None, None, None),
                      (gb, x), [(scratch, gw, N * K, N8 * K, 1, 0)])
        elif wreq:
            S = lib.query("tuber_gemm_tn_slabs", M, N, K)
            # bias gradient inside the GEMM: 1 = accumulated directly (single slab), 2 = one partial row per slab
            fuse_b = lib.query("tuber_gemm_tn_fuses_bias", M, N, K, ldg, K) if breq else 0
            part, acc = st.partial("tn", S * N * K, ws) if S > 1 else (None, 1)
            bpart = st.partial("cs", S * N, ws)[0] if fuse_b == 2 else None
            bptr = gbias if fuse_b == 1 else bpart
            defers = []
            if acc == 2:
                defers.append((part, gw, N * K, N * K, S, 0 if S <= 16 else 1))
                if fuse_b == 2:
                    defers.append((bpart, gbias, N, N, S, 1))
            if st.wq.enabled and (S == 1 or acc == 2) and st.wq.eligible(M, N, K, ldg, K):
                # nothing reads a weight gradient before the optimizer: queued, launched with its neighbours (engine.WgradQueue)
                ptr = lambda t: None if t is None else (t if isinstance(t, int) else t.data_ptr())
                st.wq.add(TnArgs(gb.data_ptr(), ldg, x.data_ptr(), K, ptr(part), gw, acc, M, N, K, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, None, None, ptr(bptr)),
                          (gb, x), defers)
            else:
                lib.call("tuber_gemm_tn", gb, ldg, x, K, part, gw, acc, M, N, K, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0, None, 0, None, None, None, bptr)
                for d in defers:
                    st.defer.add(*d)
                if fuse_b == 2 and acc != 2:
                    lib.call("tuber_reduce_rows", bpart, gbias, S, N, 1)
        if breq and not fuse_b:
            nbc = lib.query("tuber_colsum_blocks", M)
            part, acc = st.partial("cs", nbc * N, ws) if nbc > 1 else (None, 1)
            lib.call("tuber_colsum", gb, part, gbias, acc, M, N, ldg)
            if acc == 2:
                st.defer.add(part, gbias, N, N, nbc, 1)
        if not xreq or dgrad_done:
            return
        # data gradient; accumulation with an existing gradient of x and the ReLU/Dropout mask of x are GEMM epilogues
        wt = st.tshadow.data_ptr() + 2 * (toff + r0)           # W^T[:, r0:r1]: column offset, ld = ldt
        Kred = Np if (r0 == 0 and Np <= ldt) else N             # padded columns of both operands are zero
        assert Kred % 64 == 0, "row slices must be multiples of 64"
        tx = tp.target(x)
        dx = torch.empty(M, K, dtype=BF, device=dev)
        if id(tx) in tp.stack:
            lib.call("tuber_gemm_nt", gb, ldg, wt, ldt, dx, K, M, K, Kred, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                     0, None, None, 0, 0, 0, None, None, None, 0, None, None, 1.0, 0.0, None, 0, None, 0, None)
            tp.put(tx, dx)
            return
        r = tp.force(tp.g.pop(id(tx), None))
        if r is not None and (r.dtype != BF or tuple(r.shape) != (M, K) or not r.is_contiguous()):
            tp.g[id(tx)] = r
            r = None
        if id(tx) in tp.mask and r is None and tx is x:
            lib.call("tuber_gemm_nt", gb, ldg, wt, ldt, dx, K, M, K, Kred, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                     2, None, None, 0, 0, 0, None, None, x, K, None, None, tp.mask[id(tx)], 0.0, None, 0, None, 0, None)
            tp.premasked.add(id(tx))
        elif M <= 64 and Kred >= 1024 and K % 16 == 0 and ldg % 8 == 0 and not ab.on("no_in_proj_dx2"):
            # few rows, long reduction (the decoder's linear1): 16 workgroups x 4 waves over the reduction instead of 4 wave-split tiles
            lib.call("tuber_rows_dx2", gb, ldg, M, Kred, Kred, wt, ldt, K, dx, r, None)
        else:
            lib.call("tuber_gemm_nt", gb, ldg, wt, ldt, dx, K, M, K, Kred, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                     0, None, r, K, 0, 0, None, None, None, 0, None, None, 1.0, 0.0, None, 0, None, 0, None)
        tp.put(tx, dx)
    tp.rec(bwd)
    return y


def in_proj(tp, x, addend, wname, bname, rows, add_cols):
    """Packed attention in-projection with the positional embedding folded in (``with_pos_embed``, transformer.py:150-159,215-240):
        y[:, :add_cols] = (x + addend) @ W[r0:r0+add_cols]^T + b,      y[:, add_cols:] = x @ W[r0+add_cols:r1]^T + b
    as ONE GEMM (tuber_gemm_nt_addproj) -- the reference's q / k see x + pos, its v sees x; here that was an add kernel and two GEMM
    launches.  ``addend`` may carry a gradient (the decoder's query_pos) or not (the sine encoding).  Backward: weight + bias gradients
    as two queued dW GEMMs on disjoint row blocks (the first with the A + A2 operand formed on load), ONE data-gradient GEMM for x over all
    N columns, and one over the first add_cols columns for the addend when it needs a gradient."""
    st = tp.store
    M, K = x.shape
    r0, r1 = rows
    N = r1 - r0
    dev = x.device
    assert N % 64 == 0 and add_cols % 128 == 0 and 0 < add_cols <= N and tuple(addend.shape) == (M, K)
    wb = st.shadow.data_ptr() + 2 * (st.offsets[wname] + r0 * K)
    bias = st.flat.data_ptr() + 4 * (st.offsets[bname] + r0)
    y = torch.empty(M, N, dtype=BF, device=dev)
    if tp.dry:
        tp.dry_log.append(("in_proj", dict(x=x, addend=addend, y=y, w=wb, b=bias, add_cols=add_cols)))
    else:
        lib.call("tuber_gemm_nt_addproj", x, K, addend, K, add_cols, wb, K, y, N, M, N, K, bias)
    if not tp.train:
        return y
    wreq, breq = st.trainable(wname), st.trainable(bname)
    xreq, areq = tp.needs(x), tp.needs(addend)
    if not (wreq or breq or xreq or areq):
        return y
    tp.mark(y)

    def bwd():
        g = tp.take(y)
        if g is None:
            return
        assert g.dtype == BF and g.is_contiguous()
        ws = lambda k, n: workspace(dev, k, n)
        gw = st.gflat.data_ptr() + 4 * (st.offsets[wname] + r0 * K)
        gbias = st.gflat.data_ptr() + 4 * (st.offsets[bname] + r0)
        # weight / bias gradients: row block [0, add_cols) against x + addend, row block [add_cols, N) against x
        blocks = [(0, add_cols, addend)] + ([(add_cols, N, None)] if add_cols < N else [])
        for c0, c1, a2 in blocks:
            n = c1 - c0
            gptr = g.data_ptr() + 2 * c0
            if wreq:
                S = lib.query("tuber_gemm_tn_slabs", M, n, K)
                fuse_b = lib.query("tuber_gemm_tn_fuses_bias", M, n, K, N, K) if breq else 0
                assert fuse_b, "in-projection shapes are transpose-read shapes"
                part, acc = st.partial("tn", S * n * K, ws) if S > 1 else (None, 1)
                bpart = st.partial("cs", S * n, ws)[0] if fuse_b == 2 else None
                bptr = (gbias + 4 * c0) if fuse_b == 1 else bpart
                out = gw + 4 * c0 * K
                defers = []
                if acc == 2:
                    defers.append((part, out, n * K, n * K, S, 0 if S <= 16 else 1))
                    if fuse_b == 2:
                        defers.append((bpart, gbias + 4 * c0, n, n, S, 1))
                ptr = lambda t: None if t is None else (t if isinstance(t, int) else t.data_ptr())
                args = TnArgs(gptr, N, x.data_ptr(), K, ptr(part), out, acc, M, n, K, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, None, None, ptr(bptr),
                              ptr(a2), K if a2 is not None else 0)
                if st.wq.enabled and (S == 1 or acc == 2):
                    st.wq.add(args, (g, x, a2), defers)
                else:
                    arr = (TnArgs * 1)(args)
                    lib.call("tuber_gemm_tn_group", arr, 1)
                    for d in defers:
                        st.defer.add(*d)
                    if fuse_b == 2 and acc != 2:
                        lib.call("tuber_reduce_rows", bpart, gbias + 4 * c0, S, n, 1)
            elif breq:
                nbc = lib.query("tuber_colsum_blocks", M)
                part, acc = st.partial("cs", nbc * n, ws) if nbc > 1 else (None, 1)
                lib.call("tuber_colsum", gptr, part, gbias + 4 * c0, acc, M, n, N)
                if acc == 2:
                    st.defer.add(part, gbias + 4 * c0, n, n, nbc, 1)
        # data gradients
        toff, _, _, ldt = st.tinfo[wname]
        wt = st.tshadow.data_ptr() + 2 * (toff + r0)           # W^T[:, r0:r1]: column offset, ld = ldt
        plain = lambda ncols, res, out: lib.call("tuber_gemm_nt", g, N, wt, ldt, out, K, M, K, ncols, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                                                 0, None, res, K, 0, 0, None, None, None, 0, None, None, 1.0, 0.0, None, 0, None, 0, None)
        shared = False
        if xreq and areq and M <= 64 and K % 64 == 0 and not ab.on("no_in_proj_dx2"):
            # few rows (the decoder's queries): both data gradients from ONE pass over g and W -- the addend's is a prefix of x's reduction
            tx = tp.target(x)
            r = None
            if id(tx) not in tp.stack:
                r = tp.force(tp.g.pop(id(tx), None))
                if r is not None and (r.dtype != BF or tuple(r.shape) != (M, K) or not r.is_contiguous()):
                    tp.g[id(tx)] = r
                    r = None
            if r is not None or add_cols < N:            # (else the two gradients are the same tensor: the one plain GEMM below)
                dx = torch.empty(M, K, dtype=BF, device=dev)
                da = torch.empty(M, K, dtype=BF, device=dev)
                lib.call("tuber_rows_dx2", g, N, M, N, add_cols, wt, ldt, K, dx, r, da)
                tp.put(tx, dx)
                tp.put(addend, da)
                return
        if xreq:
            tx = tp.target(x)
            dx = torch.empty(M, K, dtype=BF, device=dev)
            r = None
            if id(tx) not in tp.stack:
                r = tp.force(tp.g.pop(id(tx), None))
                if r is not None and (r.dtype != BF or tuple(r.shape) != (M, K) or not r.is_contiguous()):
                    tp.g[id(tx)] = r
                    r = None
            plain(N, r, dx)                                  # an existing gradient of x is accumulated by the GEMM's residual input
            tp.put(tx, dx)
            shared = r is None and add_cols == N             # dx is the bare product g.W over the addend's columns
        if areq:
            if shared:
                tp.put(addend, dx)                           # same columns, same gradient
            else:
                da = torch.empty(M, K, dtype=BF, device=dev)
                plain(add_cols, None, da)
                tp.put(addend, da)
    tp.rec(bwd)
    return y


def layer_norm(tp, x, res, prefix, drop=0.0, out=None):
    """LayerNorm(Dropout_p(x) + res).  ``out`` = (base, row0, col0): write into base[row0:row0+M, col0:col0+E] (the consumer reads
    ``base``; the gradient is read from the same window of base's gradient) -- returns base then."""
    st = tp.store
    M, E = x.shape
    dev = x.device
    gamma = st.flat.data_ptr() + 4 * st.offsets[prefix + ".weight"]
    beta = st.flat.data_ptr() + 4 * st.offsets[prefix + ".bias"]
    if out is None:
        y, ldy, yptr = torch.empty(M, E, dtype=BF, device=dev), E, None
        yptr = y.data_ptr()
    else:
        base, row0, col0 = out
        ldy = base.shape[1]
        yptr = base.data_ptr() + 2 * (row0 * ldy + col0)
        y = base
    p = float(drop)
    salt = tp.salt() if p > 0.0 else 0
    xhat = torch.empty(M, E, dtype=BF, device=dev) if tp.train else None
    rstd = torch.empty(M, dtype=torch.float32, device=dev) if tp.train else None
    if tp.dry:
        tp.dry_log.append(("layer_norm", dict(x=x, res=res, yptr=yptr, ldy=ldy, xhat=xhat, rstd=rstd, gamma=gamma, beta=beta, p=p, salt=salt)))
    elif not tp.train and p == 0.0 and ab.eval_fp32_stream():
        # eval precision mode: the post-norm residual stream stays fp32 from LayerNorm to LayerNorm (tp.f32: fp32 twin of a bf16 LayerNorm
        # output); the bf16 result is only the operand of the GEMMs behind it.  The training path's bf16-stored stream costs ~1.3x on the actor
        # logits of a fixture with identity-dominated layers (measured on the oracle)
        x32, r32 = tp.f32.get(id(x)), (tp.f32.get(id(res)) if res is not None else None)
        y32 = torch.empty(M, E, dtype=torch.float32, device=dev) if out is None else None
        lib.call("tuber_layernorm_fwd_f32", x, x32[1] if x32 else None, res, r32[1] if r32 else None, gamma, beta, yptr, ldy, y32, M, E, 1e-5)
        if y32 is not None:
            tp.f32[id(y)] = (y, y32)             # (the bf16 tensor is held too: its id must not be recycled while the twin is listed)
    else:
        lib.call("tuber_layernorm_fwd", x, res, gamma, beta, yptr, ldy, xhat, rstd, M, E, 1e-5, p, st.seed, salt)
    if not tp.train:
        return y
    preq = st.trainable(prefix + ".weight") or st.trainable(prefix + ".bias")
    xreq, rreq = tp.needs(x), res is not None and tp.needs(res)
    if not (preq or xreq or rreq):
        return y
    tp.mark(y)
    if xreq and out is None and id(tp.target(x)) in tp.lin_out and not ab.on("no_ln_bwd_fusion"):
        tp.pair_ok.add(id(y))                            # two contributions to y's gradient stay unsummed: tuber_ln_bwd_dx adds them on load

    def bwd():
        g2 = None
        if out is None:
            g = tp.take(y, pair=True)
            if g is None:
                return
            if isinstance(g, _Pair):
                g, g2 = g.a, g.b                         # (g2 may be a pending LayerNorm backward, see _PendingLN.fuse)
            gptr, ldg = g.data_ptr(), E
        else:
            g = tp.peek(base)               # shared with the other writers of base; dropped when the tape is cleared
            if g is None:
                return
            assert g.dtype == BF and g.is_contiguous() and g.shape[1] == ldy
            gptr, ldg = g.data_ptr() + 2 * (row0 * ldy + col0), ldy
        need_res = res is not None or p == 0.0
        dx = torch.empty(M, E, dtype=BF, device=dev) if need_res else None
        dxd = torch.empty(M, E, dtype=BF, device=dev) if p > 0.0 else None

        def register(nb):
            """partial rows [nb][2E] for the kernel about to be launched + where their sums go -> (partial, dgamma, dbeta, accumulate flag)"""
            dgamma = st.gflat.data_ptr() + 4 * st.offsets[prefix + ".weight"]
            dbeta = st.gflat.data_ptr() + 4 * st.offsets[prefix + ".bias"]
            part, acc = st.partial("ln", 2 * nb * E, lambda k, n: workspace(dev, k, n))
            if not preq and acc != 2:          # frozen LayerNorm, immediate reductions: gamma/beta gradients go to scratch
                dgamma = workspace(dev, "ln_frozen", 2 * E).data_ptr()
                dbeta = dgamma + 4 * E
            if acc == 2 and preq:
                if dbeta == dgamma + 4 * E:
                    st.defer.add(part, dgamma, 2 * E, 2 * E, nb, 1)
                else:
                    st.defer.add(part, dgamma, E, 2 * E, nb, 1)
                    st.defer.add(part + 4 * E, dbeta, E, 2 * E, nb, 1)
            return part, dgamma, dbeta, acc

        rec = _PendingLN(tp, g, gptr, ldg, xhat, rstd, gamma, dx, dxd, p, salt, M, E, register, g2)
        # the backward of the linear that produced x runs next and can take the LayerNorm backward into its data-gradient launch
        # (tuber_ln_bwd_dx); everything else gets the stand-alone kernel right here
        fusable = xreq and out is None and st.defer.enabled and not ab.on("no_ln_bwd_fusion") and id(tp.target(x)) in tp.lin_out
        # ... and a LayerNorm without Dropout / residual over the OUTPUT of such a LayerNorm (the shared decoder.norm on every decoder layer's
        # output) whose input already holds a gradient: the pair stays unsummed and that LayerNorm's fused backward forms this one on load
        chained = (xreq and out is not None and p == 0.0 and res is None and E == 256 and st.defer.enabled and not ab.on("no_ln_bwd_fusion")
                   and id(tp.target(x)) in tp.pair_ok and isinstance(tp.g.get(id(tp.target(x))), torch.Tensor))
        if not (fusable or chained):
            rec.force()
        if xreq:
            tp.put(x, dxd if p > 0.0 else dx)
        if rreq:
            tp.put(res, dx)
    tp.rec(bwd)
    return y


def attention(tp, roles, geom, kpm, pdrop, *tensors):
    """Multi-head attention core on packed projections (32-wide heads).  ``tensors`` are the distinct 2-D bf16 inputs;
    ``roles`` = ((ti, col_off),)*3 locate Q, K, V; ``geom`` = (B, H, Lq, Lk, qmap, kmap) with (sL,s1,s2,B2) token maps."""
    st = tp.store
    B, H, Lq, Lk, qmap, kmap = geom
    (qi, qo), (ki, ko), (vi, vo) = roles
    tq, tk, tv = tensors[qi], tensors[ki], tensors[vi]
    dev = tq.device
    E = H * 32
    o = torch.empty(tq.shape[0], E, dtype=BF, device=dev)
    lse = torch.empty(B, H, Lq, dtype=torch.float32, device=dev)
    mq, mk, mv = _map(tq.shape[1], *qmap), _map(tk.shape[1], *kmap), _map(tv.shape[1], *kmap)
    mo = _map(E, *qmap)
    scale = 32 ** -0.5
    p = float(pdrop)
    salt = tp.salt()
    if tp.dry:
        tp.dry_log.append(("attention", dict(tensors=tensors, roles=roles, o=o, lse=lse, kpm=kpm, geom=geom, p=p, salt=salt)))
    else:
        lib.call("tuber_attn_fwd", tq.data_ptr() + 2 * qo, mq.ctypes.data, tk.data_ptr() + 2 * ko, mk.ctypes.data,
                 tv.data_ptr() + 2 * vo, mv.ctypes.data, o, mo.ctypes.data, lse, kpm, B, H, Lq, Lk, scale, float(p), st.seed, salt)
    if not tp.train:
        return o
    treq = [tp.needs(t) for t in tensors]
    if not any(treq):
        return o
    tp.mark(o)

    def bwd():
        g = tp.take(o)
        if g is None:
            return
        covered = [0] * len(tensors)
        for ti, _ in roles:
            covered[ti] += E
        grads = [torch.empty_like(t) if covered[i] >= t.shape[1] else torch.zeros_like(t) for i, t in enumerate(tensors)]
        delta = torch.empty(B, H, Lq, dtype=torch.float32, device=dev)
        lib.call("tuber_attn_bwd", tq.data_ptr() + 2 * qo, mq.ctypes.data, tk.data_ptr() + 2 * ko, mk.ctypes.data,
                 tv.data_ptr() + 2 * vo, mv.ctypes.data, o, mo.ctypes.data, lse, kpm, g, mo.ctypes.data,
                 grads[qi].data_ptr() + 2 * qo, mq.ctypes.data, grads[ki].data_ptr() + 2 * ko, mk.ctypes.data,
                 grads[vi].data_ptr() + 2 * vo, mv.ctypes.data, delta, B, H, Lq, Lk, scale, float(p), st.seed, salt)
        for t, gt, need in zip(tensors, grads, treq):
            if need:
                tp.put(t, gt)
    tp.rec(bwd)
    return o


def attention_wide(tp, q, kv, HW, T, pdrop):
    """LSTR pooling attention: one query per pixel, 8 heads of 256 (q [NQ,2048]; kv [rows,4096] = [k|v])."""
    st = tp.store
    NQ = q.shape[0]
    o = torch.empty(NQ, 2048, dtype=BF, device=q.device)
    p = float(pdrop)
    salt = tp.salt()
    lib.call("tuber_attn_wide_fwd", q, kv, o, NQ, HW, T, float(p), st.seed, salt)
    if not tp.train:
        return o
    qreq, kreq = tp.needs(q), tp.needs(kv)
    if not (qreq or kreq):
        return o
    tp.mark(o)

    def bwd():
        g = tp.take(o)
        if g is None:
            return
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        lib.call("tuber_attn_wide_bwd", q, kv, g, dq, dkv, NQ, HW, T, float(p), st.seed, salt)
        if qreq:
            tp.put(q, dq)
        if kreq:
            tp.put(kv, dkv)
    tp.rec(bwd)
    return o


def add_const(tp, a, c):
    """a + c where c carries no gradient (positional encodings): the gradient of the sum IS the gradient of a."""
    out = torch.empty_like(a)
    lib.call("tuber_axpby", a, c, out, a.numel(), 1.0, 1.0)
    if tp.train:
        tp.alias[id(out)] = a
        tp.rec(lambda keep=(out, a): None)           # keeps both tensors (and their ids) alive for the alias map
    return out


def add(tp, a, b):
    """a + b with gradients to both."""
    out = torch.empty_like(a)
    lib.call("tuber_axpby", a, b, out, a.numel(), 1.0, 1.0)
    areq, breq = tp.needs(a), tp.needs(b)
    if tp.train and (areq or breq):
        tp.mark(out)

        def bwd():
            g = tp.take(out)
            if g is not None:
                if areq:
                    tp.put(a, g)
                if breq:
                    tp.put(b, g)
        tp.rec(bwd)
    return out


def gather_sum(tp, x, fwd, bwd_map):
    """out[(a,b,c)] = mul * sum_d x[a*sa+b*sb+c*sc+d*sd]; ``bwd_map`` is the adjoint index map producing the x-shaped gradient."""
    A, B, C, D, sa, sb, sc, sd, mul = fwd
    E = x.shape[1]
    out = torch.empty(A * B * C, E, dtype=BF, device=x.device)
    lib.call("tuber_rows_gather_sum", x, out, A, B, C, D, sa, sb, sc, sd, E, float(mul))
    if tp.train and tp.needs(x):
        tp.mark(out)

        def bwd():
            g = tp.take(out)
            if g is None:
                return
            A2, B2, C2, D2, ta, tb, tc, td, mul2 = bwd_map
            assert A2 * B2 * C2 == x.shape[0]
            dx = torch.empty(x.shape[0], E, dtype=BF, device=x.device)
            lib.call("tuber_rows_gather_sum", g, dx, A2, B2, C2, D2, ta, tb, tc, td, E, float(mul2))
            tp.put(x, dx)
        tp.rec(bwd)
    return out


def param_rows(tp, name, B):
    """bf16 rows of an embedding-like parameter [Q, E], repeated B times: rows (b, q).  Its consumers' gradients are collected
    and summed once (query_embed is added to the decoder state twelve times)."""
    st = tp.store
    Q, E = st.module.get_parameter(name).shape
    src = st.shadow.data_ptr() + 2 * st.offsets[name]
    out = torch.empty(B * Q, E, dtype=BF, device=st.device)
    lib.call("tuber_rows_gather_sum", src, out, B, 1, Q, 1, 0, 0, 1, 0, E, 1.0)
    if tp.train and st.trainable(name):
        tp.mark(out)
        tp.stack[id(out)] = []

        def bwd(keep=out):
            gs = tp.stack.pop(id(keep), [])
            if not gs:
                return
            gp = st.gflat.data_ptr() + 4 * st.offsets[name]
            n = len(gs)
            if n > 1:
                allg = torch.cat(gs, dim=0)                      # [n*B*Q, E] -> rows (use, b) of [Q*E]
            else:
                allg = gs[0]
            R = lib.query("tuber_colsum_blocks", n * B)
            part, acc = st.partial("cs", R * Q * E, lambda k, m: workspace(st.device, k, m)) if R > 1 else (None, 1)
            lib.call("tuber_colsum", allg, part, gp, acc, n * B, Q * E, Q * E)
            if acc == 2:
                st.defer.add(part, gp, Q * E, Q * E, R, 1)
        tp.rec(bwd)
    return out


def dropout(tp, x, p):
    if p <= 0.0:
        return x
    st = tp.store
    salt = tp.salt()
    y = torch.empty_like(x)
    lib.call("tuber_dropout", x, y, x.numel(), float(p), st.seed, salt)
    if not tp.train or not tp.needs(x):
        return y
    tp.mark(y)

    def bwd():
        g = tp.take(y)
        if g is None:
            return
        dx = torch.empty_like(g)
        lib.call("tuber_dropout", g, dx, g.numel(), float(p), st.seed, salt)
        tp.put(x, dx)
    tp.rec(bwd)
    return y


def sigmoid(tp, x):
    y = torch.empty_like(x)
    lib.call("tuber_sigmoid_fwd", x, y, x.numel())
    if tp.train and tp.needs(x):
        tp.mark(y)

        def bwd():
            g = tp.take(y)
            if g is None:
                return
            dx = torch.empty_like(y)
            lib.call("tuber_sigmoid_bwd", g.contiguous(), y, dx, y.numel())
            tp.put(x, dx)
        tp.rec(bwd)
    return y


def temporal_max(tp, feat, B, Tp, hw):
    """feat rows (b,t,hw) -> rows (b,hw): max over the T' frames, nn.MaxPool3d((T',1,1)) (backbone_builder.py:45-47,73)."""
    C = feat.shape[1]
    out = torch.empty(B * hw, C, dtype=BF, device=feat.device)
    arg = torch.empty(B * hw, C, dtype=torch.uint8, device=feat.device) if tp.train else None
    lib.call("tuber_temporal_max_fwd", feat, out, arg, B, Tp, hw, C)
    if tp.train and tp.needs(feat):
        tp.mark(out)

        def bwd():
            g = tp.take(out)
            if g is None:
                return
            d = torch.empty_like(feat)
            lib.call("tuber_temporal_max_bwd", g.contiguous(), arg, d, B, Tp, hw, C)
            tp.put(feat, d)
        tp.rec(bwd)
    return out


def mid_frame(tp, feat, B, Tp, hw):
    """feat rows (b,t,hw) -> rows (b,hw) of the middle frame (backbone_builder.py:79-80); plain torch indexing (JHMDB only)."""
    C = feat.shape[1]
    out = feat.view(B, Tp, hw, C)[:, Tp // 2].reshape(B * hw, C)
    if tp.train and tp.needs(feat):
        tp.mark(out)

        def bwd():
            g = tp.take(out)
            if g is None:
                return
            d = torch.zeros_like(feat)
            d.view(B, Tp, hw, C)[:, Tp // 2] = g.view(B, hw, C)
            tp.put(feat, d)
        tp.rec(bwd)
    return out


def backbone(tp, runner, clips, bn_train):
    """CSN body: forward / backward are the hand-scheduled kernel sequences of CSNRunner (backbone.py)."""
    feat, saved = runner.forward(clips, bn_train)
    runner.last_shape = tuple(feat.shape)
    f2 = feat.view(-1, feat.shape[-1])
    if tp.train and runner.any_trainable():
        tp.mark(f2)

        def bwd():
            g = tp.take(f2)
            if g is not None:
                tp.store.wq.flush()          # transformer / head weight gradients: launched before the reducer is told they are final
                runner.backward(saved, g.contiguous())
        tp.rec(bwd)
    return f2
