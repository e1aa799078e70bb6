"""torch.autograd glue over the C ABI for the transformer / head part of TubeR.

Activations are 2-D token-major bf16 tensors [rows, E]; each Function launches the HIP kernels of
libtuber_hip.so for its forward and backward.  Parameter gradients are accumulated by the kernels
directly into the ParamStore's flat fp32 gradient buffer (the Functions return None for them).
A 0-d ``anchor`` tensor with requires_grad=True is threaded through Functions whose tensor inputs
do not require grad themselves, so autograd still schedules their backward.
"""
import numpy as np
import torch

from . import lib

BF = torch.bfloat16


def _ceil(x, m):
    return (x + m - 1) // m * m


_WS = {}


def workspace(dev, key, numel):
    t = _WS.get((dev, key))
    if t is None or t.numel() < numel:
        t = torch.empty(int(numel * 1.25) + 64, dtype=torch.float32, device=dev)
        _WS[(dev, key)] = t
    return t


class LinearFn(torch.autograd.Function):
    """y = [relu](x @ W[r0:r1]^T + b[r0:r1]); W is a (row slice of a) parameter in the store."""

    @staticmethod
    def forward(ctx, x, anchor, store, wname, bname, r0, r1, relu, out_f32):
        M, K = x.shape
        N = r1 - r0
        wb = store.shadow.data_ptr() + 2 * (store.offsets[wname] + r0 * K)
        bias = store.flat.data_ptr() + 4 * (store.offsets[bname] + r0) if bname else None
        y = torch.empty(M, N, dtype=torch.float32 if out_f32 else BF, device=x.device)
        lib.call("tuber_gemm_nt", x, K, wb, K, y, N, M, N, K, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                 0, bias, None, 0, 1 if relu else 0, 1 if out_f32 else 0, None, None, None, 0, None, None, 1.0, 0.0, None, 0)
        ctx.store, ctx.meta = store, (wname, bname, r0, r1, relu, out_f32, M, N, K)
        ctx.save_for_backward(x, y if relu else None)
        return y

    @staticmethod
    def backward(ctx, g):
        store = ctx.store
        wname, bname, r0, r1, relu, out_f32, M, N, K = ctx.meta
        x, y = ctx.saved_tensors
        dev = x.device
        Np = _ceil(N, 64)
        g = g.contiguous()
        if out_f32 or g.dtype != BF:
            gb = torch.empty(M, Np, dtype=BF, device=dev)
            lib.call("tuber_cast_pad_rows", g.float() if g.dtype != torch.float32 else g, gb, M, N, Np)
            ldg = Np
        elif Np != N:
            gb = torch.zeros(M, Np, dtype=BF, device=dev)
            gb[:, :N] = g
            ldg = Np
        else:
            gb, ldg = g, N
        if relu:
            assert ldg == N
            gm = torch.empty_like(gb)
            lib.call("tuber_relu_mask", gb, y, gm, M * N)
            gb = gm
        # weight / bias gradients straight into the flat gradient buffer
        gw = store.gflat.data_ptr() + 4 * (store.offsets[wname] + r0 * K)
        S = lib.query("tuber_gemm_tn_slabs", M, N, K)
        with store.side(gb, x):            # weight / bias gradients: off the critical path, on the side stream
            part = workspace(dev, "tn", S * N * K)
            lib.call("tuber_gemm_tn", gb, ldg, x, K, part, gw, 1, M, N, K, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0)
            if bname:
                gbias = store.gflat.data_ptr() + 4 * (store.offsets[bname] + r0)
                R = lib.query("tuber_colsum_blocks", M)
                lib.call("tuber_colsum", gb, workspace(dev, "cs", R * N), gbias, 1, M, N, ldg)
        dx = None
        if ctx.needs_input_grad[0]:
            toff, NN, KK, ldt = store.tinfo[wname]
            wt = store.tshadow.data_ptr() + 2 * (toff + r0)       # W^T[:, r0:r1]: column offset, ld = ldt
            # reduction runs over the (padded) output features; padded columns of both operands are zero
            Kred = Np if (r0 == 0 and Np <= ldt) else N
            assert Kred % 64 == 0, "row slices must be multiples of 64"
            dx = torch.empty(M, K, dtype=BF, device=dev)
            lib.call("tuber_gemm_nt", gb, ldg, wt, ldt, dx, K, M, K, Kred, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                     0, None, None, 0, 0, 0, None, None, None, 0, None, None, 1.0, 0.0, None, 0)
        return dx, None, None, None, None, None, None, None, None


def linear(x, store, wname, bname=None, rows=None, relu=False, out_f32=False, anchor=None):
    N = store.module.get_parameter(wname).shape[0] if rows is None else None
    r0, r1 = (0, N) if rows is None else rows
    return LinearFn.apply(x, anchor if anchor is not None else getattr(store, 'anchor', None), store, wname, bname, r0, r1, relu, out_f32)


class LayerNormFn(torch.autograd.Function):
    """y = LayerNorm(x + res) (res optional); gamma/beta are parameters in the store."""

    @staticmethod
    def forward(ctx, x, res, store, prefix):
        M, E = x.shape
        dev = x.device
        gamma = store.flat.data_ptr() + 4 * store.offsets[prefix + ".weight"]
        beta = store.flat.data_ptr() + 4 * store.offsets[prefix + ".bias"]
        y = torch.empty(M, E, dtype=BF, device=dev)
        need = any(ctx.needs_input_grad[:2]) or True
        xhat = torch.empty(M, E, dtype=BF, device=dev) if need else None
        rstd = torch.empty(M, dtype=torch.float32, device=dev) if need else None
        lib.call("tuber_layernorm_fwd", x, res, gamma, beta, y, xhat, rstd, M, E, 1e-5)
        ctx.store, ctx.prefix, ctx.has_res = store, prefix, res is not None
        ctx.save_for_backward(xhat, rstd)
        return y

    @staticmethod
    def backward(ctx, g):
        store, prefix = ctx.store, ctx.prefix
        xhat, rstd = ctx.saved_tensors
        M, E = xhat.shape
        dev = xhat.device
        g = g.contiguous()
        gamma = store.flat.data_ptr() + 4 * store.offsets[prefix + ".weight"]
        dgamma = store.gflat.data_ptr() + 4 * store.offsets[prefix + ".weight"]
        dbeta = store.gflat.data_ptr() + 4 * store.offsets[prefix + ".bias"]
        nb = lib.query("tuber_layernorm_bwd_blocks", M)
        dx = torch.empty(M, E, dtype=BF, device=dev)
        lib.call("tuber_layernorm_bwd", g, xhat, rstd, gamma, dx, workspace(dev, "ln", 2 * nb * E), dgamma, dbeta, 1, M, E)
        return dx, (dx if ctx.has_res else None), None, None


def layer_norm(x, res, store, prefix):
    return LayerNormFn.apply(x, res, store, prefix)


def _map(ld, sL, s1=0, s2=0, B2=1):
    return np.array([ld, sL, s1, s2, B2], dtype=np.int64)


class AttentionFn(torch.autograd.Function):
    """Multi-head attention core on packed projections.

    ``tensors`` are the distinct 2-D bf16 inputs; ``roles`` = ((ti, col_off), (ti, col_off), (ti, col_off)) says which
    tensor / column offset holds Q, K, V;  ``geom`` = (B, H, Lq, Lk, (sL,s1,s2,B2) for q rows, same for k/v rows);
    the output is [q_rows, H*32] in the q row order."""

    @staticmethod
    def forward(ctx, store, roles, geom, kpm, pdrop, seed, *tensors):
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
        lib.call("tuber_attn_fwd", tq.data_ptr() + 2 * qo, mq.ctypes.data, tk.data_ptr() + 2 * ko, mk.ctypes.data,
                 tv.data_ptr() + 2 * vo, mv.ctypes.data, o, mo.ctypes.data, lse, kpm, B, H, Lq, Lk, scale, float(pdrop), store.seed, int(seed))
        ctx.meta = (roles, geom, pdrop, seed, scale)
        ctx.kpm, ctx.seed_t = kpm, store.seed
        ctx.save_for_backward(o, lse, *tensors)
        return o

    @staticmethod
    def backward(ctx, g):
        roles, geom, pdrop, seed, scale = ctx.meta
        B, H, Lq, Lk, qmap, kmap = geom
        o, lse, *tensors = ctx.saved_tensors
        (qi, qo), (ki, ko), (vi, vo) = roles
        tq, tk, tv = tensors[qi], tensors[ki], tensors[vi]
        dev = tq.device
        E = H * 32
        g = g.contiguous()
        grads = [torch.empty_like(t) for t in tensors]
        covered = [0] * len(tensors)
        for ti, _ in roles:
            covered[ti] += E
        for i, t in enumerate(tensors):
            if covered[i] < t.shape[1]:
                grads[i].zero_()
        mq, mk, mv = _map(tq.shape[1], *qmap), _map(tk.shape[1], *kmap), _map(tv.shape[1], *kmap)
        mo = _map(E, *qmap)
        delta = torch.empty(B, H, Lq, dtype=torch.float32, device=dev)
        lib.call("tuber_attn_bwd", tq.data_ptr() + 2 * qo, mq.ctypes.data, tk.data_ptr() + 2 * ko, mk.ctypes.data,
                 tv.data_ptr() + 2 * vo, mv.ctypes.data, o, mo.ctypes.data, lse, ctx.kpm, g, mo.ctypes.data,
                 grads[qi].data_ptr() + 2 * qo, mq.ctypes.data, grads[ki].data_ptr() + 2 * ko, mk.ctypes.data,
                 grads[vi].data_ptr() + 2 * vo, mv.ctypes.data, delta, B, H, Lq, Lk, scale, float(pdrop), ctx.seed_t, int(seed))
        return (None, None, None, None, None, None) + tuple(grads)


def attention(store, roles, geom, kpm, pdrop, seed, *tensors):
    return AttentionFn.apply(store, roles, geom, kpm, pdrop, seed, *tensors)


class AxpbyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        out = torch.empty_like(a)
        lib.call("tuber_axpby", a, b, out, a.numel(), 1.0, 1.0)
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return AxpbyFn.apply(a, b)


class GatherSumFn(torch.autograd.Function):
    """out[(a,b,c)] = mul * sum_d in[a*sa+b*sb+c*sc+d*sd]; backward is the same kernel with the adjoint index map
    given explicitly (``bwd`` = (A,B,C,D,sa,sb,sc,sd,mul) producing the input-shaped gradient)."""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        A, B, C, D, sa, sb, sc, sd, mul = fwd
        E = x.shape[1]
        out = torch.empty(A * B * C, E, dtype=BF, device=x.device)
        lib.call("tuber_rows_gather_sum", x, out, A, B, C, D, sa, sb, sc, sd, E, float(mul))
        ctx.bwd, ctx.rows = bwd, x.shape[0]
        return out

    @staticmethod
    def backward(ctx, g):
        A, B, C, D, sa, sb, sc, sd, mul = ctx.bwd
        g = g.contiguous()
        E = g.shape[1]
        assert A * B * C == ctx.rows
        dx = torch.empty(ctx.rows, E, dtype=BF, device=g.device)
        lib.call("tuber_rows_gather_sum", g, dx, A, B, C, D, sa, sb, sc, sd, E, float(mul))
        return dx, None, None


def gather_sum(x, fwd, bwd):
    return GatherSumFn.apply(x, fwd, bwd)


class DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p, seed_t, salt):
        y = torch.empty_like(x)
        lib.call("tuber_dropout", x, y, x.numel(), float(p), seed_t, int(salt))
        ctx.p, ctx.seed_t, ctx.salt = p, seed_t, salt
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        dx = torch.empty_like(g)
        lib.call("tuber_dropout", g, dx, g.numel(), float(ctx.p), ctx.seed_t, int(ctx.salt))
        return dx, None, None, None


def dropout(x, p, training, store):
    if not training or p <= 0.0:
        return x
    store.step_seed += 1
    return DropoutFn.apply(x, p, store.seed, store.step_seed)


class SigmoidFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = torch.empty_like(x)
        lib.call("tuber_sigmoid_fwd", x, y, x.numel())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = g.contiguous()
        dx = torch.empty_like(y)
        lib.call("tuber_sigmoid_bwd", g, y, dx, y.numel())
        return dx


def sigmoid(x):
    return SigmoidFn.apply(x)


class ParamRowsFn(torch.autograd.Function):
    """bf16 rows of an embedding-like parameter [Q, E], repeated ``B`` times: rows (b, q)."""

    @staticmethod
    def forward(ctx, anchor, store, name, B):
        Q, E = store.module.get_parameter(name).shape
        src = store.shadow.data_ptr() + 2 * store.offsets[name]
        out = torch.empty(B * Q, E, dtype=BF, device=store.device)
        lib.call("tuber_rows_gather_sum", src, out, B, 1, Q, 1, 0, 0, 1, 0, E, 1.0)
        ctx.store, ctx.meta = store, (name, B, Q, E)
        return out

    @staticmethod
    def backward(ctx, g):
        store = ctx.store
        name, B, Q, E = ctx.meta
        g = g.contiguous()
        gp = store.gflat.data_ptr() + 4 * store.offsets[name]
        R = lib.query("tuber_colsum_blocks", B)
        lib.call("tuber_colsum", g, workspace(g.device, "cs", R * Q * E), gp, 1, B, Q * E, Q * E)
        return None, None, None, None


def param_rows(store, name, B, anchor):
    return ParamRowsFn.apply(anchor, store, name, B)


class BackboneFn(torch.autograd.Function):
    """CSN body as one autograd node: forward/backward are the hand-scheduled kernel sequences of CSNRunner."""

    @staticmethod
    def forward(ctx, clips, anchor, runner, train):
        feat, saved = runner.forward(clips, train)
        ctx.runner, ctx.saved = runner, saved
        runner.last_shape = tuple(feat.shape)
        return feat.view(-1, feat.shape[-1])

    @staticmethod
    def backward(ctx, g):
        ctx.runner.backward(ctx.saved, g.contiguous())
        ctx.saved = None
        return None, None, None, None


class AttentionWideFn(torch.autograd.Function):
    """LSTR pooling attention: one query per pixel, 8 heads of 256 (q [NQ,2048]; kv [rows,4096] = [k|v])."""

    @staticmethod
    def forward(ctx, q, kv, HW, T, pdrop, seed_t, salt):
        NQ = q.shape[0]
        o = torch.empty(NQ, 2048, dtype=BF, device=q.device)
        lib.call("tuber_attn_wide_fwd", q, kv, o, NQ, HW, T, float(pdrop), seed_t, int(salt))
        ctx.meta = (NQ, HW, T, pdrop, salt)
        ctx.seed_t = seed_t
        ctx.save_for_backward(q, kv)
        return o

    @staticmethod
    def backward(ctx, g):
        NQ, HW, T, pdrop, salt = ctx.meta
        q, kv = ctx.saved_tensors
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        lib.call("tuber_attn_wide_bwd", q, kv, g.contiguous(), dq, dkv, NQ, HW, T, float(pdrop), ctx.seed_t, int(salt))
        return dq, dkv, None, None, None, None, None


def attention_wide(store, q, kv, HW, T, pdrop, salt):
    return AttentionWideFn.apply(q, kv, HW, T, pdrop, store.seed, salt)
