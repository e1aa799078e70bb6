"""Per-kernel parity of the HIP kernels (called through the C ABI) against fp32 torch math on the
same bf16-rounded inputs.  Tolerances: bf16 outputs are compared at 2^-7 relative to the output scale
(one bf16 rounding + fp32 accumulation-order noise); fp32 outputs / statistics at 1e-3 relative."""
import math

import os

import pytest
import torch
import torch.nn.functional as F

from tubelet_transformer_amd import lib

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def rnd(*shape, dev, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dev)


def bfr(x):
    """round to bf16 and back (the value the kernel actually sees)."""
    return x.to(BF).float()


def close(name, got, ref, rel=2 ** -7, abs_=None):
    got, ref = got.float(), ref.float()
    scale = float(ref.abs().max()) + 1e-12
    err = float((got - ref).abs().max())
    tol = rel * scale if abs_ is None else abs_
    print("%-46s max|err| %.3e  (scale %.3e, tol %.3e)" % (name, err, scale, tol))
    assert math.isfinite(err) and err <= tol, "%s: err %.3e > tol %.3e" % (name, err, tol)


class rows64:
    """reference statistics rows in the 64-row layout: tuber_gemm_nt takes 96-row tiles for plain-A shapes with >= 8 192 rows (round 6), whose partial rows
    group the output rows differently (same column sums); kernels that are compared ROW BY ROW with tuber_gemm_nt switch that off for the reference call"""
    def __enter__(self):
        lib.query("tuber_gemm_nt_96_set", 0)

    def __exit__(self, *a):
        lib.query("tuber_gemm_nt_96_set", 1)


def gemm_nt(A, B, M, N, K, amode=0, sc=None, sh=None, gather=None, epi=0, bias=None, R=None, relu=0, out_f32=0,
            Cm=None, msc=None, msh=None, lda=None):
    dev = A.device
    C = torch.empty(M, N, device=dev, dtype=torch.float32 if out_f32 else BF)
    rows = lib.query("tuber_gemm_nt_stat_rows", M, N)
    st0 = torch.zeros(rows, N, device=dev) if epi else None
    st1 = torch.zeros(rows, N, device=dev) if epi else None
    g = gather or (0, 0, 0, 0, 0, 0, 0, 0)
    lib.call("tuber_gemm_nt", A, lda or K, B, K, C, N, M, N, K, amode, sc, sh, 1 if gather else 0, *g, epi, bias, R, N, relu,
             out_f32, st0, st1, Cm, N, msc, msh, 1.0, 0.0, None, 0, None, 0, None)
    return C, st0, st1


@pytest.mark.parametrize("M,N,K", [(70000, 64, 64), (40000, 128, 256), (5632, 256, 1024), (704, 768, 256), (30, 256, 256),
                                   (300, 80, 256), (30, 4, 256), (1000, 2048, 512), (44032, 512, 128)])
def test_gemm_nt_plain(dev, M, N, K):
    A = rnd(M, K, dev=dev, seed=1).to(BF)
    B = rnd(N, K, dev=dev, seed=2, scale=K ** -0.5).to(BF)
    bias = rnd(N, dev=dev, seed=3)
    ref = A.float() @ B.float().t() + bias
    C, _, _ = gemm_nt(A, B, M, N, K, bias=bias)
    close("gemm_nt bias %dx%dx%d" % (M, N, K), C, ref)
    C, _, _ = gemm_nt(A, B, M, N, K, bias=bias, relu=1, out_f32=1)
    close("gemm_nt bias+relu f32 %dx%dx%d" % (M, N, K), C, ref.relu(), rel=1e-3)
    R = rnd(M, N, dev=dev, seed=4).to(BF)
    C, _, _ = gemm_nt(A, B, M, N, K, R=R)
    close("gemm_nt +residual %dx%dx%d" % (M, N, K), C, A.float() @ B.float().t() + R.float())


@pytest.mark.parametrize("M,N,K", [(70000, 64, 256), (40000, 128, 512), (5000, 256, 64), (3000, 64, 128)])
def test_gemm_nt_bn_prologue_stats(dev, M, N, K):
    A = rnd(M, K, dev=dev, seed=1).to(BF)
    B = rnd(N, K, dev=dev, seed=2, scale=K ** -0.5).to(BF)
    sc = 1.0 + 0.2 * rnd(K, dev=dev, seed=5)
    sh = 0.3 * rnd(K, dev=dev, seed=6)
    a = bfr((A.float() * sc + sh).relu())
    ref = a @ B.float().t()
    C, st0, st1 = gemm_nt(A, B, M, N, K, amode=1, sc=sc, sh=sh, epi=1)
    close("gemm_nt bn_relu+stats out", C, ref)
    close("gemm_nt stats sum", st0.sum(0), ref.sum(0), rel=2e-3, abs_=2e-3 * float(ref.abs().sum(0).max()))
    close("gemm_nt stats sumsq", st1.sum(0), (ref * ref).sum(0), rel=2e-3)


@pytest.mark.parametrize("cfg", [7, 13, 0] + ([2, 12, 17] if os.environ.get("TUBER_AB_VARIANTS") else []))
@pytest.mark.parametrize("M,N,K", [(5632, 256, 1024), (5632, 1024, 256), (700, 128, 192), (130, 64, 64)])
def test_gemm_nt_forced_tile_configs(dev, cfg, M, N, K):
    """every tile configuration of the product library (13: 64x64, 7: 64x128, 0: 128x128 -- forced here also for the shapes that would
    not pick them; the rejected A/B variants 2 / 12 / 17 join when library and test run are built / started with TUBER_AB_VARIANTS=1)
    through all prologue / epilogue variants, incl. an odd number of k-tiles and a single tile"""
    assert lib.query("tuber_gemm_nt_has_cfg", cfg), "tile variant %d is not in this build of libtuber_hip.so" % cfg
    A = rnd(M, K, dev=dev, seed=1).to(BF)
    B = rnd(N, K, dev=dev, seed=2, scale=K ** -0.5).to(BF)
    sc, sh = 1.0 + 0.2 * rnd(K, dev=dev, seed=5), 0.3 * rnd(K, dev=dev, seed=6)
    bias = rnd(N, dev=dev, seed=3)
    R = rnd(M, N, dev=dev, seed=4).to(BF)
    Cm = rnd(M, N, dev=dev, seed=7).to(BF)
    lib.call("tuber_gemm_nt_set_cfg", cfg)
    try:
        ref = A.float() @ B.float().t()
        C, _, _ = gemm_nt(A, B, M, N, K, bias=bias, R=R, relu=1)
        close("cfg%d plain" % cfg, C, (ref + bias + R.float()).relu())
        a = bfr((A.float() * sc + sh).relu())
        ref1 = a @ B.float().t()
        C, st0, st1 = gemm_nt(A, B, M, N, K, amode=1, sc=sc, sh=sh, epi=1)
        close("cfg%d bn+stats out" % cfg, C, ref1)
        close("cfg%d stats sum" % cfg, st0.sum(0), ref1.sum(0), abs_=2e-3 * float(ref1.abs().sum(0).max()))
        close("cfg%d stats sumsq" % cfg, st1.sum(0), (ref1 * ref1).sum(0), rel=2e-3)
        C, st0, st1 = gemm_nt(A, B, M, N, K, epi=2, Cm=Cm)
        refm = ref * (Cm.float() > 0)
        close("cfg%d masked out" % cfg, C, refm)
        close("cfg%d masked sum" % cfg, st0.sum(0), refm.sum(0), abs_=2e-3 * float(refm.abs().sum(0).max()))
        close("cfg%d masked sum dz*c" % cfg, st1.sum(0), (refm * Cm.float()).sum(0), abs_=2e-3 * float((refm * Cm.float()).abs().sum(0).max()))
    finally:
        lib.call("tuber_gemm_nt_set_cfg", -1)


@pytest.mark.parametrize("M,N,K", [(5632, 256, 1024), (704, 256, 2048), (30, 256, 2048), (2816, 512, 2048), (700, 130, 1088), (64, 64, 1024)])
def test_gemm_nt_wave_split_k(dev, M, N, K):
    """the wave split-K form of the 64x64 kernel (taken automatically for plain A, >= 16 k-tiles, <= 512 tiles: every wave computes the
    whole tile for every fourth k-tile, the partial tiles are summed through LDS) with all four epilogues, against fp32 and against
    the shared-tile kernel (forced cfg 13) -- same products, a different summation order: agreement to accumulation rounding.
    Ragged M / N (700 x 130) and a k-tile count (17) that is not a multiple of the four waves included."""
    A = rnd(M, K, dev=dev, seed=1).to(BF)
    B = rnd(N, K, dev=dev, seed=2, scale=K ** -0.5).to(BF)
    bias = rnd(N, dev=dev, seed=3)
    R = rnd(M, N, dev=dev, seed=4).to(BF)
    Cm = rnd(M, N, dev=dev, seed=7).to(BF)
    Y = rnd(M, N, dev=dev, seed=8).to(BF)
    ref = A.float() @ B.float().t()

    def run_all():
        out = {}
        out["plain"] = gemm_nt(A, B, M, N, K, bias=bias, R=R, relu=1)[0]
        out["stats"] = gemm_nt(A, B, M, N, K, epi=1)
        out["bwd"] = gemm_nt(A, B, M, N, K, epi=2, Cm=Cm)
        rows = lib.query("tuber_gemm_nt_stat_rows", M, N)
        dz, b0, b1 = torch.empty(M, N, device=dev, dtype=BF), torch.zeros(rows, N, device=dev), torch.zeros(rows, N, device=dev)
        lib.call("tuber_gemm_nt_join", A, K, B, K, dz, N, M, N, K, R, N, Y, N, Cm, N, b0, b1)
        out["join"] = (dz, b0, b1)
        return out
    got = run_all()                                       # automatic choice: wave split-K
    lib.call("tuber_gemm_nt_set_cfg", 13)
    try:
        base = run_all()                                  # shared-tile kernel
    finally:
        lib.call("tuber_gemm_nt_set_cfg", -1)
    close("wsk plain", got["plain"], (ref + bias + R.float()).relu())
    close("wsk plain vs shared tile", got["plain"], base["plain"], rel=2 ** -8)
    close("wsk stats out", got["stats"][0], ref)
    close("wsk stats sum", got["stats"][1].sum(0), ref.sum(0), abs_=2e-3 * float(ref.abs().sum(0).max()))
    close("wsk stats sumsq", got["stats"][2].sum(0), (ref * ref).sum(0), rel=2e-3)
    tr = lib.query("tuber_gemm_nt_wsk_tile_rows", M, N, K)
    assert tr == (96 if (M, N, K) in ((5632, 256, 1024), (2816, 512, 2048)) else 64), tr      # the layer3 / layer4 shapes take the 96-row tiles
    if tr == 64:
        close("wsk stats rows vs shared tile", got["stats"][1], base["stats"][1], rel=1e-4, abs_=1e-3 * float(base["stats"][1].abs().max()))
    else:
        # 96-row tiles write ceil(M / 96) partial rows, the remaining rows of the [ceil(M / 64)][N] buffer as zero: blocks of 192 output rows
        # (2 rows here, 3 of the shared-tile kernel) must agree, and the padding must really be zero
        n96 = (M + 95) // 96
        assert float(got["stats"][1][n96:].abs().max()) == 0.0 and float(got["stats"][2][n96:].abs().max()) == 0.0
        blk = lambda t, k: torch.stack([t[i:i + k].sum(0) for i in range(0, (t.shape[0] // k) * k, k)])
        nb = M // 192
        for j in (1, 2):
            a, b = blk(got["stats"][j][:n96], 2)[:nb], blk(base["stats"][j], 3)[:nb]
            close("wsk 96-row stats blocks vs shared tile", a, b, rel=1e-4, abs_=1e-3 * float(b.abs().max()))
        a, b = blk(got["bwd"][1][:n96], 2)[:nb], blk(base["bwd"][1], 3)[:nb]
        close("wsk 96-row masked stats blocks vs shared tile", a, b, rel=1e-4, abs_=1e-3 * float(b.abs().max()))
    refm = ref * (Cm.float() > 0)
    close("wsk masked out", got["bwd"][0], refm)
    close("wsk masked sum dz*c", got["bwd"][2].sum(0), (refm * Cm.float()).sum(0), abs_=2e-3 * float((refm * Cm.float()).abs().sum(0).max()))
    refj = bfr(ref + R.float()) * (Y.float() > 0)
    close("wsk join dz", got["join"][0], refj)
    close("wsk join vs shared tile", got["join"][0], base["join"][0], rel=2 ** -8)
    close("wsk join sum dz*c", got["join"][2].sum(0), (refj * Cm.float()).sum(0), abs_=2e-3 * float((refj * Cm.float()).abs().sum(0).max()))


def test_gemm_nt_gather(dev):
    n, Ti, Hi, Wi, K, N = 2, 8, 15, 21, 256, 512
    st, ss = 2, 2
    To, Ho, Wo = (Ti - 1) // st + 1, (Hi - 1) // ss + 1, (Wi - 1) // ss + 1
    X = rnd(n, Ti, Hi, Wi, K, dev=dev, seed=1).to(BF)
    B = rnd(N, K, dev=dev, seed=2, scale=K ** -0.5).to(BF)
    Xs = X[:, ::st, ::ss, ::ss].reshape(-1, K)
    M = Xs.shape[0]
    assert M == n * To * Ho * Wo
    C, st0, _ = gemm_nt(X, B, M, N, K, gather=(To, Ho, Wo, Ti, Hi, Wi, st, ss), epi=1)
    ref = Xs.float() @ B.float().t()
    close("gemm_nt strided gather", C, ref)
    close("gemm_nt strided gather stats", st0.sum(0), ref.sum(0), abs_=2e-3 * float(ref.abs().sum(0).max()))


@pytest.mark.parametrize("M,N,K", [(20000, 64, 256), (5000, 256, 1024), (900, 2048, 256)])
def test_gemm_nt_bwd_epilogue(dev, M, N, K):
    G = rnd(M, K, dev=dev, seed=1).to(BF)
    WT = rnd(N, K, dev=dev, seed=2, scale=K ** -0.5).to(BF)
    Cm = rnd(M, N, dev=dev, seed=3).to(BF)
    msc = 1.0 + 0.2 * rnd(N, dev=dev, seed=5)
    msh = 0.3 * rnd(N, dev=dev, seed=6)
    da = G.float() @ WT.float().t()
    mask = (Cm.float() * msc + msh) > 0
    ref = da * mask
    C, st0, st1 = gemm_nt(G, WT, M, N, K, epi=2, Cm=Cm, msc=msc, msh=msh)
    close("gemm_nt bwd-mask out", C, ref)
    close("gemm_nt bwd-mask sum dz", st0.sum(0), ref.sum(0), abs_=2e-3 * float(ref.abs().sum(0).max()))
    close("gemm_nt bwd-mask sum dz*c", st1.sum(0), (ref * Cm.float()).sum(0), abs_=2e-3 * float((ref * Cm.float()).abs().sum(0).max()))


@pytest.mark.parametrize("M,N,K,amode", [(70000, 256, 64, 0), (40000, 128, 512, 1), (5632, 1024, 256, 1), (704, 256, 2048, 0),
                                         (30, 4, 256, 0), (5000, 64, 448, 0), (999, 80, 256, 0)])
def test_gemm_tn(dev, M, N, K, amode):
    G = rnd(M, N, dev=dev, seed=1).to(BF)
    A = rnd(M, K, dev=dev, seed=2).to(BF)
    sc = 1.0 + 0.2 * rnd(K, dev=dev, seed=5)
    sh = 0.3 * rnd(K, dev=dev, seed=6)
    a = bfr((A.float() * sc + sh).relu()) if amode else A.float()
    ref = G.float().t() @ a
    S = lib.query("tuber_gemm_tn_slabs", M, N, K)
    part = torch.empty(S, N, K, device=dev)
    out = torch.zeros(N, K, device=dev)
    lib.call("tuber_gemm_tn", G, N, A, K, part, out, 0, M, N, K, amode, sc if amode else None, sh if amode else None,
             0, 0, 0, 0, 0, 0, 0, 0, 0, None, 0, None, None, None, None)
    close("gemm_tn %dx%dx%d amode %d" % (M, N, K, amode), out, ref, rel=2e-3)
    lib.call("tuber_gemm_tn", G, N, A, K, part, out, 1, M, N, K, amode, sc if amode else None, sh if amode else None,
             0, 0, 0, 0, 0, 0, 0, 0, 0, None, 0, None, None, None, None)
    close("gemm_tn accumulate", out, 2 * ref, rel=2e-3)


def test_gemm_tn_gather(dev):
    n, Ti, Hi, Wi, K, N = 2, 8, 15, 21, 256, 512
    st, ss = 2, 2
    To, Ho, Wo = (Ti - 1) // st + 1, (Hi - 1) // ss + 1, (Wi - 1) // ss + 1
    X = rnd(n, Ti, Hi, Wi, K, dev=dev, seed=1).to(BF)
    Xs = X[:, ::st, ::ss, ::ss].reshape(-1, K)
    M = Xs.shape[0]
    G = rnd(M, N, dev=dev, seed=3).to(BF)
    S = lib.query("tuber_gemm_tn_slabs", M, N, K)
    part = torch.empty(S, N, K, device=dev)
    out = torch.zeros(N, K, device=dev)
    lib.call("tuber_gemm_tn", G, N, X, K, part, out, 0, M, N, K, 0, None, None, 1, To, Ho, Wo, Ti, Hi, Wi, st, ss, None, 0, None, None, None, None)
    close("gemm_tn gather", out, G.float().t() @ Xs.float(), rel=2e-3)


def dw_ref(x_raw, sc, sh, w, st, ss):
    """x_raw [N,T,H,W,C] bf16 -> fp32 NCDHW conv of relu(bn(x))."""
    a = bfr(x_raw.float() * sc + sh).relu() if sc is not None else x_raw.float()
    a = a.permute(0, 4, 1, 2, 3)
    C = a.shape[1]
    return F.conv3d(a, w.view(C, 1, 3, 3, 3), stride=(st, ss, ss), padding=1, groups=C)


@pytest.mark.parametrize("N,T,H,W,C,st,ss", [(2, 8, 16, 21, 64, 1, 1), (1, 8, 17, 22, 128, 2, 2), (2, 4, 9, 11, 256, 2, 1),
                                             (1, 4, 8, 8, 512, 1, 1), (1, 6, 13, 43, 64, 2, 2)])
def test_dwconv(dev, N, T, H, W, C, st, ss):
    x = rnd(N, T, H, W, C, dev=dev, seed=1).to(BF)
    w = rnd(C, 27, dev=dev, seed=2, scale=27 ** -0.5)
    sc = 1.0 + 0.2 * rnd(C, dev=dev, seed=5)
    sh = 0.3 * rnd(C, dev=dev, seed=6)
    To, Ho, Wo = (T - 1) // st + 1, (H - 1) // ss + 1, (W - 1) // ss + 1
    # NB the kernel computes relu(fma(x, sc, sh)) in fp32 without rounding the activation to bf16
    a = (x.float() * sc + sh).relu().permute(0, 4, 1, 2, 3).requires_grad_(True)
    wp = w.clone().requires_grad_(True)
    ref = F.conv3d(a, wp.view(C, 1, 3, 3, 3), stride=(st, ss, ss), padding=1, groups=C)
    assert ref.shape[2:] == (To, Ho, Wo)
    out = torch.empty(N, To, Ho, Wo, C, device=dev, dtype=BF)
    R = lib.query("tuber_dwconv_fwd_stat_rows", N, To, Ho, Wo)
    st0, st1 = torch.zeros(R, C, device=dev), torch.zeros(R, C, device=dev)
    lib.call("tuber_dwconv_fwd", x, sc, sh, w, out, st0, st1, N, T, H, W, To, Ho, Wo, C, st, ss)
    refl = ref.detach().permute(0, 2, 3, 4, 1)
    close("dwconv fwd", out, refl)
    close("dwconv fwd stats sum", st0.sum(0), refl.sum((0, 1, 2, 3)), abs_=2e-3 * float(refl.abs().sum((0, 1, 2, 3)).max()))
    close("dwconv fwd stats sumsq", st1.sum(0), (refl ** 2).sum((0, 1, 2, 3)), rel=2e-3)
    # no-prologue variant
    out2 = torch.empty_like(out)
    lib.call("tuber_dwconv_fwd", x, None, None, w, out2, None, None, N, T, H, W, To, Ho, Wo, C, st, ss)
    close("dwconv fwd (no bn)", out2, F.conv3d(x.float().permute(0, 4, 1, 2, 3), w.view(C, 1, 3, 3, 3), stride=(st, ss, ss),
                                                 padding=1, groups=C).permute(0, 2, 3, 4, 1))
    # backward
    g = rnd(N, To, Ho, Wo, C, dev=dev, seed=9).to(BF)
    ref.backward(g.float().permute(0, 4, 1, 2, 3))
    da = a.grad.permute(0, 2, 3, 4, 1)                 # grad wrt relu(bn(x)) includes relu mask already (a is post-relu leaf)
    mask = (x.float() * sc + sh) > 0
    dz_ref = da * mask
    dz = torch.empty(N, T, H, W, C, device=dev, dtype=BF)
    R2 = lib.query("tuber_dwconv_bwd_data_stat_rows", N, T, H, W)
    s0, s1 = torch.zeros(R2, C, device=dev), torch.zeros(R2, C, device=dev)
    lib.call("tuber_dwconv_bwd_data", g, w, x, sc, sh, dz, s0, s1, N, T, H, W, To, Ho, Wo, C, st, ss)
    close("dwconv bwd data", dz, dz_ref)
    close("dwconv bwd data sum dz", s0.sum(0), dz_ref.sum((0, 1, 2, 3)), abs_=2e-3 * float(dz_ref.abs().sum((0, 1, 2, 3)).max()))
    close("dwconv bwd data sum dz*x", s1.sum(0), (dz_ref * x.float()).sum((0, 1, 2, 3)),
          abs_=2e-3 * float((dz_ref * x.float()).abs().sum((0, 1, 2, 3)).max()))
    nb = lib.query("tuber_dwconv_bwd_weight_blocks", N, To, Ho, Wo)
    part = torch.empty(nb, 27, C, device=dev)
    dw = torch.zeros(C, 27, device=dev)
    lib.call("tuber_dwconv_bwd_weight", g, x, sc, sh, part, dw, 0, N, T, H, W, To, Ho, Wo, C, st, ss)
    close("dwconv bwd weight", dw, wp.grad, rel=2e-3)


@pytest.mark.parametrize("N,T,H,W,C", [(2, 8, 16, 22, 256), (1, 5, 13, 43, 128), (2, 9, 24, 37, 64), (1, 1, 3, 5, 64), (1, 4, 16, 22, 512)])
def test_dwconv_tile(dev, N, T, H, W, C):
    """LDS-staged stride-1 depthwise kernels (ragged 8x16 tiles, several plane chunks) against F.conv3d and its autograd."""
    x = rnd(N, T, H, W, C, dev=dev, seed=1).to(BF)
    w = rnd(C, 27, dev=dev, seed=2, scale=27 ** -0.5)
    sc = 1.0 + 0.2 * rnd(C, dev=dev, seed=5)
    sh = 0.3 * rnd(C, dev=dev, seed=6)
    a = (x.float() * sc + sh).relu().permute(0, 4, 1, 2, 3).requires_grad_(True)
    wp = w.clone().requires_grad_(True)
    ref = F.conv3d(a, wp.view(C, 1, 3, 3, 3), stride=1, padding=1, groups=C)
    out = torch.empty(N, T, H, W, C, device=dev, dtype=BF)
    R = lib.query("tuber_dwconv_tile_blocks", N, T, H, W, C)
    st0, st1 = torch.zeros(R, C, device=dev), torch.zeros(R, C, device=dev)
    lib.call("tuber_dwconv_tile_fwd", x, sc, sh, w, out, st0, st1, N, T, H, W, C)
    refl = ref.detach().permute(0, 2, 3, 4, 1)
    close("dwconv tile fwd", out, refl)
    close("dwconv tile fwd stats sum", st0.sum(0), refl.sum((0, 1, 2, 3)), abs_=2e-3 * float(refl.abs().sum((0, 1, 2, 3)).max()))
    close("dwconv tile fwd stats sumsq", st1.sum(0), (refl ** 2).sum((0, 1, 2, 3)), rel=2e-3)
    out2 = torch.empty_like(out)
    lib.call("tuber_dwconv_tile_fwd", x, None, None, w, out2, None, None, N, T, H, W, C)
    close("dwconv tile fwd (no bn)", out2, F.conv3d(x.float().permute(0, 4, 1, 2, 3), w.view(C, 1, 3, 3, 3), padding=1,
                                                      groups=C).permute(0, 2, 3, 4, 1))
    g = rnd(N, T, H, W, C, dev=dev, seed=9).to(BF)
    ref.backward(g.float().permute(0, 4, 1, 2, 3))
    dz_ref = a.grad.permute(0, 2, 3, 4, 1) * ((x.float() * sc + sh) > 0)
    dz = torch.empty(N, T, H, W, C, device=dev, dtype=BF)
    s0, s1 = torch.zeros(R, C, device=dev), torch.zeros(R, C, device=dev)
    lib.call("tuber_dwconv_tile_bwd_data", g, w, x, sc, sh, dz, s0, s1, N, T, H, W, C)
    close("dwconv tile bwd data", dz, dz_ref)
    close("dwconv tile bwd data sum dz", s0.sum(0), dz_ref.sum((0, 1, 2, 3)), abs_=2e-3 * float(dz_ref.abs().sum((0, 1, 2, 3)).max()))
    close("dwconv tile bwd data sum dz*x", s1.sum(0), (dz_ref * x.float()).sum((0, 1, 2, 3)),
          abs_=2e-3 * float((dz_ref * x.float()).abs().sum((0, 1, 2, 3)).max()))
    part = torch.empty(lib.query("tuber_dwconv_tile_wgrad_blocks", N, T, H, W, C), 27, C, device=dev)
    dw = torch.ones(C, 27, device=dev)
    lib.call("tuber_dwconv_tile_bwd_weight", g, x, sc, sh, part, dw, 1, N, T, H, W, C)
    close("dwconv tile bwd weight", dw - 1, wp.grad, rel=2e-3)


@pytest.mark.parametrize("N,T,H,W,C,R", [(2, 8, 16, 22, 256, 88), (1, 5, 13, 43, 128, 128), (2, 9, 24, 37, 64, 3), (1, 4, 16, 22, 512, 150)])
def test_dwconv_tile_forward_finalises_the_batchnorm_in_front_of_it(dev, N, T, H, W, C, R):
    """tuber_dwconv_tile_fwd_bn = tuber_bn_finalize + tuber_dwconv_tile_fwd in one launch: every output of both -- the conv output, its
    statistics rows, scale / shift / mean / invstd, the running statistics and num_batches_tracked -- bit for bit (fp64 sums of the fp32
    partial rows are exact, and the finalisation is the same expression).  reference: nn.BatchNorm3d + ReLU + conv3, ir_CSN_152.py:46-51."""
    M = N * T * H * W
    x = rnd(N, T, H, W, C, dev=dev, seed=1, scale=1.5).to(BF)
    w = rnd(C, 27, dev=dev, seed=2, scale=27 ** -0.5)
    xf = x.float().view(M, C)
    bounds = torch.linspace(0, M, R + 1).long().tolist()
    p0 = torch.stack([xf[bounds[i]:bounds[i + 1]].sum(0) for i in range(R)])
    p1 = torch.stack([(xf[bounds[i]:bounds[i + 1]] ** 2).sum(0) for i in range(R)])
    gamma, beta = 1 + 0.1 * rnd(C, dev=dev, seed=3), 0.1 * rnd(C, dev=dev, seed=4)
    rm, rv = 0.1 * rnd(C, dev=dev, seed=5), 1 + 0.1 * rnd(C, dev=dev, seed=6).abs()
    nblk = lib.query("tuber_dwconv_tile_blocks", N, T, H, W, C)

    def run(fused):
        rm2, rv2, nbt = rm.clone(), rv.clone(), torch.full((1,), 41, dtype=torch.int64, device=dev)
        scale, shift, mean, invstd = (torch.full((C,), float("nan"), device=dev) for _ in range(4))
        out = torch.empty(N, T, H, W, C, device=dev, dtype=BF)
        st0, st1 = torch.zeros(nblk, C, device=dev), torch.zeros(nblk, C, device=dev)
        if fused:
            lib.call("tuber_dwconv_tile_fwd_bn", x, p0, p1, R, float(M), gamma, beta, rm2, rv2, nbt, 0.1, 1e-3, scale, shift, mean, invstd,
                     w, out, st0, st1, N, T, H, W, C)
        else:
            lib.call("tuber_bn_finalize", p0, p1, R, C, float(M), gamma, beta, rm2, rv2, nbt, 0.1, 1e-3, scale, shift, mean, invstd)
            lib.call("tuber_dwconv_tile_fwd", x, scale, shift, w, out, st0, st1, N, T, H, W, C)
        return dict(out=out, st0=st0, st1=st1, scale=scale, shift=shift, mean=mean, invstd=invstd, rmean=rm2, rvar=rv2, nbt=nbt)

    want, got = run(False), run(True)
    assert int(got["nbt"]) == 42
    for k in want:
        assert torch.equal(got[k], want[k]), "%s differs from finalize + forward (max |d| %.3g)" % (k, float((got[k].float() - want[k].float()).abs().max()))
    # without running statistics (rmean = rvar = nbt = NULL)
    scale, shift, mean, invstd = (torch.empty(C, device=dev) for _ in range(4))
    out = torch.empty(N, T, H, W, C, device=dev, dtype=BF)
    lib.call("tuber_dwconv_tile_fwd_bn", x, p0, p1, R, float(M), gamma, beta, None, None, None, 0.1, 1e-3, scale, shift, mean, invstd,
             w, out, None, None, N, T, H, W, C)
    assert torch.equal(out, want["out"]) and torch.equal(scale, want["scale"])


def test_bn_finalize_and_bwd(dev):
    M, C = 5000, 256
    x = rnd(M, C, dev=dev, seed=1, scale=2.0) + 0.5
    gamma, beta = 1 + 0.1 * rnd(C, dev=dev, seed=2), 0.1 * rnd(C, dev=dev, seed=3)
    rm, rv = 0.1 * rnd(C, dev=dev, seed=4), 1 + 0.1 * rnd(C, dev=dev, seed=5).abs()
    R = 7
    chunks = x.chunk(R, 0)
    st0 = torch.stack([c.sum(0) for c in chunks])
    st1 = torch.stack([(c * c).sum(0) for c in chunks])
    rm2, rv2 = rm.clone(), rv.clone()
    nbt = torch.zeros(1, dtype=torch.int64, device=dev)
    scale, shift, mean, invstd = (torch.empty(C, device=dev) for _ in range(4))
    lib.call("tuber_bn_finalize", st0, st1, R, C, float(M), gamma, beta, rm2, rv2, nbt, 0.1, 1e-3, scale, shift, mean, invstd)
    xr = x.clone().requires_grad_(True)
    g_ = gamma.clone().requires_grad_(True)
    b_ = beta.clone().requires_grad_(True)
    y = F.batch_norm(xr, rm, rv, g_, b_, True, 0.1, 1e-3)
    close("bn finalize apply", x * scale + shift, y.detach(), rel=1e-4)
    close("bn running_mean", rm2, rm, rel=1e-5)
    close("bn running_var", rv2, rv, rel=1e-5)
    assert int(nbt) == 1
    dz = rnd(M, C, dev=dev, seed=7)
    y.backward(dz)
    xb, dzb = x.to(BF), dz.to(BF)
    # backward coefficients from partial sums
    dchunks, xchunks = dz.chunk(R, 0), x.chunk(R, 0)
    b0 = torch.stack([c.sum(0) for c in dchunks])
    b1 = torch.stack([(c * d).sum(0) for c, d in zip(dchunks, xchunks)])
    cA, cB, cC, dg, db = (torch.zeros(C, device=dev) for _ in range(5))
    lib.call("tuber_bn_bwd_finalize", b0, b1, R, C, float(M), gamma, mean, invstd, cA, cB, cC, dg, db, 0)
    close("bn bwd dgamma", dg, g_.grad, rel=1e-3)
    close("bn bwd dbeta", db, b_.grad, rel=1e-3)
    close("bn bwd dx (fp32 coefficients)", cA * dz + cB * x + cC, xr.grad, rel=1e-3)
    dx = torch.empty(M, C, device=dev, dtype=BF)
    lib.call("tuber_bn_bwd_apply", dzb, xb, cA, cB, cC, dx, M, C)
    close("bn bwd apply kernel", dx, cA * dzb.float() + cB * xb.float() + cC)
    sc2, sh2 = torch.empty(C, device=dev), torch.empty(C, device=dev)
    lib.call("tuber_bn_eval_affine", gamma, beta, rm, rv, 1e-3, sc2, sh2, C)
    close("bn eval affine", x * sc2 + sh2, F.batch_norm(x, rm, rv, gamma, beta, False, 0.1, 1e-3), rel=1e-5)


@pytest.mark.parametrize("M,C,ds", [(3000, 256, False), (700, 2048, True), (5000, 64, True)])
def test_block_out(dev, M, C, ds):
    c4 = rnd(M, C, dev=dev, seed=1).to(BF)
    res = rnd(M, C, dev=dev, seed=2).to(BF)
    s4, h4 = 1 + 0.1 * rnd(C, dev=dev, seed=3), 0.1 * rnd(C, dev=dev, seed=4)
    rs, rh = (1 + 0.1 * rnd(C, dev=dev, seed=5), 0.1 * rnd(C, dev=dev, seed=6)) if ds else (None, None)
    y = torch.empty(M, C, device=dev, dtype=BF)
    lib.call("tuber_block_out_fwd", c4, s4, h4, res, rs, rh, y, M, C)
    r = res.float() * rs + rh if ds else res.float()
    ref = (c4.float() * s4 + h4 + r).relu()
    close("block_out fwd", y, ref)
    dy = rnd(M, C, dev=dev, seed=7).to(BF)
    dz = torch.empty(M, C, device=dev, dtype=BF)
    R = lib.query("tuber_rowblock_count", M, C)
    a, b, c = (torch.zeros(R, C, device=dev) for _ in range(3))
    lib.call("tuber_block_out_bwd", dy, y, c4, res if ds else None, dz, a, b, c if ds else None, M, C)
    dzr = dy.float() * (y.float() > 0)
    close("block_out bwd dz", dz, dzr)
    close("block_out bwd sum dz", a.sum(0), dzr.sum(0), abs_=2e-3 * float(dzr.abs().sum(0).max()))
    close("block_out bwd sum dz*c4", b.sum(0), (dzr * c4.float()).sum(0), abs_=2e-3 * float((dzr * c4.float()).abs().sum(0).max()))
    if ds:
        close("block_out bwd sum dz*ds", c.sum(0), (dzr * res.float()).sum(0), abs_=2e-3 * float((dzr * res.float()).abs().sum(0).max()))


@pytest.mark.parametrize("M,E", [(704, 256), (30, 256), (300, 2048)])
def test_layernorm(dev, M, E):
    x = rnd(M, E, dev=dev, seed=1).to(BF)
    r = rnd(M, E, dev=dev, seed=2).to(BF)
    g, b = 1 + 0.1 * rnd(E, dev=dev, seed=3), 0.1 * rnd(E, dev=dev, seed=4)
    y = torch.empty(M, E, device=dev, dtype=BF)
    xh = torch.empty(M, E, device=dev, dtype=BF)
    rstd = torch.empty(M, device=dev)
    lib.call("tuber_layernorm_fwd", x, r, g, b, y, E, xh, rstd, M, E, 1e-5, 0.0, None, 0)
    xin = (x.float() + r.float()).requires_grad_(True)
    gp, bp = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = F.layer_norm(xin, (E,), gp, bp, 1e-5)
    close("layernorm fwd", y, ref.detach())
    dy = rnd(M, E, dev=dev, seed=5).to(BF)
    ref.backward(dy.float())
    nb = lib.query("tuber_layernorm_bwd_blocks", M)
    part = torch.empty(2 * nb * E, device=dev)
    dx = torch.empty(M, E, device=dev, dtype=BF)
    # adjacent (one reduction launch) and separate dgamma / dbeta buffers
    dgb = torch.zeros(2 * E, device=dev)
    lib.call("tuber_layernorm_bwd", dy, E, xh, rstd, g, dx, None, part, dgb, dgb.data_ptr() + 4 * E, 0, M, E, 0.0, None, 0)
    close("layernorm bwd dx", dx, xin.grad, rel=2 ** -6)
    close("layernorm bwd dgamma", dgb[:E], gp.grad, rel=1e-2)
    close("layernorm bwd dbeta", dgb[E:], bp.grad, rel=1e-2)
    dg, db = torch.ones(E, device=dev), torch.ones(E, device=dev)
    lib.call("tuber_layernorm_bwd", dy, E, xh, rstd, g, dx, None, part, dg, db, 1, M, E, 0.0, None, 0)
    close("layernorm bwd dgamma (accumulate, separate)", dg - 1, gp.grad, rel=1e-2)
    close("layernorm bwd dbeta (accumulate, separate)", db - 1, bp.grad, rel=1e-2)
    y2 = torch.empty(M, E, device=dev, dtype=BF)
    lib.call("tuber_layernorm_fwd", x, None, g, b, y2, E, None, None, M, E, 1e-5, 0.0, None, 0)
    close("layernorm fwd (no res)", y2, F.layer_norm(x.float(), (E,), g, b, 1e-5))


@pytest.mark.parametrize("M,E", [(704, 256), (77, 2048)])
def test_layernorm_dropout_and_window(dev, M, E):
    """LayerNorm(Dropout(x) + res) written into a column window of a wider buffer; the mask is the one tuber_dropout makes."""
    p, salt = 0.1, 7
    seed = torch.tensor([1234], dtype=torch.int64, device=dev)
    x = rnd(M, E, dev=dev, seed=1).to(BF)
    r = rnd(M, E, dev=dev, seed=2).to(BF)
    g, b = 1 + 0.1 * rnd(E, dev=dev, seed=3), 0.1 * rnd(E, dev=dev, seed=4)
    ones = torch.ones(M, E, device=dev, dtype=BF)
    keep = torch.empty_like(ones)
    lib.call("tuber_dropout", ones, keep, M * E, p, seed, salt)
    keep = keep.float()                                       # 0 or 1/(1-p)
    assert 0.85 < float((keep > 0).float().mean()) < 0.95
    wide = torch.zeros(M + 3, 2 * E, device=dev, dtype=BF)
    xh = torch.empty(M, E, device=dev, dtype=BF)
    rstd = torch.empty(M, device=dev)
    lib.call("tuber_layernorm_fwd", x, r, g, b, wide.data_ptr() + 2 * (2 * (2 * E) + E), 2 * E, xh, rstd, M, E, 1e-5, p, seed, salt)
    xin = x.float().requires_grad_(True)
    rin = r.float().requires_grad_(True)
    ref = F.layer_norm(xin * keep + rin, (E,), g, b, 1e-5)
    close("layernorm(dropout) fwd window", wide[2:2 + M, E:], ref.detach())
    assert float(wide[:2].abs().max()) == 0 and float(wide[2:2 + M, :E].abs().max()) == 0 and float(wide[2 + M:].abs().max()) == 0
    dyw = rnd(M + 3, 2 * E, dev=dev, seed=5).to(BF)
    ref.backward(dyw[2:2 + M, E:].float())
    nb = lib.query("tuber_layernorm_bwd_blocks", M)
    part = torch.empty(2 * nb * E, device=dev)
    dx, dxd = torch.empty(M, E, device=dev, dtype=BF), torch.empty(M, E, device=dev, dtype=BF)
    dg, db = torch.zeros(E, device=dev), torch.zeros(E, device=dev)
    lib.call("tuber_layernorm_bwd", dyw.data_ptr() + 2 * (2 * (2 * E) + E), 2 * E, xh, rstd, g, dx, dxd, part, dg, db, 0, M, E, p, seed, salt)
    close("layernorm(dropout) bwd d res", dx, rin.grad, rel=2 ** -6)
    close("layernorm(dropout) bwd d x", dxd, xin.grad, rel=2 ** -6)


@pytest.mark.parametrize("M,Kin,epi,p,two", [(30, 256, "plain", 0.0, False), (30, 256, "res", 0.1, False), (30, 2048, "mask", 0.1, True),
                                              (2816, 256, "res", 0.1, False), (2816, 2048, "mask", 0.1, False), (2816, 256, "plain", 0.0, True),
                                              (77, 2048, "mask", 0.0, False), (100, 1024, "res", 0.1, True)])
def test_layernorm_backward_fused_into_the_linear_data_gradient(dev, M, Kin, epi, p, two):
    """tuber_ln_bwd_dx == [tuber_axpby] + tuber_layernorm_bwd + tuber_gemm_nt (the chain tape.py launches for norm(x + dropout(linear(.))) in
    the backward pass): dx / dxd identical up to one bf16 ulp on rare elements (same arithmetic, another compilation unit), the dgamma / dbeta
    partial rows summed over their blocks to 1e-5, the product to MFMA summation order; every epilogue of linear.bwd's data-gradient GEMM."""
    E, salt = 256, 13
    seed = torch.tensor([4321], dtype=torch.int64, device=dev)
    dy = rnd(M, E, dev=dev, seed=1).to(BF)
    dy2 = rnd(M, E, dev=dev, seed=2).to(BF) if two else None
    xh = rnd(M, E, dev=dev, seed=3).to(BF)
    rstd = (0.5 + torch.rand(M, generator=torch.Generator().manual_seed(11))).to(dev)
    gam = 1 + 0.1 * rnd(E, dev=dev, seed=4)
    wt = (rnd(Kin, E, dev=dev, seed=5) / 16).to(BF)           # W^T rows: [Kin][E]
    res = rnd(M, Kin, dev=dev, seed=6).to(BF) if epi == "res" else None
    cm = rnd(M, Kin, dev=dev, seed=7).to(BF) if epi == "mask" else None
    alpha = 1.25 if epi == "mask" else 1.0
    # the launch chain
    g = dy
    if two:
        g = torch.empty_like(dy)
        lib.call("tuber_axpby", dy, dy2, g, dy.numel(), 1.0, 1.0)
    nb0 = lib.query("tuber_layernorm_bwd_blocks", M)
    part0 = torch.zeros(nb0, 2 * E, device=dev)
    dx0, dxd0 = torch.empty(M, E, device=dev, dtype=BF), torch.empty(M, E, device=dev, dtype=BF)
    lib.call("tuber_layernorm_bwd", g, E, xh, rstd, gam, dx0, dxd0 if p > 0 else None, part0, None, None, 2, M, E, p, seed, salt)
    a0 = dxd0 if p > 0 else dx0
    out0 = torch.empty(M, Kin, device=dev, dtype=BF)
    if epi == "mask":
        lib.call("tuber_gemm_nt", a0, E, wt, E, out0, Kin, M, Kin, E, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                 2, None, None, 0, 0, 0, None, None, cm, Kin, None, None, alpha, 0.0, None, 0, None, 0, None)
    else:
        lib.call("tuber_gemm_nt", a0, E, wt, E, out0, Kin, M, Kin, E, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                 0, None, res, Kin, 0, 0, None, None, None, 0, None, None, 1.0, 0.0, None, 0, None, 0, None)
    # one launch
    nb1 = lib.query("tuber_ln_bwd_dx_blocks", M)
    part1 = torch.zeros(nb1, 2 * E, device=dev)
    dx1 = torch.full((M, E), float("nan"), device=dev, dtype=BF)
    dxd1 = torch.full((M, E), float("nan"), device=dev, dtype=BF)
    out1 = torch.full((M, Kin), float("nan"), device=dev, dtype=BF)
    assert lib.query("tuber_ln_bwd_dx_supported", E, Kin) == 1
    lib.call("tuber_ln_bwd_dx", dy, E, dy2, E if two else 0, xh, rstd, gam, dx1, dxd1 if p > 0 else None, part1, M, E, p, seed, salt,
             wt, E, Kin, out1, res, cm, alpha, None, 0, None, None, None, None)
    torch.cuda.synchronize()
    for name, a, b in (("dx", dx1, dx0),) + ((("dxd", dxd1, dxd0),) if p > 0 else ()):
        d = (a.float() - b.float()).abs()
        assert bool(torch.isfinite(a.float()).all()), name
        assert float((d > 0).float().mean()) < 1e-3 and float(d.max()) <= 2 ** -7 * float(b.float().abs().max()), (name, float((d > 0).float().mean()), float(d.max()))
    close("ln_bwd_dx partial sums", part1.sum(0), part0.sum(0), rel=1e-5)
    assert bool(torch.isfinite(out1.float()).all())
    close("ln_bwd_dx product", out1, out0, rel=2 ** -6)
    if epi == "mask":
        assert bool(((out1.float() == 0) | (cm.float() > 0)).all())
    if two:
        # the second contribution formed ON LOAD as the backward of another LayerNorm (no Dropout / residual) from its output gradient ga:
        # same results as handing over that backward's stored output, plus that LayerNorm's partial rows
        ga = rnd(M, 2 * E, dev=dev, seed=8).to(BF)[:, E:]                     # a column window (ld = 2E), as the decoder's hs gradient is
        xha = rnd(M, E, dev=dev, seed=9).to(BF)
        rstda = (0.5 + torch.rand(M, generator=torch.Generator().manual_seed(12))).to(dev)
        gama = 1 + 0.1 * rnd(E, dev=dev, seed=10)
        parta0 = torch.zeros(nb0, 2 * E, device=dev)
        dxa = torch.empty(M, E, device=dev, dtype=BF)
        lib.call("tuber_layernorm_bwd", ga, 2 * E, xha, rstda, gama, dxa, None, parta0, None, None, 2, M, E, 0.0, None, 0)
        outs = []
        for chained in (False, True):
            pa, pb = torch.zeros(nb1, 2 * E, device=dev), torch.zeros(nb1, 2 * E, device=dev)
            dxx, dxdx, oo = (torch.full((M, n), float("nan"), device=dev, dtype=BF) for n in (E, E, Kin))
            extra = (ga, 2 * E, xha, rstda, gama, pb) if chained else (None, 0, None, None, None, None)
            lib.call("tuber_ln_bwd_dx", dy, E, None if chained else dxa, 0 if chained else E, xh, rstd, gam, dxx, dxdx if p > 0 else None, pa, M, E, p, seed, salt,
                     wt, E, Kin, oo, res, cm, alpha, *extra)
            outs.append((dxx, dxdx, oo, pa, pb))
        torch.cuda.synchronize()
        for i in range(3 if p > 0 else 1):
            a, b = outs[1][i].float(), outs[0][i].float()
            if i == 1 and p == 0:
                continue
            d = (a - b).abs()
            # (the other LayerNorm's backward is recomputed in another compilation unit: a bf16 ulp on a few of its elements moves the sum's rounding)
            assert bool(torch.isfinite(a).all()) and float((d > 0).float().mean()) < 1e-2 and float(d.max()) <= 2 ** -6 * float(b.abs().max()), i
        close("chained: product", outs[1][2], outs[0][2], rel=2 ** -6)
        close("chained: partial rows", outs[1][3].sum(0), outs[0][3].sum(0), rel=1e-4)
        close("chained: the other LayerNorm's partial rows", outs[1][4].sum(0), parta0.sum(0), rel=1e-4)


@pytest.mark.parametrize("M,N,Nq,Kin,with_res", [(30, 768, 512, 256, True), (30, 256, 256, 256, True), (30, 768, 512, 256, False), (7, 512, 256, 128, True),
                                                 (77, 768, 512, 256, True), (30, 2048, 2048, 256, True), (30, 2048, 1024, 256, False), (30, 1280, 512, 64, True)])
def test_in_projection_data_gradients_in_one_launch(dev, M, N, Nq, Kin, with_res):
    """tuber_rows_dx2 == the two tuber_gemm_nt launches of tape.py: in_proj.bwd: dx = g.W (+ res) over all N columns, dpos = g[:, :Nq].W[:Nq]"""
    g = rnd(M, N, dev=dev, seed=1).to(BF)
    wt = (rnd(Kin, N, dev=dev, seed=2) / 16).to(BF)            # W^T rows [Kin][N]
    res = rnd(M, Kin, dev=dev, seed=3).to(BF) if with_res else None
    dx0, da0 = torch.empty(M, Kin, device=dev, dtype=BF), torch.empty(M, Kin, device=dev, dtype=BF)
    for ncols, r, out in ((N, res, dx0), (Nq, None, da0)):
        lib.call("tuber_gemm_nt", g, N, wt, N, out, Kin, M, Kin, ncols, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                 0, None, r, Kin, 0, 0, None, None, None, 0, None, None, 1.0, 0.0, None, 0, None, 0, None)
    dx1 = torch.full((M, Kin), float("nan"), device=dev, dtype=BF)
    da1 = torch.full((M, Kin), float("nan"), device=dev, dtype=BF)
    lib.call("tuber_rows_dx2", g, N, M, N, Nq, wt, N, Kin, dx1, res, da1)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(dx1.float()).all()) and bool(torch.isfinite(da1.float()).all())
    close("rows_dx2 dx", dx1, dx0, rel=2 ** -6)
    close("rows_dx2 dpos", da1, da0, rel=2 ** -6)
    ref = g.float() @ wt.float().t() + (res.float() if with_res else 0)
    close("rows_dx2 dx vs fp32", dx1, ref, rel=2 ** -6)
    close("rows_dx2 dpos vs fp32", da1, g.float()[:, :Nq] @ wt.float()[:, :Nq].t(), rel=2 ** -6)
    dx2 = torch.empty(M, Kin, device=dev, dtype=BF)
    lib.call("tuber_rows_dx2", g, N, M, N, Nq, wt, N, Kin, dx2, res, None)                # the prefix result not wanted
    assert torch.equal(dx2, dx1)


def test_gemm_epilogue_dropout_and_masked_dgrad(dev):
    """FFN pieces: h = Dropout(ReLU(x W1^T + b)) as one GEMM; the backward mask alpha*g*[h>0] as the epilogue of the next dgrad."""
    M, K, N, p, salt = 300, 256, 512, 0.1, 11
    seed = torch.tensor([99], dtype=torch.int64, device=dev)
    x = rnd(M, K, dev=dev, seed=1).to(BF)
    w = (rnd(N, K, dev=dev, seed=2) / 16).to(BF)
    bias = 0.1 * rnd(N, dev=dev, seed=3)
    h = torch.empty(M, N, device=dev, dtype=BF)
    lib.call("tuber_gemm_nt", x, K, w, K, h, N, M, N, K, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0,
             0, bias, None, 0, 1, 0, None, None, None, 0, None, None, 1.0, p, seed, salt, None, 0, None)
    ones = torch.ones(M, N, device=dev, dtype=BF)
    keep = torch.empty_like(ones)
    lib.call("tuber_dropout", ones, keep, M * N, p, seed, salt)
    ref = torch.relu(x.float() @ w.float().t() + bias) * keep.float()
    close("gemm relu+dropout epilogue", h, ref)
    # masked data gradient: dpre = alpha * (g W2) * [h > 0], W2 [K2, N]
    K2 = 256
    g = rnd(M, K2, dev=dev, seed=4).to(BF)
    w2t = (rnd(N, K2, dev=dev, seed=5) / 16).to(BF)            # W2^T rows = N (output features of the dgrad GEMM)
    dpre = torch.empty(M, N, device=dev, dtype=BF)
    alpha = 1.0 / (1.0 - p)
    lib.call("tuber_gemm_nt", g, K2, w2t, K2, dpre, N, M, N, K2, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0,
             2, None, None, 0, 0, 0, None, None, h, N, None, None, alpha, 0.0, None, 0, None, 0, None)
    refd = alpha * (g.float() @ w2t.float().t()) * (h.float() > 0)
    close("gemm masked dgrad epilogue", dpre, refd)
    gm = torch.empty_like(dpre)
    lib.call("tuber_relu_mask", g.new_ones(M, N), h, gm, M * N, alpha)
    close("relu_mask alpha", gm, alpha * (h.float() > 0))


@pytest.mark.parametrize("M,N,K", [(704, 2048, 256), (30, 256, 256), (180, 256, 2048), (704, 256, 2048)])
def test_gemm_tn_fused_bias_gradient(dev, M, N, K):
    """single-slab weight gradients also produce the bias gradient (column sums of G from the LDS image) in the same launch"""
    assert lib.query("tuber_gemm_tn_fuses_bias", M, N, K, N, K) == 1
    G = rnd(M, N, dev=dev, seed=1).to(BF)
    A = rnd(M, K, dev=dev, seed=2).to(BF)
    out = torch.ones(N, K, device=dev)
    db = torch.ones(N, device=dev)
    part = torch.empty(N * K, device=dev)
    lib.call("tuber_gemm_tn", G, N, A, K, part, out, 1, M, N, K, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0, None, 0, None, None, None, db)
    close("gemm_tn (+bias) dW", out - 1, G.float().t() @ A.float(), rel=4e-3)
    close("gemm_tn fused dbias", db - 1, G.float().sum(0), abs_=2e-3 * float(G.float().abs().sum(0).max()))


@pytest.mark.parametrize("M,N,K", [(5632, 1024, 256), (16896, 256, 256), (11264, 512, 256)])
def test_gemm_tn_fused_bias_gradient_multi_slab(dev, M, N, K):
    """with several slabs the bias gradient leaves as one partial row per slab; reduced immediately (tuber_reduce_rows) or by the
    deferred tuber_multi_reduce, the weight-gradient slabs likewise"""
    assert lib.query("tuber_gemm_tn_fuses_bias", M, N, K, N, K) == 2
    S = lib.query("tuber_gemm_tn_slabs", M, N, K)
    assert S > 1
    G = rnd(M, N, dev=dev, seed=1).to(BF)
    A = rnd(M, K, dev=dev, seed=2).to(BF)
    ref_w, ref_b = G.float().t() @ A.float(), G.float().sum(0)
    tol_b = 2e-3 * float(G.float().abs().sum(0).max())
    # immediate
    out, db = torch.ones(N, K, device=dev), torch.ones(N, device=dev)
    part = torch.full((S * N * K,), float("nan"), device=dev)
    bpart = torch.full((S * N,), float("nan"), device=dev)
    lib.call("tuber_gemm_tn", G, N, A, K, part, out, 1, M, N, K, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0, None, 0, None, None, None, bpart)
    lib.call("tuber_reduce_rows", bpart, db, S, N, 1)
    close("gemm_tn multi-slab dW", out - 1, ref_w, rel=4e-3)
    close("gemm_tn multi-slab fused dbias", db - 1, ref_b, abs_=tol_b)
    # deferred: accumulate = 2 leaves the slabs, one tuber_multi_reduce launch finishes both gradients, bit-identically
    from tubelet_transformer_amd.engine import DeferredReduce
    d = DeferredReduce(dev)
    out2, db2 = torch.ones(N, K, device=dev), torch.ones(N, device=dev)
    p2, b2 = d.alloc(S * N * K), d.alloc(S * N)
    lib.call("tuber_gemm_tn", G, N, A, K, p2, out2, 2, M, N, K, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0, None, 0, None, None, None, b2)
    d.add(p2, out2.data_ptr(), N * K, N * K, S, 0 if S <= 16 else 1)
    d.add(b2, db2.data_ptr(), N, N, S, 1)
    d.flush()
    assert torch.equal(out2, out) and torch.equal(db2, db)


@pytest.mark.parametrize("M,C,ld", [(704, 256, 256), (30, 2048, 2048), (180, 3, 64), (24, 3840, 3840), (16896, 512, 512), (5000, 80, 128)])
def test_colsum(dev, M, C, ld):
    g = rnd(M, ld, dev=dev, seed=1).to(BF)
    out = torch.ones(C, device=dev)
    part = torch.empty(lib.query("tuber_colsum_blocks", M) * C, device=dev)
    lib.call("tuber_colsum", g, part, out, 1, M, C, ld)
    ref = g.float()[:, :C].sum(0)
    close("colsum M%d C%d" % (M, C), out - 1, ref, abs_=2e-3 * float(g.float().abs().sum(0).max()))


def attn_ref(q, k, v, kpm, scale):
    """q [B,H,Lq,32] etc fp32; kpm [B,Lk] bool."""
    s = (q @ k.transpose(-1, -2)) * scale
    if kpm is not None:
        s = s.masked_fill(kpm[:, None, None, :], float("-inf"))
    return torch.softmax(s, -1) @ v


@pytest.mark.parametrize("B,Lq,Lk,masked", [(2, 352, 352, True), (2, 15, 352, True), (12, 15, 1408, False), (3, 200, 77, False),
                                            (40, 4, 4, False), (40, 100, 130, True), (2, 100, 700, True), (33, 40, 1000, False)])
def test_attention(dev, B, Lq, Lk, masked):
    """MFMA kernels in both work splits (64-row workgroups when ceil(Lq/64)*H*B >= 256: (40,100,130); 16-row workgroups whose four
    waves split the tiles of the other operand otherwise, incl. fewer tiles than waves, ragged last tiles and Lq != Lk -- the forward
    / dQ split follows Lq, the dK/dV split Lk: (33,40,1000) mixes them), and the scalar kernels for (40,4,4)."""
    H, E = 8, 256
    # token-major packed layout (l, b, E): row(l,b) = l*B + b
    q = rnd(Lq, B, E, dev=dev, seed=1).to(BF)
    k = rnd(Lk, B, E, dev=dev, seed=2).to(BF)
    v = rnd(Lk, B, E, dev=dev, seed=3).to(BF)
    kpm = None
    if masked:
        kpm = torch.zeros(B, Lk, dtype=torch.bool, device=dev)
        kpm[0, Lk - 37:] = True
        kpm[1, ::5] = True
    mq = torch.tensor([E, B, 1, 0, 1], dtype=torch.int64)
    o = torch.empty(Lq, B, E, device=dev, dtype=BF)
    lse = torch.empty(B, H, Lq, device=dev)
    scale = 32 ** -0.5
    mqp = mq.numpy().ctypes.data
    lib.call("tuber_attn_fwd", q, mqp, k, mqp, v, mqp, o, mqp, lse, kpm.to(torch.uint8) if masked else None, B, H, Lq, Lk, scale, 0.0, None, 0)

    def heads(x, L):
        return x.float().view(L, B, H, 32).permute(1, 2, 0, 3).requires_grad_(True)
    qh, kh, vh = heads(q, Lq), heads(k, Lk), heads(v, Lk)
    ref = attn_ref(qh, kh, vh, kpm, scale)
    refl = ref.permute(2, 0, 1, 3).reshape(Lq, B, E)
    close("attention fwd B%d Lq%d Lk%d" % (B, Lq, Lk), o, refl.detach())
    do = rnd(Lq, B, E, dev=dev, seed=4).to(BF)
    ref.backward(do.float().view(Lq, B, H, 32).permute(1, 2, 0, 3))
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B, H, Lq, device=dev)
    kp = kpm.to(torch.uint8) if masked else None
    lib.call("tuber_attn_bwd", q, mqp, k, mqp, v, mqp, o, mqp, lse, kp, do, mqp, dq, mqp, dk, mqp, dv, mqp, delta, B, H, Lq, Lk,
             scale, 0.0, None, 0)
    close("attention bwd dq", dq, qh.grad.permute(2, 0, 1, 3).reshape(Lq, B, E), rel=2 ** -6)
    close("attention bwd dk", dk, kh.grad.permute(2, 0, 1, 3).reshape(Lk, B, E), rel=2 ** -6)
    close("attention bwd dv", dv, vh.grad.permute(2, 0, 1, 3).reshape(Lk, B, E), rel=2 ** -6)


def test_attention_strided_maps_and_dropout(dev):
    """class-branch layout: rows (lb, t, hw); sequence over t with batch (lb, hw); packed qkv [rows, 768]."""
    LB, T, HW, H, E = 3, 4, 10, 8, 256
    rows = LB * T * HW
    qkv = rnd(rows, 3 * E, dev=dev, seed=1).to(BF)
    o = torch.zeros(rows, E, device=dev, dtype=BF)
    B = LB * HW
    lse = torch.empty(B, H, T, device=dev)
    m_in = torch.tensor([3 * E, HW, T * HW, 1, HW], dtype=torch.int64)      # ld, sL, s1, s2, B2
    m_out = torch.tensor([E, HW, T * HW, 1, HW], dtype=torch.int64)
    pi, po = m_in.numpy().ctypes.data, m_out.numpy().ctypes.data
    scale = 32 ** -0.5
    lib.call("tuber_attn_fwd", qkv, pi, qkv[:, E:], pi, qkv[:, 2 * E:], pi, o, po, lse, None, B, H, T, T, scale, 0.0, None, 0)
    x = qkv.float().view(LB, T, HW, 3, H, 32)
    qh = x[:, :, :, 0].permute(0, 2, 3, 1, 4).reshape(B, H, T, 32)
    kh = x[:, :, :, 1].permute(0, 2, 3, 1, 4).reshape(B, H, T, 32)
    vh = x[:, :, :, 2].permute(0, 2, 3, 1, 4).reshape(B, H, T, 32)
    ref = attn_ref(qh, kh, vh, None, scale).view(LB, HW, H, T, 32).permute(0, 3, 1, 2, 4).reshape(rows, E)
    close("attention strided maps", o, ref)
    # dropout: statistical check -- mean preserved, and fwd is deterministic for a fixed seed
    o1, o2 = torch.zeros_like(o), torch.zeros_like(o)
    seed_t = torch.full((1,), 7, dtype=torch.int64, device=dev)
    lib.call("tuber_attn_fwd", qkv, pi, qkv[:, E:], pi, qkv[:, 2 * E:], pi, o1, po, lse, None, B, H, T, T, scale, 0.1, seed_t, 1234)
    lib.call("tuber_attn_fwd", qkv, pi, qkv[:, E:], pi, qkv[:, 2 * E:], pi, o2, po, lse, None, B, H, T, T, scale, 0.1, seed_t, 1234)
    assert torch.equal(o1, o2)
    assert not torch.equal(o1, o)
    rel = float((o1.float() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    print("dropout(0.1) relative rms deviation from the no-dropout output: %.3f" % rel)
    assert 0.02 < rel < 0.6


@pytest.mark.parametrize("Lq,Lk", [(64, 32), (100, 32), (15, 32)])
def test_attention_dropout_backward(dev, Lq, Lk):
    """Dropout on the attention weights, forward AND backward, against autograd: with Lk = 32 and V = I the forward output IS the
    dropped probability matrix, which exposes the kernel's keep mask; the mask depends on (seed, salt, b, h, q, k) only, so the
    same mask applies to a second call with random V.  Lq >= 32 takes the MFMA kernels, Lq = 15 the scalar ones."""
    B, H, E, p = 2, 8, 256, 0.25
    scale = 32 ** -0.5
    seed_t = torch.full((1,), 11, dtype=torch.int64, device=dev)
    q = rnd(Lq, B, E, dev=dev, seed=1).to(BF)
    k = rnd(Lk, B, E, dev=dev, seed=2).to(BF)
    v = rnd(Lk, B, E, dev=dev, seed=3).to(BF)
    eye = torch.eye(32, device=dev).repeat(1, H).view(Lk, 1, E).repeat(1, B, 1).to(BF).contiguous()
    mp = torch.tensor([E, B, 1, 0, 1], dtype=torch.int64).numpy()
    mpp = mp.ctypes.data
    lse = torch.empty(B, H, Lq, device=dev)
    pd = torch.empty(Lq, B, E, device=dev, dtype=BF)
    lib.call("tuber_attn_fwd", q, mpp, k, mpp, eye, mpp, pd, mpp, lse, None, B, H, Lq, Lk, scale, p, seed_t, 77)
    keep = (pd.float().view(Lq, B, H, 32).permute(1, 2, 0, 3) > 0).float() / (1 - p)          # [B,H,Lq,Lk]
    assert 0.6 < float((keep > 0).float().mean()) < 0.9

    def heads(x, L):
        return x.float().view(L, B, H, 32).permute(1, 2, 0, 3).requires_grad_(True)
    qh, kh, vh = heads(q, Lq), heads(k, Lk), heads(v, Lk)
    ref = (torch.softmax(qh @ kh.transpose(-1, -2) * scale, -1) * keep) @ vh
    o = torch.empty(Lq, B, E, device=dev, dtype=BF)
    lib.call("tuber_attn_fwd", q, mpp, k, mpp, v, mpp, o, mpp, lse, None, B, H, Lq, Lk, scale, p, seed_t, 77)
    close("attention+dropout fwd", o, ref.permute(2, 0, 1, 3).reshape(Lq, B, E).detach())
    do = rnd(Lq, B, E, dev=dev, seed=4).to(BF)
    ref.backward(do.float().view(Lq, B, H, 32).permute(1, 2, 0, 3))
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    delta = torch.empty(B, H, Lq, device=dev)
    lib.call("tuber_attn_bwd", q, mpp, k, mpp, v, mpp, o, mpp, lse, None, do, mpp, dq, mpp, dk, mpp, dv, mpp, delta, B, H, Lq, Lk, scale,
             p, seed_t, 77)
    back = lambda t, L: t.grad.permute(2, 0, 1, 3).reshape(L, B, E)
    close("attention+dropout bwd dq", dq, back(qh, Lq), rel=2 ** -5)
    close("attention+dropout bwd dk", dk, back(kh, Lk), rel=2 ** -5)
    close("attention+dropout bwd dv", dv, back(vh, Lk), rel=2 ** -5)


@pytest.mark.parametrize("N,T,H,W", [(1, 4, 30, 38), (2, 3, 64, 340), (2, 8, 96, 150)])
def test_stem(dev, N, T, H, W):
    clip = rnd(N, 3, T, H, W, dev=dev, seed=1)
    w = rnd(64, 3, 3, 7, 7, dev=dev, seed=2, scale=441 ** -0.5)
    Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
    M = N * T * Ho * Wo
    ref = F.conv3d(bfr(clip), bfr(w), stride=(1, 2, 2), padding=(1, 3, 3)).permute(0, 2, 3, 4, 1).reshape(M, 64)
    # implicit-GEMM stem conv (the shipped path): forward + BN partial stats + weight gradient
    wp = torch.zeros(64, 512, device=dev, dtype=BF)
    lib.call("tuber_stem_pack_weight", w.contiguous(), wp)
    R = lib.query("tuber_stem_conv_blocks", N, T, H, W)
    c2 = torch.empty(M, 64, device=dev, dtype=BF)
    a0, a1 = torch.zeros(R, 64, device=dev), torch.zeros(R, 64, device=dev)
    lib.call("tuber_stem_conv_fwd", clip, wp, c2, a0, a1, N, T, H, W)
    close("stem conv implicit gemm", c2, ref)
    close("stem conv implicit stats sum", a0.sum(0), ref.sum(0), abs_=2e-3 * float(ref.abs().sum(0).max()))
    close("stem conv implicit stats sumsq", a1.sum(0), (ref * ref).sum(0), rel=2e-3)
    gg = rnd(M, 64, dev=dev, seed=8).to(BF)
    wr = bfr(w).clone().requires_grad_(True)
    F.conv3d(bfr(clip), wr, stride=(1, 2, 2), padding=(1, 3, 3)).backward(gg.float().view(N, T, Ho, Wo, 64).permute(0, 4, 1, 2, 3))
    dwt = torch.zeros(64, 441, device=dev)
    part = torch.empty(lib.query("tuber_stem_conv_wgrad_blocks", N, T, H, W) * 512 * 64, device=dev)
    lib.call("tuber_stem_conv_bwd_weight", clip, gg, part, dwt, 0, N, T, H, W)
    close("stem conv implicit dW", dwt, wr.grad.view(64, 441), rel=2e-3)
    # the same kernel with the BatchNorm backward apply folded into its gradient operand: bit-identical to apply + weight gradient
    cA, cB, cC = 1 + 0.2 * rnd(64, dev=dev, seed=11), 0.1 * rnd(64, dev=dev, seed=12), 0.05 * rnd(64, dev=dev, seed=13)
    dc = torch.empty(M, 64, device=dev, dtype=BF)
    lib.call("tuber_bn_bwd_apply", gg, c2, cA, cB, cC, dc, M, 64)
    dw_a, dw_b = torch.zeros(64, 441, device=dev), torch.zeros(64, 441, device=dev)
    lib.call("tuber_stem_conv_bwd_weight", clip, dc, part, dw_a, 0, N, T, H, W)
    lib.call("tuber_stem_conv_bwd_weight_bn", clip, gg, c2, cA, cB, cC, part, dw_b, 0, N, T, H, W)
    torch.cuda.synchronize()
    assert torch.equal(dw_a, dw_b)
    assert float(dw_a.abs().max()) > 0
    # pool fwd
    sc, sh = 1 + 0.1 * rnd(64, dev=dev, seed=3), 0.1 * rnd(64, dev=dev, seed=4)
    Hp, Wp = (Ho + 2 - 3) // 2 + 1, (Wo + 2 - 3) // 2 + 1
    out = torch.empty(N * T * Hp * Wp, 64, device=dev, dtype=BF)
    arg = torch.empty(N * T * Hp * Wp, 64, device=dev, dtype=torch.uint8)
    lib.call("tuber_stem_pool_fwd", c2, sc, sh, out, arg, N * T, Ho, Wo, Hp, Wp)
    a = (c2.float() * sc + sh).relu().view(N * T, Ho, Wo, 64).permute(0, 3, 1, 2).requires_grad_(True)
    pr = F.max_pool2d(a, 3, 2, 1)
    close("stem pool fwd", out, pr.detach().permute(0, 2, 3, 1).reshape(-1, 64))
    g = rnd(N * T * Hp * Wp, 64, dev=dev, seed=5).to(BF)
    pr.backward(g.float().view(N * T, Hp, Wp, 64).permute(0, 3, 1, 2))
    da = a.grad.permute(0, 2, 3, 1).reshape(M, 64)
    dz_ref = da * ((c2.float() * sc + sh) > 0)
    dz = torch.empty(M, 64, device=dev, dtype=BF)
    R = lib.query("tuber_stem_pool_bwd_stat_rows", M)
    s0, s1 = torch.zeros(R, 64, device=dev), torch.zeros(R, 64, device=dev)
    lib.call("tuber_stem_pool_bwd", g, arg, c2, sc, sh, dz, s0, s1, N * T, Ho, Wo, Hp, Wp)
    close("stem pool bwd dz", dz, dz_ref)
    close("stem pool bwd sum dz", s0.sum(0), dz_ref.sum(0), abs_=2e-3 * float(dz_ref.abs().sum(0).max()))
    close("stem pool bwd sum dz*x", s1.sum(0), (dz_ref * c2.float()).sum(0), abs_=2e-3 * float((dz_ref * c2.float()).abs().sum(0).max()))


def test_elementwise(dev):
    n = 8 * 1000
    a, b = rnd(n, dev=dev, seed=1).to(BF), rnd(n, dev=dev, seed=2).to(BF)
    out = torch.empty_like(a)
    lib.call("tuber_axpby", a, b, out, n, 1.0, 1.0)
    close("axpby", out, a.float() + b.float())
    W = rnd(300, 70, dev=dev, seed=3)
    WT = torch.zeros(70, 304, device=dev, dtype=BF)
    lib.call("tuber_cast_transpose", W, WT, 300, 70, 304)
    close("cast_transpose", WT[:, :300], W.t())
    wb = torch.empty(300 * 70, device=dev, dtype=BF)
    lib.call("tuber_cast_f32_bf16", W, wb, 300 * 70)
    assert torch.equal(wb, W.to(BF).view(-1))
    # temporal average pool as gather-sum: x [N, T, S, C] -> [N, S, C]
    N, T, S, C = 2, 4, 30, 64
    x = rnd(N, T, S, C, dev=dev, seed=4).to(BF)
    o = torch.empty(N, S, C, device=dev, dtype=BF)
    lib.call("tuber_rows_gather_sum", x, o, N, 1, S, T, T * S, 0, 1, S, C, 0.25)
    close("avgpool_t via gather-sum", o, x.float().mean(1))
    rep = torch.empty(6, N * S, C, device=dev, dtype=BF)
    lib.call("tuber_rows_gather_sum", o, rep, 6, 1, N * S, 1, 0, 0, 1, 0, C, 1.0)
    assert torch.equal(rep, o.view(1, N * S, C).expand(6, -1, -1))
    y = torch.empty(n, device=dev)
    xx = rnd(n, dev=dev, seed=5)
    lib.call("tuber_sigmoid_fwd", xx, y, n)
    close("sigmoid", y, torch.sigmoid(xx), rel=1e-5)
    d = torch.empty_like(a)
    lib.call("tuber_dropout", a, d, n, 0.5, None, 77)
    kept = (d.float() != 0).float().mean().item()
    print("dropout keep fraction %.3f" % kept)
    assert 0.45 < kept < 0.55
    close("dropout scaling", d.float()[d.float() != 0], 2 * a.float()[d.float() != 0])
    mask = torch.zeros(2, 1, 5, 7, dtype=torch.bool, device=dev)
    mask[1, :, 4:, :] = True
    mask[1, :, :, 5:] = True
    pos = torch.empty(2 * 5 * 7, 256, device=dev, dtype=BF)
    lib.call("tuber_posenc", mask.to(torch.uint8), pos, 2, 1, 5, 7, 256)
    from oracle import tuber_oracle as O
    ref = O.position_embedding_sine_3d(mask.cpu(), 256).permute(0, 2, 3, 4, 1).reshape(-1, 256).to(dev)
    close("posenc", pos, ref, abs_=1e-2)


@pytest.mark.parametrize("B,T,hw,E", [(2, 4, 352, 2048), (3, 5, 7, 64)])
def test_temporal_max_pool(dev, B, T, hw, E):
    """TEMPORAL_DS_STRATEGY 'max' (backbone_builder.py:45-47,73): forward = nn.MaxPool3d((T,1,1)) exactly (bf16 values, ties are frequent
    and must go to the earliest frame like torch), backward routes the gradient to that frame"""
    x = (rnd(B, T, hw, E, dev=dev, seed=1) * 2).to(BF)
    x[0, :, 0, :8] = 1.0                                        # a full tie
    out = torch.empty(B * hw, E, device=dev, dtype=BF)
    arg = torch.empty(B * hw, E, device=dev, dtype=torch.uint8)
    lib.call("tuber_temporal_max_fwd", x, out, arg, B, T, hw, E)
    xr = x.float().permute(0, 3, 1, 2).reshape(B, E, T, hw, 1).requires_grad_(True)
    ref, idx = F.max_pool3d(xr, (T, 1, 1), return_indices=True)
    assert torch.equal(out.view(B, hw, E).float(), ref.reshape(B, E, hw).permute(0, 2, 1))
    assert torch.equal(arg.view(B, hw, E).long(), (idx.reshape(B, E, hw) // hw).permute(0, 2, 1))
    g = rnd(B * hw, E, dev=dev, seed=2).to(BF)
    dx = torch.full((B, T, hw, E), float("nan"), device=dev, dtype=BF)
    lib.call("tuber_temporal_max_bwd", g, arg, dx, B, T, hw, E)
    ref.backward(g.float().view(B, hw, E).permute(0, 2, 1).reshape(B, E, 1, hw, 1))
    assert torch.equal(dx.float(), xr.grad.reshape(B, E, T, hw).permute(0, 2, 3, 1))
    out2 = torch.empty_like(out)
    lib.call("tuber_temporal_max_fwd", x, out2, None, B, T, hw, E)     # eval: no argmax
    assert torch.equal(out2, out)


def test_gemm_tn_group_is_bit_identical_to_single_launches(dev):
    """tuber_gemm_tn_group: several weight-gradient GEMMs (plain and BN+ReLU operand, one with the strided row gather of the
    down_sample conv, single- and multi-slab, direct accumulation and slab partials) in one launch == the individual launches"""
    import ctypes
    from tubelet_transformer_amd.engine import TnArgs
    assert lib.query("tuber_gemm_tn_args_bytes") == ctypes.sizeof(TnArgs)
    n_, Ti, Hi, Wi, st, ss = 2, 4, 8, 10, 2, 2
    To, Ho, Wo = Ti // st, Hi // ss, Wi // ss
    probs = [(5632, 1024, 256, 1, None), (5632, 256, 1024, 0, None), (704, 512, 2048, 0, None), (n_ * To * Ho * Wo, 512, 256, 0, (To, Ho, Wo, Ti, Hi, Wi, st, ss)),
             (44032, 128, 512, 0, None), (704, 2048, 256, 0, "bias"), (16896, 256, 256, 0, "bias")]
    entries, keep, singles = [], [], []
    for i, (M, N, K, amode, gather) in enumerate(probs):
        bias = gather == "bias"                      # fused bias gradient (nn.Linear weight gradients of the transformer)
        gather = None if bias else gather
        rows_a = n_ * Ti * Hi * Wi if gather else M
        G = rnd(M, N, dev=dev, seed=10 + i).to(BF)
        A = rnd(rows_a, K, dev=dev, seed=20 + i).to(BF)
        sc, sh = 1.0 + 0.2 * rnd(K, dev=dev, seed=30 + i), 0.3 * rnd(K, dev=dev, seed=40 + i)
        S = lib.query("tuber_gemm_tn_slabs", M, N, K)
        g = gather or (0,) * 8
        res = []
        for grouped in (False, True):
            out = torch.full((N, K), 0.25, device=dev)
            part = torch.full((max(S, 1) * N * K,), float("nan"), device=dev) if S > 1 else None
            acc = 2 if S > 1 else 1
            bg = torch.full((max(S, 1) * N,), 0.5, device=dev) if bias else None
            if grouped:
                entries.append(TnArgs(G.data_ptr(), N, A.data_ptr(), K, part.data_ptr() if part is not None else None, out.data_ptr(), acc, M, N, K,
                                      amode, 1 if gather else 0, *g, sc.data_ptr() if amode else None, sh.data_ptr() if amode else None,
                                      bg.data_ptr() if bias else None))
            else:
                lib.call("tuber_gemm_tn", G, N, A, K, part, out, acc, M, N, K, amode, sc if amode else None, sh if amode else None,
                         1 if gather else 0, *g, None, 0, None, None, None, bg)
            res.append((out, part, bg))
        keep.append((G, A, sc, sh))
        singles.append(res)
    arr = (TnArgs * len(entries))(*entries)
    lib.call("tuber_gemm_tn_group", arr, len(entries))
    torch.cuda.synchronize()
    for (M, N, K, amode, gather), ((o1, p1, b1), (o2, p2, b2)) in zip(probs, singles):
        assert torch.equal(o1, o2), (M, N, K)
        if b1 is not None:
            assert torch.equal(b1, b2) and float((b1 - 0.5).abs().max()) > 0, (M, N, K)
        if p1 is not None:
            assert torch.equal(p1, p2), (M, N, K)
            assert bool(torch.isfinite(p1).all())
        else:
            assert float((o1 - 0.25).abs().max()) > 0


@pytest.mark.parametrize("M,N,K,res", [(5632, 1024, 256, True), (44032, 512, 128, True), (2816, 2048, 512, False), (1000, 256, 64, True)])
def test_gemm_nt_join_equals_dgrad_then_block_out_bwd(dev, M, N, K, res):
    """tuber_gemm_nt_join == tuber_gemm_nt(+R) followed by tuber_block_out_bwd: dz bit-identical, partial statistics equal after
    summing their rows (the two paths cut the rows into different blocks)"""
    A = rnd(M, K, dev=dev, seed=1).to(BF)
    B = rnd(N, K, dev=dev, seed=2, scale=K ** -0.5).to(BF)
    Rr = rnd(M, N, dev=dev, seed=3).to(BF) if res else None
    Y = rnd(M, N, dev=dev, seed=4).to(BF).relu()
    C4 = rnd(M, N, dev=dev, seed=5).to(BF)
    dx = torch.empty(M, N, device=dev, dtype=BF)
    lib.call("tuber_gemm_nt", A, K, B, K, dx, N, M, N, K, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0,
             0, None, Rr, N, 0, 0, None, None, None, 0, None, None, 1.0, 0.0, None, 0, None, 0, None)
    R0 = lib.query("tuber_rowblock_count", M, N)
    a0, a1 = torch.zeros(R0, N, device=dev), torch.zeros(R0, N, device=dev)
    dz0 = torch.empty(M, N, device=dev, dtype=BF)
    lib.call("tuber_block_out_bwd", dx, Y, C4, None, dz0, a0, a1, None, M, N)
    R1 = lib.query("tuber_gemm_nt_stat_rows", M, N)
    b0, b1 = torch.full((R1, N), float("nan"), device=dev), torch.full((R1, N), float("nan"), device=dev)
    dz1 = torch.full((M, N), float("nan"), device=dev, dtype=BF)
    lib.call("tuber_gemm_nt_join", A, K, B, K, dz1, N, M, N, K, Rr, N, Y, N, C4, N, b0, b1)
    torch.cuda.synchronize()
    assert torch.equal(dz0, dz1)
    ref = (A.float() @ B.float().t() + (Rr.float() if res else 0)) * (Y.float() > 0)
    close("join dz vs torch", dz1, ref)
    close("join sum dz", b0.sum(0), a0.sum(0), abs_=1e-4 * float(dz0.float().abs().sum(0).max()))
    close("join sum dz*c4", b1.sum(0), a1.sum(0), abs_=1e-4 * float((dz0.float() * C4.float()).abs().sum(0).max()))
    # below a stage's first block: a third statistics row for the projection shortcut's BatchNorm (tuber_gemm_nt_join_ds)
    Cd = rnd(M, N, dev=dev, seed=6).to(BF)
    a2 = torch.zeros(R0, N, device=dev)
    lib.call("tuber_block_out_bwd", dx, Y, C4, Cd, dz0, a0, a1, a2, M, N)
    d0, d1, d2 = (torch.full((R1, N), float("nan"), device=dev) for _ in range(3))
    dz2 = torch.full((M, N), float("nan"), device=dev, dtype=BF)
    lib.call("tuber_gemm_nt_join_ds", A, K, B, K, dz2, N, M, N, K, Rr, N, Y, N, C4, N, Cd, N, d0, d1, d2)
    torch.cuda.synchronize()
    assert torch.equal(dz2, dz1)
    close("ds join sum dz", d0.sum(0), a0.sum(0), abs_=1e-4 * float(dz0.float().abs().sum(0).max()))
    close("ds join sum dz*c4", d1.sum(0), a1.sum(0), abs_=1e-4 * float((dz0.float() * C4.float()).abs().sum(0).max()))
    close("ds join sum dz*cd", d2.sum(0), a2.sum(0), abs_=1e-4 * float((dz0.float() * Cd.float()).abs().sum(0).max()))


@pytest.mark.parametrize("n,Ti,Hi,Wi,st,ss,N,K", [(2, 8, 16, 22, 2, 2, 512, 128), (2, 4, 9, 11, 2, 2, 256, 64), (1, 8, 16, 22, 2, 1, 1024, 512),
                                                  (2, 6, 10, 12, 1, 2, 256, 128)])
def test_gemm_nt_join_strided_residual(dev, n, Ti, Hi, Wi, st, ss, N, K):
    """tuber_gemm_nt_join_strided (a stage boundary: the join below a block whose projection shortcut is STRIDED) == tuber_gemm_nt followed by
    tuber_rows_scatter_add of the shortcut's gradient and tuber_block_out_bwd: dz bit-identical, statistics equal after summing their rows.
    Odd grids (rows that are not sampled at the high end), temporal-only and spatial-only strides (layer4 with LAST_STRIDE False)."""
    To, Ho, Wo = (Ti - 1) // st + 1, (Hi - 1) // ss + 1, (Wi - 1) // ss + 1
    M, Mo = n * Ti * Hi * Wi, n * To * Ho * Wo
    A = rnd(M, K, dev=dev, seed=1).to(BF)
    B = rnd(N, K, dev=dev, seed=2, scale=K ** -0.5).to(BF)
    Rs = rnd(Mo, N, dev=dev, seed=3).to(BF)                      # the projection shortcut's data gradient, one row per sampled position
    Y = rnd(M, N, dev=dev, seed=4).to(BF).relu()
    C4 = rnd(M, N, dev=dev, seed=5).to(BF)
    dx = torch.empty(M, N, device=dev, dtype=BF)
    lib.call("tuber_gemm_nt", A, K, B, K, dx, N, M, N, K, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0,
             0, None, None, N, 0, 0, None, None, None, 0, None, None, 1.0, 0.0, None, 0, None, 0, None)
    lib.call("tuber_rows_scatter_add", dx, Rs, Mo, To, Ho, Wo, Ti, Hi, Wi, st, ss, N)
    R0 = lib.query("tuber_rowblock_count", M, N)
    a0, a1 = torch.zeros(R0, N, device=dev), torch.zeros(R0, N, device=dev)
    dz0 = torch.empty(M, N, device=dev, dtype=BF)
    lib.call("tuber_block_out_bwd", dx, Y, C4, None, dz0, a0, a1, None, M, N)
    R1 = lib.query("tuber_gemm_nt_stat_rows", M, N)
    b0, b1 = torch.full((R1, N), float("nan"), device=dev), torch.full((R1, N), float("nan"), device=dev)
    dz1 = torch.full((M, N), float("nan"), device=dev, dtype=BF)
    lib.call("tuber_gemm_nt_join_strided", A, K, B, K, dz1, N, M, N, K, Rs, N, To, Ho, Wo, Ti, Hi, Wi, st, ss, Y, N, C4, N, b0, b1)
    torch.cuda.synchronize()
    # (the unfused path rounds dx to bf16 before the scatter-add adds the shortcut's gradient; the fused epilogue adds in fp32 and rounds once)
    full = torch.zeros(n, Ti, Hi, Wi, N, device=dev)
    full[:, ::st, ::ss, ::ss] = Rs.float().view(n, To, Ho, Wo, N)
    ref = (A.float() @ B.float().t() + full.view(M, N)) * (Y.float() > 0)
    close("strided join dz vs torch", dz1, ref)
    ndiff = int((dz0 != dz1).sum())
    frac = Mo / M           # only the sampled rows receive a residual, i.e. can round differently
    assert ndiff <= 0.4 * frac * dz1.numel(), "strided join: dz differs from the three-kernel path in %d of %d elements" % (ndiff, dz1.numel())
    close("strided join dz vs three kernels", dz1, dz0.float(), rel=2 ** -6)
    close("strided join sum dz", b0.sum(0), a0.sum(0), abs_=2e-3 * float(dz0.float().abs().sum(0).max()))
    close("strided join sum dz*c4", b1.sum(0), a1.sum(0), abs_=2e-3 * float((dz0.float() * C4.float()).abs().sum(0).max()))


@pytest.mark.parametrize("B,H,W,h,w", [(2, 256, 340, 16, 22), (2, 288, 384, 18, 24), (3, 64, 96, 4, 6), (1, 224, 224, 14, 14), (2, 250, 333, 16, 21)])
def test_mask_resize_matches_interpolate_nearest(dev, B, H, W, h, w):
    """tuber_mask_resize == F.interpolate(mask[None].float(), size=(h, w)).to(bool)[0] (backbone_builder.py:85-86) for ragged padding masks"""
    g = torch.Generator().manual_seed(B * H + W)
    mask = torch.zeros(B, H, W, dtype=torch.bool)
    for b in range(B):
        hh, ww = int(torch.randint(H // 2, H + 1, (1,), generator=g)), int(torch.randint(W // 2, W + 1, (1,), generator=g))
        mask[b, hh:, :] = True
        mask[b, :, ww:] = True
    mask = mask.to(dev)
    out = torch.empty(B, h * w, dtype=torch.uint8, device=dev)
    lib.call("tuber_mask_resize", mask, out, B, H, W, h, w)
    ref = F.interpolate(mask[None].float(), size=(h, w)).to(torch.bool)[0].reshape(B, h * w)
    assert torch.equal(out.bool(), ref)


@pytest.mark.parametrize("ava", [True, False])
def test_class_error_kernel(dev, ava):
    g = torch.Generator().manual_seed(3)
    B, Q, C, Tmax = 3, 15, (80 if ava else 22), 8
    logits = torch.randn(B, Q, C, generator=g).to(dev)
    match = torch.full((B, Tmax), -1, dtype=torch.int32)
    counts = [3, 0, 5]
    for b, n in enumerate(counts):
        match[b, :n] = torch.randperm(Q, generator=g)[:n].int()
    match = match.to(dev)
    if ava:
        tl = (torch.rand(B, Tmax, C, generator=g) < 0.06).float()
        tl[:, :, 11] = 1.0
        # make some rows exactly right: labels = top-k of the matched query
        for b, t in ((0, 0), (2, 1), (2, 3)):
            k = int(tl[b, t].sum())
            top = logits[b, int(match[b, t])].topk(k).indices.cpu()
            tl[b, t] = 0
            tl[b, t, top] = 1.0
        tl = tl.to(dev)
        rows = torch.gather(logits, 1, match.clamp(min=0).long()[:, :, None].expand(-1, -1, C))
        lab = tl > 0.5
        lo = torch.where(lab, rows, rows.new_full((), float("inf"))).amin(-1)
        hi = torch.where(lab, rows.new_full((), float("-inf")), rows).amax(-1)
        ok = (lo > hi) & (match >= 0)
    else:
        tl = torch.randint(0, C - 1, (B, Tmax), generator=g).float()
        for b, t in ((0, 1), (2, 0), (2, 4)):
            tl[b, t] = float(logits[b, int(match[b, t])].argmax())
        tl = tl.to(dev)
        rows = torch.gather(logits, 1, match.clamp(min=0).long()[:, :, None].expand(-1, -1, C))
        ok = (rows.argmax(-1) == tl.long()) & (match >= 0)
    want = 100.0 - 100.0 * float(ok.sum()) / max(int((match >= 0).sum()), 1)
    out = torch.empty(1, device=dev)
    lib.call("tuber_class_error", logits, match, tl, B, Q, C, Tmax, 1 if ava else 0, out)
    assert abs(float(out) - want) < 1e-4 and int(ok.sum()) == 3, (float(out), want)


@pytest.mark.parametrize("M,C,R", [(5632, 1024, 88), (5632, 256, 88), (2816, 2048, 44), (1408, 2048, 22), (1000, 128, 7), (2816, 512, 128)])
def test_bn_bwd_one_launch_matches_finalize_plus_apply(dev, M, C, R):
    """tuber_bn_bwd_fa (finalize + apply in one launch, short partial lists) against tuber_bn_bwd_finalize + tuber_bn_bwd_apply:
    same fp64 arithmetic in another fixed summation order -> dx equal up to one bf16 ulp on rare elements, dgamma / dbeta to 1e-6 rel"""
    dz = rnd(M, C, dev=dev, seed=1).to(BF)
    x = rnd(M, C, dev=dev, seed=2).to(BF)
    gamma = 1 + 0.1 * rnd(C, dev=dev, seed=3)
    mean, invstd = 0.1 * rnd(C, dev=dev, seed=4), (1 + 0.2 * torch.rand(C, generator=torch.Generator().manual_seed(5))).to(dev)
    # partial rows: R row blocks of the true sums
    dch, xch = dz.float().chunk(R, 0), x.float().chunk(R, 0)
    b0 = torch.stack([c.sum(0) for c in dch]).contiguous()
    b1 = torch.stack([(c * d).sum(0) for c, d in zip(dch, xch)]).contiguous()
    R = b0.shape[0]
    cA, cB, cC = (torch.zeros(C, device=dev) for _ in range(3))
    dg0, db0 = torch.full((C,), 0.5, device=dev), torch.full((C,), 0.25, device=dev)
    lib.call("tuber_bn_bwd_finalize", b0, b1, R, C, float(M), gamma, mean, invstd, cA, cB, cC, dg0, db0, 1)
    dx0 = torch.empty(M, C, device=dev, dtype=BF)
    lib.call("tuber_bn_bwd_apply", dz, x, cA, cB, cC, dx0, M, C)
    try:
        for kr in (0, 4, 8, 11):         # rows per workgroup = 16 * kr: the launcher's own choice (0), then each form forced
            lib.query("tuber_bn_bwd_fa_rows_set", kr)
            dg1, db1 = torch.full((C,), 0.5, device=dev), torch.full((C,), 0.25, device=dev)
            dx1 = torch.full((M, C), float("nan"), device=dev, dtype=BF)
            lib.call("tuber_bn_bwd_fa", b0, b1, R, C, float(M), gamma, mean, invstd, dg1, db1, dz, x, dx1, M)
            torch.cuda.synchronize()
            assert bool(torch.isfinite(dx1.float()).all())
            diff = (dx0.float() - dx1.float()).abs()
            assert float((diff > 0).float().mean()) < 1e-3 and float(diff.max()) <= 2 ** -7 * float(dx0.float().abs().max())
            close("fa dgamma", dg1, dg0, rel=1e-5)
            close("fa dbeta", db1, db0, rel=1e-5)
            # frozen BatchNorm: no dgamma / dbeta
            dx2 = torch.empty(M, C, device=dev, dtype=BF)
            lib.call("tuber_bn_bwd_fa", b0, b1, R, C, float(M), gamma, mean, invstd, None, None, dz, x, dx2, M)
            assert torch.equal(dx2, dx1)
    finally:
        lib.query("tuber_bn_bwd_fa_rows_set", 0)
    assert lib.query("tuber_bn_bwd_fa_rows", M, C) in (64, 128, 176)


@pytest.mark.parametrize("M,N,K,add_cols", [(704, 768, 256, 512), (30, 768, 256, 512), (30, 256, 256, 256), (704, 512, 256, 256), (2816, 768, 256, 512)])
def test_gemm_nt_addproj_and_its_weight_gradient(dev, M, N, K, add_cols):
    """packed in-projection with the positional embedding folded in: columns [0, add_cols) see bf16(x + pos), the rest x -- bit-identical to
    add kernel + two GEMM launches; and the dW GEMM with the A + A2 operand formed on load (tuber_gemm_tn_group entry with A2)"""
    from tubelet_transformer_amd.engine import TnArgs
    x = rnd(M, K, dev=dev, seed=1).to(BF)
    pos = rnd(M, K, dev=dev, seed=2).to(BF)
    W = rnd(N, K, dev=dev, seed=3, scale=K ** -0.5).to(BF)
    bias = rnd(N, dev=dev, seed=4)
    y = torch.full((M, N), float("nan"), device=dev, dtype=BF)
    lib.call("tuber_gemm_nt_addproj", x, K, pos, K, add_cols, W, K, y, N, M, N, K, bias)
    xs = torch.empty_like(x)
    lib.call("tuber_axpby", x, pos, xs, x.numel(), 1.0, 1.0)
    y0 = torch.empty(M, N, device=dev, dtype=BF)
    for (c0, c1, a) in ((0, add_cols, xs), (add_cols, N, x)):
        if c1 > c0:
            out = torch.empty(M, c1 - c0, device=dev, dtype=BF)
            lib.call("tuber_gemm_nt", a, K, W[c0:c1].contiguous(), K, out, c1 - c0, M, c1 - c0, K, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                     0, bias[c0:c1].contiguous(), None, 0, 0, 0, None, None, None, 0, None, None, 1.0, 0.0, None, 0, None, 0, None)
            y0[:, c0:c1] = out
    torch.cuda.synchronize()
    assert torch.equal(y, y0)
    # weight gradient of the first row block: G = g[:, :add_cols] (ld N), A = x + pos formed on load
    g = rnd(M, N, dev=dev, seed=5).to(BF)
    S = lib.query("tuber_gemm_tn_slabs", M, add_cols, K)
    dW = torch.zeros(add_cols, K, device=dev)
    db = torch.zeros(max(S, 1) * add_cols, device=dev)
    part = torch.zeros(max(S, 1) * add_cols * K, device=dev)
    arr = (TnArgs * 1)(TnArgs(g.data_ptr(), N, x.data_ptr(), K, part.data_ptr() if S > 1 else None, dW.data_ptr(), 2 if S > 1 else 1, M, add_cols, K,
                              0, 0, 0, 0, 0, 0, 0, 0, 0, 0, None, None, db.data_ptr(), pos.data_ptr(), K))
    lib.call("tuber_gemm_tn_group", arr, 1)
    got = part.view(S, add_cols, K).sum(0) if S > 1 else dW
    gb = db.view(max(S, 1), add_cols).sum(0)
    close("addproj dW", got, g[:, :add_cols].float().t() @ xs.float(), rel=4e-3)
    close("addproj dbias", gb, g[:, :add_cols].float().sum(0), abs_=2e-3 * float(g.float().abs().sum(0).max()))


@pytest.mark.parametrize("N,T,H,W,C,R", [(2, 8, 16, 22, 256, 88), (2, 4, 16, 22, 512, 44), (1, 6, 9, 21, 64, 7), (2, 16, 32, 43, 128, 64)])
def test_dwconv_tile_backward_with_bn_backward_folded_in(dev, N, T, H, W, C, R):
    """tuber_dwconv_tile_bwd_{data,weight}_bn: the BatchNorm backward of the layer above (dc3 = cA*dz3 + cB*c3 + cC, coefficients from R
    partial rows) formed inside the depthwise backward kernels == tuber_bn_bwd_fa followed by the plain kernels, up to the bf16
    rounding of the dc3 tensor the unfused path stores (the fused path keeps it in fp32): compared against fp32 torch math of the
    composite; dgamma / dbeta of the folded BatchNorm identical to the stand-alone kernel's"""
    M = N * T * H * W
    dz3 = rnd(M, C, dev=dev, seed=1).to(BF)
    c3 = (rnd(M, C, dev=dev, seed=2) * 1.5 + 0.3).to(BF)
    c1 = rnd(M, C, dev=dev, seed=3).to(BF)
    w = rnd(C, 27, dev=dev, seed=4) / 5
    sc1, sh1 = 1.0 + 0.2 * rnd(C, dev=dev, seed=5), 0.3 * rnd(C, dev=dev, seed=6)
    gamma = 1.0 + 0.1 * rnd(C, dev=dev, seed=7)
    mean, invstd = 0.3 + 0.1 * rnd(C, dev=dev, seed=8), 1.0 / (1.5 + 0.1 * rnd(C, dev=dev, seed=9).abs())
    # partial rows whose column sums are the true statistics (split unevenly over R rows)
    s_dz, s_dzx = dz3.float().sum(0), (dz3.float() * c3.float()).sum(0)
    wts = (torch.rand(R, 1, generator=torch.Generator().manual_seed(6)) + 0.1).to(dev)
    wts = wts / wts.sum()
    st0, st1 = (wts * s_dz).contiguous(), (wts * s_dzx).contiguous()
    # fp32 reference of the composite
    xhat_sum = (s_dzx - mean * s_dz) * invstd
    cA, cB = gamma * invstd, -gamma * invstd * invstd * (xhat_sum / M)
    cC = -cB * mean - gamma * invstd * (s_dz / M)
    dc3 = cA * dz3.float() + cB * c3.float() + cC
    a1 = bfr((c1.float() * sc1 + sh1).relu()).view(N, T, H, W, C).permute(0, 4, 1, 2, 3)
    a1 = a1.detach().requires_grad_(True)
    wt = w.view(C, 1, 3, 3, 3).detach().requires_grad_(True)
    out = F.conv3d(a1, wt, padding=1, groups=C)
    out.backward(dc3.view(N, T, H, W, C).permute(0, 4, 1, 2, 3))
    mask = ((c1.float() * sc1 + sh1) > 0).view(N, T, H, W, C)
    dz1_ref = a1.grad.permute(0, 2, 3, 4, 1) * mask
    dw_ref = wt.grad.view(C, 27)
    # fused kernels
    dg, db = torch.full((C,), 0.5, device=dev), torch.full((C,), 0.25, device=dev)
    Rb = lib.query("tuber_dwconv_tile_blocks", N, T, H, W, C)
    o0, o1 = torch.empty(Rb, C, device=dev), torch.empty(Rb, C, device=dev)
    dz1 = torch.empty(M, C, device=dev, dtype=BF)
    lib.call("tuber_dwconv_tile_bwd_data_bn", dz3, c3, st0, st1, R, float(M), gamma, mean, invstd, dg, db, w, c1, sc1, sh1, dz1, o0, o1, N, T, H, W, C)
    close("dw bwd data with bn3 folded in", dz1.view(N, T, H, W, C), dz1_ref)
    close("bn3 dgamma", dg - 0.5, xhat_sum, rel=1e-4, abs_=1e-3 * float(xhat_sum.abs().max()))
    close("bn3 dbeta", db - 0.25, s_dz, rel=1e-4, abs_=1e-3 * float(s_dz.abs().max()))
    close("dz1 stats sum", o0.sum(0), dz1_ref.reshape(M, C).sum(0), abs_=2e-3 * float(dz1_ref.abs().sum(dim=(0, 1, 2, 3)).max()))
    nb = lib.query("tuber_dwconv_tile_wgrad_blocks", N, T, H, W, C)
    part = torch.empty(nb * 27 * C, device=dev)
    dwg = torch.zeros(C, 27, device=dev)
    lib.call("tuber_dwconv_tile_bwd_weight_bn", dz3, c3, st0, st1, R, float(M), gamma, mean, invstd, c1, sc1, sh1, part, dwg, 0, N, T, H, W, C)
    close("dw weight gradient with bn3 folded in", dwg, dw_ref, rel=3e-3, abs_=3e-3 * float(dw_ref.abs().max()))
    # both gradients in ONE launch from ONE staged ring (round 4): dz1 and dgamma / dbeta bit-identical to the data-gradient kernel; the
    # statistics rows and the weight gradient are the same sums in another order (2-channel x 8-column thread map; products grouped by
    # the position of the activation) -> equal to the two-launch form up to fp32 rounding
    dgm, dbm = torch.full((C,), 0.5, device=dev), torch.full((C,), 0.25, device=dev)
    m0, m1 = torch.full((Rb, C), float("nan"), device=dev), torch.full((Rb, C), float("nan"), device=dev)
    dz1m = torch.full((M, C), float("nan"), device=dev, dtype=BF)
    part2 = torch.full((Rb * 27 * C,), float("nan"), device=dev)
    lib.call("tuber_dwconv_tile_bwd_both_bn", dz3, c3, st0, st1, R, float(M), gamma, mean, invstd, dgm, dbm, w, c1, sc1, sh1, dz1m, m0, m1, part2,
             N, T, H, W, C)
    torch.cuda.synchronize()
    assert torch.equal(dz1m, dz1) and torch.equal(dgm, dg) and torch.equal(dbm, db)
    for got, want, what in ((m0, o0, "sum dz rows"), (m1, o1, "sum dz*x rows")):
        close("one-launch " + what, got.sum(0), want.sum(0), abs_=1e-5 * float(want.abs().sum(0).max()))
    dw_both = part2.view(Rb, 27, C).sum(0).t()
    close("one-launch weight gradient vs two-launch", dw_both, dwg, abs_=2e-4 * float(dwg.abs().max()))
    close("one-launch weight gradient vs fp32 reference", dw_both, dw_ref, rel=3e-3, abs_=3e-3 * float(dw_ref.abs().max()))
    # the unfused sequence agrees to the rounding of its bf16 dc3
    dc3_t = torch.empty(M, C, device=dev, dtype=BF)
    dg2, db2 = torch.full((C,), 0.5, device=dev), torch.full((C,), 0.25, device=dev)
    if C % 128 == 0:
        lib.call("tuber_bn_bwd_fa", st0, st1, R, C, float(M), gamma, mean, invstd, dg2, db2, dz3, c3, dc3_t, M)
        assert torch.equal(dg2, dg) and torch.equal(db2, db)        # same fp64 derivation
        dz1_u = torch.empty(M, C, device=dev, dtype=BF)
        lib.call("tuber_dwconv_tile_bwd_data", dc3_t, w, c1, sc1, sh1, dz1_u, o0, o1, N, T, H, W, C)
        close("fused vs unfused dz1", dz1, dz1_u.float(), rel=2 ** -6)


@pytest.mark.parametrize("M", [64 * 700, 64 * 37 + 29, 50])
def test_conv4_bwd_fused(dev, M):
    """tuber_conv4_bwd_fused (layer1's bn4 backward apply + conv4 data gradient + conv4 weight gradient as one persistent kernel)
    against fp32 torch math of the three-kernel sequence it replaces: dc4 = bf16(cA*dz + cB*c4 + cC); dz3 = (dc4 . W4) masked by
    relu'(bn3(c3)) with the per-64-row statistics rows; dW4 = dc4^T . bf16(relu(bn3(c3))) summed over the workgroup slabs.
    Ragged M (last tile partly empty), fewer tiles than workgroups, and many tiles per workgroup."""
    C4, P = 256, 64
    assert lib.query("tuber_conv4_bwd_supported", C4, P) == 1 and lib.query("tuber_conv4_bwd_supported", 512, 128) == 0
    dz = rnd(M, C4, dev=dev, seed=1).to(BF)
    c4 = (rnd(M, C4, dev=dev, seed=2) * 1.3 + 0.2).to(BF)
    c3 = rnd(M, P, dev=dev, seed=3).to(BF)
    W4 = rnd(C4, P, dev=dev, seed=4, scale=P ** -0.5)                      # conv4.weight [C4][P]
    ldw = 256
    w4t = torch.zeros(P, ldw, device=dev, dtype=BF)
    w4t[:, :C4] = W4.t().to(BF)
    cA, cB, cC = 1 + 0.2 * rnd(C4, dev=dev, seed=5), 0.1 * rnd(C4, dev=dev, seed=6), 0.05 * rnd(C4, dev=dev, seed=7)
    sc3, sh3 = 1 + 0.2 * rnd(P, dev=dev, seed=8), 0.3 * rnd(P, dev=dev, seed=9)
    S = lib.query("tuber_conv4_bwd_slabs", M)
    tiles = (M + 63) // 64
    assert S == min(tiles, 512)
    dz3 = torch.full((M, P), float("nan"), device=dev, dtype=BF)
    st0, st1 = torch.full((tiles, P), float("nan"), device=dev), torch.full((tiles, P), float("nan"), device=dev)
    slab = torch.full((S, C4, P), float("nan"), device=dev)
    lib.call("tuber_conv4_bwd_fused", dz, c4, c3, w4t, ldw, cA, cB, cC, sc3, sh3, dz3, st0, st1, slab, M)
    torch.cuda.synchronize()
    dc4 = bfr(cA * dz.float() + cB * c4.float() + cC)
    z3 = c3.float() * sc3 + sh3
    ref3 = (dc4 @ W4.to(BF).float()) * (z3 > 0)
    close("conv4 bwd fused dz3", dz3, ref3)
    ref3r = bfr(ref3)
    pad = tiles * 64 - M
    rows = torch.cat([ref3, torch.zeros(pad, P, device=dev)]).view(tiles, 64, P)
    crow = torch.cat([c3.float(), torch.zeros(pad, P, device=dev)]).view(tiles, 64, P)
    close("conv4 bwd fused stats sum dz3", st0, rows.sum(1), abs_=2e-3 * float(rows.abs().sum(1).max()) + 1e-6)
    close("conv4 bwd fused stats sum dz3*c3", st1, (rows * crow).sum(1), abs_=2e-3 * float((rows * crow).abs().sum(1).max()) + 1e-6)
    dW = slab.sum(0)
    refW = dc4.t() @ bfr(z3.relu())
    close("conv4 bwd fused dW4", dW, refW, rel=2e-3)
    assert bool(torch.isfinite(slab).all())


@pytest.mark.parametrize("M", [40, 64 * 41 + 17, 64 * 700])
def test_entry_conv_fwd_fused(dev, M):
    """tuber_entry_conv_fwd (conv1 + projection-shortcut conv of layer1's first bottleneck from one pass over the block input) against
    the two tuber_gemm_nt(epi 1) launches it replaces (same MFMA k-order) and fp32 torch math; statistics rows per 64-row tile;
    ragged M, fewer tiles than workgroups, several tiles per workgroup; eval mode without statistics"""
    CI, P, C4 = 64, 64, 256
    assert lib.query("tuber_entry_conv_supported", CI, P, C4) == 1 and lib.query("tuber_entry_conv_supported", 256, 64, 256) == 0
    x = rnd(M, CI, dev=dev, seed=1).to(BF)
    W1 = rnd(P, CI, dev=dev, seed=2, scale=CI ** -0.5).to(BF)
    Wd = rnd(C4, CI, dev=dev, seed=3, scale=CI ** -0.5).to(BF)
    tiles = (M + 63) // 64
    c1 = torch.full((M, P), float("nan"), device=dev, dtype=BF)
    cd = torch.full((M, C4), float("nan"), device=dev, dtype=BF)
    a0, a1 = torch.full((tiles, P), float("nan"), device=dev), torch.full((tiles, P), float("nan"), device=dev)
    d0, d1 = torch.full((tiles, C4), float("nan"), device=dev), torch.full((tiles, C4), float("nan"), device=dev)
    lib.call("tuber_entry_conv_fwd", x, W1, CI, Wd, CI, c1, cd, a0, a1, d0, d1, M)
    torch.cuda.synchronize()
    for name, out, W, N, s0, s1 in (("c1", c1, W1, P, a0, a1), ("cd", cd, Wd, C4, d0, d1)):
        with rows64():
            g, r0, r1 = gemm_nt(x, W, M, N, CI, epi=1)
        assert torch.equal(out, g), name                     # K = 64: the same two MFMA steps in the same order
        ref = x.float() @ W.float().t()
        close("entry conv %s vs fp32" % name, out, ref)
        pad = tiles * 64 - M
        rows = torch.cat([ref, torch.zeros(pad, N, device=dev)]).view(tiles, 64, N)
        close("entry conv %s stats sum" % name, s0, rows.sum(1), abs_=2e-3 * float(rows.abs().sum(1).max()) + 1e-6)
        close("entry conv %s stats sumsq" % name, s1, (rows * rows).sum(1), rel=2e-3)
        if r0.shape == s0.shape:
            close("entry conv %s stats rows vs gemm_nt" % name, s0, r0, abs_=1e-3 * float(r0.abs().max()) + 1e-5)
    c1e = torch.empty(M, P, device=dev, dtype=BF)
    cde = torch.empty(M, C4, device=dev, dtype=BF)
    lib.call("tuber_entry_conv_fwd", x, W1, CI, Wd, CI, c1e, cde, None, None, None, None, M)
    torch.cuda.synchronize()
    assert torch.equal(c1e, c1) and torch.equal(cde, cd)
    with pytest.raises(Exception):                          # statistics pointers: all four or none
        lib.call("tuber_entry_conv_fwd", x, W1, CI, Wd, CI, c1e, cde, a0, a1, None, None, M)


@pytest.mark.parametrize("M", [64 * 37 + 9, 64 * 1200])
def test_conv4_bwd_fused_projection_form(dev, M):
    """tuber_conv4_bwd_fused with sc3 = sh3 = NULL: the projection shortcut of a stage's first block (down_sample conv + BatchNorm,
    ir_CSN_152.py:86-87) -- its BatchNorm backward apply, data gradient (no mask: the conv input is the block input, which may be
    negative) and weight gradient (operand = the block input as it is) against fp32 torch math; no statistics rows are written"""
    C4, P = 256, 64
    dz = rnd(M, C4, dev=dev, seed=1).to(BF)
    cd = (rnd(M, C4, dev=dev, seed=2) * 1.3 + 0.2).to(BF)
    x = rnd(M, P, dev=dev, seed=3).to(BF)                                  # about half of it negative
    Wd = rnd(C4, P, dev=dev, seed=4, scale=P ** -0.5)                      # down_sample.0.weight [C4][P]
    wdt = Wd.t().contiguous().to(BF)
    cA, cB, cC = 1 + 0.2 * rnd(C4, dev=dev, seed=5), 0.1 * rnd(C4, dev=dev, seed=6), 0.05 * rnd(C4, dev=dev, seed=7)
    S = lib.query("tuber_conv4_bwd_slabs", M)
    dx = torch.full((M, P), float("nan"), device=dev, dtype=BF)
    slab = torch.full((S, C4, P), float("nan"), device=dev)
    guard = torch.full((4, P), 7.0, device=dev)
    lib.call("tuber_conv4_bwd_fused", dz, cd, x, wdt, C4, cA, cB, cC, None, None, dx, None, None, slab, M)
    torch.cuda.synchronize()
    dcd = bfr(cA * dz.float() + cB * cd.float() + cC)
    close("projection form dx", dx, dcd @ Wd.to(BF).float())
    close("projection form dWd", slab.sum(0), dcd.t() @ x.float(), rel=2e-3)
    assert bool(torch.isfinite(slab).all()) and bool((guard == 7.0).all())
    # the statistics pointers are mandatory in the conv4 form
    with pytest.raises(Exception):
        lib.call("tuber_conv4_bwd_fused", dz, cd, x, wdt, C4, cA, cB, cC, cA[:P], cA[:P], dx, None, None, slab, M)


@pytest.mark.parametrize("M,PN,proj", [(64 * 600, 64, False), (64 * 41 + 17, 64, True), (64 * 300, 128, False), (40, 128, True)])
def test_blockout_conv1_fwd_fused(dev, M, PN, proj):
    """tuber_blockout_conv1_fwd (layer1's residual join + the next bottleneck's conv1 as one persistent kernel) against the two
    kernels it replaces: y bit-identical to tuber_block_out_fwd, c1 and its statistics rows equal to tuber_gemm_nt(epi 1) on that y
    (same MFMA k-order: compared to accumulation rounding) and to fp32 torch math; identity and projection shortcut, ragged M"""
    C = 256
    assert lib.query("tuber_blockout_conv1_supported", C, PN) == 1 and lib.query("tuber_blockout_conv1_supported", 512, 128) == 0
    c4 = rnd(M, C, dev=dev, seed=1).to(BF)
    res = rnd(M, C, dev=dev, seed=2).to(BF)
    s4, h4 = 1 + 0.2 * rnd(C, dev=dev, seed=3), 0.3 * rnd(C, dev=dev, seed=4)
    rs, rh = (1 + 0.2 * rnd(C, dev=dev, seed=5), 0.3 * rnd(C, dev=dev, seed=6)) if proj else (None, None)
    W = rnd(PN, C, dev=dev, seed=7, scale=C ** -0.5).to(BF)
    tiles = (M + 63) // 64
    y = torch.empty(M, C, device=dev, dtype=BF)
    c1 = torch.full((M, PN), float("nan"), device=dev, dtype=BF)
    st0, st1 = torch.full((tiles, PN), float("nan"), device=dev), torch.full((tiles, PN), float("nan"), device=dev)
    lib.call("tuber_blockout_conv1_fwd", c4, s4, h4, res, rs, rh, y, W, C, c1, st0, st1, M, PN)
    y_ref = torch.empty(M, C, device=dev, dtype=BF)
    lib.call("tuber_block_out_fwd", c4, s4, h4, res, rs, rh, y_ref, M, C)
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref)
    with rows64():
        c1_ref, r0, r1 = gemm_nt(y_ref, W, M, PN, C, epi=1)
    close("fused c1 vs gemm_nt", c1, c1_ref.float(), rel=2 ** -8)
    ref = y_ref.float() @ W.float().t()
    close("fused c1 vs fp32", c1, ref)
    close("fused stats sum", st0.sum(0), ref.sum(0), abs_=2e-3 * float(ref.abs().sum(0).max()))
    close("fused stats sumsq", st1.sum(0), (ref * ref).sum(0), rel=2e-3)
    close("fused stats rows vs gemm_nt", st0, r0, abs_=1e-3 * float(r0.abs().max()) + 1e-5)
    # eval mode: no statistics rows
    c1e = torch.empty(M, PN, device=dev, dtype=BF)
    lib.call("tuber_blockout_conv1_fwd", c4, s4, h4, res, rs, rh, y, W, C, c1e, None, None, M, PN)
    assert torch.equal(c1e, c1)


@pytest.mark.parametrize("M,join,with_r,with_dw", [(64 * 500, True, True, True), (64 * 23 + 11, True, True, True), (64 * 300, False, True, True),
                                                   (64 * 40, False, False, False), (30, True, False, True), (64 * 310 + 5, 2, True, True)])
def test_conv1_bwd_fused(dev, M, join, with_r, with_dw):
    """tuber_conv1_bwd_fused (layer1's bn1 backward apply + conv1 data gradient [+ the lower block's residual join] + conv1 weight
    gradient as one persistent kernel) against the kernels it replaces: dc1 = bf16(cA*dz1 + cB*c1 + cC); the data gradient through
    tuber_gemm_nt_join (join form: bit-level comparison of dz, statistics rows to accumulation rounding) or plain fp32 math; the weight
    gradient dW1 = dc1^T . x summed over the workgroup slabs.  Ragged M, no residual, frozen conv1 (no slab)."""
    C, P = 256, 64
    assert lib.query("tuber_conv1_bwd_supported", C, P) == 1 and lib.query("tuber_conv1_bwd_supported", 512, 128) == 0
    dz1 = rnd(M, P, dev=dev, seed=1).to(BF)
    c1 = (rnd(M, P, dev=dev, seed=2) * 1.2 + 0.1).to(BF)
    cA, cB, cC = 1 + 0.2 * rnd(P, dev=dev, seed=3), 0.1 * rnd(P, dev=dev, seed=4), 0.05 * rnd(P, dev=dev, seed=5)
    W1 = rnd(P, C, dev=dev, seed=6, scale=P ** -0.5)                        # conv1.weight [P][C]
    ldw = 64
    w1t = W1.t().contiguous().to(BF)                                        # [C][P]
    R = rnd(M, C, dev=dev, seed=7).to(BF) if with_r else None
    X = rnd(M, C, dev=dev, seed=8).to(BF)
    Cm = rnd(M, C, dev=dev, seed=9).to(BF) if join else None
    tiles = (M + 63) // 64
    S = lib.query("tuber_conv1_bwd_slabs", M)
    assert S == min(tiles, 256)
    out = torch.full((M, C), float("nan"), device=dev, dtype=BF)
    st0 = torch.full((tiles, C), float("nan"), device=dev) if join else None
    st1 = torch.full((tiles, C), float("nan"), device=dev) if join else None
    slab = torch.full((S, P, C), float("nan"), device=dev) if with_dw else None
    # join == 2: the lower block is a stage's first block -- a third statistics row (sum dz * cd) for its projection shortcut's BatchNorm
    Cd = rnd(M, C, dev=dev, seed=10).to(BF) if join == 2 else None
    st2 = torch.full((tiles, C), float("nan"), device=dev) if join == 2 else None
    lib.call("tuber_conv1_bwd_fused", dz1, c1, cA, cB, cC, w1t, ldw, R, X, Cm, Cd, out, st0, st1, st2, slab, M)
    torch.cuda.synchronize()
    dc1 = (cA * dz1.float() + cB * c1.float() + cC).to(BF)
    dx = dc1.float() @ W1.to(BF).float() + (R.float() if with_r else 0.0)
    if join:
        ref = bfr(dx) * (X.float() > 0)
        close("conv1 bwd fused dz (join)", out, ref)
        dz_j = torch.empty(M, C, device=dev, dtype=BF)
        rows = lib.query("tuber_gemm_nt_stat_rows", M, C)
        assert rows == tiles
        j0, j1 = torch.zeros(rows, C, device=dev), torch.zeros(rows, C, device=dev)
        lib.call("tuber_gemm_nt_join", dc1, P, w1t, ldw, dz_j, C, M, C, P, R, C, X, C, Cm, C, j0, j1)
        torch.cuda.synchronize()
        ndiff = int((out != dz_j).sum())                 # same products, another MFMA blocking: bf16 ties may round the other way
        assert ndiff <= 1e-4 * out.numel() + 2, "join form: dz differs from tuber_gemm_nt_join in %d elements" % ndiff
        close("conv1 bwd fused dz vs gemm_nt_join", out, dz_j.float(), rel=2 ** -7)
        close("conv1 bwd fused stats sum dz vs gemm_nt_join", st0, j0, abs_=1e-3 * float(j0.abs().max()) + 1e-5)
        close("conv1 bwd fused stats sum dz*c4 vs gemm_nt_join", st1, j1, abs_=1e-3 * float(j1.abs().max()) + 1e-5)
        if join == 2:
            want = (out.float() * Cd.float()).view(-1, C)
            pad = tiles * 64 - M
            want = torch.cat([want, torch.zeros(pad, C, device=dev)]).view(tiles, 64, C).sum(1)
            close("conv1 bwd fused stats sum dz*cd (projection shortcut)", st2, want, abs_=1e-3 * float(want.abs().max()) + 1e-5)
    else:
        close("conv1 bwd fused dx (plain)", out, dx)
    if with_dw:
        close("conv1 bwd fused dW1", slab.sum(0), dc1.float().t() @ X.float(), rel=2e-3)
        assert bool(torch.isfinite(slab).all())


# ---- fp32 kernels of the eval precision mode (csrc/eval_f32.hip): fp32 matrix-pipe linears and the fp32 attention core against fp64 torch math;
# tolerance 2e-5 of the output scale (fp32 accumulation order over K <= 2048) ----
@pytest.mark.parametrize("M,N,K,add_cols,act", [(30, 768, 256, 512, 0), (30, 256, 256, 0, 0), (30, 256, 256, 256, 0), (30, 2048, 256, 0, 1), (30, 256, 2048, 0, 0),
                                                (180, 3, 256, 0, 0), (180, 256, 256, 0, 1), (180, 4, 256, 0, 2), (704, 512, 256, 256, 0), (640, 768, 256, 512, 0),
                                                (257, 100, 32, 0, 1), (1, 16, 64, 0, 0), (17, 33, 96, 0, 2)])
def test_linear_f32(dev, M, N, K, add_cols, act):
    x, add = rnd(M, K + 8, dev=dev, seed=1)[:, :K], rnd(M, K, dev=dev, seed=2)             # x with a leading dimension
    W, b = rnd(N, K, dev=dev, seed=3, scale=K ** -0.5), rnd(N, dev=dev, seed=4)
    y = torch.full((M, N + 4), 7.0, device=dev)                                            # y with a leading dimension: the pad must stay untouched
    lib.call("tuber_linear_f32", x, K + 8, add if add_cols else None, K, add_cols, W, K, b, y, N + 4, M, N, K, act)
    ref = x.double() @ W.double().t() + b.double()
    if add_cols:
        ref[:, :add_cols] += add.double() @ W[:add_cols].double().t()
    ref = ref.relu() if act == 1 else ref.sigmoid() if act == 2 else ref
    close("linear_f32 %dx%dx%d add %d act %d" % (M, N, K, add_cols, act), y[:, :N], ref.float(), rel=2e-5)
    assert bool((y[:, N:] == 7.0).all())


def test_linear_f32_batched_equals_single_launches(dev):
    M, N, K, nb = 704, 512, 256, 6
    x, add = rnd(M, K, dev=dev, seed=1), rnd(M, K, dev=dev, seed=2)
    Wall, ball = rnd(nb, 3 * N // 2 + 4, K, dev=dev, seed=3, scale=K ** -0.5), rnd(nb, N + 12, dev=dev, seed=4)   # weight sets at a constant stride, like the flat parameter buffer
    y = torch.empty(nb, M, N, device=dev)
    lib.call("tuber_linear_f32_batched", x, K, add, K, 256, Wall, K, ball, y, N, M, N, K, 0, nb, Wall.stride(0), ball.stride(0), M * N)
    for z in range(nb):
        y1 = torch.empty(M, N, device=dev)
        lib.call("tuber_linear_f32", x, K, add, K, 256, Wall[z], K, ball[z], y1, N, M, N, K, 0)
        assert torch.equal(y[z], y1), z
    ref = torch.cat([(x + add).double() @ Wall[0, :256].double().t(), x.double() @ Wall[0, 256:N].double().t()], 1) + ball[0, :N].double()
    close("linear_f32_batched set 0", y[0], ref.float(), rel=2e-5)


@pytest.mark.parametrize("B,H,Lq,Lk,masked", [(2, 8, 15, 15, False), (2, 8, 15, 352, True), (2, 8, 15, 352, False), (1, 8, 320, 320, False), (2, 8, 40, 2160, True),
                                              (3, 4, 1, 7, True), (1, 2, 33, 1300, False)])
def test_attention_f32(dev, B, H, Lq, Lk, masked):
    E = 32 * H
    q, kv = rnd(B * Lq, E, dev=dev, seed=1), rnd(B * Lk, 2 * E, dev=dev, seed=2)
    kpm = None
    if masked:
        g = torch.Generator(device="cpu").manual_seed(5)
        kpm = (torch.rand(B, Lk, generator=g) < 0.3)
        kpm[:, 0] = False                                                                   # never a fully masked row (NaN in the reference too)
        kpm = kpm.to(dev).to(torch.uint8)
    o = torch.empty(B * Lq, E, device=dev)
    scale = 32 ** -0.5
    lib.call("tuber_attention_f32", q, E, kv, 2 * E, kv.data_ptr() + 4 * E, 2 * E, o, E, kpm, B, H, Lq, Lk, scale)
    qd = q.double().view(B, Lq, H, 32).transpose(1, 2)
    kd = kv[:, :E].double().view(B, Lk, H, 32).transpose(1, 2)
    vd = kv[:, E:].double().view(B, Lk, H, 32).transpose(1, 2)
    s = qd @ kd.transpose(-1, -2) * scale
    if masked:
        s = s.masked_fill(kpm.bool().view(B, 1, 1, Lk), float("-inf"))
    ref = (s.softmax(-1) @ vd).transpose(1, 2).reshape(B * Lq, E)
    close("attention_f32 B%d H%d Lq%d Lk%d" % (B, H, Lq, Lk), o, ref.float(), rel=2e-5)


@pytest.mark.parametrize("M,N,K", [(5632, 1024, 256), (44032, 512, 128), (704, 2048, 512), (1000, 256, 64)])
def test_gemm_nt_bn_out_eval_tail(dev, M, N, K):
    """conv4 + eval-mode bn4 + residual join + ReLU in one GEMM (tuber_gemm_nt_bn_out) against fp32 math on the operands the kernel sees, and
    against the two launches it replaces (tuber_gemm_nt amode 1 -> bf16 c4 -> tuber_block_out_fwd_f32)."""
    A = rnd(M, K, dev=dev, seed=1).to(BF)
    W = rnd(N, K, dev=dev, seed=2, scale=K ** -0.5).to(BF)
    sc, sh = rnd(K, dev=dev, seed=3).abs() + 0.5, rnd(K, dev=dev, seed=4) * 0.3
    s4, h4 = rnd(N, dev=dev, seed=5).abs() + 0.5, rnd(N, dev=dev, seed=6) * 0.3
    R32 = rnd(M, N, dev=dev, seed=7).abs()
    y, y32 = torch.empty(M, N, device=dev, dtype=BF), torch.empty(M, N, device=dev)
    lib.call("tuber_gemm_nt_bn_out", A, K, sc, sh, W, K, s4, h4, R32, N, y, N, y32, N, M, N, K)
    a = bfr((A.float() * sc + sh).relu())
    ref = (bfr(a @ W.float().t()) * s4 + h4 + R32).relu()            # c4 passes through bf16 (in registers) like in the two-launch path
    close("gemm_nt_bn_out y32 %dx%dx%d" % (M, N, K), y32, ref)
    close("gemm_nt_bn_out y   %dx%dx%d" % (M, N, K), y, ref)
    assert torch.equal(y, y32.to(BF))
    # the two-launch path it replaces: bit-identical
    c4, _, _ = gemm_nt(A, W, M, N, K, amode=1, sc=sc, sh=sh)
    z, z32 = torch.empty(M, N, device=dev, dtype=BF), torch.empty(M, N, device=dev)
    x_bf = torch.zeros(M, N, device=dev, dtype=BF)
    lib.call("tuber_block_out_fwd_f32", c4, s4, h4, x_bf, None, None, R32, z, z32, M, N)
    assert torch.equal(y32, z32) and torch.equal(y, z)


@pytest.mark.parametrize("M,N,K", [(44032, 128, 512), (16896, 256, 2048), (16896, 2048, 256), (8200, 64, 128), (348160, 64, 256)])
def test_gemm_nt_96_row_tiles(dev, M, N, K):
    """round 6: plain-A GEMMs with >= 8 192 rows run on 96 x 64 tiles of the regular pipeline (gemm.hip: nt_use_96).  Against the round-5 tile choice
    (hook tuber_gemm_nt_96_set(0)): outputs BIT-identical (same k order) for the plain (+bias, residual, ReLU), statistics and masked-backward
    epilogues; the statistics rows are per 96 output rows with the unused rows of the [ceil(M / 64)][N] buffer written as zero, so their column sums
    agree to fp32 summation order; and against fp32 math."""
    A = rnd(M, K, dev=dev, seed=1).to(BF)
    B = rnd(N, K, dev=dev, seed=2, scale=K ** -0.5).to(BF)
    bias = rnd(N, dev=dev, seed=3)
    R = rnd(M, N, dev=dev, seed=4).to(BF)
    Cm = rnd(M, N, dev=dev, seed=7).to(BF)
    msc, msh = rnd(N, dev=dev, seed=8).abs() + 0.5, rnd(N, dev=dev, seed=9)
    rows = lib.query("tuber_gemm_nt_stat_rows", M, N)

    def run_all():
        out = {"plain": gemm_nt(A, B, M, N, K, bias=bias, R=R, relu=1)[0]}
        for name, kw in (("stats", dict(epi=1)), ("bwd", dict(epi=2, Cm=Cm, msc=msc, msh=msh))):
            C = torch.empty(M, N, device=dev, dtype=BF)
            st0, st1 = torch.full((rows, N), 7.0, device=dev), torch.full((rows, N), 7.0, device=dev)      # poisoned: every row must be written
            lib.call("tuber_gemm_nt", A, K, B, K, C, N, M, N, K, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0, kw["epi"], None, None, N, 0, 0, st0, st1,
                     kw.get("Cm"), N, kw.get("msc"), kw.get("msh"), 1.0, 0.0, None, 0, None, 0, None)
            out[name] = (C, st0, st1)
        return out
    got = run_all()
    lib.query("tuber_gemm_nt_96_set", 0)
    try:
        base = run_all()
    finally:
        lib.query("tuber_gemm_nt_96_set", 1)
    ref = A.float() @ B.float().t()
    close("96-row plain vs fp32", got["plain"], (ref + bias + R.float()).relu())
    assert torch.equal(got["plain"], base["plain"])
    for name in ("stats", "bwd"):
        assert torch.equal(got[name][0], base[name][0]), name
        for k in (1, 2):
            g, b = got[name][k].double().sum(0), base[name][k].double().sum(0)
            close("96-row %s rows, column sums %d" % (name, k), g.float(), b.float(), rel=2e-4)
        tiles96 = (M + 95) // 96
        assert bool((got[name][1][tiles96:] == 0).all()) and bool((got[name][2][tiles96:] == 0).all()), "rows beyond ceil(M / 96) are zero"
    close("96-row stats sum vs fp32", got["stats"][1].sum(0), ref.sum(0), abs_=2e-3 * float(ref.abs().sum(0).max()))


@pytest.mark.parametrize("M,N,K", [(5632, 1024, 256), (44032, 512, 128), (2816, 2048, 512), (1000, 256, 64)])
def test_join_backward_reads_the_relu_mask_as_a_bit_field(dev, M, N, K):
    """round 6: tuber_block_out_fwd_mask writes y AND the bit field [y > 0] ([M][N / 8] bytes); tuber_gemm_nt_join_mask reads the bits where
    tuber_gemm_nt_join reads y.  Both pairs must be bit-identical: y, dz and both statistics rows."""
    c4 = rnd(M, N, dev=dev, seed=1).to(BF)
    res = rnd(M, N, dev=dev, seed=2).to(BF)
    s4, h4 = rnd(N, dev=dev, seed=3).abs() + 0.5, rnd(N, dev=dev, seed=4) * 0.3
    y0, y1 = torch.empty(M, N, device=dev, dtype=BF), torch.empty(M, N, device=dev, dtype=BF)
    ymask = torch.full((M, N // 8), 0xAA, device=dev, dtype=torch.uint8)
    lib.call("tuber_block_out_fwd", c4, s4, h4, res, None, None, y0, M, N)
    lib.call("tuber_block_out_fwd_mask", c4, s4, h4, res, None, None, y1, ymask, M, N)
    assert torch.equal(y0, y1)
    bits = ((ymask.view(M, N // 8, 1).to(torch.int32) >> torch.arange(8, device=dev, dtype=torch.int32)) & 1).view(M, N).bool()
    assert torch.equal(bits, y0.float() > 0)
    frac = float(bits.float().mean())
    assert 0.2 < frac < 0.8, frac                                       # a mask that masks
    # the join backward on top: dz = (A . B^T + R) * [y > 0], statistics rows sum dz, sum dz * c4
    A = rnd(M, K, dev=dev, seed=5).to(BF)
    W = rnd(N, K, dev=dev, seed=6, scale=K ** -0.5).to(BF)
    R = rnd(M, N, dev=dev, seed=7).to(BF)
    rows = lib.query("tuber_gemm_nt_stat_rows", M, N)
    outs = []
    for masked in (False, True):
        dz = torch.empty(M, N, device=dev, dtype=BF)
        b0, b1 = torch.full((rows, N), 7.0, device=dev), torch.full((rows, N), 7.0, device=dev)
        if masked:
            lib.call("tuber_gemm_nt_join_mask", A, K, W, K, dz, N, M, N, K, R, N, ymask, c4, N, b0, b1)
        else:
            lib.call("tuber_gemm_nt_join", A, K, W, K, dz, N, M, N, K, R, N, y0, N, c4, N, b0, b1)
        outs.append((dz, b0, b1))
    for a, b in zip(outs[0], outs[1]):
        assert torch.equal(a, b)
    ref = (A.float() @ W.float().t() + R.float()) * (y0.float() > 0)
    close("join with bit mask vs fp32", outs[1][0], ref)


def _bits_of(y):
    """the [M][N / 8] bit field of y > 0 (the layout tuber_block_out_fwd_mask writes)"""
    M, N = y.shape
    b = (y.float() > 0).view(M, N // 8, 8).to(torch.int32)
    w = (b << torch.arange(8, device=y.device, dtype=torch.int32)).sum(-1)
    return w.to(torch.uint8).contiguous()


def test_strided_and_first_block_joins_with_the_bit_field_mask(dev):
    """tuber_gemm_nt_join_strided_mask / tuber_gemm_nt_join_ds_mask == their y-reading forms, bit for bit (dz and every statistics row);
    tuber_blockout_conv1_fwd_mask == tuber_blockout_conv1_fwd + the bit field of its y."""
    # strided form: a stage boundary (n, Ti, Hi, Wi) = (2, 4, 16, 22), spatial stride 2
    n, Ti, Hi, Wi, st, ss, N, K = 2, 4, 16, 22, 1, 2, 512, 256
    To, Ho, Wo = (Ti - 1) // st + 1, (Hi - 1) // ss + 1, (Wi - 1) // ss + 1
    M, Mo = n * Ti * Hi * Wi, n * To * Ho * Wo
    A = rnd(M, K, dev=dev, seed=1).to(BF)
    B = rnd(N, K, dev=dev, seed=2, scale=K ** -0.5).to(BF)
    Rs = rnd(Mo, N, dev=dev, seed=3).to(BF)
    Y = rnd(M, N, dev=dev, seed=4).to(BF).relu()
    C4, Cd, Rr = rnd(M, N, dev=dev, seed=5).to(BF), rnd(M, N, dev=dev, seed=6).to(BF), rnd(M, N, dev=dev, seed=7).to(BF)
    ym = _bits_of(Y)
    R1 = lib.query("tuber_gemm_nt_stat_rows", M, N)
    res = []
    for masked in (False, True):
        dz = torch.empty(M, N, device=dev, dtype=BF)
        b0, b1 = torch.full((R1, N), 7.0, device=dev), torch.full((R1, N), 7.0, device=dev)
        if masked:
            lib.call("tuber_gemm_nt_join_strided_mask", A, K, B, K, dz, N, M, N, K, Rs, N, To, Ho, Wo, Ti, Hi, Wi, st, ss, ym, C4, N, b0, b1)
        else:
            lib.call("tuber_gemm_nt_join_strided", A, K, B, K, dz, N, M, N, K, Rs, N, To, Ho, Wo, Ti, Hi, Wi, st, ss, Y, N, C4, N, b0, b1)
        res.append((dz, b0, b1))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    res = []
    for masked in (False, True):
        dz = torch.empty(M, N, device=dev, dtype=BF)
        d0, d1, d2 = (torch.full((R1, N), 7.0, device=dev) for _ in range(3))
        if masked:
            lib.call("tuber_gemm_nt_join_ds_mask", A, K, B, K, dz, N, M, N, K, Rr, N, ym, C4, N, Cd, N, d0, d1, d2)
        else:
            lib.call("tuber_gemm_nt_join_ds", A, K, B, K, dz, N, M, N, K, Rr, N, Y, N, C4, N, Cd, N, d0, d1, d2)
        res.append((dz, d0, d1, d2))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # the layer1 join + next conv1 kernel with the mask output
    Mb, C, PN = 20000, 256, 128
    c4, rs_in = rnd(Mb, C, dev=dev, seed=11).to(BF), rnd(Mb, C, dev=dev, seed=12).to(BF)
    s4, h4 = rnd(C, dev=dev, seed=13).abs() + 0.5, rnd(C, dev=dev, seed=14) * 0.3
    W = rnd(PN, C, dev=dev, seed=15, scale=C ** -0.5).to(BF)
    rows = lib.query("tuber_gemm_nt_stat_rows", Mb, PN)
    outs = []
    for masked in (False, True):
        y, c1 = torch.empty(Mb, C, device=dev, dtype=BF), torch.empty(Mb, PN, device=dev, dtype=BF)
        t0, t1 = torch.zeros(rows, PN, device=dev), torch.zeros(rows, PN, device=dev)
        mk = torch.zeros(Mb, C // 8, device=dev, dtype=torch.uint8)
        if masked:
            lib.call("tuber_blockout_conv1_fwd_mask", c4, s4, h4, rs_in, None, None, y, mk, W, C, c1, t0, t1, Mb, PN)
        else:
            lib.call("tuber_blockout_conv1_fwd", c4, s4, h4, rs_in, None, None, y, W, C, c1, t0, t1, Mb, PN)
        outs.append((y, c1, t0, t1, mk))
    for a, b in zip(outs[0][:4], outs[1][:4]):
        assert torch.equal(a, b)
    assert torch.equal(outs[1][4], _bits_of(outs[1][0]))
