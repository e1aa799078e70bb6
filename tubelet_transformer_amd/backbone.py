"""ir-CSN-50/152 backbone on the HIP kernels (forward + hand-scheduled backward).

API mirror of the reference: ``build_CSN`` / ``ResNeXt`` / ``ResNeXtBottleneck``
(models/backbones/ir_CSN_152.py:33-210, ir_CSN_50.py) -- same module tree and state_dict keys
(``conv1, bn1, layer{1-4}.{i}.{conv1,bn1,conv3,bn3,conv4,bn4,down_sample.{0,1}}``, CSN-50 ``out_fc``).
The nn.Conv3d / nn.BatchNorm3d children are PARAMETER CONTAINERS only: arithmetic runs through
libtuber_hip.so on NDHWC bf16 activations; there is no eager fallback.

Forward schedule per bottleneck (ir_CSN_152.py:70-90), all BN statistics fused into the producers:
    c1 = gemm_nt(x, W1)            [+stats]   -> bn_finalize(bn1)
    c3 = dwconv(relu(bn1(c1)), w3) [+stats]   -> bn_finalize(bn3)
    c4 = gemm_nt(relu(bn3(c3)), W4)[+stats]   -> bn_finalize(bn4)
    cd = gemm_nt(gather(x), Wd)    [+stats]   -> bn_finalize(down_sample.1)      (first block of a stage)
    y  = relu(bn4(c4) + (bn_d(cd) | x))
"""
import torch
from torch import nn

from . import ab, lib
from .engine import TnArgs, WgradQueue

BN_EPS = 1e-3       # ir_CSN_152.py:15
BN_MOM = 0.1        # ir_CSN_152.py:16
BF = torch.bfloat16
CMAX = 2048


class ResNeXtBottleneck(nn.Module):
    """Parameter container with the reference's attribute names (ir_CSN_152.py:33-68)."""

    def __init__(self, in_planes, planes, stride=1, temporal_stride=1, down_sample=None, expansion=2):
        super().__init__()
        self.expansion = expansion
        self.conv1 = nn.Conv3d(in_planes, planes, kernel_size=1, bias=False)
        self.bn1 = nn.BatchNorm3d(planes, eps=BN_EPS, momentum=BN_MOM)
        self.conv3 = nn.Conv3d(planes, planes, kernel_size=3, bias=False, stride=(temporal_stride, stride, stride),
                               padding=1, groups=planes)
        self.bn3 = nn.BatchNorm3d(planes, eps=BN_EPS, momentum=BN_MOM)
        self.conv4 = nn.Conv3d(planes, planes * expansion, kernel_size=1, bias=False)
        self.bn4 = nn.BatchNorm3d(planes * expansion, eps=BN_EPS, momentum=BN_MOM)
        self.relu = nn.ReLU(inplace=True)
        self.down_sample = down_sample
        self.stride = stride
        self.temporal_stride = temporal_stride

    def forward(self, x):
        raise RuntimeError("ResNeXtBottleneck is executed by the fused HIP schedule of ResNeXt.forward")


class ResNeXt(nn.Module):
    """CSN body (ir_CSN_152.py:93-186).  ``forward`` takes an fp32 NCDHW clip batch on the GPU and returns
    ``(features, None)`` where features is the NDHWC bf16 tensor [B, T/8, H/16, W/16, 2048]."""

    def __init__(self, block_nums, num_classes=400, last_stride=True, with_out_fc=False):
        super().__init__()
        self.conv1 = nn.Conv3d(3, 64, kernel_size=(3, 7, 7), stride=(1, 2, 2), padding=(1, 3, 3), bias=False)
        self.bn1 = nn.BatchNorm3d(64, eps=BN_EPS, momentum=BN_MOM)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool3d(kernel_size=(1, 3, 3), stride=(1, 2, 2), padding=(0, 1, 1))
        self.layer1 = self._make_layer(64, 64, block_nums[0], 1, 1)
        self.layer2 = self._make_layer(256, 128, block_nums[1], 2, 2)
        self.layer3 = self._make_layer(512, 256, block_nums[2], 2, 2)
        self.layer4 = self._make_layer(1024, 512, block_nums[3], 2 if last_stride else 1, 2)
        self.avgpool = nn.AdaptiveAvgPool3d(output_size=(1, 1, 1))
        if with_out_fc:  # CSN-50 only (ir_CSN_50.py:137-138); never used in forward
            self.out_fc = nn.Linear(2048, num_classes)
            self.sigmoid = nn.Sigmoid()
        self._runner = None

    @staticmethod
    def _make_layer(in_planes, planes, blocks, stride, temporal_stride, expansion=4):
        ds = nn.Sequential(
            nn.Conv3d(in_planes, planes * expansion, kernel_size=1, stride=(temporal_stride, stride, stride), bias=False),
            nn.BatchNorm3d(planes * expansion, eps=BN_EPS, momentum=BN_MOM))
        layers = [ResNeXtBottleneck(in_planes, planes, stride, temporal_stride, ds, expansion)]
        for _ in range(1, blocks):
            layers.append(ResNeXtBottleneck(planes * expansion, planes, expansion=expansion))
        return nn.Sequential(*layers)

    def forward(self, x):
        raise RuntimeError("ResNeXt runs through tubelet_transformer_amd.tuber.DETR (needs the model's ParamStore)")


def build_CSN(cfg):
    """models/backbones/ir_CSN_{50,152}.py:build_CSN: block counts [3,4,6,3] / [3,8,36,3]."""
    name = cfg.CONFIG.MODEL.BACKBONE_NAME
    if name == "CSN-152":
        m = ResNeXt([3, 8, 36, 3], cfg.CONFIG.DATA.NUM_CLASSES, cfg.CONFIG.MODEL.LAST_STRIDE, with_out_fc=False)
    elif name == "CSN-50":
        m = ResNeXt([3, 4, 6, 3], cfg.CONFIG.DATA.NUM_CLASSES, cfg.CONFIG.MODEL.LAST_STRIDE, with_out_fc=True)
    elif name == "CSN-TEST":   # shallow test-only body (2 blocks per stage) used by the well-conditioned bf16 parity tests
        m = ResNeXt([2, 2, 2, 2], cfg.CONFIG.DATA.NUM_CLASSES, cfg.CONFIG.MODEL.LAST_STRIDE, with_out_fc=False)
    else:
        raise ValueError("unsupported BACKBONE_NAME %r (CSN-50 / CSN-152)" % name)
    if cfg.CONFIG.MODEL.PRETRAINED:
        from .checkpoint import load_csn_mat
        load_csn_mat(m, cfg.CONFIG.MODEL.PRETRAIN_BACKBONE_DIR, name)
    return m


# ------------------------------------------------------------------------------------------------
# fused schedule
# ------------------------------------------------------------------------------------------------
class _BN:
    """Raw device pointers of one BatchNorm layer (params in the flat store + per-layer scratch)."""
    __slots__ = ("C", "gamma", "beta", "rmean", "rvar", "nbt", "dgamma", "dbeta", "scale", "shift", "mean", "invstd",
                 "cA", "cB", "cC")


class CSNRunner:
    """Executes ResNeXt forward/backward for one ParamStore.  Pointers are cached as ints; every launch
    goes through the C ABI on torch's current stream."""

    def __init__(self, body: ResNeXt, prefix: str, store):
        self.body, self.store = body, store
        dev = store.device
        self.dev = dev
        bns = [m for m in body.modules() if isinstance(m, nn.BatchNorm3d)]
        self.scratch = torch.zeros(len(bns), 7, CMAX, dtype=torch.float32, device=dev)
        self._bn_index = {}
        self._bn_rows = []
        self.blocks = []
        sp = self.scratch.data_ptr()

        def mk_bn(mod_prefix, mod):
            b = _BN()
            i = len(self._bn_index)
            self._bn_index[mod_prefix] = i
            f, g = store.flat.data_ptr(), store.gflat.data_ptr()
            ow, ob = store.offsets[mod_prefix + ".weight"], store.offsets[mod_prefix + ".bias"]
            b.C = mod.num_features
            b.gamma, b.beta = f + 4 * ow, f + 4 * ob
            b.dgamma, b.dbeta = g + 4 * ow, g + 4 * ob
            b.rmean, b.rvar, b.nbt = mod.running_mean.data_ptr(), mod.running_var.data_ptr(), mod.num_batches_tracked.data_ptr()
            base = sp + 4 * i * 7 * CMAX
            b.scale, b.shift, b.mean, b.invstd, b.cA, b.cB, b.cC = (base + 4 * k * CMAX for k in range(7))
            self._bn_rows.append((i, b))
            return b

        def wptr(name):
            o = store.offsets[name]
            return store.shadow.data_ptr() + 2 * o, store.flat.data_ptr() + 4 * o, store.gflat.data_ptr() + 4 * o

        def tptr(name):
            toff, N, K, ldt = store.tinfo[name]
            return store.tshadow.data_ptr() + 2 * toff, ldt

        self.stem_w32 = store.flat.data_ptr() + 4 * store.offsets[prefix + "conv1.weight"]
        self.stem_g = store.gflat.data_ptr() + 4 * store.offsets[prefix + "conv1.weight"]
        self.stem_wpad = torch.zeros(64, 512, dtype=BF, device=dev)      # k' = (c,kt,kh)*8 + kw packing of conv1.weight
        self.stem_bn = mk_bn(prefix + "bn1", body.bn1)
        for li in range(1, 5):
            layer = getattr(body, "layer%d" % li)
            for bi, blk in enumerate(layer):
                p = "%slayer%d.%d." % (prefix, li, bi)
                d = {"cin": blk.conv1.in_channels, "p": blk.conv1.out_channels, "st": blk.temporal_stride, "ss": blk.stride,
                     "ds": blk.down_sample is not None}
                d["off0"] = store.offsets[p + "conv1.weight"]
                d["stage"], d["first"] = li, bi == 0
                d["mod"] = blk
                d["w1"], _, d["g1"] = wptr(p + "conv1.weight")
                d["w1t"], d["ld1t"] = tptr(p + "conv1.weight")
                _, d["w3"], d["g3"] = wptr(p + "conv3.weight")
                d["w4"], _, d["g4"] = wptr(p + "conv4.weight")
                d["w4t"], d["ld4t"] = tptr(p + "conv4.weight")
                d["bn1"], d["bn3"], d["bn4"] = mk_bn(p + "bn1", blk.bn1), mk_bn(p + "bn3", blk.bn3), mk_bn(p + "bn4", blk.bn4)
                if d["ds"]:
                    d["wd"], _, d["gd"] = wptr(p + "down_sample.0.weight")
                    d["wdt"], d["lddt"] = tptr(p + "down_sample.0.weight")
                    d["bnd"] = mk_bn(p + "down_sample.1", blk.down_sample[1])
                self.blocks.append(d)
        # eval mode: the affine form of EVERY BatchNorm in one launch at the start of the forward (tuber_bn_eval_affine_multi)
        rows = sorted(self._bn_rows, key=lambda r: r[0])
        self._bn_table = torch.tensor([[b.gamma, b.beta, b.rmean, b.rvar, b.scale, b.shift, b.C, 0] for _, b in rows], dtype=torch.int64, device=dev)
        self._bn_cmax = max(b.C for _, b in rows)
        self._affine_ready = False
        self._ws = {}
        self._fa_max = lib.query("tuber_bn_bwd_fa_max_rows")
        # flat offset where the parameters after the CSN body begin (gradient all-reduce slicing, ddp.py)
        body = [store.offsets[n] + (q.numel() + 63) // 64 * 64 for n, q in zip(store.names, store.params) if n.startswith(prefix)]
        self.body_end = max(body)
        self.body_begin = min(store.offsets[n] for n in store.names if n.startswith(prefix))

    # -- workspaces (serialised on the stream, so one of each kind suffices) ---------------------
    def ws(self, key, numel, dtype=torch.float32):
        t = self._ws.get(key)
        if t is None or t.numel() < numel:
            t = torch.empty(int(numel * 1.25) + 64, dtype=dtype, device=self.dev)
            self._ws[key] = t
        return t.data_ptr()

    def _stat_rows(self, st0, st1, R, C):
        """long partial-statistics lists (layer1) get a wide first-stage reduction before the (few-block) finalize kernel"""
        R2 = lib.query("tuber_stat_rows_reduced", R)
        if R2 >= R:
            return st0, st1, R
        o0, o1 = self.ws("st0r", R2 * C), self.ws("st1r", R2 * C)
        lib.call("tuber_stat_rows_reduce", st0, st1, R, C, o0, o1)
        return o0, o1, R2

    def _bn_train(self, bn, st0, st1, R, count):
        st0, st1, R = self._stat_rows(st0, st1, R, bn.C)
        lib.call("tuber_bn_finalize", st0, st1, R, bn.C, float(count), bn.gamma, bn.beta, bn.rmean, bn.rvar, bn.nbt, BN_MOM, BN_EPS,
                 bn.scale, bn.shift, bn.mean, bn.invstd)

    def _bn_eval(self, bn):
        if not self._affine_ready:          # a block range run on its own (tests); forward() has every layer's affine form from one launch
            lib.call("tuber_bn_eval_affine", bn.gamma, bn.beta, bn.rmean, bn.rvar, BN_EPS, bn.scale, bn.shift, bn.C)

    def _gemm_stats(self, A, lda, Wb, ldb, C, M, N, K, amode, sc, sh, gather, bn, train, defer=False):
        """conv as GEMM; in training mode also the following BatchNorm's statistics (``defer``: return the statistics rows (st0, st1, R)
        instead of finalising them -- the consumer does)."""
        g = gather or (0, 0, 0, 0, 0, 0, 0, 0)
        if train:
            R = lib.query("tuber_gemm_nt_stat_rows", M, N)
            st0, st1 = self.ws("st0", R * N), self.ws("st1", R * N)
            lib.call("tuber_gemm_nt", A, lda, Wb, ldb, C, N, M, N, K, amode, sc, sh, 1 if gather else 0, *g, 1, None, None, 0, 0, 0,
                     st0, st1, None, 0, None, None, 1.0, 0.0, None, 0, None, 0, None)
            if defer:
                return st0, st1, R
            self._bn_train(bn, st0, st1, R, M)
        else:
            lib.call("tuber_gemm_nt", A, lda, Wb, ldb, C, N, M, N, K, amode, sc, sh, 1 if gather else 0, *g, 0, None, None, 0, 0, 0,
                     None, None, None, 0, None, None, 1.0, 0.0, None, 0, None, 0, None)
            self._bn_eval(bn)

    # -- forward ------------------------------------------------------------------------------------
    def forward(self, clips, train):
        """clips fp32 [B,3,T,H,W] (contiguous, on device).  Returns (features [B,T',h,w,2048] bf16, saved)."""
        B, _, T, H, W = clips.shape
        dev = self.dev
        Ho, Wo = (H + 6 - 7) // 2 + 1, (W + 6 - 7) // 2 + 1
        M0 = B * T * Ho * Wo
        # stem conv: implicit GEMM straight from the fp32 clip (no patch matrix in HBM), BN statistics fused
        lib.call("tuber_stem_pack_weight", self.stem_w32, self.stem_wpad)
        if not train:
            lib.call("tuber_bn_eval_affine_multi", self._bn_table, self._bn_table.shape[0], self._bn_cmax, BN_EPS)
        self._affine_ready = not train
        try:
            c0 = torch.empty(M0, 64, dtype=BF, device=dev)
            bn0 = self.stem_bn
            if train:
                R = lib.query("tuber_stem_conv_blocks", B, T, H, W)
                st0, st1 = self.ws("st0", R * 64), self.ws("st1", R * 64)
                lib.call("tuber_stem_conv_fwd", clips, self.stem_wpad, c0, st0, st1, B, T, H, W)
                self._bn_train(bn0, st0, st1, R, M0)
            else:
                lib.call("tuber_stem_conv_fwd", clips, self.stem_wpad, c0, None, None, B, T, H, W)
                self._bn_eval(bn0)
            Hp, Wp = (Ho + 2 - 3) // 2 + 1, (Wo + 2 - 3) // 2 + 1
            x = torch.empty(B * T * Hp * Wp, 64, dtype=BF, device=dev)
            arg = torch.empty(B * T * Hp * Wp, 64, dtype=torch.uint8, device=dev) if train else None
            lib.call("tuber_stem_pool_fwd", c0, self.stem_bn.scale, self.stem_bn.shift, x, arg, B * T, Ho, Wo, Hp, Wp)
            saved = {"stem": (clips if train else None, None, c0, arg, (B, T, Ho, Wo, Hp, Wp)), "blocks": [], "lo": 0}
            x, (Ti, Hi, Wi) = self._forward_blocks(x, B, (T, Hp, Wp), 0, len(self.blocks), train, saved["blocks"])
        finally:
            self._affine_ready = False          # a block range run on its own afterwards (run_blocks) derives its own
        feat = x.view(B, Ti, Hi, Wi, 2048)
        return feat, saved

    def _forward_blocks(self, x, B, geom, lo, hi, train, out_saved):
        """bottlenecks [lo, hi) on x = bf16 rows [B*Ti*Hi*Wi, cin] (NDHWC); appends the saved-for-backward tuples to ``out_saved``"""
        dev = self.dev
        Ti, Hi, Wi = geom
        pre_c1 = None               # the next block's conv1 output when the previous block's join kernel already produced it
        pend = None                 # bn1's statistics rows (st0, st1, R, count) when its finalisation is left to the depthwise kernel
        fold1 = train and not ab.on("no_bn1_in_dw_fwd") and not ab.on("dw_register_tiled")
        # eval precision mode (round 6): the residual stream between the bottlenecks stays fp32 (y32), the bf16 copy y is only the operand of
        # the next block's GEMMs -- the rounding points of a bf16-rounded execution of the reference graph (tests/parity_util.py), which the
        # bf16-STORED stream of the training path exceeds by 1.7x on the actor logits and 2x on the boxes (measured on the oracle)
        precise = not train and ab.eval_fp32_stream()
        y32 = None

        def bn1_stats(blk, s0, s1, R, count):
            """bn1 of a stride-1 block is finalised INSIDE its depthwise forward kernel (tuber_dwconv_tile_fwd_bn): one launch less per block"""
            if fold1 and blk["st"] == 1 and blk["ss"] == 1:
                return (s0, s1, R, count)
            self._bn_train(blk["bn1"], s0, s1, R, count)
            return None

        for bi in range(lo, hi):
            d = self.blocks[bi]
            cin, P, st, ss = d["cin"], d["p"], d["st"], d["ss"]
            To, Hq, Wq = (Ti - 1) // st + 1, (Hi - 1) // ss + 1, (Wi - 1) // ss + 1
            Min, Mout = B * Ti * Hi * Wi, B * To * Hq * Wq
            cd = ymask = None
            if pre_c1 is not None:
                c1, pre_c1 = pre_c1, None
            elif (not ab.on("no_entry_conv") and d["ds"] and st == 1 and ss == 1 and lib.query("tuber_entry_conv_supported", cin, P, 4 * P) == 1):
                # layer1's first block: conv1 and the projection-shortcut conv read the same [M, 64] input -- one persistent kernel
                # produces both outputs (and both BatchNorms' statistics rows) from one pass over it (csrc/entry_conv.hip)
                c1 = torch.empty(Min, P, dtype=BF, device=dev)
                cd = torch.empty(Min, 4 * P, dtype=BF, device=dev)
                if train:
                    Rt = (Min + 63) // 64
                    a0, a1 = self.ws("st0", Rt * P), self.ws("st1", Rt * P)
                    e0, e1 = self.ws("std0", Rt * 4 * P), self.ws("std1", Rt * 4 * P)
                else:
                    a0 = a1 = e0 = e1 = None
                lib.call("tuber_entry_conv_fwd", x, d["w1"], cin, d["wd"], cin, c1, cd, a0, a1, e0, e1, Min)
                if train:
                    pend = bn1_stats(d, a0, a1, Rt, Min)
                    self._bn_train(d["bnd"], e0, e1, Rt, Min)
                else:
                    self._bn_eval(d["bn1"])
                    self._bn_eval(d["bnd"])
            else:
                c1 = torch.empty(Min, P, dtype=BF, device=dev)
                if fold1 and st == 1 and ss == 1:
                    pend = self._gemm_stats(x, cin, d["w1"], cin, c1, Min, P, cin, 0, None, None, None, d["bn1"], train, defer=True) + (Min,)
                else:
                    self._gemm_stats(x, cin, d["w1"], cin, c1, Min, P, cin, 0, None, None, None, d["bn1"], train)
            c3 = torch.empty(Mout, P, dtype=BF, device=dev)
            b1, b3, b4 = d["bn1"], d["bn3"], d["bn4"]
            tile = st == 1 and ss == 1 and not ab.on("dw_register_tiled")        # LDS-staged kernels for the stride-1 blocks (47 of 50)
            if train:
                R = lib.query("tuber_dwconv_tile_blocks", B, Ti, Hi, Wi, P) if tile else lib.query("tuber_dwconv_fwd_stat_rows", B, To, Hq, Wq)
                # (its own pair of buffers when the kernel also READS bn1's rows, which sit in st0 / st1)
                st0, st1 = (self.ws("st0b", R * P), self.ws("st1b", R * P)) if pend is not None else (self.ws("st0", R * P), self.ws("st1", R * P))
            else:
                st0 = st1 = None
            if pend is not None:
                p0, p1, pR = self._stat_rows(pend[0], pend[1], pend[2], P)
                lib.call("tuber_dwconv_tile_fwd_bn", c1, p0, p1, pR, float(pend[3]), b1.gamma, b1.beta, b1.rmean, b1.rvar, b1.nbt, BN_MOM, BN_EPS,
                         b1.scale, b1.shift, b1.mean, b1.invstd, d["w3"], c3, st0, st1, B, Ti, Hi, Wi, P)
                pend = None
            elif tile:
                lib.call("tuber_dwconv_tile_fwd", c1, b1.scale, b1.shift, d["w3"], c3, st0, st1, B, Ti, Hi, Wi, P)
            else:
                lib.call("tuber_dwconv_fwd", c1, b1.scale, b1.shift, d["w3"], c3, st0, st1, B, Ti, Hi, Wi, To, Hq, Wq, P, st, ss)
            if train:
                self._bn_train(b3, st0, st1, R, Mout)
            else:
                self._bn_eval(b3)
            y = torch.empty(Mout, 4 * P, dtype=BF, device=dev)
            if precise and not d["ds"] and y32 is not None and not ab.on("no_eval_conv4_join"):
                # eval precision mode, identity block: an eval-mode bn4 is a constant affine map, so conv4 + bn4 + the residual join + ReLU are ONE
                # GEMM (tuber_gemm_nt_bn_out): c4 never reaches HBM, y leaves as the bf16 operand of the next block and as the fp32 stream
                self._bn_eval(b4)
                y32n = torch.empty(Mout, 4 * P, dtype=torch.float32, device=dev)
                lib.call("tuber_gemm_nt_bn_out", c3, P, b3.scale, b3.shift, d["w4"], P, b4.scale, b4.shift, y32, 4 * P, y, 4 * P, y32n, 4 * P, Mout, 4 * P, P)
                y32 = y32n
                x = y
                Ti, Hi, Wi = To, Hq, Wq
                continue
            c4 = torch.empty(Mout, 4 * P, dtype=BF, device=dev)
            self._gemm_stats(c3, P, d["w4"], P, c4, Mout, 4 * P, P, 1, b3.scale, b3.shift, None, b4, train)
            if d["ds"] and cd is None:
                cd = torch.empty(Mout, 4 * P, dtype=BF, device=dev)
                strided = st != 1 or ss != 1
                gather = (To, Hq, Wq, Ti, Hi, Wi, st, ss) if strided else None
                self._gemm_stats(x, cin, d["wd"], cin, cd, Mout, 4 * P, cin, 0, None, None, gather, d["bnd"], train)
            res, rs, rh = (cd, d["bnd"].scale, d["bnd"].shift) if d["ds"] else (x, None, None)
            # layer1 (256-channel block output, the widest activations): the residual join AND the next bottleneck's conv1 (+ its
            # BatchNorm statistics) run as one persistent kernel that keeps the y tile in LDS (csrc/blockout_conv1.hip): y is written
            # once and not read back.  The next block may be layer2's first one (its conv1 is dense; the stride sits on the depthwise conv).
            nxt = self.blocks[bi + 1] if bi + 1 < hi else None
            if precise:
                # (all four stages.  Leaving layer1's three blocks -- 356 MB tensors, most of the mode's cost: 5.8 instead of 6.3 ms per 2-clip eval batch -- on
                #  the bf16 stream and their fused kernels was built and measured: the oracle puts that at +6 % on the actor logits, the MI355X at 1.15e-2 ->
                #  1.96e-2 on config 3, 0.04e-2 under the tolerance; the mode is there for the margin, so it keeps all of them)
                y32n = torch.empty(Mout, 4 * P, dtype=torch.float32, device=dev)
                if d["ds"]:
                    lib.call("tuber_block_out_fwd_f32", c4, b4.scale, b4.shift, cd, rs, rh, None, y, y32n, Mout, 4 * P)
                else:                   # identity block: the fp32 stream of the block below (a segment that starts here has only the bf16 rows)
                    lib.call("tuber_block_out_fwd_f32", c4, b4.scale, b4.shift, x, None, None, y32, y, y32n, Mout, 4 * P)
                y32 = y32n
            elif (not ab.on("no_blockout_conv1") and nxt is not None and nxt["cin"] == 4 * P
                    and lib.query("tuber_blockout_conv1_supported", 4 * P, nxt["p"]) == 1):
                PN = nxt["p"]
                pre_c1 = torch.empty(Mout, PN, dtype=BF, device=dev)
                if train:
                    Rn = lib.query("tuber_gemm_nt_stat_rows", Mout, PN)
                    n0, n1 = self.ws("st0", Rn * PN), self.ws("st1", Rn * PN)
                else:
                    n0 = n1 = None
                if train and nxt["stage"] != d["stage"] and not ab.on("no_join_mask"):
                    # the last block of layer1: its join backward runs in layer2's first conv1 data-gradient GEMM (strided form) and reads the mask as a bit field
                    ymask = torch.empty(Mout, P // 2, dtype=torch.uint8, device=dev)
                    lib.call("tuber_blockout_conv1_fwd_mask", c4, b4.scale, b4.shift, res, rs, rh, y, ymask, nxt["w1"], nxt["cin"], pre_c1, n0, n1, Mout, PN)
                else:
                    lib.call("tuber_blockout_conv1_fwd", c4, b4.scale, b4.shift, res, rs, rh, y, nxt["w1"], nxt["cin"], pre_c1, n0, n1, Mout, PN)
                if train:
                    pend = bn1_stats(nxt, n0, n1, Rn, Mout)
                else:
                    self._bn_eval(nxt["bn1"])
            elif train and not ab.on("no_join_mask"):
                # training: the ReLU mask of y also leaves as a bit field -- what this block's join backward (inside the conv1
                # data-gradient GEMM of the block above, tuber_gemm_nt_join_mask) reads instead of y: 1 / 16 of the bytes of a side operand of a
                # launch that runs at the bandwidth of its side operands
                ymask = torch.empty(Mout, P // 2, dtype=torch.uint8, device=dev)
                lib.call("tuber_block_out_fwd_mask", c4, b4.scale, b4.shift, res, rs, rh, y, ymask, Mout, 4 * P)
            else:
                lib.call("tuber_block_out_fwd", c4, b4.scale, b4.shift, res, rs, rh, y, Mout, 4 * P)
            if train:
                out_saved.append((x, c1, c3, c4, cd, y, (Ti, Hi, Wi, To, Hq, Wq), ymask))
            x = y
            Ti, Hi, Wi = To, Hq, Wq
        return x, (Ti, Hi, Wi)

    # -- teacher-forced segments (tests: every bottleneck of the real-depth body in isolation, on the oracle's activations) ----------
    def run_blocks(self, x, geom, lo, hi, train=True):
        """bottlenecks [lo, hi) alone: x = bf16 rows [B*Ti*Hi*Wi, cin] of block ``lo``'s input, geom = (B, Ti, Hi, Wi).
        Returns (y rows of block hi-1, (To, Ho, Wo), saved) -- ``saved`` feeds ``backward_blocks``."""
        B, Ti, Hi, Wi = geom
        saved = {"blocks": [], "lo": lo, "B": B}
        y, g = self._forward_blocks(x.contiguous(), B, (Ti, Hi, Wi), lo, hi, train, saved["blocks"])
        return y, g, saved

    def backward_blocks(self, saved, dy, need_dx=True):
        """backward of a ``run_blocks`` segment through the SAME code path as the full body (queued / grouped weight gradients, join
        fusion inside the segment, deferred second-stage reductions -- flushed here): parameter gradients are accumulated into the
        flat gradient buffer; returns the gradient rows of the segment's input."""
        lo = saved["lo"]
        plans, _, _ = self.trainable_plan()
        dx = self._backward_blocks(saved["blocks"], lo, dy.contiguous(), saved["B"], lo, lo + len(saved["blocks"]), plans, need_dx, None)
        self.flush_wgrads()
        self.store.defer.flush()
        return dx

    # -- trainability (requires_grad) ---------------------------------------------------------------------
    # The reference freezes by ``requires_grad = False`` (pretrained recipe: stem + layer1 + layer2, ir_CSN_152.py:251-254,301-303;
    # LR_BACKBONE <= 0: the whole body, backbone_builder.py:38-40) and autograd then neither computes those gradients nor walks the
    # graph below the first trainable tensor.  Same here: weight-gradient kernels and dgamma/dbeta of frozen tensors are not
    # launched / written (their slices of the flat gradient buffer stay zero, ``p.grad`` is None) and the data-gradient chain stops
    # at the lowest block that still has a trainable tensor.  BatchNorm keeps using batch statistics and updating its running
    # buffers in train mode, as the reference's frozen-but-train-mode BatchNorm3d does.
    def trainable_plan(self):
        """([per-block flag dicts], stem flags, index of the lowest block whose backward must run (len(blocks) = none),
        stem backward needed).  Read from ``requires_grad`` on every call: freezing may change between steps."""
        plans = []
        for d in self.blocks:
            m = d["mod"]
            f = {"w1": m.conv1.weight.requires_grad, "w3": m.conv3.weight.requires_grad, "w4": m.conv4.weight.requires_grad,
                 "bn1": m.bn1.weight.requires_grad or m.bn1.bias.requires_grad,
                 "bn3": m.bn3.weight.requires_grad or m.bn3.bias.requires_grad,
                 "bn4": m.bn4.weight.requires_grad or m.bn4.bias.requires_grad}
            if d["ds"]:
                f["wd"] = m.down_sample[0].weight.requires_grad
                f["bnd"] = m.down_sample[1].weight.requires_grad or m.down_sample[1].bias.requires_grad
            f["any"] = any(f.values())
            plans.append(f)
        b = self.body
        stem = {"w": b.conv1.weight.requires_grad, "bn": b.bn1.weight.requires_grad or b.bn1.bias.requires_grad}
        stem["any"] = stem["w"] or stem["bn"]
        lowest = len(self.blocks)
        if stem["any"]:
            lowest = 0
        else:
            for i, f in enumerate(plans):
                if f["any"]:
                    lowest = i
                    break
        return plans, stem, lowest

    def any_trainable(self):
        plans, stem, lowest = self.trainable_plan()
        return stem["any"] or lowest < len(self.blocks)

    # -- backward -------------------------------------------------------------------------------------
    def _bn_bwd(self, bn, st0, st1, R, count, dz, x, M, train=True, apply=True):
        """finalize coefficients (+ dgamma/dbeta into the flat grads when the layer is trainable) and apply: returns dx tensor [M, C]
        (None when ``apply`` is off: only dgamma/dbeta were wanted).
        (Forming dx inside the consuming GEMMs instead -- tuber_gemm_nt amode 2 / tuber_gemm_tn G2 -- removes this kernel and
        7.6 GB/step of HBM traffic but was measured 0.85 ms/step SLOWER on MI355X: the GEMMs are instruction/latency bound,
        not bandwidth bound, and the two-operand prologue costs them more than the apply kernel)"""
        fa = apply and not ab.on("no_bn_bwd_fa") and bn.C % 128 == 0
        if fa and R > self._fa_max and not ab.on("no_bn_bwd_fa_after_reduce"):
            st0, st1, R = self._stat_rows(st0, st1, R, bn.C)       # layer1 / layer2: 64 rows after the first stage -> finalize + apply as one launch
        if fa and R <= self._fa_max:
            # short partial lists (layer3 / layer4 directly): every workgroup of the apply derives its strip's coefficients itself -- one launch
            dx = torch.empty(M, bn.C, dtype=BF, device=self.dev)
            lib.call("tuber_bn_bwd_fa", st0, st1, R, bn.C, float(count), bn.gamma, bn.mean, bn.invstd,
                     bn.dgamma if train else None, bn.dbeta if train else None, dz, x, dx, M)
            return dx
        st0, st1, R = self._stat_rows(st0, st1, R, bn.C)
        lib.call("tuber_bn_bwd_finalize", st0, st1, R, bn.C, float(count), bn.gamma, bn.mean, bn.invstd, bn.cA, bn.cB, bn.cC,
                 bn.dgamma if train else None, bn.dbeta if train else None, 1)
        if not apply:
            return None
        dx = torch.empty(M, bn.C, dtype=BF, device=self.dev)
        lib.call("tuber_bn_bwd_apply", dz, x, bn.cA, bn.cB, bn.cC, dx, M, bn.C)
        return dx

    def _wgrad(self, G, ldg, A, lda, out, M, N, K, amode=0, sc=None, sh=None, gather=None):
        """weight gradient dW[N,K] += G^T f(A).  Nothing consumes it before the optimizer, so it is only QUEUED (engine.WgradQueue:
        operand tensors kept alive) and launched together with its neighbours in one tuber_gemm_tn_group launch."""
        S = lib.query("tuber_gemm_tn_slabs", M, N, K)
        part, acc = self.store.partial("tn", S * N * K, self.ws) if S > 1 else (None, 1)
        g = gather or (0, 0, 0, 0, 0, 0, 0, 0)
        outp = out if isinstance(out, int) else out.data_ptr()
        wq = self.store.wq
        if wq.enabled and (S == 1 or acc == 2) and WgradQueue.eligible(M, N, K, ldg, lda):
            ptr = lambda t: None if t is None else (t if isinstance(t, int) else t.data_ptr())
            wq.add(TnArgs(ptr(G), ldg, ptr(A), lda, ptr(part), outp, acc, M, N, K, amode, 1 if gather else 0, *g, ptr(sc), ptr(sh), None),
                   (G, A), [(part, outp, N * K, N * K, S, 0 if S <= 16 else 1)] if acc == 2 else [])
            return
        lib.call("tuber_gemm_tn", G, ldg, A, lda, part, out, acc, M, N, K, amode, sc, sh, 1 if gather else 0, *g, None, 0, None, None, None, None)
        if acc == 2:
            self.store.defer.add(part, outp, N * K, N * K, S, 0 if S <= 16 else 1)

    def flush_wgrads(self):
        self.store.wq.flush()

    def _backward_blocks(self, sblocks, base, dy, B, lowest, top, plans, dx_below, red):
        """bottlenecks [lowest, top) in reverse; ``sblocks[i - base]`` holds block i's saved tensors; ``dx_below``: the gradient of
        block ``lowest``'s input is wanted (something trainable, or a caller, sits below it).  Returns that gradient (or None)."""
        dev = self.dev
        pre = None          # (dz, sum-dz rows, sum-dz*c4 rows, R) of this block's join backward, produced by the block above (tuber_gemm_nt_join)
        wq = self.store.wq
        for bi in range(top - 1, lowest - 1, -1):
            d, sv, f = self.blocks[bi], sblocks[bi - base], plans[bi]
            x, c1, c3, c4, cd, y, (Ti, Hi, Wi, To, Hq, Wq), _ymask = sv
            cin, P, st, ss = d["cin"], d["p"], d["st"], d["ss"]
            C4 = 4 * P
            Min, Mout = B * Ti * Hi * Wi, B * To * Hq * Wq
            b1, b3, b4 = d["bn1"], d["bn3"], d["bn4"]
            need_dx = dx_below or bi > lowest
            # how deep the chain inside this block has to go: 7 = input gradient, 6 = conv1 weight, 5 = bn1, 4 = conv3 weight,
            # 3 = bn3, 2 = conv4 weight, 1 = bn4 / shortcut only
            depth = 7 if need_dx else (6 if f["w1"] else 5 if f["bn1"] else 4 if f["w3"] else 3 if f["bn3"] else 2 if f["w4"] else 1)
            # join backward: dz + stats of bn4 (and the shortcut BN)
            if pre is not None:
                dz, sa, sb, sc_, R = pre        # (sc_: the projection shortcut's statistics rows when the join of a stage's first block was fused)
                pre = None
            else:
                R = lib.query("tuber_rowblock_count", Mout, C4)
                sa, sb, sc_ = self.ws("st0", R * C4), self.ws("st1", R * C4), self.ws("st2", R * C4)
                dz = torch.empty(Mout, C4, dtype=BF, device=dev)
                lib.call("tuber_block_out_bwd", dy, y, c4, cd, dz, sa, sb, sc_ if d["ds"] else None, Mout, C4)
            dc4 = None
            # layer1 (C4 = 256, P = 64: the widest activations): bn4's backward apply, the conv4 data gradient and the conv4 weight
            # gradient run as ONE persistent kernel that reads dz and c4 once and never writes dc4 (csrc/conv4_bwd.hip)
            fuse4 = (not ab.on("no_conv4_bwd_fused") and depth >= 3 and f["w4"]
                     and lib.query("tuber_conv4_bwd_supported", C4, P) == 1)
            if depth >= 2 or f["bn4"]:
                dc4 = self._bn_bwd(b4, sa, sb, R, Mout, dz, c4, Mout, train=f["bn4"], apply=depth >= 2 and not fuse4)
            dcd = None
            # layer1's projection shortcut (64 -> 256 channels, stride 1): the same persistent kernel in its plain form does the shortcut
            # BatchNorm's backward apply, the projection's data gradient and its weight gradient in one pass over dz and cd
            fused = (not ab.on("no_proj_bwd_fused") and d["ds"] and st == 1 and ss == 1 and need_dx and f["wd"]
                     and lib.query("tuber_conv4_bwd_supported", C4, cin) == 1)
            if d["ds"] and (need_dx or f["wd"] or f["bnd"]):
                dcd = self._bn_bwd(d["bnd"], sa, sc_, R, Mout, dz, cd, Mout, train=f["bnd"], apply=(need_dx or f["wd"]) and not fused)
            # conv4: weight grad (A = relu(bn3(c3)) recomputed on load) and data grad fused with relu/bn3 backward
            if f["w4"] and not fuse4:
                self._wgrad(dc4, C4, c3, P, d["g4"], Mout, C4, P, 1, b3.scale, b3.shift)
            dc3 = None
            tile = st == 1 and ss == 1 and not ab.on("dw_register_tiled")
            # bn3's backward apply (dc3 = cA*dz3 + cB*c3 + cC) is formed INSIDE the two depthwise backward kernels of the stride-1 blocks
            # while they load their gradient operand: every workgroup derives the coefficients of its 64 channels from the partial rows
            # of the conv4 data-gradient GEMM -- the bn_bwd_fa launch and the dc3 round trip through HBM disappear
            R3 = lib.query("tuber_gemm_nt_stat_rows", Mout, P)
            fuse3 = (not ab.on("no_bn3_in_dw") and tile and depth >= 5 and P % 64 == 0
                     and (R3 <= self._fa_max or lib.query("tuber_stat_rows_reduced", R3) <= self._fa_max))
            if depth >= 3:
                s0, s1 = self.ws("st0u" if fuse3 else "st0", R3 * P), self.ws("st1u" if fuse3 else "st1", R3 * P)
                dz3 = torch.empty(Mout, P, dtype=BF, device=dev)
                if fuse4:
                    S4 = lib.query("tuber_conv4_bwd_slabs", Mout)
                    part4, acc4 = self.store.partial("c4f", S4 * C4 * P, self.ws)
                    lib.call("tuber_conv4_bwd_fused", dz, c4, c3, d["w4t"], d["ld4t"], b4.cA, b4.cB, b4.cC, b3.scale, b3.shift,
                             dz3, s0, s1, part4, Mout)
                    g4 = d["g4"]
                    if acc4 == 2:       # second stage deferred to the step's tuber_multi_reduce (mode 1 = the summation order of tuber_reduce_rows)
                        self.store.defer.add(part4, g4 if isinstance(g4, int) else g4.data_ptr(), C4 * P, C4 * P, S4, 1)
                    else:
                        lib.call("tuber_reduce_rows", part4, g4, S4, C4 * P, 1)
                else:
                    lib.call("tuber_gemm_nt", dc4, C4, d["w4t"], d["ld4t"], dz3, P, Mout, P, C4, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                             2, None, None, 0, 0, 0, s0, s1, c3, P, b3.scale, b3.shift, 1.0, 0.0, None, 0, None, 0, None)
                if fuse3:
                    bs0, bs1, bR = (s0, s1, R3) if R3 <= self._fa_max else self._stat_rows(s0, s1, R3, P)
                    bn3 = (dz3, c3, bs0, bs1, bR, float(Mout), b3.gamma, b3.mean, b3.invstd)
                else:
                    dc3 = self._bn_bwd(b3, s0, s1, R3, Mout, dz3, c3, Mout, train=f["bn3"], apply=depth >= 4)
            # depthwise conv: weight grad, data grad fused with relu/bn1 backward
            # (stride-1 blocks with the bn3 fold: ONE launch forms both gradients from one staged ring of dc3 -- 4 tensor passes instead of
            #  the 7 of two kernels; csrc/dwconv_tile.hip: dwconv_tile_bwd_both_kernel)
            both = not ab.on("no_dw_bwd_one_launch") and fuse3 and f["w3"] and depth >= 5
            if both:
                R1 = nb = lib.query("tuber_dwconv_tile_blocks", B, Ti, Hi, Wi, P)      # one [27][P] weight-gradient block per workgroup of the data-gradient grid
                part, acc = self.store.partial("tn", nb * 27 * P, self.ws)
                s0, s1 = self.ws("st0", R1 * P), self.ws("st1", R1 * P)
                dz1 = torch.empty(Min, P, dtype=BF, device=dev)
                lib.call("tuber_dwconv_tile_bwd_both_bn", *bn3, b3.dgamma if f["bn3"] else None, b3.dbeta if f["bn3"] else None,
                         d["w3"], c1, b1.scale, b1.shift, dz1, s0, s1, part, B, Ti, Hi, Wi, P)
                g3 = d["g3"]
                if acc == 2:
                    self.store.defer.add(part, g3 if isinstance(g3, int) else g3.data_ptr(), 27 * P, 27 * P, nb, 1, P)
                else:           # immediate second stage (TUBER_AB=immediate_reduce): the same block sum, launched right here
                    lib.call("tuber_dw_wgrad_reduce", part, g3, nb, P, 1)
            elif f["w3"]:
                nb = lib.query("tuber_dwconv_tile_wgrad_blocks", B, Ti, Hi, Wi, P) if tile else lib.query("tuber_dwconv_bwd_weight_blocks", B, To, Hq, Wq)
                part, acc = self.store.partial("tn", nb * 27 * P, self.ws)

                def dw_wgrad(dc3=dc3, c1=c1, b1=b1, part=part, acc=acc, d=d, nb=nb, tile=tile, geo=(B, Ti, Hi, Wi, To, Hq, Wq, P, st, ss),
                             bn3=bn3 if fuse3 else None):
                    B_, Ti_, Hi_, Wi_, To_, Hq_, Wq_, P_, st_, ss_ = geo
                    if bn3 is not None:
                        lib.call("tuber_dwconv_tile_bwd_weight_bn", *bn3, c1, b1.scale, b1.shift, part, d["g3"], acc, B_, Ti_, Hi_, Wi_, P_)
                    elif tile:
                        lib.call("tuber_dwconv_tile_bwd_weight", dc3, c1, b1.scale, b1.shift, part, d["g3"], acc, B_, Ti_, Hi_, Wi_, P_)
                    else:
                        lib.call("tuber_dwconv_bwd_weight", dc3, c1, b1.scale, b1.shift, part, d["g3"], acc, B_, Ti_, Hi_, Wi_,
                                 To_, Hq_, Wq_, P_, st_, ss_)
                    if acc == 2:
                        g3 = d["g3"]
                        self.store.defer.add(part, g3 if isinstance(g3, int) else g3.data_ptr(), 27 * P_, 27 * P_, nb, 1, P_)
                dw_wgrad()
            dc1 = None
            if depth >= 5:
                if not both:
                    R1 = lib.query("tuber_dwconv_tile_blocks", B, Ti, Hi, Wi, P) if tile else lib.query("tuber_dwconv_bwd_data_stat_rows", B, Ti, Hi, Wi)
                    s0, s1 = self.ws("st0", R1 * P), self.ws("st1", R1 * P)
                    dz1 = torch.empty(Min, P, dtype=BF, device=dev)
                if both:
                    pass
                elif fuse3:
                    lib.call("tuber_dwconv_tile_bwd_data_bn", *bn3, b3.dgamma if f["bn3"] else None, b3.dbeta if f["bn3"] else None,
                             d["w3"], c1, b1.scale, b1.shift, dz1, s0, s1, B, Ti, Hi, Wi, P)
                elif tile:
                    lib.call("tuber_dwconv_tile_bwd_data", dc3, d["w3"], c1, b1.scale, b1.shift, dz1, s0, s1, B, Ti, Hi, Wi, P)
                else:
                    lib.call("tuber_dwconv_bwd_data", dc3, d["w3"], c1, b1.scale, b1.shift, dz1, s0, s1, B, Ti, Hi, Wi, To, Hq, Wq, P, st, ss)
                # layer1 (256-channel block input, P = 64): bn1's backward apply, the conv1 data gradient (with the lower block's join
                # when that is an identity block) and the conv1 weight gradient run as ONE persistent kernel (csrc/conv1_bwd.hip)
                strided_ds = d["ds"] and (st != 1 or ss != 1)
                fuse1 = (not ab.on("no_conv1_bwd_fused") and need_dx and not strided_ds
                         and lib.query("tuber_conv1_bwd_supported", cin, P) == 1)
                dc1 = self._bn_bwd(b1, s0, s1, R1, Min, dz1, c1, Min, train=f["bn1"], apply=depth >= 6 and not fuse1)
            else:
                fuse1 = False
            # conv1: weight grad and data grad (+ identity shortcut gradient as residual)
            if f["w1"] and not fuse1:
                self._wgrad(dc1, P, x, cin, d["g1"], Min, P, cin)
            strided = st != 1 or ss != 1
            gather = (To, Hq, Wq, Ti, Hi, Wi, st, ss) if (d["ds"] and strided) else None
            if d["ds"] and f["wd"] and not fused:
                self._wgrad(dcd, C4, x, cin, d["gd"], Mout, C4, cin, 0, None, None, gather)
            if need_dx:
                res = dz if not d["ds"] else None
                if fused:
                    bd = d["bnd"]
                    dxd = torch.empty(Mout, cin, dtype=BF, device=dev)
                    Sd = lib.query("tuber_conv4_bwd_slabs", Mout)
                    partd, accd = self.store.partial("cdf", Sd * C4 * cin, self.ws)
                    lib.call("tuber_conv4_bwd_fused", dz, cd, x, d["wdt"], d["lddt"], bd.cA, bd.cB, bd.cC, None, None, dxd, None, None, partd, Mout)
                    gd = d["gd"]
                    if accd == 2:
                        self.store.defer.add(partd, gd if isinstance(gd, int) else gd.data_ptr(), C4 * cin, C4 * cin, Sd, 1)
                    else:
                        lib.call("tuber_reduce_rows", partd, gd, Sd, C4 * cin, 1)
                    res = dxd
                elif d["ds"]:
                    dxd = torch.empty(Mout, cin, dtype=BF, device=dev)
                    lib.call("tuber_gemm_nt", dcd, C4, d["wdt"], d["lddt"], dxd, cin, Mout, cin, C4, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0,
                             0, 0, None, None, 0, 0, 0, None, None, None, 0, None, None, 1.0, 0.0, None, 0, None, 0, None)
                    if not strided:
                        res = dxd           # stride-1 projection shortcut: its dense data gradient is the residual input
                # The input gradient dx IS the gradient of the block below's output y (= this block's x).  When that block is an
                # identity block and dx is complete after this GEMM, its join backward (dz = dx * [y > 0] + the bn4 statistics) runs
                # as the GEMM's epilogue: dx never reaches HBM and the block_out_bwd launch of the next iteration is gone.
                fuse = not ab.on("no_join_fusion") and bi - 1 >= lowest and not self.blocks[bi - 1]["ds"] and not (d["ds"] and strided)
                # layer1: the persistent conv1-backward kernel also takes the join of the stage's FIRST block below it (one more LDS image: the
                # projection shortcut's raw output, for its BatchNorm's statistics row) -- that join was a five-tensor block_out_bwd pass (170 us)
                fuse_sr = (not ab.on("no_join_fusion") and not ab.on("no_strided_join_fusion") and bi - 1 >= lowest and not self.blocks[bi - 1]["ds"]
                           and d["ds"] and strided and not fuse1 and Min % (Ti * Hi * Wi) == 0)
                fuse_ds = (not ab.on("no_join_fusion") and not ab.on("no_ds_join_fusion") and bi - 1 >= lowest and self.blocks[bi - 1]["ds"]
                           and not (d["ds"] and strided))
                if fuse1:
                    part1 = None
                    if f["w1"]:
                        S1 = lib.query("tuber_conv1_bwd_slabs", Min)
                        part1, acc1 = self.store.partial("c1f", S1 * P * cin, self.ws)
                    outx = torch.empty(Min, cin, dtype=BF, device=dev)
                    cdl, jc = None, None
                    if fuse or fuse_ds:
                        c4l = sblocks[bi - 1 - base][3]
                        Rj = lib.query("tuber_gemm_nt_stat_rows", Min, cin)
                        ja, jb = self.ws("stj0", Rj * cin), self.ws("stj1", Rj * cin)
                        if fuse_ds:
                            cdl, jc = sblocks[bi - 1 - base][4], self.ws("stj2", Rj * cin)
                    else:
                        c4l, ja, jb = None, None, None
                    lib.call("tuber_conv1_bwd_fused", dz1, c1, b1.cA, b1.cB, b1.cC, d["w1t"], d["ld1t"], res, x, c4l, cdl, outx, ja, jb, jc, part1, Min)
                    if part1 is not None:
                        g1 = d["g1"]
                        if acc1 == 2:
                            self.store.defer.add(part1, g1 if isinstance(g1, int) else g1.data_ptr(), P * cin, P * cin, S1, 1)
                        else:
                            lib.call("tuber_reduce_rows", part1, g1, S1, P * cin, 1)
                    if fuse or fuse_ds:
                        pre = (outx, ja, jb, jc, Rj)
                        dy = None
                    else:
                        dy = outx
                elif fuse:
                    c4l = sblocks[bi - 1 - base][3]
                    Rj = lib.query("tuber_gemm_nt_stat_rows", Min, cin)
                    ja, jb = self.ws("stj0", Rj * cin), self.ws("stj1", Rj * cin)
                    dzl = torch.empty(Min, cin, dtype=BF, device=dev)
                    ym = sblocks[bi - 1 - base][7]                 # the lower block's ReLU mask as a bit field (tuber_block_out_fwd_mask), or None
                    if ym is not None:
                        lib.call("tuber_gemm_nt_join_mask", dc1, P, d["w1t"], d["ld1t"], dzl, cin, Min, cin, P, res, cin, ym, c4l, cin, ja, jb)
                    else:
                        lib.call("tuber_gemm_nt_join", dc1, P, d["w1t"], d["ld1t"], dzl, cin, Min, cin, P, res, cin, x, cin, c4l, cin, ja, jb)
                    pre = (dzl, ja, jb, None, Rj)
                    dy = None
                elif fuse_ds:
                    # the block below is its stage's first block (layer2 / layer3 / layer4): the join epilogue also takes the statistics row of its
                    # projection shortcut's BatchNorm (sum dz*cd) -- no stand-alone five-tensor block_out_bwd
                    sv_l = sblocks[bi - 1 - base]
                    Rj = lib.query("tuber_gemm_nt_stat_rows", Min, cin)
                    ja, jb, jc = self.ws("stj0", Rj * cin), self.ws("stj1", Rj * cin), self.ws("stj2", Rj * cin)
                    dzl = torch.empty(Min, cin, dtype=BF, device=dev)
                    if sv_l[7] is not None:
                        lib.call("tuber_gemm_nt_join_ds_mask", dc1, P, d["w1t"], d["ld1t"], dzl, cin, Min, cin, P, res, cin, sv_l[7], sv_l[3], cin, sv_l[4], cin, ja, jb, jc)
                    else:
                        lib.call("tuber_gemm_nt_join_ds", dc1, P, d["w1t"], d["ld1t"], dzl, cin, Min, cin, P, res, cin, x, cin, sv_l[3], cin, sv_l[4], cin, ja, jb, jc)
                    pre = (dzl, ja, jb, jc, Rj)
                    dy = None
                elif fuse_sr:
                    # a stage's first block above an identity block (layer1 | layer2, layer2 | layer3, layer3 | layer4): the strided projection
                    # shortcut's gradient dxd is added at its sampled rows INSIDE the join epilogue -- no dx tensor, no scatter-add launch, no
                    # stand-alone block_out_bwd pass over the previous stage's widest tensors (128 us at the layer1 | layer2 boundary)
                    c4l = sblocks[bi - 1 - base][3]
                    Rj = lib.query("tuber_gemm_nt_stat_rows", Min, cin)
                    ja, jb = self.ws("stj0", Rj * cin), self.ws("stj1", Rj * cin)
                    dzl = torch.empty(Min, cin, dtype=BF, device=dev)
                    ym = sblocks[bi - 1 - base][7]
                    if ym is not None:
                        lib.call("tuber_gemm_nt_join_strided_mask", dc1, P, d["w1t"], d["ld1t"], dzl, cin, Min, cin, P, dxd, cin, To, Hq, Wq, Ti, Hi, Wi, st, ss,
                                 ym, c4l, cin, ja, jb)
                    else:
                        lib.call("tuber_gemm_nt_join_strided", dc1, P, d["w1t"], d["ld1t"], dzl, cin, Min, cin, P, dxd, cin, To, Hq, Wq, Ti, Hi, Wi, st, ss,
                                 x, cin, c4l, cin, ja, jb)
                    pre = (dzl, ja, jb, None, Rj)
                    dy = None
                else:
                    dx = torch.empty(Min, cin, dtype=BF, device=dev)
                    lib.call("tuber_gemm_nt", dc1, P, d["w1t"], d["ld1t"], dx, cin, Min, cin, P, 0, None, None, 0, 0, 0, 0, 0, 0, 0, 0, 0,
                             0, None, res, cin, 0, 0, None, None, None, 0, None, None, 1.0, 0.0, None, 0, None, 0, None)
                    if d["ds"] and strided:
                        lib.call("tuber_rows_scatter_add", dx, dxd, Mout, To, Hq, Wq, Ti, Hi, Wi, st, ss, cin)
                    dy = dx
            # layer1 / layer2 weight gradients are long GEMMs: launched per bottleneck (their operands are 45-180 MB each);
            # layer3 / layer4 ones are short: up to 8 (four bottlenecks) share a launch
            # stage boundaries at which the gradient windows above them are made FINAL (queued weight-gradient groups launched, deferred
            # second-stage sums landed): always where layer3 ends; the graph-mode DDP step (training.GraphedTrainStep) adds the end of
            # layer4 -- set BEFORE its eager warm-up, so warm-up and capture build the same launch groups and reduce tables
            at_cut = d["first"] and d["stage"] in getattr(self, "cut_stages", (3,))
            if d["stage"] <= 2 or red is not None or at_cut:
                self.flush_wgrads()
            if red is not None:
                self.store.defer.flush()         # the slice handed to RCCL must include the deferred second-stage reductions
                red.notify(d["off0"])
            hook = getattr(self, "split_hook", None)
            if at_cut:
                self.store.defer.flush()
            if hook is not None and at_cut and need_dx:
                # every parameter at flat offsets >= off0 (this stage, the stages above it, everything behind the body) and everything
                # laid out in front of the body (transformer, heads) is final here: the graph-mode DDP step cuts its hipGraph at this
                # point and all-reduces those windows under the backward of the stages below
                hook(d["off0"])
        return dy

    def backward(self, saved, dfeat):
        """dfeat bf16 [B*T'*h*w, 2048] (gradient of the returned features).  Parameter gradients of the TRAINABLE tensors are
        accumulated into the ParamStore's flat gradient buffer; the chain stops at the lowest block with a trainable tensor."""
        dev = self.dev
        dy = dfeat
        B = saved["stem"][4][0]
        red = getattr(self.store, "reducer", None)
        plans, stem_plan, lowest = self.trainable_plan()
        if red is not None:           # everything behind the body (pool decoder of the 'decode' configs) is final ...
            self.store.defer.flush()  # ... once its deferred second-stage sums (LayerNorm / bias / dW partials) have landed
            red.notify(self.body_end, force=True)
        nblk = len(self.blocks)
        dy = self._backward_blocks(saved["blocks"], saved.get("lo", 0), dy, B, lowest, nblk, plans, stem_plan["any"], red)
        self.flush_wgrads()
        if not stem_plan["any"]:
            return
        # stem: pool + relu + bn backward, then the 3->64 conv weight gradient (implicit GEMM over the clip)
        clips, _, c0, arg, (B, T, Ho, Wo, Hp, Wp) = saved["stem"]
        M0 = B * T * Ho * Wo
        R = lib.query("tuber_stem_pool_bwd_stat_rows", M0)
        s0, s1 = self.ws("st0", R * 64), self.ws("st1", R * 64)
        dz0 = torch.empty(M0, 64, dtype=BF, device=dev)
        bn = self.stem_bn
        lib.call("tuber_stem_pool_bwd", dy, arg, c0, bn.scale, bn.shift, dz0, s0, s1, B * T, Ho, Wo, Hp, Wp)
        # the BatchNorm backward apply (dc0 = cA*dz0 + cB*c0 + cC, a 3-pass elementwise kernel over [M0, 64]) is formed inside the
        # weight-gradient kernel while it stages its gradient operand: dc0 never exists in HBM
        fold = not ab.on("no_stem_bn_in_wgrad") and stem_plan["w"]
        dc0 = self._bn_bwd(bn, s0, s1, R, M0, dz0, c0, M0, train=stem_plan["bn"], apply=stem_plan["w"] and not fold)
        if stem_plan["w"]:
            H, W = clips.shape[-2:]
            nwg = lib.query("tuber_stem_conv_wgrad_blocks", B, T, H, W)
            if fold:
                lib.call("tuber_stem_conv_bwd_weight_bn", clips, dz0, c0, bn.cA, bn.cB, bn.cC, self.ws("tn", nwg * 512 * 64), self.stem_g, 1, B, T, H, W)
            else:
                lib.call("tuber_stem_conv_bwd_weight", clips, dc0, self.ws("tn", nwg * 512 * 64), self.stem_g, 1, B, T, H, W)
