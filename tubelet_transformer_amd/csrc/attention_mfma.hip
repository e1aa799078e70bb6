// MFMA flash attention for 32-wide heads (gfx950, wave64): the Lq >= 32 cases of tuber_attn_fwd / tuber_attn_bwd --
// the DETR encoder self-attention (352 tokens per clip) and the class branch's factorised spatial attention
// (48 x 8 (batch, head) pairs of 352 x 352), i.e. ~3/4 of the attention time of the training step.
// reference: nn.MultiheadAttention, models/transformer/transformer.py:159, transformer_layers.py:81,88.
//
// One MFMA 16x16x32 covers the whole head dimension, so QK^T costs one instruction per 16 x 16 score tile.  All products are
// issued TRANSPOSED (keys / head dim as MFMA rows, queries or keys as MFMA columns) so that the accumulator layout of a score
// tile (lane = one column, 4 consecutive rows) is already the B-operand layout of the following product -- P never goes
// through LDS.  The 8 k-slots a lane contributes to that second product are rows {g*4..g*4+3} of two stacked 16-row tiles;
// the A operand (V^T, K^T, Q^T, dO^T) is read with the same slot order from the ROW-MAJOR LDS image of V / K / Q / dO by the
// gfx950 hardware transpose read (ds_read_b64_tr_b16) -- no transposed copies are ever written.
//   forward : S^T = K Q^T, online softmax per query column (cross-lane-group max / sum = 2 shuffles), O^T += V^T P^T
//   dQ      : S^T = K Q^T, dP^T = V dO^T, dS^T = P^T (dP^T - delta) scale, dQ^T += K^T dS^T
//   dK, dV  : S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS
// Dropout regenerates the forward mask from (seed, salt, ((b*H+h)*Lq+q)*Lk+k) like the scalar kernels of attention.hip.
//
// Two work splits of the same inner loops (template parameter SPLIT):
//   (backward: dQ and dK / dV are two bodies of ONE launch, attn_mfma_bwd_kernel)
//   SPLIT = false: workgroup = 64 rows (16 per wave), the four waves share each staged 64-row tile of the other operand
//                  (class branch: 384 (b, h) pairs x 6 tiles fill the chip).
//   SPLIT = true : workgroup = 16 rows; all four waves own the SAME 16 rows and take every fourth tile of the loop, each staging
//                  its tiles in a wave-private LDS image (no workgroup barrier inside the loop); the partial results -- online-
//                  softmax states (m, l, O) in the forward, plain sums in the backward -- are merged through LDS at the end.
//                  For the DETR encoder / decoder of a 2-clip batch (16 (b, h) pairs; 96 resp. 16 workgroups of 64 rows) the
//                  kernel time is one wave's dependent chain over 6 key tiles; this cuts the chain to 2 tiles and puts 352 instead
//                  of 96 workgroups on the 256 CUs.
#include "attention.h"

#include "attention_mfma_fwd.h"       // helpers (trow, tile / wave staging, transpose-read fragments) + attn_mfma_fwd_body

namespace {

template <bool SPLIT>
__global__ __launch_bounds__(256) void attn_mfma_fwd_kernel(AttnArgs a) { attn_mfma_fwd_body<SPLIT>(a, blockIdx.x, blockIdx.y, blockIdx.z); }

// ---------------------------------------------------------------------------------------------------------------------
// dQ (and delta = dO . O for the dK/dV kernel): workgroup = 64 queries of one (b, h), loop over key tiles
template <bool SPLIT>
struct BwdLds {                 // staging of ONE backward body (dQ: K, V, key mask;  dK/dV: Q, dO, lse, delta); the merged launch overlays both
    static constexpr int NI = SPLIT ? 4 : 1;
    static constexpr int bytes = 2 * NI * TL * RP * (int)sizeof(bf16) + 2 * NI * TL * (int)sizeof(float);
};
template <bool SPLIT>
__device__ __forceinline__ void attn_mfma_bwd_dq_body(const AttnArgs& a, const int bx, char* sm) {
    constexpr int NI = SPLIT ? 4 : 1;
    bf16 (*ks)[TL][RP] = (bf16 (*)[TL][RP])sm;
    bf16 (*vs)[TL][RP] = (bf16 (*)[TL][RP])(sm + NI * TL * RP * sizeof(bf16));
    uint8_t (*msk)[TL] = (uint8_t (*)[TL])(sm + 2 * NI * TL * RP * sizeof(bf16));
    const int b = blockIdx.z, h = blockIdx.y, q0 = bx * (SPLIT ? 16 : 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    const int im = SPLIT ? wave : 0;
    const int qi = q0 + (SPLIT ? 0 : wave * 16) + li;
    const bool qok = qi < a.Lq;
    const int qc = qok ? qi : a.Lq - 1;
    const bf16x8 qf = as_bf16x8(*(const uint4*)(a.Q + trow(a.mq, qc, b) * a.mq.ld + h * 32 + g * 8));
    const bf16x8 dof = as_bf16x8(*(const uint4*)(a.dO + trow(a.mdo, qc, b) * a.mdo.ld + h * 32 + g * 8));
    float delta;
    {
        const bf16x8 of = as_bf16x8(*(const uint4*)(a.O + trow(a.mo, qc, b) * a.mo.ld + h * 32 + g * 8));
        float d = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) d = fmaf(bf2f(dof[e]), bf2f(of[e]), d);
        delta = group_sum(d);
    }
    const float lse = a.lse[((long)b * a.H + h) * a.Lq + qc];
    if (qok && g == 0 && (!SPLIT || wave == 0)) a.delta[((long)b * a.H + h) * a.Lq + qi] = delta;
    const float inv_keep = dropout_inv_keep(a.pdrop);
    const uint64_t seed = a.pdrop > 0.f ? eseed(a.seed_ptr, a.salt) : 0ull;
    const uint64_t rbase = ((uint64_t)(b * a.H + h) * a.Lq + qc) * (uint64_t)a.Lk;
    f32x4 dq0 = {0.f, 0.f, 0.f, 0.f}, dq1 = {0.f, 0.f, 0.f, 0.f};
    const int kfirst = SPLIT ? wave * TL : 0, kstep = SPLIT ? 4 * TL : TL;
    uint4 kr, vr;
    WaveTile kw, vw;
    if (SPLIT) { if (kfirst < a.Lk) { kw = wave_fetch(a.K, a.mk, b, h, kfirst, a.Lk); vw = wave_fetch(a.V, a.mv, b, h, kfirst, a.Lk); } }
    else { kr = tile_fetch(a.K, a.mk, b, h, 0, a.Lk); vr = tile_fetch(a.V, a.mv, b, h, 0, a.Lk); }
    for (int k0 = kfirst; k0 < a.Lk; k0 += kstep) {
        if (SPLIT) {
            wave_park(ks[im], kw);
            wave_park(vs[im], vw);
            msk[im][lane] = (k0 + lane >= a.Lk) || (a.kpm && a.kpm[(long)b * a.Lk + k0 + lane]);
            if (k0 + kstep < a.Lk) { kw = wave_fetch(a.K, a.mk, b, h, k0 + kstep, a.Lk); vw = wave_fetch(a.V, a.mv, b, h, k0 + kstep, a.Lk); }
            __builtin_amdgcn_wave_barrier();
        } else {
            __syncthreads();
            park_rm(ks[0], kr);
            park_rm(vs[0], vr);
            if (threadIdx.x < TL) msk[0][threadIdx.x] = (k0 + threadIdx.x >= a.Lk) || (a.kpm && a.kpm[(long)b * a.Lk + k0 + threadIdx.x]);
            if (k0 + TL < a.Lk) { kr = tile_fetch(a.K, a.mk, b, h, k0 + TL, a.Lk); vr = tile_fetch(a.V, a.mv, b, h, k0 + TL, a.Lk); }
            __syncthreads();
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int c0 = sub * 32;
            bf16x8 dsf;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x4 s = mfma(frag_rm(ks[im], c0 + t * 16, li, g), qf, f32x4{0.f, 0.f, 0.f, 0.f});
                const f32x4 dp = mfma(frag_rm(vs[im], c0 + t * 16, li, g), dof, f32x4{0.f, 0.f, 0.f, 0.f});
                bool keep[4] = {true, true, true, true};
                if (a.pdrop > 0.f) dropout_keep_run<4>(seed, rbase + k0 + c0 + t * 16 + g * 4, a.thresh, keep);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kl = c0 + t * 16 + g * 4 + r;
                    const float p = msk[im][kl] ? 0.f : __expf(s[r] * a.scale - lse);
                    float dpv = dp[r];
                    if (a.pdrop > 0.f) dpv = keep[r] ? dpv * inv_keep : 0.f;
                    dsf[t * 4 + r] = f2bf(p * (dpv - delta) * a.scale);
                }
            }
            dq0 = mfma(frag_tr_rm(ks[im], 0, c0, li, g), dsf, dq0);
            dq1 = mfma(frag_tr_rm(ks[im], 16, c0, li, g), dsf, dq1);
        }
        if (SPLIT) __builtin_amdgcn_wave_barrier();
    }
    if (SPLIT) {                                           // sum the four waves' partial dQ of the same 16 queries
        __syncthreads();
        float* cmb = (float*)&ks[0][0][0];                 // [4][64][8]
        float* mine = cmb + (wave * 64 + lane) * 8;
#pragma unroll
        for (int r = 0; r < 4; ++r) { mine[r] = dq0[r]; mine[4 + r] = dq1[r]; }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int w = 1; w < 4; ++w)
#pragma unroll
            for (int r = 0; r < 4; ++r) { dq0[r] += cmb[(w * 64 + lane) * 8 + r]; dq1[r] += cmb[(w * 64 + lane) * 8 + 4 + r]; }
    }
    if (qok) {
        bf16* orow = a.dQ + trow(a.mdq, qi, b) * a.mdq.ld + h * 32;
        bf16x4 y0, y1;
#pragma unroll
        for (int r = 0; r < 4; ++r) { y0[r] = f2bf(dq0[r]); y1[r] = f2bf(dq1[r]); }
        *(uint2*)(orow + g * 4) = as_uint2(y0);
        *(uint2*)(orow + 16 + g * 4) = as_uint2(y1);
    }
}

// dK, dV: workgroup = 64 keys of one (b, h), loop over query tiles
// delta = dO . O of one query row (the dK / dV body computes it itself since round 4: it used to read what the dQ kernel had written,
// which made the two launches dependent; now they are ONE launch)
__device__ __forceinline__ float row_delta(const WaveTile& d, const WaveTile& o) {
    float acc = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const bf16x8 x = as_bf16x8(d.c[c]), y = as_bf16x8(o.c[c]);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc = fmaf(bf2f(x[e]), bf2f(y[e]), acc);
    }
    return acc;
}
__device__ __forceinline__ WaveTile row_fetch(const bf16* base, const TokMap& m, int b, int h, int r, int L) {
    WaveTile t;
    const uint4* p = (const uint4*)(base + trow(m, min(r, L - 1), b) * m.ld + h * 32);
#pragma unroll
    for (int c = 0; c < 4; ++c) t.c[c] = p[c];
    return t;
}
// OWN: form delta here (merged launch); !OWN: read what the dQ launch in front of this one has written (the two-launch form)
template <bool SPLIT, bool OWN = true>
__device__ __forceinline__ void attn_mfma_bwd_dkv_body(const AttnArgs& a, const int bx, char* sm) {
    constexpr int NI = SPLIT ? 4 : 1;
    bf16 (*qs)[TL][RP] = (bf16 (*)[TL][RP])sm;
    bf16 (*dos)[TL][RP] = (bf16 (*)[TL][RP])(sm + NI * TL * RP * sizeof(bf16));
    float (*lse_s)[TL] = (float (*)[TL])(sm + 2 * NI * TL * RP * sizeof(bf16));
    float (*del_s)[TL] = (float (*)[TL])(sm + 2 * NI * TL * RP * sizeof(bf16) + NI * TL * sizeof(float));
    const int b = blockIdx.z, h = blockIdx.y, kb = bx * (SPLIT ? 16 : 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    const int im = SPLIT ? wave : 0;
    const int ki = kb + (SPLIT ? 0 : wave * 16) + li;
    const bool kin = ki < a.Lk;
    const int kc = kin ? ki : a.Lk - 1;
    const bool kdead = !kin || (a.kpm && a.kpm[(long)b * a.Lk + kc]);
    const bf16x8 kf = as_bf16x8(*(const uint4*)(a.K + trow(a.mk, kc, b) * a.mk.ld + h * 32 + g * 8));
    const bf16x8 vf = as_bf16x8(*(const uint4*)(a.V + trow(a.mv, kc, b) * a.mv.ld + h * 32 + g * 8));
    const float inv_keep = dropout_inv_keep(a.pdrop);
    const uint64_t seed = a.pdrop > 0.f ? eseed(a.seed_ptr, a.salt) : 0ull;
    const uint64_t bh = (uint64_t)(b * a.H + h);
    f32x4 dk0 = {0.f, 0.f, 0.f, 0.f}, dk1 = {0.f, 0.f, 0.f, 0.f}, dv0 = {0.f, 0.f, 0.f, 0.f}, dv1 = {0.f, 0.f, 0.f, 0.f};
    const int qfirst = SPLIT ? wave * TL : 0, qstep = SPLIT ? 4 * TL : TL;
    const int srow = SPLIT ? lane : (int)threadIdx.x;      // which row of the tile this thread carries lse / delta for
    uint4 qr, dr;
    WaveTile qw, dw, ow;                                   // SPLIT: lane = row of the wave's tile: q, dO and O rows (delta = dO . O at park time)
    WaveTile dro, oro;                                     // !SPLIT: threads 0 .. 63 carry one row of dO and O each for the same purpose
    float lr = 0.f;
    if (SPLIT) {
        if (qfirst < a.Lq) {
            qw = wave_fetch(a.Q, a.mq, b, h, qfirst, a.Lq); dw = wave_fetch(a.dO, a.mdo, b, h, qfirst, a.Lq);
            ow = wave_fetch(a.O, a.mo, b, h, qfirst, a.Lq);
        }
    } else {
        qr = tile_fetch(a.Q, a.mq, b, h, 0, a.Lq); dr = tile_fetch(a.dO, a.mdo, b, h, 0, a.Lq);
        if (OWN && srow < TL) { dro = row_fetch(a.dO, a.mdo, b, h, srow, a.Lq); oro = row_fetch(a.O, a.mo, b, h, srow, a.Lq); }
    }
    float er = 0.f;
    if (srow < TL && qfirst + srow < a.Lq) { lr = a.lse[bh * a.Lq + qfirst + srow]; if (!OWN) er = a.delta[bh * a.Lq + qfirst + srow]; }
    for (int q0 = qfirst; q0 < a.Lq; q0 += qstep) {
        if (SPLIT) {
            wave_park(qs[im], qw);
            wave_park(dos[im], dw);
            lse_s[im][lane] = lr; del_s[im][lane] = row_delta(dw, ow);         // (rows beyond Lq are zero-filled: delta 0)
            if (q0 + qstep < a.Lq) {
                qw = wave_fetch(a.Q, a.mq, b, h, q0 + qstep, a.Lq); dw = wave_fetch(a.dO, a.mdo, b, h, q0 + qstep, a.Lq);
                ow = wave_fetch(a.O, a.mo, b, h, q0 + qstep, a.Lq);
                const int qn = q0 + qstep + lane;
                lr = qn < a.Lq ? a.lse[bh * a.Lq + qn] : 0.f;
            }
            __builtin_amdgcn_wave_barrier();
        } else {
            __syncthreads();
            park_rm(qs[0], qr);
            park_rm(dos[0], dr);
            if (threadIdx.x < TL) { lse_s[0][threadIdx.x] = lr; del_s[0][threadIdx.x] = !OWN ? er : q0 + (int)threadIdx.x < a.Lq ? row_delta(dro, oro) : 0.f; }
            if (q0 + TL < a.Lq) {
                qr = tile_fetch(a.Q, a.mq, b, h, q0 + TL, a.Lq); dr = tile_fetch(a.dO, a.mdo, b, h, q0 + TL, a.Lq);
                const int qn = q0 + TL + threadIdx.x;
                if (threadIdx.x < TL) {
                    lr = qn < a.Lq ? a.lse[bh * a.Lq + qn] : 0.f;
                    if (OWN) { dro = row_fetch(a.dO, a.mdo, b, h, qn, a.Lq); oro = row_fetch(a.O, a.mo, b, h, qn, a.Lq); }
                    else er = qn < a.Lq ? a.delta[bh * a.Lq + qn] : 0.f;
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub) {
            const int c0 = sub * 32;
            bf16x8 pf, dsf;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const f32x4 s = mfma(frag_rm(qs[im], c0 + t * 16, li, g), kf, f32x4{0.f, 0.f, 0.f, 0.f});      // S[q][key li]
                const f32x4 dp = mfma(frag_rm(dos[im], c0 + t * 16, li, g), vf, f32x4{0.f, 0.f, 0.f, 0.f});    // dP[q][key li]
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ql = c0 + t * 16 + g * 4 + r, qq = q0 + ql;
                    const bool live = !kdead && qq < a.Lq;
                    const float p = live ? __expf(s[r] * a.scale - lse_s[im][ql]) : 0.f;
                    float pd = p, dpv = dp[r];
                    if (a.pdrop > 0.f) {
                        const bool keep = dropout_keep(seed, (bh * a.Lq + (uint64_t)(live ? qq : 0)) * (uint64_t)a.Lk + kc, a.thresh);
                        pd = keep ? p * inv_keep : 0.f;
                        dpv = keep ? dpv * inv_keep : 0.f;
                    }
                    pf[t * 4 + r] = f2bf(pd);
                    dsf[t * 4 + r] = f2bf(p * (dpv - del_s[im][ql]) * a.scale);
                }
            }
            dv0 = mfma(frag_tr_rm(dos[im], 0, c0, li, g), pf, dv0);
            dv1 = mfma(frag_tr_rm(dos[im], 16, c0, li, g), pf, dv1);
            dk0 = mfma(frag_tr_rm(qs[im], 0, c0, li, g), dsf, dk0);
            dk1 = mfma(frag_tr_rm(qs[im], 16, c0, li, g), dsf, dk1);
        }
        if (SPLIT) __builtin_amdgcn_wave_barrier();
    }
    if (SPLIT) {                                           // sum the four waves' partial dK / dV of the same 16 keys
        __syncthreads();
        float* cmb = (float*)&qs[0][0][0];                 // [4][64][16] = 16 KB of the 20 KB
        float* mine = cmb + (wave * 64 + lane) * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) { mine[r] = dk0[r]; mine[4 + r] = dk1[r]; mine[8 + r] = dv0[r]; mine[12 + r] = dv1[r]; }
        __syncthreads();
        if (wave != 0) return;
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float* o = cmb + (w * 64 + lane) * 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) { dk0[r] += o[r]; dk1[r] += o[4 + r]; dv0[r] += o[8 + r]; dv1[r] += o[12 + r]; }
        }
    }
    if (kin) {
        bf16* krow = a.dK + trow(a.mdk, ki, b) * a.mdk.ld + h * 32;
        bf16* vrow = a.dV + trow(a.mdv, ki, b) * a.mdv.ld + h * 32;
        bf16x4 y0, y1, z0, z1;
#pragma unroll
        for (int r = 0; r < 4; ++r) { y0[r] = f2bf(dk0[r]); y1[r] = f2bf(dk1[r]); z0[r] = f2bf(dv0[r]); z1[r] = f2bf(dv1[r]); }
        *(uint2*)(krow + g * 4) = as_uint2(y0);
        *(uint2*)(krow + 16 + g * 4) = as_uint2(y1);
        *(uint2*)(vrow + g * 4) = as_uint2(z0);
        *(uint2*)(vrow + 16 + g * 4) = as_uint2(z1);
    }
}

// dQ and dK / dV of one attention in ONE launch: workgroups [0, nq) are query tiles, [nq, gridDim.x) key tiles.  The two halves share no
// data (each forms delta itself), so they fill the chip side by side: the encoder's backward was two dependent launches of 352
// workgroups, 11 + 13 us, twelve times per step.
template <bool SQ, bool SK>
__global__ __launch_bounds__(256) void attn_mfma_bwd_kernel(AttnArgs a, int nq) {
    constexpr int BYTES = BwdLds<SQ>::bytes > BwdLds<SK>::bytes ? BwdLds<SQ>::bytes : BwdLds<SK>::bytes;
    __shared__ __attribute__((aligned(16))) char sm[BYTES];
    if ((int)blockIdx.x < nq) attn_mfma_bwd_dq_body<SQ>(a, blockIdx.x, sm);
    else attn_mfma_bwd_dkv_body<SK>(a, (int)blockIdx.x - nq, sm);
}

// the two bodies as launches of their own: the 64-row form on grids that fill the chip several times over (class branch: 2 304 workgroups
// each) -- there the merged kernel's register budget (the larger of the two bodies) costs occupancy: 154.7 vs 127.3 us per backward
__global__ __launch_bounds__(256) void attn_mfma_bwd_dq_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char sm[BwdLds<false>::bytes];
    attn_mfma_bwd_dq_body<false>(a, blockIdx.x, sm);
}
__global__ __launch_bounds__(256) void attn_mfma_bwd_dkv_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) char sm[BwdLds<false>::bytes];
    attn_mfma_bwd_dkv_body<false, false>(a, blockIdx.x, sm);
}

}  // namespace

// entry points used by tuber_attn_fwd / tuber_attn_bwd (attention.hip) for Lq >= 32; `args` is an AttnArgs.
// The wave-split kernels take over when the 64-row workgroups would leave most of the 256 CUs idle.
static bool attn_split(int rows, int H, int B) {
    constexpr int lim = 256;      // one 64-row workgroup per CU: below that the wave-split form wins (round 2: -0.21 ms/step)
    return (long)ceil_div(rows, 64) * H * B < lim;
}
extern "C" __attribute__((visibility("hidden"))) void tuber_attn_mfma_fwd_launch(const void* args, hipStream_t stream) {
    const AttnArgs& a = *(const AttnArgs*)args;
    if (attn_split(a.Lq, a.H, a.B)) hipLaunchKernelGGL(attn_mfma_fwd_kernel<true>, dim3(ceil_div(a.Lq, 16), a.H, a.B), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(attn_mfma_fwd_kernel<false>, dim3(ceil_div(a.Lq, 64), a.H, a.B), dim3(256), 0, stream, a);
}
extern "C" __attribute__((visibility("hidden"))) void tuber_attn_mfma_bwd_launch(const void* args, hipStream_t stream) {
    const AttnArgs& a = *(const AttnArgs*)args;
    const bool sq = attn_split(a.Lq, a.H, a.B), sk = attn_split(a.Lk, a.H, a.B);
    const int nq = ceil_div(a.Lq, sq ? 16 : 64), nk = ceil_div(a.Lk, sk ? 16 : 64);
    const dim3 grid(nq + nk, a.H, a.B), block(256);
    if (sq && sk) hipLaunchKernelGGL((attn_mfma_bwd_kernel<true, true>), grid, block, 0, stream, a, nq);
    else if (sq) hipLaunchKernelGGL((attn_mfma_bwd_kernel<true, false>), grid, block, 0, stream, a, nq);
    else if (sk) hipLaunchKernelGGL((attn_mfma_bwd_kernel<false, true>), grid, block, 0, stream, a, nq);
    else {
        hipLaunchKernelGGL(attn_mfma_bwd_dq_kernel, dim3(nq, a.H, a.B), block, 0, stream, a);
        hipLaunchKernelGGL(attn_mfma_bwd_dkv_kernel, dim3(nk, a.H, a.B), block, 0, stream, a);
    }
}
