// Multi-head attention core (head_dim = 32): softmax(scale * Q K^T + key_padding_mask) [dropout] V
// reference: nn.MultiheadAttention (models/transformer/transformer.py:159,227,237; tuber_ava.py:138;
// transformer_layers.py:81,88) and the hand-rolled copy (transformer_layers.py:156-167,306-366).
//
// Token-major operands are read IN PLACE from the packed projection outputs through a strided
// token map, so no permute / reshape kernels are ever run:
//     row(l, b) = l*sL + (b / B2)*s1 + (b % B2)*s2 ;  element = row*ld + h*32 + d
// This covers the DETR encoder/decoder (batch = clip), the class branch's factorised attention
// (sequence over h*w with batch (layer, clip, t); sequence over t with batch (layer, clip, h*w))
// and the tubelet-query cross attention without materialising any of the reference's
// view/permute/contiguous copies (tuber_ava.py:133-139, transformer_layers.py:77-91).
//
// Round-1 kernel: flash-style online softmax in fp32 on bf16 storage; every query (key) row is split over
// 16 lanes that each walk a strided subset of the keys (queries) and are merged with wave shuffles.
// Sequences are <= 1728 tokens.  Backward recomputes P from the saved
// log-sum-exp (no P tensor in HBM).
#include <cstdlib>

#include "attention.h"

__device__ __forceinline__ long tok_row(const TokMap& m, int l, int b) {
    return (long)l * m.sL + (long)(b / m.B2) * m.s1 + (long)(b % m.B2) * m.s2;
}

__device__ __forceinline__ uint64_t eff_seed(const uint64_t* seed_ptr, uint64_t salt) {
    return (seed_ptr ? *seed_ptr : 0ull) * 0x9E3779B97F4A7C15ull + salt;
}
__device__ __forceinline__ void load_row32(const bf16* p, float (&v)[32]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bf16x8 t = as_bf16x8(((const uint4*)p)[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[i * 8 + e] = bf2f(t[e]);
    }
}
__device__ __forceinline__ void store_row32(bf16* p, const float (&v)[32]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        bf16x8 t;
#pragma unroll
        for (int e = 0; e < 8; ++e) t[e] = f2bf(v[i * 8 + e]);
        ((uint4*)p)[i] = as_uint4(t);
    }
}
__device__ __forceinline__ float dot32(const float (&a)[32], const float (&b)[32]) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int d = 0; d < 32; d += 4) {
        s0 = fmaf(a[d], b[d], s0); s1 = fmaf(a[d + 1], b[d + 1], s1);
        s2 = fmaf(a[d + 2], b[d + 2], s2); s3 = fmaf(a[d + 3], b[d + 3], s3);
    }
    return (s0 + s1) + (s2 + s3);
}

// (round 5: the scalar flash kernels attn_fwd / attn_bwd_dq / attn_bwd_dkv of round 1 -- 16 rows x 16 split lanes -- are gone: their last caller,
// the DETR decoder's self-attention over 15 tubelet queries, runs on the wave-split MFMA kernels of attention_mfma.hip like every other
// attention above SMALL_L)

// ---- tiny sequences (Lq, Lk <= 8: the class branch's attention over the 4 temporal slots, batch = layers x clips x h*w) ----
// one thread per (batch, head, row); every row is a handful of 64-byte loads.  Thread order (round 6): head fastest, then the row, then the batch
// entry -- eight neighbouring lanes read the eight heads of ONE token row (512 contiguous bytes) and the sibling rows of a (batch, head) unit
// sit in the same wave (their K / V rows hit in L1); with the row fastest every lane's 64 bytes came from a different token row, h*w rows apart.
#define SMALL_L 8
__global__ __launch_bounds__(256) void attn_small_fwd_kernel(AttnArgs a) {
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)a.B * a.H * a.Lq) return;
    const int h = (int)(t % a.H), qi = (int)((t / a.H) % a.Lq), b = (int)(t / ((long)a.H * a.Lq));
    const long li = ((long)b * a.H + h) * a.Lq + qi;      // position in the [B][H][Lq] statistics / dropout index space
    float q[32], o[32], s[SMALL_L];
    load_row32(a.Q + tok_row(a.mq, qi, b) * a.mq.ld + h * 32, q);
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < SMALL_L; ++k) {
        s[k] = -INFINITY;
        if (k < a.Lk && !(a.kpm && a.kpm[(long)b * a.Lk + k])) {
            float kv[32];
            load_row32(a.K + tok_row(a.mk, k, b) * a.mk.ld + h * 32, kv);
            s[k] = dot32(q, kv) * a.scale;
            mx = fmaxf(mx, s[k]);
        }
    }
    float l = 0.f;
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = 0.f;
    const float inv_keep = dropout_inv_keep(a.pdrop);
    const uint64_t seed = a.pdrop > 0.f ? eff_seed(a.seed_ptr, a.salt) : 0ull;
#pragma unroll
    for (int k = 0; k < SMALL_L; ++k) {
        if (k < a.Lk && s[k] != -INFINITY) {
            const float p = __expf(s[k] - mx);
            l += p;
            float pv = p;
            if (a.pdrop > 0.f) pv = dropout_keep(seed, (uint64_t)li * a.Lk + k, a.thresh) ? p * inv_keep : 0.f;
            float vv[32];
            load_row32(a.V + tok_row(a.mv, k, b) * a.mv.ld + h * 32, vv);
#pragma unroll
            for (int d = 0; d < 32; ++d) o[d] = fmaf(pv, vv[d], o[d]);
        }
    }
    const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] *= inv;
    store_row32(a.O + tok_row(a.mo, qi, b) * a.mo.ld + h * 32, o);
    if (a.lse) a.lse[li] = mx + __logf(l);
}

// thread i of a (batch, head): dQ of query i (i < Lq) and dK, dV of key i (i < Lk); delta recomputed locally
__global__ __launch_bounds__(256) void attn_small_bwd_kernel(AttnArgs a) {
    const int Lm = max(a.Lq, a.Lk);
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    if (t >= (long)a.B * a.H * Lm) return;
    const int h = (int)(t % a.H), i = (int)((t / a.H) % Lm), b = (int)(t / ((long)a.H * Lm));      // head fastest (see attn_small_fwd_kernel)
    const long bh = (long)b * a.H + h;
    const float inv_keep = dropout_inv_keep(a.pdrop);
    const uint64_t seed = a.pdrop > 0.f ? eff_seed(a.seed_ptr, a.salt) : 0ull;
    if (i < a.Lq) {
        float q[32], dov[32], ov[32], dq[32];
        load_row32(a.Q + tok_row(a.mq, i, b) * a.mq.ld + h * 32, q);
        load_row32(a.dO + tok_row(a.mdo, i, b) * a.mdo.ld + h * 32, dov);
        load_row32(a.O + tok_row(a.mo, i, b) * a.mo.ld + h * 32, ov);
        const float delta = dot32(dov, ov);
        const long li = bh * a.Lq + i;
        const float lse = a.lse[li];
        a.delta[li] = delta;
#pragma unroll
        for (int d = 0; d < 32; ++d) dq[d] = 0.f;
        for (int k = 0; k < a.Lk; ++k) {
            if (a.kpm && a.kpm[(long)b * a.Lk + k]) continue;
            float kv[32], vv[32];
            load_row32(a.K + tok_row(a.mk, k, b) * a.mk.ld + h * 32, kv);
            load_row32(a.V + tok_row(a.mv, k, b) * a.mv.ld + h * 32, vv);
            const float p = __expf(dot32(q, kv) * a.scale - lse);
            float dp = dot32(dov, vv);
            if (a.pdrop > 0.f) dp = dropout_keep(seed, (uint64_t)li * a.Lk + k, a.thresh) ? dp * inv_keep : 0.f;
            const float ds = p * (dp - delta) * a.scale;
#pragma unroll
            for (int d = 0; d < 32; ++d) dq[d] = fmaf(ds, kv[d], dq[d]);
        }
        store_row32(a.dQ + tok_row(a.mdq, i, b) * a.mdq.ld + h * 32, dq);
    }
    if (i < a.Lk) {
        float k[32], v[32], dk[32], dv[32];
        load_row32(a.K + tok_row(a.mk, i, b) * a.mk.ld + h * 32, k);
        load_row32(a.V + tok_row(a.mv, i, b) * a.mv.ld + h * 32, v);
#pragma unroll
        for (int d = 0; d < 32; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
        const bool masked = a.kpm && a.kpm[(long)b * a.Lk + i];
        for (int qq = 0; qq < a.Lq && !masked; ++qq) {
            float qv[32], dov[32], ov[32];
            load_row32(a.Q + tok_row(a.mq, qq, b) * a.mq.ld + h * 32, qv);
            load_row32(a.dO + tok_row(a.mdo, qq, b) * a.mdo.ld + h * 32, dov);
            load_row32(a.O + tok_row(a.mo, qq, b) * a.mo.ld + h * 32, ov);
            const long li = bh * a.Lq + qq;
            const float p = __expf(dot32(qv, k) * a.scale - a.lse[li]);
            float dp = dot32(dov, v), pd = p;
            if (a.pdrop > 0.f) {
                const bool keep = dropout_keep(seed, (uint64_t)li * a.Lk + i, a.thresh);
                pd = keep ? p * inv_keep : 0.f;
                dp = keep ? dp * inv_keep : 0.f;
            }
            const float ds = p * (dp - dot32(dov, ov)) * a.scale;
#pragma unroll
            for (int d = 0; d < 32; ++d) { dv[d] = fmaf(pd, dov[d], dv[d]); dk[d] = fmaf(ds, qv[d], dk[d]); }
        }
        store_row32(a.dK + tok_row(a.mdk, i, b) * a.mdk.ld + h * 32, dk);
        store_row32(a.dV + tok_row(a.mdv, i, b) * a.mdv.ld + h * 32, dv);
    }
}

// ---------------------------------------------------------------------------------------------
// Wide-head, single-query attention of the LSTR pooling decoder (TEMPORAL_DS_STRATEGY 'decode',
// backbone_builder.py:74-78; transformer_layers.py:156-167,306-366): d_model 2048, 8 heads of 256,
// ONE query per pixel attending over the T <= 8 temporal slots of that pixel.
//   q  [NQ, 2048]               rows (b, hw)
//   kv [rows, 4096] = [k | v]   row(pixel, t) = (b*T + t)*HW + hw   (T == 1: row = pixel, self-attention)
// One 256-thread block per pixel: thread = (head, 8-dim slice); the 256-dim dot products are reduced
// over the 32 lanes of a head with shuffles; softmax over T in registers.
// ---------------------------------------------------------------------------------------------
#define WT_MAX 8
__device__ __forceinline__ float half_wave_sum(float v) {   // over the 32 lanes sharing a head
    return xor16_sum(quad16_sum(v));
}
__device__ __forceinline__ void load8(const bf16* p, float (&v)[8]) {
    const bf16x8 t = as_bf16x8(*(const uint4*)p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = bf2f(t[e]);
}
__device__ __forceinline__ void store8(bf16* p, const float (&v)[8]) {
    bf16x8 t;
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = f2bf(v[e]);
    *(uint4*)p = as_uint4(t);
}

template <bool BWD>
__global__ __launch_bounds__(256) void attn_wide_kernel(const bf16* __restrict__ q, const bf16* __restrict__ kv, bf16* __restrict__ o,
                                                        const bf16* __restrict__ dO, bf16* __restrict__ dq, bf16* __restrict__ dkv,
                                                        int HW, int T, float scale, float pdrop, uint32_t thresh, const uint64_t* seed_ptr, uint64_t salt) {
    const uint64_t seed = pdrop > 0.f ? eff_seed(seed_ptr, salt) : 0ull;
    const int pix = blockIdx.x, head = threadIdx.x >> 5, d0 = threadIdx.x * 8;   // d0 = head*256 + lane32*8
    const int b = pix / HW, hwi = pix % HW;
    float qv[8];
    load8(q + (long)pix * 2048 + d0, qv);
    float s[WT_MAX], kk[WT_MAX][8], vv[WT_MAX][8];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < WT_MAX; ++t) {
        if (t < T) {
            const long row = T == 1 ? pix : ((long)b * T + t) * HW + hwi;
            load8(kv + row * 4096 + d0, kk[t]);
            load8(kv + row * 4096 + 2048 + d0, vv[t]);
            float a = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) a = fmaf(qv[e], kk[t][e], a);
            s[t] = half_wave_sum(a) * scale;
            mx = fmaxf(mx, s[t]);
        }
    }
    float l = 0.f, p[WT_MAX], keep[WT_MAX];
    const float inv_keep = dropout_inv_keep(pdrop);
#pragma unroll
    for (int t = 0; t < WT_MAX; ++t)
        if (t < T) { p[t] = __expf(s[t] - mx); l += p[t]; }
#pragma unroll
    for (int t = 0; t < WT_MAX; ++t)
        if (t < T) {
            p[t] /= l;
            keep[t] = 1.f;
            if (pdrop > 0.f) keep[t] = dropout_keep(seed, ((uint64_t)pix * 8 + head) * WT_MAX + t, thresh) ? inv_keep : 0.f;
        }
    if (!BWD) {
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < WT_MAX; ++t)
            if (t < T) {
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(p[t] * keep[t], vv[t][e], acc[e]);
            }
        store8(o + (long)pix * 2048 + d0, acc);
    } else {
        float g[8];
        load8(dO + (long)pix * 2048 + d0, g);
        float dp[WT_MAX], dsum = 0.f;
#pragma unroll
        for (int t = 0; t < WT_MAX; ++t)
            if (t < T) {
                float a = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) a = fmaf(g[e], vv[t][e], a);
                dp[t] = half_wave_sum(a) * keep[t];
                dsum += p[t] * dp[t];
            }
        float dqv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < WT_MAX; ++t)
            if (t < T) {
                const float ds = p[t] * (dp[t] - dsum) * scale;
                float dk[8], dv[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) { dqv[e] = fmaf(ds, kk[t][e], dqv[e]); dk[e] = ds * qv[e]; dv[e] = p[t] * keep[t] * g[e]; }
                const long row = T == 1 ? pix : ((long)b * T + t) * HW + hwi;
                store8(dkv + row * 4096 + d0, dk);
                store8(dkv + row * 4096 + 2048 + d0, dv);
            }
        store8(dq + (long)pix * 2048 + d0, dqv);
    }
}

// everything but short-query x short-key calls takes the MFMA kernels (attention_mfma.hip); measured on MI355X the tubelet-query
// cross attentions (Lq = 15 against 352 / 1408 keys) are faster there too even with one active wave per 64-query block

extern "C" {

// maps: 5 longs each = {ld, sL, s1, s2, B2}
static TokMap mk_map(const long* m) { TokMap t; t.ld = m[0]; t.sL = m[1]; t.s1 = m[2]; t.s2 = m[3]; t.B2 = (int)m[4]; return t; }

int tuber_attn_fwd(const void* Q, const long* mq, const void* K, const long* mk, const void* V, const long* mv, void* O,
                   const long* mo, float* lse, const void* key_padding_mask, int B, int H, int Lq, int Lk, float scale,
                   float pdrop, const void* seed_ptr, unsigned long long salt, hipStream_t stream) {
    if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0 || pdrop < 0.f || pdrop >= 1.f) return TUBER_EINVAL;
    AttnArgs a{};
    a.Q = (const bf16*)Q; a.mq = mk_map(mq); a.K = (const bf16*)K; a.mk = mk_map(mk); a.V = (const bf16*)V; a.mv = mk_map(mv);
    a.O = (bf16*)O; a.mo = mk_map(mo); a.lse = lse; a.kpm = (const uint8_t*)key_padding_mask;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.scale = scale; a.pdrop = pdrop;
    a.thresh = (uint32_t)((double)pdrop * 4294967296.0); a.seed_ptr = (const uint64_t*)seed_ptr; a.salt = salt;
    if (Lq <= SMALL_L && Lk <= SMALL_L)
        hipLaunchKernelGGL(attn_small_fwd_kernel, dim3(ceil_div((long)B * H * Lq, 256)), dim3(256), 0, stream, a);
    else
        tuber_attn_mfma_fwd_launch(&a, stream);
    TUBER_RETURN_LAUNCH();
}

int tuber_attn_bwd(const void* Q, const long* mq, const void* K, const long* mk, const void* V, const long* mv, const void* O,
                   const long* mo, const float* lse, const void* key_padding_mask, const void* dO, const long* mdo, void* dQ,
                   const long* mdq, void* dK, const long* mdk, void* dV, const long* mdv, float* delta, int B, int H, int Lq,
                   int Lk, float scale, float pdrop, const void* seed_ptr, unsigned long long salt, hipStream_t stream) {
    if (B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0 || pdrop < 0.f || pdrop >= 1.f) return TUBER_EINVAL;
    AttnArgs a{};
    a.Q = (const bf16*)Q; a.mq = mk_map(mq); a.K = (const bf16*)K; a.mk = mk_map(mk); a.V = (const bf16*)V; a.mv = mk_map(mv);
    a.O = (bf16*)O; a.mo = mk_map(mo); a.lse = (float*)lse; a.kpm = (const uint8_t*)key_padding_mask;
    a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.scale = scale; a.pdrop = pdrop;
    a.thresh = (uint32_t)((double)pdrop * 4294967296.0); a.seed_ptr = (const uint64_t*)seed_ptr; a.salt = salt;
    a.dO = (const bf16*)dO; a.mdo = mk_map(mdo); a.dQ = (bf16*)dQ; a.mdq = mk_map(mdq); a.dK = (bf16*)dK; a.mdk = mk_map(mdk);
    a.dV = (bf16*)dV; a.mdv = mk_map(mdv); a.delta = delta;
    if (Lq <= SMALL_L && Lk <= SMALL_L) {
        hipLaunchKernelGGL(attn_small_bwd_kernel, dim3(ceil_div((long)B * H * (Lq > Lk ? Lq : Lk), 256)), dim3(256), 0, stream, a);
    } else {
        tuber_attn_mfma_bwd_launch(&a, stream);     // dQ and dK / dV bodies of one launch (two for the class branch's 2 x 2304 workgroups)
    }
    TUBER_RETURN_LAUNCH();
}

int tuber_attn_wide_fwd(const void* q, const void* kv, void* o, int NQ, int HW, int T, float pdrop, const void* seed_ptr,
                        unsigned long long salt, hipStream_t stream) {
    if (T < 1 || T > WT_MAX || NQ <= 0 || pdrop < 0.f || pdrop >= 1.f) return TUBER_EINVAL;
    hipLaunchKernelGGL(attn_wide_kernel<false>, dim3(NQ), dim3(256), 0, stream, (const bf16*)q, (const bf16*)kv, (bf16*)o, nullptr,
                       nullptr, nullptr, HW, T, 0.0625f, pdrop, (uint32_t)((double)pdrop * 4294967296.0), (const uint64_t*)seed_ptr, (uint64_t)salt);
    TUBER_RETURN_LAUNCH();
}
int tuber_attn_wide_bwd(const void* q, const void* kv, const void* dO, void* dq, void* dkv, int NQ, int HW, int T, float pdrop,
                        const void* seed_ptr, unsigned long long salt, hipStream_t stream) {
    if (T < 1 || T > WT_MAX || NQ <= 0 || pdrop < 0.f || pdrop >= 1.f) return TUBER_EINVAL;
    hipLaunchKernelGGL(attn_wide_kernel<true>, dim3(NQ), dim3(256), 0, stream, (const bf16*)q, (const bf16*)kv, nullptr, (const bf16*)dO,
                       (bf16*)dq, (bf16*)dkv, HW, T, 0.0625f, pdrop, (uint32_t)((double)pdrop * 4294967296.0), (const uint64_t*)seed_ptr, (uint64_t)salt);
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
