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
// Round-1 kernel: flash-style online softmax, one query row per thread, K/V tiles of 64 keys
// staged in LDS and broadcast-read; fp32 math on bf16 storage.  Sequences are <= 1728 tokens
// (SURVEY.md section 5.7).  Backward recomputes P from the saved log-sum-exp (no P tensor in HBM).
#include "common.h"

struct TokMap { long ld; long sL, s1, s2; int B2; };
__device__ __forceinline__ long tok_row(const TokMap& m, int l, int b) {
    return (long)l * m.sL + (long)(b / m.B2) * m.s1 + (long)(b % m.B2) * m.s2;
}

struct AttnArgs {
    const bf16* Q; TokMap mq;
    const bf16* K; TokMap mk;
    const bf16* V; TokMap mv;
    bf16* O; TokMap mo;
    float* lse;                   // [B][H][Lq]
    const uint8_t* kpm;           // [B][Lk] (1 = padded key) or null
    int B, H, Lq, Lk;
    float scale, pdrop; uint32_t thresh; const uint64_t* seed_ptr; uint64_t salt;
    // backward
    const bf16* dO; TokMap mdo;
    bf16* dQ; TokMap mdq;
    bf16* dK; TokMap mdk;
    bf16* dV; TokMap mdv;
    float* delta;                 // [B][H][Lq]
};

#define KT 64
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
// stage rows [r0, r0+KT) of a strided [L][32] bf16 operand into LDS [KT][32]
__device__ __forceinline__ void stage_tile(bf16 (*dst)[32], const bf16* base, const TokMap& m, int b, int h, int r0, int L) {
    for (int i = threadIdx.x; i < KT * 4; i += blockDim.x) {
        const int r = i >> 2, c = i & 3;
        uint4 v = make_uint4(0, 0, 0, 0);
        if (r0 + r < L) v = *(const uint4*)(base + tok_row(m, r0 + r, b) * m.ld + h * 32 + c * 8);
        *(uint4*)&dst[r][c * 8] = v;
    }
}

__device__ __forceinline__ uint64_t eff_seed(const uint64_t* seed_ptr, uint64_t salt) {
    return (seed_ptr ? *seed_ptr : 0ull) * 0x9E3779B97F4A7C15ull + salt;
}

__global__ __launch_bounds__(128) void attn_fwd_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) bf16 ks[KT][32];
    __shared__ __attribute__((aligned(16))) bf16 vs[KT][32];
    __shared__ uint8_t msk[KT];
    const int b = blockIdx.z, h = blockIdx.y;
    const int qi = blockIdx.x * 128 + threadIdx.x;
    const bool active = qi < a.Lq;
    float q[32], o[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) { q[d] = 0.f; o[d] = 0.f; }
    if (active) load_row32(a.Q + tok_row(a.mq, qi, b) * a.mq.ld + h * 32, q);
#pragma unroll
    for (int d = 0; d < 32; ++d) q[d] *= a.scale;
    float mx = -INFINITY, l = 0.f;
    const float inv_keep = a.pdrop > 0.f ? 1.f / (1.f - a.pdrop) : 1.f;
    const uint64_t rbase = ((uint64_t)(b * a.H + h) * a.Lq + (active ? qi : 0)) * (uint64_t)a.Lk;
    const uint64_t seed = a.pdrop > 0.f ? eff_seed(a.seed_ptr, a.salt) : 0ull;
    for (int k0 = 0; k0 < a.Lk; k0 += KT) {
        __syncthreads();
        stage_tile(ks, a.K, a.mk, b, h, k0, a.Lk);
        stage_tile(vs, a.V, a.mv, b, h, k0, a.Lk);
        if (threadIdx.x < KT) msk[threadIdx.x] = (a.kpm && k0 + threadIdx.x < a.Lk) ? a.kpm[(long)b * a.Lk + k0 + threadIdx.x] : 0;
        __syncthreads();
        const int kn = min(KT, a.Lk - k0);
        for (int c0 = 0; c0 < kn; c0 += 16) {
            float s[16];
            float cmax = -INFINITY;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int kk = c0 + j;
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bf16x8 t = as_bf16x8(*(const uint4*)&ks[kk][i * 8]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc = fmaf(q[i * 8 + e], bf2f(t[e]), acc);
                }
                const bool valid = kk < kn && !msk[kk];
                s[j] = valid ? acc : -INFINITY;
                cmax = fmaxf(cmax, s[j]);
            }
            const float mnew = fmaxf(mx, cmax);
            if (mnew == -INFINITY) continue;          // every key so far masked
            const float alpha = __expf(mx - mnew);
            l *= alpha;
#pragma unroll
            for (int d = 0; d < 32; ++d) o[d] *= alpha;
            mx = mnew;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int kk = c0 + j;
                const float p = __expf(s[j] - mx);     // exp(-inf) = 0 for masked / tail keys
                l += p;
                float pv = p;
                if (a.pdrop > 0.f) pv = dropout_keep(seed, rbase + k0 + kk, a.thresh) ? p * inv_keep : 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bf16x8 t = as_bf16x8(*(const uint4*)&vs[kk][i * 8]);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[i * 8 + e] = fmaf(pv, bf2f(t[e]), o[i * 8 + e]);
                }
            }
        }
    }
    if (active) {
        const float inv = l > 0.f ? 1.f / l : 0.f;
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] *= inv;
        store_row32(a.O + tok_row(a.mo, qi, b) * a.mo.ld + h * 32, o);
        if (a.lse) a.lse[((long)b * a.H + h) * a.Lq + qi] = mx + __logf(l);
    }
}

// dQ: one query row per thread.  Also writes delta = dO . O for the dK/dV kernel.
__global__ __launch_bounds__(128) void attn_bwd_dq_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) bf16 ks[KT][32];
    __shared__ __attribute__((aligned(16))) bf16 vs[KT][32];
    __shared__ uint8_t msk[KT];
    const int b = blockIdx.z, h = blockIdx.y;
    const int qi = blockIdx.x * 128 + threadIdx.x;
    const bool active = qi < a.Lq;
    float q[32], dov[32], dq[32];
    float lse = 0.f, delta = 0.f;
#pragma unroll
    for (int d = 0; d < 32; ++d) { q[d] = 0.f; dov[d] = 0.f; dq[d] = 0.f; }
    if (active) {
        load_row32(a.Q + tok_row(a.mq, qi, b) * a.mq.ld + h * 32, q);
        load_row32(a.dO + tok_row(a.mdo, qi, b) * a.mdo.ld + h * 32, dov);
        float ov[32];
        load_row32(a.O + tok_row(a.mo, qi, b) * a.mo.ld + h * 32, ov);
#pragma unroll
        for (int d = 0; d < 32; ++d) delta = fmaf(dov[d], ov[d], delta);
        lse = a.lse[((long)b * a.H + h) * a.Lq + qi];
        a.delta[((long)b * a.H + h) * a.Lq + qi] = delta;
    }
    const float inv_keep = a.pdrop > 0.f ? 1.f / (1.f - a.pdrop) : 1.f;
    const uint64_t rbase = ((uint64_t)(b * a.H + h) * a.Lq + (active ? qi : 0)) * (uint64_t)a.Lk;
    const uint64_t seed = a.pdrop > 0.f ? eff_seed(a.seed_ptr, a.salt) : 0ull;
    for (int k0 = 0; k0 < a.Lk; k0 += KT) {
        __syncthreads();
        stage_tile(ks, a.K, a.mk, b, h, k0, a.Lk);
        stage_tile(vs, a.V, a.mv, b, h, k0, a.Lk);
        if (threadIdx.x < KT) msk[threadIdx.x] = (a.kpm && k0 + threadIdx.x < a.Lk) ? a.kpm[(long)b * a.Lk + k0 + threadIdx.x] : 0;
        __syncthreads();
        const int kn = min(KT, a.Lk - k0);
        for (int kk = 0; kk < kn; ++kk) {
            if (msk[kk]) continue;
            float s = 0.f, dp = 0.f;
            float kv[32];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16x8 t = as_bf16x8(*(const uint4*)&ks[kk][i * 8]);
                const bf16x8 u = as_bf16x8(*(const uint4*)&vs[kk][i * 8]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    kv[i * 8 + e] = bf2f(t[e]);
                    s = fmaf(q[i * 8 + e], kv[i * 8 + e], s);
                    dp = fmaf(dov[i * 8 + e], bf2f(u[e]), dp);
                }
            }
            const float p = __expf(s * a.scale - lse);
            if (a.pdrop > 0.f) dp = dropout_keep(seed, rbase + k0 + kk, a.thresh) ? dp * inv_keep : 0.f;
            const float ds = p * (dp - delta) * a.scale;
#pragma unroll
            for (int d = 0; d < 32; ++d) dq[d] = fmaf(ds, kv[d], dq[d]);
        }
    }
    if (active) store_row32(a.dQ + tok_row(a.mdq, qi, b) * a.mdq.ld + h * 32, dq);
}

// dK, dV: one key row per thread; Q / dO tiles (+ lse, delta) staged in LDS.
__global__ __launch_bounds__(128) void attn_bwd_dkv_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) bf16 qs[KT][32];
    __shared__ __attribute__((aligned(16))) bf16 dos[KT][32];
    __shared__ float lses[KT], dels[KT];
    const int b = blockIdx.z, h = blockIdx.y;
    const int ki = blockIdx.x * 128 + threadIdx.x;
    const bool active = ki < a.Lk;
    const bool masked = active && a.kpm && a.kpm[(long)b * a.Lk + ki];
    float k[32], v[32], dk[32], dv[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) { k[d] = 0.f; v[d] = 0.f; dk[d] = 0.f; dv[d] = 0.f; }
    if (active) {
        load_row32(a.K + tok_row(a.mk, ki, b) * a.mk.ld + h * 32, k);
        load_row32(a.V + tok_row(a.mv, ki, b) * a.mv.ld + h * 32, v);
    }
    const float inv_keep = a.pdrop > 0.f ? 1.f / (1.f - a.pdrop) : 1.f;
    const uint64_t seed = a.pdrop > 0.f ? eff_seed(a.seed_ptr, a.salt) : 0ull;
    for (int q0 = 0; q0 < a.Lq; q0 += KT) {
        __syncthreads();
        stage_tile(qs, a.Q, a.mq, b, h, q0, a.Lq);
        stage_tile(dos, a.dO, a.mdo, b, h, q0, a.Lq);
        if (threadIdx.x < KT) {
            const bool ok = q0 + threadIdx.x < a.Lq;
            lses[threadIdx.x] = ok ? a.lse[((long)b * a.H + h) * a.Lq + q0 + threadIdx.x] : 0.f;
            dels[threadIdx.x] = ok ? a.delta[((long)b * a.H + h) * a.Lq + q0 + threadIdx.x] : 0.f;
        }
        __syncthreads();
        if (!active || masked) continue;
        const int qn = min(KT, a.Lq - q0);
        for (int qq = 0; qq < qn; ++qq) {
            float s = 0.f, dp = 0.f;
            float qv[32], dv_[32];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16x8 t = as_bf16x8(*(const uint4*)&qs[qq][i * 8]);
                const bf16x8 u = as_bf16x8(*(const uint4*)&dos[qq][i * 8]);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    qv[i * 8 + e] = bf2f(t[e]);
                    dv_[i * 8 + e] = bf2f(u[e]);
                    s = fmaf(qv[i * 8 + e], k[i * 8 + e], s);
                    dp = fmaf(dv_[i * 8 + e], v[i * 8 + e], dp);
                }
            }
            const float p = __expf(s * a.scale - lses[qq]);
            float pd = p;
            if (a.pdrop > 0.f) {
                const uint64_t idx = ((uint64_t)(b * a.H + h) * a.Lq + (q0 + qq)) * (uint64_t)a.Lk + ki;
                const bool keep = dropout_keep(seed, idx, a.thresh);
                pd = keep ? p * inv_keep : 0.f;
                dp = keep ? dp * inv_keep : 0.f;
            }
            const float ds = p * (dp - dels[qq]) * a.scale;
#pragma unroll
            for (int d = 0; d < 32; ++d) { dv[d] = fmaf(pd, dv_[d], dv[d]); dk[d] = fmaf(ds, qv[d], dk[d]); }
        }
    }
    if (active) {
        store_row32(a.dK + tok_row(a.mdk, ki, b) * a.mdk.ld + h * 32, dk);
        store_row32(a.dV + tok_row(a.mdv, ki, b) * a.mdv.ld + h * 32, dv);
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
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4); v += __shfl_xor(v, 8); v += __shfl_xor(v, 16);
    return v;
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
    const float inv_keep = pdrop > 0.f ? 1.f / (1.f - pdrop) : 1.f;
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
    hipLaunchKernelGGL(attn_fwd_kernel, dim3(ceil_div(Lq, 128), H, B), dim3(128), 0, stream, a);
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
    hipLaunchKernelGGL(attn_bwd_dq_kernel, dim3(ceil_div(Lq, 128), H, B), dim3(128), 0, stream, a);
    hipLaunchKernelGGL(attn_bwd_dkv_kernel, dim3(ceil_div(Lk, 128), H, B), dim3(128), 0, stream, a);
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
