// Helpers + forward body of the MFMA flash attention (attention_mfma.hip), shared with the cooperative decoder kernel
// (decoder_coop.hip).  See attention_mfma.hip for the scheme.  Everything lives in an anonymous namespace of the including file.
#pragma once
#include "attention.h"

namespace {

constexpr int TL = 64;      // rows (keys or queries) per staged tile
constexpr int RP = 40;      // row-major image: 80-byte rows (ds_read_b128 of 16 different rows is conflict-free)

__device__ __forceinline__ long trow(const TokMap& m, int l, int b) {
    return (long)l * m.sL + (long)(b / m.B2) * m.s1 + (long)(b % m.B2) * m.s2;
}
__device__ __forceinline__ uint64_t eseed(const uint64_t* seed_ptr, uint64_t salt) {
    return (seed_ptr ? *seed_ptr : 0ull) * 0x9E3779B97F4A7C15ull + salt;
}
__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// one staged tile: thread t -> row r = t >> 2, 16-byte chunk c = t & 3 (8 head-dim elements)
__device__ __forceinline__ uint4 tile_fetch(const bf16* base, const TokMap& m, int b, int h, int r0, int L) {
    const int r = threadIdx.x >> 2, c = threadIdx.x & 3;
    return (r0 + r < L) ? *(const uint4*)(base + trow(m, r0 + r, b) * m.ld + h * 32 + c * 8) : make_uint4(0, 0, 0, 0);
}
__device__ __forceinline__ void park_rm(bf16 (*dst)[RP], uint4 v) {
    *(uint4*)&dst[threadIdx.x >> 2][(threadIdx.x & 3) * 8] = v;
}
// A operand from a row-major image: MFMA row i = image row (r0 + li), k = head dim g*8 .. g*8+7
__device__ __forceinline__ bf16x8 frag_rm(const bf16 (*src)[RP], int r0, int li, int g) {
    return as_bf16x8(*(const uint4*)&src[r0 + li][g * 8]);
}
// the same A operand straight from a ROW-MAJOR image via the gfx950 LDS transpose read: the 16-lane group reads the [4 rows][16 d]
// blocks at rows r0 + g*4 and r0 + 16 + g*4 (lane i supplies the address of chunk i = row i/4, columns (i%4)*4.. and receives
// column i), i.e. lane (li, g) gets rows {r0+g*4..+3, r0+16+g*4..+3} of column d0 + li -- no transposed copy in LDS
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 frag_tr_rm(const bf16 (*src)[RP], int d0, int r0, int li, int g) {
    const bf16* p = &src[r0 + g * 4 + (li >> 2)][d0 + (li & 3) * 4];
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 16 * RP));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
// SPLIT staging: one wave stages a whole 64-row tile (lane = row, four 16-byte chunks) into its private image
struct WaveTile { uint4 c[4]; };
__device__ __forceinline__ WaveTile wave_fetch(const bf16* base, const TokMap& m, int b, int h, int r0, int L) {
    const int r = r0 + (threadIdx.x & 63);
    WaveTile t;
    if (r < L) {
        const uint4* p = (const uint4*)(base + trow(m, r, b) * m.ld + h * 32);
#pragma unroll
        for (int c = 0; c < 4; ++c) t.c[c] = p[c];
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c) t.c[c] = make_uint4(0, 0, 0, 0);
    }
    return t;
}
__device__ __forceinline__ void wave_park(bf16 (*dst)[RP], const WaveTile& t) {
#pragma unroll
    for (int c = 0; c < 4; ++c) *(uint4*)&dst[threadIdx.x & 63][c * 8] = t.c[c];
}
__device__ __forceinline__ float group_max(float v) { return xor32_max(xor16_max(v)); }
__device__ __forceinline__ float group_sum(float v) { return xor32_sum(xor16_sum(v)); }

// ---------------------------------------------------------------------------------------------------------------------
// the forward body for the workgroup (bx, by = head, bz = batch) of a (ceil(Lq / (SPLIT ? 16 : 64)), H, B) grid: the kernel of
// attention_mfma.hip calls it with blockIdx; the cooperative decoder kernel (decoder_coop.hip) with the (clip, head) unit it owns
template <bool SPLIT>
__device__ __forceinline__ void attn_mfma_fwd_body(const AttnArgs& a, const int bx, const int by, const int bz) {
    constexpr int NI = SPLIT ? 4 : 1;                     // staged images (one per wave when the waves split the key tiles)
    __shared__ __attribute__((aligned(16))) bf16 ks[NI][TL][RP];
    __shared__ __attribute__((aligned(16))) bf16 vs[NI][TL][RP];
    __shared__ uint8_t msk[NI][TL];
    const int b = bz, h = by, q0 = bx * (SPLIT ? 16 : 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    const int im = SPLIT ? wave : 0;
    const int qi = q0 + (SPLIT ? 0 : wave * 16) + li;
    const bool qok = qi < a.Lq;
    const int qc = qok ? qi : a.Lq - 1;
    const bf16x8 qf = as_bf16x8(*(const uint4*)(a.Q + trow(a.mq, qc, b) * a.mq.ld + h * 32 + g * 8));
    const float inv_keep = dropout_inv_keep(a.pdrop);
    const uint64_t seed = a.pdrop > 0.f ? eseed(a.seed_ptr, a.salt) : 0ull;
    const uint64_t rbase = ((uint64_t)(b * a.H + h) * a.Lq + qc) * (uint64_t)a.Lk;
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    float mx = -INFINITY, l = 0.f;
    const int kfirst = SPLIT ? wave * TL : 0, kstep = SPLIT ? 4 * TL : TL;
    uint4 kr, vr;
    WaveTile kw, vw;
    if (SPLIT) { if (kfirst < a.Lk) { kw = wave_fetch(a.K, a.mk, b, h, kfirst, a.Lk); vw = wave_fetch(a.V, a.mv, b, h, kfirst, a.Lk); } }
    else { kr = tile_fetch(a.K, a.mk, b, h, 0, a.Lk); vr = tile_fetch(a.V, a.mv, b, h, 0, a.Lk); }
    for (int k0 = kfirst; k0 < a.Lk; k0 += kstep) {
        if (SPLIT) {
            // wave-private images: the LDS queue keeps one wave's reads and writes in order, no workgroup barrier
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
            f32x4 s[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) s[t] = mfma(frag_rm(ks[im], c0 + t * 16, li, g), qf, f32x4{0.f, 0.f, 0.f, 0.f});
            float p[8];
            float cmax = -INFINITY;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool dead = msk[im][c0 + t * 16 + g * 4 + r];
                    p[t * 4 + r] = dead ? -INFINITY : s[t][r] * a.scale;
                    cmax = fmaxf(cmax, p[t * 4 + r]);
                }
            cmax = group_max(cmax);
            const float mnew = fmaxf(mx, cmax);
            const float alpha = mnew == -INFINITY ? 1.f : __expf(mx - mnew);
            float ls = 0.f;
            bf16x8 pf;
            bool keep[2][4] = {{true, true, true, true}, {true, true, true, true}};     // two runs of four consecutive keys per lane
            if (a.pdrop > 0.f) {
                dropout_keep_run<4>(seed, rbase + k0 + c0 + g * 4, a.thresh, keep[0]);
                dropout_keep_run<4>(seed, rbase + k0 + c0 + 16 + g * 4, a.thresh, keep[1]);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float pe = mnew == -INFINITY ? 0.f : __expf(p[e] - mnew);
                ls += pe;
                if (a.pdrop > 0.f) pe = keep[e >> 2][e & 3] ? pe * inv_keep : 0.f;
                pf[e] = f2bf(pe);
            }
            l = l * alpha + group_sum(ls);
            mx = mnew;
#pragma unroll
            for (int r = 0; r < 4; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
            o0 = mfma(frag_tr_rm(vs[im], 0, c0, li, g), pf, o0);
            o1 = mfma(frag_tr_rm(vs[im], 16, c0, li, g), pf, o1);
        }
        if (SPLIT) __builtin_amdgcn_wave_barrier();
    }
    bool writer = true;
    if (SPLIT) {
        // merge the four waves' online-softmax states of the same 16 queries: m = max m_w, l = sum l_w e^(m_w - m), O likewise
        __syncthreads();
        float* cmb = (float*)&ks[0][0][0];                 // [4 waves][64 lanes][10]: 10 KB of the 20 KB the images occupy
        float* mine = cmb + (wave * 64 + lane) * 10;
        mine[0] = mx; mine[1] = l;
#pragma unroll
        for (int r = 0; r < 4; ++r) { mine[2 + r] = o0[r]; mine[6 + r] = o1[r]; }
        __syncthreads();
        writer = wave == 0;                                // (a device function: the other waves fall through instead of returning)
        if (writer) {
            float m = -INFINITY;
#pragma unroll
            for (int w = 0; w < 4; ++w) m = fmaxf(m, cmb[(w * 64 + lane) * 10]);
            l = 0.f;
            o0 = f32x4{0.f, 0.f, 0.f, 0.f}; o1 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const float* o = cmb + (w * 64 + lane) * 10;
                const float f = o[0] == -INFINITY ? 0.f : __expf(o[0] - m);
                l += o[1] * f;
#pragma unroll
                for (int r = 0; r < 4; ++r) { o0[r] += o[2 + r] * f; o1[r] += o[6 + r] * f; }
            }
            mx = m;
        }
    }
    if (writer && qok) {
        const float inv = l > 0.f ? 1.f / l : 0.f;
        bf16* orow = a.O + trow(a.mo, qi, b) * a.mo.ld + h * 32;
        bf16x4 y0, y1;
#pragma unroll
        for (int r = 0; r < 4; ++r) { y0[r] = f2bf(o0[r] * inv); y1[r] = f2bf(o1[r] * inv); }
        *(uint2*)(orow + g * 4) = as_uint2(y0);
        *(uint2*)(orow + 16 + g * 4) = as_uint2(y1);
        if (a.lse && g == 0) a.lse[((long)b * a.H + h) * a.Lq + qi] = mx + __logf(l);
    }
}


}  // namespace
