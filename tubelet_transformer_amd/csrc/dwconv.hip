// Depthwise 3x3x3 Conv3d (groups = C, padding 1, stride (st, ss, ss)) on NDHWC bf16 -- gfx950.
// reference: ResNeXtBottleneck.conv3, models/backbones/ir_CSN_152.py:48-51 (the stride of every
// stage sits on this conv).  HBM-bound (11.7 FLOP/B): channels are the contiguous dimension, a
// 16-lane group covers one 128-byte cache line of 64 channels per position, each thread owns 4
// channels x 4 consecutive output columns so every loaded input vector is reused across taps;
// the 27 x 64 fp32 weight slice of the block's channel chunk lives in LDS.
// Fused: BatchNorm-apply + ReLU of the producer's raw output on load (zero padding applied AFTER
// the activation, like the reference's padded conv over relu(bn1(.))), per-channel partial
// statistics for the following training-mode BatchNorm on store.
#include "common.h"

struct DwGeom { int N, Ti, Hi, Wi, To, Ho, Wo, C, st, ss; };

__device__ __forceinline__ void load_w_chunk(float (*wl)[64], const float* __restrict__ w, int c0, int C) {
    for (int i = threadIdx.x; i < 27 * 64; i += 256) {
        const int tap = i >> 6, c = i & 63;
        wl[tap][c] = (c0 + c < C) ? w[(long)(c0 + c) * 27 + tap] : 0.f;
    }
}

template <int SS>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(
    const bf16* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh, const float* __restrict__ w,
    bf16* __restrict__ out, float* __restrict__ st0, float* __restrict__ st1, DwGeom g, int iters) {
    constexpr int NIN = 3 * SS + 3;   // input columns feeding 4 consecutive outputs
    __shared__ __attribute__((aligned(16))) float wl[27][64];
    __shared__ float red[2][16][64];
    const int cl = threadIdx.x & 15, ps = threadIdx.x >> 4;
    const int c0 = blockIdx.y * 64, c = c0 + cl * 4;
    load_w_chunk(wl, w, c0, g.C);
    float a4[4] = {1.f, 1.f, 1.f, 1.f}, b4[4] = {0.f, 0.f, 0.f, 0.f};
    if (sc) {
        const float4 s = *(const float4*)(sc + c), h = *(const float4*)(sh + c);
        a4[0] = s.x; a4[1] = s.y; a4[2] = s.z; a4[3] = s.w; b4[0] = h.x; b4[1] = h.y; b4[2] = h.z; b4[3] = h.w;
    }
    __syncthreads();
    const int Wg = (g.Wo + 3) >> 2;
    const long segs = (long)g.N * g.To * g.Ho * Wg;
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        const long seg = ((long)blockIdx.x * iters + it) * 16 + ps;
        if (seg >= segs) break;
        const int sg32 = (int)seg;
        int wg = sg32 % Wg; int r = sg32 / Wg;
        const int ho = r % g.Ho; r /= g.Ho;
        const int to = r % g.To; const int n = r / g.To;
        const int wo0 = wg * 4;
        float acc[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
        // all 9 x NIN input vectors are loaded unconditionally from clamped addresses (one latency, not nine) and
        // zeroed afterwards where the tap falls into the padding
        uint2 raw[9][NIN];
        unsigned okr = 0;                           // bit r: row (dt,dh) inside the volume
        int wok = 0;                                // bit i: column i inside the row
#pragma unroll
        for (int i = 0; i < NIN; ++i) { const int wi = wo0 * SS - 1 + i; wok |= (wi >= 0 && wi < g.Wi) ? (1 << i) : 0; }
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int dt = r / 3, dh = r % 3;
            const int ti = to * g.st + dt - 1, hi = ho * SS + dh - 1;
            const bool ok = ti >= 0 && ti < g.Ti && hi >= 0 && hi < g.Hi;
            okr |= ok ? (1u << r) : 0u;
            const int tc = min(max(ti, 0), g.Ti - 1), hc = min(max(hi, 0), g.Hi - 1);
            const bf16* row = x + (((long)n * g.Ti + tc) * g.Hi + hc) * (long)g.Wi * g.C + c;
#pragma unroll
            for (int i = 0; i < NIN; ++i) {
                const int wc = min(max(wo0 * SS - 1 + i, 0), g.Wi - 1);
                raw[r][i] = *(const uint2*)(row + (long)wc * g.C);
            }
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            if (!((okr >> r) & 1)) continue;
            float in[NIN][4];
#pragma unroll
            for (int i = 0; i < NIN; ++i) {
                const bf16x4 v = as_bf16x4(raw[r][i]);
                const bool ok = (wok >> i) & 1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float f = fmaf(bf2f(v[e]), a4[e], b4[e]);
                    in[i][e] = ok ? (sc ? fmaxf(f, 0.f) : f) : 0.f;
                }
            }
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const float4 wv = *(const float4*)&wl[r * 3 + dw][cl * 4];
                const float ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(in[j * SS + dw][e], ww[e], acc[j][e]);
            }
        }
        bf16* orow = out + ((((long)n * g.To + to) * g.Ho + ho) * (long)g.Wo + wo0) * g.C + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (wo0 + j < g.Wo) {
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) { o[e] = f2bf(acc[j][e]); s0[e] += acc[j][e]; s1[e] += acc[j][e] * acc[j][e]; }
                *(uint2*)(orow + (long)j * g.C) = as_uint2(o);
            }
        }
    }
    if (st0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[0][ps][cl * 4 + e] = s0[e]; red[1][ps][cl * 4 + e] = s1[e]; }
        __syncthreads();
        if (threadIdx.x < 64 && c0 + threadIdx.x < g.C) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) { a += red[0][s][threadIdx.x]; b += red[1][s][threadIdx.x]; }
            st0[(long)blockIdx.x * g.C + c0 + threadIdx.x] = a;
            st1[(long)blockIdx.x * g.C + c0 + threadIdx.x] = b;
        }
    }
}

// data gradient: da[i] = sum_taps w[tap] * g[o(i,tap)], then (fused) through relu(bn1(x)):
// dz = da * [x*sc+sh > 0]; partial stats sum dz, sum dz*x for bn1's backward.
template <int SS>
__global__ __launch_bounds__(256) void dwconv_bwd_data_kernel(
    const bf16* __restrict__ gout, const float* __restrict__ w, const bf16* __restrict__ x, const float* __restrict__ sc,
    const float* __restrict__ sh, bf16* __restrict__ dz, float* __restrict__ st0, float* __restrict__ st1, DwGeom g, int iters) {
    constexpr int NG = SS == 1 ? 6 : 3;
    __shared__ __attribute__((aligned(16))) float wl[27][64];
    __shared__ float red[2][16][64];
    const int cl = threadIdx.x & 15, ps = threadIdx.x >> 4;
    const int c0 = blockIdx.y * 64, c = c0 + cl * 4;
    load_w_chunk(wl, w, c0, g.C);
    float a4[4], b4[4];
    {
        const float4 s = *(const float4*)(sc + c), h = *(const float4*)(sh + c);
        a4[0] = s.x; a4[1] = s.y; a4[2] = s.z; a4[3] = s.w; b4[0] = h.x; b4[1] = h.y; b4[2] = h.z; b4[3] = h.w;
    }
    __syncthreads();
    const int Wg = (g.Wi + 3) >> 2;
    const long segs = (long)g.N * g.Ti * g.Hi * Wg;
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
        const long seg = ((long)blockIdx.x * iters + it) * 16 + ps;
        if (seg >= segs) break;
        const int sg32 = (int)seg;
        int wg = sg32 % Wg; int r = sg32 / Wg;
        const int hi = r % g.Hi; r /= g.Hi;
        const int ti = r % g.Ti; const int n = r / g.Ti;
        const int wi0 = wg * 4;
        float acc[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[j][e] = 0.f;
        // all 9 x NG gradient vectors are loaded unconditionally from clamped addresses (one latency) and masked afterwards
        uint2 raw[9][NG];
        unsigned okr = 0;
        int wok = 0;
#pragma unroll
        for (int i = 0; i < NG; ++i) { const int wo = SS == 1 ? wi0 - 1 + i : (wi0 >> 1) + i; wok |= (wo >= 0 && wo < g.Wo) ? (1 << i) : 0; }
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            const int dt = r / 3, dh = r % 3;
            const int tn = ti + 1 - dt, hn = hi + 1 - dh;
            const int to = tn / g.st, ho = hn / SS;
            const bool ok = tn >= 0 && (tn % g.st) == 0 && to < g.To && hn >= 0 && (hn % SS) == 0 && ho < g.Ho;
            okr |= ok ? (1u << r) : 0u;
            const int tc = min(max(to, 0), g.To - 1), hc = min(max(ho, 0), g.Ho - 1);
            const bf16* row = gout + (((long)n * g.To + tc) * g.Ho + hc) * (long)g.Wo * g.C + c;
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const int wo = SS == 1 ? wi0 - 1 + i : (wi0 >> 1) + i;
                raw[r][i] = *(const uint2*)(row + (long)min(max(wo, 0), g.Wo - 1) * g.C);
            }
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            if (!((okr >> r) & 1)) continue;
            float gw[NG][4];
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const bf16x4 v = as_bf16x4(raw[r][i]);
                const bool ok = (wok >> i) & 1;
#pragma unroll
                for (int e = 0; e < 4; ++e) gw[i][e] = ok ? bf2f(v[e]) : 0.f;
            }
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const float4 wv = *(const float4*)&wl[r * 3 + dw][cl * 4];
                const float ww[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int num = j + 1 - dw;           // wo*SS = wi0 + num
                    if (SS == 1) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(gw[num + 1][e], ww[e], acc[j][e]);
                    } else if (num >= 0 && (num & 1) == 0) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[j][e] = fmaf(gw[num >> 1][e], ww[e], acc[j][e]);
                    }
                }
            }
        }
        const long pos = (((long)n * g.Ti + ti) * g.Hi + hi) * (long)g.Wi + wi0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (wi0 + j < g.Wi) {
                const bf16x4 xv = as_bf16x4(*(const uint2*)(x + (pos + j) * g.C + c));
                bf16x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xf = bf2f(xv[e]);
                    const float v = fmaf(xf, a4[e], b4[e]) > 0.f ? acc[j][e] : 0.f;
                    o[e] = f2bf(v);
                    s0[e] += v; s1[e] += v * xf;
                }
                *(uint2*)(dz + (pos + j) * g.C + c) = as_uint2(o);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][ps][cl * 4 + e] = s0[e]; red[1][ps][cl * 4 + e] = s1[e]; }
    __syncthreads();
    if (threadIdx.x < 64 && c0 + threadIdx.x < g.C) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) { a += red[0][s][threadIdx.x]; b += red[1][s][threadIdx.x]; }
        st0[(long)blockIdx.x * g.C + c0 + threadIdx.x] = a;
        st1[(long)blockIdx.x * g.C + c0 + threadIdx.x] = b;
    }
}

// weight gradient partials: P[block][tap][C] = sum over the block's output positions of
// g[o][c] * relu(bn1(x))[in(o,tap)][c].  Same register tiling as the forward kernel: a thread owns
// 4 channels x 4 consecutive output columns, so every activated input vector it loads feeds up to
// 3 taps x 4 outputs; 27 x 4 fp32 accumulators live in registers across the block's segments and
// are reduced over the 16 position slots (wave shuffles + LDS) once at the end.
template <int SS>
__global__ __launch_bounds__(256) void dwconv_bwd_weight_kernel(
    const bf16* __restrict__ gout, const bf16* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
    float* __restrict__ P, DwGeom g, int iters) {
    constexpr int NIN = 3 * SS + 3;
    __shared__ float red[4][27][64];
    const int cl = threadIdx.x & 15, ps = threadIdx.x >> 4;
    const int c0 = blockIdx.y * 64, c = c0 + cl * 4;
    float a4[4], b4[4];
    {
        const float4 s = *(const float4*)(sc + c), h = *(const float4*)(sh + c);
        a4[0] = s.x; a4[1] = s.y; a4[2] = s.z; a4[3] = s.w; b4[0] = h.x; b4[1] = h.y; b4[2] = h.z; b4[3] = h.w;
    }
    float acc[27][4];
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[t][e] = 0.f;
    const int Wg = (g.Wo + 3) >> 2;
    const long segs = (long)g.N * g.To * g.Ho * Wg;
    for (int it = 0; it < iters; ++it) {
        const long seg = ((long)blockIdx.x * iters + it) * 16 + ps;
        if (seg >= segs) break;
        const int sg32 = (int)seg;
        int wg = sg32 % Wg; int r = sg32 / Wg;
        const int ho = r % g.Ho; r /= g.Ho;
        const int to = r % g.To; const int n = r / g.To;
        const int wo0 = wg * 4;
        float gv[4][4];
        const bf16* grow = gout + ((((long)n * g.To + to) * g.Ho + ho) * (long)g.Wo + wo0) * g.C + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (wo0 + j < g.Wo) {
                const bf16x4 v = as_bf16x4(*(const uint2*)(grow + (long)j * g.C));
#pragma unroll
                for (int e = 0; e < 4; ++e) gv[j][e] = bf2f(v[e]);
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) gv[j][e] = 0.f;
            }
        }
        int wok = 0;
#pragma unroll
        for (int i = 0; i < NIN; ++i) { const int wi = wo0 * SS - 1 + i; wok |= (wi >= 0 && wi < g.Wi) ? (1 << i) : 0; }
#pragma unroll
        for (int dt = 0; dt < 3; ++dt) {
            const int ti = to * g.st + dt - 1;
            if (ti < 0 || ti >= g.Ti) continue;           // wave-uniform for most waves (a wave spans <= 4 segments)
            // the 3 x NIN activation vectors of this temporal slab: unconditional clamped loads, masked afterwards
            uint2 raw[3][NIN];
            int okh = 0;
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) {
                const int hi = ho * SS + dh - 1;
                okh |= (hi >= 0 && hi < g.Hi) ? (1 << dh) : 0;
                const bf16* row = x + (((long)n * g.Ti + ti) * g.Hi + min(max(hi, 0), g.Hi - 1)) * (long)g.Wi * g.C + c;
#pragma unroll
                for (int i = 0; i < NIN; ++i)
                    raw[dh][i] = *(const uint2*)(row + (long)min(max(wo0 * SS - 1 + i, 0), g.Wi - 1) * g.C);
            }
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) {
                if (!((okh >> dh) & 1)) continue;
                float in[NIN][4];
#pragma unroll
                for (int i = 0; i < NIN; ++i) {
                    const bf16x4 v = as_bf16x4(raw[dh][i]);
                    const bool ok = (wok >> i) & 1;
#pragma unroll
                    for (int e = 0; e < 4; ++e) in[i][e] = ok ? fmaxf(fmaf(bf2f(v[e]), a4[e], b4[e]), 0.f) : 0.f;
                }
#pragma unroll
                for (int dw = 0; dw < 3; ++dw) {
                    const int tap = (dt * 3 + dh) * 3 + dw;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[tap][e] = fmaf(gv[j][e], in[j * SS + dw][e], acc[tap][e]);
                }
            }
        }
    }
    // reduce over the 16 position slots: lanes (ps & 3) within a wave by shuffles (lane = (ps&3)*16 + cl), waves via LDS
    const int wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 27; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[t][e];
            v = xor32_sum(xor16_sum(v));
            if ((threadIdx.x & 63) < 16) red[wave][t][cl * 4 + e] = v;
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 27 * 64; i += 256) {
        const int tap = i >> 6, cc = i & 63;
        if (c0 + cc < g.C)
            P[((long)blockIdx.x * 27 + tap) * g.C + c0 + cc] = red[0][tap][cc] + red[1][tap][cc] + red[2][tap][cc] + red[3][tap][cc];
    }
}

// out[c][tap] (+)= sum_r P[r][tap][c]: 32 row-groups x 32 (tap,c) columns per block
__global__ __launch_bounds__(1024) void dw_wgrad_reduce_kernel(const float* __restrict__ P, float* __restrict__ out, int R, int C, int accumulate) {
    __shared__ float red[32][33];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cl;            // over tap*C + c
    const int n = 27 * C;
    float a = 0.f;
    if (i < n) for (int r = rg; r < R; r += 32) a += P[(long)r * n + i];
    red[rg][cl] = a;
    __syncthreads();
    if (rg == 0 && i < n) {
        a = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) a += red[k][cl];
        float* o = out + (long)(i % C) * 27 + i / C;
        *o = accumulate ? *o + a : a;
    }
}

static int dw_blocks(long segs, int* iters, int max_blocks = 4096) {
    long groups = (segs + 15) / 16;
    int it = (int)((groups + max_blocks - 1) / max_blocks);   // <= 4096 blocks: one or two segments per thread, latency hidden by occupancy
    if (it < 1) it = 1;
    *iters = it;
    return (int)((groups + it - 1) / it);
}
// the weight-gradient kernel ends with a 108-accumulator cross-thread reduction and a 6.9 KB partial per workgroup: few, long-lived
// workgroups (2 per CU) amortise that tail and keep the partial buffer (and its reduce pass) small
static const int DW_WGRAD_BLOCKS = 512;

extern "C" {

static bool dw_ok(int C, int st, int ss) { return (C & 63) == 0 && (st == 1 || st == 2) && (ss == 1 || ss == 2); }

// number of partial-stat rows the forward kernel writes (= grid.x)
int tuber_dwconv_fwd_stat_rows(int N, int To, int Ho, int Wo) {
    int it;
    return dw_blocks((long)N * To * Ho * ((Wo + 3) / 4), &it);
}
int tuber_dwconv_bwd_data_stat_rows(int N, int Ti, int Hi, int Wi) {
    int it;
    return dw_blocks((long)N * Ti * Hi * ((Wi + 3) / 4), &it);
}
int tuber_dwconv_bwd_weight_blocks(int N, int To, int Ho, int Wo) {
    int it;
    return dw_blocks((long)N * To * Ho * ((Wo + 3) / 4), &it, DW_WGRAD_BLOCKS);
}

int tuber_dwconv_fwd(const void* x, const float* sc, const float* sh, const float* w, void* out, float* st0, float* st1,
                     int N, int Ti, int Hi, int Wi, int To, int Ho, int Wo, int C, int st, int ss, hipStream_t stream) {
    if (!dw_ok(C, st, ss)) return TUBER_EINVAL;
    DwGeom g{N, Ti, Hi, Wi, To, Ho, Wo, C, st, ss};
    int iters;
    const int nb = dw_blocks((long)N * To * Ho * ((Wo + 3) / 4), &iters);
    dim3 grid(nb, C / 64), block(256);
    if (ss == 1) hipLaunchKernelGGL(dwconv_fwd_kernel<1>, grid, block, 0, stream, (const bf16*)x, sc, sh, w, (bf16*)out, st0, st1, g, iters);
    else hipLaunchKernelGGL(dwconv_fwd_kernel<2>, grid, block, 0, stream, (const bf16*)x, sc, sh, w, (bf16*)out, st0, st1, g, iters);
    TUBER_RETURN_LAUNCH();
}

int tuber_dwconv_bwd_data(const void* gout, const float* w, const void* x, const float* sc, const float* sh, void* dz,
                          float* st0, float* st1, int N, int Ti, int Hi, int Wi, int To, int Ho, int Wo, int C, int st, int ss,
                          hipStream_t stream) {
    if (!dw_ok(C, st, ss)) return TUBER_EINVAL;
    DwGeom g{N, Ti, Hi, Wi, To, Ho, Wo, C, st, ss};
    int iters;
    const int nb = dw_blocks((long)N * Ti * Hi * ((Wi + 3) / 4), &iters);
    dim3 grid(nb, C / 64), block(256);
    if (ss == 1) hipLaunchKernelGGL(dwconv_bwd_data_kernel<1>, grid, block, 0, stream, (const bf16*)gout, w, (const bf16*)x, sc, sh, (bf16*)dz, st0, st1, g, iters);
    else hipLaunchKernelGGL(dwconv_bwd_data_kernel<2>, grid, block, 0, stream, (const bf16*)gout, w, (const bf16*)x, sc, sh, (bf16*)dz, st0, st1, g, iters);
    TUBER_RETURN_LAUNCH();
}

// dw[c][tap] (+)= sum_r partial[r][tap][c]
int tuber_dw_wgrad_reduce(const float* partial, float* dw, int R, int C, int accumulate, hipStream_t stream) {
    hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3(ceil_div(27 * C, 32)), dim3(1024), 0, stream, partial, dw, R, C, accumulate);
    TUBER_RETURN_LAUNCH();
}

// partial must hold blocks * 27 * C floats; dw is the [C][27] fp32 weight gradient
int tuber_dwconv_bwd_weight(const void* gout, const void* x, const float* sc, const float* sh, float* partial, float* dw,
                            int accumulate, int N, int Ti, int Hi, int Wi, int To, int Ho, int Wo, int C, int st, int ss,
                            hipStream_t stream) {
    if (!dw_ok(C, st, ss)) return TUBER_EINVAL;
    DwGeom g{N, Ti, Hi, Wi, To, Ho, Wo, C, st, ss};
    int iters;
    const int nb = dw_blocks((long)N * To * Ho * ((Wo + 3) / 4), &iters, DW_WGRAD_BLOCKS);
    dim3 grid(nb, C / 64), block(256);
    if (ss == 1) hipLaunchKernelGGL(dwconv_bwd_weight_kernel<1>, grid, block, 0, stream, (const bf16*)gout, (const bf16*)x, sc, sh, partial, g, iters);
    else hipLaunchKernelGGL(dwconv_bwd_weight_kernel<2>, grid, block, 0, stream, (const bf16*)gout, (const bf16*)x, sc, sh, partial, g, iters);
    if (accumulate != 2)                       // accumulate == 2: partial blocks reduced later by tuber_multi_reduce
        hipLaunchKernelGGL(dw_wgrad_reduce_kernel, dim3(ceil_div(27 * C, 32)), dim3(1024), 0, stream, partial, dw, nb, C, accumulate);
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
