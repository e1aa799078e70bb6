// CSN stem: Conv3d(3,64,k=(3,7,7),s=(1,2,2),p=(1,3,3)) -> BN -> ReLU -> MaxPool3d((1,3,3),s=(1,2,2),p=(0,1,1))
// reference: models/backbones/ir_CSN_152.py:109-122,172-179.
// The conv itself is the implicit MFMA GEMM of stem_conv.hip; here: BN-apply + ReLU fused into the max-pool read, and the pool
// backward fused with the ReLU mask and the BN-backward partial statistics.
#include "common.h"

// out = max over the 3x3 window of relu(x*sc+sh); arg = window tap index (first maximum, like ATen)
__global__ __launch_bounds__(256) void stem_pool_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ sc,
                                                            const float* __restrict__ sh, bf16* __restrict__ out,
                                                            uint8_t* __restrict__ arg, int NT, int Hs, int Ws, int Hp, int Wp) {
    const int cg = threadIdx.x & 7;     // 8 channels each, C = 64
    float a[8], b[8];
    {
        const float4 s0 = *(const float4*)(sc + cg * 8), s1 = *(const float4*)(sc + cg * 8 + 4);
        const float4 h0 = *(const float4*)(sh + cg * 8), h1 = *(const float4*)(sh + cg * 8 + 4);
        a[0] = s0.x; a[1] = s0.y; a[2] = s0.z; a[3] = s0.w; a[4] = s1.x; a[5] = s1.y; a[6] = s1.z; a[7] = s1.w;
        b[0] = h0.x; b[1] = h0.y; b[2] = h0.z; b[3] = h0.w; b[4] = h1.x; b[5] = h1.y; b[6] = h1.z; b[7] = h1.w;
    }
    const long total = (long)NT * Hp * Wp;
    for (long p = (long)blockIdx.x * 32 + (threadIdx.x >> 3); p < total; p += (long)gridDim.x * 32) {
        const int wp = (int)(p % Wp); long r = p / Wp;
        const int hp = (int)(r % Hp); const long nt = r / Hp;
        float best[8]; int bi[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int hi = hp * 2 + dh - 1;
            if (hi < 0 || hi >= Hs) continue;
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int wi = wp * 2 + dw - 1;
                if (wi < 0 || wi >= Ws) continue;
                const bf16x8 v = as_bf16x8(*(const uint4*)(x + ((nt * Hs + hi) * Ws + wi) * 64 + cg * 8));
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float f = fmaxf(fmaf(bf2f(v[e]), a[e], b[e]), 0.f);
                    if (f > best[e]) { best[e] = f; bi[e] = dh * 3 + dw; }
                }
            }
        }
        bf16x8 o;
        uint32_t lo = 0, hi4 = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = f2bf(best[e]);
            if (e < 4) lo |= (uint32_t)bi[e] << (8 * e); else hi4 |= (uint32_t)bi[e] << (8 * (e - 4));
        }
        *(uint4*)(out + p * 64 + cg * 8) = as_uint4(o);
        if (arg) *(uint2*)(arg + p * 64 + cg * 8) = make_uint2(lo, hi4);
    }
}

// gradient back through pool + relu(bn): for every stem position gather the pooled gradients whose argmax points at it, mask by
// relu, write dz and the BN-backward partial statistics.
// One thread = one 2x2 block of stem positions (rows 2a, 2a+1; columns 2b, 2b+1) x 8 channels.  With the 3x3 / stride-2 / pad-1
// window those four positions are fed by exactly the four pooled cells (a..a+1, b..b+1), through nine fixed (position, tap) pairs:
// every pooled gradient / argmax word is loaded once and there is no per-tap divergence (the per-position form walked nine
// half-masked taps: 158 us for 423 MB).  8 lanes x 8 channels per block, 32 blocks per pass, consecutive blocks along w.
__global__ __launch_bounds__(256) void stem_pool_bwd_kernel(const bf16* __restrict__ gpool, const uint8_t* __restrict__ arg,
                                                            const bf16* __restrict__ x, const float* __restrict__ sc,
                                                            const float* __restrict__ sh, bf16* __restrict__ dz,
                                                            float* __restrict__ st0, float* __restrict__ st1, int NT, int Hs,
                                                            int Ws, int Hp, int Wp, int HB, int WB, int blocks_per_wg) {
    __shared__ float red[2][32][64];
    const int cg = threadIdx.x & 7, rs = threadIdx.x >> 3;
    float a[8], b[8], s0[8], s1[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a[e] = sc[cg * 8 + e]; b[e] = sh[cg * 8 + e]; s0[e] = 0.f; s1[e] = 0.f; }
    const int total = NT * HB * WB;
    const int q0 = blockIdx.x * blocks_per_wg, q1 = min(total, q0 + blocks_per_wg);
    for (int q = q0 + rs; q < q1; q += 32) {
        const int bw = q % WB; const int r = q / WB;
        const int bh = r % HB; const int nt = r / HB;
        const int h0 = 2 * bh, w0 = 2 * bw;
        const bool h1ok = h0 + 1 < Hs, w1ok = w0 + 1 < Ws;
        // the four pooled cells: c00 = (bh, bw), c01 = (bh, bw+1), c10 = (bh+1, bw), c11 = (bh+1, bw+1)
        const bool cok[4] = {true, bw + 1 < Wp, bh + 1 < Hp, bh + 1 < Hp && bw + 1 < Wp};
        bf16x8 g[4]; uint2 ai[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long o = ((long)(nt * Hp + bh + (k >> 1)) * Wp + bw + (k & 1)) * 64 + cg * 8;
            g[k] = cok[k] ? as_bf16x8(*(const uint4*)(gpool + o)) : as_bf16x8(make_uint4(0, 0, 0, 0));
            ai[k] = cok[k] ? *(const uint2*)(arg + o) : make_uint2(0xffffffffu, 0xffffffffu);       // tap 255: matches nothing
        }
        // the four positions p = 2*dy + dx of the block
        const bool pok[4] = {true, w1ok, h1ok, h1ok && w1ok};
        bf16x8 xv[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const long o = ((long)(nt * Hs + h0 + (p >> 1)) * Ws + w0 + (p & 1)) * 64 + cg * 8;
            xv[p] = pok[p] ? as_bf16x8(*(const uint4*)(x + o)) : as_bf16x8(make_uint4(0, 0, 0, 0));
        }
        float acc[4][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int t[4]; float gv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                t[k] = (int)(((e < 4 ? ai[k].x : ai[k].y) >> (8 * (e & 3))) & 0xff);
                gv[k] = bf2f(g[k][e]);
            }
            // (cell, tap) pairs in the accumulation order of the per-position loop (dh, dw ascending)
            acc[0][e] = t[0] == 4 ? gv[0] : 0.f;
            acc[1][e] = (t[1] == 3 ? gv[1] : 0.f) + (t[0] == 5 ? gv[0] : 0.f);
            acc[2][e] = (t[2] == 1 ? gv[2] : 0.f) + (t[0] == 7 ? gv[0] : 0.f);
            acc[3][e] = (((t[3] == 0 ? gv[3] : 0.f) + (t[2] == 2 ? gv[2] : 0.f)) + (t[1] == 6 ? gv[1] : 0.f)) + (t[0] == 8 ? gv[0] : 0.f);
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (!pok[p]) continue;
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xf = bf2f(xv[p][e]);
                const float v = fmaf(xf, a[e], b[e]) > 0.f ? acc[p][e] : 0.f;
                o[e] = f2bf(v);
                s0[e] += v; s1[e] += v * xf;
            }
            *(uint4*)(dz + ((long)(nt * Hs + h0 + (p >> 1)) * Ws + w0 + (p & 1)) * 64 + cg * 8) = as_uint4(o);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][rs][cg * 8 + e] = s0[e]; red[1][rs][cg * 8 + e] = s1[e]; }
    __syncthreads();
    if (threadIdx.x < 64) {
        float u = 0.f, v = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) { u += red[0][s][threadIdx.x]; v += red[1][s][threadIdx.x]; }
        st0[(long)blockIdx.x * 64 + threadIdx.x] = u;
        st1[(long)blockIdx.x * 64 + threadIdx.x] = v;
    }
}

extern "C" {

int tuber_stem_pool_fwd(const void* x, const float* sc, const float* sh, void* out, void* arg, int NT, int Hs, int Ws, int Hp, int Wp,
                        hipStream_t stream) {
    const long total = (long)NT * Hp * Wp;
    long nb = (total + 31) / 32;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(stem_pool_fwd_kernel, dim3((int)nb), dim3(256), 0, stream, (const bf16*)x, sc, sh, (bf16*)out, (uint8_t*)arg,
                       NT, Hs, Ws, Hp, Wp);
    TUBER_RETURN_LAUNCH();
}

int tuber_stem_pool_bwd_stat_rows(long positions) {
    constexpr int cap = 2048;
    long nb = (positions + 255) / 256;
    return (int)(nb > cap ? cap : nb);
}

int tuber_stem_pool_bwd(const void* gpool, const void* arg, const void* x, const float* sc, const float* sh, void* dz, float* st0,
                        float* st1, int NT, int Hs, int Ws, int Hp, int Wp, hipStream_t stream) {
    const long total = (long)NT * Hs * Ws;
    if (total <= 0 || total >= (1L << 31) / 64) return TUBER_EINVAL;
    if (Hp != (Hs - 1) / 2 + 1 || Wp != (Ws - 1) / 2 + 1) return TUBER_EINVAL;        // MaxPool (3x3, s 2, p 1) geometry
    const int nb = tuber_stem_pool_bwd_stat_rows(total);              // rows of partial statistics = workgroups
    const int HB = (Hs + 1) / 2, WB = (Ws + 1) / 2;                   // 2x2 position blocks
    const long blocks = (long)NT * HB * WB;
    const int bpw = (int)((blocks + nb - 1) / nb);
    hipLaunchKernelGGL(stem_pool_bwd_kernel, dim3(nb), dim3(256), 0, stream, (const bf16*)gpool, (const uint8_t*)arg, (const bf16*)x,
                       sc, sh, (bf16*)dz, st0, st1, NT, Hs, Ws, Hp, Wp, HB, WB, bpw);
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
