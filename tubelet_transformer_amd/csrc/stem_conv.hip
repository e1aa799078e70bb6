// CSN stem Conv3d(3,64,k=(3,7,7),s=(1,2,2),p=(1,3,3)) as an IMPLICIT GEMM on MFMA (gfx950) -- forward and weight gradient.
// reference: models/backbones/ir_CSN_152.py:109-115,172-174.  No patch matrix ever reaches HBM (the explicit 1.25 GB
// im2col of the first round cost ~7 GB/step of traffic, profiles/r01_d_pmc_hbm_traffic_per_kernel.txt).
//
// Work unit ("tile"): 64 consecutive output columns wo of one (clip, t, ho) output row, all 64 channels.  Its receptive
// field is 63 "runs" (c, kt, kh) of 2*64+5 = 133 consecutive input pixels; the fp32 NCDHW clip is read run-wise
// (coalesced along w), converted to bf16 and staged in LDS as P[run][x] (17 KB, double buffered).
// K is laid out as k' = run*8 + kw (kw = 7 and run = 63 are zero padding -> K' = 512), so that an MFMA operand fragment
// (8 consecutive k') of output column w is the 8 consecutive pixels P[run][2w .. 2w+7]: four 4-byte LDS reads.
//   forward : wave = 16 output channels; its 16 weight fragments (all of K') stay in 64 VGPRs for the whole persistent
//             kernel; per k-step 4 patch fragments (one per 16-column m-tile) + 4 MFMAs.  BN partial statistics fused.
//   dW      : D[k'][n] += sum_m P^T[k'][m] G[m][n]; wave = 8 k'-tiles x 4 n-tiles (128 accumulator VGPRs, kept across
//             the persistent loop); G tile transposed into LDS with 4x4 register transposes; patch^T fragments are
//             8 strided 2-byte LDS reads.  fp32 partials per workgroup, reduced + scattered to [64][441] afterwards.
#include "common.h"

#define SW 64
#define PXW 160         // row pitch 80 dwords = 16 mod 32: the four run rows of a fragment read (ds_read_b32: 32 banks, 32-lane groups) fall on disjoint banks (136 pixels used)
#define KP 512

struct StemGeom { int B, T, H, W, Ho, Wo, tilesW, ntiles; };

// pairs (run r, x = 2*xp, 2*xp+1): 64 runs x 68 pairs = 4352 = 17 per thread
#define NPAIR 17
__device__ __forceinline__ void stem_load_patch(const float* __restrict__ clip, const StemGeom& g, int tile, float2 (&v)[NPAIR]) {
    const int wt = tile % g.tilesW; int r_ = tile / g.tilesW;
    const int ho = r_ % g.Ho; r_ /= g.Ho;
    const int t = r_ % g.T; const int b = r_ / g.T;
    const int wbase = 2 * wt * SW - 3;
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int r = idx / 68, xp = idx % 68;
        float2 o = make_float2(0.f, 0.f);
        if (r < 63) {
            const int c = r / 21, kt = (r / 7) % 3, kh = r % 7;
            const int ti = t + kt - 1, hi = 2 * ho + kh - 3;
            if (ti >= 0 && ti < g.T && hi >= 0 && hi < g.H) {
                const float* row = clip + ((((long)b * 3 + c) * g.T + ti) * g.H + hi) * (long)g.W;
                const int w0 = wbase + 2 * xp;
                if (w0 >= 0 && w0 < g.W) o.x = row[w0];
                if (w0 + 1 >= 0 && w0 + 1 < g.W) o.y = row[w0 + 1];
            }
        }
        v[i] = o;
    }
}
__device__ __forceinline__ void stem_store_patch(bf16 (*P)[PXW], const float2 (&v)[NPAIR]) {
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int r = idx / 68, xp = idx % 68;
        bf16x2 p = {f2bf(v[i].x), f2bf(v[i].y)};
        *(bf16x2*)&P[r][2 * xp] = p;
    }
}
__device__ __forceinline__ bf16x8 stem_patch_frag(const bf16 (*P)[PXW], int run, int w) {   // P[run][2w .. 2w+7]
    const uint32_t* p = (const uint32_t*)&P[run][2 * w];
    uint4 u = make_uint4(p[0], p[1], p[2], p[3]);
    return as_bf16x8(u);
}

__global__ __launch_bounds__(256) void stem_conv_fwd_kernel(const float* __restrict__ clip, const bf16* __restrict__ Wp,
                                                            bf16* __restrict__ out, float* __restrict__ st0,
                                                            float* __restrict__ st1, StemGeom g) {
    __shared__ __attribute__((aligned(16))) bf16 P[2][64][PXW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, gq = lane >> 4;
    // this wave's 16 channels: weight fragments for all 16 k-steps (A operand: row i = channel, 8 consecutive k')
    bf16x8 wf[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) wf[ks] = as_bf16x8(*(const uint4*)(Wp + (long)(wave * 16 + li) * KP + ks * 32 + gq * 8));
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    float2 pre[NPAIR];
    int tile = blockIdx.x, buf = 0;
    if (tile < g.ntiles) {
        stem_load_patch(clip, g, tile, pre);
        stem_store_patch(P[0], pre);
    }
    __syncthreads();
    for (; tile < g.ntiles; tile += gridDim.x) {
        const int next = tile + gridDim.x;
        if (next < g.ntiles) stem_load_patch(clip, g, next, pre);       // in flight during the MFMAs below
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int run = ks * 4 + gq;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], stem_patch_frag(P[buf], run, mt * 16 + li), acc[mt], 0, 0, 0);
        }
        // D[i = channel][j = column]: lane holds column mt*16+li, channels wave*16 + gq*4 + 0..3
        const int wt = tile % g.tilesW;
        const long row0 = (long)(tile / g.tilesW) * g.Wo + wt * SW;        // (b,t,ho) row start + column offset
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int w = wt * SW + mt * 16 + li;
            if (w < g.Wo) {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) { o[r] = f2bf(acc[mt][r]); s0[r] += acc[mt][r]; s1[r] += acc[mt][r] * acc[mt][r]; }
                *(uint2*)(out + (row0 + mt * 16 + li) * 64 + wave * 16 + gq * 4) = as_uint2(o);
            }
        }
        if (next < g.ntiles) stem_store_patch(P[buf ^ 1], pre);
        __syncthreads();
        buf ^= 1;
    }
    if (st0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = quad16_sum(s0[r]), b2 = quad16_sum(s1[r]);
            if (li == 0) {
                st0[(long)blockIdx.x * 64 + wave * 16 + gq * 4 + r] = a;
                st1[(long)blockIdx.x * 64 + wave * 16 + gq * 4 + r] = b2;
            }
        }
    }
}

// ---- weight gradient ----
// D[k'][n] += sum_m patch^T[k'][m] G[m][n].  The MFMA A fragment needs 8 consecutive m of one k' = (run, kw), i.e. the pixels
// P[run][2m + kw]: stride 2.  The patch is therefore staged DE-INTERLEAVED (PE = even, PO = odd pixel columns), which makes
// the fragment 8 consecutive bf16 of PE/PO[run] starting at m + kw/2; and the 16 rows of an MFMA are 16 RUNS with one common
// kw, so the sub-dword start offset is wave-uniform: one 16-byte + one 8-byte LDS read and (for odd offsets) 4 v_alignbit.
#define EOW 72
#define NQUAD 9          // (run, 4 consecutive pixels): 64 runs x 34 quads = 2176 = 8.5 per thread
__device__ __forceinline__ void stem_load_patch4(const float* __restrict__ clip, const StemGeom& g, int tile, float4 (&v)[NQUAD]) {
    const int wt = tile % g.tilesW; int r_ = tile / g.tilesW;
    const int ho = r_ % g.Ho; r_ /= g.Ho;
    const int t = r_ % g.T; const int b = r_ / g.T;
    const int wbase = 2 * wt * SW - 3;
#pragma unroll
    for (int i = 0; i < NQUAD; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int r = idx / 34, q = idx % 34;
        float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < 63) {
            const int c = r / 21, kt = (r / 7) % 3, kh = r % 7;
            const int ti = t + kt - 1, hi = 2 * ho + kh - 3;
            if (ti >= 0 && ti < g.T && hi >= 0 && hi < g.H) {
                const float* row = clip + ((((long)b * 3 + c) * g.T + ti) * g.H + hi) * (long)g.W;
                const int w0 = wbase + 4 * q;
                if (w0 >= 0 && w0 < g.W) o.x = row[w0];
                if (w0 + 1 >= 0 && w0 + 1 < g.W) o.y = row[w0 + 1];
                if (w0 + 2 >= 0 && w0 + 2 < g.W) o.z = row[w0 + 2];
                if (w0 + 3 >= 0 && w0 + 3 < g.W) o.w = row[w0 + 3];
            }
        }
        v[i] = o;
    }
}
__device__ __forceinline__ void stem_store_patch_eo(bf16 (*PE)[EOW], bf16 (*PO)[EOW], const float4 (&v)[NQUAD]) {
#pragma unroll
    for (int i = 0; i < NQUAD; ++i) {
        const int idx = threadIdx.x + 256 * i;
        const int r = idx / 34, q = idx % 34;
        if (r < 64) {
            *(bf16x2*)&PE[r][2 * q] = bf16x2{f2bf(v[i].x), f2bf(v[i].z)};     // pixels 4q, 4q+2   -> even columns 2q, 2q+1
            *(bf16x2*)&PO[r][2 * q] = bf16x2{f2bf(v[i].y), f2bf(v[i].w)};     // pixels 4q+1, 4q+3 -> odd columns
        }
    }
}
// 8 consecutive bf16 of row[] starting at element m0 + s (m0 % 8 == 0, s = 0..3 wave-uniform)
__device__ __forceinline__ bf16x8 stem_frag_shift(const bf16* row, int m0, int s) {
    const uint4 a = *(const uint4*)(row + m0);
    const uint2 b = *(const uint2*)(row + m0 + 8);
    uint32_t w[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
    uint4 o;
    if (s & 2) { w[0] = w[1]; w[1] = w[2]; w[2] = w[3]; w[3] = w[4]; w[4] = w[5]; }
    if (s & 1) {
        o.x = __builtin_amdgcn_alignbit(w[1], w[0], 16); o.y = __builtin_amdgcn_alignbit(w[2], w[1], 16);
        o.z = __builtin_amdgcn_alignbit(w[3], w[2], 16); o.w = __builtin_amdgcn_alignbit(w[4], w[3], 16);
    } else {
        o = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return as_bf16x8(o);
}

// dW'[k'][n] partials: one [512][64] fp32 slab per workgroup.  MFMA tile id = wave*7 + kt (28 tiles: kw = id >> 2 in 0..6,
// run block rb = id & 3); tile row i <-> run = rb*16 + i.
__global__ __launch_bounds__(256) void stem_conv_bwd_w_kernel(const float* __restrict__ clip, const bf16* __restrict__ G,
                                                              float* __restrict__ partial, StemGeom g) {
    __shared__ __attribute__((aligned(16))) bf16 PE[64][EOW];
    __shared__ __attribute__((aligned(16))) bf16 PO[64][EOW];
    __shared__ __attribute__((aligned(16))) bf16 GT[64][72];            // [n][m], m contiguous, 144-byte rows (16 B aligned)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, gq = lane >> 4;
    f32x4 acc[7][4];
#pragma unroll
    for (int a = 0; a < 7; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // G tile staging: 64 m x 64 n in 4x4 blocks, one block per thread: mi = 0..15 (x4 rows), ci = 0..15 (x4 cols)
    const int ci = (lane & 3) | ((lane >> 4) << 2), mi = ((lane >> 2) & 3) + 4 * wave;
    // software pipeline over the tiles: the patch / gradient loads of tile t+1 are issued right after tile t has been parked
    // in LDS, so their latency hides behind tile t's 56 MFMAs (one workgroup per CU: nothing else would hide it)
    float4 pre[NQUAD];
    uint2 gr[4];
    auto fetch = [&](int tile) {
        stem_load_patch4(clip, g, tile, pre);
        const int wt = tile % g.tilesW;
        const long row0 = (long)(tile / g.tilesW) * g.Wo + wt * SW;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int w = wt * SW + mi * 4 + j;
            gr[j] = w < g.Wo ? *(const uint2*)(G + (row0 + mi * 4 + j) * 64 + ci * 4) : make_uint2(0, 0);
        }
    };
    if (blockIdx.x < g.ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < g.ntiles; tile += gridDim.x) {
        __syncthreads();                                     // previous tile's MFMAs are done with PE / PO / GT
        stem_store_patch_eo(PE, PO, pre);
        {
            bf16x4 x[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] = as_bf16x4(gr[j]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                bf16x4 y = {x[0][c], x[1][c], x[2][c], x[3][c]};
                *(uint2*)&GT[ci * 4 + c][mi * 4] = as_uint2(y);
            }
        }
        if (tile + (int)gridDim.x < g.ntiles) fetch(tile + gridDim.x);
        __syncthreads();
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) {
            bf16x8 gb[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) gb[nt] = as_bf16x8(*(const uint4*)&GT[nt * 16 + li][ms * 32 + gq * 8]);
#pragma unroll
            for (int kt = 0; kt < 7; ++kt) {
                const int id = wave * 7 + kt, kw = id >> 2, rb = id & 3;
                const bf16* row = (kw & 1) ? &PO[rb * 16 + li][0] : &PE[rb * 16 + li][0];
                const bf16x8 pa = stem_frag_shift(row, ms * 32 + gq * 8, kw >> 1);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) acc[kt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, gb[nt], acc[kt][nt], 0, 0, 0);
            }
        }
    }
    // D[i = run][j = n]: lane holds n = nt*16 + li, run = rb*16 + gq*4 + r; slab layout [k' = run*8 + kw][n] (kw = 7 rows zero)
    float* o = partial + (long)blockIdx.x * KP * 64;
#pragma unroll
    for (int kt = 0; kt < 7; ++kt) {
        const int id = wave * 7 + kt, kw = id >> 2, rb = id & 3;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                o[(long)((rb * 16 + gq * 4 + r) * 8 + kw) * 64 + nt * 16 + li] = acc[kt][nt][r];
    }
}

// W[64][441] fp32 -> Wp[64][512] bf16 with k' = run*8 + kw (zero padding)
__global__ void stem_pack_w_kernel(const float* __restrict__ W, bf16* __restrict__ Wp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * KP) return;
    const int n = i / KP, kp = i % KP, run = kp >> 3, kw = kp & 7;
    Wp[i] = (run < 63 && kw < 7) ? f2bf(W[n * 441 + run * 7 + kw]) : (bf16)0.f;
}
// dW[n][run*7+kw] (+)= sum_wg partial[wg][k'][n]
__global__ __launch_bounds__(1024) void stem_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW, int nwg, int accumulate) {
    __shared__ float red[32][33];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cl;                   // over k'*64 + n
    float a = 0.f;
    for (int s = rg; s < nwg; s += 32) a += partial[(long)s * KP * 64 + i];
    red[rg][cl] = a;
    __syncthreads();
    if (rg == 0) {
        a = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) a += red[k][cl];
        const int kp = i / 64, n = i % 64, run = kp >> 3, kw = kp & 7;
        if (run < 63 && kw < 7) {
            float* o = dW + n * 441 + run * 7 + kw;
            *o = accumulate ? *o + a : a;
        }
    }
}

static StemGeom stem_geom(int B, int T, int H, int W) {
    StemGeom g;
    g.B = B; g.T = T; g.H = H; g.W = W;
    g.Ho = (H + 6 - 7) / 2 + 1; g.Wo = (W + 6 - 7) / 2 + 1;
    g.tilesW = (g.Wo + SW - 1) / SW;
    g.ntiles = B * T * g.Ho * g.tilesW;
    return g;
}

extern "C" {

// persistent grid of both kernels (= partial-statistics rows of the forward, slabs of the weight gradient)
int tuber_stem_conv_blocks(int B, int T, int H, int W) {
    const StemGeom g = stem_geom(B, T, H, W);
    return g.ntiles < 1024 ? g.ntiles : 1024;
}

int tuber_stem_pack_weight(const float* W, void* Wp, hipStream_t stream) {
    hipLaunchKernelGGL(stem_pack_w_kernel, dim3(64 * KP / 256), dim3(256), 0, stream, W, (bf16*)Wp);
    TUBER_RETURN_LAUNCH();
}

// out [B*T*Ho*Wo, 64] bf16 raw conv output (NDHWC); st0/st1 [blocks][64] partial (sum, sum^2) or NULL
int tuber_stem_conv_fwd(const float* clip, const void* Wp, void* out, float* st0, float* st1, int B, int T, int H, int W,
                        hipStream_t stream) {
    if (B <= 0 || T <= 0 || H < 7 || W < 7) return TUBER_EINVAL;
    const StemGeom g = stem_geom(B, T, H, W);
    hipLaunchKernelGGL(stem_conv_fwd_kernel, dim3(tuber_stem_conv_blocks(B, T, H, W)), dim3(256), 0, stream, clip, (const bf16*)Wp,
                       (bf16*)out, st0, st1, g);
    TUBER_RETURN_LAUNCH();
}

// dW [64][441] fp32 (+)= conv weight gradient from G = d(loss)/d(raw conv output) [M,64] bf16;
// partial must hold blocks * 512 * 64 floats
int tuber_stem_conv_bwd_weight(const float* clip, const void* G, float* partial, float* dW, int accumulate, int B, int T, int H, int W,
                               hipStream_t stream) {
    if (B <= 0 || T <= 0 || H < 7 || W < 7) return TUBER_EINVAL;
    const StemGeom g = stem_geom(B, T, H, W);
    int nwg = tuber_stem_conv_blocks(B, T, H, W);
    if (nwg > 256) nwg = 256;
    hipLaunchKernelGGL(stem_conv_bwd_w_kernel, dim3(nwg), dim3(256), 0, stream, clip, (const bf16*)G, partial, g);
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(KP * 64 / 32), dim3(1024), 0, stream, partial, dW, nwg, accumulate);
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
