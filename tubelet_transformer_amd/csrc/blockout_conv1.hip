// Block output + the NEXT bottleneck's first pointwise conv as ONE persistent kernel, for the wide-activation stage -- gfx950.
// reference: ResNeXtBottleneck.forward, models/backbones/ir_CSN_152.py:84-90 (out = relu(bn4(conv4(.)) + residual)) followed by :72-74
// of the next block (out = conv1(x); bn1 statistics).
//
// In layer1 (M = 348 160 rows of 256 channels) the residual join is a 3-pass elementwise kernel (reads c4 and the shortcut, writes y)
// and the next block's conv1 reads y again.  Here a workgroup walks 64-row tiles: it forms the y tile (same arithmetic as
// tuber_block_out_fwd, bit for bit), writes it to HBM for the backward pass, keeps it in LDS as bf16 and multiplies it with the next
// block's conv1 weight from there -- y is not read back: 3 passes over [M, 256] instead of 4.  The conv output c1 and its per-64-row
// statistics rows (sum, sum of squares) are what tuber_gemm_nt(epi 1) would have written.  LDS: y image 32 KB + weight image
// (32 KB for 64 output channels: two workgroups per CU; 64 KB for the 128 of layer2's first block).
// Bound: HBM, 2*M*(3*256 + PN) bytes.
#include "common.h"

namespace {

constexpr int C = 256, TR = 64, GP = 256;

// the swizzle of conv4_bwd.hip's [rows][256] images (conflict-free for 16 consecutive rows read as 16-byte pieces)
__device__ __forceinline__ int gkey(int row) { return (row & 3) | (((row >> 3) & 1) << 2) | (((row >> 2) & 1) << 3); }
__device__ __forceinline__ int goff(int row, int col) {
    return row * GP + ((((col >> 4) ^ gkey(row)) << 4) | ((col & 15) ^ (((row >> 2) & 1) << 3)));
}

struct BoC1Args {
    const bf16* c4; const float* s4; const float* h4;      // bn4(c4) = c4*s4 + h4
    const bf16* res; const float* rs; const float* rh;     // shortcut: res (identity) or res*rs + rh (projection + its BatchNorm)
    bf16* y;                                               // [M, 256] out
    uint8_t* ymask;                                        // optional: the ReLU mask of y as a bit field [M][32] (tuber_gemm_nt_join_strided_mask reads it)
    const bf16* w; long ldw;                               // next conv1 weight [PN][ldw] bf16 (row = output channel)
    bf16* c1;                                              // [M, PN] out
    float* st0; float* st1;                                // [tiles][PN] out (NULL in eval mode)
    long M;
};

template <int PN>
__global__ __launch_bounds__(256, PN == 64 ? 2 : 1) void blockout_conv1_kernel(BoC1Args a) {
    constexpr int NB = PN / 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* yimg = (bf16*)smem_raw;                       // [64][256]
    bf16* wimg = yimg + TR * GP;                        // [PN][256]
    float* red = (float*)(wimg + PN * GP);              // [4 waves][PN][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const long ntiles = (a.M + TR - 1) / TR;
    const int gr = tid >> 5, gch = tid & 31;             // staging: rows gr + 8 h, channels gch*8 ..
#pragma unroll
    for (int h = 0; h < PN / 8; ++h) {
        const int p = gr + 8 * h;
        *(uint4*)(wimg + goff(p, gch * 8)) = *(const uint4*)(a.w + (long)p * a.ldw + gch * 8);
    }
    float a4[8], b4[8], ar[8], br[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a4[e] = a.s4[gch * 8 + e]; b4[e] = a.h4[gch * 8 + e];
        ar[e] = a.rs ? a.rs[gch * 8 + e] : 1.f; br[e] = a.rs ? a.rh[gch * 8 + e] : 0.f;
    }
    const bool proj = a.rs != nullptr;
    uint4 rcv[8], rrv[8];
    auto load_tile = [&](long t) {
        const long m0 = t * TR;
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const long m = min(m0 + gr + 8 * h, a.M - 1);
            rcv[h] = *(const uint4*)(a.c4 + m * C + gch * 8);
            rrv[h] = *(const uint4*)(a.res + m * C + gch * 8);
        }
    };
    long t = blockIdx.x;
    load_tile(t);
    for (; t < ntiles; t += gridDim.x) {
        const long m0 = t * TR;
#pragma unroll
        for (int h = 0; h < 8; ++h) {
            const int row = gr + 8 * h;
            const bf16x8 c = as_bf16x8(rcv[h]), r = as_bf16x8(rrv[h]);
            bf16x8 o;
            unsigned bits = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float rv = bf2f(r[e]);
                if (proj) rv = fmaf(rv, ar[e], br[e]);
                o[e] = f2bf(fmaxf(fmaf(bf2f(c[e]), a4[e], b4[e]) + rv, 0.f));
                bits |= (bf2f(o[e]) > 0.f ? 1u : 0u) << e;
            }
            const uint4 ov = as_uint4(o);
            if (m0 + row < a.M) {
                *(uint4*)(a.y + (m0 + row) * C + gch * 8) = ov;
                if (a.ymask) a.ymask[(m0 + row) * (C / 8) + gch] = (uint8_t)bits;
            }
            *(uint4*)(yimg + goff(row, gch * 8)) = ov;
        }
        load_tile(min(t + (long)gridDim.x, ntiles - 1));
        __syncthreads();
        // ---- c1[m][p] = sum_c y[m][c] * W[p][c]; issued as D[p][m] so a lane ends up with 4 consecutive p of one row m ----
        f32x4 acc[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int ks = 0; ks < C / 32; ++ks) {
            const int kc = ks * 32 + g * 8;
            const bf16x8 fy = as_bf16x8(*(const uint4*)(yimg + goff(16 * wave + li, kc)));
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const bf16x8 fw = as_bf16x8(*(const uint4*)(wimg + goff(n * 16 + li, kc)));
                acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, fy, acc[n], 0, 0, 0);
            }
        }
        {
            const int row = 16 * wave + li;
            const bool ok = m0 + row < a.M;
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const int p0 = n * 16 + g * 4;
                bf16x4 o;
                float s0[4], s1[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[n][r];
                    o[r] = f2bf(v);
                    s0[r] = ok ? v : 0.f;
                    s1[r] = ok ? v * v : 0.f;
                }
                if (ok) *(uint2*)(a.c1 + (m0 + row) * PN + p0) = as_uint2(o);
                if (a.st0) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float x = quad16_sum(s0[r]), y2 = quad16_sum(s1[r]);
                        if (li == 0) {
                            red[(wave * PN + p0 + r) * 2 + 0] = x;
                            red[(wave * PN + p0 + r) * 2 + 1] = y2;
                        }
                    }
                }
            }
        }
        __syncthreads();
        if (a.st0 && tid < 2 * PN) {
            const int which = tid / PN, p = tid % PN;
            const float v = (red[(0 * PN + p) * 2 + which] + red[(1 * PN + p) * 2 + which]) + (red[(2 * PN + p) * 2 + which] + red[(3 * PN + p) * 2 + which]);
            (which ? a.st1 : a.st0)[t * PN + p] = v;
        }
    }
}

template <int PN>
int launch(const BoC1Args& a, hipStream_t stream) {
    const size_t lds = (size_t)(TR * GP + PN * GP) * sizeof(bf16) + 4 * PN * 2 * sizeof(float);
    static LdsOptIn opt;
    TUBER_LDS_OPT_IN(opt, blockout_conv1_kernel<PN>, lds);
    const long tiles = (a.M + TR - 1) / TR;
    const long slots = PN == 64 ? 512 : 256;
    hipLaunchKernelGGL(blockout_conv1_kernel<PN>, dim3((unsigned)(tiles < slots ? tiles : slots)), dim3(256), lds, stream, a);
    TUBER_RETURN_LAUNCH();
}

}  // namespace

extern "C" {

// shapes the fused forward kernel takes: 256-channel block output feeding a conv1 with 64 or 128 output channels (layer1, and layer1 -> layer2)
int tuber_blockout_conv1_supported(int c, int pn) { return c == C && (pn == 64 || pn == 128); }

// y = relu(c4*s4 + h4 + (rs ? res*rs + rh : res))  [M, 256] bf16 (= tuber_block_out_fwd);  c1 = y . w^T  [M, pn] bf16 with the per-64-row
// statistics rows st0 / st1 [ceil(M / 64)][pn] (sum, sum of squares; NULL in eval mode) that tuber_gemm_nt(epi 1) writes.
// w = the next block's conv1 weight [pn][ldw] bf16.
int tuber_blockout_conv1_fwd(const void* c4, const float* s4, const float* h4, const void* res, const float* rs, const float* rh,
                             void* y, const void* w, long ldw, void* c1, float* st0, float* st1, long M, int pn, hipStream_t stream) {
    if (!c4 || !s4 || !h4 || !res || !y || !w || !c1 || M <= 0 || ldw < C || (ldw & 7) || (rs && !rh) || (st0 && !st1)) return TUBER_EINVAL;
    BoC1Args a;
    a.c4 = (const bf16*)c4; a.s4 = s4; a.h4 = h4; a.res = (const bf16*)res; a.rs = rs; a.rh = rh; a.y = (bf16*)y;
    a.ymask = nullptr;
    a.w = (const bf16*)w; a.ldw = ldw; a.c1 = (bf16*)c1; a.st0 = st0; a.st1 = st1; a.M = M;
    if (pn == 64) return launch<64>(a, stream);
    if (pn == 128) return launch<128>(a, stream);
    return TUBER_EINVAL;
}

// the same launch, also writing the ReLU mask of y as a bit field ([M][32] bytes, bit e of byte (m, c / 8) = y[m][c + e] > 0) for the join backward across the
// stage boundary (tuber_gemm_nt_join_strided_mask)
int tuber_blockout_conv1_fwd_mask(const void* c4, const float* s4, const float* h4, const void* res, const float* rs, const float* rh,
                                  void* y, void* ymask, const void* w, long ldw, void* c1, float* st0, float* st1, long M, int pn, hipStream_t stream) {
    if (!c4 || !s4 || !h4 || !res || !y || !ymask || !w || !c1 || M <= 0 || ldw < C || (ldw & 7) || (rs && !rh) || (st0 && !st1)) return TUBER_EINVAL;
    BoC1Args a;
    a.c4 = (const bf16*)c4; a.s4 = s4; a.h4 = h4; a.res = (const bf16*)res; a.rs = rs; a.rh = rh; a.y = (bf16*)y; a.ymask = (uint8_t*)ymask;
    a.w = (const bf16*)w; a.ldw = ldw; a.c1 = (bf16*)c1; a.st0 = st0; a.st1 = st1; a.M = M;
    if (pn == 64) return launch<64>(a, stream);
    if (pn == 128) return launch<128>(a, stream);
    return TUBER_EINVAL;
}

}  // extern "C"
