// Backward of the bottleneck's second pointwise conv (conv4) for the wide-activation stage, as ONE persistent kernel -- gfx950.
// reference: autograd of  out = bn4(conv4(relu(bn3(c3))))  in ResNeXtBottleneck.forward, models/backbones/ir_CSN_152.py:58-64,78-84.
//
// In layer1 (M = 348 160 rows, C4 = 256, P = 64) the three kernels this replaces are pure data movement over [M, C4] tensors:
//     bn_bwd_fa        dc4 = cA*dz + cB*c4 + cC                     reads dz, c4   writes dc4        (3 passes)
//     gemm_nt (epi 2)  dz3 = (dc4 . W4) * [bn3(c3) > 0] + stats     reads dc4                       (1 pass)
//     gemm_tn          dW4 = dc4^T . relu(bn3(c3))                  reads dc4                       (1 pass)
// Here a workgroup walks 64-row tiles: it loads dz, c4 (and the small c3) ONCE, forms the bf16 dc4 tile in LDS, and runs both GEMMs
// from that image -- the data gradient reads it row-major (k = channel contiguous), the weight gradient through the gfx950 LDS
// transpose read (ds_read_b64_tr_b16) -- so dc4 never exists in HBM: 2 passes over [M, C4] instead of 5.  The weight gradient stays in
// 64 accumulator registers per thread over all of a workgroup's tiles and leaves as one fp32 slab per workgroup (reduced by the
// caller's deferred tuber_multi_reduce, like every other weight gradient).  BatchNorm-backward coefficients come from
// tuber_bn_bwd_finalize (cA / cB / cC per channel); statistics rows of dz3 (sum dz3, sum dz3*c3 per 64-row tile) are written exactly
// as tuber_gemm_nt(epi 2) writes them, so the consumers (bn3's backward inside the depthwise kernels) do not change.
// Bound: HBM, 2*M*(2*C4 + 2*P) bytes per launch (435 MB in layer1).  LDS: dc4 image 32 KB + W4^T 32 KB + raw c3 image 8 KB (the
// activation relu(bn3(.)) is applied to the transposed fragments in registers) = 74.5 KB, two workgroups per CU.
#include "common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int C4 = 256, P = 64, TR = 64;          // channels of dz / c4, channels of c3 / dz3, rows per tile
constexpr int GP = 256;                            // pitch (bf16) of the dc4 and W4^T images

// [rows][256] bf16 images: the 32-byte units of a 512-byte row are XOR-swizzled by a 4-bit row key that is distinct (a) over the 8
// rows {r..r+3, r+8..r+11} a 32-lane transposed read touches and (b) over 16 consecutive rows a 16-lane row-major 16-byte read touches;
// the two 16-byte halves of a unit swap with row bit 2 so that case (b)'s units u and u+8 (same banks) use different halves.
__device__ __forceinline__ int gkey(int row) { return (row & 3) | (((row >> 3) & 1) << 2) | (((row >> 2) & 1) << 3); }
__device__ __forceinline__ int goff(int row, int col) {
    return row * GP + ((((col >> 4) ^ gkey(row)) << 4) | ((col & 15) ^ (((row >> 2) & 1) << 3)));
}
// [64][64] bf16 images (a3, raw c3): the layout of gemm.hip's transpose-read kernel (128-byte rows, 2-bit key on the 32-byte units)
__device__ __forceinline__ int akey(int row) { return ((row >> 1) & 1) | ((row >> 2) & 2); }
__device__ __forceinline__ int aoff(int row, int col) { return row * 64 + ((((col >> 4) ^ akey(row)) << 4) | (col & 15)); }

__device__ __forceinline__ s16x4 tr_read(const bf16* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
}
// 8 consecutive rows m0..m0+7 of column col0 + li (li = lane & 15) from a row-major image: two transposed [4 m][16 cols] reads
__device__ __forceinline__ bf16x8 g_tr_frag(const bf16* img, int m0, int col0, int li) {
    const int r = m0 + (li >> 2), c = col0 + (li & 3) * 4;
    const s16x4 lo = tr_read(img + goff(r, c)), hi = tr_read(img + goff(r + 4, c));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 a_tr_frag(const bf16* img, int m0, int col0, int li) {
    const bf16* p = img + aoff(m0 + (li >> 2), col0 + (li & 3) * 4);          // row + 4 has the same key (m0 % 8 == 0)
    const s16x4 lo = tr_read(p), hi = tr_read(p + 4 * 64);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

struct Conv4BwdArgs {
    const bf16* dz; const bf16* c4;        // [M, C4]: gradient of bn4's output (after the join mask), bn4's input
    const bf16* c3;                        // [M, P]: bn3's input (raw depthwise-conv output)
    const bf16* w4t; long ldw;             // conv4 weight TRANSPOSED: [P][ldw] bf16, element (p, c) = W4[c][p]
    const float* cA; const float* cB; const float* cC;     // bn4 backward: dc4 = cA*dz + cB*c4 + cC per channel
    const float* sc3; const float* sh3;    // bn3 apply: a3 = relu(c3*sc3 + sh3)
    bf16* dz3;                             // [M, P] out: (dc4 . W4) * [a3 > 0]
    float* st0; float* st1;                // [tiles][P] out: per-tile sum dz3, sum dz3*c3
    float* slab;                           // [gridDim.x][C4][P] fp32 out: this workgroup's part of dW4 = dc4^T . a3
    long M;
};

// PLAIN: the projection-shortcut form (down_sample conv + BatchNorm of a stage's first block, ir_CSN_152.py:86-87): the conv input is the
// block input itself -- no BatchNorm / ReLU in front of it, so no activation on the weight-gradient operand, no mask on the data
// gradient and no statistics rows.
template <bool PLAIN>
__global__ __launch_bounds__(256, 2) void conv4_bwd_kernel(Conv4BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* gimg = (bf16*)smem_raw;                       // [64][256] dc4 tile
    bf16* wimg = gimg + TR * GP;                        // [64 p][256 c] W4^T
    bf16* cimg = wimg + P * GP;                         // [64][64] raw c3 (bn3's input)
    float* red = (float*)(cimg + TR * 64);              // [4 waves][64 p][2]
    float* tab3 = red + 4 * P * 2;                      // [2][64] bn3 scale / shift
    float* tab4 = tab3 + 2 * P;                         // [3][256] bn4 backward coefficients cA | cB | cC
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const long ntiles = (a.M + TR - 1) / TR;

    // ---- once per workgroup: W4^T image, per-thread coefficient registers ----
    {
        const int r = tid >> 5, ch = tid & 31;           // 8 rows x 32 chunks of 8 columns per pass
#pragma unroll
        for (int h = 0; h < P / 8; ++h) {
            const int p = r + 8 * h;
            *(uint4*)(wimg + goff(p, ch * 8)) = *(const uint4*)(a.w4t + (long)p * a.ldw + ch * 8);
        }
    }
    const int gr = tid >> 5, gch = tid & 31;             // dz / c4 staging: rows gr + 8 h, 16-byte chunk gch (channels gch*8 ..)
    tab4[tid] = a.cA[tid]; tab4[C4 + tid] = a.cB[tid]; tab4[2 * C4 + tid] = a.cC[tid];        // 256 threads = 256 channels
    const int cr = tid >> 3, cch = tid & 7;              // c3 staging: rows cr + 32 h, chunk cch (channels cch*8 ..)
    if (!PLAIN && tid < P) { tab3[tid] = a.sc3[tid]; tab3[P + tid] = a.sh3[tid]; }
    float sA[4], hA[4];                                  // bn3 scale / shift of this lane's weight-gradient columns p = i*16 + li
#pragma unroll
    for (int i = 0; i < 4; ++i) { sA[i] = PLAIN ? 1.f : a.sc3[i * 16 + li]; hA[i] = PLAIN ? 0.f : a.sh3[i * 16 + li]; }

    f32x4 wacc[4][4];                                    // dW block (p block i, c block j): p = i*16 + li, c = 64*wave + j*16 + g*4 + r
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) wacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // prefetch registers as named scalars: as arrays (indexed in unrolled loops, in a lambda or a macro) the compiler left some of them in
    // scratch memory, and the scratch store right behind the loads made every prefetch wait for them at issue
    uint4 rz0, rz1, rz2, rz3, rz4, rz5, rz6, rz7, rc0, rc1, rc2, rc3, rc4, rc5, rc6, rc7, r30, r31;
#define LOAD_ROW(h, m0_)                                                          \
    do {                                                                          \
        const long m_ = min((m0_) + gr + 8 * h, a.M - 1);                         \
        rz##h = *(const uint4*)(a.dz + m_ * C4 + gch * 8);                        \
        rc##h = *(const uint4*)(a.c4 + m_ * C4 + gch * 8);                        \
    } while (0)
#define LOAD_C3(h, m0_)                                                           \
    do {                                                                          \
        const long m_ = min((m0_) + cr + 32 * h, a.M - 1);                        \
        r3##h = *(const uint4*)(a.c3 + m_ * P + cch * 8);                         \
    } while (0)
#define LOAD_TILE(tt)                                                             \
    do {                                                                          \
        const long m0_ = (tt) * TR;                                               \
        LOAD_ROW(0, m0_); LOAD_ROW(1, m0_); LOAD_ROW(2, m0_); LOAD_ROW(3, m0_);   \
        LOAD_ROW(4, m0_); LOAD_ROW(5, m0_); LOAD_ROW(6, m0_); LOAD_ROW(7, m0_);   \
        LOAD_C3(0, m0_); LOAD_C3(1, m0_);                                         \
    } while (0)
#define PUT_ROW(h)                                                                \
    do {                                                                          \
        const int row = gr + 8 * h;                                               \
        const bf16x8 z = as_bf16x8(rz##h), x = as_bf16x8(rc##h);                  \
        bf16x8 o;                                                                 \
        const bool ok = m0 + row < a.M;                                           \
        _Pragma("unroll") for (int e = 0; e < 8; ++e)                             \
            o[e] = f2bf(fmaf(kA[e], bf2f(z[e]), fmaf(kB[e], bf2f(x[e]), kC[e])));  \
        /* (rows beyond M were loaded from row M-1: computed, then cleared as a whole -- a select per element became a branch each) */ \
        *(uint4*)(gimg + goff(row, gch * 8)) = ok ? as_uint4(o) : make_uint4(0, 0, 0, 0); \
    } while (0)
    long t = blockIdx.x;
    LOAD_TILE(t);                                        // the grid never exceeds the tile count
    __syncthreads();                                     // tab3 / tab4 / wimg are in place
    for (; t < ntiles; t += gridDim.x) {
        const long m0 = t * TR;
        // ---- registers -> LDS images (rows beyond M contribute zeros) ----
        float kA[8], kB[8], kC[8];                        // this thread's 8 channels (from LDS each tile: 24 registers less across the MFMA phase)
        {
            const float4 a0 = *(const float4*)(tab4 + gch * 8), a1 = *(const float4*)(tab4 + gch * 8 + 4);
            const float4 b0 = *(const float4*)(tab4 + C4 + gch * 8), b1 = *(const float4*)(tab4 + C4 + gch * 8 + 4);
            const float4 c0 = *(const float4*)(tab4 + 2 * C4 + gch * 8), c1 = *(const float4*)(tab4 + 2 * C4 + gch * 8 + 4);
            kA[0] = a0.x; kA[1] = a0.y; kA[2] = a0.z; kA[3] = a0.w; kA[4] = a1.x; kA[5] = a1.y; kA[6] = a1.z; kA[7] = a1.w;
            kB[0] = b0.x; kB[1] = b0.y; kB[2] = b0.z; kB[3] = b0.w; kB[4] = b1.x; kB[5] = b1.y; kB[6] = b1.z; kB[7] = b1.w;
            kC[0] = c0.x; kC[1] = c0.y; kC[2] = c0.z; kC[3] = c0.w; kC[4] = c1.x; kC[5] = c1.y; kC[6] = c1.z; kC[7] = c1.w;
        }
        PUT_ROW(0); PUT_ROW(1); PUT_ROW(2); PUT_ROW(3); PUT_ROW(4); PUT_ROW(5); PUT_ROW(6); PUT_ROW(7);
        *(uint4*)(cimg + aoff(cr, cch * 8)) = r30;              // (rows beyond M: a copy of row M-1; their dc4 rows are zero)
        *(uint4*)(cimg + aoff(cr + 32, cch * 8)) = r31;
        LOAD_TILE(min(t + (long)gridDim.x, ntiles - 1));      // next tile's loads run under this tile's MFMA work (unconditional; the last one re-reads a valid tile)
        __syncthreads();

        // ---- data gradient: D[p][m] = sum_c W4^T[p][c] * dc4[m][c]; wave w owns rows m = 16 w + li, all 64 p ----
        f32x4 dacc[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) dacc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
        for (int ks = 0; ks < C4 / 32; ++ks) {
            const int kc = ks * 32 + g * 8;
            const bf16x8 fg = as_bf16x8(*(const uint4*)(gimg + goff(16 * wave + li, kc)));
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const bf16x8 fw = as_bf16x8(*(const uint4*)(wimg + goff(n * 16 + li, kc)));
                dacc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw, fg, dacc[n], 0, 0, 0);
            }
        }
        // ---- data-gradient epilogue: lane holds row m = 16 w + li, columns p = n*16 + g*4 .. +3 ----
        {
            const int row = 16 * wave + li;
            const bool ok = m0 + row < a.M;
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const int p0 = n * 16 + g * 4;
                bf16x4 o;
                if (PLAIN) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = f2bf(dacc[n][r]);
                    if (ok) *(uint2*)(a.dz3 + (m0 + row) * P + p0) = as_uint2(o);
                    continue;
                }
                const bf16x4 cv = as_bf16x4(*(const uint2*)(cimg + aoff(row, p0)));
                const float4 sc = *(const float4*)(tab3 + p0), sh = *(const float4*)(tab3 + P + p0);
                const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
                float s0[4], s1[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float d = (ok && fmaf(bf2f(cv[r]), scv[r], shv[r]) > 0.f) ? dacc[n][r] : 0.f;
                    o[r] = f2bf(d);
                    s0[r] = d;
                    s1[r] = d * bf2f(cv[r]);
                }
                if (ok) *(uint2*)(a.dz3 + (m0 + row) * P + p0) = as_uint2(o);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float x = quad16_sum(s0[r]), y = quad16_sum(s1[r]);
                    if (li == 0) {
                        red[(wave * P + p0 + r) * 2 + 0] = x;
                        red[(wave * P + p0 + r) * 2 + 1] = y;
                    }
                }
            }
        }
        // ---- weight gradient: D[c][p] += sum_m dc4[m][c] * a3[m][p]; wave w owns c = 64 w .. 64 w + 63 ----
#pragma unroll
        for (int ks = 0; ks < TR / 32; ++ks) {
            const int mm = ks * 32 + g * 8;
            bf16x8 fc[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fc[j] = g_tr_frag(gimg, mm, 64 * wave + j * 16, li);
#pragma unroll
            for (int i = 0; i < 4; ++i) {                // a3 = relu(bn3(c3)) formed on the transposed fragment (column p = i*16 + li)
                const bf16x8 x = a_tr_frag(cimg, mm, i * 16, li);
                bf16x8 fp = x;
                if (!PLAIN) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) fp[e] = f2bf(fmaxf(fmaf(bf2f(x[e]), sA[i], hA[i]), 0.f));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) wacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fc[j], fp, wacc[i][j], 0, 0, 0);
            }
        }
        __syncthreads();                                  // images are free again; the partial statistics are complete
        if (!PLAIN && tid < 2 * P) {
            const int which = tid >> 6, p = tid & 63;
            const float v = (red[(0 * P + p) * 2 + which] + red[(1 * P + p) * 2 + which]) + (red[(2 * P + p) * 2 + which] + red[(3 * P + p) * 2 + which]);
            (which ? a.st1 : a.st0)[t * P + p] = v;
        }
        // (red is rewritten only after the next tile's first barrier)
    }
    // ---- this workgroup's slab of dW4 [C4][P] ----
    float* out = a.slab + (long)blockIdx.x * C4 * P;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[(long)(64 * wave + j * 16 + g * 4 + r) * P + i * 16 + li] = wacc[i][j][r];
}

#undef LOAD_TILE
#undef LOAD_ROW
#undef LOAD_C3
#undef PUT_ROW
constexpr size_t kLds = (size_t)(TR * GP + P * GP + TR * 64) * sizeof(bf16) + (4 * P * 2 + 2 * P + 3 * C4) * sizeof(float);

}  // namespace

extern "C" {

// workgroups (= fp32 slabs of the weight gradient) tuber_conv4_bwd_fused launches for M rows: two per CU, never more than tiles
int tuber_conv4_bwd_slabs(long M) {
    const long tiles = (M + TR - 1) / TR;
    return (int)(tiles < 512 ? tiles : 512);
}

// shapes the fused kernel takes (the wide-activation stage of the CSN body: C4 = 256 output channels, P = 64 bottleneck channels)
int tuber_conv4_bwd_supported(int c4, int p) { return c4 == C4 && p == P; }

// dz, c4 [M, 256] bf16; c3 [M, 64] bf16; w4t = conv4 weight transposed [64][ldw] bf16; cA / cB / cC [256], sc3 / sh3 [64] fp32;
// dz3 [M, 64] bf16 out; st0 / st1 [ceil(M / 64)][64] fp32 out (same rows as tuber_gemm_nt epi 2); slab [tuber_conv4_bwd_slabs(M)][256][64]
// fp32 out (sum over the slabs = dW4 [256][64], conv4.weight layout).
// sc3 = sh3 = NULL: the projection-shortcut form (down_sample conv of a stage's first block: c4 = its output, c3 = the block input x,
// w4t = its weight transposed): dz3 = dc4 . W with no mask, the weight-gradient operand is x as it is, st0 / st1 are not written (may be NULL).
int tuber_conv4_bwd_fused(const void* dz, const void* c4, const void* c3, const void* w4t, long ldw, const float* cA, const float* cB,
                          const float* cC, const float* sc3, const float* sh3, void* dz3, float* st0, float* st1, float* slab,
                          long M, hipStream_t stream) {
    const bool plain = !sc3 && !sh3;
    if (!dz || !c4 || !c3 || !w4t || !cA || !cB || !cC || (!plain && (!sc3 || !sh3 || !st0 || !st1)) || !dz3 || !slab || M <= 0 || ldw < C4 || (ldw & 7))
        return TUBER_EINVAL;
    Conv4BwdArgs a;
    a.dz = (const bf16*)dz; a.c4 = (const bf16*)c4; a.c3 = (const bf16*)c3; a.w4t = (const bf16*)w4t; a.ldw = ldw;
    a.cA = cA; a.cB = cB; a.cC = cC; a.sc3 = sc3; a.sh3 = sh3; a.dz3 = (bf16*)dz3; a.st0 = st0; a.st1 = st1; a.slab = slab; a.M = M;
    static LdsOptIn opt[2];
    TUBER_LDS_OPT_IN(opt[0], conv4_bwd_kernel<false>, kLds);
    TUBER_LDS_OPT_IN(opt[1], conv4_bwd_kernel<true>, kLds);
    if (plain) hipLaunchKernelGGL(conv4_bwd_kernel<true>, dim3(tuber_conv4_bwd_slabs(M)), dim3(256), kLds, stream, a);
    else hipLaunchKernelGGL(conv4_bwd_kernel<false>, dim3(tuber_conv4_bwd_slabs(M)), dim3(256), kLds, stream, a);
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
