// First bottleneck of the wide-activation stage: conv1 AND the projection-shortcut conv from ONE read of the block input -- gfx950.
// reference: ResNeXtBottleneck.forward, models/backbones/ir_CSN_152.py:72-74 (out = conv1(x); bn1) and :86-87 (residual =
// down_sample(x) = BatchNorm3d(Conv3d(64, 256, 1))(x)) for layer1's first block (stride 1).
//
// Both convs are 1x1x1 over the same [M, 64] input (M = 348 160 rows: the pooled stem output); as two GEMMs the input is read twice and
// the wide one (64 -> 256) ran at 2.8 TB/s.  Here a 512-thread workgroup walks 64-row tiles: the x tile (8 KB) goes to LDS once, both
// weight matrices stay resident in LDS ([64 + 256][64] bf16 = 40 KB), a wave computes 16 rows x 160 of the 320 output columns
// (K = 64: two MFMA k-steps), both output tiles are staged in LDS and leave as whole rows.  Per-64-row statistics rows (sum, sum of
// squares of the fp32 accumulators) are the rows tuber_gemm_nt(epi 1) writes for either conv (to 2^-16); the column sums run on
// the matrix pipe.
// Bound: HBM, 2*M*(64 + 64 + 256) bytes.
#include "common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int CI = 64, P1 = 64, CD = 256, NT = (P1 + CD) / 16, TR = 64, NTH = 512;
constexpr int OP = 256;                                  // pitch (bf16) of the staged projection-output image

// [rows][64] bf16 images (x tile, weights): 128-byte rows, the 16-byte chunks XOR-swizzled by 3 row bits (conflict-free for the 16 rows
// a row-major 16-byte read touches per k group)
__device__ __forceinline__ int xoff(int row, int col) { return row * 64 + ((((col >> 3) ^ (row & 7)) << 3) | (col & 7)); }
// [64][256] bf16 output image: 8-byte pieces written by (row li, columns n*16 + g*4), read back as 16-byte pieces of a row
__device__ __forceinline__ int ooff(int row, int col) { return row * OP + ((((col >> 3) ^ (row & 15)) << 3) | (col & 7)); }

// two / four floats -> packed bf16 (one v_cvt_pk_bf16_f32 per pair); the halves of a packed pair back as floats
__device__ __forceinline__ uint32_t pack2(float a, float b) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) { return make_uint2(pack2(a, b), pack2(c, d)); }
__device__ __forceinline__ float lo16(uint32_t p) { return __builtin_bit_cast(float, p << 16); }
__device__ __forceinline__ float hi16(uint32_t p) { return __builtin_bit_cast(float, p & 0xffff0000u); }

struct EntryArgs {
    const bf16* x;                          // [M, 64]
    const bf16* w1; long ldw1;              // conv1 weight [64][ldw1] bf16 (row = output channel)
    const bf16* wd; long ldwd;              // projection weight [256][ldwd]
    bf16* c1; bf16* cd;                     // [M, 64], [M, 256] out
    float* a0; float* a1;                   // [tiles][64] statistics rows of c1 (NULL in eval mode)
    float* d0; float* d1;                   // [tiles][256] statistics rows of cd
    long M;
};

__global__ __launch_bounds__(NTH, 1) void entry_conv_kernel(EntryArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* ximg = (bf16*)smem_raw;                       // [64][64]
    bf16* wimg = ximg + TR * CI;                        // [320][64]: rows 0..63 conv1, 64..319 projection
    bf16* oimg = wimg + (P1 + CD) * CI;                 // [64][256] projection output tile
    bf16* cimg = oimg + TR * OP;                        // [64][64] conv1 output tile
    float* red = (float*)(cimg + TR * P1);              // [4 row waves][320][2]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int rw = wave & 3, nh = wave >> 2;             // rows 16 rw .. + 15; column tiles nh*10 .. nh*10 + 9
    const long ntiles = (a.M + TR - 1) / TR;
    const bool stats = a.a0 != nullptr;
    {   // weights -> LDS: 320 rows x 8 chunks of 16 bytes
        for (int i = tid; i < (P1 + CD) * 8; i += NTH) {
            const int r = i >> 3, ch = i & 7;
            const bf16* src = r < P1 ? a.w1 + (long)r * a.ldw1 : a.wd + (long)(r - P1) * a.ldwd;
            *(uint4*)(wimg + xoff(r, ch * 8)) = *(const uint4*)(src + ch * 8);
        }
    }
    const int sr = tid >> 3, sch = tid & 7;              // x staging / c1 copy-out: row sr, chunk sch
    uint4 rx;
    long t = blockIdx.x;
    rx = *(const uint4*)(a.x + min(t * TR + sr, a.M - 1) * CI + sch * 8);
    for (; t < ntiles; t += gridDim.x) {
        const long m0 = t * TR;
        *(uint4*)(ximg + xoff(sr, sch * 8)) = rx;
        rx = *(const uint4*)(a.x + min(min(t + (long)gridDim.x, ntiles - 1) * TR + sr, a.M - 1) * CI + sch * 8);     // next tile
        __syncthreads();
        bf16x8 fx[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) fx[ks] = as_bf16x8(*(const uint4*)(ximg + xoff(16 * rw + li, ks * 32 + g * 8)));
        const int rbase = 16 * rw + g * 4;               // this lane's 4 output rows: rbase + r
        // rows beyond M (last tile only) get zero weight in the statistics: their k entries of the A operands are cleared
        const uint32_t k0 = m0 + rbase + 0 < a.M ? 0x0000ffffu : 0u, k1 = m0 + rbase + 1 < a.M ? 0xffff0000u : 0u;
        const uint32_t k2 = m0 + rbase + 2 < a.M ? 0x0000ffffu : 0u, k3 = m0 + rbase + 3 < a.M ? 0xffff0000u : 0u;
        const uint2 keep = make_uint2(k0 | k1, k2 | k3);
        const uint2 onesk = make_uint2(0x3F803F80u & keep.x, 0x3F803F80u & keep.y);      // bf16 1.0 per kept row
        const uint2 halfk = make_uint2(0x3F003F00u & keep.x, 0x3F003F00u & keep.y);      // bf16 0.5 per kept row (the remainder operand is stored doubled)
        const bool diag = g == (li >> 2);                // this lane holds element (li, li) of a 16 x 16 product, in register li & 3
#pragma unroll
        for (int j = 0; j < NT / 2; ++j) {
            const int n = nh * (NT / 2) + j;             // column tile: 0..3 conv1, 4..19 projection
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 fw = as_bf16x8(*(const uint4*)(wimg + xoff(n * 16 + li, ks * 32 + g * 8)));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fx[ks], fw, acc, 0, 0, 0);      // D[m][p]: lane = column p = n*16 + li, rows m = rbase + r
            }
            const uint2 hi = pack4(acc[0], acc[1], acc[2], acc[3]);
            const int q = n * 16 + li;                   // this lane's column among the 320
            {
                bf16* dst = n < P1 / 16 ? cimg : oimg;
                const bf16x4 h = as_bf16x4(hi);
#pragma unroll
                for (int r = 0; r < 4; ++r) dst[n < P1 / 16 ? xoff(rbase + r, q) : ooff(rbase + r, q - P1)] = h[r];
            }
            if (stats) {
                // Column statistics of the wave's 16 rows on the matrix pipe (the 16-lane DPP butterflies they replace -- 320 columns x 2
                // sums per tile -- made the kernel VALU-bound).  The accumulators v are split as H + L (bf16 head = the output values, bf16
                // remainder: ~16 mantissa bits, i.e. the fp32 accumulators to 2^-17).  This lane's packed H / L [16 m][16 p] are at once the
                // B operand (k = m, column p) and the A operand (row p, k = m) of a 16x16x16 MFMA:  ones . (H + L) = column sums;
                // H^T . H + 2 H^T . L has the sums of squares on its diagonal (L^T . L, 2^-18 of it, is dropped).
                const uint2 l2 = pack4(2.f * (acc[0] - lo16(hi.x)), 2.f * (acc[1] - hi16(hi.x)), 2.f * (acc[2] - lo16(hi.y)), 2.f * (acc[3] - hi16(hi.y)));
                const uint2 hk = make_uint2(hi.x & keep.x, hi.y & keep.y);
                f32x4 S = {0.f, 0.f, 0.f, 0.f}, Q = {0.f, 0.f, 0.f, 0.f};
                S = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, onesk), __builtin_bit_cast(s16x4, hi), S, 0, 0, 0);
                Q = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, hk), __builtin_bit_cast(s16x4, hi), Q, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, halfk), __builtin_bit_cast(s16x4, l2), S, 0, 0, 0);
                Q = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, hk), __builtin_bit_cast(s16x4, l2), Q, 0, 0, 0);
                const float qd = (li & 2) ? ((li & 1) ? Q[3] : Q[2]) : ((li & 1) ? Q[1] : Q[0]);
                if (diag) *(float2*)(red + (rw * 320 + q) * 2) = make_float2(S[0], qd);
            }
        }
        __syncthreads();
        {   // output tiles -> HBM as whole rows: projection 64 rows x 32 chunks of 16 bytes, conv1 64 rows x 8 chunks
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int r = (tid >> 5) + 16 * h, ch = tid & 31;
                if (m0 + r < a.M) *(uint4*)(a.cd + (m0 + r) * CD + ch * 8) = *(const uint4*)(oimg + ooff(r, ch * 8));
            }
            if (m0 + sr < a.M) *(uint4*)(a.c1 + (m0 + sr) * P1 + sch * 8) = *(const uint4*)(cimg + xoff(sr, sch * 8));
        }
        if (stats) {
            for (int i = tid; i < 2 * 320; i += NTH) {
                const int which = i >= 320, q = i - which * 320;
                const float v = (red[(0 * 320 + q) * 2 + which] + red[(1 * 320 + q) * 2 + which]) + (red[(2 * 320 + q) * 2 + which] + red[(3 * 320 + q) * 2 + which]);
                if (q < P1) (which ? a.a1 : a.a0)[t * P1 + q] = v;
                else (which ? a.d1 : a.d0)[t * CD + q - P1] = v;
            }
        }
        // (ximg is rewritten before the next barrier, oimg / cimg / red only after it)
    }
}

constexpr size_t kLds = (size_t)(TR * CI + (P1 + CD) * CI + TR * OP + TR * P1) * sizeof(bf16) + 4 * 320 * 2 * sizeof(float);

}  // namespace

extern "C" {

// shapes the fused kernel takes: 64-channel block input, conv1 to 64 channels, projection shortcut to 256 (layer1's first block)
int tuber_entry_conv_supported(int cin, int p, int c4) { return cin == CI && p == P1 && c4 == CD; }

// c1 = x . w1^T [M, 64] and cd = x . wd^T [M, 256] (bf16) from one pass over x [M, 64]; a0 / a1 [ceil(M / 64)][64] and d0 / d1
// [ceil(M / 64)][256]: per-64-row sums and sums of squares of the fp32 results (the rows tuber_gemm_nt(epi 1) writes; all four NULL in
// eval mode).  w1 [64][ldw1], wd [256][ldwd] bf16.
int tuber_entry_conv_fwd(const void* x, const void* w1, long ldw1, const void* wd, long ldwd, void* c1, void* cd,
                         float* a0, float* a1, float* d0, float* d1, long M, hipStream_t stream) {
    const int ns = (a0 != nullptr) + (a1 != nullptr) + (d0 != nullptr) + (d1 != nullptr);
    if (!x || !w1 || !wd || !c1 || !cd || M <= 0 || ldw1 < CI || ldwd < CI || ((ldw1 | ldwd) & 7) || (ns != 0 && ns != 4)) return TUBER_EINVAL;
    EntryArgs a;
    a.x = (const bf16*)x; a.w1 = (const bf16*)w1; a.ldw1 = ldw1; a.wd = (const bf16*)wd; a.ldwd = ldwd;
    a.c1 = (bf16*)c1; a.cd = (bf16*)cd; a.a0 = a0; a.a1 = a1; a.d0 = d0; a.d1 = d1; a.M = M;
    static LdsOptIn opt;
    TUBER_LDS_OPT_IN(opt, entry_conv_kernel, kLds);
    const long tiles = (M + TR - 1) / TR;
    hipLaunchKernelGGL(entry_conv_kernel, dim3((unsigned)(tiles < 256 ? tiles : 256)), dim3(NTH), kLds, stream, a);
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
