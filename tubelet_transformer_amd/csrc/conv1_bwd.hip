// Backward of the bottleneck's first pointwise conv (conv1) for the wide-activation stage, as ONE persistent kernel -- gfx950.
// reference: autograd of  out = bn1(conv1(x))  in ResNeXtBottleneck.forward, models/backbones/ir_CSN_152.py:72-74, plus the residual
// join of the block below (:84-90) when that block is an identity block.
//
// In layer1 (M = 348 160 rows, 256-channel block input x, P = 64) this replaces
//     bn_bwd_apply / bn_bwd_fa   dc1 = cA*dz1 + cB*c1 + cC                          reads dz1, c1   writes dc1
//     gemm_nt (plain | join)     dx = dc1 . W1 + R   [join: dz = dx * [x > 0] + statistics of the lower block's bn4]
//     gemm_tn                    dW1 = dc1^T . x                                     reads dc1 and x AGAIN
// A 512-thread workgroup walks 64-row tiles: dz1 / c1 are read once and dc1 lives only in LDS; the x tile -- mask source of the join and
// operand of the weight gradient -- is read ONCE and parked in LDS for both uses; the weight gradient stays in accumulator registers
// over all of a workgroup's tiles (one fp32 slab per workgroup for the caller's deferred reduction).  Wave w owns the 32 input channels
// c = 32 w .. 32 w + 31 of every row: its slice of W1 sits in registers, its statistics columns need no cross-wave reduction.
// Results: dz / dx, the per-64-row statistics rows (sum dz, sum dz*c4_lower: exactly the rows tuber_gemm_nt_join writes) and dW1.
// Bound: HBM, join form 2*M*(4*256 + 2*64) bytes per launch.  LDS: x, R and (join) c4 images 32 KB each -- every operand arrives as
// coalesced 16-byte loads one tile ahead and is read back in the MFMA epilogue layout -- + two 8 KB images of dc1 (row-major / transposed use).
// (A first version fetched R / c4 in the epilogue layout straight from HBM, 8 bytes per lane one row block ahead: 257 us per launch --
// as slow as the three kernels it replaced.)
#include "common.h"

namespace {

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

constexpr int C = 256, P = 64, TR = 64, GP = 256, NTH = 512;

__device__ __forceinline__ int gkey(int row) { return (row & 3) | (((row >> 3) & 1) << 2) | (((row >> 2) & 1) << 3); }
__device__ __forceinline__ int goff(int row, int col) {
    return row * GP + ((((col >> 4) ^ gkey(row)) << 4) | ((col & 15) ^ (((row >> 2) & 1) << 3)));
}
// [64][64] image read through ds_read_b64_tr_b16 (gemm.hip's transpose-read layout)
__device__ __forceinline__ int akey(int row) { return ((row >> 1) & 1) | ((row >> 2) & 2); }
__device__ __forceinline__ int aoff(int row, int col) { return row * 64 + ((((col >> 4) ^ akey(row)) << 4) | (col & 15)); }
// [64][64] image read row-major in 16-byte pieces by 16 consecutive rows: 16-byte chunks XOR-swizzled like gemm_nt's activation tile
__device__ __forceinline__ int roff(int row, int col) { return row * 64 + ((((col >> 3) ^ ((row >> 1) & 7)) << 3) | (col & 7)); }

__device__ __forceinline__ s16x4 tr_read(const bf16* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
}
__device__ __forceinline__ bf16x8 g_tr_frag(const bf16* img, int m0, int col0, int li) {
    const int r = m0 + (li >> 2), c = col0 + (li & 3) * 4;
    const s16x4 lo = tr_read(img + goff(r, c)), hi = tr_read(img + goff(r + 4, c));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ bf16x8 a_tr_frag(const bf16* img, int m0, int col0, int li) {
    const bf16* p = img + aoff(m0 + (li >> 2), col0 + (li & 3) * 4);
    const s16x4 lo = tr_read(p), hi = tr_read(p + 4 * 64);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

struct Conv1BwdArgs {
    const bf16* dz1; const bf16* c1;       // [M, P]: gradient of bn1's output (after the ReLU mask), bn1's input
    const float* cA; const float* cB; const float* cC;   // bn1 backward coefficients [P]
    const bf16* w1t; long ldw;             // conv1 weight transposed: [C][ldw], element (c, p) = W1[p][c]
    const bf16* R;                         // [M, C] gradient already flowing into x (identity shortcut / projection data gradient) or NULL
    const bf16* X;                         // [M, C] the block input x (= the lower block's output y)
    const bf16* Cm;                        // join: the lower block's raw conv4 output [M, C]; NULL = plain form (out = dx)
    const bf16* Cd;                        // join below a stage's FIRST block: the raw output of its projection shortcut [M, C] (statistics of its BatchNorm) or NULL
    bf16* out;                             // [M, C]: dz of the lower block (join) or dx
    float* st0; float* st1;                // join: [tiles][C] statistics rows
    float* st2;                            // join with Cd: [tiles][C] rows of sum dz * cd
    float* slab;                           // [gridDim.x][P][C] fp32: this workgroup's part of dW1 (NULL: conv1 frozen)
    long M;
};

// JOIN: 0 = plain (out = dx), 1 = with the residual join of the identity block below, 2 = ... of the stage's first block below (a third
// statistics row for the projection shortcut's BatchNorm: its backward used to be a five-tensor block_out_bwd pass, 170 us in layer1)
template <int JOIN>
__global__ __launch_bounds__(NTH, 1) void conv1_bwd_kernel(Conv1BwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    bf16* ximg = (bf16*)smem_raw;                       // [64 m][256 c]
    bf16* rimg = ximg + TR * GP;                        // [64 m][256 c] residual gradient R
    bf16* cimg = rimg + TR * GP;                        // [64 m][256 c] lower block's c4 (join form)
    bf16* dimg = cimg + (JOIN ? TR * GP : 0);           // [64 m][256 c] lower block's projection output (JOIN == 2)
    bf16* dr = dimg + (JOIN == 2 ? TR * GP : 0);        // [64 m][64 p] dc1, row-major reads (data gradient)
    bf16* dt = dr + TR * 64;                            // [64 m][64 p] dc1, transposed reads (weight gradient)
    float* tab = (float*)(dt + TR * 64);                // [3][64] cA | cB | cC
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;     // 8 waves
    const int li = lane & 15, g = lane >> 4;
    const int cw = 32 * wave;                            // this wave's input-channel range
    const long ntiles = (a.M + TR - 1) / TR;
    if (tid < P) { tab[tid] = a.cA[tid]; tab[P + tid] = a.cB[tid]; tab[2 * P + tid] = a.cC[tid]; }
    // W1 slice of this wave as MFMA operands (row index li <-> c = cw + n*16 + li, 8 consecutive p at ks*32 + g*8), kept in registers
    bf16x8 wf[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int n = 0; n < 2; ++n) wf[ks][n] = as_bf16x8(*(const uint4*)(a.w1t + (long)(cw + n * 16 + li) * a.ldw + ks * 32 + g * 8));
    f32x4 wacc[4][2];                                    // dW block (p block i, c block j): p = i*16 + g*4 + r, c = cw + j*16 + li
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) wacc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int sr = tid >> 3, sch = tid & 7;              // dz1 / c1 staging: row sr (0..63), 16-byte chunk sch
    const int gr = tid >> 5, gch = tid & 31;             // x staging: rows gr + 16 h (h < 4), chunk gch
    // prefetch registers as named scalars: as arrays (indexed in unrolled loops, in a lambda or a macro) the compiler kept two of the three
    // in scratch memory, which made every prefetch wait for its loads at issue
    uint4 rz, rc, rx0, rx1, rx2, rx3, rr0, rr1, rr2, rr3, rm0, rm1, rm2, rm3, rd0, rd1, rd2, rd3;
    const bf16* Rp = a.R ? a.R : a.X;                    // no residual gradient: the loads stay unconditional, zeroed below
    const bool hasR = a.R != nullptr;
#define LOAD_ROW(h, m0_)                                                          \
    do {                                                                          \
        const long mx_ = min((m0_) + gr + 16 * h, a.M - 1);                       \
        rx##h = *(const uint4*)(a.X + mx_ * C + gch * 8);                         \
        rr##h = *(const uint4*)(Rp + mx_ * C + gch * 8);                          \
    } while (0)
#define LOAD_TILE(tt)                                                             \
    do {                                                                          \
        const long m0_ = (tt) * TR;                                               \
        const long m_ = min(m0_ + sr, a.M - 1);                                   \
        rz = *(const uint4*)(a.dz1 + m_ * P + sch * 8);                           \
        rc = *(const uint4*)(a.c1 + m_ * P + sch * 8);                            \
        LOAD_ROW(0, m0_); LOAD_ROW(1, m0_); LOAD_ROW(2, m0_); LOAD_ROW(3, m0_);   \
    } while (0)
// the lower block's conv4 output (JOIN) and projection output (JOIN == 2) are fetched LATE -- behind the MFMA / epilogue section of the previous tile -- so that their
// registers are not live across that section (as a fifth member of the tile-ahead prefetch set they sent 20 registers to scratch
// memory and every prefetch waited for its loads: 249 us per launch)
#define LOAD_D(tt)                                                                \
    do {                                                                          \
        const long m0_ = (tt) * TR;                                               \
        rd0 = *(const uint4*)(a.Cd + min(m0_ + gr, a.M - 1) * C + gch * 8);       \
        rd1 = *(const uint4*)(a.Cd + min(m0_ + gr + 16, a.M - 1) * C + gch * 8);  \
        rd2 = *(const uint4*)(a.Cd + min(m0_ + gr + 32, a.M - 1) * C + gch * 8);  \
        rd3 = *(const uint4*)(a.Cd + min(m0_ + gr + 48, a.M - 1) * C + gch * 8);  \
    } while (0)
#define LOAD_M(tt)                                                                \
    do {                                                                          \
        const long m0_ = (tt) * TR;                                               \
        rm0 = *(const uint4*)(a.Cm + min(m0_ + gr, a.M - 1) * C + gch * 8);       \
        rm1 = *(const uint4*)(a.Cm + min(m0_ + gr + 16, a.M - 1) * C + gch * 8);  \
        rm2 = *(const uint4*)(a.Cm + min(m0_ + gr + 32, a.M - 1) * C + gch * 8);  \
        rm3 = *(const uint4*)(a.Cm + min(m0_ + gr + 48, a.M - 1) * C + gch * 8);  \
    } while (0)
#define PUT_ROW(h)                                                                \
    do {                                                                          \
        *(uint4*)(ximg + goff(gr + 16 * h, gch * 8)) = rx##h;                     \
        *(uint4*)(rimg + goff(gr + 16 * h, gch * 8)) = hasR ? rr##h : make_uint4(0, 0, 0, 0); \
        if (JOIN) *(uint4*)(cimg + goff(gr + 16 * h, gch * 8)) = rm##h;           \
        if (JOIN == 2) *(uint4*)(dimg + goff(gr + 16 * h, gch * 8)) = rd##h;      \
    } while (0)
    rm0 = rm1 = rm2 = rm3 = make_uint4(0, 0, 0, 0);
    rd0 = rd1 = rd2 = rd3 = make_uint4(0, 0, 0, 0);
    long t = blockIdx.x;
    LOAD_TILE(t);                                        // the grid never exceeds the tile count
    if (JOIN) LOAD_M(t);
    if (JOIN == 2) LOAD_D(t);
    __syncthreads();                                     // tab
    for (; t < ntiles; t += gridDim.x) {
        const long m0 = t * TR;
        {   // dc1 tile -> both images (rows beyond M: zeros)
            const bf16x8 z = as_bf16x8(rz), x = as_bf16x8(rc);
            bf16x8 o;
            const bool ok = m0 + sr < a.M;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int p = sch * 8 + e;
                o[e] = f2bf(ok ? fmaf(tab[p], bf2f(z[e]), fmaf(tab[P + p], bf2f(x[e]), tab[2 * P + p])) : 0.f);
            }
            *(uint4*)(dr + roff(sr, sch * 8)) = as_uint4(o);
            *(uint4*)(dt + aoff(sr, sch * 8)) = as_uint4(o);
        }
        PUT_ROW(0); PUT_ROW(1); PUT_ROW(2); PUT_ROW(3);
        LOAD_TILE(min(t + (long)gridDim.x, ntiles - 1));      // next tile (unconditional: a predicated prefetch sent the registers through scratch); the last one re-reads a valid tile
        __syncthreads();

        // ---- weight gradient: D[p][c] += sum_m dc1[m][p] * x[m][c] (this wave: c = cw .. cw + 31) ----
        if (a.slab) {
#pragma unroll
            for (int ks = 0; ks < TR / 32; ++ks) {
                const int mm = ks * 32 + g * 8;
                bf16x8 fc[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) fc[j] = g_tr_frag(ximg, mm, cw + j * 16, li);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bf16x8 fp = a_tr_frag(dt, mm, i * 16, li);
#pragma unroll
                    for (int j = 0; j < 2; ++j) wacc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp, fc[j], wacc[i][j], 0, 0, 0);
                }
            }
        }
        // ---- data gradient per 16-row block: D[c][m] = sum_p W1[p][c] * dc1[m][p]; lane: row m = mb*16 + li, c = cw + n*16 + g*4 + r ----
        float s0[8], s1[8], s2[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { s0[q] = 0.f; s1[q] = 0.f; s2[q] = 0.f; }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 fd = as_bf16x8(*(const uint4*)(dr + roff(mb * 16 + li, ks * 32 + g * 8)));
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks][n], fd, acc[n], 0, 0, 0);
            }
            const int row = mb * 16 + li;
            const bool ok = m0 + row < a.M;
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int c0 = cw + n * 16 + g * 4;
                const bf16x4 rv = as_bf16x4(*(const uint2*)(rimg + goff(row, c0)));
                bf16x4 o;
                if (JOIN) {
                    const bf16x4 yv = as_bf16x4(*(const uint2*)(ximg + goff(row, c0)));
                    const bf16x4 cv = as_bf16x4(*(const uint2*)(cimg + goff(row, c0)));
                    bf16x4 dv = bf16x4{};
                    if (JOIN == 2) dv = as_bf16x4(*(const uint2*)(dimg + goff(row, c0)));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        // the stored dz is bf16: the statistics are taken of the ROUNDED value, like tuber_gemm_nt_join / tuber_block_out_bwd
                        const bf16 q = f2bf(acc[n][r] + bf2f(rv[r]));
                        const float d = (ok && bf2f(yv[r]) > 0.f) ? bf2f(q) : 0.f;
                        o[r] = f2bf(d);
                        s0[n * 4 + r] += d;
                        s1[n * 4 + r] += d * bf2f(cv[r]);
                        if (JOIN == 2) s2[n * 4 + r] += d * bf2f(dv[r]);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = f2bf(acc[n][r] + bf2f(rv[r]));
                }
                *(uint2*)(rimg + goff(row, c0)) = as_uint2(o);      // in place of the R piece just consumed: the tile leaves as 16-byte stores below
            }
        }
        if (JOIN) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const float x = quad16_sum(s0[q]), y = quad16_sum(s1[q]);
                if (li == 0) {
                    const int c = cw + (q >> 2) * 16 + g * 4 + (q & 3);
                    a.st0[t * C + c] = x;
                    a.st1[t * C + c] = y;
                }
                if (JOIN == 2) {
                    const float z = quad16_sum(s2[q]);
                    if (li == 0) a.st2[t * C + (cw + (q >> 2) * 16 + g * 4 + (q & 3))] = z;
                }
            }
        }
        __syncthreads();                                  // the output tile is complete in rimg
        if (JOIN) LOAD_M(min(t + (long)gridDim.x, ntiles - 1));
        if (JOIN == 2) LOAD_D(min(t + (long)gridDim.x, ntiles - 1));
#pragma unroll
        for (int h = 0; h < 4; ++h) {                     // coalesced 16-byte stores (8-byte pieces straight from the MFMA layout ran at 2.9 TB/s)
            const int row = gr + 16 * h;
            if (m0 + row < a.M) *(uint4*)(a.out + (m0 + row) * C + gch * 8) = *(const uint4*)(rimg + goff(row, gch * 8));
        }
        __syncthreads();                                  // images are rewritten by the next tile
    }
    if (a.slab) {
        float* out = a.slab + (long)blockIdx.x * P * C;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) out[(long)(i * 16 + g * 4 + r) * C + cw + j * 16 + li] = wacc[i][j][r];
    }
}

#undef LOAD_TILE
#undef LOAD_D
#undef LOAD_M
constexpr size_t kLdsJoin = (size_t)(3 * TR * GP + 2 * TR * 64) * sizeof(bf16) + 3 * P * sizeof(float);
constexpr size_t kLdsJoin2 = (size_t)(4 * TR * GP + 2 * TR * 64) * sizeof(bf16) + 3 * P * sizeof(float);
constexpr size_t kLdsPlain = (size_t)(2 * TR * GP + 2 * TR * 64) * sizeof(bf16) + 3 * P * sizeof(float);

}  // namespace

extern "C" {

// workgroups (= fp32 slabs of dW1) tuber_conv1_bwd_fused launches for M rows: one 512-thread workgroup per CU, never more than tiles
int tuber_conv1_bwd_slabs(long M) {
    const long tiles = (M + TR - 1) / TR;
    return (int)(tiles < 256 ? tiles : 256);
}

int tuber_conv1_bwd_supported(int cin, int p) { return cin == C && p == P; }

// dz1, c1 [M, 64] bf16; cA / cB / cC [64] fp32 (tuber_bn_bwd_finalize of bn1); w1t = conv1 weight transposed [256][ldw] bf16;
// R [M, 256] bf16 or NULL; X [M, 256] bf16 (block input); Cm [M, 256] bf16 (the lower block's raw conv4 output) selects the JOIN form:
// out = (dc1 . W1 + R) * [X > 0] with statistics rows st0 / st1 [ceil(M / 64)][256] (= tuber_gemm_nt_join); Cm NULL: out = dc1 . W1 + R.
// Cd [M, 256] bf16 (with Cm): the lower block is a stage's first block -- Cd = the raw output of its projection shortcut, st2 receives the rows
// sum dz * cd its BatchNorm backward needs (what tuber_block_out_bwd writes as its third statistics buffer).
// slab [tuber_conv1_bwd_slabs(M)][64][256] fp32 (sum over slabs = dW1, conv1.weight layout) or NULL when conv1 is frozen.
int tuber_conv1_bwd_fused(const void* dz1, const void* c1, const float* cA, const float* cB, const float* cC, const void* w1t, long ldw,
                          const void* R, const void* X, const void* Cm, const void* Cd, void* out, float* st0, float* st1, float* st2, float* slab,
                          long M, hipStream_t stream) {
    if (!dz1 || !c1 || !cA || !cB || !cC || !w1t || !X || !out || M <= 0 || ldw < P || (ldw & 7) || (Cm && (!st0 || !st1)) || (Cd && (!Cm || !st2)))
        return TUBER_EINVAL;
    Conv1BwdArgs a;
    a.dz1 = (const bf16*)dz1; a.c1 = (const bf16*)c1; a.cA = cA; a.cB = cB; a.cC = cC; a.w1t = (const bf16*)w1t; a.ldw = ldw;
    a.R = (const bf16*)R; a.X = (const bf16*)X; a.Cm = (const bf16*)Cm; a.Cd = (const bf16*)Cd; a.out = (bf16*)out;
    a.st0 = st0; a.st1 = st1; a.st2 = st2; a.slab = slab; a.M = M;
    static LdsOptIn opt[3];
    const dim3 grid(tuber_conv1_bwd_slabs(M)), block(NTH);
    if (Cd) {
        TUBER_LDS_OPT_IN(opt[2], conv1_bwd_kernel<2>, kLdsJoin2);
        hipLaunchKernelGGL(conv1_bwd_kernel<2>, grid, block, kLdsJoin2, stream, a);
    } else if (Cm) {
        TUBER_LDS_OPT_IN(opt[1], conv1_bwd_kernel<1>, kLdsJoin);
        hipLaunchKernelGGL(conv1_bwd_kernel<1>, grid, block, kLdsJoin, stream, a);
    } else {
        TUBER_LDS_OPT_IN(opt[0], conv1_bwd_kernel<0>, kLdsPlain);
        hipLaunchKernelGGL(conv1_bwd_kernel<0>, grid, block, kLdsPlain, stream, a);
    }
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
