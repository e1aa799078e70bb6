// Shared device helpers for the TubeR gfx950 kernels (wave64, MFMA bf16, LDS-tiled).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define TUBER_OK 0
#define TUBER_EINVAL (-1)

// Every extern "C" launcher ends with this: report a launch failure as its hipError_t code.
#define TUBER_RETURN_LAUNCH()                      \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        return e__ == hipSuccess ? TUBER_OK : (int)e__; \
    } while (0)

// Opt a kernel in to more than 64 KB of dynamic LDS.  The attribute is PER DEVICE, so the "done" state is a bit per device id and the
// runtime's answer is checked (a process that drives a second GPU must opt in there too).  Returns TUBER_OK or the hipError_t.
struct LdsOptIn {
    unsigned long long done = 0;       // benign race: two threads may both set the attribute, the value is the same
};
static inline int lds_opt_in(LdsOptIn& st, const void* fn, size_t bytes) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return (int)hipGetLastError();
    const unsigned long long bit = 1ull << (dev & 63);
    if (st.done & bit) return TUBER_OK;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); return (int)e; }
    st.done |= bit;
    return TUBER_OK;
}
#define TUBER_LDS_OPT_IN(st, fn, bytes)                          \
    do {                                                         \
        int rc__ = lds_opt_in(st, (const void*)(fn), bytes);     \
        if (rc__ != TUBER_OK) return rc__;                       \
    } while (0)

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }

// 16-byte vector <-> 8 bf16
__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ uint4 as_uint4(bf16x8 v) { return __builtin_bit_cast(uint4, v); }
__device__ __forceinline__ bf16x4 as_bf16x4(uint2 v) { return __builtin_bit_cast(bf16x4, v); }
__device__ __forceinline__ uint2 as_uint2(bf16x4 v) { return __builtin_bit_cast(uint2, v); }

// XCD-aware block remap (8 XCDs; block b runs on XCD b % 8): give each XCD a contiguous
// range of logical tile ids so neighbouring tiles share an L2.  Bijective for any grid size.
__device__ __forceinline__ int xcd_remap(int b, int nb) {
    const int q = nb >> 3, r = nb & 7, xcd = b & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (b >> 3);
}

// Lane exchange inside a 16-lane DPP row: one VALU operand modifier, no LDS traffic (``__shfl_xor`` compiles to ds_bpermute_b32,
// an LDS-pipe instruction plus address arithmetic).  xor 1 / xor 2 are quad permutes; once the four lanes of a quad hold the same
// value, row_half_mirror (lane l <- 7-l inside each 8) delivers the neighbouring quad's value and row_mirror (l <- 15-l) the other
// half-row's: the results are bit-identical to the xor butterflies they replace.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
#define DPP_XOR1 0xB1          /* quad_perm [1,0,3,2] */
#define DPP_XOR2 0x4E          /* quad_perm [2,3,0,1] */
#define DPP_HALF_MIRROR 0x141  /* after xor1+xor2: acts as xor 4 */
#define DPP_MIRROR 0x140       /* after xor1+xor2+xor4: acts as xor 8 */
#define DPP_ROR8 0x128         /* row_ror:8 = lane ^ 8 inside the 16-lane row, for arbitrary values */

// sum over the 16 lanes that share (lane >> 4); every lane gets the total
__device__ __forceinline__ float quad16_sum(float v) {
    v += dpp_f32<DPP_XOR1>(v);
    v += dpp_f32<DPP_XOR2>(v);
    v += dpp_f32<DPP_HALF_MIRROR>(v);
    v += dpp_f32<DPP_MIRROR>(v);
    return v;
}
__device__ __forceinline__ float quad16_max(float v) {
    v = fmaxf(v, dpp_f32<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f32<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f32<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f32<DPP_MIRROR>(v));
    return v;
}
// Exchange ACROSS the 16-lane rows: the gfx950 row swaps (v_permlane16_swap / v_permlane32_swap: VALU, no LDS).  Given two copies of
// v, the swap leaves {own row pair's even row} in one and {odd row} in the other for every lane, so their sum / max is what
// v (+|max) __shfl_xor(v, 16 | 32) computes -- bit-identical (the operands are the same two values), without the two ds_bpermute
// round trips that sat in the dependent chain of every softmax column maximum.
// (The two results are copied into scalars before the bit cast: __builtin_bit_cast applied to the vector ELEMENT r[1] read element 0
// with this compiler, which made every such reduction x + x -- found by the depthwise weight-gradient test.)
__device__ __forceinline__ void row_swap16(float v, float& x, float& y) {
    const unsigned a = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(a, a, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    x = __builtin_bit_cast(float, r0); y = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ void row_swap32(float v, float& x, float& y) {
    const unsigned a = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    const unsigned r0 = r[0], r1 = r[1];
    x = __builtin_bit_cast(float, r0); y = __builtin_bit_cast(float, r1);
}
__device__ __forceinline__ float xor16_sum(float v) { float x, y; row_swap16(v, x, y); return x + y; }
__device__ __forceinline__ float xor32_sum(float v) { float x, y; row_swap32(v, x, y); return x + y; }
__device__ __forceinline__ float xor16_max(float v) { float x, y; row_swap16(v, x, y); return fmaxf(x, y); }
__device__ __forceinline__ float xor32_max(float v) { float x, y; row_swap32(v, x, y); return fmaxf(x, y); }
// fp64 forms (both 32-bit halves take the same lane exchange).  Without these overloads a double argument converts to float silently:
// bn_partial_sums reduced its fp64 partial sums across lanes in fp32 until round 3.
__device__ __forceinline__ double f64_of(unsigned lo, unsigned hi) { return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo); }
__device__ __forceinline__ double xor16_sum(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
    const auto rl = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
    const unsigned l0 = rl[0], l1 = rl[1], h0 = rh[0], h1 = rh[1];
    return f64_of(l0, h0) + f64_of(l1, h1);
}
__device__ __forceinline__ double xor32_sum(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
    const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const unsigned l0 = rl[0], l1 = rl[1], h0 = rh[0], h1 = rh[1];
    return f64_of(l0, h0) + f64_of(l1, h1);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xF, 0xF, true);
    const unsigned hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xF, 0xF, true);
    return f64_of(lo, hi);
}
__device__ __forceinline__ float wave_sum(float v) { return xor32_sum(xor16_sum(quad16_sum(v))); }
__device__ __forceinline__ float wave_max(float v) { return xor32_max(xor16_max(quad16_max(v))); }

// Counter-based RNG for dropout: one 32-bit hash per (seed, element index).  Deterministic and
// stateless so the backward pass regenerates the forward mask instead of storing it.
__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
// One 32-bit hash word decides TWO consecutive elements: element idx takes the 16-bit field (idx & 1) of word(idx >> 1) and is kept iff
// field >= thresh >> 16 (thresh = p * 2^32, so the drop probability is floor(p * 65536) / 65536: 0.099991 for p = 0.1).  The two
// 32-bit multiplies of the hash are quarter-rate instructions -- the mask of the class-branch FFN activation alone (34.6 M elements) cost
// 14 us of VALU time per pass with one word per element -- so every site that owns a run of consecutive elements (GEMM epilogue rows,
// LayerNorm rows, four keys of a score column) calls dropout_keep_run and hashes once per pair.
__device__ __forceinline__ uint32_t dropout_word(uint64_t seed, uint64_t pair) {
    // the inner hash depends on the seed and the HIGH word of the pair index only: for the first 2^33 elements of a tensor (all of them,
    // in this model) it is loop-invariant and hoisted; the branch keeps the stream defined for larger tensors
    const uint32_t hi = (uint32_t)(pair >> 32);
    uint32_t inner = hash_u32((uint32_t)seed);
    if (__builtin_expect(hi != 0, 0)) inner = hash_u32(hi + (uint32_t)seed);
    return hash_u32((uint32_t)pair ^ inner ^ (uint32_t)(seed >> 32) * 0x9E3779B9U);
}
// the scale of the kept elements, from the QUANTISED drop probability floor(p * 65536) / 65536 the masks realise (so the expectation of
// dropout(x) is exactly x, and a p below 2^-16 -- which drops nothing -- scales by 1)
__host__ __device__ __forceinline__ float dropout_inv_keep(float p) {
    const uint32_t t16 = (uint32_t)((double)p * 4294967296.0) >> 16;
    return t16 ? 65536.f / (float)(65536u - t16) : 1.f;
}
__device__ __forceinline__ bool dropout_keep(uint64_t seed, uint64_t idx, uint32_t thresh) {
    const uint32_t w = dropout_word(seed, idx >> 1);
    return ((idx & 1) ? (w >> 16) : (w & 0xFFFFu)) >= (thresh >> 16);
}
// keep[e] = dropout_keep(seed, idx0 + e, thresh) for a run of N (even) consecutive elements: N / 2 words when the run starts on an even
// element (every call site of the model: row lengths are even), N words otherwise
template <int N>
__device__ __forceinline__ void dropout_keep_run(uint64_t seed, uint64_t idx0, uint32_t thresh, bool (&keep)[N]) {
    static_assert(N % 2 == 0, "runs of pairs");
    const uint32_t t16 = thresh >> 16;
    if ((idx0 & 1) == 0) {
#pragma unroll
        for (int j = 0; j < N / 2; ++j) {
            const uint32_t w = dropout_word(seed, (idx0 >> 1) + j);
            keep[2 * j] = (w & 0xFFFFu) >= t16;
            keep[2 * j + 1] = (w >> 16) >= t16;
        }
    } else {
#pragma unroll
        for (int e = 0; e < N; ++e) keep[e] = dropout_keep(seed, idx0 + e, thresh);
    }
}
