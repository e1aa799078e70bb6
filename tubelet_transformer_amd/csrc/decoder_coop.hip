// The DETR decoder stack (post-norm layers over <= 32 tubelet-query rows) as ONE cooperative launch on ONE XCD -- gfx950.
// reference: TransformerDecoder.forward / TransformerDecoderLayer.forward_post, models/transformer/transformer.py:99-128,218-249
// (self-attention with q = k = tgt + query_pos, cross-attention over the encoder memory, FFN 256 -> 2048 -> 256, three LayerNorms,
// the shared decoder.norm applied to every layer's output).
//
// Why: per layer the launch chain is 13 kernels of 1-32 workgroups on 30 rows x 256 -- ~80 dependent launches forward for the stack,
// each ~4.5 us on an otherwise idle chip.  The single-workgroup form of round 4 lost because ONE CU streams the
// 2.9 MB of weights per layer at ~30 GB/s.  Here 16 workgroups that the hardware places on ONE XCD (blockIdx % 8 == 0) split every
// weight matrix by output columns, exchange the 30 x 256 activations through that XCD's L2 and meet at an XCD-local barrier
// (scripts/microbench/grid_barrier_bench.hip: 0.9 us bare, 1.7 us with data) -- 8 barriers per layer:
//     in-proj | self-attention (one (clip, head) per workgroup) | out-proj | [norm1] q-proj | cross-attention | out-proj |
//     [norm2] FFN linear1 | FFN linear2 (four waves split K = 2048) | [norm3, decoder.norm] next layer's in-proj
// LayerNorm(Dropout(sublayer) + residual) is recomputed by EVERY workgroup in front of the GEMM that consumes it (30 rows: cheaper
// than a barrier); workgroup 0 stores its outputs.  Rounding points, dropout streams (seed, salt, element index) and every saved
// tensor (packed projections, attention outputs + log-sum-exp, LayerNorm xhat / rstd, the FFN activation) are those of the launch
// chain (tape.py: in_proj / attention / linear / layer_norm), so the EXISTING backward closures run unchanged on what this writes.
// The memory-side projections [(memory + pos) W_k | memory W_v] do not depend on the decoder state and stay GEMM launches in front.
//
// Visibility inside the launch: exchanged tensors are distinct buffers per layer, written once and read only AFTER the barrier that
// follows their producer, so no CU can hold a stale L1 line of them (L1 is cold at kernel start); stores are drained
// (s_waitcnt vmcnt(0)) by every writing wave before the arrival.  That is enough when the 16 workgroups share an L2 (one XCD).  The
// kernel does not TRUST the placement: the first barrier is agent-scope and collects every workgroup's XCC id; unless they are all
// equal, every barrier is bracketed by agent-scope release / acquire fences (slower, still correct).  Spins are bounded: a barrier
// that cannot complete raises the error word instead of hanging the GPU.
#include "attention_mfma_fwd.h"

namespace {

constexpr int E = 256, NH = 8, FF = 2048, MAXL = 6;
constexpr int G = 16;                      // cooperating workgroups = (clip, head) units of a 2-clip batch
constexpr int NT = 256;                    // threads per workgroup
constexpr int MR = 32;                     // activation rows held (two 16-row MFMA tiles); B * Q <= 32
constexpr int PX = 264;                    // bf16 pitch of a [32][256] operand image (528-byte rows: conflict-free 16-byte reads of 16 rows)
constexpr float LN_EPS = 1e-5f;

struct DLayer {
    const bf16 *w_in, *w_o1, *w_q, *w_o2, *w_f1, *w_f2;       // bf16 row-major [N][K]; w_q = rows [0, 256) of multihead_attn.in_proj_weight
    const float *b_in, *b_o1, *b_q, *b_o2, *b_f1, *b_f2;
    const float *g1, *e1, *g2, *e2, *g3, *e3;                 // norm1 .. norm3 weight / bias
    const bf16* kv;                                           // [B * Lm][512]: (memory + pos) W_k | memory W_v of this layer
    // saved tensors (what tape.py's ops of the launch chain allocate and their backward closures read)
    bf16* qkv; bf16* o1; float* lse1; bf16* a1;
    bf16* y1; bf16* xh1; float* rs1;
    bf16* q; bf16* o2; float* lse2; bf16* a2;
    bf16* y2; bf16* xh2; float* rs2;
    bf16* h; bf16* f2;
    bf16* y3; bf16* xh3; float* rs3;
    bf16* xhN; float* rsN;
    uint64_t salt_a1, salt_n1, salt_a2, salt_n2, salt_f, salt_n3;
};

struct DecCoopArgs {
    DLayer L[MAXL];
    int nl, B, Q, Lm, R;
    const bf16* qpos;                       // [R][256] rows (b, q): query_embed rows as the chain's param_rows gives them
    const float* gN; const float* eN;       // decoder.norm
    bf16* hs;                               // [nl][R][256]
    const uint8_t* kpm;                     // [B][Lm] or null
    float pdrop, pattn;                     // Dropout of the sublayer outputs / FFN activation, of the attention weights
    const uint64_t* seed_ptr;
    unsigned* sync;                         // [0] arrival counter, [1] XCC mask, [2] error word; zero at launch, left zero by a clean run
};

__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }      // HW_REG_XCC_ID[3:0]

struct Ctx {
    unsigned* sync;
    unsigned phase;        // barriers passed
    bool same_xcd;         // every workgroup reported the same XCC id: L2 is the coherence point
};

// all G workgroups arrive; bounded spin.  Every wave has drained its stores before the __syncthreads in front of the arrival.
__device__ __forceinline__ void coop_barrier(Ctx& c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned target = (c.phase + 1) * G;
        // bounded: ~0.1 s, then the error word is raised and every later barrier of every workgroup falls through at once
        if (c.same_xcd) {
            // arrival: an atomic that executes in the L2; poll: a relaxed AGENT-scope load (= an sc1 load: served by the L2, never by this CU's L1
            // -- a workgroup-scope "fetch_add 0" poll is folded into a plain load by the compiler and re-reads a stale L1 line: 130 us per
            // barrier in the first version of this kernel)
            __hip_atomic_fetch_add(c.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(c.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 1023) == 0 && (spins > 1000000 || __hip_atomic_load(c.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(c.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
        } else {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_fetch_add(c.sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (__hip_atomic_load(c.sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(2);
                if ((++spins & 1023) == 0 && (spins > 500000 || __hip_atomic_load(c.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
                    __hip_atomic_store(c.sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    break;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
    }
    __syncthreads();
    // (acquire side of the XCD-local path, ADVICE r05: an L1 invalidate -- buffer_inv sc1 -- per wave behind every barrier was built and
    //  measured in round 6: the launch went from 300 to 443 us, +0.14 ms per step, because the weight and LayerNorm-parameter lines the
    //  next stage re-reads went with it; removed.  What the path relies on instead, and why it holds: (1) every exchanged tensor is
    //  written exactly once per launch and FIRST read after the barrier that follows its producer, so a CU's vector L1 -- cold at kernel
    //  start -- cannot hold a line of it from before that barrier; four workgroups' 32-byte slices sharing one 128-byte line does not
    //  change this: the line is not in any reader's L1 until its first read, which is behind the barrier, when all four slices are in
    //  the shared L2 (stores drained by s_waitcnt vmcnt(0) in front of the arrival); (2) no two exchanged tensors share a cache line:
    //  each is its own allocation of the caching allocator (512-byte granularity); (3) the placement is not assumed: if the census
    //  finds two XCC ids the fenced path below runs.)
    ++c.phase;
}

__device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// acc[rt] += W[col0 + li][k0 .. k0 + KS*32) . X^T for the two 16-row activation tiles: weights straight from the row-major bf16 matrix
// (one 16-byte load per lane and k-step = the MFMA A fragment), activations from an LDS image (XL = true) or from global rows.
// Issued swapped (weights = A operand), so lane (li, g) ends with acc[rt][r] = out[row rt*16 + li][col0 + g*4 + r].
template <int KS>
__device__ __forceinline__ void wload(const bf16* __restrict__ W, long ldw, int col0, int k0, int li, int g, uint4 (&wf)[KS]) {
    const bf16* wp = W + (long)(col0 + li) * ldw + k0 + g * 8;
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) wf[kk] = *(const uint4*)(wp + kk * 32);
}
template <int KS, bool XL>
__device__ __forceinline__ void wmma(const uint4 (&wf)[KS], int k0, const bf16* X, long ldx, int rows_ok, int li, int g, f32x4 (&acc)[2]) {
    uint4 xf[2][KS];
    if constexpr (!XL) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int r = min(rt * 16 + li, rows_ok - 1);                 // rows beyond R read a valid row; their outputs are never stored
            const bf16* xp = X + (long)r * ldx + k0 + g * 8;
#pragma unroll
            for (int kk = 0; kk < KS; ++kk) xf[rt][kk] = *(const uint4*)(xp + kk * 32);
        }
    }
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const bf16x8 x = XL ? as_bf16x8(*(const uint4*)(X + (long)(rt * 16 + li) * ldx + k0 + kk * 32 + g * 8)) : as_bf16x8(xf[rt][kk]);
            acc[rt] = mma(as_bf16x8(wf[kk]), x, acc[rt]);
        }
    }
}
template <int KS, bool XL>
__device__ __forceinline__ void gemm_cols(const bf16* __restrict__ W, long ldw, int col0, int k0, const bf16* X, long ldx, int rows_ok,
                                          int li, int g, f32x4 (&acc)[2]) {
    uint4 wf[KS];
    wload<KS>(W, ldw, col0, k0, li, g, wf);
    wmma<KS, XL>(wf, k0, X, ldx, rows_ok, li, g, acc);
}

// out[r][col0 + g*4 .. +3] = bf16(f(acc + bias)) for the rows r < R this lane holds
template <bool RELU, bool DROP>
__device__ __forceinline__ void store_cols(const f32x4 (&acc)[2], const float* __restrict__ bias, bf16* __restrict__ out, long ldo, int N, int col0,
                                           int R, int li, int g, uint64_t seed, uint32_t thresh, float inv_keep) {
    const float4 bv = *(const float4*)(bias + col0 + g * 4);
    const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int r = rt * 16 + li;
        if (r >= R) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = acc[rt][e] + bb[e]; if (RELU) v[e] = fmaxf(v[e], 0.f); }
        if (DROP) {
            bool keep[4];
            dropout_keep_run<4>(seed, (uint64_t)r * N + col0 + g * 4, thresh, keep);       // the stream of gemm_nt's epilogue: element m * N + n
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = keep[e] ? v[e] * inv_keep : 0.f;
        }
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = f2bf(v[e]);
        *(uint2*)(out + (long)r * ldo + col0 + g * 4) = as_uint2(o);
    }
}

// y = LayerNorm(Dropout(x) + res) for rows [0, R), one wave per row (the arithmetic of layernorm_fwd_kernel<4>, norm.hip, expression for
// expression): x from global (written by the other workgroups before the last barrier), res from the LDS image `resl` (or none), result
// into the LDS image `dst` (bf16, what the launch chain would have stored and re-read); `save`: also to y / xhat / rstd in global.
template <bool DROP>
__device__ __forceinline__ void ln_rows(const bf16* __restrict__ x, const bf16* resl, const float* __restrict__ gamma, const float* __restrict__ beta,
                                        bf16* dst, int R, int save_j, bf16* __restrict__ y, long ldy, bf16* __restrict__ xhat, float* __restrict__ rstd_out,
                                        uint64_t seed, uint32_t thresh, float inv_keep, bool x_is_lds, bool only_mine = false) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane * 4;
    const float4 gq = *(const float4*)(gamma + col), bq = *(const float4*)(beta + col);
    const float gg[4] = {gq.x, gq.y, gq.z, gq.w}, bb[4] = {bq.x, bq.y, bq.z, bq.w};
    // the wave's rows (wave, wave + 4, ...: at most 8) are fetched TOGETHER: one row per trip exposed one L2 round trip per row (~0.8 us x 8)
    uint2 xin[MR / 4];
#pragma unroll
    for (int i = 0; i < MR / 4; ++i) {
        const int row = min(wave + 4 * i, R - 1);
        xin[i] = x_is_lds ? *(const uint2*)(x + (long)row * PX + col) : *(const uint2*)(x + (long)row * E + col);
    }
#pragma unroll
    for (int i = 0; i < MR / 4; ++i) {
        const int row = wave + 4 * i;
        if (row >= R) break;
        const bool save = (row & (G - 1)) == save_j;             // workgroup j stores rows j and j + 16
        if (only_mine && !save) continue;
        const bf16x4 a = as_bf16x4(xin[i]);
        bf16x4 r = bf16x4{};
        if (resl) r = as_bf16x4(*(const uint2*)(resl + (long)row * PX + col));
        bool keep[4] = {true, true, true, true};
        if (DROP) dropout_keep_run<4>(seed, (uint64_t)((long)row * E + col), thresh, keep);
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float xv = bf2f(a[e]);
            if (DROP) xv = keep[e] ? xv * inv_keep : 0.f;
            v[e] = xv + (resl ? bf2f(r[e]) : 0.f);
        }
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) s += v[e];
        const float mean = wave_sum(s) * (1.f / E);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = v[e] - mean; q += d * d; }
        const float rstd = rsqrtf(wave_sum(q) * (1.f / E) + LN_EPS);
        bf16x4 o, xh;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float hh = (v[e] - mean) * rstd;
            xh[e] = f2bf(hh);
            o[e] = f2bf(fmaf(hh, gg[e], bb[e]));
        }
        if (dst) *(uint2*)(dst + (long)row * PX + col) = as_uint2(o);
        if (save) {
            if (y) *(uint2*)(y + (long)row * ldy + col) = as_uint2(o);
            *(uint2*)(xhat + (long)row * E + col) = as_uint2(xh);
            if (lane == 0) rstd_out[row] = rstd;
        }
    }
}

__device__ __forceinline__ TokMap tmap(long ld, long sL, long s1) { TokMap m; m.ld = ld; m.sL = sL; m.s1 = s1; m.s2 = 0; m.B2 = 1; return m; }

template <bool DROP>
__global__ __launch_bounds__(NT, 1) void decoder_coop_fwd_kernel(DecCoopArgs a) {
    if (blockIdx.x & 7) return;                                   // the 16 workgroups the hardware places on XCD 0 do the work
    const int j = blockIdx.x >> 3;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    bf16* Xs = (bf16*)smem_raw;                                   // current decoder state tgt (bf16, as the chain stores it)
    bf16* XPs = Xs + MR * PX;                                     // bf16(tgt + query_pos): the q / k operand of the in-projections
    bf16* QPs = XPs + MR * PX;                                    // query_pos
    float* red = (float*)(QPs + MR * PX);                         // [4 waves][2 row tiles][64 lanes][4]: linear2's partial sums (8 KB)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, g = lane >> 4;
    const int R = a.R;
    Ctx c;
    c.sync = a.sync; c.phase = 0; c.same_xcd = false;

    // ---- images: tgt = 0 (transformer.py:60), query_pos ----
    for (int i = tid; i < MR * (E / 8); i += NT) {
        const int r = i / (E / 8), q8 = (i % (E / 8)) * 8;
        uint4 qv = make_uint4(0, 0, 0, 0);
        if (r < R) qv = *(const uint4*)(a.qpos + (long)r * E + q8);
        *(uint4*)(QPs + r * PX + q8) = qv;
        *(uint4*)(XPs + r * PX + q8) = qv;                        // tgt + query_pos with tgt = 0
        *(uint4*)(Xs + r * PX + q8) = make_uint4(0, 0, 0, 0);
    }
    // ---- placement census behind an agent-scope barrier: are the 16 workgroups on one XCD? ----
    if (tid == 0) __hip_atomic_fetch_or(a.sync + 1, 1u << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    coop_barrier(c);
    {
        const unsigned mask = __hip_atomic_load(a.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c.same_xcd = (mask & (mask - 1)) == 0;
    }

    const uint64_t seed0 = DROP ? (a.seed_ptr ? *a.seed_ptr : 0ull) * 0x9E3779B97F4A7C15ull : 0ull;
    const uint32_t th = (uint32_t)((double)a.pdrop * 4294967296.0);
    const float ik = dropout_inv_keep(a.pdrop);
    const int ub = j >> 3, uh = j & 7;                            // this workgroup's (clip, head) attention unit
    AttnArgs at{};
    at.kpm = nullptr; at.B = a.B; at.H = NH; at.Lq = a.Q; at.scale = 0.17677669529663687f;     // 32 ** -0.5
    at.pdrop = DROP ? a.pattn : 0.f; at.thresh = (uint32_t)((double)at.pdrop * 4294967296.0); at.seed_ptr = a.seed_ptr;

    uint4 win_next[8];                                            // this wave's in-projection weight tile of the coming layer
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) win_next[kk] = make_uint4(0, 0, 0, 0);
    for (int l = 0; l < a.nl; ++l) {
        const DLayer& W = a.L[l];
        // ---- A: packed self-attention in-projection: q | k rows see tgt + query_pos, v rows see tgt (tape.in_proj) ----
        if (wave < 3) {
            const int t = wave * G + j;                           // 48 column tiles of 16: tile t
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            if (l == 0) wload<8>(W.w_in, E, t * 16, 0, li, g, win_next);
            wmma<8, true>(win_next, 0, t < 32 ? XPs : Xs, PX, R, li, g, acc);
            store_cols<false, false>(acc, W.b_in, W.qkv, 3 * E, 3 * E, t * 16, R, li, g, 0, 0, 1.f);
        }
        coop_barrier(c);
        // ---- B: self-attention, one (clip, head) per workgroup (rows (b, q): row = b * Q + q) ----
        at.Q = W.qkv; at.mq = tmap(3 * E, 1, a.Q);
        at.K = W.qkv + E; at.mk = at.mq;
        at.V = W.qkv + 2 * E; at.mv = at.mq;
        at.O = W.o1; at.mo = tmap(E, 1, a.Q);
        at.lse = W.lse1; at.kpm = nullptr; at.Lk = a.Q; at.salt = W.salt_a1;
        attn_mfma_fwd_body<true>(at, 0, uh, ub);
        // (measured and rejected: fetching the weights of the next GEMM stage BEFORE the barrier that starts it -- the barrier's own
        // store drain then waits for them: out-proj 4.0 -> 3.3 us, but the attention stages + 1 us and linear1 / linear2 + 2.3 / + 3.8 us)
        coop_barrier(c);
        // ---- C: self-attention out-projection (column tile j; the operand rows come straight from global) ----
        if (wave == 0) {
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            gemm_cols<8, false>(W.w_o1, E, j * 16, 0, W.o1, E, R, li, g, acc);
            store_cols<false, false>(acc, W.b_o1, W.a1, E, E, j * 16, R, li, g, 0, 0, 1.f);
        }
        coop_barrier(c);
        // ---- D: norm1 (every workgroup, workgroup 0 saves) + cross-attention q-projection of tgt + query_pos ----
        uint4 wq[8];
        if (wave == 0) wload<8>(W.w_q, E, j * 16, 0, li, g, wq);           // the weight fetch runs under the LayerNorm
        ln_rows<DROP>(W.a1, Xs, W.g1, W.e1, Xs, R, j, W.y1, E, W.xh1, W.rs1, seed0 + W.salt_n1, th, ik, false);
        __syncthreads();
        for (int i = tid; i < MR * (E / 8); i += NT) {            // XPs = bf16(tgt + query_pos), as gemm_nt's A_ADD operand forms it
            const int r = i / (E / 8), q8 = (i % (E / 8)) * 8;
            const bf16x8 x = as_bf16x8(*(const uint4*)(Xs + r * PX + q8)), p = as_bf16x8(*(const uint4*)(QPs + r * PX + q8));
            bf16x8 y;
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = f2bf(bf2f(x[e]) + bf2f(p[e]));
            *(uint4*)(XPs + r * PX + q8) = as_uint4(y);
        }
        __syncthreads();
        if (wave == 0) {
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            wmma<8, true>(wq, 0, XPs, PX, R, li, g, acc);
            store_cols<false, false>(acc, W.b_q, W.q, E, E, j * 16, R, li, g, 0, 0, 1.f);
        }
        coop_barrier(c);
        // ---- E: cross-attention over the encoder memory (keys / values of this layer's packed projection) ----
        at.Q = W.q; at.mq = tmap(E, 1, a.Q);
        at.K = W.kv; at.mk = tmap(2 * E, 1, a.Lm);
        at.V = W.kv + E; at.mv = at.mk;
        at.O = W.o2; at.mo = tmap(E, 1, a.Q);
        at.lse = W.lse2; at.kpm = a.kpm; at.Lk = a.Lm; at.salt = W.salt_a2;
        attn_mfma_fwd_body<true>(at, 0, uh, ub);
        coop_barrier(c);
        // ---- F: cross-attention out-projection ----
        if (wave == 0) {
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            gemm_cols<8, false>(W.w_o2, E, j * 16, 0, W.o2, E, R, li, g, acc);
            store_cols<false, false>(acc, W.b_o2, W.a2, E, E, j * 16, R, li, g, 0, 0, 1.f);
        }
        coop_barrier(c);
        // ---- G: norm2 + FFN linear1 (ReLU, Dropout): hidden columns [j * 128, j * 128 + 128), two tiles per wave ----
        uint4 wf1[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u) wload<8>(W.w_f1, E, j * 128 + (wave * 2 + u) * 16, 0, li, g, wf1[u]);      // both tiles' weights under the LayerNorm
        ln_rows<DROP>(W.a2, Xs, W.g2, W.e2, Xs, R, j, W.y2, E, W.xh2, W.rs2, seed0 + W.salt_n2, th, ik, false);
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int col0 = j * 128 + (wave * 2 + u) * 16;
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            wmma<8, true>(wf1[u], 0, Xs, PX, R, li, g, acc);
            store_cols<true, DROP>(acc, W.b_f1, W.h, FF, FF, col0, R, li, g, seed0 + W.salt_f, th, ik);
        }
        coop_barrier(c);
        // ---- H: FFN linear2, column tile j; the four waves take K = 2048 in quarters and are summed in wave order ----
        {
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
            uint4 w2[2][8];
#pragma unroll
            for (int kq = 0; kq < 2; ++kq) wload<8>(W.w_f2, FF, j * 16, wave * 512 + kq * 256, li, g, w2[kq]);      // all 16 fragments in flight together
#pragma unroll
            for (int kq = 0; kq < 2; ++kq) wmma<8, false>(w2[kq], wave * 512 + kq * 256, W.h, FF, R, li, g, acc);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) *(f32x4*)(red + ((wave * 2 + rt) * 64 + lane) * 4) = acc[rt];
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int w = 0; w < 4; ++w) t += *(const f32x4*)(red + ((w * 2 + rt) * 64 + lane) * 4);
                    acc[rt] = t;
                }
                store_cols<false, false>(acc, W.b_f2, W.f2, E, E, j * 16, R, li, g, 0, 0, 1.f);
            }
        }
        coop_barrier(c);
        // ---- I: norm3 -> the next layer's tgt; decoder.norm of it -> hs[l] (workgroup 0 saves both) ----
        uint4 win[8];
        const bool more = l + 1 < a.nl;
        if (more && wave < 3) wload<8>(a.L[l + 1].w_in, E, (wave * G + j) * 16, 0, li, g, win);      // the next in-projection's weights under the LayerNorms
        ln_rows<DROP>(W.f2, Xs, W.g3, W.e3, Xs, R, j, W.y3, E, W.xh3, W.rs3, seed0 + W.salt_n3, th, ik, false);
        __syncthreads();
        // decoder.norm of rows j and j + 16 only (hs, xhat, rstd go to global; no workgroup needs the result)
        ln_rows<false>(Xs, nullptr, a.gN, a.eN, nullptr, R, j, a.hs + (long)l * R * E, E, W.xhN, W.rsN, 0, 0, 1.f, true, true);
        for (int i = tid; i < MR * (E / 8); i += NT) {
            const int r = i / (E / 8), q8 = (i % (E / 8)) * 8;
            const bf16x8 x = as_bf16x8(*(const uint4*)(Xs + r * PX + q8)), p = as_bf16x8(*(const uint4*)(QPs + r * PX + q8));
            bf16x8 y;
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = f2bf(bf2f(x[e]) + bf2f(p[e]));
            *(uint4*)(XPs + r * PX + q8) = as_uint4(y);
        }
        __syncthreads();
        if (more && wave < 3) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) win_next[kk] = win[kk];
        }
    }
    // ---- leave the synchronisation words zero for the next launch: a workgroup may still be re-reading the arrival counter of the last
    // barrier when the first ones are through, so departures are counted and the LAST workgroup out resets the words ----
    coop_barrier(c);
    // ---- fail safe: a barrier that timed out anywhere (a workgroup that was not co-resident: another kernel held its CU) leaves the error
    // word set, every later barrier falls through and what the stack computed is garbage.  Every workgroup that sees the word at its exit
    // overwrites the WHOLE output hs with NaN -- a workgroup cannot be past a barrier the others timed out at, and the laggard itself
    // checks here last -- so the heads, the loss and the gradient norm become NaN and the optimizer's non-finite guard (optim.hip:
    // clip coefficient -1) skips the update on the device; the host switches this engine to the launch chain when it next reads the
    // word (engine.ParamStore.coop_failed).
    if (__hip_atomic_load(a.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
        const uint4 nan4 = make_uint4(0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u, 0x7FC07FC0u);
        const long n8 = (long)a.nl * R * (E / 8);
        for (long i = tid; i < n8; i += NT) ((uint4*)a.hs)[i] = nan4;
    }
    if (tid == 0) {
        const unsigned left = __hip_atomic_fetch_add(a.sync + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (left == G - 1) {                                     // the last workgroup out resets the words (error word stays)
            __hip_atomic_store(a.sync + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(a.sync + 3, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

constexpr size_t kLds = (size_t)3 * MR * PX * sizeof(bf16) + 4 * 2 * 64 * 4 * sizeof(float);

}  // namespace

extern "C" {

// 1 when tuber_decoder_coop_fwd takes this decoder: 256-wide model, 8 heads, 2048-wide FFN, B * Q <= 32 rows, B * 8 == 16 attention
// units (a 2-clip batch), at most 6 layers
int tuber_decoder_coop_supported(int d_model, int nhead, int dim_ff, int batch, int num_queries, int num_layers) {
    return d_model == E && nhead == NH && dim_ff == FF && batch * NH == G && batch * num_queries <= MR && num_queries <= 16 && num_layers >= 1 && num_layers <= MAXL;
}
// pointers per layer in `layer_ptrs` (HOST array, 40 per layer, the order of DLayer's pointer members) and salts per layer in
// `layer_salts` (HOST array, 6 per layer: self-attention, norm1, cross-attention, norm2, FFN dropout, norm3)
int tuber_decoder_coop_ptrs_per_layer(void) { return 40; }

// The decoder stack forward (training or eval) as one cooperative launch.  sync: 4 zeroed 32-bit words in device memory (arrival counter,
// XCC census, error word, departure counter); a clean run leaves them zero, sync[2] != 0 afterwards = a barrier timed out: hs is all NaN then.
int tuber_decoder_coop_fwd(const void* const* layer_ptrs, const unsigned long long* layer_salts, int num_layers, const void* qpos,
                           const float* norm_weight, const float* norm_bias, void* hs, const void* kpm, int B, int Q, int Lm,
                           float pdrop, float pattn, const void* seed_ptr, void* sync, hipStream_t stream) {
    if (!layer_ptrs || !layer_salts || !qpos || !norm_weight || !norm_bias || !hs || !sync) return TUBER_EINVAL;
    if (!tuber_decoder_coop_supported(E, NH, FF, B, Q, num_layers) || Lm <= 0) return TUBER_EINVAL;
    if (pdrop < 0.f || pdrop >= 1.f || pattn < 0.f || pattn >= 1.f) return TUBER_EINVAL;
    DecCoopArgs a{};
    for (int l = 0; l < num_layers; ++l) {
        const void* const* p = layer_ptrs + 40 * l;
        for (int i = 0; i < 40; ++i) if (!p[i]) return TUBER_EINVAL;
        DLayer& L = a.L[l];
        int i = 0;
        L.w_in = (const bf16*)p[i++]; L.w_o1 = (const bf16*)p[i++]; L.w_q = (const bf16*)p[i++]; L.w_o2 = (const bf16*)p[i++];
        L.w_f1 = (const bf16*)p[i++]; L.w_f2 = (const bf16*)p[i++];
        L.b_in = (const float*)p[i++]; L.b_o1 = (const float*)p[i++]; L.b_q = (const float*)p[i++]; L.b_o2 = (const float*)p[i++];
        L.b_f1 = (const float*)p[i++]; L.b_f2 = (const float*)p[i++];
        L.g1 = (const float*)p[i++]; L.e1 = (const float*)p[i++]; L.g2 = (const float*)p[i++]; L.e2 = (const float*)p[i++];
        L.g3 = (const float*)p[i++]; L.e3 = (const float*)p[i++];
        L.kv = (const bf16*)p[i++];
        L.qkv = (bf16*)p[i++]; L.o1 = (bf16*)p[i++]; L.lse1 = (float*)p[i++]; L.a1 = (bf16*)p[i++];
        L.y1 = (bf16*)p[i++]; L.xh1 = (bf16*)p[i++]; L.rs1 = (float*)p[i++];
        L.q = (bf16*)p[i++]; L.o2 = (bf16*)p[i++]; L.lse2 = (float*)p[i++]; L.a2 = (bf16*)p[i++];
        L.y2 = (bf16*)p[i++]; L.xh2 = (bf16*)p[i++]; L.rs2 = (float*)p[i++];
        L.h = (bf16*)p[i++]; L.f2 = (bf16*)p[i++];
        L.y3 = (bf16*)p[i++]; L.xh3 = (bf16*)p[i++]; L.rs3 = (float*)p[i++];
        L.xhN = (bf16*)p[i++]; L.rsN = (float*)p[i++];
        const unsigned long long* s = layer_salts + 6 * l;
        L.salt_a1 = s[0]; L.salt_n1 = s[1]; L.salt_a2 = s[2]; L.salt_n2 = s[3]; L.salt_f = s[4]; L.salt_n3 = s[5];
    }
    a.nl = num_layers; a.B = B; a.Q = Q; a.Lm = Lm; a.R = B * Q;
    a.qpos = (const bf16*)qpos; a.gN = norm_weight; a.eN = norm_bias; a.hs = (bf16*)hs; a.kpm = (const uint8_t*)kpm;
    a.pdrop = pdrop; a.pattn = pattn; a.seed_ptr = (const uint64_t*)seed_ptr; a.sync = (unsigned*)sync;
    const bool drop = pdrop > 0.f || pattn > 0.f;
    dim3 grid(G * 8), block(NT);
    static LdsOptIn opt[2];                 // static (attention staging) + dynamic LDS exceed 64 KB together
    TUBER_LDS_OPT_IN(opt[0], decoder_coop_fwd_kernel<true>, kLds);
    TUBER_LDS_OPT_IN(opt[1], decoder_coop_fwd_kernel<false>, kLds);
    if (drop) hipLaunchKernelGGL(decoder_coop_fwd_kernel<true>, grid, block, kLds, stream, a);
    else hipLaunchKernelGGL(decoder_coop_fwd_kernel<false>, grid, block, kLds, stream, a);
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
