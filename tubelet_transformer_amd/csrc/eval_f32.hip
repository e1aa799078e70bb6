// fp32 kernels of the EVAL precision mode (round 6): the DETR decoder stack and the box / actor heads in fp32.
// reference: TransformerDecoder / TransformerDecoderLayer.forward_post (models/transformer/transformer.py:99-128,218-249), the heads of
// DETR.forward (models/tuber_ava.py:121-125,142), MLP (models/criterion.py:485-497).
//
// Why: eval has no batch statistics and is not the throughput metric, and the published numbers of the reference (mAP 29.7 / 31.1) live
// there.  The bf16 path's error on the actor logits is carried in roughly equal parts by the body's convs, the decoder's linears and the
// heads (measured on the oracle with selective rounding); the decoder works on <= 640 rows x 256 -- a few MFLOP per
// layer -- so it runs in fp32 outright under model.eval(): fp32 MFMA / FMA kernels, fp32 master weights straight from the flat parameter
// buffer, fp32 softmax, the residual stream never rounded.  The 704-row memory-side projections are the only part with real work
// (185 MFLOP per layer).  Training keeps the bf16 MFMA path (tuber_decoder_coop_fwd / the launch chain).
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

// y[m][n] = act( sum_k (x[m][k] + (n < add_cols ? add[m][k] : 0)) * W[n][k] + bias[n] ),  act: 0 none, 1 ReLU, 2 sigmoid -- on the fp32
// matrix pipe (v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32: true fp32 multiply-adds), operand fragments straight from global memory:
// a lane holds 4 consecutive k of one row (one 16-byte load) and feeds them to 4 MFMAs -- which k a lane group contributes to which MFMA
// does not matter as long as both operands agree.  blockIdx.z walks a batch of weight sets (the six layers' memory projections).
struct LinArgs {
    const float* x; long ldx; const float* add; long ldadd; int add_cols;
    const float* W; long ldw; const float* bias; float* y; long ldy;
    int M, N, K, act; long wz, bz, yz;
};

__device__ __forceinline__ float lin_act(float v, int act) {
    if (act == 1) return fmaxf(v, 0.f);
    if (act == 2) return 1.f / (1.f + expf(-v));
    return v;
}

constexpr int LIN_PF = 8;          // k steps of operand loads in flight per wave

// FEW ROWS (the decoder's 30, the heads' 180): a workgroup owns ONE 16 x 16 output tile and its four waves take every fourth 16-wide k
// chunk -- the first version (32 x 64 LDS tiles, FMA from broadcast reads) ran the decoder's 30-row layers as one latency chain of K / 32
// fetch -> park -> FMA rounds on 4 workgroups (~16 us per layer call); here a 30 x 256 x 256 layer is 32 workgroups of 4 + 4 loads per lane.
// The four partial tiles meet in LDS and are summed in wave order (deterministic).
__global__ __launch_bounds__(256) void linear_f32_rows_kernel(LinArgs a) {
    __shared__ float red[4][256];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = blockIdx.x * 16, n0 = blockIdx.y * 16;
    const int r = lane & 15, kq = lane >> 4;
    const bool with_add = a.add != nullptr && n0 < a.add_cols;
    const float* xp = a.x + (long)min(m0 + r, a.M - 1) * a.ldx + 4 * kq;
    const float* ap = with_add ? a.add + (long)min(m0 + r, a.M - 1) * a.ldadd + 4 * kq : nullptr;
    const float* wp = a.W + blockIdx.z * a.wz + (long)min(n0 + r, a.N - 1) * a.ldw + 4 * kq;
    const int nchunks = a.K >> 4;
    const int ns = nchunks > w ? (nchunks - w + 3) >> 2 : 0;          // chunks w, w + 4, ...
    float4 xa[LIN_PF], wb[LIN_PF];
    auto ld = [&](int s, float4& xv, float4& wv) {
        const long k0 = 16L * min(4 * s + w, nchunks - 1);
        xv = *(const float4*)(xp + k0);
        if (with_add) { const float4 t = *(const float4*)(ap + k0); xv.x += t.x; xv.y += t.y; xv.z += t.z; xv.w += t.w; }
        wv = *(const float4*)(wp + k0);
    };
#pragma unroll
    for (int u = 0; u < LIN_PF; ++u) ld(u, xa[u], wb[u]);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < ns; s0 += LIN_PF) {
#pragma unroll
        for (int u = 0; u < LIN_PF; ++u) {
            if (s0 + u < ns) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u].x, wb[u].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u].y, wb[u].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u].z, wb[u].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[u].w, wb[u].w, acc, 0, 0, 0);
            }
            if (s0 + u + LIN_PF < ns) ld(s0 + u + LIN_PF, xa[u], wb[u]);
        }
    }
    // D[i][j]: j = lane % 16, i = 4 * (lane / 16) + v
#pragma unroll
    for (int v = 0; v < 4; ++v) red[w][(4 * kq + v) * 16 + r] = acc[v];
    __syncthreads();
    const int m = m0 + (tid >> 4), n = n0 + (tid & 15);
    if (m < a.M && n < a.N) {
        float v = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
        if (a.bias) v += a.bias[blockIdx.z * a.bz + n];
        a.y[blockIdx.z * a.yz + (long)m * a.ldy + n] = lin_act(v, a.act);
    }
}

// MANY ROWS (the 704-row memory projections): 64 x 64 per workgroup, a wave owns a 32 x 32 block over the whole reduction.
__global__ __launch_bounds__(256) void linear_f32_tile_kernel(LinArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = blockIdx.x * 64 + (w >> 1) * 32, n0 = blockIdx.y * 64 + (w & 1) * 32;
    if (m0 >= a.M || n0 >= a.N) return;
    const int r = lane & 31, kh = lane >> 5;
    const bool with_add = a.add != nullptr && n0 < a.add_cols;
    const float* xp = a.x + (long)min(m0 + r, a.M - 1) * a.ldx + 4 * kh;
    const float* ap = with_add ? a.add + (long)min(m0 + r, a.M - 1) * a.ldadd + 4 * kh : nullptr;
    const float* wp = a.W + blockIdx.z * a.wz + (long)min(n0 + r, a.N - 1) * a.ldw + 4 * kh;
    const int ns = a.K >> 3;
    float4 xa[LIN_PF], wb[LIN_PF];
    auto ld = [&](int s, float4& xv, float4& wv) {
        const long k0 = 8L * min(s, ns - 1);
        xv = *(const float4*)(xp + k0);
        if (with_add) { const float4 t = *(const float4*)(ap + k0); xv.x += t.x; xv.y += t.y; xv.z += t.z; xv.w += t.w; }
        wv = *(const float4*)(wp + k0);
    };
#pragma unroll
    for (int u = 0; u < LIN_PF; ++u) ld(u, xa[u], wb[u]);
    f32x16 acc;
#pragma unroll
    for (int v = 0; v < 16; ++v) acc[v] = 0.f;
    for (int s0 = 0; s0 < ns; s0 += LIN_PF) {
#pragma unroll
        for (int u = 0; u < LIN_PF; ++u) {
            if (s0 + u < ns) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[u].x, wb[u].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[u].y, wb[u].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[u].z, wb[u].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[u].w, wb[u].w, acc, 0, 0, 0);
            }
            if (s0 + u + LIN_PF < ns) ld(s0 + u + LIN_PF, xa[u], wb[u]);
        }
    }
    // D[i][j]: j = lane % 32, i = 8 * (v / 4) + 4 * (lane / 32) + v % 4
    const int n = n0 + r;
    if (n >= a.N) return;
    const float b = a.bias ? a.bias[blockIdx.z * a.bz + n] : 0.f;
    float* yp = a.y + blockIdx.z * a.yz + n;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const int m = m0 + 8 * (v >> 2) + 4 * kh + (v & 3);
        if (m < a.M) yp[(long)m * a.ldy] = lin_act(acc[v] + b, a.act);
    }
}

// one workgroup per (clip b, head h, chunk of <= QCAP queries): o[b*Lq + i][h*D + d] = sum_j softmax_j(scale * q_i . k_j) v_j[d], keys masked by
// kpm[b][j] != 0.  q rows b*Lq + i, k / v rows b*Lk + j; D = 32.  The queries of the chunk share every K / V row: thread t scores key
// j = t, t + 256, ... against all of them (its K row in registers, the queries as LDS broadcasts) into the LDS score table [QCAP][Lk]; a wave per
// query takes max / exp / sum; then thread (key slice t / 32, d = t % 32) accumulates P . V for ALL queries over every eighth key -- 8 value rows
// in flight per thread, 64 per workgroup, the first batch (like the first K row) fetched before anything else -- and the eight partial sums meet in LDS
// in slice order.  History: one wave per (b, h, query) re-read the same 90 KB of K / V fifteen times and walked the keys as one dependent chain;
// the first shared form kept the V walk per query group (44 exposed round trips: 69 us per cross-attention call at 15 x 352).
template <int QCAP>
__global__ __launch_bounds__(256) void attention_f32_kernel(const float* __restrict__ q, long ldq, const float* __restrict__ k, long ldk,
                                                            const float* __restrict__ v, long ldv, float* __restrict__ o, long ldo,
                                                            const uint8_t* __restrict__ kpm, int H, int LqAll, int Lk, float scale) {
    constexpr int D = 32;
    extern __shared__ __attribute__((aligned(16))) float sm[];     // [QCAP][D] queries (pre-scaled), [QCAP] 1 / l, [8][QCAP][D] partial outputs, [QCAP][Lk] scores -> probabilities
    float* qs = sm;
    float* linv = sm + QCAP * D;
    float* red = linv + QCAP;
    float* sc = red + 8 * QCAP * D;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int nqc = (LqAll + QCAP - 1) / QCAP;
    int u0 = blockIdx.x;
    const int i0 = (u0 % nqc) * QCAP; u0 /= nqc;
    const int h = u0 % H, b = u0 / H;
    const int Lq = min(QCAP, LqAll - i0);
    q += (long)(b * LqAll + i0) * ldq;
    o += (long)(b * LqAll + i0) * ldo;
    const float* kb = k + (long)b * Lk * ldk + h * D;
    const int d = tid & 31, ks = tid >> 5;
    const float* vp = v + (long)b * Lk * ldv + h * D + d;
    // everything that does not depend on another thread is requested first: this thread's first K row and first batch of V rows
    float kr[D], vv[8];
    {
        const float* kp = kb + (long)min(tid, Lk - 1) * ldk;
#pragma unroll
        for (int e = 0; e < D; e += 4) { const float4 t = *(const float4*)(kp + e); kr[e] = t.x; kr[e + 1] = t.y; kr[e + 2] = t.z; kr[e + 3] = t.w; }
#pragma unroll
        for (int t = 0; t < 8; ++t) vv[t] = vp[(long)min(ks + 8 * t, Lk - 1) * ldv];
    }
    for (int e = tid; e < QCAP * D; e += 256) qs[e] = e < Lq * D ? q[(long)(e / D) * ldq + h * D + (e % D)] * scale : 0.f;
    __syncthreads();
    for (int j = tid; j < Lk; j += 256) {
        if (j != tid) {
            const float* kp = kb + (long)j * ldk;
#pragma unroll
            for (int e = 0; e < D; e += 4) { const float4 t = *(const float4*)(kp + e); kr[e] = t.x; kr[e + 1] = t.y; kr[e + 2] = t.z; kr[e + 3] = t.w; }
        }
        const bool masked = kpm && kpm[(long)b * Lk + j];
        for (int i = 0; i < Lq; ++i) {
            const float4* q4 = (const float4*)(qs + i * D);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < D; e += 4) {
                const float4 t = q4[e >> 2];
                s = fmaf(t.x, kr[e], s); s = fmaf(t.y, kr[e + 1], s); s = fmaf(t.z, kr[e + 2], s); s = fmaf(t.w, kr[e + 3], s);
            }
            sc[i * Lk + j] = masked ? -INFINITY : s;
        }
    }
    __syncthreads();
    for (int i = w; i < QCAP; i += 4) {
        float* row = sc + i * Lk;
        if (i >= Lq) {                                 // unused rows of the chunk: zero probabilities, so the P . V walk needs no bounds on i
            for (int j = lane; j < Lk; j += 64) row[j] = 0.f;
            continue;
        }
        float mx = -INFINITY;
        for (int j = lane; j < Lk; j += 64) mx = fmaxf(mx, row[j]);
        mx = wave_max(mx);
        float l = 0.f;
        for (int j = lane; j < Lk; j += 64) { const float p = expf(row[j] - mx); row[j] = p; l += p; }
        l = wave_sum(l);
        if (lane == 0) linv[i] = 1.f / l;
    }
    __syncthreads();
    float acc[QCAP];
#pragma unroll
    for (int i = 0; i < QCAP; ++i) acc[i] = 0.f;
    for (int j0 = ks; j0 < Lk; j0 += 64) {             // keys j0, j0 + 8, ..., j0 + 56 of this slice
        float cur[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) cur[t] = vv[t];
        if (j0 + 64 < Lk) {
#pragma unroll
            for (int t = 0; t < 8; ++t) vv[t] = vp[(long)min(j0 + 64 + 8 * t, Lk - 1) * ldv];
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int j = j0 + 8 * t;
            if (j < Lk) {
#pragma unroll
                for (int i = 0; i < QCAP; ++i) acc[i] = fmaf(sc[i * Lk + j], cur[t], acc[i]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < QCAP; ++i) red[(ks * QCAP + i) * D + d] = acc[i];
    __syncthreads();
    for (int e = tid; e < Lq * D; e += 256) {
        float r = 0.f;
#pragma unroll
        for (int sl = 0; sl < 8; ++sl) r += red[sl * QCAP * D + e];
        o[(long)(e / D) * ldo + h * D + (e % D)] = r * linv[e / D];
    }
}

}  // namespace

extern "C" {

static int linear_f32_launch(const float* x, long ldx, const float* add, long ldadd, int add_cols, const float* W, long ldw, const float* bias,
                             float* y, long ldy, int M, int N, int K, int act, int nbatch, long wz, long bz, long yz, hipStream_t stream) {
    if (!x || !W || !y || M <= 0 || N <= 0 || K <= 0 || (K & 31) || (ldx & 3) || (ldw & 3) || (wz & 3) || (add && ((ldadd & 3) || (add_cols & 63))) || act < 0 || act > 2 ||
        nbatch <= 0 || nbatch > 65535)
        return TUBER_EINVAL;
    LinArgs a{x, ldx, add, ldadd, add ? add_cols : 0, W, ldw, bias, y, ldy, M, N, K, act, wz, bz, yz};
    if (M <= 256)
        hipLaunchKernelGGL(linear_f32_rows_kernel, dim3(ceil_div(M, 16), ceil_div(N, 16), nbatch), dim3(256), 0, stream, a);
    else
        hipLaunchKernelGGL(linear_f32_tile_kernel, dim3(ceil_div(M, 64), ceil_div(N, 64), nbatch), dim3(256), 0, stream, a);
    TUBER_RETURN_LAUNCH();
}

// fp32 linear layer y = act((x [+ add for the output columns < add_cols]) . W^T + bias): x [M, K] (ldx), add [M, K] or NULL (with_pos_embed:
// the q / k rows of a packed in-projection see x + pos, the v rows x), W [N, K] (ldw) and bias [N] fp32 -- the master parameters --,
// y [M, N] (ldy).  act: 0 none, 1 ReLU, 2 sigmoid.  K % 32 == 0, add_cols % 64 == 0, 16-byte aligned rows.
int tuber_linear_f32(const float* x, long ldx, const float* add, long ldadd, int add_cols, const float* W, long ldw, const float* bias,
                     float* y, long ldy, int M, int N, int K, int act, hipStream_t stream) {
    return linear_f32_launch(x, ldx, add, ldadd, add_cols, W, ldw, bias, y, ldy, M, N, K, act, 1, 0, 0, 0, stream);
}

// the same layer for `nbatch` weight sets over ONE input in one launch: set z reads W + z * w_stride, bias + z * bias_stride and writes
// y + z * y_stride (strides in elements) -- the memory-side K / V projections of all decoder layers, which do not depend on the decoder state.
int tuber_linear_f32_batched(const float* x, long ldx, const float* add, long ldadd, int add_cols, const float* W, long ldw, const float* bias,
                             float* y, long ldy, int M, int N, int K, int act, int nbatch, long w_stride, long bias_stride, long y_stride,
                             hipStream_t stream) {
    return linear_f32_launch(x, ldx, add, ldadd, add_cols, W, ldw, bias, y, ldy, M, N, K, act, nbatch, w_stride, bias_stride, y_stride, stream);
}

// fp32 multi-head attention core, head dimension 32 (nn.MultiheadAttention of the DETR decoder, transformer.py:218-240): q rows (b, i),
// k / v rows (b, j), heads side by side in the columns; kpm [B][Lk] bytes (non-zero = padded key) or NULL; o rows (b, i).
int tuber_attention_f32(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, float* o, long ldo, const void* kpm,
                        int B, int H, int Lq, int Lk, float scale, hipStream_t stream) {
    if (!q || !k || !v || !o || B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0 || (ldq & 3) || (ldk & 3) || (ldv & 3)) return TUBER_EINVAL;
    // queries per workgroup: 32 when there are more than 16 and their score rows fit into the 160 KB of LDS beside the fixed part, else 16
    auto lds_bytes = [&](int qcap) { return (size_t)(qcap * 32 + qcap + 8 * qcap * 32 + (long)qcap * Lk) * sizeof(float); };
    const int QCAP = Lq > 16 && lds_bytes(32) <= 160 * 1024 ? 32 : 16;
    const size_t lds = lds_bytes(QCAP);
    if (lds > 160 * 1024) return TUBER_EINVAL;                      // Lk > ~2 250 keys: not a shape of this model
    if (lds > 48 * 1024) {                                          // one-time opt-in per kernel and device (the library's pattern: common.h)
        static LdsOptIn opt[2];
        if (QCAP == 32) TUBER_LDS_OPT_IN(opt[1], attention_f32_kernel<32>, 160 * 1024);
        else TUBER_LDS_OPT_IN(opt[0], attention_f32_kernel<16>, 160 * 1024);
    }
    const dim3 grid(B * H * ceil_div(Lq, QCAP));
    if (QCAP == 32)
        hipLaunchKernelGGL(attention_f32_kernel<32>, grid, dim3(256), lds, stream, q, ldq, k, ldk, v, ldv, o, ldo, (const uint8_t*)kpm, H, Lq, Lk, scale);
    else
        hipLaunchKernelGGL(attention_f32_kernel<16>, grid, dim3(256), lds, stream, q, ldq, k, ldk, v, ldv, o, ldo, (const uint8_t*)kpm, H, Lq, Lk, scale);
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
