// fp32 kernels of the EVAL precision mode (round 6): the DETR decoder stack and the box / actor heads in fp32.
// reference: TransformerDecoder / TransformerDecoderLayer.forward_post (models/transformer/transformer.py:99-128,218-249), the heads of
// DETR.forward (models/tuber_ava.py:121-125,142), MLP (models/criterion.py:485-497).
//
// Why: eval has no batch statistics and is not the throughput metric, and the published numbers of the reference (mAP 29.7 / 31.1) live
// there.  The bf16 path's error on the actor logits is carried in roughly equal parts by the body's convs, the decoder's linears and the
// heads (measured on the oracle with selective rounding); the decoder works on <= 640 rows x 256 -- a few MFLOP per
// layer -- so it runs in fp32 outright under model.eval(): plain FMA kernels, fp32 master weights straight from the flat parameter
// buffer, fp32 softmax, the residual stream never rounded.  The 704-row memory-side projections are the only part with real work
// (185 MFLOP per layer).  Training keeps the bf16 MFMA path (tuber_decoder_coop_fwd / the launch chain).
#include "common.h"

namespace {

constexpr int LT_M = 32, LT_N = 64, LT_K = 32;

// y[m][n] = act( sum_k (x[m][k] + (n < add_cols ? add[m][k] : 0)) * W[n][k] + bias[n] ),  act: 0 none, 1 ReLU, 2 sigmoid.
// Few-row, long-K layers (the decoder's linear2: 30 rows, K = 2048, 4 column tiles) would be one latency chain of K / 32 fetch -> park ->
// FMA rounds on 4 workgroups: the k range is split over blockIdx.z (raw partial tiles to `part`, summed in slab order by
// linear_f32_reduce_kernel -- deterministic) and the next k-chunk's operands are fetched into registers under the current chunk's FMAs.
__global__ __launch_bounds__(256) void linear_f32_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ add, long ldadd, int add_cols,
                                                         const float* __restrict__ W, long ldw, const float* __restrict__ bias,
                                                         float* __restrict__ y, long ldy, float* __restrict__ part, int M, int N, int K, int kslab, int act) {
    __shared__ float As[LT_M][LT_K + 1];
    __shared__ float Ws[LT_K][LT_N + 1];
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int m0 = blockIdx.x * LT_M, n0 = blockIdx.y * LT_N;
    const int kb = blockIdx.z * kslab, ke = min(K, kb + kslab);
    const bool with_add = add != nullptr && n0 < add_cols;          // add_cols is a multiple of the column tile (checked by the launcher)
    float acc[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) acc[r] = 0.f;
    const int ar = tid >> 3, k4 = (tid & 7) * 4;
    const float* xp = x + (long)min(m0 + ar, M - 1) * ldx + k4;
    const float* ap = with_add ? add + (long)min(m0 + ar, M - 1) * ldadd + k4 : nullptr;
    const float* wp0 = W + (long)min(n0 + ar, N - 1) * ldw + k4;
    const float* wp1 = W + (long)min(n0 + ar + 32, N - 1) * ldw + k4;
    float4 ra, rw0, rw1;
    auto fetch = [&](int k0) {
        ra = *(const float4*)(xp + k0);
        if (with_add) { const float4 a = *(const float4*)(ap + k0); ra.x += a.x; ra.y += a.y; ra.z += a.z; ra.w += a.w; }
        rw0 = *(const float4*)(wp0 + k0);
        rw1 = *(const float4*)(wp1 + k0);
    };
    fetch(kb);
    for (int k0 = kb; k0 < ke; k0 += LT_K) {
        As[ar][k4] = ra.x; As[ar][k4 + 1] = ra.y; As[ar][k4 + 2] = ra.z; As[ar][k4 + 3] = ra.w;
        Ws[k4][ar] = rw0.x; Ws[k4 + 1][ar] = rw0.y; Ws[k4 + 2][ar] = rw0.z; Ws[k4 + 3][ar] = rw0.w;
        Ws[k4][ar + 32] = rw1.x; Ws[k4 + 1][ar + 32] = rw1.y; Ws[k4 + 2][ar + 32] = rw1.z; Ws[k4 + 3][ar + 32] = rw1.w;
        __syncthreads();
        if (k0 + LT_K < ke) fetch(k0 + LT_K);
#pragma unroll
        for (int kk = 0; kk < LT_K; ++kk) {
            const float w = Ws[kk][tx];
#pragma unroll
            for (int r = 0; r < 8; ++r) acc[r] = fmaf(As[ty * 8 + r][kk], w, acc[r]);
        }
        __syncthreads();
    }
    const int n = n0 + tx;
    if (n >= N) return;
    if (gridDim.z > 1) {                                   // raw partial tile of this k slab
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int m = m0 + ty * 8 + r;
            if (m < M) part[((long)blockIdx.z * M + m) * N + n] = acc[r];
        }
        return;
    }
    const float b = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int m = m0 + ty * 8 + r;
        if (m >= M) continue;
        float v = acc[r] + b;
        if (act == 1) v = fmaxf(v, 0.f);
        else if (act == 2) v = 1.f / (1.f + expf(-v));
        y[(long)m * ldy + n] = v;
    }
}

__global__ __launch_bounds__(256) void linear_f32_reduce_kernel(const float* __restrict__ part, const float* __restrict__ bias, float* __restrict__ y, long ldy,
                                                                int M, int N, int S, int act) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)M * N) return;
    const int m = (int)(i / N), n = (int)(i % N);
    float v = 0.f;
    for (int s = 0; s < S; ++s) v += part[(long)s * M * N + i];
    v += bias ? bias[n] : 0.f;
    if (act == 1) v = fmaxf(v, 0.f);
    else if (act == 2) v = 1.f / (1.f + expf(-v));
    y[(long)m * ldy + n] = v;
}

// one wave per (clip b, head h, query i): o[b*Lq + i][h*D + d] = sum_j softmax_j(scale * q . k_j) v_j[d], keys masked by kpm[b][j] != 0.
// q rows b*Lq + i, k / v rows b*Lk + j; D = 32.
__global__ __launch_bounds__(64) void attention_f32_kernel(const float* __restrict__ q, long ldq, const float* __restrict__ k, long ldk,
                                                           const float* __restrict__ v, long ldv, float* __restrict__ o, long ldo,
                                                           const uint8_t* __restrict__ kpm, int H, int Lq, int Lk, float scale) {
    constexpr int D = 32;
    extern __shared__ float sc[];            // [Lk] scores, then probabilities
    const int lane = threadIdx.x;
    int u = blockIdx.x;
    const int i = u % Lq; u /= Lq;
    const int h = u % H; const int b = u / H;
    const float* qp = q + (long)(b * Lq + i) * ldq + h * D;
    float qr[D];
#pragma unroll
    for (int d = 0; d < D; d += 4) { const float4 t = *(const float4*)(qp + d); qr[d] = t.x; qr[d + 1] = t.y; qr[d + 2] = t.z; qr[d + 3] = t.w; }
    float mx = -INFINITY;
    for (int j = lane; j < Lk; j += 64) {
        const float* kp = k + (long)(b * Lk + j) * ldk + h * D;
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < D; d += 4) { const float4 t = *(const float4*)(kp + d); s = fmaf(qr[d], t.x, s); s = fmaf(qr[d + 1], t.y, s); s = fmaf(qr[d + 2], t.z, s); s = fmaf(qr[d + 3], t.w, s); }
        s *= scale;
        if (kpm && kpm[(long)b * Lk + j]) s = -INFINITY;
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float l = 0.f;
    for (int j = lane; j < Lk; j += 64) { const float p = expf(sc[j] - mx); sc[j] = p; l += p; }
    l = wave_sum(l);
    __syncthreads();
    const int d = lane & 31, half = lane >> 5;
    float acc = 0.f;
    for (int j = half; j < Lk; j += 2) acc = fmaf(sc[j], v[(long)(b * Lk + j) * ldv + h * D + d], acc);
    acc += __shfl_xor(acc, 32);
    if (half == 0) o[(long)(b * Lq + i) * ldo + h * D + d] = acc / l;
}

}  // namespace

extern "C" {

// k slabs tuber_linear_f32 splits (M, N, K) into: few-row layers with a long reduction are spread over more workgroups (each slab is a
// multiple of 256); 1 = no split.  The caller's workspace must hold slabs * M * N floats when this is > 1.
int tuber_linear_f32_slabs(int M, int N, int K) {
    const long tiles = (long)ceil_div(M, LT_M) * ceil_div(N, LT_N);
    if (tiles >= 64 || K < 512) return 1;
    int S = K / 256;
    while (S > 1 && tiles * S > 256) S >>= 1;
    return S < 1 ? 1 : S;
}

// fp32 linear layer y = act((x [+ add for the output columns < add_cols]) . W^T + bias): x [M, K] (ldx), add [M, K] or NULL (with_pos_embed:
// the q / k rows of a packed in-projection see x + pos, the v rows x), W [N, K] (ldw) and bias [N] fp32 -- the master parameters --,
// y [M, N] (ldy).  act: 0 none, 1 ReLU, 2 sigmoid.  K % 32 == 0, add_cols % 64 == 0, 16-byte aligned rows.  workspace: tuber_linear_f32_slabs(M, N, K)
// * M * N floats (may be NULL when that is 1).
int tuber_linear_f32(const float* x, long ldx, const float* add, long ldadd, int add_cols, const float* W, long ldw, const float* bias,
                     float* y, long ldy, int M, int N, int K, int act, float* workspace, hipStream_t stream) {
    if (!x || !W || !y || M <= 0 || N <= 0 || K <= 0 || (K & 31) || (ldx & 3) || (ldw & 3) || (add && ((ldadd & 3) || (add_cols & 63))) || act < 0 || act > 2)
        return TUBER_EINVAL;
    const int S = workspace ? tuber_linear_f32_slabs(M, N, K) : 1;
    const int kslab = ceil_div(ceil_div(K, S), LT_K) * LT_K;
    hipLaunchKernelGGL(linear_f32_kernel, dim3(ceil_div(M, LT_M), ceil_div(N, LT_N), S), dim3(256), 0, stream, x, ldx, add, ldadd, add ? add_cols : 0,
                       W, ldw, bias, y, ldy, workspace, M, N, K, kslab, act);
    if (S > 1)
        hipLaunchKernelGGL(linear_f32_reduce_kernel, dim3(ceil_div((long)M * N, 256)), dim3(256), 0, stream, workspace, bias, y, ldy, M, N, S, act);
    TUBER_RETURN_LAUNCH();
}

// fp32 multi-head attention core, head dimension 32 (nn.MultiheadAttention of the DETR decoder, transformer.py:218-240): q rows (b, i),
// k / v rows (b, j), heads side by side in the columns; kpm [B][Lk] bytes (non-zero = padded key) or NULL; o rows (b, i).
int tuber_attention_f32(const float* q, long ldq, const float* k, long ldk, const float* v, long ldv, float* o, long ldo, const void* kpm,
                        int B, int H, int Lq, int Lk, float scale, hipStream_t stream) {
    if (!q || !k || !v || !o || B <= 0 || H <= 0 || Lq <= 0 || Lk <= 0 || Lk > 12288 || (ldq & 3) || (ldk & 3) || (ldv & 3)) return TUBER_EINVAL;
    hipLaunchKernelGGL(attention_f32_kernel, dim3(B * H * Lq), dim3(64), (size_t)Lk * sizeof(float), stream, q, ldq, k, ldk, v, ldv, o, ldo,
                       (const uint8_t*)kpm, H, Lq, Lk, scale);
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
