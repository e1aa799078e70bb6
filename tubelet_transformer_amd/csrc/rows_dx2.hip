// The two data gradients of a packed attention in-projection with the positional embedding folded in, as ONE launch -- gfx950.
// reference: autograd of q = k = with_pos_embed(tgt, query_pos); self_attn(q, k, tgt) / multihead_attn(with_pos_embed(tgt, query_pos), ...)
// (models/transformer/transformer.py:150-159,215-240): the projection y = [(x + pos) W_qk^T | x W_v^T] hands back
//     dx   = g . W            over all N columns of g   (+ the gradient x already holds)
//     dpos = g[:, :Nq] . W[:Nq]   over the q / k columns only
// tape.py: in_proj.bwd launched them as two tuber_gemm_nt calls of 4 workgroups each (30 decoder rows: ~5 + ~8 us, launch-bound).
// dpos is a PREFIX of dx's reduction: one pass over g and W, two results.
#include "common.h"

namespace {

constexpr int RB = 32;                     // rows per workgroup (two 16-row MFMA tiles)

struct Dx2Args {
    const bf16* g; long ldg; int M, N, Nq;        // g [M][ldg], reduction length N, prefix Nq (multiples of 32)
    const bf16* wt; long ldt; int Kin;            // W^T rows [Kin][ldt >= N]
    bf16* dx; const bf16* res; bf16* dpos;        // [M][Kin] each; res may be NULL
};

__device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void store_tile(const f32x4 (&acc)[2], bf16* out, const bf16* res, long ld, int r0, int rows, int col, int li, int g) {
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) {
        const int r = rt * 16 + li;
        if (r >= rows) continue;
        const long off = (long)(r0 + r) * ld + col + g * 4;
        bf16x4 o;
        if (res) {
            const bf16x4 sv = as_bf16x4(*(const uint2*)(res + off));
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[rt][e] + bf2f(sv[e]));
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = f2bf(acc[rt][e]);
        }
        *(uint2*)(out + off) = as_uint2(o);
    }
}

// grid (row blocks of 32, Kin / 16): a workgroup owns 16 output columns; its four waves take consecutive spans of the reduction (256 elements
// each up to N = 1024, 512 up to 2048, ...), so the 30 x 768 in-projection runs on 16 workgroups x 3 waves instead of 4 x 1 (12.5 -> ~6 us) and
// linear1's data gradient (N = 2048) on 16 x 4.  Weights = MFMA A operand (one 16-byte load per lane and k-step straight from the row-major
// W^T), activations = B operand from global rows (L2-resident); partial accumulators meet in LDS, wave 0 adds them in wave order: the waves
// whose span lies below Nq give dpos, all of them dx.  Nq is a multiple of the span (host check).
__global__ __launch_bounds__(256) void rows_dx2_kernel(Dx2Args a) {
    __shared__ f32x4 red[4][2][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int r0 = blockIdx.x * RB, rows = min(RB, a.M - r0);
    const int col0 = blockIdx.y * 16;
    const int span = 256 * ((a.N + 1023) / 1024);
    const int kb = w * span, ke = min(a.N, kb + span);
    const bf16* wp = a.wt + (long)(col0 + li) * a.ldt + g * 8;
    const bf16* xp[2];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) xp[rt] = a.g + (long)(r0 + min(rt * 16 + li, rows - 1)) * a.ldg + g * 8;
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    // chunks of 256 reduction elements: 8 weight + 16 activation fragments in flight per lane
    for (int k0 = kb; k0 < ke; k0 += 256) {
        uint4 wf[8], xf[2][8];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const bool ok = k0 + kk * 32 < ke;
            wf[kk] = ok ? *(const uint4*)(wp + k0 + kk * 32) : make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) xf[rt][kk] = ok ? *(const uint4*)(xp[rt] + k0 + kk * 32) : make_uint4(0, 0, 0, 0);
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) acc[rt] = mma(as_bf16x8(wf[kk]), as_bf16x8(xf[rt][kk]), acc[rt]);
    }
    red[w][0][lane] = acc[0]; red[w][1][lane] = acc[1];
    __syncthreads();
    if (w) return;
    f32x4 tot[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}}, pre[2] = {tot[0], tot[1]};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i * span >= a.N) break;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) tot[rt] += red[i][rt][lane];
        if ((i + 1) * span == a.Nq || ((i + 1) * span > a.N && a.Nq == a.N)) { pre[0] = tot[0]; pre[1] = tot[1]; }
    }
    if (a.dpos) store_tile(pre, a.dpos, nullptr, a.Kin, r0, rows, col0, li, g);
    store_tile(tot, a.dx, a.res, a.Kin, r0, rows, col0, li, g);
}

}  // namespace

extern "C" {

// dx[M][Kin] = g[M][:N] . W (+ res);  dpos[M][Kin] = g[M][:Nq] . W[:Nq]  (bare; NULL: not wanted) -- W^T given as rows wt[Kin][ldt].
// N a multiple of 32, Nq = N or a multiple of the per-wave span (256 up to N = 1024, 512 up to 2048), Kin a multiple of 16.  Meant for the few-row case
// (the DETR decoder's 30 query rows); with dpos = NULL it is a plain few-row data-gradient GEMM (linear1's, N = 2048).
int tuber_rows_dx2(const void* g, long ldg, int M, int N, int Nq, const void* wt, long ldt, int Kin, void* dx, const void* res, void* dpos,
                   hipStream_t stream) {
    const int span = 256 * ((N + 1023) / 1024);
    if (M <= 0 || N <= 0 || (N & 31) || N > 4 * span || Nq <= 0 || Nq > N || (Nq != N && (Nq % span)) || Kin <= 0 || (Kin & 15) || ldg < N || (ldg & 7) ||
        ldt < N || (ldt & 7) || !dx)
        return TUBER_EINVAL;
    Dx2Args a;
    a.g = (const bf16*)g; a.ldg = ldg; a.M = M; a.N = N; a.Nq = Nq;
    a.wt = (const bf16*)wt; a.ldt = ldt; a.Kin = Kin;
    a.dx = (bf16*)dx; a.res = (const bf16*)res; a.dpos = (bf16*)dpos;
    hipLaunchKernelGGL(rows_dx2_kernel, dim3(ceil_div(M, RB), Kin / 16), dim3(256), 0, stream, a);
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
