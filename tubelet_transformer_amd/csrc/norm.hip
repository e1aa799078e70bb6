// BatchNorm3d (training + eval) split into the pieces that fuse into neighbouring kernels,
// the bottleneck residual join, and LayerNorm -- gfx950, HBM-bound, 16-byte vector access.
//
// Training-mode BatchNorm (reference: nn.BatchNorm3d(eps=1e-3, momentum=0.1),
// models/backbones/ir_CSN_152.py:15-16,46,56,64,119,154) is a two-phase scheme:
//   producer conv epilogue  -> per-tile partial (sum, sum of squares) per channel
//   bn_finalize             -> mean / var / running-stat update / scale = g*invstd, shift = b - mean*scale
//   consumer prologue       -> relu(x*scale + shift)       (gemm.hip A_BN_RELU, dwconv.hip)
// Backward mirrors it: partial (sum dz, sum dz*x) -> bn_bwd_finalize -> dx = A*dz + B*x + C.
#include "common.h"

// ---------------------------------------------------------------------------------------------
// finalize kernels: one block per 32 channels, 1024 threads = 32 row-groups x 32 channels
// ---------------------------------------------------------------------------------------------
// column sums of the partial-statistics rows st0/st1 [R][C] for channels c0..c0+31 (NTH threads): thread = (row group
// of NTH / 8, channel quad); 16-byte loads, 4 row iterations in flight, double accumulation; the result for channel c0+cl is
// returned in threads 0..31 (cl = threadIdx.x).  NTH = 256 for the short lists (R <= 128: every list of layer3 / layer4 and every
// first-stage-reduced one): device timestamps showed the 1024-thread workgroups of this tiny kernel starting up to 1.0 us apart
// (16 waves each to dispatch) for a 2.6 us life -- with 4 waves per workgroup the stagger is a quarter of that.
template <int NTH>
__device__ __forceinline__ void bn_partial_sums(const float* __restrict__ st0, const float* __restrict__ st1, int R, int C, int c0,
                                                double (&red)[2][32][33], double& a_out, double& b_out) {
    constexpr int RG = NTH / 8, NW = NTH / 64;
    const int qd = threadIdx.x & 7, rq = threadIdx.x >> 3;       // 8 quads x RG row groups
    const int cq = c0 + qd * 4;
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
    if (cq < C) {                                                // C % 4 == 0
        int r = rq;
        for (; r + 3 * RG < R; r += 4 * RG) {
            float4 x[4], y[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                x[u] = *(const float4*)(st0 + (long)(r + RG * u) * C + cq);
                y[u] = *(const float4*)(st1 + (long)(r + RG * u) * C + cq);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[0] += x[u].x; a[1] += x[u].y; a[2] += x[u].z; a[3] += x[u].w;
                b[0] += y[u].x; b[1] += y[u].y; b[2] += y[u].z; b[3] += y[u].w;
            }
        }
        {   // tail: up to three more rows, issued together
            float4 x[3], y[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const bool ok = r + RG * u < R;
                const long ro = (long)(ok ? r + RG * u : 0) * C + cq;
                x[u] = *(const float4*)(st0 + ro);
                y[u] = *(const float4*)(st1 + ro);
                if (!ok) { x[u] = make_float4(0.f, 0.f, 0.f, 0.f); y[u] = make_float4(0.f, 0.f, 0.f, 0.f); }
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                a[0] += x[u].x; a[1] += x[u].y; a[2] += x[u].z; a[3] += x[u].w;
                b[0] += y[u].x; b[1] += y[u].y; b[2] += y[u].z; b[3] += y[u].w;
            }
        }
    }
    // RG row groups -> NW (one per wave) via shuffles over the lanes sharing qd (lane bits 3..5), then LDS
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        a[e] = xor32_sum(xor16_sum(a[e] + dpp_f64<DPP_ROR8>(a[e])));       // (the fp64 overloads: the float ones would narrow silently)
        b[e] = xor32_sum(xor16_sum(b[e] + dpp_f64<DPP_ROR8>(b[e])));
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane < 8) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[0][wave][lane * 4 + e] = a[e]; red[1][wave][lane * 4 + e] = b[e]; }
    }
    __syncthreads();
    a_out = 0.0; b_out = 0.0;
    if (threadIdx.x < 32) {
#pragma unroll
        for (int i = 0; i < NW; ++i) { a_out += red[0][i][threadIdx.x]; b_out += red[1][i][threadIdx.x]; }
    }
}

template <int NTH>
__global__ __launch_bounds__(NTH) void bn_finalize_kernel(
    const float* __restrict__ st0, const float* __restrict__ st1, int R, int C, float count,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    float* __restrict__ rmean, float* __restrict__ rvar, long long* __restrict__ nbt,
    float momentum, float eps,
    float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out, float* __restrict__ invstd_out) {
    __shared__ double red[2][32][33];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    // per-channel parameters are fetched before the partial-row sweep so their latency hides behind it
    const bool fin = threadIdx.x < 32 && c < C;
    const float gam = fin ? gamma[c] : 0.f, bet = fin ? beta[c] : 0.f;
    const float rm0 = (fin && rmean) ? rmean[c] : 0.f, rv0 = (fin && rmean) ? rvar[c] : 0.f;
    double a, b;
    bn_partial_sums<NTH>(st0, st1, R, C, blockIdx.x * 32, red, a, b);
    if (rg == 0 && c < C) {
        const double mean = a / (double)count;
        double var = b / (double)count - mean * mean;
        if (var < 0.0) var = 0.0;
        const float invstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gam * invstd;
        scale[c] = sc;
        shift[c] = bet - (float)mean * sc;
        mean_out[c] = (float)mean;
        invstd_out[c] = invstd;
        if (rmean) {
            const double unbiased = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
            rmean[c] = (1.f - momentum) * rm0 + momentum * (float)mean;
            rvar[c] = (1.f - momentum) * rv0 + momentum * (float)unbiased;
        }
    }
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
}

__global__ void bn_eval_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ rmean, const float* __restrict__ rvar, float eps,
                                      float* __restrict__ scale, float* __restrict__ shift, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float sc = gamma[c] / sqrtf(rvar[c] + eps);
    scale[c] = sc;
    shift[c] = beta[c] - rmean[c] * sc;
}

// the same for EVERY BatchNorm of a network in one launch (eval forward: 155 launches of the kernel above were 0.6 ms of a 5.5 ms forward):
// blockIdx.y walks a device table of 8-word rows {gamma, beta, running_mean, running_var, scale, shift, C, unused}.
struct BnAffineRow { const float* gamma; const float* beta; const float* rmean; const float* rvar; float* scale; float* shift; long C; long pad; };
__global__ void bn_eval_affine_multi_kernel(const BnAffineRow* __restrict__ table, float eps) {
    const BnAffineRow r = table[blockIdx.y];
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= r.C) return;
    const float sc = r.gamma[c] / sqrtf(r.rvar[c] + eps);
    r.scale[c] = sc;
    r.shift[c] = r.beta[c] - r.rmean[c] * sc;
}

// dL/dx = A*dz + B*x + C per channel;  dgamma = sum dz*xhat, dbeta = sum dz
template <int NTH>
__global__ __launch_bounds__(NTH) void bn_bwd_finalize_kernel(
    const float* __restrict__ st0, const float* __restrict__ st1, int R, int C, float count,
    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
    float* __restrict__ cA, float* __restrict__ cB, float* __restrict__ cC,
    float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
    __shared__ double red[2][32][33];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    const bool fin = threadIdx.x < 32 && c < C;
    const float mu_f = fin ? mean[c] : 0.f, r_f = fin ? invstd[c] : 0.f, g_f = fin ? gamma[c] : 0.f;
    const float dg0 = (fin && dgamma && accumulate) ? dgamma[c] : 0.f, db0 = (fin && dgamma && accumulate) ? dbeta[c] : 0.f;
    double a, b;
    bn_partial_sums<NTH>(st0, st1, R, C, blockIdx.x * 32, red, a, b);
    if (rg == 0 && c < C) {
        const double mu = mu_f, r = r_f, g = g_f;
        const double sum_dz = a, sum_dz_xhat = (b - mu * a) * r;
        const double m1 = sum_dz / count, m2 = sum_dz_xhat / count;
        cA[c] = (float)(g * r);
        cB[c] = (float)(-g * r * r * m2);
        cC[c] = (float)(g * r * r * m2 * mu - g * r * m1);
        if (dgamma) {
            dgamma[c] = dg0 + (float)sum_dz_xhat;
            dbeta[c] = db0 + (float)sum_dz;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// row-major [M, C] elementwise kernels: thread = 8 channels (16 B); TPR = C/8 threads per row
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void load8f(const float* p, float (&v)[8]) {
    const float4 a = ((const float4*)p)[0], b = ((const float4*)p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// y = relu( c4*s4+h4 + (rs ? res*rs+rh : res) )     [block output; ir_CSN_152.py:84-90]
__global__ __launch_bounds__(256) void block_out_fwd_kernel(
    const bf16* __restrict__ c4, const float* __restrict__ s4, const float* __restrict__ h4,
    const bf16* __restrict__ res, const float* __restrict__ rs, const float* __restrict__ rh,
    bf16* __restrict__ y, uint8_t* __restrict__ ymask, long M, int C) {
    const int tpr = C >> 3;
    const int cg = threadIdx.x % tpr;
    const long rpp = 256 / tpr;
    float a4[8], b4[8], ar[8], br[8];
    load8f(s4 + cg * 8, a4); load8f(h4 + cg * 8, b4);
    if (rs) { load8f(rs + cg * 8, ar); load8f(rh + cg * 8, br); }
    for (long row = (long)blockIdx.x * rpp + threadIdx.x / tpr; row < M; row += (long)gridDim.x * rpp) {
        const long off = row * C + cg * 8;
        const bf16x8 c = as_bf16x8(*(const uint4*)(c4 + off));
        const bf16x8 r = as_bf16x8(*(const uint4*)(res + off));
        bf16x8 o;
        unsigned bits = 0;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float rv = bf2f(r[e]);
            if (rs) rv = fmaf(rv, ar[e], br[e]);
            o[e] = f2bf(fmaxf(fmaf(bf2f(c[e]), a4[e], b4[e]) + rv, 0.f));
            bits |= (bf2f(o[e]) > 0.f ? 1u : 0u) << e;
        }
        *(uint4*)(y + off) = as_uint4(o);
        // ReLU mask of the block output, one bit per element ([M][C / 8] bytes, bit e of byte (m, c / 8) = y[m][c + e] > 0): what the join
        // backward of this block reads instead of the 16 x larger y (tuber_gemm_nt_join_mask, round 6)
        if (ymask) ymask[row * tpr + cg] = (uint8_t)bits;
    }
}

// the same join for the EVAL precision mode (round 6): the residual stream stays fp32 -- res32 = the previous block's fp32 output (identity
// blocks) or the BatchNorm of the bf16 projection-shortcut conv output (res / rs / rh) -- and the block output is written twice: y32 (fp32,
// the next block's residual input) and y (bf16, the operand of the next block's conv1 / projection GEMMs).  That is where the bf16-rounded
// execution of the oracle (tests/parity_util.py) rounds: conv operands and results, never the stream between the blocks.
__global__ __launch_bounds__(256) void block_out_fwd_f32_kernel(
    const bf16* __restrict__ c4, const float* __restrict__ s4, const float* __restrict__ h4,
    const bf16* __restrict__ res, const float* __restrict__ rs, const float* __restrict__ rh, const float* __restrict__ res32,
    bf16* __restrict__ y, float* __restrict__ y32, long M, int C) {
    const int tpr = C >> 3;
    const int cg = threadIdx.x % tpr;
    const long rpp = 256 / tpr;
    float a4[8], b4[8], ar[8], br[8];
    load8f(s4 + cg * 8, a4); load8f(h4 + cg * 8, b4);
    if (rs) { load8f(rs + cg * 8, ar); load8f(rh + cg * 8, br); }
    for (long row = (long)blockIdx.x * rpp + threadIdx.x / tpr; row < M; row += (long)gridDim.x * rpp) {
        const long off = row * C + cg * 8;
        const bf16x8 c = as_bf16x8(*(const uint4*)(c4 + off));
        float rv[8];
        if (res32) {
            load8f(res32 + off, rv);
        } else {
            const bf16x8 r = as_bf16x8(*(const uint4*)(res + off));
#pragma unroll
            for (int e = 0; e < 8; ++e) rv[e] = rs ? fmaf(bf2f(r[e]), ar[e], br[e]) : bf2f(r[e]);
        }
        bf16x8 o;
        float of[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            of[e] = fmaxf(fmaf(bf2f(c[e]), a4[e], b4[e]) + rv[e], 0.f);
            o[e] = f2bf(of[e]);
        }
        *(uint4*)(y + off) = as_uint4(o);
        ((float4*)(y32 + off))[0] = make_float4(of[0], of[1], of[2], of[3]);
        ((float4*)(y32 + off))[1] = make_float4(of[4], of[5], of[6], of[7]);
    }
}

// dz = dy * [y > 0]  (bf16 out) ; partial stats per block: sum dz, sum dz*c4, (sum dz*cds)
__global__ __launch_bounds__(256) void block_out_bwd_kernel(
    const bf16* __restrict__ dy, const bf16* __restrict__ y, const bf16* __restrict__ c4, const bf16* __restrict__ cds,
    bf16* __restrict__ dz, float* __restrict__ st_dz, float* __restrict__ st_c4, float* __restrict__ st_ds,
    long M, int C, long rows_per_block) {
    __shared__ float red[3][256][8 + 1];
    const int tpr = C >> 3;
    const int cg = threadIdx.x % tpr, rs = threadIdx.x / tpr;
    const int rpp = 256 / tpr;
    float s0[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; s2[e] = 0.f; }
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    for (long row = r0 + rs; row < r1; row += rpp) {
        const long off = row * C + cg * 8;
        const bf16x8 g = as_bf16x8(*(const uint4*)(dy + off));
        const bf16x8 yy = as_bf16x8(*(const uint4*)(y + off));
        const bf16x8 c = as_bf16x8(*(const uint4*)(c4 + off));
        bf16x8 d = bf16x8{};
        if (cds) d = as_bf16x8(*(const uint4*)(cds + off));
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = bf2f(yy[e]) > 0.f ? bf2f(g[e]) : 0.f;
            o[e] = f2bf(v);
            s0[e] += v; s1[e] += v * bf2f(c[e]);
            if (cds) s2[e] += v * bf2f(d[e]);
        }
        *(uint4*)(dz + off) = as_uint4(o);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][threadIdx.x][e] = s0[e]; red[1][threadIdx.x][e] = s1[e]; red[2][threadIdx.x][e] = s2[e]; }
    __syncthreads();
    // thread t < C reduces channel t over the rpp row slots
    for (int ch = threadIdx.x; ch < C; ch += 256) {
        const int g8 = ch >> 3, e = ch & 7;
        float a = 0.f, b = 0.f, c = 0.f;
        for (int s = 0; s < rpp; ++s) { const int t = s * tpr + g8; a += red[0][t][e]; b += red[1][t][e]; c += red[2][t][e]; }
        st_dz[(long)blockIdx.x * C + ch] = a;
        st_c4[(long)blockIdx.x * C + ch] = b;
        if (st_ds) st_ds[(long)blockIdx.x * C + ch] = c;
    }
}

// generic "relu(bn(x))-masked gradient + stats":  dz = g * [x*sc+sh > 0]; partial sum dz, sum dz*x
// (used for the stem BN; the bottleneck BNs get this fused into gemm / dwconv epilogues)
__global__ __launch_bounds__(256) void relu_bn_bwd_reduce_kernel(
    const bf16* __restrict__ g, const bf16* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
    bf16* __restrict__ dz, float* __restrict__ st0, float* __restrict__ st1, long M, int C, long rows_per_block) {
    __shared__ float red[2][256][8 + 1];
    const int tpr = C >> 3;
    const int cg = threadIdx.x % tpr, rs = threadIdx.x / tpr;
    const int rpp = 256 / tpr;
    float a8[8], b8[8], s0[8], s1[8];
    load8f(sc + cg * 8, a8); load8f(sh + cg * 8, b8);
#pragma unroll
    for (int e = 0; e < 8; ++e) { s0[e] = 0.f; s1[e] = 0.f; }
    const long r0 = (long)blockIdx.x * rows_per_block;
    const long r1 = min(M, r0 + rows_per_block);
    for (long row = r0 + rs; row < r1; row += rpp) {
        const long off = row * C + cg * 8;
        const bf16x8 gg = as_bf16x8(*(const uint4*)(g + off));
        const bf16x8 xx = as_bf16x8(*(const uint4*)(x + off));
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xv = bf2f(xx[e]);
            const float v = fmaf(xv, a8[e], b8[e]) > 0.f ? bf2f(gg[e]) : 0.f;
            o[e] = f2bf(v);
            s0[e] += v; s1[e] += v * xv;
        }
        *(uint4*)(dz + off) = as_uint4(o);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][threadIdx.x][e] = s0[e]; red[1][threadIdx.x][e] = s1[e]; }
    __syncthreads();
    for (int ch = threadIdx.x; ch < C; ch += 256) {
        const int g8 = ch >> 3, e = ch & 7;
        float a = 0.f, b = 0.f;
        for (int s = 0; s < rpp; ++s) { const int t = s * tpr + g8; a += red[0][t][e]; b += red[1][t][e]; }
        st0[(long)blockIdx.x * C + ch] = a;
        st1[(long)blockIdx.x * C + ch] = b;
    }
}

// dx = A*dz + B*x + C   (BatchNorm backward apply)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(
    const bf16* __restrict__ dz, const bf16* __restrict__ x, const float* __restrict__ cA, const float* __restrict__ cB,
    const float* __restrict__ cC, bf16* __restrict__ dx, long M, int C) {
    const int tpr = C >> 3;
    const int cg = threadIdx.x % tpr;
    const long rpp = 256 / tpr;
    float a[8], b[8], c[8];
    load8f(cA + cg * 8, a); load8f(cB + cg * 8, b); load8f(cC + cg * 8, c);
    for (long row = (long)blockIdx.x * rpp + threadIdx.x / tpr; row < M; row += (long)gridDim.x * rpp) {
        const long off = row * C + cg * 8;
        const bf16x8 d = as_bf16x8(*(const uint4*)(dz + off));
        const bf16x8 xx = as_bf16x8(*(const uint4*)(x + off));
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(fmaf(a[e], bf2f(d[e]), fmaf(b[e], bf2f(xx[e]), c[e])));
        *(uint4*)(dx + off) = as_uint4(o);
    }
}


// BatchNorm backward for SHORT partial-statistics lists (layer3 / layer4: R <= 128 rows): finalize + apply in ONE launch.
// A dependent launch costs ~5 us + a ~1.7 us gap on this GPU whatever it computes, and the finalize kernel is exactly that; here
// every workgroup derives the coefficients of ITS 128-channel strip itself -- R x 128 x 2 floats from L2, one dependent round
// trip, 8 row groups of float4 loads all in flight, fp64 sums -- and goes straight on to dx = cA*dz + cB*x + cC over its
// 128 rows x 128 channels (256-byte row segments).  Row chunk 0 of each strip also writes dgamma / dbeta (+=).  The summation order over
// the partial rows is fixed (deterministic) but differs from bn_bwd_finalize's, so results agree with the two-launch path to fp64
// rounding, not bit for bit.  (Round 1's one-launch variant used 32-channel strips = 64-byte segments and 1024-thread workgroups and
// lost 0.5 ms/step.)
// Rows per workgroup = 16 * KR, chosen per launch (fa_rows_per_thread) so that the grid sits at or just under a multiple of the 256 CUs:
// layer3's bn4 (M = 5632, C = 1024) is 44 x 8 = 352 workgroups at 128 rows -- the launch runs at the pace of the CUs that carry two --
// and 32 x 8 = 256 at 176 rows.
#define FA_CS 128          // channels per strip
template <int KR>
__global__ __launch_bounds__(256) void bn_bwd_fa_kernel(
    const float* __restrict__ st0, const float* __restrict__ st1, int R, int C, float count,
    const float* __restrict__ gamma, const float* __restrict__ mean, const float* __restrict__ invstd,
    float* __restrict__ dgamma, float* __restrict__ dbeta,
    const bf16* __restrict__ dz, const bf16* __restrict__ x, bf16* __restrict__ dx, long M) {
    constexpr int FA_ROWS = 16 * KR;
    __shared__ double red[2][8][FA_CS];
    __shared__ float coef[3][FA_CS];
    const int c0 = blockIdx.y * FA_CS;
    const int tid = threadIdx.x;
    // ---- the apply phase's operands go out FIRST (thread = row slot of 16 x 8-channel group of 16; FA_ROWS / 16 = 8 rows per thread):
    // their latency then runs under the derive phase instead of after it (one exposed memory round trip per workgroup, not two)
    const int cg = tid & 15, rs = tid >> 4;
    const long r0 = (long)blockIdx.x * FA_ROWS;
    const long r1 = min(M, r0 + FA_ROWS);
    uint4 pd[FA_ROWS / 16], px[FA_ROWS / 16];
#pragma unroll
    for (int k = 0; k < FA_ROWS / 16; ++k) {
        const long row = r0 + rs + 16 * k;
        const long off = min(row, M - 1) * C + c0 + cg * 8;
        pd[k] = *(const uint4*)(dz + off);
        px[k] = *(const uint4*)(x + off);
    }
    // ---- derive: thread = (row group rg of 8, channel quad q of 32) ----
    {
        const int q = tid & 31, rg = tid >> 5;
        double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
        const float* p0 = st0 + c0 + q * 4;
        const float* p1 = st1 + c0 + q * 4;
        int r = rg;
        for (; r + 24 < R; r += 32) {
            float4 u[4], v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { u[k] = *(const float4*)(p0 + (long)(r + 8 * k) * C); v[k] = *(const float4*)(p1 + (long)(r + 8 * k) * C); }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a[0] += u[k].x; a[1] += u[k].y; a[2] += u[k].z; a[3] += u[k].w;
                b[0] += v[k].x; b[1] += v[k].y; b[2] += v[k].z; b[3] += v[k].w;
            }
        }
        for (; r < R; r += 8) {
            const float4 u = *(const float4*)(p0 + (long)r * C), v = *(const float4*)(p1 + (long)r * C);
            a[0] += u.x; a[1] += u.y; a[2] += u.z; a[3] += u.w;
            b[0] += v.x; b[1] += v.y; b[2] += v.z; b[3] += v.w;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[0][rg][q * 4 + e] = a[e]; red[1][rg][q * 4 + e] = b[e]; }
    }
    __syncthreads();
    if (tid < FA_CS) {
        double a = 0.0, b = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { a += red[0][k][tid]; b += red[1][k][tid]; }
        const int c = c0 + tid;
        const double mu = mean[c], rr = invstd[c], g = gamma[c];
        const double sum_dz = a, sum_dz_xhat = (b - mu * a) * rr;
        const double m1 = sum_dz / count, m2 = sum_dz_xhat / count;
        coef[0][tid] = (float)(g * rr);
        coef[1][tid] = (float)(-g * rr * rr * m2);
        coef[2][tid] = (float)(g * rr * rr * m2 * mu - g * rr * m1);
        if (blockIdx.x == 0 && dgamma) {
            dgamma[c] += (float)sum_dz_xhat;
            dbeta[c] += (float)sum_dz;
        }
    }
    __syncthreads();
    // ---- apply ----
    float ca[8], cb[8], cc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ca[e] = coef[0][cg * 8 + e]; cb[e] = coef[1][cg * 8 + e]; cc[e] = coef[2][cg * 8 + e]; }
#pragma unroll
    for (int k = 0; k < FA_ROWS / 16; ++k) {
        const long row = r0 + rs + 16 * k;
        if (row >= r1) break;
        const bf16x8 d = as_bf16x8(pd[k]), xx = as_bf16x8(px[k]);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(fmaf(ca[e], bf2f(d[e]), fmaf(cb[e], bf2f(xx[e]), cc[e])));
        *(uint4*)(dx + row * C + c0 + cg * 8) = as_uint4(o);
    }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm(x + res) over the last dim E (256 or 2048), eps 1e-5, one wave per row.
// reference: nn.LayerNorm in models/transformer/transformer.py:163-167,229-247 (post-norm).
// ---------------------------------------------------------------------------------------------
template <int EPL>   // elements per lane = E / 64 (4 for E=256, 32 for E=2048)
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(
    const bf16* __restrict__ x, const bf16* __restrict__ res, const float* __restrict__ gamma, const float* __restrict__ beta,
    bf16* __restrict__ y, long ldy, bf16* __restrict__ xhat_out, float* __restrict__ rstd_out, int M, float eps,
    uint32_t thresh, float inv_keep, const uint64_t* __restrict__ seed_ptr, uint64_t salt) {
    constexpr int E = EPL * 64;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float v[EPL];
    const long base = (long)row * E;
    const uint64_t seed = thresh ? (seed_ptr ? *seed_ptr : 0ull) * 0x9E3779B97F4A7C15ull + salt : 0ull;
#pragma unroll
    for (int i = 0; i < EPL / 4; ++i) {
        const int col = (i * 64 + lane) * 4;
        const bf16x4 a = as_bf16x4(*(const uint2*)(x + base + col));
        bf16x4 r = bf16x4{};
        if (res) r = as_bf16x4(*(const uint2*)(res + base + col));
        bool keep[4] = {true, true, true, true};
        if (thresh) dropout_keep_run<4>(seed, (uint64_t)(base + col), thresh, keep);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float xv = bf2f(a[e]);
            if (thresh) xv = keep[e] ? xv * inv_keep : 0.f;   // Dropout(x) + res
            v[i * 4 + e] = xv + (res ? bf2f(r[e]) : 0.f);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) s += v[i];
    const float mean = wave_sum(s) * (1.f / E);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) { const float d = v[i] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) * (1.f / E) + eps);
#pragma unroll
    for (int i = 0; i < EPL / 4; ++i) {
        const int col = (i * 64 + lane) * 4;
        const float4 g = *(const float4*)(gamma + col), b = *(const float4*)(beta + col);
        const float gg[4] = {g.x, g.y, g.z, g.w}, bb[4] = {b.x, b.y, b.z, b.w};
        bf16x4 o, xh;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float h = (v[i * 4 + e] - mean) * rstd;
            xh[e] = f2bf(h);
            o[e] = f2bf(fmaf(h, gg[e], bb[e]));
        }
        *(uint2*)(y + (long)row * ldy + col) = as_uint2(o);
        if (xhat_out) *(uint2*)(xhat_out + base + col) = as_uint2(xh);
    }
    if (rstd_out && lane == 0) rstd_out[row] = rstd;
}

// eval precision mode (round 6): y = LayerNorm(x + res) with the residual STREAM in fp32 -- x32 / res32 (fp32 [M, E]) take precedence over
// the bf16 x / res when given -- and the result written as bf16 (y, leading dimension ldy: the operand of the next GEMM) and, when y32 is
// not NULL, as fp32 [M, E] (the next LayerNorm's residual input).  No dropout, nothing saved for a backward.
template <int EPL>
__global__ __launch_bounds__(256) void layernorm_fwd_f32_kernel(
    const bf16* __restrict__ x, const float* __restrict__ x32, const bf16* __restrict__ res, const float* __restrict__ res32,
    const float* __restrict__ gamma, const float* __restrict__ beta, bf16* __restrict__ y, long ldy, float* __restrict__ y32, int M, float eps) {
    constexpr int E = EPL * 64;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float v[EPL];
    const long base = (long)row * E;
#pragma unroll
    for (int i = 0; i < EPL / 4; ++i) {
        const int col = (i * 64 + lane) * 4;
        float xv[4], rv[4] = {0.f, 0.f, 0.f, 0.f};
        if (x32) { const float4 t = *(const float4*)(x32 + base + col); xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w; }
        else { const bf16x4 a = as_bf16x4(*(const uint2*)(x + base + col)); xv[0] = bf2f(a[0]); xv[1] = bf2f(a[1]); xv[2] = bf2f(a[2]); xv[3] = bf2f(a[3]); }
        if (res32) { const float4 t = *(const float4*)(res32 + base + col); rv[0] = t.x; rv[1] = t.y; rv[2] = t.z; rv[3] = t.w; }
        else if (res) { const bf16x4 r = as_bf16x4(*(const uint2*)(res + base + col)); rv[0] = bf2f(r[0]); rv[1] = bf2f(r[1]); rv[2] = bf2f(r[2]); rv[3] = bf2f(r[3]); }
#pragma unroll
        for (int e = 0; e < 4; ++e) v[i * 4 + e] = xv[e] + rv[e];
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) s += v[i];
    const float mean = wave_sum(s) * (1.f / E);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < EPL; ++i) { const float d = v[i] - mean; q += d * d; }
    const float rstd = rsqrtf(wave_sum(q) * (1.f / E) + eps);
#pragma unroll
    for (int i = 0; i < EPL / 4; ++i) {
        const int col = (i * 64 + lane) * 4;
        const float4 g = *(const float4*)(gamma + col), b = *(const float4*)(beta + col);
        const float gg[4] = {g.x, g.y, g.z, g.w}, bb[4] = {b.x, b.y, b.z, b.w};
        bf16x4 o;
        float of[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            of[e] = fmaf((v[i * 4 + e] - mean) * rstd, gg[e], bb[e]);
            o[e] = f2bf(of[e]);
        }
        *(uint2*)(y + (long)row * ldy + col) = as_uint2(o);
        if (y32) *(float4*)(y32 + base + col) = make_float4(of[0], of[1], of[2], of[3]);
    }
}

// dx = rstd * (g*dy - mean(g*dy) - xhat*mean(g*dy*xhat)); partial dgamma = sum dy*xhat, dbeta = sum dy
// over the rows of each block (blocks write [gridDim.x][E] partials).
template <int EPL>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(
    const bf16* __restrict__ dy, long lddy, const bf16* __restrict__ xhat, const float* __restrict__ rstd, const float* __restrict__ gamma,
    bf16* __restrict__ dx, bf16* __restrict__ dxd, float* __restrict__ part, int M, int rows_per_block,
    uint32_t thresh, float inv_keep, const uint64_t* __restrict__ seed_ptr, uint64_t salt) {
    constexpr int E = EPL * 64;
    __shared__ float red[2][4][E];
    const uint64_t seed = thresh ? (seed_ptr ? *seed_ptr : 0ull) * 0x9E3779B97F4A7C15ull + salt : 0ull;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    float ag[EPL], ab[EPL];
#pragma unroll
    for (int i = 0; i < EPL; ++i) { ag[i] = 0.f; ab[i] = 0.f; }
    float gm[EPL];
#pragma unroll
    for (int i = 0; i < EPL / 4; ++i) {
        const float4 g = *(const float4*)(gamma + (i * 64 + lane) * 4);
        gm[i * 4] = g.x; gm[i * 4 + 1] = g.y; gm[i * 4 + 2] = g.z; gm[i * 4 + 3] = g.w;
    }
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    for (int row = r0 + w; row < r1; row += 4) {
        const long base = (long)row * E;
        float d[EPL], h[EPL];
#pragma unroll
        for (int i = 0; i < EPL / 4; ++i) {
            const int col = (i * 64 + lane) * 4;
            const bf16x4 a = as_bf16x4(*(const uint2*)(dy + (long)row * lddy + col));
            const bf16x4 b = as_bf16x4(*(const uint2*)(xhat + base + col));
#pragma unroll
            for (int e = 0; e < 4; ++e) { d[i * 4 + e] = bf2f(a[e]); h[i * 4 + e] = bf2f(b[e]); }
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < EPL; ++i) {
            ag[i] += d[i] * h[i]; ab[i] += d[i];
            const float gd = d[i] * gm[i];
            s1 += gd; s2 += gd * h[i];
        }
        s1 = wave_sum(s1) * (1.f / E);
        s2 = wave_sum(s2) * (1.f / E);
        const float rs = rstd[row];
#pragma unroll
        for (int i = 0; i < EPL / 4; ++i) {
            const int col = (i * 64 + lane) * 4;
            bf16x4 o, od;
            bool keep[4] = {true, true, true, true};
            if (thresh) dropout_keep_run<4>(seed, (uint64_t)(base + col), thresh, keep);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gxe = rs * (d[i * 4 + e] * gm[i * 4 + e] - s1 - h[i * 4 + e] * s2);
                o[e] = f2bf(gxe);
                od[e] = keep[e] ? f2bf(gxe * inv_keep) : (bf16)0.f;
            }
            if (dx) *(uint2*)(dx + base + col) = as_uint2(o);              // gradient of the residual input
            if (dxd) *(uint2*)(dxd + base + col) = as_uint2(od);           // gradient of x through Dropout
        }
    }
#pragma unroll
    for (int i = 0; i < EPL / 4; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = (i * 64 + lane) * 4 + e;
            red[0][w][col] = ag[i * 4 + e];
            red[1][w][col] = ab[i * 4 + e];
        }
    __syncthreads();
    for (int col = threadIdx.x; col < E; col += 256) {
        part[(long)blockIdx.x * 2 * E + col] = red[0][0][col] + red[0][1][col] + red[0][2][col] + red[0][3][col];
        part[(long)blockIdx.x * 2 * E + E + col] = red[1][0][col] + red[1][1][col] + red[1][2][col] + red[1][3][col];
    }
}

// out[c] (+)= sum_r P[r][c]   (column reduce of partial rows; also used for bias gradients)
__global__ __launch_bounds__(1024) void reduce_rows_kernel(const float* __restrict__ P, float* __restrict__ out, int R, int C, int accumulate,
                                                           long ldp) {
    __shared__ float red[32][33];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + cl;
    float a = 0.f;
    if (c < C) for (int r = rg; r < R; r += 32) a += P[(long)r * ldp + c];
    red[rg][cl] = a;
    __syncthreads();
    if (rg == 0 && c < C) {
        a = 0.f;
        for (int i = 0; i < 32; ++i) a += red[i][cl];
        out[c] = accumulate ? out[c] + a : a;
    }
}

// one entry of tuber_multi_reduce: out[j] += sum_{s<S} P[s*stride + j], j < n.
//   mode 0: one thread per element (four when vec: n, stride % 4 == 0 and 16-byte aligned bases), s ascending (few slabs);
//   mode 1: 32 slab groups x 32 elements per block, LDS tree;  C > 0 (mode 1): P is [S][27][C] and the result goes to out[c*27 + tap]
//   (depthwise weight gradient).  next >= 0 chains a further contribution to the SAME out (a parameter used several times, e.g. the
//   shared decoder norm): the chain is added in order by the same thread, so the result equals the sequence of immediate reductions.
struct MultiReduceEntry {
    const float* P;
    float* out;
    long n, stride;
    int S, mode, C, next;
};
__global__ __launch_bounds__(1024) void multi_reduce_kernel(const MultiReduceEntry* __restrict__ table, const int2* __restrict__ blk) {
    __shared__ float red[32][33];
    const int2 be = blk[blockIdx.x];
    MultiReduceEntry e = table[be.x];
    // mode bit 3 (round 6): the gradient window is known to be zero (zero_grad ran, nothing has written it since), so the head entry does not READ it:
    // out = sum instead of out += sum -- the same value (0 + a == a), one read of the gradient buffer less per step
    const bool fresh = (e.mode & 8) != 0;
    e.mode &= 7;
    if (e.mode == 0) {                       // scalar: 1024 elements per block
        const long i = (long)be.y * 1024 + threadIdx.x;
        if (i >= e.n) return;
        float o = fresh ? 0.f : e.out[i];
        for (;;) {
            float a = 0.f;
            for (int s = 0; s < e.S; ++s) a += e.P[(long)s * e.stride + i];
            o += a;
            if (e.next < 0) break;
            e = table[e.next];
        }
        e.out[i] = o;
    } else if (e.mode == 2) {                // the same sums, four elements per thread: 4096 elements per block
        const long i = ((long)be.y * 1024 + threadIdx.x) * 4;
        if (i >= e.n) return;
        float4 o = fresh ? make_float4(0.f, 0.f, 0.f, 0.f) : *(const float4*)(e.out + i);
        for (;;) {
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
            for (int s = 0; s < e.S; ++s) {
                const float4 v = *(const float4*)(e.P + (long)s * e.stride + i);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
            o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
            if (e.next < 0) break;
            e = table[e.next];
        }
        *(float4*)(e.out + i) = o;
    } else {
        const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
        const long j = (long)be.y * 32 + cl;
        float o = 0.f;
        float* op = nullptr;
        if (rg == 0 && j < e.n) { op = e.C > 0 ? e.out + (j % e.C) * 27 + j / e.C : e.out + j; o = fresh ? 0.f : *op; }
        for (;;) {
            float a = 0.f;
            if (j < e.n) for (int s = rg; s < e.S; s += 32) a += e.P[(long)s * e.stride + j];
            red[rg][cl] = a;
            __syncthreads();
            if (rg == 0 && j < e.n) {
                a = 0.f;
#pragma unroll
                for (int i = 0; i < 32; ++i) a += red[i][cl];
                o += a;
            }
            if (e.next < 0) break;
            e = table[e.next];
            __syncthreads();
        }
        if (op) *op = o;
    }
}

// partial column sums of a bf16 [M, ld] matrix (bias gradient): block (bx, by) sums rows [bx*rpb, (bx+1)*rpb) of the
// 64 columns by*64.. ; thread = (row lane of 32, 8-column group of 8): 128-byte row segments, 4 rows in flight.
// With one row block the result goes straight to out (+= when accumulate); otherwise to P[bx][C].
__global__ __launch_bounds__(256) void colsum_partial_kernel(const bf16* __restrict__ g, float* __restrict__ P, float* __restrict__ out,
                                                             int accumulate, long M, int C, long ld, long rows_per_block) {
    __shared__ float red[32][8][9];
    const int groups = (C + 7) >> 3;
    const int gl = threadIdx.x & 7, rs = threadIdx.x >> 3;
    const int cg = blockIdx.y * 8 + gl;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    if (cg < groups) {
        long r = r0 + rs;
        for (; r + 96 < r1; r += 128) {
            uint4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const uint4*)(g + (r + 32 * u) * ld + cg * 8);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bf16x8 x = as_bf16x8(v[u]);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += bf2f(x[e]);
            }
        }
        for (; r < r1; r += 32) {
            const bf16x8 x = as_bf16x8(*(const uint4*)(g + r * ld + cg * 8));
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += bf2f(x[e]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rs][gl][e] = acc[e];
    __syncthreads();
    const int og = threadIdx.x >> 3, oe = threadIdx.x & 7;       // threads 0..63 = 8 groups x 8 columns
    const int col = (blockIdx.y * 8 + og) * 8 + oe;
    if (threadIdx.x < 64 && col < C) {
        float a = 0.f;
#pragma unroll
        for (int s = 0; s < 32; ++s) a += red[s][og][oe];
        if (gridDim.x == 1) out[col] = accumulate ? out[col] + a : a;
        else P[(long)blockIdx.x * C + col] = a;
    }
}

static inline int ew_grid(long M, int C) {
    const long rpp = 256 / (C >> 3);
    long nb = (M + rpp - 1) / rpp;
    if (nb > 4096) nb = 4096;
    return (int)nb;
}

// first stage for long partial-statistics lists (layer1: thousands of rows): [R][C] x2 -> [R2][C] x2, one block per
// (32 channels, row chunk); 256 threads = 8 channel quads x 32 row lanes
__global__ __launch_bounds__(256) void stat_rows_reduce_kernel(const float* __restrict__ st0, const float* __restrict__ st1, int R, int C,
                                                               float* __restrict__ o0, float* __restrict__ o1, int chunk) {
    __shared__ float red[2][32][33];
    const int qd = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int cq = blockIdx.x * 32 + qd * 4;
    const int r0 = blockIdx.y * chunk, r1 = min(R, r0 + chunk);
    float a[4] = {0.f, 0.f, 0.f, 0.f}, b[4] = {0.f, 0.f, 0.f, 0.f};
    if (cq < C) {
        for (int r = r0 + rl; r < r1; r += 32) {
            const float4 x = *(const float4*)(st0 + (long)r * C + cq), y = *(const float4*)(st1 + (long)r * C + cq);
            a[0] += x.x; a[1] += x.y; a[2] += x.z; a[3] += x.w;
            b[0] += y.x; b[1] += y.y; b[2] += y.z; b[3] += y.w;
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][rl][qd * 4 + e] = a[e]; red[1][rl][qd * 4 + e] = b[e]; }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int which = threadIdx.x >> 5, cl = threadIdx.x & 31;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) v += red[which][i][cl];
        const int c = blockIdx.x * 32 + cl;
        if (c < C) (which ? o1 : o0)[(long)blockIdx.y * C + c] = v;
    }
}

extern "C" {

// [R][C] partial-statistics rows -> [R2][C] (R2 = tuber_stat_rows_reduced(R) < R): cheap first stage before the finalize kernels
int tuber_stat_rows_reduced(int R) {
    return R > 512 ? 64 : R;
}

int tuber_stat_rows_reduce(const float* st0, const float* st1, int R, int C, float* out0, float* out1, hipStream_t stream) {
    const int R2 = tuber_stat_rows_reduced(R);
    if (R2 >= R || (C & 3)) return TUBER_EINVAL;
    const int chunk = ceil_div(R, R2);
    hipLaunchKernelGGL(stat_rows_reduce_kernel, dim3(ceil_div(C, 32), R2), dim3(256), 0, stream, st0, st1, R, C, out0, out1, chunk);
    TUBER_RETURN_LAUNCH();
}

int tuber_bn_finalize(const float* st0, const float* st1, int R, int C, float count, const float* gamma, const float* beta,
                      float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps,
                      float* scale, float* shift, float* mean, float* invstd, hipStream_t stream) {
    if (R <= 0 || C <= 0) return TUBER_EINVAL;
    if (R <= 128)
        hipLaunchKernelGGL(bn_finalize_kernel<256>, dim3(ceil_div(C, 32)), dim3(256), 0, stream, st0, st1, R, C, count, gamma, beta,
                           running_mean, running_var, num_batches_tracked, momentum, eps, scale, shift, mean, invstd);
    else
        hipLaunchKernelGGL(bn_finalize_kernel<1024>, dim3(ceil_div(C, 32)), dim3(1024), 0, stream, st0, st1, R, C, count, gamma, beta,
                           running_mean, running_var, num_batches_tracked, momentum, eps, scale, shift, mean, invstd);
    TUBER_RETURN_LAUNCH();
}

int tuber_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                         float* scale, float* shift, int C, hipStream_t stream) {
    hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(ceil_div(C, 256)), dim3(256), 0, stream, gamma, beta, running_mean, running_var,
                       eps, scale, shift, C);
    TUBER_RETURN_LAUNCH();
}

// tuber_bn_eval_affine for n BatchNorm layers in ONE launch.  table: n rows of 8 64-bit words in DEVICE memory -- the pointers gamma, beta,
// running_mean, running_var, scale, shift, then the channel count and one unused word; cmax >= every row's channel count.
int tuber_bn_eval_affine_multi(const void* table, int n, int cmax, float eps, hipStream_t stream) {
    if (!table || n <= 0 || n > 65535 || cmax <= 0) return TUBER_EINVAL;
    hipLaunchKernelGGL(bn_eval_affine_multi_kernel, dim3(ceil_div(cmax, 256), n), dim3(256), 0, stream, (const BnAffineRow*)table, eps);
    TUBER_RETURN_LAUNCH();
}

int tuber_bn_bwd_finalize(const float* st0, const float* st1, int R, int C, float count, const float* gamma, const float* mean,
                          const float* invstd, float* cA, float* cB, float* cC, float* dgamma, float* dbeta, int accumulate,
                          hipStream_t stream) {
    if (R <= 0 || C <= 0) return TUBER_EINVAL;
    if (R <= 128)
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<256>, dim3(ceil_div(C, 32)), dim3(256), 0, stream, st0, st1, R, C, count, gamma, mean,
                           invstd, cA, cB, cC, dgamma, dbeta, accumulate);
    else
        hipLaunchKernelGGL(bn_bwd_finalize_kernel<1024>, dim3(ceil_div(C, 32)), dim3(1024), 0, stream, st0, st1, R, C, count, gamma, mean,
                           invstd, cA, cB, cC, dgamma, dbeta, accumulate);
    TUBER_RETURN_LAUNCH();
}

static inline bool chan_ok(int C) { return C >= 8 && C <= 2048 && (256 % (C >> 3)) == 0 && (C & 7) == 0; }

int tuber_block_out_fwd(const void* c4, const float* s4, const float* h4, const void* res, const float* rs, const float* rh,
                        void* y, long M, int C, hipStream_t stream) {
    if (!chan_ok(C)) return TUBER_EINVAL;
    hipLaunchKernelGGL(block_out_fwd_kernel, dim3(ew_grid(M, C)), dim3(256), 0, stream, (const bf16*)c4, s4, h4, (const bf16*)res,
                       rs, rh, (bf16*)y, (uint8_t*)nullptr, M, C);
    TUBER_RETURN_LAUNCH();
}

// tuber_block_out_fwd that ALSO writes the ReLU mask of its output as a bit field: ymask [M][C / 8] bytes, bit e of byte (m, c / 8) =
// y[m][c + e] > 0.  The join backward of this block (tuber_gemm_nt_join_mask) reads it instead of y: 1 / 16 of the bytes.
int tuber_block_out_fwd_mask(const void* c4, const float* s4, const float* h4, const void* res, const float* rs, const float* rh,
                             void* y, void* ymask, long M, int C, hipStream_t stream) {
    if (!chan_ok(C) || !ymask) return TUBER_EINVAL;
    hipLaunchKernelGGL(block_out_fwd_kernel, dim3(ew_grid(M, C)), dim3(256), 0, stream, (const bf16*)c4, s4, h4, (const bf16*)res,
                       rs, rh, (bf16*)y, (uint8_t*)ymask, M, C);
    TUBER_RETURN_LAUNCH();
}

// tuber_block_out_fwd with an fp32 residual stream (eval precision mode): res32 (fp32 [M, C], the previous block's y32) or, when NULL, the
// bf16 shortcut `res` with its optional BatchNorm (rs / rh); writes y (bf16) AND y32 (fp32).
int tuber_block_out_fwd_f32(const void* c4, const float* s4, const float* h4, const void* res, const float* rs, const float* rh,
                            const float* res32, void* y, float* y32, long M, int C, hipStream_t stream) {
    if (!chan_ok(C) || (!res && !res32) || !y || !y32) return TUBER_EINVAL;
    hipLaunchKernelGGL(block_out_fwd_f32_kernel, dim3(ew_grid(M, C)), dim3(256), 0, stream, (const bf16*)c4, s4, h4, (const bf16*)res,
                       rs, rh, res32, (bf16*)y, y32, M, C);
    TUBER_RETURN_LAUNCH();
}

// rows of partial stats written by the row-blocked reduce kernels for an [M, C] tensor: ~16K elements per block
// (8 passes of 256 threads x 8 channels), at most 1024 blocks
int tuber_rowblock_count(long M, int C) {
    long nb = (M * (long)C + 16383) / 16384;
    if (nb > 1024) nb = 1024;
    if (nb < 1) nb = 1;
    return (int)nb;
}
static inline long rows_per_block(long M, int C) { const int nb = tuber_rowblock_count(M, C); return (M + nb - 1) / nb; }

int tuber_block_out_bwd(const void* dy, const void* y, const void* c4, const void* cds, void* dz, float* st_dz, float* st_c4,
                        float* st_ds, long M, int C, hipStream_t stream) {
    if (!chan_ok(C)) return TUBER_EINVAL;
    hipLaunchKernelGGL(block_out_bwd_kernel, dim3(tuber_rowblock_count(M, C)), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)y,
                       (const bf16*)c4, (const bf16*)cds, (bf16*)dz, st_dz, st_c4, st_ds, M, C, rows_per_block(M, C));
    TUBER_RETURN_LAUNCH();
}

int tuber_relu_bn_bwd_reduce(const void* g, const void* x, const float* sc, const float* sh, void* dz, float* st0, float* st1,
                             long M, int C, hipStream_t stream) {
    if (!chan_ok(C)) return TUBER_EINVAL;
    hipLaunchKernelGGL(relu_bn_bwd_reduce_kernel, dim3(tuber_rowblock_count(M, C)), dim3(256), 0, stream, (const bf16*)g,
                       (const bf16*)x, sc, sh, (bf16*)dz, st0, st1, M, C, rows_per_block(M, C));
    TUBER_RETURN_LAUNCH();
}

// BatchNorm backward, finalize + apply in one launch, for short partial lists (R <= tuber_bn_bwd_fa_max_rows(), C % 128 == 0):
// dx = cA*dz + cB*x + cC with the coefficients derived per workgroup from the partial rows; dgamma / dbeta are ACCUMULATED (+=) unless
// NULL (frozen BatchNorm).  Same arithmetic as tuber_bn_bwd_finalize + tuber_bn_bwd_apply (fp64 sums, another fixed order).
int tuber_bn_bwd_fa_max_rows(void) { return 128; }
// rows per thread (x 16 = rows per workgroup) of tuber_bn_bwd_fa: the candidate with the least (rounds of 256 workgroups) x (rows per
// thread + 6, the derive prologue every workgroup pays, in row units)
static int fa_rows_per_thread(long M, int C) {
    if (ceil_div(M, 128L) * (C / FA_CS) >= 1024) return 8;      // many rounds either way (layer1 / layer2): 128 rows measured best
    const int cand[3] = {4, 8, 11};
    int best = 8;
    long best_cost = -1;
    for (int i = 0; i < 3; ++i) {
        const long wgs = ceil_div(M, 16L * cand[i]) * (C / FA_CS);
        const long cost = ceil_div(wgs, 256L) * (cand[i] + 6);
        if (best_cost < 0 || cost < best_cost) { best_cost = cost; best = cand[i]; }
    }
    return best;
}
int tuber_bn_bwd_fa_rows(long M, int C) { return 16 * fa_rows_per_thread(M, C); }
static int g_fa_rows_forced = 0;     // test hook: rows per thread forced (0 = heuristic), so the kernel test covers every instantiation at every shape
int tuber_bn_bwd_fa_rows_set(int kr) { g_fa_rows_forced = kr; return 0; }
int tuber_bn_bwd_fa(const float* st0, const float* st1, int R, int C, float count, const float* gamma, const float* mean,
                    const float* invstd, float* dgamma, float* dbeta, const void* dz, const void* x, void* dx, long M,
                    hipStream_t stream) {
    if (R <= 0 || R > 128 || C <= 0 || (C % FA_CS) || M <= 0) return TUBER_EINVAL;
    const int kr = g_fa_rows_forced ? g_fa_rows_forced : fa_rows_per_thread(M, C);
#define FA_LAUNCH(KR) hipLaunchKernelGGL(bn_bwd_fa_kernel<KR>, dim3(ceil_div(M, 16L * KR), C / FA_CS), dim3(256), 0, stream, st0, st1, R, C, \
                                         count, gamma, mean, invstd, dgamma, dbeta, (const bf16*)dz, (const bf16*)x, (bf16*)dx, M)
    if (kr == 4) FA_LAUNCH(4);
    else if (kr == 11) FA_LAUNCH(11);
    else FA_LAUNCH(8);
#undef FA_LAUNCH
    TUBER_RETURN_LAUNCH();
}

int tuber_bn_bwd_apply(const void* dz, const void* x, const float* cA, const float* cB, const float* cC, void* dx, long M, int C,
                       hipStream_t stream) {
    if (!chan_ok(C)) return TUBER_EINVAL;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_grid(M, C)), dim3(256), 0, stream, (const bf16*)dz, (const bf16*)x, cA, cB, cC,
                       (bf16*)dx, M, C);
    TUBER_RETURN_LAUNCH();
}

// y = LayerNorm(Dropout_p(x) + res): res may be NULL, p may be 0; y rows have leading dimension ldy (>= E) so two LayerNorms can
// write the halves of one concatenated buffer; xhat [M,E] and rstd [M] are saved for the backward (NULL in eval).
int tuber_layernorm_fwd(const void* x, const void* res, const float* gamma, const float* beta, void* y, long ldy, void* xhat, float* rstd,
                        int M, int E, float eps, float p, const void* seed_ptr, unsigned long long salt, hipStream_t stream) {
    if (ldy < E || (ldy & 3) || p < 0.f || p >= 1.f) return TUBER_EINVAL;
    dim3 grid(ceil_div(M, 4)), block(256);
    const uint32_t th = (uint32_t)((double)p * 4294967296.0);
    const float ik = dropout_inv_keep(p);
#define LNF(EPL) hipLaunchKernelGGL(layernorm_fwd_kernel<EPL>, grid, block, 0, stream, (const bf16*)x, (const bf16*)res, gamma, beta, \
                                    (bf16*)y, ldy, (bf16*)xhat, rstd, M, eps, th, ik, (const uint64_t*)seed_ptr, (uint64_t)salt)
    if (E == 256) LNF(4);
    else if (E == 2048) LNF(32);
    else return TUBER_EINVAL;
#undef LNF
    TUBER_RETURN_LAUNCH();
}

// eval precision mode: LayerNorm(x + res) with fp32 stream operands (x32 / res32 override x / res when not NULL; res, res32 both NULL =
// no residual) -> y (bf16, leading dimension ldy) and y32 (fp32 [M, E], may be NULL).
int tuber_layernorm_fwd_f32(const void* x, const float* x32, const void* res, const float* res32, const float* gamma, const float* beta,
                            void* y, long ldy, float* y32, int M, int E, float eps, hipStream_t stream) {
    if (ldy < E || (ldy & 3) || (!x && !x32) || !y) return TUBER_EINVAL;
    dim3 grid(ceil_div(M, 4)), block(256);
#define LNF(EPL) hipLaunchKernelGGL(layernorm_fwd_f32_kernel<EPL>, grid, block, 0, stream, (const bf16*)x, x32, (const bf16*)res, res32, gamma, beta, \
                                    (bf16*)y, ldy, y32, M, eps)
    if (E == 256) LNF(4);
    else if (E == 2048) LNF(32);
    else return TUBER_EINVAL;
#undef LNF
    TUBER_RETURN_LAUNCH();
}

// 16 rows (4 per wave) per block up to 1024 blocks: the per-block chain is what bounds these small launches
int tuber_layernorm_bwd_blocks(int M) { const int nb = ceil_div(M, 16); return nb > 1024 ? 1024 : nb; }

// backward of tuber_layernorm_fwd: dx = gradient w.r.t. res (and w.r.t. x when p == 0), dxd = gradient w.r.t. x through the
// dropout mask (either may be NULL); dy rows have leading dimension lddy.  partial must hold 2 * blocks * E floats;
// dgamma/dbeta are accumulated into or written.
int tuber_layernorm_bwd(const void* dy, long lddy, const void* xhat, const float* rstd, const float* gamma, void* dx, void* dxd,
                        float* partial, float* dgamma, float* dbeta, int accumulate, int M, int E,
                        float p, const void* seed_ptr, unsigned long long salt, hipStream_t stream) {
    if (lddy < E || (lddy & 3) || p < 0.f || p >= 1.f) return TUBER_EINVAL;
    const int nb = tuber_layernorm_bwd_blocks(M);
    int rpb = ceil_div(M, nb);
    rpb = ceil_div(rpb, 4) * 4;
    dim3 grid(nb), block(256);
    const uint32_t th = (uint32_t)((double)p * 4294967296.0);
    const float ik = dropout_inv_keep(p);
#define LNB(EPL) hipLaunchKernelGGL(layernorm_bwd_kernel<EPL>, grid, block, 0, stream, (const bf16*)dy, lddy, (const bf16*)xhat, rstd, gamma, \
                                    (bf16*)dx, (bf16*)dxd, partial, M, rpb, th, ik, (const uint64_t*)seed_ptr, (uint64_t)salt)
    if (E == 256) LNB(4);
    else if (E == 2048) LNB(32);
    else return TUBER_EINVAL;
#undef LNB
    if (accumulate == 2) {            // the caller reduces partial [nb][2E] later (tuber_multi_reduce)
    } else if (dbeta == dgamma + E) { // weight and bias adjacent in the flat gradient buffer: one reduction over [nb][2E]
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(ceil_div(2 * E, 32)), dim3(1024), 0, stream, partial, dgamma, nb, 2 * E, accumulate, (long)2 * E);
    } else {
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(ceil_div(E, 32)), dim3(1024), 0, stream, partial, dgamma, nb, E, accumulate, (long)2 * E);
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(ceil_div(E, 32)), dim3(1024), 0, stream, partial + E, dbeta, nb, E, accumulate, (long)2 * E);
    }
    TUBER_RETURN_LAUNCH();
}

int tuber_reduce_rows(const float* P, float* out, int R, int C, int accumulate, hipStream_t stream) {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(ceil_div(C, 32)), dim3(1024), 0, stream, P, out, R, C, accumulate, (long)C);
    TUBER_RETURN_LAUNCH();
}

// dbias[c] (+)= sum_m g[m][c] for bf16 g [M, ld] (ld % 8 == 0, readable up to ceil8(C) columns);
// partial must hold tuber_colsum_blocks(M) * C floats
int tuber_colsum_blocks(long M) {
    if (M <= 4096) return 1;
    long nb = (M + 255) / 256;
    return (int)(nb > 256 ? 256 : nb);
}

int tuber_colsum(const void* g, float* partial, float* out, int accumulate, long M, int C, long ld, hipStream_t stream) {
    if (M <= 0 || C <= 0 || (ld & 7) || ld < ((C + 7) & ~7)) return TUBER_EINVAL;
    const int nb = tuber_colsum_blocks(M);
    const long rpb = (M + nb - 1) / nb;
    const int groups = (C + 7) / 8;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nb, (groups + 7) / 8), dim3(256), 0, stream, (const bf16*)g, partial, out, accumulate,
                       M, C, ld, rpb);
    if (nb > 1 && accumulate != 2)             // accumulate == 2: partial [nb][C] reduced later by tuber_multi_reduce
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(ceil_div(C, 32)), dim3(1024), 0, stream, partial, out, nb, C, accumulate, (long)C);
    TUBER_RETURN_LAUNCH();
}

// All deferred second-stage reductions of one backward pass in ONE launch.  The weight-gradient GEMMs, depthwise weight gradients,
// LayerNorm and bias gradients each leave per-workgroup fp32 partials (accumulate == 2 in their launchers); ~240 five-microsecond
// reduce launches per step collapse into this one.  table[e] = MultiReduceEntry, blk[b] = (entry, block index inside the entry).
// Summation orders are those of reduce_slabs_flat_kernel (mode 0) and reduce_rows / reduce_slabs / dw_wgrad_reduce (mode 1), so the
// result is bit-identical to the immediate path.
int tuber_multi_reduce(const void* table, const void* blk, int nblocks, hipStream_t stream) {
    if (nblocks <= 0) return TUBER_EINVAL;
    hipLaunchKernelGGL(multi_reduce_kernel, dim3(nblocks), dim3(1024), 0, stream, (const MultiReduceEntry*)table, (const int2*)blk);
    TUBER_RETURN_LAUNCH();
}
int tuber_multi_reduce_entry_bytes() { return (int)sizeof(MultiReduceEntry); }

}  // extern "C"
