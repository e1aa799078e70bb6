// LDS-staged depthwise 3x3x3 Conv3d for the stride-1 bottlenecks (47 of the 50 CSN-152 blocks) -- gfx950.
// reference: ResNeXtBottleneck.conv3, models/backbones/ir_CSN_152.py:48-51 (groups = C, padding 1).
//
// The register-tiled kernels of dwconv.hip re-activate (BatchNorm apply + ReLU + padding select) every input vector in
// every thread that touches it -- 13.5x redundant VALU work, which is what bounds them (layer1: 94 us for 89 MB).  Here a
// 512-thread workgroup owns an 8 x 16 output tile of 64 channels and slides over a range of output planes t:
//   * each input plane (10 x 18 positions with halo) is fetched ONCE per workgroup, activated once, and parked in a
//     3-plane LDS ring as fp32 (3 x 46 KB); the fetch of plane t+2 is in flight while plane t is computed;
//   * a thread owns 4 channels x 4 consecutive output columns of one tile row and reads its 9 x 6 input vectors with
//     conflict-free ds_read_b128; the 27 x 4 filter taps live in registers; the 432 FMAs per plane are written on float2 so
//     the compiler can pair them (v_pk_fma_f32);
//   * zero padding is applied AFTER the activation (the reference pads relu(bn1(.))).
// Modes: forward (+ partial statistics of bn3), data gradient (flipped taps; fused with the backward of relu(bn1(.)) and
// bn1's partial statistics), weight gradient (27 x 4 accumulators per thread, one partial [27][64] per workgroup).
#include <cstdlib>

#include "common.h"

typedef float f32x2 __attribute__((ext_vector_type(2)));

extern "C" int tuber_dw_wgrad_reduce(const float* partial, float* dw, int R, int C, int accumulate, hipStream_t stream);

namespace {

constexpr int TH = 8, TW = 16, PH = TH + 2, PW = TW + 2, NPOS = PH * PW;     // 180 staged positions per plane
constexpr int PLANE = NPOS * 64;                                             // floats
constexpr int NLD = (NPOS * 16 + 511) / 512;                                 // 6 vector loads per thread per plane

struct TileGeom {
    int N, T, H, W, C;
    int tc, tchunks, htiles, wtiles;
};

enum { M_FWD = 0, M_BWD_DATA = 1, M_BWD_WEIGHT = 2 };

struct TileArgs {
    const bf16* in;        // staged tensor: x (fwd / wgrad: activated on load when sc != NULL) or gout (bwd data)
    const float* sc; const float* sh;
    const float* w;        // [C][27] fp32
    bf16* out;             // fwd: conv output; bwd data: dz
    const bf16* aux;       // bwd data: x at the output position (mask + statistics);  wgrad: gout
    float* st0; float* st1;   // partial statistics rows [gridDim.x][C]
    float* P;              // wgrad partials [gridDim.x][27][C]
    // BNG variants: the gradient operand (bwd data: the staged tensor; wgrad: the side input) is not read from HBM as a finished
    // tensor but FORMED on load as cA*dzu + cB*xu + cC -- the BatchNorm backward of the layer above (bn3): dzu = its masked output
    // gradient (`in` resp. `aux`), xu = its input (the raw depthwise-conv output c3), coefficients derived by every workgroup for
    // its 64 channels from the R partial rows (sum dz, sum dz*x) like tuber_bn_bwd_fa does
    const bf16* xu;
    const float* bst0; const float* bst1; int bR; float bcount;
    const float* bgamma; const float* bmean; const float* binvstd;
    float* bdgamma; float* bdbeta;      // accumulated (+=) by the workgroups with blockIdx.x == 0 when not NULL
    // FIN (forward): the training-mode BatchNorm IN FRONT of the conv (bn1) is finalised here instead of by a tuber_bn_finalize launch:
    // every workgroup derives scale / shift of its 64 channels from the producer's bR partial rows (bst0 = sum x, bst1 = sum x^2,
    // bcount, bgamma as above); the workgroups with blockIdx.x == 0 publish scale / shift / mean / invstd for the backward and update
    // the running statistics
    const float* fbeta; float* frmean; float* frvar; long long* fnbt; float fmom, feps;
    float* fscale; float* fshift; float* fmean; float* finvstd;
    TileGeom g;
};

// (bx, by) = the workgroup's position in ITS problem's grid: the kernels below pass blockIdx, the merged backward launch an offset one
template <int MODE, bool BNG = false, bool ACT = true, bool FIN = false>
__device__ __forceinline__ void dwconv_tile_body(const TileArgs& a, const int bx, const int by) {
    static_assert(!FIN || (MODE == M_FWD && ACT && !BNG), "FIN: the forward conv behind a training-mode BatchNorm + ReLU");
    extern __shared__ __attribute__((aligned(16))) float smem[];        // ring[3][PLANE] | red
    const TileGeom g = a.g;
    const int tid = threadIdx.x;
    const int cl = tid & 15, slot = tid >> 4, row = slot >> 2, cg = slot & 3;
    const int c0 = by * 64, c = c0 + cl * 4;
    int b = bx;
    const int wt = b % g.wtiles; b /= g.wtiles;
    const int ht = b % g.htiles; b /= g.htiles;
    const int tk = b % g.tchunks; const int n = b / g.tchunks;
    const int h0 = ht * TH, w0 = wt * TW;
    const int t0 = tk * g.tc, t1 = min(g.T, t0 + g.tc);

    // ---- per-thread staging slots (loop invariant over t) ----
    int s_off[NLD], s_lds[NLD];
    bool s_ok[NLD];
    float sa[4] = {1.f, 1.f, 1.f, 1.f}, sb[4] = {0.f, 0.f, 0.f, 0.f};   // (tid + 512 i) & 15 == cl: one channel quad per thread
    constexpr bool act = MODE != M_BWD_DATA && ACT;     // forward / weight gradient stage relu(bn1(.)); ACT = false: the plain forward conv (no scale / shift given)
    if (act && !FIN) {
        const float4 s = *(const float4*)(a.sc + c), h = *(const float4*)(a.sh + c);
        sa[0] = s.x; sa[1] = s.y; sa[2] = s.z; sa[3] = s.w; sb[0] = h.x; sb[1] = h.y; sb[2] = h.z; sb[3] = h.w;
    }
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + 512 * i;
        const int pos = idx >> 4, q = idx & 15;
        const int pr = pos / PW, pc = pos % PW;
        const int hi = h0 - 1 + pr, wi = w0 - 1 + pc;
        s_ok[i] = pos < NPOS && hi >= 0 && hi < g.H && wi >= 0 && wi < g.W;
        s_off[i] = s_ok[i] ? (hi * g.W + wi) * g.C + c0 + q * 4 : 0;
        s_lds[i] = pos < NPOS ? pos * 64 + q * 4 : -1;
    }
    const long plane_elems = (long)g.H * g.W * g.C;
    const bf16* in_n = a.in + (long)n * g.T * plane_elems;
    constexpr bool STG2 = BNG && MODE == M_BWD_DATA;        // the staged tensor is formed from two tensors
    const bf16* in2_n = STG2 ? a.xu + (long)n * g.T * plane_elems : nullptr;
    uint2 regs[NLD], regs_a[NLD], regs_b[NLD];     // regs: the steady-state prefetch set; _a / _b: the two extra planes of the prologue
    uint2 regx[STG2 ? NLD : 1], regx_a[STG2 ? NLD : 1], regx_b[STG2 ? NLD : 1];
    // input plane t -> registers.  The loads are UNCONDITIONAL: slots outside the volume read element 0 of a valid plane (s_off = 0,
    // plane 0 when t is outside) and are cleared when they are parked -- a predicated load is a saveexec / branch / restore around every
    // one of the 18-36 loads of a prologue that is most of the kernel in the short-T stages (phase timestamps: 1.2-2.3 us of a 6.5-10 us
    // workgroup life pass before the first load is out, at two waves per SIMD every instruction counts twice).
    auto fetch = [&](int t, uint2 (&regs)[NLD], uint2 (&rx)[STG2 ? NLD : 1]) {
        const bool tok = t >= 0 && t < g.T;
        const bf16* p = in_n + (long)(tok ? t : 0) * plane_elems;
#pragma unroll
        for (int i = 0; i < NLD; ++i) regs[i] = *(const uint2*)(p + s_off[i]);
        if constexpr (STG2) {
            const bf16* p2 = in2_n + (long)(tok ? t : 0) * plane_elems;
#pragma unroll
            for (int i = 0; i < NLD; ++i) rx[i] = *(const uint2*)(p2 + s_off[i]);
        }
    };
    float kA[4] = {1.f, 1.f, 1.f, 1.f}, kB[4] = {0.f, 0.f, 0.f, 0.f}, kC[4] = {0.f, 0.f, 0.f, 0.f};     // BNG: this thread's channel quad
    auto park = [&](int t, const uint2 (&regs)[NLD], const uint2 (&rx)[STG2 ? NLD : 1]) {   // registers -> activated fp32 in ring slot t mod 3
        const bool tok = t >= 0 && t < g.T;
        float* dst = smem + ((t + 3) % 3) * PLANE;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (s_lds[i] < 0) continue;
            const bf16x4 v = as_bf16x4(regs[i]);
            float4 o;
            if constexpr (STG2) {
                const bool ok = tok && s_ok[i];          // the gradient is zero outside the volume (cC must not leak into the padding)
                const bf16x4 u = as_bf16x4(rx[i]);
                o.x = ok ? fmaf(kA[0], bf2f(v[0]), fmaf(kB[0], bf2f(u[0]), kC[0])) : 0.f;
                o.y = ok ? fmaf(kA[1], bf2f(v[1]), fmaf(kB[1], bf2f(u[1]), kC[1])) : 0.f;
                o.z = ok ? fmaf(kA[2], bf2f(v[2]), fmaf(kB[2], bf2f(u[2]), kC[2])) : 0.f;
                o.w = ok ? fmaf(kA[3], bf2f(v[3]), fmaf(kB[3], bf2f(u[3]), kC[3])) : 0.f;
            } else if (act) {
                const bool ok = tok && s_ok[i];
                o.x = ok ? fmaxf(fmaf(bf2f(v[0]), sa[0], sb[0]), 0.f) : 0.f;
                o.y = ok ? fmaxf(fmaf(bf2f(v[1]), sa[1], sb[1]), 0.f) : 0.f;
                o.z = ok ? fmaxf(fmaf(bf2f(v[2]), sa[2], sb[2]), 0.f) : 0.f;
                o.w = ok ? fmaxf(fmaf(bf2f(v[3]), sa[3], sb[3]), 0.f) : 0.f;
            } else {
                const bool ok = tok && s_ok[i];
                o.x = ok ? bf2f(v[0]) : 0.f; o.y = ok ? bf2f(v[1]) : 0.f; o.z = ok ? bf2f(v[2]) : 0.f; o.w = ok ? bf2f(v[3]) : 0.f;
            }
            *(float4*)(dst + s_lds[i]) = o;
        }
    };

    // ---- all three planes of the prologue go out together (one memory latency instead of three dependent fetch -> park rounds:
    // the short-T stages run one output plane per workgroup, so the prologue IS the kernel) ----
    // FIN: the partial-statistics rows (16 channel quads x 32 row groups, four rows per thread in flight together) and the per-channel
    // constants go out AHEAD of the planes -- vector loads complete in order, so these L2 hits are waited for alone and the
    // finalisation runs under the planes' latency
    float4 fu[FIN ? 4 : 1], fv[FIN ? 4 : 1];
    float f_gam = 0.f, f_bet = 0.f, f_rm = 0.f, f_rv = 0.f;
    if constexpr (FIN) {
        const float* p0 = a.bst0 + c0 + (tid & 15) * 4;
        const float* p1 = a.bst1 + c0 + (tid & 15) * 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long rr = min((tid >> 4) + 32 * k, a.bR - 1);
            fu[k] = *(const float4*)(p0 + rr * g.C); fv[k] = *(const float4*)(p1 + rr * g.C);
        }
        const int cc = c0 + (tid & 63);
        f_gam = a.bgamma[cc]; f_bet = a.fbeta[cc];
        if (a.frmean) { f_rm = a.frmean[cc]; f_rv = a.frvar[cc]; }
    }
    fetch(t0 - 1, regs_a, regx_a);
    fetch(t0, regs_b, regx_b);
    fetch(t0 + 1, regs, regx);

    if constexpr (FIN) {
        double* red = (double*)smem;                     // [2][32][64] in the (still empty) ring
        float* coef = smem + 3 * PLANE + 27 * 64;        // [2][64] behind the filter taps
        {
            const int q = tid & 15, rg = tid >> 4;
            double ua[4] = {0, 0, 0, 0}, ub[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; ++k) {                // rows beyond bR were read from row bR - 1 and are added as zero
                const bool ok = rg + 32 * k < a.bR;
                ua[0] += ok ? fu[k].x : 0.f; ua[1] += ok ? fu[k].y : 0.f; ua[2] += ok ? fu[k].z : 0.f; ua[3] += ok ? fu[k].w : 0.f;
                ub[0] += ok ? fv[k].x : 0.f; ub[1] += ok ? fv[k].y : 0.f; ub[2] += ok ? fv[k].z : 0.f; ub[3] += ok ? fv[k].w : 0.f;
            }
            const float* p0 = a.bst0 + c0 + q * 4;
            const float* p1 = a.bst1 + c0 + q * 4;
            for (int r = rg + 128; r < a.bR; r += 32) {  // longer lists than the model's shapes produce
                const float4 u = *(const float4*)(p0 + (long)r * g.C), v = *(const float4*)(p1 + (long)r * g.C);
                ua[0] += u.x; ua[1] += u.y; ua[2] += u.z; ua[3] += u.w;
                ub[0] += v.x; ub[1] += v.y; ub[2] += v.z; ub[3] += v.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { red[(0 * 32 + rg) * 64 + q * 4 + e] = ua[e]; red[(1 * 32 + rg) * 64 + q * 4 + e] = ub[e]; }
        }
        __syncthreads();
        double* red2 = red + 2 * 32 * 64;                // [2][4][64]
        {
            const int which = tid >> 8, part = (tid >> 6) & 3, ch = tid & 63;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += red[(which * 32 + part * 8 + k) * 64 + ch];
            red2[(which * 4 + part) * 64 + ch] = s;
        }
        __syncthreads();
        if (tid < 64) {
            // the arithmetic of bn_finalize_kernel (norm.hip), expression for expression: fp64 sums of the fp32 partial rows are exact,
            // so scale / shift / mean / invstd and the running statistics are the two-launch path's values
            const double sx = (red2[(0 * 4 + 0) * 64 + tid] + red2[(0 * 4 + 1) * 64 + tid]) + (red2[(0 * 4 + 2) * 64 + tid] + red2[(0 * 4 + 3) * 64 + tid]);
            const double sxx = (red2[(1 * 4 + 0) * 64 + tid] + red2[(1 * 4 + 1) * 64 + tid]) + (red2[(1 * 4 + 2) * 64 + tid] + red2[(1 * 4 + 3) * 64 + tid]);
            const int cc = c0 + tid;
            const float count = a.bcount;
            const double mean = sx / (double)count;
            double var = sxx / (double)count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)a.feps));
            const float scl = f_gam * invstd;
            const float shf = f_bet - (float)mean * scl;
            coef[tid] = scl;
            coef[64 + tid] = shf;
            if (bx == 0) {
                a.fscale[cc] = scl;
                a.fshift[cc] = shf;
                a.fmean[cc] = (float)mean;
                a.finvstd[cc] = invstd;
                if (a.frmean) {
                    const double unbiased = count > 1.f ? var * (double)count / ((double)count - 1.0) : var;
                    a.frmean[cc] = (1.f - a.fmom) * f_rm + a.fmom * (float)mean;
                    a.frvar[cc] = (1.f - a.fmom) * f_rv + a.fmom * (float)unbiased;
                }
                if (a.fnbt && by == 0 && tid == 0) *a.fnbt += 1;
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) { sa[e] = coef[cl * 4 + e]; sb[e] = coef[64 + cl * 4 + e]; }
        __syncthreads();                                  // the ring is written next
    }

    // ---- BNG: coefficients of the BatchNorm backward above, derived under the latency of the fetches just issued ----
    if constexpr (BNG) {
        double* red = (double*)smem;                     // [2][32][64] in the (still empty) ring
        float* coef = smem + 3 * PLANE + 27 * 64;        // [3][64] behind the filter taps
        {
            const int q = tid & 15, rg = tid >> 4;       // 16 channel quads x 32 row groups
            double sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
            const float* p0 = a.bst0 + c0 + q * 4;
            const float* p1 = a.bst1 + c0 + q * 4;
            // four partial rows per trip, all eight loads in flight together (a one-row trip waited for its own L2 round trip: bR = 88
            // in layer3 was three dependent latencies of a 16 us launch); rows beyond bR are read from row bR - 1 and added as zero
            for (int r = rg; r < a.bR; r += 128) {
                float4 u[4], v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const long rr = min(r + 32 * k, a.bR - 1);
                    u[k] = *(const float4*)(p0 + rr * g.C); v[k] = *(const float4*)(p1 + rr * g.C);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const bool ok = r + 32 * k < a.bR;
                    sa[0] += ok ? u[k].x : 0.f; sa[1] += ok ? u[k].y : 0.f; sa[2] += ok ? u[k].z : 0.f; sa[3] += ok ? u[k].w : 0.f;
                    sb[0] += ok ? v[k].x : 0.f; sb[1] += ok ? v[k].y : 0.f; sb[2] += ok ? v[k].z : 0.f; sb[3] += ok ? v[k].w : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { red[(0 * 32 + rg) * 64 + q * 4 + e] = sa[e]; red[(1 * 32 + rg) * 64 + q * 4 + e] = sb[e]; }
        }
        __syncthreads();
        // two levels (all 512 threads sum 8 row groups each, then 64 threads sum 4): the 32-entry serial fp64 chain of one wave was ~2 us of
        // every workgroup's prologue.  (fp64 sums of fp32 partial rows are exact, so the association does not change the result.)
        double* red2 = red + 2 * 32 * 64;                // [2][4][64]
        {
            const int which = tid >> 8, part = (tid >> 6) & 3, ch = tid & 63;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += red[(which * 32 + part * 8 + k) * 64 + ch];
            red2[(which * 4 + part) * 64 + ch] = s;
        }
        __syncthreads();
        if (tid < 64) {
            const double sa = (red2[(0 * 4 + 0) * 64 + tid] + red2[(0 * 4 + 1) * 64 + tid]) + (red2[(0 * 4 + 2) * 64 + tid] + red2[(0 * 4 + 3) * 64 + tid]);
            const double sb = (red2[(1 * 4 + 0) * 64 + tid] + red2[(1 * 4 + 1) * 64 + tid]) + (red2[(1 * 4 + 2) * 64 + tid] + red2[(1 * 4 + 3) * 64 + tid]);
            const int cc = c0 + tid;
            const double mu = a.bmean[cc], rr = a.binvstd[cc], gm = a.bgamma[cc];
            const double sum_dz = sa, sum_dz_xhat = (sb - mu * sa) * rr;
            const double m1 = sum_dz / a.bcount, m2 = sum_dz_xhat / a.bcount;
            coef[0 * 64 + tid] = (float)(gm * rr);
            coef[1 * 64 + tid] = (float)(-gm * rr * rr * m2);
            coef[2 * 64 + tid] = (float)(gm * rr * rr * m2 * mu - gm * rr * m1);
            if (bx == 0 && a.bdgamma) {
                a.bdgamma[cc] += (float)sum_dz_xhat;
                a.bdbeta[cc] += (float)sum_dz;
            }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) { kA[e] = coef[0 * 64 + cl * 4 + e]; kB[e] = coef[1 * 64 + cl * 4 + e]; kC[e] = coef[2 * 64 + cl * 4 + e]; }
        __syncthreads();                                  // the ring is written next
    }

    // ---- filter taps of this thread's 4 channels (bwd data: flipped) ----
    // filter taps [27][64] stay in LDS behind the ring (bwd data: flipped); every lane group reads its quad as a broadcast --
    // 108 more VGPRs for register-resident taps spill this kernel
    float* wl = smem + 3 * PLANE;
    if constexpr (MODE != M_BWD_WEIGHT) {
        for (int i = tid; i < 27 * 64; i += 512) {
            const int cc = i / 27, tap = i % 27;
            wl[(MODE == M_BWD_DATA ? 26 - tap : tap) * 64 + cc] = a.w[(long)c0 * 27 + i];
        }
    }
    f32x2 wacc[MODE == M_BWD_WEIGHT ? 27 : 1][2];
    if constexpr (MODE == M_BWD_WEIGHT) {
#pragma unroll
        for (int t = 0; t < 27; ++t) { wacc[t][0] = f32x2{0.f, 0.f}; wacc[t][1] = f32x2{0.f, 0.f}; }
    }
    float a4[4] = {1.f, 1.f, 1.f, 1.f}, b4[4] = {0.f, 0.f, 0.f, 0.f};      // bwd data: bn1 scale/shift of the mask
    if (MODE == M_BWD_DATA) {
        const float4 s = *(const float4*)(a.sc + c), h = *(const float4*)(a.sh + c);
        a4[0] = s.x; a4[1] = s.y; a4[2] = s.z; a4[3] = s.w; b4[0] = h.x; b4[1] = h.y; b4[2] = h.z; b4[3] = h.w;
    }
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};

    // ---- prologue: planes t0-1 and t0 parked, t0+1 in flight ----
    const int ho = h0 + row, wo0 = w0 + cg * 4;
    const bool row_ok = ho < g.H;
    // side inputs of an output plane (global, 8 B per column): bwd data: x (needed after the taps);  wgrad: gout (needed BEFORE the
    // taps -- so the weight gradient fetches them one plane ahead, the first ones together with the prologue planes)
    constexpr bool SD2 = BNG && MODE == M_BWD_WEIGHT;     // the side input (gout at the output position) is formed from two tensors
    auto side_fetch = [&](int t, uint2 (&sd)[4], uint2 (&sx)[SD2 ? 4 : 1]) {      // unconditional, from the clamped position (see fetch)
        const long ob = (((long)n * g.T + t) * g.H + min(ho, g.H - 1)) * (long)g.W * g.C + c;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const long oc = (long)min(wo0 + j, g.W - 1) * g.C;
            sd[j] = *(const uint2*)(a.aux + ob + oc);
            if constexpr (SD2) sx[j] = *(const uint2*)(a.xu + ob + oc);
        }
    };
    uint2 side_nx[4] = {make_uint2(0, 0), make_uint2(0, 0), make_uint2(0, 0), make_uint2(0, 0)};
    uint2 sidx_nx[SD2 ? 4 : 1];
    if constexpr (MODE == M_BWD_WEIGHT) side_fetch(t0, side_nx, sidx_nx);
    park(t0 - 1, regs_a, regx_a);
    park(t0, regs_b, regx_b);
    for (int t = t0; t < t1; ++t) {
        park(t + 1, regs, regx);
        if (t + 1 < t1) fetch(t + 2, regs, regx);
        uint2 side[4];
        uint2 sidx[SD2 ? 4 : 1];
        const long obase = (((long)n * g.T + t) * g.H + ho) * (long)g.W * g.C + c;
        if constexpr (MODE == M_BWD_WEIGHT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { side[j] = side_nx[j]; if constexpr (SD2) sidx[j] = sidx_nx[j]; }
            if (t + 1 < t1) side_fetch(t + 1, side_nx, sidx_nx);
        } else if (MODE != M_FWD) {
            side_fetch(t, side, sidx);
        }
        __syncthreads();
        f32x2 acc[4][2];
#pragma unroll
        for (int j = 0; j < 4; ++j) { acc[j][0] = f32x2{0.f, 0.f}; acc[j][1] = f32x2{0.f, 0.f}; }
        f32x2 gv[4][2];
        if constexpr (MODE == M_BWD_WEIGHT) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bf16x4 v = as_bf16x4(side[j]);
                if constexpr (SD2) {
                    const bool ok = row_ok && wo0 + j < g.W;      // zero gradient outside the volume
                    const bf16x4 u = as_bf16x4(sidx[j]);
                    float q[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) q[e] = ok ? fmaf(kA[e], bf2f(v[e]), fmaf(kB[e], bf2f(u[e]), kC[e])) : 0.f;
                    gv[j][0] = f32x2{q[0], q[1]};
                    gv[j][1] = f32x2{q[2], q[3]};
                } else {
                    const bool ok = row_ok && wo0 + j < g.W;      // zero gradient outside the volume
                    gv[j][0] = f32x2{ok ? bf2f(v[0]) : 0.f, ok ? bf2f(v[1]) : 0.f};
                    gv[j][1] = f32x2{ok ? bf2f(v[2]) : 0.f, ok ? bf2f(v[3]) : 0.f};
                }
            }
        }
        auto tap_plane = [&](int dt) {
            const float* pl = smem + ((t + dt - 1 + 3) % 3) * PLANE;
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) {
                const float* rp = pl + ((row + dh) * PW + cg * 4) * 64 + cl * 4;
                f32x2 in[6][2];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    const float4 v = *(const float4*)(rp + i * 64);
                    in[i][0] = f32x2{v.x, v.y};
                    in[i][1] = f32x2{v.z, v.w};
                }
                f32x2 wt2[3][2];
                if constexpr (MODE != M_BWD_WEIGHT) {
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw) {
                        const float4 v = *(const float4*)(wl + ((dt * 3 + dh) * 3 + dw) * 64 + cl * 4);
                        wt2[dw][0] = f32x2{v.x, v.y};
                        wt2[dw][1] = f32x2{v.z, v.w};
                    }
                }
#pragma unroll
                for (int dw = 0; dw < 3; ++dw) {
                    const int tap = (dt * 3 + dh) * 3 + dw;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if constexpr (MODE == M_BWD_WEIGHT) {
                            wacc[tap][0] = gv[j][0] * in[j + dw][0] + wacc[tap][0];
                            wacc[tap][1] = gv[j][1] * in[j + dw][1] + wacc[tap][1];
                        } else {
                            acc[j][0] = in[j + dw][0] * wt2[dw][0] + acc[j][0];
                            acc[j][1] = in[j + dw][1] * wt2[dw][1] + acc[j][1];
                        }
                    }
                }
            }
                };
        if constexpr (MODE == M_BWD_WEIGHT) {         // accumulators are indexed by the tap: needs compile-time dt
            tap_plane(0); tap_plane(1); tap_plane(2);
        } else {
#pragma unroll 1                                      // one temporal tap per pass keeps the live LDS reads at 18 vectors (no spills)
            for (int dt = 0; dt < 3; ++dt) tap_plane(dt);
        }
        if constexpr (MODE != M_BWD_WEIGHT) if (row_ok) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (wo0 + j >= g.W) continue;
                float v[4] = {acc[j][0][0], acc[j][0][1], acc[j][1][0], acc[j][1][1]};
                bf16x4 o;
                if constexpr (MODE == M_FWD) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { o[e] = f2bf(v[e]); s0[e] += v[e]; s1[e] += v[e] * v[e]; }
                } else {
                    const bf16x4 xv = as_bf16x4(side[j]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float xf = bf2f(xv[e]);
                        const float d = fmaf(xf, a4[e], b4[e]) > 0.f ? v[e] : 0.f;
                        o[e] = f2bf(d);
                        s0[e] += d; s1[e] += d * xf;
                    }
                }
                *(uint2*)(a.out + obase + (long)(wo0 + j) * g.C) = as_uint2(o);
            }
        }
        __syncthreads();                              // ring slot (t-1) mod 3 is overwritten by the next park
    }

    // ---- workgroup reduction over the 32 position slots ----
    float* red = smem;                                // the ring is dead now
    if constexpr (MODE == M_BWD_WEIGHT) {
        // [27][64] partial: lanes sharing cl within a wave by shuffles (slot bits 0..1 = lane bits 4..5), waves via LDS
        const int wave = tid >> 6;
#pragma unroll
        for (int t = 0; t < 27; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = wacc[t][e >> 1][e & 1];
                v = xor32_sum(xor16_sum(v));
                if ((tid & 63) < 16) red[(wave * 27 + t) * 64 + cl * 4 + e] = v;
            }
        __syncthreads();
        for (int i = tid; i < 27 * 64; i += 512) {
            float s = 0.f;
#pragma unroll
            for (int wv8 = 0; wv8 < 8; ++wv8) s += red[wv8 * 27 * 64 + i];
            const int tap = i >> 6, cc = i & 63;
            a.P[((long)bx * 27 + tap) * g.C + c0 + cc] = s;
        }
    } else if (a.st0) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[slot * 64 + cl * 4 + e] = s0[e]; red[(32 + slot) * 64 + cl * 4 + e] = s1[e]; }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, cc = tid & 63;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 32; ++k) s += red[(which * 32 + k) * 64 + cc];
            (which ? a.st1 : a.st0)[(long)bx * g.C + c0 + cc] = s;
        }
    }
}

template <int MODE, bool BNG = false, bool ACT = true>
// Workgroup -> tile: the hardware deals consecutive workgroups round-robin over the 8 XCDs, so with bx = blockIdx.x the spatial neighbours
// of a tile (which share its halo columns / rows) sit on 8 different L2s and every XCD fetches its own copy of every halo from memory.
// xcd_remap gives each XCD a CONTIGUOUS range of tile ids -- whole planes of spatially adjacent tiles -- so halos are L2 hits.  The tile id
// also indexes the statistics rows / partial blocks, so results do not depend on the mapping.
__global__ __launch_bounds__(512) void dwconv_tile_kernel(TileArgs a) { dwconv_tile_body<MODE, BNG, ACT>(a, xcd_remap(blockIdx.x, gridDim.x), blockIdx.y); }
// the forward conv that also finalises the BatchNorm in front of it (TileArgs: FIN)
__global__ __launch_bounds__(512) void dwconv_tile_fwd_fin_kernel(TileArgs a) { dwconv_tile_body<M_FWD, false, true, true>(a, xcd_remap(blockIdx.x, gridDim.x), blockIdx.y); }

// ---------------------------------------------------------------------------------------------------------------------------------
// Round 4: data gradient AND weight gradient of a stride-1 depthwise conv from ONE staged ring (with the BatchNorm backward of the
// layer above folded into the gradient operand).  Both are sums over the same pairs of positions
//     dx[p]   = sum_off w[off] * g[p - off]            (the data gradient: flipped taps over the ring of g = dc3 with halo)
//     dW[off] = sum_p   a[p]   * g[p - off]            (a = relu(bn1(x)) at the CENTRE position p, g from the same ring)
// so a workgroup that has parked g (formed from dzu and xu) with halo and holds x at its own output positions -- the mask operand it
// needs anyway -- has every operand of both.  The weight-gradient grid of round 3's two-grid launch (which staged x with halo and
// re-read dzu / xu at the centre: 7 tensor passes, 2.26x the algorithmic traffic by PMC, two rounds of workgroups in layer3) is gone:
// reads dzu and xu with halo + x, writes dz = the 4 passes of a depthwise backward.  g is zero outside the volume (parked as zero)
// and a is cleared for the overhang positions of a tile, so every (p, p - off) pair inside the volume is counted exactly once.
//
// Thread map: 27 x 4 weight-gradient accumulators per thread (the 4-channel x 4-column map of the kernels above) plus the data
// gradient's state does not fit 256 VGPRs (415 spilled).  Here a thread owns 2 channels x 8 consecutive columns of one tile row:
// 54 accumulators, ds_read_b64 operand reads (32 lanes x 8 B = all 64 banks once), packed FMAs on the channel pair.
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void dwconv_tile_bwd_both_kernel(TileArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];        // ring[3][PLANE] | taps [27][64] | coef [3][64]
    const TileGeom g = a.g;
    const int bx = xcd_remap(blockIdx.x, gridDim.x), by = blockIdx.y;      // XCD-contiguous tile ids (see dwconv_tile_kernel)
    const int tid = threadIdx.x;
    const int cp = tid & 31, slot = tid >> 5, row = slot >> 1, cg = slot & 1;      // channel pair, tile row, 8-column half
    const int c0 = by * 64, c = c0 + cp * 2;
    int b = bx;
    const int wt = b % g.wtiles; b /= g.wtiles;
    const int ht = b % g.htiles; b /= g.htiles;
    const int tk = b % g.tchunks; const int n = b / g.tchunks;
    const int h0 = ht * TH, w0 = wt * TW;
    const int t0 = tk * g.tc, t1 = min(g.T, t0 + g.tc);

    // ---- staging slots: thread -> (position, channel quad) of a plane, loop invariant over t (as in dwconv_tile_body) ----
    int s_off[NLD];
    bool s_ok[NLD];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int idx = tid + 512 * i;
        const int pos = idx >> 4, q = idx & 15;
        const int pr = pos / PW, pc = pos % PW;
        const int hi = h0 - 1 + pr, wi = w0 - 1 + pc;
        s_ok[i] = pos < NPOS && hi >= 0 && hi < g.H && wi >= 0 && wi < g.W;
        s_off[i] = s_ok[i] ? (hi * g.W + wi) * g.C + c0 + q * 4 : 0;
    }
    const long plane_elems = (long)g.H * g.W * g.C;
    const bf16* in_n = a.in + (long)n * g.T * plane_elems;
    const bf16* in2_n = a.xu + (long)n * g.T * plane_elems;
    uint2 regs[NLD], regs_a[NLD], regs_b[NLD], regx[NLD], regx_a[NLD], regx_b[NLD];
    auto fetch = [&](int t, uint2 (&rg)[NLD], uint2 (&rx)[NLD]) {          // unconditional loads (see dwconv_tile_body)
        const bool tok = t >= 0 && t < g.T;
        const bf16* p = in_n + (long)(tok ? t : 0) * plane_elems;
        const bf16* p2 = in2_n + (long)(tok ? t : 0) * plane_elems;
#pragma unroll
        for (int i = 0; i < NLD; ++i) rg[i] = *(const uint2*)(p + s_off[i]);
#pragma unroll
        for (int i = 0; i < NLD; ++i) rx[i] = *(const uint2*)(p2 + s_off[i]);
    };
    auto park = [&](int t, const uint2 (&rg)[NLD], const uint2 (&rx)[NLD]) {      // g = cA*dzu + cB*xu + cC, fp32, zero outside the volume
        const bool tok = t >= 0 && t < g.T;
        float* dst = smem + ((t + 3) % 3) * PLANE;
        const float* coef = smem + 3 * PLANE + 27 * 64 + (tid & 15) * 4;       // this thread's channel quad (the same for all its slots)
        const float4 cA = *(const float4*)coef, cB = *(const float4*)(coef + 64), cC = *(const float4*)(coef + 128);
        const float kA[4] = {cA.x, cA.y, cA.z, cA.w}, kB[4] = {cB.x, cB.y, cB.z, cB.w}, kC[4] = {cC.x, cC.y, cC.z, cC.w};
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (i == NLD - 1 && tid + 512 * i >= NPOS * 16) continue;
            const bf16x4 v = as_bf16x4(rg[i]), u = as_bf16x4(rx[i]);
            const bool ok = tok && s_ok[i];
            float4 o;
            o.x = ok ? fmaf(kA[0], bf2f(v[0]), fmaf(kB[0], bf2f(u[0]), kC[0])) : 0.f;
            o.y = ok ? fmaf(kA[1], bf2f(v[1]), fmaf(kB[1], bf2f(u[1]), kC[1])) : 0.f;
            o.z = ok ? fmaf(kA[2], bf2f(v[2]), fmaf(kB[2], bf2f(u[2]), kC[2])) : 0.f;
            o.w = ok ? fmaf(kA[3], bf2f(v[3]), fmaf(kB[3], bf2f(u[3]), kC[3])) : 0.f;
            *(float4*)(dst + 4 * (tid + 512 * i)) = o;              // (position * 64 + quad * 4 == 4 * slot index)
        }
    };
    // ---- the SMALL loads go out first: the partial-statistics rows of the coefficient derivation (16 channel quads x 32 row groups,
    // four rows per thread in flight together) and the filter taps.  Vector loads complete in order, so issued ahead of the three
    // plane fetches they are waited for alone (an L2 round trip) and the derivation -- three barriers and two fp64 LDS reductions --
    // runs UNDER the planes' latency; issued behind them (as until round 4) it started when the last plane had landed, one row trip
    // after the other: 2 - 4 us of every workgroup's 16 - 35 us life
    const int dq = tid & 15, drg = tid >> 4;
    const float* const dp0 = a.bst0 + c0 + dq * 4;
    const float* const dp1 = a.bst1 + c0 + dq * 4;
    float4 du[4], dv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long rr = min(drg + 32 * k, a.bR - 1);
        du[k] = *(const float4*)(dp0 + rr * g.C); dv[k] = *(const float4*)(dp1 + rr * g.C);
    }
    float wreg[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) wreg[k] = a.w[(long)c0 * 27 + min(tid + 512 * k, 27 * 64 - 1)];
    const int dch = c0 + (tid & 63);
    const float d_mu = a.bmean[dch], d_rr = a.binvstd[dch], d_gm = a.bgamma[dch];      // used by threads 0 .. 63 below
    const float2 sc2 = *(const float2*)(a.sc + c), sh2 = *(const float2*)(a.sh + c);      // bn1 scale / shift of the channel pair
    fetch(t0 - 1, regs_a, regx_a);
    fetch(t0, regs_b, regx_b);
    fetch(t0 + 1, regs, regx);

    // ---- x at this thread's output positions (mask + statistics operand of the data gradient, activation operand of the weight
    // gradient): 4 B per column, fetched one plane ahead, the first ones together with the prologue planes ----
    const int ho = h0 + row, wo0 = w0 + cg * 8;
    const bool row_ok = ho < g.H;
    auto side_fetch = [&](int t, uint32_t (&sd)[8]) {          // unconditional, from the clamped position
        const long ob = (((long)n * g.T + t) * g.H + min(ho, g.H - 1)) * (long)g.W * g.C + c;
#pragma unroll
        for (int j = 0; j < 8; ++j) sd[j] = *(const uint32_t*)(a.aux + ob + (long)min(wo0 + j, g.W - 1) * g.C);
    };
    uint32_t side_nx[8];
    side_fetch(t0, side_nx);

    // ---- coefficients of the BatchNorm backward above, derived under the latency of the fetches just issued (identical arithmetic
    // to dwconv_tile_body<., BNG>: dgamma / dbeta and the coefficients are bit-identical) ----
    {
        double* red = (double*)smem;                     // [2][32][64] in the (still empty) ring
        float* coef = smem + 3 * PLANE + 27 * 64;        // [3][64] behind the filter taps
        {
            const int q = dq, rg = drg;
            double sa[4] = {0, 0, 0, 0}, sb[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; ++k) {               // rows beyond bR were read from row bR - 1 and are added as zero
                const bool ok = rg + 32 * k < a.bR;
                sa[0] += ok ? du[k].x : 0.f; sa[1] += ok ? du[k].y : 0.f; sa[2] += ok ? du[k].z : 0.f; sa[3] += ok ? du[k].w : 0.f;
                sb[0] += ok ? dv[k].x : 0.f; sb[1] += ok ? dv[k].y : 0.f; sb[2] += ok ? dv[k].z : 0.f; sb[3] += ok ? dv[k].w : 0.f;
            }
            for (int r = rg + 128; r < a.bR; r += 32) {      // longer lists than the model's shapes produce
                const float4 u = *(const float4*)(dp0 + (long)r * g.C), v = *(const float4*)(dp1 + (long)r * g.C);
                sa[0] += u.x; sa[1] += u.y; sa[2] += u.z; sa[3] += u.w;
                sb[0] += v.x; sb[1] += v.y; sb[2] += v.z; sb[3] += v.w;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { red[(0 * 32 + rg) * 64 + q * 4 + e] = sa[e]; red[(1 * 32 + rg) * 64 + q * 4 + e] = sb[e]; }
        }
        __syncthreads();
        double* red2 = red + 2 * 32 * 64;                // [2][4][64]
        {
            const int which = tid >> 8, part = (tid >> 6) & 3, ch = tid & 63;
            double s = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += red[(which * 32 + part * 8 + k) * 64 + ch];
            red2[(which * 4 + part) * 64 + ch] = s;
        }
        __syncthreads();
        if (tid < 64) {
            const double sa = (red2[(0 * 4 + 0) * 64 + tid] + red2[(0 * 4 + 1) * 64 + tid]) + (red2[(0 * 4 + 2) * 64 + tid] + red2[(0 * 4 + 3) * 64 + tid]);
            const double sb = (red2[(1 * 4 + 0) * 64 + tid] + red2[(1 * 4 + 1) * 64 + tid]) + (red2[(1 * 4 + 2) * 64 + tid] + red2[(1 * 4 + 3) * 64 + tid]);
            const int cc = c0 + tid;
            const double mu = d_mu, rr = d_rr, gm = d_gm;
            const double sum_dz = sa, sum_dz_xhat = (sb - mu * sa) * rr;
            const double m1 = sum_dz / a.bcount, m2 = sum_dz_xhat / a.bcount;
            coef[0 * 64 + tid] = (float)(gm * rr);
            coef[1 * 64 + tid] = (float)(-gm * rr * rr * m2);
            coef[2 * 64 + tid] = (float)(gm * rr * rr * m2 * mu - gm * rr * m1);
            if (bx == 0 && a.bdgamma) {
                a.bdgamma[cc] += (float)sum_dz_xhat;
                a.bdbeta[cc] += (float)sum_dz;
            }
        }
        __syncthreads();                                  // the ring is written next (and the coefficients read) by every thread
    }

    // ---- flipped filter taps [27][64] in LDS behind the ring ----
    float* wl = smem + 3 * PLANE;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = tid + 512 * k;
        if (i < 27 * 64) wl[(26 - i % 27) * 64 + i / 27] = wreg[k];
    }
    f32x2 wacc[27];
    f32x2 s0 = f32x2{0.f, 0.f}, s1 = f32x2{0.f, 0.f};

    park(t0 - 1, regs_a, regx_a);
    park(t0, regs_b, regx_b);
    // the 54 accumulators come to life HERE, behind the prologue's 72 prefetch registers (an ordinary zero initialisation is hoisted to
    // the top of the kernel, and the allocator then keeps the prefetched planes in scratch memory: every load waited for its store)
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        float z0, z1;
        asm volatile("v_mov_b32 %0, 0\n\tv_mov_b32 %1, 0" : "=v"(z0), "=v"(z1));
        wacc[k] = f32x2{z0, z1};
    }
    for (int t = t0; t < t1; ++t) {
        park(t + 1, regs, regx);
        if (t + 1 < t1) fetch(t + 2, regs, regx);
        uint32_t side[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) side[j] = side_nx[j];
        if (t + 1 < t1) side_fetch(t + 1, side_nx);
        const long obase = (((long)n * g.T + t) * g.H + ho) * (long)g.W * g.C + c;
        __syncthreads();
        f32x2 acc[8], av[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[j] = f32x2{0.f, 0.f};
            const bf16x2 x2 = __builtin_bit_cast(bf16x2, side[j]);
            const bool ok = row_ok && wo0 + j < g.W;      // nothing outside the volume enters the weight gradient
            av[j] = f32x2{ok ? fmaxf(fmaf(bf2f(x2[0]), sc2.x, sh2.x), 0.f) : 0.f, ok ? fmaxf(fmaf(bf2f(x2[1]), sc2.y, sh2.y), 0.f) : 0.f};
        }
        // one temporal tap per pass (a real loop: unrolled over dt the compiler keeps far more operand reads in flight than 256 VGPRs
        // hold -- 255 spilled).  The weight-gradient accumulators are indexed by the tap, so a pass collects its 9 taps in a
        // zero-initialised group w9 and folds it into the persistent group of its dt: taps (2 - dt) * 9 + (8 - k9).
#pragma unroll 1
        for (int dt = 0; dt < 3; ++dt) {
            const float* pl = smem + ((t + dt - 1 + 3) % 3) * PLANE;
            const float* wd = wl + dt * 9 * 64 + cp * 2;
            f32x2 w9[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) w9[k] = f32x2{0.f, 0.f};
#pragma unroll
            for (int dh = 0; dh < 3; ++dh) {
                const float* rp = pl + ((row + dh) * PW + cg * 8) * 64 + cp * 2;
                f32x2 in[10];
#pragma unroll
                for (int i = 0; i < 10; ++i) { const float2 v = *(const float2*)(rp + i * 64); in[i] = f32x2{v.x, v.y}; }
                f32x2 w3[3];
#pragma unroll
                for (int dw = 0; dw < 3; ++dw) { const float2 wv = *(const float2*)(wd + (dh * 3 + dw) * 64); w3[dw] = f32x2{wv.x, wv.y}; }
                // column-major issue order: consecutive FMAs go to different accumulators (acc[j] / the three taps of this row), so no
                // dependent pair is back to back (tap-major order compiled to 8-long dependent chains with a wait state per link)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
#pragma unroll
                    for (int dw = 0; dw < 3; ++dw) {
                        const int k9 = dh * 3 + dw;
                        acc[j] = in[j + dw] * w3[dw] + acc[j];           // ring position p + d  <->  filter offset -d ...
                        w9[8 - k9] = av[j] * in[j + dw] + w9[8 - k9];    // ... i.e. weight-gradient tap 26 - tap = (2 - dt) * 9 + (8 - k9)
                    }
                }
            }
            // the group of this pass is always the LAST register group; the three groups rotate by one per pass, so after the three passes
            // of a plane every group is back in its place (a branch on dt that adds into one of three groups makes the register
            // allocator keep copies of all 27 accumulators across the join: 100+ spills)
#pragma unroll
            for (int k = 0; k < 9; ++k) { const f32x2 g2 = wacc[18 + k] + w9[k]; wacc[18 + k] = wacc[9 + k]; wacc[9 + k] = wacc[k]; wacc[k] = g2; }
        }
        if (row_ok) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (wo0 + j >= g.W) continue;
                const bf16x2 x2 = __builtin_bit_cast(bf16x2, side[j]);
                const float x0 = bf2f(x2[0]), x1 = bf2f(x2[1]);
                const float d0 = fmaf(x0, sc2.x, sh2.x) > 0.f ? acc[j][0] : 0.f;
                const float d1 = fmaf(x1, sc2.y, sh2.y) > 0.f ? acc[j][1] : 0.f;
                bf16x2 o;
                o[0] = f2bf(d0); o[1] = f2bf(d1);
                s0 = s0 + f32x2{d0, d1};
                s1 = s1 + f32x2{d0 * x0, d1 * x1};
                *(uint32_t*)(a.out + obase + (long)(wo0 + j) * g.C) = __builtin_bit_cast(uint32_t, o);
            }
        }
        __syncthreads();                              // ring slot (t-1) mod 3 is overwritten by the next park
    }

    // ---- workgroup reductions (the ring is dead now): [27][64] weight-gradient block, then the statistics rows ----
    float* red = smem;
    {
        const int wave = tid >> 6;
#pragma unroll
        for (int t = 0; t < 27; ++t)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const float v = xor32_sum(wacc[t][e]);          // the two slots of a wave (lane bit 5) share the channel pair
                if ((tid & 63) < 32) red[(wave * 27 + t) * 64 + cp * 2 + e] = v;
            }
    }
    float* red_s = smem + 8 * 27 * 64;
    red_s[slot * 64 + cp * 2] = s0[0]; red_s[slot * 64 + cp * 2 + 1] = s0[1];
    red_s[(16 + slot) * 64 + cp * 2] = s1[0]; red_s[(16 + slot) * 64 + cp * 2 + 1] = s1[1];
    __syncthreads();
    for (int i = tid; i < 27 * 64; i += 512) {
        float s = 0.f;
#pragma unroll
        for (int wv8 = 0; wv8 < 8; ++wv8) s += red[wv8 * 27 * 64 + i];
        const int tap = i >> 6, cc = i & 63;
        a.P[((long)bx * 27 + tap) * g.C + c0 + cc] = s;
    }
    if (tid < 128) {
        const int which = tid >> 6, cc = tid & 63;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += red_s[(which * 16 + k) * 64 + cc];
        (which ? a.st1 : a.st0)[(long)bx * g.C + c0 + cc] = s;
    }
}

TileGeom make_geom(int N, int T, int H, int W, int C, bool wgrad = false) {
    TileGeom g;
    g.N = N; g.T = T; g.H = H; g.W = W; g.C = C;
    g.htiles = (H + TH - 1) / TH; g.wtiles = (W + TW - 1) / TW;
    const long base = (long)N * g.htiles * g.wtiles * (C / 64);
    // planes per workgroup: one workgroup is resident per CU (145 KB of LDS), so the kernel runs in ceil(WGs / 256) rounds of
    // (tc + 2) plane steps (2 = halo planes staged before the first output); pick the tc that minimises rounds x steps
    // (layer1: tc = 4 -> 768 workgroups = 3 full rounds of 6 steps instead of 576 = 2.25 -> 3 rounds of 8)
    long best_cost = -1, tc = 1;
    for (long c = 1; c <= T; ++c) {
        const long chunks = (T + c - 1) / c;
        const long rounds = (base * chunks + 255) / 256;
        const long cost = rounds * (c + 2);
        // ties: forward / data gradient take the split with more workgroups, the weight gradient the one with fewer (each of
        // its workgroups ends with a 108-accumulator reduction and a 6.9 KB partial)
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && wgrad)) { best_cost = cost; tc = c; }
    }
    g.tc = (int)tc;
    g.tchunks = (T + g.tc - 1) / g.tc;
    return g;
}

template <int MODE, bool BNG = false>
int launch_tile(TileArgs& a, hipStream_t stream) {
    const TileGeom& g = a.g;
    dim3 grid(g.N * g.tchunks * g.htiles * g.wtiles, g.C / 64), block(512);
    const size_t lds = (3 * PLANE + 27 * 64 + 3 * 64) * sizeof(float);       // ring + filter taps + BNG coefficients (wgrad reuses the ring for its reduction)
    static LdsOptIn opt;                                   // (one per instantiation) > 64 KB of dynamic LDS needs the opt-in once per kernel and device
    TUBER_LDS_OPT_IN(opt, (dwconv_tile_kernel<MODE, BNG>), lds);
    hipLaunchKernelGGL((dwconv_tile_kernel<MODE, BNG>), grid, block, lds, stream, a);
    TUBER_RETURN_LAUNCH();
}

}  // namespace

extern "C" {

// rows of partial statistics (forward, data gradient) = workgroups along x = partial blocks of the weight gradient
int tuber_dwconv_tile_blocks(int N, int T, int H, int W, int C) {
    const TileGeom g = make_geom(N, T, H, W, C);
    return g.N * g.tchunks * g.htiles * g.wtiles;
}

// partial blocks of the weight gradient (its own plane chunking)
int tuber_dwconv_tile_wgrad_blocks(int N, int T, int H, int W, int C) {
    const TileGeom g = make_geom(N, T, H, W, C, true);
    return g.N * g.tchunks * g.htiles * g.wtiles;
}

int tuber_dwconv_tile_fwd(const void* x, const float* sc, const float* sh, const float* w, void* out, float* st0, float* st1,
                          int N, int T, int H, int W, int C, hipStream_t stream) {
    if ((C & 63) || (sc == nullptr) != (sh == nullptr)) return TUBER_EINVAL;
    TileArgs a{};
    a.in = (const bf16*)x; a.sc = sc; a.sh = sh; a.w = w; a.out = (bf16*)out; a.st0 = st0; a.st1 = st1;
    a.g = make_geom(N, T, H, W, C);
    if (!sc) {                                   // plain conv (no BatchNorm + ReLU in front): its own instantiation
        const TileGeom& g = a.g;
        const size_t lds = (3 * PLANE + 27 * 64 + 3 * 64) * sizeof(float);
        static LdsOptIn opt;
        TUBER_LDS_OPT_IN(opt, (dwconv_tile_kernel<M_FWD, false, false>), lds);
        hipLaunchKernelGGL((dwconv_tile_kernel<M_FWD, false, false>), dim3(g.N * g.tchunks * g.htiles * g.wtiles, g.C / 64), dim3(512), lds, stream, a);
        TUBER_RETURN_LAUNCH();
    }
    return launch_tile<M_FWD>(a, stream);
}

// tuber_dwconv_tile_fwd with the training-mode BatchNorm in front of it (bn1) finalised INSIDE the launch: pst0 / pst1 = the producing
// conv's R partial rows [R][C] (sum x, sum x^2), count = rows behind them.  Writes scale / shift / mean / invstd [C] and updates
// rmean / rvar / nbt (NULL: no running statistics) exactly as tuber_bn_finalize does -- that launch disappears from the forward chain.
int tuber_dwconv_tile_fwd_bn(const void* x, const float* pst0, const float* pst1, int R, float count, const float* gamma, const float* beta,
                             float* rmean, float* rvar, long long* nbt, float momentum, float eps, float* scale, float* shift,
                             float* mean, float* invstd, const float* w, void* out, float* st0, float* st1,
                             int N, int T, int H, int W, int C, hipStream_t stream) {
    if ((C & 63) || R <= 0 || !pst0 || !pst1 || !gamma || !beta || !scale || !shift || !mean || !invstd || (rmean == nullptr) != (rvar == nullptr))
        return TUBER_EINVAL;
    TileArgs a{};
    a.in = (const bf16*)x; a.w = w; a.out = (bf16*)out; a.st0 = st0; a.st1 = st1;
    a.bst0 = pst0; a.bst1 = pst1; a.bR = R; a.bcount = count; a.bgamma = gamma;
    a.fbeta = beta; a.frmean = rmean; a.frvar = rvar; a.fnbt = nbt; a.fmom = momentum; a.feps = eps;
    a.fscale = scale; a.fshift = shift; a.fmean = mean; a.finvstd = invstd;
    a.g = make_geom(N, T, H, W, C);
    const TileGeom& g = a.g;
    const size_t lds = (3 * PLANE + 27 * 64 + 3 * 64) * sizeof(float);
    static LdsOptIn opt;
    TUBER_LDS_OPT_IN(opt, dwconv_tile_fwd_fin_kernel, lds);
    hipLaunchKernelGGL(dwconv_tile_fwd_fin_kernel, dim3(g.N * g.tchunks * g.htiles * g.wtiles, g.C / 64), dim3(512), lds, stream, a);
    TUBER_RETURN_LAUNCH();
}

int tuber_dwconv_tile_bwd_data(const void* gout, const float* w, const void* x, const float* sc, const float* sh, void* dz,
                               float* st0, float* st1, int N, int T, int H, int W, int C, hipStream_t stream) {
    if ((C & 63) || !sc || !sh) return TUBER_EINVAL;
    TileArgs a{};
    a.in = (const bf16*)gout; a.sc = sc; a.sh = sh; a.w = w; a.out = (bf16*)dz; a.aux = (const bf16*)x; a.st0 = st0; a.st1 = st1;
    a.g = make_geom(N, T, H, W, C);
    return launch_tile<M_BWD_DATA>(a, stream);
}

// partial must hold tuber_dwconv_tile_wgrad_blocks * 27 * C floats; dw is the [C][27] fp32 weight gradient
int tuber_dwconv_tile_bwd_weight(const void* gout, const void* x, const float* sc, const float* sh, float* partial, float* dw,
                                 int accumulate, int N, int T, int H, int W, int C, hipStream_t stream) {
    if ((C & 63) || !sc || !sh) return TUBER_EINVAL;
    TileArgs a{};
    a.in = (const bf16*)x; a.sc = sc; a.sh = sh; a.aux = (const bf16*)gout; a.P = partial;
    a.g = make_geom(N, T, H, W, C, true);
    const int rc = launch_tile<M_BWD_WEIGHT>(a, stream);
    if (rc || accumulate == 2) return rc;      // accumulate == 2: partial blocks reduced later by tuber_multi_reduce
    return tuber_dw_wgrad_reduce(partial, dw, a.g.N * a.g.tchunks * a.g.htiles * a.g.wtiles, C, accumulate, stream);
}

// The two backward kernels with the BatchNorm backward of the layer ABOVE the depthwise conv (bn3) folded into their gradient
// operand: instead of a finished dc3 they take bn3's masked output gradient dzu [M, C], bn3's input xu = c3 [M, C] and the R <= 128
// partial rows (sum dz, sum dz*x) the producer of dzu wrote, derive cA / cB / cC for their 64 channels and form
// dc3 = cA*dzu + cB*xu + cC (fp32) on load.  The stand-alone tuber_bn_bwd_fa launch and the dc3 tensor disappear.
// dgamma / dbeta of bn3 (accumulated, +=) are written by the data-gradient kernel unless NULL.
int tuber_dwconv_tile_bwd_data_bn(const void* dzu, const void* xu, const float* bst0, const float* bst1, int R, float count,
                                  const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                  const float* w, const void* x, const float* sc, const float* sh, void* dz,
                                  float* st0, float* st1, int N, int T, int H, int W, int C, hipStream_t stream) {
    if ((C & 63) || !sc || !sh || R <= 0 || R > 128 || !bst0 || !bst1) return TUBER_EINVAL;
    TileArgs a{};
    a.in = (const bf16*)dzu; a.xu = (const bf16*)xu; a.sc = sc; a.sh = sh; a.w = w; a.out = (bf16*)dz; a.aux = (const bf16*)x;
    a.st0 = st0; a.st1 = st1;
    a.bst0 = bst0; a.bst1 = bst1; a.bR = R; a.bcount = count; a.bgamma = gamma; a.bmean = mean; a.binvstd = invstd;
    a.bdgamma = dgamma; a.bdbeta = dbeta;
    a.g = make_geom(N, T, H, W, C);
    return launch_tile<M_BWD_DATA, true>(a, stream);
}

int tuber_dwconv_tile_bwd_weight_bn(const void* dzu, const void* xu, const float* bst0, const float* bst1, int R, float count,
                                    const float* gamma, const float* mean, const float* invstd,
                                    const void* x, const float* sc, const float* sh, float* partial, float* dw, int accumulate,
                                    int N, int T, int H, int W, int C, hipStream_t stream) {
    if ((C & 63) || !sc || !sh || R <= 0 || R > 128 || !bst0 || !bst1) return TUBER_EINVAL;
    TileArgs a{};
    a.in = (const bf16*)x; a.sc = sc; a.sh = sh; a.aux = (const bf16*)dzu; a.xu = (const bf16*)xu; a.P = partial;
    a.bst0 = bst0; a.bst1 = bst1; a.bR = R; a.bcount = count; a.bgamma = gamma; a.bmean = mean; a.binvstd = invstd;
    a.g = make_geom(N, T, H, W, C, true);
    const int rc = launch_tile<M_BWD_WEIGHT, true>(a, stream);
    if (rc || accumulate == 2) return rc;
    return tuber_dw_wgrad_reduce(partial, dw, a.g.N * a.g.tchunks * a.g.htiles * a.g.wtiles, C, accumulate, stream);
}

// Data and weight gradient of the same conv in ONE launch and ONE pass over the operands (dwconv_tile_bwd_both_kernel): dz and
// dgamma / dbeta bit-identical to tuber_dwconv_tile_bwd_data_bn; the statistics rows and the weight gradient are the same sums in another
// order (other thread map; the weight-gradient products are grouped by the position of the ACTIVATION, not of the gradient), i.e. equal to
// the two-launch form up to fp32 rounding.  `partial` receives tuber_dwconv_tile_blocks(N, T, H, W, C) blocks of [27][C]
// (NOT ..._wgrad_blocks), reduced by the caller.
int tuber_dwconv_tile_bwd_both_bn(const void* dzu, const void* xu, const float* bst0, const float* bst1, int R, float count,
                                  const float* gamma, const float* mean, const float* invstd, float* dgamma, float* dbeta,
                                  const float* w, const void* x, const float* sc, const float* sh, void* dz, float* st0, float* st1,
                                  float* partial, int N, int T, int H, int W, int C, hipStream_t stream) {
    if ((C & 63) || !sc || !sh || R <= 0 || R > 128 || !bst0 || !bst1 || !partial || !dz || !st0 || !st1) return TUBER_EINVAL;
    TileArgs a{};
    a.in = (const bf16*)dzu; a.xu = (const bf16*)xu; a.sc = sc; a.sh = sh; a.w = w; a.out = (bf16*)dz; a.aux = (const bf16*)x;
    a.st0 = st0; a.st1 = st1; a.P = partial;
    a.bst0 = bst0; a.bst1 = bst1; a.bR = R; a.bcount = count; a.bgamma = gamma; a.bmean = mean; a.binvstd = invstd;
    a.bdgamma = dgamma; a.bdbeta = dbeta;
    a.g = make_geom(N, T, H, W, C);
    const size_t lds = (3 * PLANE + 27 * 64 + 3 * 64) * sizeof(float);
    static LdsOptIn opt;
    TUBER_LDS_OPT_IN(opt, dwconv_tile_bwd_both_kernel, lds);
    hipLaunchKernelGGL(dwconv_tile_bwd_both_kernel, dim3(a.g.N * a.g.tchunks * a.g.htiles * a.g.wtiles, C / 64), dim3(512), lds, stream, a);
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
