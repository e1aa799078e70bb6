// LayerNorm backward fused INTO the data-gradient GEMM of the linear layer in front of it -- gfx950.
// reference: autograd of `tgt = norm(tgt + dropout(sublayer(...)))`, models/transformer/transformer.py:160-167,229-247 (post-norm encoder /
// decoder layers): the sublayer ends in a linear (attention out_proj or the FFN's linear2), so the backward chain has
//     layernorm_bwd (gradient of the normalised rows)  ->  gemm_nt (its product with W, the gradient of the linear's input)
// as two dependent launches of 2 - 352 workgroups, ~5 us each on tensors of 30 x 256 (decoder) or 2816 x 256 (encoder): launch-bound.
// Here every workgroup owns 32 rows: it runs the LayerNorm backward of those rows itself (one wave per row, the arithmetic of
// layernorm_bwd_kernel<4>, norm.hip, expression for expression), keeps the bf16 result -- the operand the GEMM launch would have read
// back -- as an LDS image, and goes straight on to its 64 output columns of the product on MFMA.  The column-0 workgroup of
// each row block also stores what the two-launch path stores: dx (gradient of the residual input), dxd (gradient of the linear's
// output through the Dropout mask: the operand of the queued weight-gradient GEMM) and the block's dgamma / dbeta partial row.
// The LayerNorm part is recomputed by the other column workgroups (32 rows x 256: ~1 us from L2) -- cheaper than the launch boundary.
// Epilogues of the product = those tape.py: linear.bwd chooses for the stand-alone GEMM: plain, + an existing gradient of the
// linear's input (res), or the ReLU / Dropout mask of that input (cm > 0, scaled by alpha).
#include "common.h"

namespace {

constexpr int E = 256;                     // LayerNorm width = reduction length of the product
constexpr int RB = 32;                     // rows per workgroup (two 16-row MFMA tiles)
constexpr int PX = 264;                    // bf16 pitch of the [32][256] operand image (528-byte rows: conflict-free 16-byte reads of 16 rows)

struct LnDxArgs {
    const bf16* dy; long lddy; const bf16* dy2; long lddy2;      // gradient of the LayerNorm output (dy2: a second contribution, or NULL)
    // ... or that second contribution is itself the backward of ANOTHER LayerNorm over the same rows (the decoder's shared decoder.norm on each
    // layer's output, no Dropout, no residual): its output gradient ga, saved xhat / rstd, gamma and partial rows (ga != NULL excludes dy2)
    const bf16* ga; long ldga; const bf16* xhat_a; const float* rstd_a; const float* gamma_a; float* part_a;
    const bf16* xhat; const float* rstd; const float* gamma;
    bf16* dx; bf16* dxd; float* part;                             // what layernorm_bwd stores ([blocks][2E] partials)
    int M; uint32_t thresh; float inv_keep; const uint64_t* seed_ptr; uint64_t salt;
    const bf16* wt; long ldt; int Kin;                            // W^T rows [Kin][ldt >= 256]
    bf16* out; const bf16* res; const bf16* cm; float alpha;      // out [M][Kin]
};

__device__ __forceinline__ f32x4 mma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

constexpr int CT = 1;  // 16-column tiles per wave: the workgroup covers 64 * CT output columns (CT = 4 was measured and lost, see the launcher)
__global__ __launch_bounds__(256) void ln_bwd_dx_kernel(LnDxArgs a) {
    __shared__ __attribute__((aligned(16))) bf16 img[RB * PX];
    __shared__ float red[2][4][E];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r0 = blockIdx.x * RB, rows = min(RB, a.M - r0);
    const bool writer = blockIdx.y == 0;
    const int li = lane & 15, g = lane >> 4;
    const int colbase = blockIdx.y * 64 * CT + w * 16 * CT;

    // ---- the weight fragments of this wave's first column tile go out first: their latency runs under the LayerNorm phase ----
    uint4 wf[8];
    {
        const bf16* wp = a.wt + (long)(colbase + li) * a.ldt + g * 8;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) wf[kk] = *(const uint4*)(wp + kk * 32);
    }

    // ---- LayerNorm backward of rows r0 + w, w + 4, ... (one wave per row, lane = 4 consecutive columns) ----
    const uint64_t seed = a.thresh ? (a.seed_ptr ? *a.seed_ptr : 0ull) * 0x9E3779B97F4A7C15ull + a.salt : 0ull;
    float gm[4];
    {
        const float4 gv = *(const float4*)(a.gamma + lane * 4);
        gm[0] = gv.x; gm[1] = gv.y; gm[2] = gv.z; gm[3] = gv.w;
    }
    float ag[4] = {0.f, 0.f, 0.f, 0.f}, ab[4] = {0.f, 0.f, 0.f, 0.f};
    // all eight rows of the wave are fetched together (the rows are independent; one row per trip waited for its own round trip)
    uint2 rd[RB / 4], rh[RB / 4], rd2[RB / 4], rha[RB / 4];
    float rrs[RB / 4], rrsa[RB / 4];
    float gma[4] = {0.f, 0.f, 0.f, 0.f}, aga[4] = {0.f, 0.f, 0.f, 0.f}, aba[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.ga) {
        const float4 gv = *(const float4*)(a.gamma_a + lane * 4);
        gma[0] = gv.x; gma[1] = gv.y; gma[2] = gv.z; gma[3] = gv.w;
    }
#pragma unroll
    for (int k = 0; k < RB / 4; ++k) {
        const int row = r0 + min(w + 4 * k, rows - 1);
        rd[k] = *(const uint2*)(a.dy + (long)row * a.lddy + lane * 4);
        rh[k] = *(const uint2*)(a.xhat + (long)row * E + lane * 4);
        if (a.dy2) rd2[k] = *(const uint2*)(a.dy2 + (long)row * a.lddy2 + lane * 4);
        rrs[k] = a.rstd[row];
        if (a.ga) {
            rd2[k] = *(const uint2*)(a.ga + (long)row * a.ldga + lane * 4);
            rha[k] = *(const uint2*)(a.xhat_a + (long)row * E + lane * 4);
            rrsa[k] = a.rstd_a[row];
        }
    }
#pragma unroll
    for (int k = 0; k < RB / 4; ++k) {
        const int lr = w + 4 * k;                       // row inside the block
        bf16x4 o = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f}, od = o;
        if (lr < rows) {
            const long base = (long)(r0 + lr) * E;
            const bf16x4 da = as_bf16x4(rd[k]), hb = as_bf16x4(rh[k]);
            float d[4], h[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) { d[e] = bf2f(da[e]); h[e] = bf2f(hb[e]); }
            if (a.dy2) {                                // the sum of the two contributions as the axpby launch would have stored it
                const bf16x4 db = as_bf16x4(rd2[k]);
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = bf2f(f2bf(d[e] + bf2f(db[e])));
            } else if (a.ga) {                          // the other LayerNorm's backward first (layernorm_bwd_kernel's arithmetic, bf16 result), then the sum
                const bf16x4 da2 = as_bf16x4(rd2[k]), ha = as_bf16x4(rha[k]);
                float d2[4], h2[4], t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    d2[e] = bf2f(da2[e]); h2[e] = bf2f(ha[e]);
                    aga[e] += d2[e] * h2[e]; aba[e] += d2[e];
                    const float gd = d2[e] * gma[e];
                    t1 += gd; t2 += gd * h2[e];
                }
                t1 = wave_sum(t1) * (1.f / E);
                t2 = wave_sum(t2) * (1.f / E);
#pragma unroll
                for (int e = 0; e < 4; ++e) d[e] = bf2f(f2bf(d[e] + bf2f(f2bf(rrsa[k] * (d2[e] * gma[e] - t1 - h2[e] * t2)))));
            }
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                ag[e] += d[e] * h[e]; ab[e] += d[e];
                const float gd = d[e] * gm[e];
                s1 += gd; s2 += gd * h[e];
            }
            s1 = wave_sum(s1) * (1.f / E);
            s2 = wave_sum(s2) * (1.f / E);
            const float rs = rrs[k];
            bool keep[4] = {true, true, true, true};
            if (a.thresh) dropout_keep_run<4>(seed, (uint64_t)(base + lane * 4), a.thresh, keep);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float gxe = rs * (d[e] * gm[e] - s1 - h[e] * s2);
                o[e] = f2bf(gxe);
                od[e] = keep[e] ? f2bf(gxe * a.inv_keep) : (bf16)0.f;
            }
            if (writer) {
                if (a.dx) *(uint2*)(a.dx + base + lane * 4) = as_uint2(o);
                if (a.dxd) *(uint2*)(a.dxd + base + lane * 4) = as_uint2(od);
            }
        }
        *(uint2*)(img + lr * PX + lane * 4) = as_uint2(a.thresh ? od : o);       // rows beyond M: zeros
    }
    if (writer) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[0][w][lane * 4 + e] = ag[e]; red[1][w][lane * 4 + e] = ab[e]; }
    }
    __syncthreads();
    if (writer) {
        const int col = tid;                            // 256 threads = E columns
        a.part[(long)blockIdx.x * 2 * E + col] = red[0][0][col] + red[0][1][col] + red[0][2][col] + red[0][3][col];
        a.part[(long)blockIdx.x * 2 * E + E + col] = red[1][0][col] + red[1][1][col] + red[1][2][col] + red[1][3][col];
        if (a.ga) {                                     // (uniform branch: a.ga is a kernel argument)
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 4; ++e) { red[0][w][lane * 4 + e] = aga[e]; red[1][w][lane * 4 + e] = aba[e]; }
            __syncthreads();
            a.part_a[(long)blockIdx.x * 2 * E + col] = red[0][0][col] + red[0][1][col] + red[0][2][col] + red[0][3][col];
            a.part_a[(long)blockIdx.x * 2 * E + E + col] = red[1][0][col] + red[1][1][col] + red[1][2][col] + red[1][3][col];
        }
    }

    // ---- out[r0 + .][colbase + ..] = img . W^T: weights = MFMA A operand, so lane (li, g) ends with acc[rt][r] = out[rt*16 + li][col + g*4 + r] ----
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
        const int col0 = colbase + ct * 16;
        f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        uint4 wn[8];
        if (ct + 1 < CT) {                              // next tile's fragments under this tile's MFMAs
            const bf16* wp = a.wt + (long)(col0 + 16 + li) * a.ldt + g * 8;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) wn[kk] = *(const uint4*)(wp + kk * 32);
        }
        // side operands of the epilogue, fetched before the MFMAs
        uint2 side[2];
        const bf16* sp = a.res ? a.res : a.cm;
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int r = min(rt * 16 + li, rows - 1);
            side[rt] = sp ? *(const uint2*)(sp + (long)(r0 + r) * a.Kin + col0 + g * 4) : make_uint2(0, 0);
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const bf16x8 x = as_bf16x8(*(const uint4*)(img + (rt * 16 + li) * PX + kk * 32 + g * 8));
                acc[rt] = mma(as_bf16x8(wf[kk]), x, acc[rt]);
            }
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int r = rt * 16 + li;
            if (r >= rows) continue;
            const bf16x4 sv = as_bf16x4(side[rt]);
            bf16x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[rt][e];
                if (a.res) v += bf2f(sv[e]);
                else if (a.cm) v = bf2f(sv[e]) > 0.f ? v * a.alpha : 0.f;
                o[e] = f2bf(v);
            }
            *(uint2*)(a.out + (long)(r0 + r) * a.Kin + col0 + g * 4) = as_uint2(o);
        }
        if (ct + 1 < CT) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) wf[kk] = wn[kk];
        }
    }
}

}  // namespace

extern "C" {

// partial rows ([blocks][2 * 256] floats) tuber_ln_bwd_dx writes for M rows
int tuber_ln_bwd_dx_blocks(int M) { return ceil_div(M, RB); }

// is the fused form available for a LayerNorm of width E in front of a linear with Kin inputs?
int tuber_ln_bwd_dx_supported(int E_, int Kin) { return E_ == E && Kin >= 64 && (Kin % 64) == 0 ? 1 : 0; }

// where the one-launch form is faster than the two launches (measured): the decoder's 30 rows at any width, many rows only in front of a
// 256-input linear (attention out_proj); the encoder's linear2 (2816 rows x 2048 inputs) recomputes the LayerNorm part 32 times and loses
int tuber_ln_bwd_dx_pays(int M, int E_, int Kin) { return tuber_ln_bwd_dx_supported(E_, Kin) && (M <= 64 || Kin <= 256) ? 1 : 0; }

// LayerNorm backward (tuber_layernorm_bwd with accumulate = 2: partial rows only) + the data-gradient product of the linear in front of it:
//   dxn = LN'(dy [+ dy2]);  dx = dxn (if given);  dxd = Dropout-masked dxn (if given; required when p > 0);
//   out[M][Kin] = (p > 0 ? dxd : dxn) . W   with W^T given as rows wt[Kin][ldt]  [+ res | masked by cm > 0 and scaled by alpha]
// ga .. partial_a (or NULLs): the second contribution is the backward of another LayerNorm over the same rows (no Dropout / residual), formed
// on load from ITS output gradient and saved xhat / rstd; its dgamma / dbeta partial rows go to partial_a (same block count)
int tuber_ln_bwd_dx(const void* dy, long lddy, const void* dy2, long lddy2, const void* xhat, const float* rstd, const float* gamma,
                    void* dx, void* dxd, float* partial, int M, int E_, float p, const void* seed_ptr, unsigned long long salt,
                    const void* wt, long ldt, int Kin, void* out, const void* res, const void* cm, float alpha,
                    const void* ga, long ldga, const void* xhat_a, const float* rstd_a, const float* gamma_a, float* partial_a, hipStream_t stream) {
    if (ga && (dy2 || ldga < E || (ldga & 3) || !xhat_a || !rstd_a || !gamma_a || !partial_a)) return TUBER_EINVAL;
    if (!tuber_ln_bwd_dx_supported(E_, Kin) || M <= 0 || lddy < E || (lddy & 3) || (dy2 && (lddy2 < E || (lddy2 & 3))) || ldt < E || (ldt & 7) ||
        p < 0.f || p >= 1.f || (p > 0.f && !dxd) || !partial || !out || (res && cm))
        return TUBER_EINVAL;
    LnDxArgs a;
    a.dy = (const bf16*)dy; a.lddy = lddy; a.dy2 = (const bf16*)dy2; a.lddy2 = lddy2;
    a.ga = (const bf16*)ga; a.ldga = ldga; a.xhat_a = (const bf16*)xhat_a; a.rstd_a = rstd_a; a.gamma_a = gamma_a; a.part_a = partial_a;
    a.xhat = (const bf16*)xhat; a.rstd = rstd; a.gamma = gamma;
    a.dx = (bf16*)dx; a.dxd = (bf16*)dxd; a.part = partial;
    a.M = M; a.thresh = (uint32_t)((double)p * 4294967296.0); a.inv_keep = dropout_inv_keep(p);
    a.seed_ptr = (const uint64_t*)seed_ptr; a.salt = (uint64_t)salt;
    a.wt = (const bf16*)wt; a.ldt = ldt; a.Kin = Kin;
    a.out = (bf16*)out; a.res = (const bf16*)res; a.cm = (const bf16*)cm; a.alpha = alpha;
    // (a 256-column workgroup form for wide outputs of many row blocks -- the encoder's linear2, 2816 x 2048 -- was measured at 27 us against
    // 5.3 + 6.4 us for the two launches: tuber_ln_bwd_dx_pays says where the fused form is used)
    hipLaunchKernelGGL(ln_bwd_dx_kernel, dim3(ceil_div(M, RB), Kin / 64), dim3(256), 0, stream, a);
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
