// Small HBM-bound helpers of the TubeR path (gfx950): dtype casts / weight repacks, row gathers
// with broadcast or reduction, temporal pooling, sigmoid, dropout, adds.  16-byte accesses.
#include "common.h"

__global__ void cast_f32_bf16_kernel(const float* __restrict__ s, bf16* __restrict__ d, long n) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (i + 8 <= n) {
        const float4 a = *(const float4*)(s + i), b = *(const float4*)(s + i + 4);
        bf16x8 v = {f2bf(a.x), f2bf(a.y), f2bf(a.z), f2bf(a.w), f2bf(b.x), f2bf(b.y), f2bf(b.z), f2bf(b.w)};
        *(uint4*)(d + i) = as_uint4(v);
    } else {
        for (long j = i; j < n; ++j) d[j] = f2bf(s[j]);
    }
}
__global__ void cast_bf16_f32_kernel(const bf16* __restrict__ s, float* __restrict__ d, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] = bf2f(s[i]);
}

// d = float(s) * scale, 8 elements (16 B in, 32 B out) per thread per trip; n and both pointers 16-byte aligned except the tail
__global__ __launch_bounds__(256) void cast_bf16_f32_scale_kernel(const bf16* __restrict__ s, float* __restrict__ d, long n, float scale) {
    const long n8 = n >> 3;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const bf16x8 v = as_bf16x8(((const uint4*)s)[i]);
        float4 a, b;
        a.x = bf2f(v[0]) * scale; a.y = bf2f(v[1]) * scale; a.z = bf2f(v[2]) * scale; a.w = bf2f(v[3]) * scale;
        b.x = bf2f(v[4]) * scale; b.y = bf2f(v[5]) * scale; b.z = bf2f(v[6]) * scale; b.w = bf2f(v[7]) * scale;
        ((float4*)d)[2 * i] = a;
        ((float4*)d)[2 * i + 1] = b;
    }
    if (blockIdx.x == 0)
        for (long i = (n8 << 3) + threadIdx.x; i < n; i += blockDim.x) d[i] = bf2f(s[i]) * scale;
}

// W[R][C] fp32 -> WT[C][ldt] bf16 (transposed repack for the data-gradient GEMM), 32x32 LDS tiles
__global__ __launch_bounds__(256) void cast_transpose_kernel(const float* __restrict__ W, bf16* __restrict__ WT, int R, int C, int ldt) {
    __shared__ float t[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int j = ty; j < 32; j += 8) t[j][tx] = (r0 + j < R && c0 + tx < C) ? W[(long)(r0 + j) * C + c0 + tx] : 0.f;
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (c0 + j < C && r0 + tx < R) WT[(long)(c0 + j) * ldt + r0 + tx] = f2bf(t[tx][j]);
}

// out[(a,b,c)][:] = sum_{d<D} in[a*sa + b*sb + c*sc + d*sd][:]  (rows of E bf16, E % 8 == 0)
__global__ __launch_bounds__(256) void rows_gather_sum_kernel(const bf16* __restrict__ in, bf16* __restrict__ out, int A, int B, int Cc,
                                                              int D, long sa, long sb, long sc, long sd, int E, float mul) {
    const int epr = E >> 3;
    const long total = (long)A * B * Cc * epr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % epr); long r = i / epr;
        const int c = (int)(r % Cc); long q = r / Cc;
        const int b = (int)(q % B); const int a = (int)(q / B);
        const long src = (long)a * sa + (long)b * sb + (long)c * sc;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int d = 0; d < D; ++d) {
            const bf16x8 v = as_bf16x8(*(const uint4*)(in + (src + (long)d * sd) * E + ch * 8));
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += bf2f(v[e]);
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(acc[e] * mul);
        *(uint4*)(out + r * E + ch * 8) = as_uint4(o);
    }
}

// temporal max pool over all T frames: out[(b,p)][e] = max_t x[(b,t,p)][e], arg = first t that attains it (nn.MaxPool3d((T,1,1)),
// models/backbone_builder.py:45-47,73); backward routes the gradient to that frame only.  8 channels per thread.
__global__ __launch_bounds__(256) void temporal_max_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ out, uint8_t* __restrict__ arg,
                                                               int B, int T, long hw, int E) {
    const int epr = E >> 3;
    const long total = (long)B * hw * epr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % epr);
        const long r = i / epr;
        const long pos = r % hw, b = r / hw;
        float best[8];
        uint8_t at[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; at[e] = 0; }
        for (int t = 0; t < T; ++t) {
            const bf16x8 v = as_bf16x8(*(const uint4*)(x + (((long)b * T + t) * hw + pos) * E + ch * 8));
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = bf2f(v[e]);
                if (f > best[e] || (t == 0)) { best[e] = f; at[e] = (uint8_t)t; }      // strict >: ties keep the earliest frame (NaN-free data)
            }
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(best[e]);
        *(uint4*)(out + r * E + ch * 8) = as_uint4(o);
        if (arg) {
            uint2 a;
            a.x = at[0] | (at[1] << 8) | (at[2] << 16) | ((uint32_t)at[3] << 24);
            a.y = at[4] | (at[5] << 8) | (at[6] << 16) | ((uint32_t)at[7] << 24);
            *(uint2*)(arg + r * E + ch * 8) = a;
        }
    }
}
__global__ __launch_bounds__(256) void temporal_max_bwd_kernel(const bf16* __restrict__ g, const uint8_t* __restrict__ arg, bf16* __restrict__ dx,
                                                               int B, int T, long hw, int E) {
    const int epr = E >> 3;
    const long total = (long)B * T * hw * epr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % epr);
        long r = i / epr;
        const long pos = r % hw; r /= hw;
        const int t = (int)(r % T);
        const long b = r / T;
        const long orow = b * hw + pos;
        const bf16x8 gv = as_bf16x8(*(const uint4*)(g + orow * E + ch * 8));
        const uint2 a = *(const uint2*)(arg + orow * E + ch * 8);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const uint32_t at = ((e < 4 ? a.x : a.y) >> (8 * (e & 3))) & 255u;
            o[e] = at == (uint32_t)t ? gv[e] : f2bf(0.f);
        }
        *(uint4*)(dx + ((b * T + t) * hw + pos) * E + ch * 8) = as_uint4(o);
    }
}

// out = alpha*a + beta*b   (bf16, n % 8 == 0)
__global__ void axpby_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, bf16* __restrict__ out, long n8, float alpha, float beta) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const bf16x8 x = as_bf16x8(((const uint4*)a)[i]);
        bf16x8 y = bf16x8{};
        if (b) y = as_bf16x8(((const uint4*)b)[i]);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(alpha * bf2f(x[e]) + (b ? beta * bf2f(y[e]) : 0.f));
        ((uint4*)out)[i] = as_uint4(o);
    }
}

// in-place (or out-of-place) dropout with the stateless hash RNG: y = keep ? x/(1-p) : 0
__global__ void dropout_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, long n8, uint32_t thresh, float inv_keep,
                               const uint64_t* __restrict__ seed_ptr, uint64_t salt) {
    const uint64_t seed = (seed_ptr ? *seed_ptr : 0ull) * 0x9E3779B97F4A7C15ull + salt;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const bf16x8 v = as_bf16x8(((const uint4*)x)[i]);
        bf16x8 o;
        bool keep[8];
        dropout_keep_run<8>(seed, (uint64_t)i * 8, thresh, keep);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = keep[e] ? f2bf(bf2f(v[e]) * inv_keep) : (bf16)0.f;
        ((uint4*)y)[i] = as_uint4(o);
    }
}

// y = sigmoid(x) (fp32);  bwd: dx = dy * y * (1 - y)
__global__ void sigmoid_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = 1.f / (1.f + __expf(-x[i]));
}
__global__ void sigmoid_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dx[i] = dy[i] * y[i] * (1.f - y[i]);
}

// dx = alpha * dy * [h > 0]   (ReLU backward from the saved post-activation; alpha = 1/(1-p) when Dropout followed the ReLU:
// a dropped element has h == 0 as well)
__global__ void relu_mask_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ h, bf16* __restrict__ dx, long n8, float alpha) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
        const bf16x8 g = as_bf16x8(((const uint4*)dy)[i]);
        const bf16x8 a = as_bf16x8(((const uint4*)h)[i]);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = bf2f(a[e]) > 0.f ? f2bf(bf2f(g[e]) * alpha) : (bf16)0.f;
        ((uint4*)dx)[i] = as_uint4(o);
    }
}

// PositionEmbeddingSine_3D (models/transformer/position_encoding.py:32-72) for a (B,T,H,W) bool mask,
// written token-major: out[((b*T + t)*H + y)*W + x][hidden] bf16 (and fp32 NCDHW-free); normalize=True,
// scale 2*pi, eps 1e-6, temperature 1e4; channels = [t: hidden/4 | y: 3*hidden/8 | x: 3*hidden/8].
__global__ void posenc_kernel(const uint8_t* __restrict__ mask, bf16* __restrict__ out, int B, int T, int H, int W, int hidden) {
    const long tok = blockIdx.x;
    const int x = (int)(tok % W); long r = tok / W;
    const int y = (int)(r % H); r /= H;
    const int t = (int)(r % T); const int b = (int)(r / T);
    __shared__ float emb[3];
    if (threadIdx.x < 3) {
        // cumulative count of non-masked positions up to (and including) this one, and the axis total
        float cum = 0.f, tot = 0.f;
        if (threadIdx.x == 0) { for (int i = 0; i < T; ++i) { const float v = mask[(((long)b * T + i) * H + y) * W + x] ? 0.f : 1.f; tot += v; if (i <= t) cum += v; } }
        if (threadIdx.x == 1) { for (int i = 0; i < H; ++i) { const float v = mask[(((long)b * T + t) * H + i) * W + x] ? 0.f : 1.f; tot += v; if (i <= y) cum += v; } }
        if (threadIdx.x == 2) { for (int i = 0; i < W; ++i) { const float v = mask[(((long)b * T + t) * H + y) * W + i] ? 0.f : 1.f; tot += v; if (i <= x) cum += v; } }
        emb[threadIdx.x] = cum / (tot + 1e-6f) * 6.283185307179586f;
    }
    __syncthreads();
    const int nt = hidden / 4, ns = hidden * 3 / 8;
    for (int c = threadIdx.x; c < hidden; c += blockDim.x) {
        int axis, i, nf;
        if (c < nt) { axis = 0; i = c; nf = nt; } else if (c < nt + ns) { axis = 1; i = c - nt; nf = ns; } else { axis = 2; i = c - nt - ns; nf = ns; }
        const float div = powf(10000.f, (float)(2 * (i / 2)) / (float)nf);
        const float v = emb[axis] / div;
        out[tok * hidden + c] = f2bf((i & 1) ? cosf(v) : sinf(v));
    }
}


// batched W[R][C] -> W^T[C][ldt], bf16 -> bf16, over a device table of matrices (one launch for every GEMM weight of the model).
// table[i] = {src_off, dst_off, R, C, ldt, tile_begin, tiles_x}; tiles are 64 x 64.  The source is the bf16 SHADOW of the parameters
// (tuber_cast_f32_bf16 has just written it: half the bytes of the fp32 masters, and the same rounding); 16-byte loads and stores.
// Round 1 - 3 transposed 32 x 32 tiles from the fp32 masters with 4-byte loads and 2-byte stores: 48 108 workgroups whose life was a
// binary search over the table and one 4 KB tile -- 105 us per step for 240 MB.
struct TrEntry { long src_off, dst_off; int R, C, ldt, tile_begin, tiles_x, pad; };
__global__ __launch_bounds__(256) void multi_transpose_bf16_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst,
                                                                   const TrEntry* __restrict__ table, int nmat) {
    constexpr int PITCH = 72;                       // bf16 per row of the TRANSPOSED image: 144 B, 16-byte aligned rows
    __shared__ __attribute__((aligned(16))) bf16 tt[64][PITCH];     // tt[c][r]
    int lo = 0, hi = nmat - 1;
    const int tile = blockIdx.x;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (table[mid].tile_begin <= tile) lo = mid; else hi = mid - 1; }
    const TrEntry e = table[lo];
    const int lt = tile - e.tile_begin;
    const int c0 = (lt % e.tiles_x) * 64, r0 = (lt / e.tiles_x) * 64;
    const bf16* W = src + e.src_off;
    bf16* WT = dst + e.dst_off;
    const int q = threadIdx.x & 7, rr = threadIdx.x >> 3;
    const bool vec = (e.C & 7) == 0;                // rows of W start 16-byte aligned (src_off is a multiple of 64 elements)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = rr + 32 * i;
        bf16x8 v;
        if (vec) {
            v = (r0 + r < e.R && c0 + q * 8 < e.C) ? as_bf16x8(*(const uint4*)(W + (long)(r0 + r) * e.C + c0 + q * 8)) : as_bf16x8(make_uint4(0, 0, 0, 0));
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = (r0 + r < e.R && c0 + q * 8 + k < e.C) ? W[(long)(r0 + r) * e.C + c0 + q * 8 + k] : (bf16)0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) tt[q * 8 + k][r] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = rr + 32 * i;                  // output row = source column; 8 consecutive source rows per thread
        if (c0 + c < e.C && r0 + q * 8 < e.ldt) *(uint4*)(WT + (long)(c0 + c) * e.ldt + r0 + q * 8) = *(const uint4*)&tt[c][q * 8];
    }
}

// src[R][C] fp32 -> dst[R][ldd] bf16, columns C..ldd-1 zero filled
__global__ void cast_pad_rows_kernel(const float* __restrict__ src, bf16* __restrict__ dst, int R, int C, int ldd) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)R * ldd) return;
    const int r = (int)(i / ldd), c = (int)(i % ldd);
    dst[i] = c < C ? f2bf(src[(long)r * C + c]) : (bf16)0.f;
}

// dst[map(m)][:] += src[m][:] for the strided row map of a down_sample conv (rows unique: no atomics)
__global__ __launch_bounds__(256) void rows_scatter_add_kernel(bf16* __restrict__ dst, const bf16* __restrict__ src, long Mo, int To,
                                                               int Ho, int Wo, int Ti, int Hi, int Wi, int st, int ss, int E) {
    const int epr = E >> 3;
    const long total = Mo * epr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ch = (int)(i % epr); const long m = i / epr;
        int w = (int)(m % Wo); long r = m / Wo;
        int h = (int)(r % Ho); r /= Ho;
        int t = (int)(r % To); const long n = r / To;
        const long d = ((n * Ti + (long)t * st) * Hi + (long)h * ss) * Wi + (long)w * ss;
        const bf16x8 a = as_bf16x8(*(const uint4*)(src + m * E + ch * 8));
        bf16x8 b = as_bf16x8(*(const uint4*)(dst + d * E + ch * 8));
#pragma unroll
        for (int e = 0; e < 8; ++e) b[e] = f2bf(bf2f(a[e]) + bf2f(b[e]));
        *(uint4*)(dst + d * E + ch * 8) = as_uint4(b);
    }
}

static inline int grid1(long n, int cap = 8192) { long b = (n + 255) / 256; return (int)(b > cap ? cap : (b < 1 ? 1 : b)); }

extern "C" {

int tuber_cast_f32_bf16(const float* src, void* dst, long n, hipStream_t stream) {
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ceil_div(ceil_div(n, 8), 256)), dim3(256), 0, stream, src, (bf16*)dst, n);
    TUBER_RETURN_LAUNCH();
}
int tuber_cast_bf16_f32(const void* src, float* dst, long n, hipStream_t stream) {
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, (const bf16*)src, dst, n);
    TUBER_RETURN_LAUNCH();
}
int tuber_cast_bf16_f32_scale(const void* src, float* dst, long n, float scale, hipStream_t stream) {
    if (n <= 0) return TUBER_EINVAL;
    long nb = ceil_div(ceil_div(n, 8), 256);
    if (nb > 8192) nb = 8192;
    hipLaunchKernelGGL(cast_bf16_f32_scale_kernel, dim3((int)nb), dim3(256), 0, stream, (const bf16*)src, dst, n, scale);
    TUBER_RETURN_LAUNCH();
}
int tuber_cast_transpose(const float* W, void* WT, int R, int C, int ldt, hipStream_t stream) {
    hipLaunchKernelGGL(cast_transpose_kernel, dim3(ceil_div(C, 32), ceil_div(R, 32)), dim3(256), 0, stream, W, (bf16*)WT, R, C, ldt);
    TUBER_RETURN_LAUNCH();
}
int tuber_rows_gather_sum(const void* in, void* out, int A, int B, int C, int D, long sa, long sb, long sc, long sd, int E, float mul,
                          hipStream_t stream) {
    if (E & 7) return TUBER_EINVAL;
    hipLaunchKernelGGL(rows_gather_sum_kernel, dim3(grid1((long)A * B * C * (E / 8))), dim3(256), 0, stream, (const bf16*)in, (bf16*)out,
                       A, B, C, D, sa, sb, sc, sd, E, mul);
    TUBER_RETURN_LAUNCH();
}
int tuber_temporal_max_fwd(const void* x, void* out, void* arg, int B, int T, long hw, int E, hipStream_t stream) {
    if ((E & 7) || T <= 0 || T > 255) return TUBER_EINVAL;
    hipLaunchKernelGGL(temporal_max_fwd_kernel, dim3(grid1((long)B * hw * (E / 8))), dim3(256), 0, stream, (const bf16*)x, (bf16*)out,
                       (uint8_t*)arg, B, T, hw, E);
    TUBER_RETURN_LAUNCH();
}
int tuber_temporal_max_bwd(const void* g, const void* arg, void* dx, int B, int T, long hw, int E, hipStream_t stream) {
    if ((E & 7) || T <= 0 || T > 255) return TUBER_EINVAL;
    hipLaunchKernelGGL(temporal_max_bwd_kernel, dim3(grid1((long)B * T * hw * (E / 8))), dim3(256), 0, stream, (const bf16*)g,
                       (const uint8_t*)arg, (bf16*)dx, B, T, hw, E);
    TUBER_RETURN_LAUNCH();
}
int tuber_axpby(const void* a, const void* b, void* out, long n, float alpha, float beta, hipStream_t stream) {
    if (n & 7) return TUBER_EINVAL;
    hipLaunchKernelGGL(axpby_kernel, dim3(grid1(n / 8)), dim3(256), 0, stream, (const bf16*)a, (const bf16*)b, (bf16*)out, n / 8, alpha, beta);
    TUBER_RETURN_LAUNCH();
}
int tuber_dropout(const void* x, void* y, long n, float p, const void* seed_ptr, unsigned long long salt, hipStream_t stream) {
    if ((n & 7) || p < 0.f || p >= 1.f) return TUBER_EINVAL;
    hipLaunchKernelGGL(dropout_kernel, dim3(grid1(n / 8)), dim3(256), 0, stream, (const bf16*)x, (bf16*)y, n / 8,
                       (uint32_t)((double)p * 4294967296.0), dropout_inv_keep(p), (const uint64_t*)seed_ptr, (uint64_t)salt);
    TUBER_RETURN_LAUNCH();
}
int tuber_sigmoid_fwd(const float* x, float* y, long n, hipStream_t stream) {
    hipLaunchKernelGGL(sigmoid_fwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, x, y, n);
    TUBER_RETURN_LAUNCH();
}
int tuber_sigmoid_bwd(const float* dy, const float* y, float* dx, long n, hipStream_t stream) {
    hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, dy, y, dx, n);
    TUBER_RETURN_LAUNCH();
}
int tuber_relu_mask(const void* dy, const void* h, void* dx, long n, float alpha, hipStream_t stream) {
    if (n & 7) return TUBER_EINVAL;
    hipLaunchKernelGGL(relu_mask_kernel, dim3(grid1(n / 8)), dim3(256), 0, stream, (const bf16*)dy, (const bf16*)h, (bf16*)dx, n / 8, alpha);
    TUBER_RETURN_LAUNCH();
}
int tuber_posenc(const void* mask, void* out, int B, int T, int H, int W, int hidden, hipStream_t stream) {
    if (hidden % 8) return TUBER_EINVAL;
    hipLaunchKernelGGL(posenc_kernel, dim3(B * T * H * W), dim3(64), 0, stream, (const uint8_t*)mask, (bf16*)out, B, T, H, W, hidden);
    TUBER_RETURN_LAUNCH();
}

int tuber_multi_transpose_bf16(const void* src, void* dst, const void* table, int nmat, int total_tiles, hipStream_t stream) {
    if (nmat <= 0 || total_tiles <= 0) return TUBER_EINVAL;
    hipLaunchKernelGGL(multi_transpose_bf16_kernel, dim3(total_tiles), dim3(256), 0, stream, (const bf16*)src, (bf16*)dst, (const TrEntry*)table, nmat);
    TUBER_RETURN_LAUNCH();
}
int tuber_cast_pad_rows(const float* src, void* dst, int R, int C, int ldd, hipStream_t stream) {
    if (ldd < C) return TUBER_EINVAL;
    hipLaunchKernelGGL(cast_pad_rows_kernel, dim3(ceil_div((long)R * ldd, 256)), dim3(256), 0, stream, src, (bf16*)dst, R, C, ldd);
    TUBER_RETURN_LAUNCH();
}
int tuber_rows_scatter_add(void* dst, const void* src, long Mo, int To, int Ho, int Wo, int Ti, int Hi, int Wi, int st, int ss, int E,
                           hipStream_t stream) {
    if (E & 7) return TUBER_EINVAL;
    hipLaunchKernelGGL(rows_scatter_add_kernel, dim3(grid1(Mo * (E / 8))), dim3(256), 0, stream, (bf16*)dst, (const bf16*)src, Mo, To, Ho,
                       Wo, Ti, Hi, Wi, st, ss, E);
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
