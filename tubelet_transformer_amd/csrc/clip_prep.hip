// Clip input pre-pass of the TubeR path on gfx950: decoded uint8 frames -> the padded, normalised fp32
// batch the stem convolution reads, plus the padding mask.  Replaces, on the device, what the reference does per sample on CPU workers:
//   datasets/ava_frame.py:146-150        PIL.Image.resize (Pillow 8-bit two-pass fixed-point bicubic)   -> frames_resize_{h,v}_kernel
//   datasets/video_transforms.py:69-85   hflip, :20-66 crop, :333-369 ColorJitter (OpenCV 8-bit HSV, hue range 180),
//   :308-322 ToTensor + Normalize, utils/misc.py:367-425 zero padding to the batch maximum + bool mask  -> clip_prepare_kernel
// All of it is byte / integer work bound by HBM: 12 B read and 12 B (+1 B mask) written per pixel, no LDS staging beyond the 3 KB of
// look-up tables, 4 output pixels per thread so every store is 16 bytes.
#include "common.h"

namespace {

struct ClipDesc {              // one per clip, in device memory; mirrors TuberClipDesc in include/tuber_hip.h
    long long src_off;         // byte offset of the clip's first frame in `frames` (uint8 [T][H][W][3])
    int H, W;                  // frame size
    int y1, x1, h, w;          // crop window in the (possibly flipped) frame; output pixel (y,x) reads frame pixel (y1+y, x1+x)
    int flip, jitter;          // horizontal flip before the crop; HSV jitter on/off
    int hue, sat, val;         // the three jitter shifts (hue in OpenCV half-degrees)
    int pad_;
};

constexpr int PRECISION_BITS = 32 - 8 - 2;     // Pillow Resample.c

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PRECISION_BITS;
    return (uint8_t)min(max(v, 0), 255);
}

// horizontal pass: dst[img][y][xo][c] = sum_k src[img][y0+y][lo(xo)+k][c] * kk[xo][k].  One workgroup per source row: the row is staged in
// LDS with 4-byte loads from the enclosing aligned window (rows start at arbitrary byte offsets), taps are LDS byte reads, and the output
// row goes back through LDS so the stores are 4 bytes wide too.  [buf_lo, buf_hi) = the bytes of `src` that may be touched.
constexpr int HROW_MAX = 3 * 4096;             // widest row staged in LDS (bytes); wider frames take the direct kernel
__global__ __launch_bounds__(256) void frames_resize_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                              const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                              int nimg, int H, int W, int Wo, int y0, int rows) {
    __shared__ uint32_t s_in[HROW_MAX / 4 + 2];
    __shared__ uint32_t s_out[HROW_MAX / 4 + 2];
    const uint8_t* buf_lo = src;
    const uint8_t* buf_hi = src + (long)nimg * H * W * 3;
    const int inb = W * 3, outb = Wo * 3;
    for (long rix = blockIdx.x; rix < (long)nimg * rows; rix += gridDim.x) {
        const int y = (int)(rix % rows), img = (int)(rix / rows);
        const uint8_t* rp = src + ((long)img * H + y0 + y) * inb;
        const int sh = (int)((uintptr_t)rp & 3);
        const uint8_t* ap = rp - sh;
        __syncthreads();                                               // previous row fully consumed / written
        for (int i = threadIdx.x; i * 4 < sh + inb; i += blockDim.x) {
            const uint8_t* g = ap + i * 4;
            uint32_t v;
            if (g >= buf_lo && g + 4 <= buf_hi) v = *(const uint32_t*)g;
            else {
                v = 0;
                for (int e = 0; e < 4; ++e)
                    if (g + e >= buf_lo && g + e < buf_hi) v |= (uint32_t)g[e] << (8 * e);
            }
            s_in[i] = v;
        }
        __syncthreads();
        const uint8_t* row = (const uint8_t*)s_in + sh;
        uint8_t* orow = (uint8_t*)s_out;
        for (int xo = threadIdx.x; xo < Wo; xo += blockDim.x) {
            const int lo = bounds[2 * xo], n = bounds[2 * xo + 1];
            const uint8_t* p = row + lo * 3;
            const int* k = kk + (long)xo * ksize;
            int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
            for (int t = 0; t < n; ++t) {
                const int w = k[t];
                a0 += p[3 * t] * w; a1 += p[3 * t + 1] * w; a2 += p[3 * t + 2] * w;
            }
            orow[3 * xo] = clip8(a0); orow[3 * xo + 1] = clip8(a1); orow[3 * xo + 2] = clip8(a2);
        }
        __syncthreads();
        uint8_t* op = dst + rix * outb;
        if ((((uintptr_t)op | (uintptr_t)outb) & 3) == 0) {
            for (int i = threadIdx.x; i * 4 < outb; i += blockDim.x) ((uint32_t*)op)[i] = s_out[i];
        } else {
            for (int i = threadIdx.x; i < outb; i += blockDim.x) op[i] = orow[i];
        }
    }
}

// same arithmetic without staging, one thread per output pixel (rows wider than HROW_MAX bytes)
__global__ __launch_bounds__(256) void frames_resize_h_direct_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                                     const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                                     int nimg, int H, int W, int Wo, int y0, int rows) {
    const long total = (long)nimg * rows * Wo;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xo = (int)(i % Wo);
        const long r = i / Wo;
        const int y = (int)(r % rows), img = (int)(r / rows);
        const int lo = bounds[2 * xo], n = bounds[2 * xo + 1];
        const uint8_t* p = src + (((long)img * H + y0 + y) * W + lo) * 3;
        const int* k = kk + (long)xo * ksize;
        int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
        for (int t = 0; t < n; ++t) {
            const int w = k[t];
            a0 += p[3 * t] * w; a1 += p[3 * t + 1] * w; a2 += p[3 * t + 2] * w;
        }
        uint8_t* o = dst + i * 3;
        o[0] = clip8(a0); o[1] = clip8(a1); o[2] = clip8(a2);
    }
}

// vertical pass over rows of rowb = W*3 bytes: dst[img][yo][j] = sum_k src[img][lo(yo)+k][j] * kk[yo][k]; VEC consecutive bytes per thread
// (VEC = 16 / 4: one 16- / 4-byte load per tap when rows and bases are that aligned; VEC = 1: any layout)
template <int VEC>
__global__ __launch_bounds__(256) void frames_resize_v_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                              const int* __restrict__ bounds, const int* __restrict__ kk, int ksize,
                                                              int nimg, int H, int rowb, int Ho) {
    constexpr int NW = VEC >= 4 ? VEC / 4 : 1;
    const int groups = (rowb + VEC - 1) / VEC;
    const long total = (long)nimg * Ho * groups;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int gq = (int)(i % groups);
        const long r = i / groups;
        const int yo = (int)(r % Ho), img = (int)(r / Ho);
        const int lo = bounds[2 * yo], n = bounds[2 * yo + 1];
        const uint8_t* p = src + ((long)img * H + lo) * rowb + gq * VEC;
        const int* k = kk + (long)yo * ksize;
        int a[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] = 1 << (PRECISION_BITS - 1);
        for (int t = 0; t < n; ++t) {
            const int w = k[t];
            const uint8_t* q = p + (long)t * rowb;
            if (VEC == 1) {
                a[0] += q[0] * w;
            } else {
                uint32_t v[NW];
                if (VEC == 16) { const uint4 u = *(const uint4*)q; v[0] = u.x; v[1 % NW] = u.y; v[2 % NW] = u.z; v[3 % NW] = u.w; }
                else v[0] = *(const uint32_t*)q;
#pragma unroll
                for (int e = 0; e < VEC; ++e) a[e] += (int)((v[e >> 2] >> (8 * (e & 3))) & 255u) * w;
            }
        }
        uint8_t* o = dst + ((long)img * Ho + yo) * rowb + gq * VEC;
        if (VEC == 1) {
            o[0] = clip8(a[0]);
        } else {
            uint32_t v[NW];
#pragma unroll
            for (int j = 0; j < NW; ++j)
                v[j] = clip8(a[4 * j]) | (clip8(a[4 * j + 1]) << 8) | (clip8(a[4 * j + 2]) << 16) | ((uint32_t)clip8(a[4 * j + 3]) << 24);
            if (VEC == 16) *(uint4*)o = make_uint4(v[0], v[1 % NW], v[2 % NW], v[3 % NW]);
            else *(uint32_t*)o = v[0];
        }
    }
}

// The library is built with -ffp-contract=fast, which lets the backend fuse a*b+c whatever the source says; OpenCV's scalar HSV -> RGB code
// rounds every product on its own and exact .5 ties are common in 8-bit data, so each product that feeds an add goes through this.
__device__ __forceinline__ float rounded(float x) {
    asm volatile("" : "+v"(x));
    return x;
}

// OpenCV 8-bit RGB -> HSV (hue range 180) -> shifts -> RGB, exactly in the order ColorJitter applies them.  The HSV -> RGB leg is fp32
// with every product and difference rounded on its own (OpenCV's scalar code is not contracted), then round-half-even to 8 bits.
__device__ __forceinline__ void jitter_pixel(int& r, int& g, int& b, const int* __restrict__ sdiv, const int* __restrict__ hdiv, int hue, int sat,
                                             int val) {
    constexpr int S = 12;
    const int v = max(max(r, g), b), vmin = min(min(r, g), b);
    const int diff = v - vmin;
    int s = (diff * sdiv[v] + (1 << (S - 1))) >> S;
    int h = v == r ? g - b : (v == g ? b - r + 2 * diff : r - g + 4 * diff);
    h = (h * hdiv[diff] + (1 << (S - 1))) >> S;
    h += h < 0 ? 180 : 0;
    h = min(max(h, 0), 255);
    s &= 255;
    // jitter (video_transforms.py:350-357)
    h = (h + hue + 180) % 180;
    s = min(max(s + sat, 0), 255);
    const int vv = min(max(v + val, 0), 255);
    // HSV -> RGB
    const float fs = rounded((float)s * (1.0f / 255.0f)), fv = (float)vv * (1.0f / 255.0f);
    float fb, fg, fr;
    if (s == 0) {
        fb = fg = fr = fv;
    } else {
        float fh = rounded((float)h * (6.0f / 180.0f));
        if (fh >= 6.f) fh -= 6.f;
        int sector = (int)floorf(fh);
        fh -= (float)sector;
        if ((unsigned)sector >= 6u) { sector = 0; fh = 0.f; }
        const float t1 = fv * (1.f - fs);
        const float t2 = fv * (1.f - rounded(fs * fh));
        const float t3 = fv * (1.f - rounded(fs * (1.f - fh)));
        // sector table (b,g,r): {1,3,0} {1,0,2} {3,0,1} {0,2,1} {0,1,3} {2,1,0}
        fb = sector == 0 ? t1 : sector == 1 ? t1 : sector == 2 ? t3 : sector == 3 ? fv : sector == 4 ? fv : t2;
        fg = sector == 0 ? t3 : sector == 1 ? fv : sector == 2 ? fv : sector == 3 ? t2 : sector == 4 ? t1 : t1;
        fr = sector == 0 ? fv : sector == 1 ? t2 : sector == 2 ? t1 : sector == 3 ? t1 : sector == 4 ? t3 : fv;
    }
    r = min(max((int)__builtin_rintf(fr * 255.0f), 0), 255);
    g = min(max((int)__builtin_rintf(fg * 255.0f), 0), 255);
    b = min(max((int)__builtin_rintf(fb * 255.0f), 0), 255);
}

// one thread = 4 consecutive output pixels of one (clip, frame, row); V4: Wmax % 4 == 0 -> float4 / packed mask stores
template <bool V4>
__global__ __launch_bounds__(256) void clip_prepare_kernel(const uint8_t* __restrict__ frames, const ClipDesc* __restrict__ desc,
                                                           const float* __restrict__ lut, const int* __restrict__ hsv_tab,
                                                           float* __restrict__ out, uint8_t* __restrict__ mask, int N, int T, int Hmax,
                                                           int Wmax) {
    __shared__ float s_lut[3 * 256];
    __shared__ int s_div[2 * 256];
    for (int i = threadIdx.x; i < 768; i += blockDim.x) s_lut[i] = lut[i];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) s_div[i] = hsv_tab[i];
    __syncthreads();
    const int quads = (Wmax + 3) >> 2;
    const long total = (long)N * T * Hmax * quads;
    const long plane = (long)Hmax * Wmax;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xq = (int)(i % quads);
        long r = i / quads;
        const int y = (int)(r % Hmax); r /= Hmax;
        const int t = (int)(r % T), n = (int)(r / T);
        const ClipDesc d = desc[n];
        const bool row_in = y < d.h;
        const uint8_t* row = frames + d.src_off + ((long)t * d.H + (row_in ? d.y1 + y : 0)) * d.W * 3;
        float o[3][4];
        uint8_t m[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int x = xq * 4 + j;
            const bool in = row_in && x < d.w;
            int sx = in ? d.x1 + x : 0;
            if (d.flip) sx = d.W - 1 - sx;
            const uint8_t* p = row + sx * 3;
            int cr = p[0], cg = p[1], cb = p[2];
            if (d.jitter) jitter_pixel(cr, cg, cb, s_div, s_div + 256, d.hue, d.sat, d.val);
            o[0][j] = in ? s_lut[cr] : 0.f;
            o[1][j] = in ? s_lut[256 + cg] : 0.f;
            o[2][j] = in ? s_lut[512 + cb] : 0.f;
            m[j] = in ? 0 : 1;
        }
        const long base = (((long)n * 3) * T + t) * plane + (long)y * Wmax + xq * 4;
        if (V4) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                *(float4*)(out + base + (long)c * T * plane) = make_float4(o[c][0], o[c][1], o[c][2], o[c][3]);
            if (t == 0) *(uint32_t*)(mask + ((long)n * Hmax + y) * Wmax + xq * 4) = m[0] | (m[1] << 8) | (m[2] << 16) | (m[3] << 24);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (xq * 4 + j >= Wmax) break;
#pragma unroll
                for (int c = 0; c < 3; ++c) out[base + (long)c * T * plane + j] = o[c][j];
                if (t == 0) mask[((long)n * Hmax + y) * Wmax + xq * 4 + j] = m[j];
            }
        }
    }
}

}  // namespace

extern "C" {

// Pillow-exact bicubic resize of `nimg` packed RGB frames [H][W][3] -> [Ho][Wo][3].  bounds_*/kk_* are the per-output (first tap, tap
// count) pairs and fixed-point weights the host precomputes (input_pipeline.resize_coeffs; bounds_v already shifted by y0); the
// horizontal pass only produces source rows [y0, y0+rows), as Pillow does; `tmp` holds nimg*rows*Wo*3 bytes.  A pass whose size does
// not change is skipped (Wo == W: vertical only, straight from src with y0 = 0 expected; Ho == H: horizontal only, straight to dst).
int tuber_frames_resize(const void* src, void* tmp, void* dst, int nimg, int H, int W, int Ho, int Wo, const void* bounds_h,
                        const void* kk_h, int ksize_h, const void* bounds_v, const void* kk_v, int ksize_v, int y0, int rows,
                        hipStream_t stream) {
    if (nimg <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return TUBER_EINVAL;
    const bool need_h = Wo != W, need_v = Ho != H;
    if (!need_h && !need_v) return TUBER_EINVAL;          // nothing to do: the caller keeps the source
    if (need_h && need_v && !tmp) return TUBER_EINVAL;
    const uint8_t* vin = (const uint8_t*)src;
    int vH = H;
    if (need_h) {
        uint8_t* hout = need_v ? (uint8_t*)tmp : (uint8_t*)dst;
        const int hy0 = need_v ? y0 : 0, hrows = need_v ? rows : H;
        if ((long)W * 3 + 4 <= HROW_MAX && (long)Wo * 3 <= HROW_MAX) {
            const long nrows = (long)nimg * hrows;
            frames_resize_h_kernel<<<(int)min(nrows, 65536L * 4), 256, 0, stream>>>(
                (const uint8_t*)src, hout, (const int*)bounds_h, (const int*)kk_h, ksize_h, nimg, H, W, Wo, hy0, hrows);
        } else {
            const long total = (long)nimg * hrows * Wo;
            frames_resize_h_direct_kernel<<<(int)min((total + 255) / 256, 65536L * 16), 256, 0, stream>>>(
                (const uint8_t*)src, hout, (const int*)bounds_h, (const int*)kk_h, ksize_h, nimg, H, W, Wo, hy0, hrows);
        }
        vin = hout;
        vH = hrows;
    }
    if (need_v) {
        const int rowb = Wo * 3;
        const uintptr_t al = (uintptr_t)vin | (uintptr_t)dst | (uintptr_t)rowb;
        const int vec = (al & 15) == 0 ? 16 : (al & 3) == 0 ? 4 : 1;
        const long total = (long)nimg * Ho * ((rowb + vec - 1) / vec);
        const int blocks = (int)min((total + 255) / 256, 65536L * 16);
        const int *bv = (const int*)bounds_v, *kv = (const int*)kk_v;
        if (vec == 16) frames_resize_v_kernel<16><<<blocks, 256, 0, stream>>>(vin, (uint8_t*)dst, bv, kv, ksize_v, nimg, vH, rowb, Ho);
        else if (vec == 4) frames_resize_v_kernel<4><<<blocks, 256, 0, stream>>>(vin, (uint8_t*)dst, bv, kv, ksize_v, nimg, vH, rowb, Ho);
        else frames_resize_v_kernel<1><<<blocks, 256, 0, stream>>>(vin, (uint8_t*)dst, bv, kv, ksize_v, nimg, vH, rowb, Ho);
    }
    TUBER_RETURN_LAUNCH();
}

// frames (uint8, the clips' [T][H][W][3] frames at desc[n].src_off) -> out fp32 [N][3][T][Hmax][Wmax] (zero padded) and
// mask uint8/bool [N][Hmax][Wmax] (1 = padding).  desc: N TuberClipDesc in device memory; lut: fp32 [3][256] ToTensor+Normalize
// table; hsv_tab: int32 [2][256] OpenCV sdiv / hdiv180 tables.
int tuber_clip_prepare(const void* frames, const void* desc, const void* lut, const void* hsv_tab, void* out, void* mask, int N, int T,
                       int Hmax, int Wmax, hipStream_t stream) {
    if (N <= 0 || T <= 0 || Hmax <= 0 || Wmax <= 0) return TUBER_EINVAL;
    const long total = (long)N * T * Hmax * ((Wmax + 3) / 4);
    const int blocks = (int)min((total + 255) / 256, 65536L * 16);
    if (Wmax % 4 == 0)
        clip_prepare_kernel<true><<<blocks, 256, 0, stream>>>((const uint8_t*)frames, (const ClipDesc*)desc, (const float*)lut,
                                                              (const int*)hsv_tab, (float*)out, (uint8_t*)mask, N, T, Hmax, Wmax);
    else
        clip_prepare_kernel<false><<<blocks, 256, 0, stream>>>((const uint8_t*)frames, (const ClipDesc*)desc, (const float*)lut,
                                                               (const int*)hsv_tab, (float*)out, (uint8_t*)mask, N, T, Hmax, Wmax);
    TUBER_RETURN_LAUNCH();
}

int tuber_clip_desc_bytes() { return (int)sizeof(ClipDesc); }

}  // extern "C"
