// Fused gradient-norm clipping + AdamW over the flat fp32 parameter / gradient buffers (gfx950).
// reference: torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1) + torch.optim.AdamW(param_groups).step()
// (utils/video_action_recognition.py:153-154, train_tuber_ava.py:41-58).  The reference touches 684 tensors with
// several passes each; here the whole model is three launches per learning-rate segment: a partial sum of squares,
// and one streaming pass that clips, decays and updates p / exp_avg / exp_avg_sq (HBM-bound: 5 streams of 4 B).
#include "common.h"

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, float* __restrict__ partial) {
    __shared__ float red[4];
    float acc = 0.f;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const float4 v = ((const float4*)g)[i];
        acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) acc += g[i] * g[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// norm_out[0] = sqrt(sum partial), norm_out[1] = clip coefficient min(1, max_norm / (norm + 1e-6)) (1 if max_norm <= 0).
// A NON-FINITE norm (a NaN / Inf anywhere in the gradient buffer: a non-finite loss, a poisoned cooperative-decoder launch, a peer rank's
// NaN arriving through the all-reduce) makes the coefficient -1 = "skip": adamw_kernel leaves parameters and moments untouched and the
// step count *step_ptr does not advance, so no invalid update is ever applied -- the reference stops BEFORE optimizer.step() on a
// non-finite loss (utils/video_action_recognition.py:195-198); here the host learns about it when it next reads the loss.
__global__ void clip_coef_kernel(const float* __restrict__ partial, int np, float max_norm, float* __restrict__ norm_out, int* __restrict__ step_ptr) {
    __shared__ double red[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) a += (double)partial[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(red[0]);
        norm_out[0] = norm;
        float c = 1.f;
        if (max_norm > 0.f) { c = max_norm / (norm + 1e-6f); if (c > 1.f) c = 1.f; }
        const bool ok = norm == norm && norm <= 3.0e38f;
        norm_out[1] = ok ? c : -1.f;
        if (step_ptr && ok) *step_ptr += 1;
    }
}

// AdamW on [begin, end): g <- g * clip;  p <- p*(1 - lr*wd);  m, v updates;  p <- p - step_size * m / (sqrt(v)/bc2s + eps)
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, const float* __restrict__ clip, float lr,
                                                    float beta1, float beta2, float eps, float wd, const int* __restrict__ step_ptr,
                                                    int write_clipped_grad, const float* __restrict__ hyper) {
    if (hyper) { lr = hyper[0]; wd = hyper[1]; }     // per-group lr / weight decay in DEVICE memory: schedulers act on replayed hipGraphs
    const float c = clip ? clip[1] : 1.f;
    if (c < 0.f) return;                             // non-finite gradient norm: the step is skipped (clip_coef_kernel)
    const float tstep = (float)(*step_ptr);
    const float bc1 = 1.f - powf(beta1, tstep), bc2_sqrt = sqrtf(1.f - powf(beta2, tstep));
    const float step_size = lr / bc1;
    const float decay = 1.f - lr * wd;
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 pv = ((float4*)p)[i], gv = ((float4*)g)[i], mv = ((float4*)m)[i], vv = ((float4*)v)[i];
        float* pp = (float*)&pv; float* gg = (float*)&gv; float* mm = (float*)&mv; float* vvp = (float*)&vv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gg[e] * c;
            gg[e] = gr;
            pp[e] *= decay;
            mm[e] = beta1 * mm[e] + (1.f - beta1) * gr;
            vvp[e] = beta2 * vvp[e] + (1.f - beta2) * gr * gr;
            pp[e] -= step_size * mm[e] / (sqrtf(vvp[e]) / bc2_sqrt + eps);
        }
        ((float4*)p)[i] = pv; ((float4*)m)[i] = mv; ((float4*)v)[i] = vv;
        if (write_clipped_grad) ((float4*)g)[i] = gv;
    }
    if (blockIdx.x == 0) {
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) {
            const float gr = g[i] * c;
            if (write_clipped_grad) g[i] = gr;
            float pv = p[i] * decay;
            const float mv = beta1 * m[i] + (1.f - beta1) * gr;
            const float vv = beta2 * v[i] + (1.f - beta2) * gr * gr;
            pv -= step_size * mv / (sqrtf(vv) / bc2_sqrt + eps);
            p[i] = pv; m[i] = mv; v[i] = vv;
        }
    }
}

__global__ void scale_kernel(float* __restrict__ x, long n, const float* __restrict__ coef_dev, float coef) {
    const float c = coef_dev ? fmaxf(coef_dev[1], 0.f) * coef : coef;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) x[i] *= c;
}

extern "C" {

#define NORM_BLOCKS 1024

// total L2 norm of g[0..n) and the clip coefficient, both left ON THE DEVICE in norm_out[0..1] (no host sync); norm_out[1] = -1 when the
// norm is not finite (tuber_adamw_segment then skips its update).  step_ptr (optional, device int): the AdamW step count, advanced by one
// iff the norm is finite.  partial must hold 1024 floats.
int tuber_grad_norm_clip_coef(const float* g, long n, float max_norm, float* partial, float* norm_out, int* step_ptr, hipStream_t stream) {
    if (n <= 0) return TUBER_EINVAL;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(NORM_BLOCKS), dim3(256), 0, stream, g, n, partial);
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(256), 0, stream, partial, NORM_BLOCKS, max_norm, norm_out, step_ptr);
    TUBER_RETURN_LAUNCH();
}

// one AdamW step on a contiguous segment; `clip` = norm_out of tuber_grad_norm_clip_coef (or NULL); clip[1] < 0 = skip (non-finite norm);
// the step count t (for the bias corrections 1 - beta^t) is read from DEVICE memory so a captured hipGraph
// replays with the right value every step; so are lr / weight_decay when `hyper` (device float[2] = {lr, weight_decay}) is given
// (lr_scheduler.step() between replays), else the by-value arguments are used.
int tuber_adamw_segment(float* p, float* g, float* exp_avg, float* exp_avg_sq, long n, const float* clip, float lr, float beta1,
                        float beta2, float eps, float weight_decay, const int* step_ptr, int write_clipped_grad,
                        const float* hyper, hipStream_t stream) {
    if (n <= 0) return TUBER_EINVAL;
    long nb = (n / 4 + 255) / 256;
    if (nb > 4096) nb = 4096;
    if (nb < 1) nb = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((int)nb), dim3(256), 0, stream, p, g, exp_avg, exp_avg_sq, n, clip, lr, beta1, beta2, eps,
                       weight_decay, step_ptr, write_clipped_grad, hyper);
    TUBER_RETURN_LAUNCH();
}

// x *= coef (* coef_dev[1] when given): gradient averaging after the all-reduce, in-place clipping
int tuber_scale_f32(float* x, long n, const float* coef_dev, float coef, hipStream_t stream) {
    if (n <= 0) return TUBER_EINVAL;
    long nb = (n + 255) / 256;
    if (nb > 4096) nb = 4096;
    hipLaunchKernelGGL(scale_kernel, dim3((int)nb), dim3(256), 0, stream, x, n, coef_dev, coef);
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
