// Hungarian-matched set criterion of TubeR on the device (fp32), all decoder layers in one launch each.
// reference: HungarianMatcher.forward (models/detr/matcher.py:37-81, matcher_ucf.py:37-88),
// SetCriterionAVA.loss_labels / loss_boxes (models/criterion.py:42-81,97-117), SetCriterion.loss_labels (:237-262),
// generalized_box_iou (utils/box_ops.py:41-65).
//
//   criterion_cost : C[l,b,q,j] = w_bbox*|box_q - box_j|_1 - w_giou*GIoU(box_q, box_j) - w_class*p   for every layer l
//                    -> ONE device-to-host copy per step feeds the host assignment solver (tuber_lsap)
//   criterion_loss : given match[l,b,j] = query matched to target j (or -1): the four loss terms of every layer
//                    AND their gradients w.r.t. logits / actor logits / boxes (the backward pass only scales them)
// Targets are padded to Tmax per clip (tcount[b] valid rows), so shapes are static: the step can be hipGraph-captured.
#include "common.h"

struct CritArgs {
    const float* logits;     // [L,B,Q,C]
    const float* logits_b;   // [L,B,Q,3] (AVA) or unused
    const float* boxes;      // [L,B,Q,4] cxcywh
    const float* tboxes;     // [B,Tmax,4] cxcywh
    const float* tlabels;    // AVA: [B,Tmax,C] multi-hot ; JHMDB: [B,Tmax] class id stored as float
    const int* tcount;       // [B]
    int L, B, Q, C, Tmax, ava;
    float w_class, w_bbox, w_giou;
    // loss kernel
    const int* match;        // [L,B,Tmax]
    float eos, pos_weight;   // EOS_COF, LOSS_COFS.WEIGHT (1 when evaluation)
    float* losses;           // [L,4] = ce, ce_b, bbox, giou
    float* g_logits;         // [L,B,Q,C]
    float* g_logits_b;       // [L,B,Q,3]
    float* g_bbox;           // [L,B,Q,4]
    float* g_giou;           // [L,B,Q,4]
};

__device__ __forceinline__ float softplus_clamped(float x) {   // min(softplus(x), 100): BCE's log clamp at -100
    const float sp = x > 20.f ? x : log1pf(__expf(x));
    return fminf(sp, 100.f);
}

// GIoU of a predicted cxcywh box p and a target t, optionally with d(1 - giou)/d(p)
__device__ __forceinline__ float giou_loss(const float* p, const float* t, float* grad) {
    const float x0 = p[0] - 0.5f * p[2], x1 = p[0] + 0.5f * p[2], y0 = p[1] - 0.5f * p[3], y1 = p[1] + 0.5f * p[3];
    const float X0 = t[0] - 0.5f * t[2], X1 = t[0] + 0.5f * t[2], Y0 = t[1] - 0.5f * t[3], Y1 = t[1] + 0.5f * t[3];
    const float A1 = (x1 - x0) * (y1 - y0), A2 = (X1 - X0) * (Y1 - Y0);
    const float iw_raw = fminf(x1, X1) - fmaxf(x0, X0), ih_raw = fminf(y1, Y1) - fmaxf(y0, Y0);
    const float iw = fmaxf(iw_raw, 0.f), ih = fmaxf(ih_raw, 0.f);
    const float I = iw * ih, U = A1 + A2 - I;
    const float cw = fmaxf(fmaxf(x1, X1) - fminf(x0, X0), 0.f), ch = fmaxf(fmaxf(y1, Y1) - fminf(y0, Y0), 0.f);
    const float Ca = cw * ch;
    const float iou = I / U;
    const float giou = iou - (Ca - U) / Ca;
    if (grad) {
        // partials w.r.t. (x0, y0, x1, y1); subgradients follow torch (max/min route to the selected operand)
        const float diw[4] = {(iw_raw > 0.f && x0 > X0) ? -1.f : 0.f, 0.f, (iw_raw > 0.f && x1 < X1) ? 1.f : 0.f, 0.f};
        const float dih[4] = {0.f, (ih_raw > 0.f && y0 > Y0) ? -1.f : 0.f, 0.f, (ih_raw > 0.f && y1 < Y1) ? 1.f : 0.f};
        const float dcw[4] = {x0 < X0 ? -1.f : 0.f, 0.f, x1 > X1 ? 1.f : 0.f, 0.f};
        const float dch[4] = {0.f, y0 < Y0 ? -1.f : 0.f, 0.f, y1 > Y1 ? 1.f : 0.f};
        const float dA1[4] = {-(y1 - y0), -(x1 - x0), (y1 - y0), (x1 - x0)};
        float d[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dI = diw[k] * ih + iw * dih[k];
            const float dU = dA1[k] - dI;
            const float dCa = dcw[k] * ch + cw * dch[k];
            const float diou = (dI * U - I * dU) / (U * U);
            const float dUC = (dU * Ca - U * dCa) / (Ca * Ca);     // d(U/Ca)
            d[k] = -(diou + dUC);                                   // loss = 2 - iou - U/Ca
        }
        grad[0] = d[0] + d[2];
        grad[1] = d[1] + d[3];
        grad[2] = 0.5f * (d[2] - d[0]);
        grad[3] = 0.5f * (d[3] - d[1]);
    }
    return 1.f - giou;
}

__global__ __launch_bounds__(256) void criterion_cost_kernel(CritArgs a, float* __restrict__ cost) {
    const int lb = blockIdx.x, l = lb / a.B, b = lb % a.B;
    const int n = a.tcount[b];
    for (int i = threadIdx.x; i < a.Q * a.Tmax; i += 256) {
        const int q = i / a.Tmax, j = i % a.Tmax;
        float c = 0.f;
        if (j < n) {
            const float* p = a.boxes + ((long)(l * a.B + b) * a.Q + q) * 4;
            const float* t = a.tboxes + ((long)b * a.Tmax + j) * 4;
            const float l1 = fabsf(p[0] - t[0]) + fabsf(p[1] - t[1]) + fabsf(p[2] - t[2]) + fabsf(p[3] - t[3]);
            const float gl = giou_loss(p, t, nullptr);            // 1 - giou
            float prob;
            if (a.ava) {
                const float* z = a.logits_b + ((long)(l * a.B + b) * a.Q + q) * 3;
                const float m = fmaxf(z[0], fmaxf(z[1], z[2]));
                const float e0 = __expf(z[0] - m), e1 = __expf(z[1] - m), e2 = __expf(z[2] - m);
                prob = e1 / (e0 + e1 + e2);
            } else {
                const float* z = a.logits + ((long)(l * a.B + b) * a.Q + q) * a.C;
                const int cls = (int)a.tlabels[(long)b * a.Tmax + j];
                float m = -INFINITY;
                for (int k = 0; k < a.C; ++k) m = fmaxf(m, z[k]);
                float s = 0.f;
                for (int k = 0; k < a.C; ++k) s += __expf(z[k] - m);
                prob = __expf(z[cls] - m) / s;
            }
            c = a.w_bbox * l1 + a.w_giou * (gl - 1.f) - a.w_class * prob;
        }
        cost[((long)(l * a.B + b) * a.Q + q) * a.Tmax + j] = c;
    }
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// one block per decoder layer
__global__ __launch_bounds__(256) void criterion_loss_kernel(CritArgs a) {
    extern __shared__ int tq[];          // [B*Q] matched target of (b,q) or -1
    __shared__ float red[4];
    const int l = blockIdx.x;
    const int BQ = a.B * a.Q;
    for (int i = threadIdx.x; i < BQ; i += 256) tq[i] = -1;
    __syncthreads();
    int nb_local = 0;
    for (int i = threadIdx.x; i < a.B * a.Tmax; i += 256) {
        const int b = i / a.Tmax, j = i % a.Tmax;
        if (j < a.tcount[b]) {
            ++nb_local;
            const int q = a.match[((long)l * a.B + b) * a.Tmax + j];
            if (q >= 0 && q < a.Q) tq[b * a.Q + q] = j;
        }
    }
    const float num_boxes = fmaxf(block_sum((float)nb_local, red), 1.f);
    int nm_local = 0;
    for (int i = threadIdx.x; i < BQ; i += 256) nm_local += tq[i] >= 0;
    const float nmatched = block_sum((float)nm_local, red);
    const long base = (long)l * BQ;

    // ---- actor / no-object classification -------------------------------------------------
    float ce_b = 0.f, ce = 0.f;
    if (a.ava) {
        const float wsum = nmatched + (BQ - nmatched) * a.eos;        // targets: 1 (matched, w 1) / 2 (unmatched, w eos)
        for (int i = threadIdx.x; i < BQ; i += 256) {
            const float* z = a.logits_b + (base + i) * 3;
            const int t = tq[i] >= 0 ? 1 : 2;
            const float w = t == 2 ? a.eos : 1.f;
            const float m = fmaxf(z[0], fmaxf(z[1], z[2]));
            const float e0 = __expf(z[0] - m), e1 = __expf(z[1] - m), e2 = __expf(z[2] - m), s = e0 + e1 + e2;
            const float p[3] = {e0 / s, e1 / s, e2 / s};
            ce_b += w * -(z[t] - m - __logf(s));
            float* g = a.g_logits_b + (base + i) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) g[k] = w * (p[k] - (k == t ? 1.f : 0.f)) / wsum;
        }
        ce_b = block_sum(ce_b, red) / wsum;
        // weighted multi-label BCE on logits (== F.binary_cross_entropy(sigmoid(x), t, w), mean over B*Q*C)
        const float inv = 1.f / ((float)BQ * a.C);
        for (long i = threadIdx.x; i < (long)BQ * a.C; i += 256) {
            const int bq = (int)(i / a.C), c = (int)(i % a.C);
            const int j = tq[bq];
            const float x = a.logits[(base + bq) * a.C + c];
            const float t = j >= 0 ? a.tlabels[((long)(bq / a.Q) * a.Tmax + j) * a.C + c] : 0.f;
            const float w = j >= 0 ? a.pos_weight : 1.f;
            ce += w * (t * softplus_clamped(-x) + (1.f - t) * softplus_clamped(x));
            const float sg = 1.f / (1.f + __expf(-x));
            a.g_logits[(base + bq) * a.C + c] = w * (sg - t) * inv;
        }
        ce = block_sum(ce, red) * inv;
    } else {
        // (C)-way CE with the last class = no-object weighted eos  (C here already counts the no-object slot)
        const float wsum = nmatched + (BQ - nmatched) * a.eos;
        for (int i = threadIdx.x; i < BQ; i += 256) {
            const float* z = a.logits + (base + i) * a.C;
            const int j = tq[i];
            const int t = j >= 0 ? (int)a.tlabels[(long)(i / a.Q) * a.Tmax + j] : a.C - 1;
            const float w = t == a.C - 1 ? a.eos : 1.f;
            float m = -INFINITY;
            for (int k = 0; k < a.C; ++k) m = fmaxf(m, z[k]);
            float s = 0.f;
            for (int k = 0; k < a.C; ++k) s += __expf(z[k] - m);
            ce += w * -(z[t] - m - __logf(s));
            float* g = a.g_logits + (base + i) * a.C;
            for (int k = 0; k < a.C; ++k) g[k] = w * (__expf(z[k] - m) / s - (k == t ? 1.f : 0.f)) / wsum;
        }
        ce = block_sum(ce, red) / wsum;
    }

    // ---- boxes ---------------------------------------------------------------------------------
    float l1 = 0.f, gi = 0.f;
    for (int i = threadIdx.x; i < BQ; i += 256) {
        float* gb = a.g_bbox + (base + i) * 4;
        float* gg = a.g_giou + (base + i) * 4;
        const int j = tq[i];
        if (j < 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { gb[k] = 0.f; gg[k] = 0.f; }
            continue;
        }
        const float* p = a.boxes + (base + i) * 4;
        const float* t = a.tboxes + ((long)(i / a.Q) * a.Tmax + j) * 4;
        float g4[4];
        gi += giou_loss(p, t, g4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float d = p[k] - t[k];
            l1 += fabsf(d);
            gb[k] = (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / num_boxes;
            gg[k] = g4[k] / num_boxes;
        }
    }
    l1 = block_sum(l1, red) / num_boxes;
    gi = block_sum(gi, red) / num_boxes;
    if (threadIdx.x == 0) {
        a.losses[l * 4 + 0] = ce;
        a.losses[l * 4 + 1] = ce_b;
        a.losses[l * 4 + 2] = l1;
        a.losses[l * 4 + 3] = gi;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Hungarian assignment ON THE DEVICE: one thread per (decoder layer, clip) problem -- the same shortest-augmenting-path
// algorithm with SciPy's tie-breaking as the host tuber_lsap (lsap.cpp; call sites models/detr/matcher.py:80,
// matcher_ucf.py:82), in double precision on the fp32 costs, so the result is identical.  The problems are tiny
// (15 queries x <= a few targets), so the point is not speed but removing the step's only device->host round trip:
// with the assignment on the device the whole training step is ONE hipGraph.
// ---------------------------------------------------------------------------------------------------------------------
#define LSAP_MAX 128
#define LSAP_CACHE 1024
__global__ void lsap_kernel(const float* __restrict__ cost, const int* __restrict__ tcount, int* __restrict__ match,
                            int L, int B, int Q, int Tmax) {
    // one 64-thread block per problem: the lanes stage the costs and clear the output, lane 0 runs the (inherently serial)
    // augmenting-path search out of LDS (its latency, not private scratch in HBM, is what a serial GPU thread pays per access)
    const int prob = blockIdx.x;
    const int b = prob % B;
    const float* C = cost + (long)prob * Q * Tmax;
    int* out = match + (long)prob * Tmax;
    __shared__ double u[LSAP_MAX], v[LSAP_MAX], spc[LSAP_MAX], cs[LSAP_CACHE];
    __shared__ int path[LSAP_MAX], col4row[LSAP_MAX], row4col[LSAP_MAX], remaining[LSAP_MAX];
    __shared__ bool SR[LSAP_MAX], SC[LSAP_MAX];
    for (int t = threadIdx.x; t < Tmax; t += blockDim.x) out[t] = -1;
    const int T = tcount[b];
    if (T <= 0 || Q <= 0) return;
    // rows = the smaller side (SciPy transposes when there are more rows than columns)
    const bool tr = T < Q;
    const int nr = tr ? T : Q, nc = tr ? Q : T;
    const bool cached = nr * nc <= LSAP_CACHE;
    if (cached)
        for (int e = threadIdx.x; e < nr * nc; e += blockDim.x) {
            const int i = e / nc, j = e % nc;
            cs[e] = tr ? (double)C[(long)j * Tmax + i] : (double)C[(long)i * Tmax + j];
        }
    __syncthreads();
    if (threadIdx.x != 0) return;
    auto cst = [&](int i, int j) -> double {
        return cached ? cs[i * nc + j] : (tr ? (double)C[(long)j * Tmax + i] : (double)C[(long)i * Tmax + j]);
    };
    const double INF = __longlong_as_double(0x7ff0000000000000LL);
    for (int i = 0; i < nr; ++i) { u[i] = 0.0; col4row[i] = -1; }
    for (int j = 0; j < nc; ++j) { v[j] = 0.0; row4col[j] = -1; path[j] = -1; }
    for (int cur = 0; cur < nr; ++cur) {
        double min_val = 0.0;
        int num_remaining = nc, i = cur, sink = -1;
        for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
        for (int k = 0; k < nr; ++k) SR[k] = false;
        for (int k = 0; k < nc; ++k) { SC[k] = false; spc[k] = INF; }
        while (sink == -1) {
            int index = -1;
            double lowest = INF;
            SR[i] = true;
            for (int it = 0; it < num_remaining; ++it) {
                const int j = remaining[it];
                const double r = min_val + cst(i, j) - u[i] - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
            }
            min_val = lowest;
            if (min_val == INF) return;                      // infeasible (NaN / inf costs): leave the clip unmatched
            const int j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC[j] = true;
            remaining[index] = remaining[--num_remaining];
        }
        u[cur] += min_val;
        for (int k = 0; k < nr; ++k)
            if (SR[k] && k != cur) u[k] += min_val - spc[col4row[k]];
        for (int k = 0; k < nc; ++k)
            if (SC[k]) v[k] -= min_val - spc[k];
        int j = sink;
        while (true) {
            const int k = path[j];
            row4col[j] = k;
            const int tmp = col4row[k]; col4row[k] = j; j = tmp;
            if (k == cur) break;
        }
    }
    // match[target] = query
    for (int i = 0; i < nr; ++i) {
        if (tr) out[i] = col4row[i];          // rows are targets, columns queries
        else out[col4row[i]] = i;             // rows are queries, columns targets
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// class_error of the matched queries (utils/misc.py:497-518 accuracy_sigmoid / :521-539 accuracy, logged by criterion.py:76-78,
// 258-260): 100 - 100 * (#matched queries whose prediction is exactly right) / (#matched queries), one workgroup, no host sync.
//   AVA (multi-label):  right <=> top-k(logits) == label set <=> min logit over the labels > max logit over the rest
//   JHMDB (single label): right <=> argmax(logits) == label
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void class_error_kernel(const float* __restrict__ logits, const int* __restrict__ match,
                                                           const float* __restrict__ tlabels, int B, int Q, int C, int Tmax, int ava,
                                                           float* __restrict__ out) {
    __shared__ int sh_ok[256], sh_n[256];
    int ok = 0, n = 0;
    for (int i = threadIdx.x; i < B * Tmax; i += 256) {
        const int q = match[i];
        if (q < 0) continue;
        const int b = i / Tmax;
        const float* row = logits + ((long)b * Q + q) * C;
        ++n;
        if (ava) {
            const float* lab = tlabels + (long)i * C;
            float lo = INFINITY, hi = -INFINITY;
            for (int c = 0; c < C; ++c) {
                const float v = row[c];
                if (lab[c] > 0.5f) lo = fminf(lo, v); else hi = fmaxf(hi, v);
            }
            ok += lo > hi;
        } else {
            int am = 0;
            float best = row[0];
            for (int c = 1; c < C; ++c) if (row[c] > best) { best = row[c]; am = c; }
            ok += am == (int)tlabels[i];
        }
    }
    sh_ok[threadIdx.x] = ok; sh_n[threadIdx.x] = n;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { sh_ok[threadIdx.x] += sh_ok[threadIdx.x + s]; sh_n[threadIdx.x] += sh_n[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = 100.f - 100.f * (float)sh_ok[0] / (float)(sh_n[0] > 0 ? sh_n[0] : 1);
}

// key-padding mask of the feature grid: nearest-neighbour resize of the clip padding mask, F.interpolate(mask[None].float(),
// size=(h, w)).to(bool) (models/backbone_builder.py:85-86): src = min(floor(dst * (float)in / out), in - 1), ATen's rule
__global__ void mask_resize_kernel(const uint8_t* __restrict__ mask, uint8_t* __restrict__ out, int B, int H, int W, int h, int w,
                                   float sy, float sx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * h * w) return;
    const int x = i % w, y = (i / w) % h, b = i / (w * h);
    const int yy = min((int)floorf(y * sy), H - 1), xx = min((int)floorf(x * sx), W - 1);
    out[i] = mask[((long)b * H + yy) * W + xx] ? 1 : 0;
}


// ---- padded target layout of one batch in ONE launch (criterion.PaddedTargets.refill) ----------------------------------------------
// The captured step reads the targets from static [B, Tmax] device buffers; refilling them from the per-clip target tensors was two
// memsets and two sliced copies per clip (7+ launches a step).  The per-clip pointers travel by value in the kernel arguments.
#define TP_MAX_CLIPS 16
struct TargetPack {
    const float* boxes[TP_MAX_CLIPS];      // [n_b][5] fp32 (column 0 = key-frame index, dropped: matcher.py:64)
    const void* labels[TP_MAX_CLIPS];      // AVA: [n_b][C] fp32 multi-hot;  JHMDB: [n_b] int64 class ids
    int n[TP_MAX_CLIPS];
    int B, Tmax, C, ava;
    float* tboxes; float* tlabels; int* tcount;
};
__global__ void targets_pack_kernel(TargetPack a) {
    const int b = blockIdx.x;
    const int n = a.n[b];
    if (threadIdx.x == 0) a.tcount[b] = n;
    for (int i = threadIdx.x; i < a.Tmax * 4; i += blockDim.x) {
        const int t = i >> 2, c = i & 3;
        a.tboxes[((long)b * a.Tmax + t) * 4 + c] = t < n ? a.boxes[b][t * 5 + 1 + c] : 0.f;
    }
    if (a.ava) {
        for (int i = threadIdx.x; i < a.Tmax * a.C; i += blockDim.x) {
            const int t = i / a.C;
            a.tlabels[(long)b * a.Tmax * a.C + i] = t < n ? ((const float*)a.labels[b])[i] : 0.f;
        }
    } else {
        for (int t = threadIdx.x; t < a.Tmax; t += blockDim.x)
            a.tlabels[(long)b * a.Tmax + t] = t < n ? (float)((const long long*)a.labels[b])[t] : 0.f;
    }
}

__global__ __launch_bounds__(256) void criterion_scale_kernel(const float* __restrict__ g, const float* __restrict__ g_l, const float* __restrict__ g_b,
                                                              const float* __restrict__ g_x, const float* __restrict__ g_g, long nl, long nb, long nx,
                                                              float* __restrict__ gl, float* __restrict__ gb, float* __restrict__ gx) {
    const int l = blockIdx.y;
    const float c0 = g[l * 4 + 0], c1 = g[l * 4 + 1], c2 = g[l * 4 + 2], c3 = g[l * 4 + 3];
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < nl) { gl[l * nl + i] = c0 * g_l[l * nl + i]; return; }
    i -= nl;
    if (i < nb) { gb[l * nb + i] = c1 * g_b[l * nb + i]; return; }
    i -= nb;
    if (i < nx) gx[l * nx + i] = __fadd_rn(__fmul_rn(c2, g_x[l * nx + i]), __fmul_rn(c3, g_g[l * nx + i]));      // (two rounded products, as the ATen ops formed it)
}

__global__ __launch_bounds__(256) void weighted_sum_kernel(const float* __restrict__ a, const float* __restrict__ w, int n, float* __restrict__ out,
                                                           const float* __restrict__ gout, float* __restrict__ out_g) {
    if (gout) {
        const float c = gout[0];
        for (int i = threadIdx.x; i < n; i += 256) out_g[i] = c * w[i];
        return;
    }
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += a[i] * w[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

extern "C" {

int tuber_criterion_cost(const float* logits, const float* logits_b, const float* boxes, const float* tboxes, const float* tlabels,
                         const int* tcount, int L, int B, int Q, int C, int Tmax, int ava, float w_class, float w_bbox, float w_giou,
                         float* cost, hipStream_t stream) {
    if (L <= 0 || B <= 0 || Q <= 0 || C <= 0 || Tmax <= 0) return TUBER_EINVAL;
    CritArgs a{};
    a.logits = logits; a.logits_b = logits_b; a.boxes = boxes; a.tboxes = tboxes; a.tlabels = tlabels; a.tcount = tcount;
    a.L = L; a.B = B; a.Q = Q; a.C = C; a.Tmax = Tmax; a.ava = ava; a.w_class = w_class; a.w_bbox = w_bbox; a.w_giou = w_giou;
    hipLaunchKernelGGL(criterion_cost_kernel, dim3(L * B), dim3(256), 0, stream, a, cost);
    TUBER_RETURN_LAUNCH();
}

int tuber_criterion_loss(const float* logits, const float* logits_b, const float* boxes, const float* tboxes, const float* tlabels,
                         const int* tcount, const int* match, int L, int B, int Q, int C, int Tmax, int ava, float eos,
                         float pos_weight, float* losses, float* g_logits, float* g_logits_b, float* g_bbox, float* g_giou,
                         hipStream_t stream) {
    if (L <= 0 || B <= 0 || Q <= 0 || C <= 0 || Tmax <= 0 || (long)B * Q > 12000) return TUBER_EINVAL;
    CritArgs a{};
    a.logits = logits; a.logits_b = logits_b; a.boxes = boxes; a.tboxes = tboxes; a.tlabels = tlabels; a.tcount = tcount;
    a.match = match; a.L = L; a.B = B; a.Q = Q; a.C = C; a.Tmax = Tmax; a.ava = ava; a.eos = eos; a.pos_weight = pos_weight;
    a.losses = losses; a.g_logits = g_logits; a.g_logits_b = g_logits_b; a.g_bbox = g_bbox; a.g_giou = g_giou;
    hipLaunchKernelGGL(criterion_loss_kernel, dim3(L), dim3(256), (size_t)B * Q * sizeof(int), stream, a);
    TUBER_RETURN_LAUNCH();
}

// backward of tuber_criterion_loss's [L][4] loss table in ONE launch: the stored per-layer gradients scaled by the incoming gradient g[L][4]
//   gl = g[:,0] * g_logits;  gb = g[:,1] * g_logits_b (ava);  gx = g[:,2] * g_bbox + g[:,3] * g_giou
// (five elementwise ATen launches before).  nl / nb / nx = elements per layer of the three tensors.
int tuber_criterion_scale(const float* g, const float* g_logits, const float* g_logits_b, const float* g_bbox, const float* g_giou, int L, long nl,
                          long nb, long nx, float* gl, float* gb, float* gx, hipStream_t stream) {
    if (L <= 0 || nl <= 0 || nx <= 0 || !g || !gl || !gx || (gb && !g_logits_b)) return TUBER_EINVAL;
    const long per = nl + (gb ? nb : 0) + nx;
    hipLaunchKernelGGL(criterion_scale_kernel, dim3((unsigned)ceil_div(per, 256), L), dim3(256), 0, stream, g, g_logits, g_logits_b, g_bbox, g_giou,
                       nl, gb ? nb : 0, nx, gl, gb, gx);
    TUBER_RETURN_LAUNCH();
}

// out[0] = sum_i a[i] * w[i] (n <= 4096, fixed summation order) and, for the backward, out_g[i] = gout[0] * w[i] when gout is given instead of a
// (`(losses * W).sum()` and its autograd: two + one ATen launches before)
int tuber_weighted_sum(const float* a, const float* w, int n, float* out, const float* gout, float* out_g, hipStream_t stream) {
    if (n <= 0 || n > 4096 || !w || (!a && !gout) || (a && !out) || (gout && !out_g)) return TUBER_EINVAL;
    hipLaunchKernelGGL(weighted_sum_kernel, dim3(1), dim3(256), 0, stream, a, w, n, out, gout, out_g);
    TUBER_RETURN_LAUNCH();
}

// match[l][b][t] = query assigned to target t of clip b by decoder layer l (-1 beyond tcount[b]); cost is the [L,B,Q,Tmax] fp32
// tensor of tuber_criterion_cost.  Q and Tmax must not exceed 128 (callers fall back to the host tuber_lsap otherwise).
int tuber_lsap_device(const float* cost, const int* tcount, int* match, int L, int B, int Q, int Tmax, hipStream_t stream) {
    if (L <= 0 || B <= 0 || Q <= 0 || Tmax <= 0 || Q > LSAP_MAX || Tmax > LSAP_MAX) return TUBER_EINVAL;
    hipLaunchKernelGGL(lsap_kernel, dim3(L * B), dim3(64), 0, stream, cost, tcount, match, L, B, Q, Tmax);
    TUBER_RETURN_LAUNCH();
}

// class_error [1] (fp32, device) of one decoder layer's matched queries; logits [B,Q,C] fp32, match [B,Tmax] (query or -1),
// tlabels [B,Tmax,C] multi-hot (ava) or [B,Tmax] class index stored as float (jhmdb)
int tuber_class_error(const float* logits, const int* match, const float* tlabels, int B, int Q, int C, int Tmax, int ava, float* out,
                      hipStream_t stream) {
    if (B <= 0 || Q <= 0 || C <= 0 || Tmax <= 0 || !out) return TUBER_EINVAL;
    hipLaunchKernelGGL(class_error_kernel, dim3(1), dim3(256), 0, stream, logits, match, tlabels, B, Q, C, Tmax, ava, out);
    TUBER_RETURN_LAUNCH();
}

// mask [B,H,W] (bool / uint8, nonzero = padding) -> out [B,h,w] uint8
// boxes / labels / sizes: HOST arrays of B device pointers / target counts (B <= tuber_targets_pack_max()); writes tboxes [B][Tmax][4],
// tlabels [B][Tmax][C] (AVA) or [B][Tmax] (class ids as float), tcount [B] -- zero padding included
int tuber_targets_pack_max(void) { return TP_MAX_CLIPS; }
int tuber_targets_pack(const void* const* boxes, const void* const* labels, const int* sizes, int B, int Tmax, int C, int ava,
                       float* tboxes, float* tlabels, int* tcount, hipStream_t stream) {
    if (B <= 0 || B > TP_MAX_CLIPS || Tmax <= 0 || C <= 0 || !boxes || !labels || !sizes || !tboxes || !tlabels || !tcount) return TUBER_EINVAL;
    TargetPack a{};
    for (int b = 0; b < B; ++b) {
        if (sizes[b] < 0 || sizes[b] > Tmax || (sizes[b] > 0 && (!boxes[b] || !labels[b]))) return TUBER_EINVAL;
        a.boxes[b] = (const float*)boxes[b]; a.labels[b] = labels[b]; a.n[b] = sizes[b];
    }
    a.B = B; a.Tmax = Tmax; a.C = C; a.ava = ava; a.tboxes = tboxes; a.tlabels = tlabels; a.tcount = tcount;
    hipLaunchKernelGGL(targets_pack_kernel, dim3(B), dim3(256), 0, stream, a);
    TUBER_RETURN_LAUNCH();
}

int tuber_mask_resize(const void* mask, void* out, int B, int H, int W, int h, int w, hipStream_t stream) {
    if (B <= 0 || H <= 0 || W <= 0 || h <= 0 || w <= 0) return TUBER_EINVAL;
    const int n = B * h * w;
    hipLaunchKernelGGL(mask_resize_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, (const uint8_t*)mask, (uint8_t*)out, B, H, W, h, w,
                       (float)H / (float)h, (float)W / (float)w);
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
