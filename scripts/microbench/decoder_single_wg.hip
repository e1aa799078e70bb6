// The DETR decoder stack (6 post-norm layers over <= 16 tubelet queries per clip) as ONE launch -- gfx950.
// reference: TransformerDecoder.forward / TransformerDecoderLayer.forward_post, models/transformer/transformer.py:99-128,218-249
// (self-attention with q = k = tgt + query_pos, cross-attention over the encoder memory, FFN 256 -> 2048 -> 256, three LayerNorms,
// the shared decoder.norm applied to every layer's output).
//
// Why one launch: per layer the unfused path is 13 launches of 1-12 workgroups on 30 rows x 256 (in-projection, attention, out-projection,
// LayerNorm, ... -- ~80 launches forward for the stack, each a dependent 4-11 us step on an otherwise idle chip: 0.46 ms of the
// profiled step for 0.8 GFLOP).  Here a workgroup of 16 waves owns the rows of two clips (2 x 16 query slots) for the whole stack:
//   * the residual stream stays in LDS as FP32 (the 256-wide decoder state was rounded to bf16 eleven times per layer;
//     now only the MFMA operands are bf16), together with query_pos and the bf16 operand images the GEMMs read;
//   * every linear layer is "activation rows (16 per clip) x weight rows streamed from L2": the weight fragment of a 16x16x32 MFMA is
//     ONE 16-byte global load per lane straight from the row-major bf16 weight (no LDS staging), issued 8 k-steps ahead; the two clips
//     reuse each fragment;
//   * attention is the transposed-product scheme of attention_mfma.hip (scores as MFMA rows = keys, columns = queries, so the
//     probabilities are already the B operand of P.V): one wave per (clip, head); the encoder-memory keys come straight from the packed
//     [k | v] projection in global memory, the V tiles through a wave-private LDS image read with ds_read_b64_tr_b16;
//   * LayerNorm(Dropout(sublayer) + residual) is an in-place pass over the fp32 rows (one wave per row).
// The encoder-memory projections [(memory + pos) W_k | memory W_v] of the six layers do not depend on the decoder state: they stay
// separate GEMM launches in FRONT of this kernel (tape.in_proj), and so do all weight gradients behind the backward kernel.
#include "common.h"

namespace {

constexpr int E = 256, NH = 8, FF = 2048, MAXL = 6;
constexpr int NW = 8, NT = NW * 64;       // waves / threads per workgroup (256 VGPRs per lane: the weight ring needs 128)
constexpr int RT = 2, R = RT * 16;        // clips (16-row tiles) per workgroup
constexpr int PA = 264;                   // bf16 pitch of a [R][256] operand image: 528-byte rows, 16-byte reads of 16 rows hit 64 distinct banks
constexpr int PQ = 776;                   // ... of the [R][768] q | k | v image
constexpr int PH = 1032;                  // ... of one 1024-wide half of the FFN hidden image
constexpr int VP = 40;                    // ... of a staged [32 keys][32 dims] V tile (as attention_mfma.hip)
constexpr float LN_EPS = 1e-5f;

struct LayerW {
    const bf16* w_in; const float* b_in;      // self_attn.in_proj_weight [768][256], bias [768]
    const bf16* w_o1; const float* b_o1;      // self_attn.out_proj
    const float* g1; const float* e1;         // norm1 weight / bias
    const bf16* w_q; const float* b_q;        // multihead_attn.in_proj rows [0, 256)
    const bf16* w_o2; const float* b_o2;      // multihead_attn.out_proj
    const float* g2; const float* e2;
    const bf16* w_f1; const float* b_f1;      // linear1 [2048][256]
    const bf16* w_f2; const float* b_f2;      // linear2 [256][2048]
    const float* g3; const float* e3;
    const bf16* kv;                           // [B * Lm][ldkv]: (memory + pos) W_k | memory W_v of this layer
};

struct DecArgs {
    LayerW L[MAXL];
    int nl;
    const float* gN; const float* eN;         // decoder.norm
    const bf16* qpos;                         // query_embed rows as bf16 [Q][256]
    long ldkv;
    const uint8_t* kpm;                       // [B][Lm], 1 = padded key (or null)
    int B, Q, Lm;
    bf16* hs;                                 // [nl][B][Q][256]
    float* hs32;                              // the same rows in fp32 (or null)
    float inv_keep, inv_keep_a; uint32_t th, th_a;      // dropout (th = p * 2^32; 0 = off): sublayer outputs / attention weights
    const uint64_t* seed_ptr; uint64_t salt;
};

__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float group_max(float v) { return xor32_max(xor16_max(v)); }
__device__ __forceinline__ float group_sum(float v) { return xor32_sum(xor16_sum(v)); }

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
// A operand "column d0 + li of rows {r0 + g*4 .. +3} and {r1 + g*4 .. +3}" of a ROW-MAJOR bf16 image (pitch in elements) through the
// gfx950 LDS transpose read -- the slot order of the probabilities a lane holds after the score MFMAs (attention_mfma.hip)
__device__ __forceinline__ bf16x8 frag_tr(const bf16* img, int pitch, int d0, int r0, int r1, int li, int g) {
    const bf16* p0 = img + (long)(r0 + g * 4 + (li >> 2)) * pitch + d0 + (li & 3) * 4;
    const bf16* p1 = img + (long)(r1 + g * 4 + (li >> 2)) * pitch + d0 + (li & 3) * 4;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p0);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p1);
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// ---- the weight stream ----
// Every linear layer is "activation rows in LDS x weight rows streamed from L2": the weight fragment of a 16x16x32 MFMA is ONE 16-byte
// global load per lane from the row-major bf16 weight.  A UNIT = one 16-feature column tile over 256 k (8 fragments = 8 KB per wave).  The 44
// units a wave consumes per layer (6 in-projection tiles, 2 + 2 + 2 for the out- / query- / out-projections, and per 1024-wide FFN half 8
// linear1 tiles and 2 x 4 k-chunks of linear2) form a fixed sequence that does not depend on the activations, so the loads run a RING of
// four units (32 fragments, 128 VGPRs) AHEAD of the MFMAs, across the workgroup barriers and the attention phases: the first version
// loaded a tile and waited for it -- one exposed L2 round trip per tile, 847 us for the stack against ~380 us for the launches it replaced.
constexpr int UNITS = 44, RING = 4;
struct Ring { uint4 f[RING][8]; };

// wave-uniform base address of unit u of a layer (wave w; weight row ct*16, k offset k0); the lane adds its own 32-bit offset
// (row li, k g*8: offE for the 256-wide weights, offF for linear2) -- addresses formed per unit from 64-bit lane values were hoisted
// out of the layer loop by the compiler and spilled (60-140 VGPRs in scratch; every reload drains the load queue the ring lives in)
__device__ __forceinline__ const char* unit_base(const LayerW& W, int u, int w) {
    if (u < 6) return (const char*)(W.w_in + (long)(w + NW * u) * 16 * E);
    if (u < 8) return (const char*)(W.w_o1 + (long)(w + NW * (u - 6)) * 16 * E);
    if (u < 10) return (const char*)(W.w_q + (long)(w + NW * (u - 8)) * 16 * E);
    if (u < 12) return (const char*)(W.w_o2 + (long)(w + NW * (u - 10)) * 16 * E);
    const int half = (u - 12) / 16, v = (u - 12) % 16;
    if (v < 8) return (const char*)(W.w_f1 + (long)(half * 1024 + (w + NW * v) * 16) * E);
    const int t = (v - 8) >> 2, kc = (v - 8) & 3;
    return (const char*)(W.w_f2 + (long)(w + NW * t) * 16 * FF + half * 1024 + kc * 256);
}
__device__ __forceinline__ bool unit_is_f2(int u) { return u >= 12 && ((u - 12) % 16) >= 8; }
template <int SLOT>
__device__ __forceinline__ void ring_issue(Ring& r, const char* base, unsigned lane_off) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) r.f[SLOT][kk] = *(const uint4*)(base + lane_off + kk * 64);
    __builtin_amdgcn_sched_barrier(0);
}
// acc[rt] += W_unit . X^T: issued swapped (weights = MFMA A operand, activations = B), so lane (li, g) holds
// acc[rt][r] = out[row rt*16 + li][feature ct*16 + g*4 + r] -- four consecutive features of one activation row
template <int SLOT>
__device__ __forceinline__ void ring_consume(const Ring& r, const bf16* xs, int pitch, int k0, int li, int g, f32x4 (&acc)[RT]) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const bf16x8 x = as_bf16x8(*(const uint4*)(xs + (long)(rt * 16 + li) * pitch + k0 + kk * 32 + g * 8));
            acc[rt] = mfma(as_bf16x8(r.f[SLOT][kk]), x, acc[rt]);
        }
    }
    __builtin_amdgcn_sched_barrier(0);      // (the scheduler otherwise hoists the operand reads of several units and spills the ring)
}

// Dropout masks: keep element idx of (layer, site) iff hash_u32(idx ^ key) >= thresh -- the stream of common.h's dropout_keep for 32-bit
// indices with the seed folded into one wave-uniform 32-bit key per site (64-bit index arithmetic per element cost ~100 VGPRs here)
__device__ __forceinline__ uint32_t site_key(const DecArgs& a, int layer, int site) {
    const uint64_t seed = (a.seed_ptr ? *a.seed_ptr : 0ull) * 0x9E3779B97F4A7C15ull + a.salt + (uint64_t)(layer * 16 + site) * 0x100000001B3ull;
    return __builtin_amdgcn_readfirstlane(hash_u32((uint32_t)seed) ^ ((uint32_t)(seed >> 32) * 0x9E3779B9U));
}
__device__ __forceinline__ bool keep32(uint32_t key, uint32_t idx, uint32_t thresh) { return hash_u32(idx ^ key) >= thresh; }

// ---- attention of one (clip, head) by one wave.  q: LDS image rows of the clip (16 queries x 32 dims at q + h*32).
// SELF: keys / values are 16 rows of the same LDS image (k at +256, v at +512): one 16-key tile, slots 4..7 of the P.V product are empty.
// !SELF: keys / values of the encoder memory in global memory (row (b*Lm + key)*ldkv, k at h*32, v at 256 + h*32), 32 keys per step, the
//        V tile staged in this wave's LDS image.  Output: 16 x 32 bf16 into out[row][h*32 ..] (LDS, pitch PA).
template <bool SELF, bool DROP>
__device__ __forceinline__ void attention_head(const DecArgs& a, const bf16* qimg, int qpitch, const bf16* kvimg, const bf16* kvg, long ldkv,
                                               const uint8_t* kpm_b, int Lk, bf16* vtile, bf16* out, int h, int lane, uint32_t key,
                                               uint32_t rbase_q_stride, int bglob) {
    const int li = lane & 15, g = lane >> 4;
    const float scale = 0.17677669529663687f;      // 32^-0.5
    const bf16x8 qf = as_bf16x8(*(const uint4*)(qimg + (long)li * qpitch + h * 32 + g * 8));
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    float mx = -INFINITY, l = 0.f;
    const uint32_t rbase = ((uint32_t)(bglob * NH + h) * 16 + li) * rbase_q_stride;
    const int kstep = SELF ? 16 : 32;
    // cross-attention: the K fragments and the V rows of step i + 1 are in flight while step i is computed (32 keys per step; the
    // encoder memory of a clip is 352 keys = 11 dependent steps)
    uint4 kn[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)}, vn[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
    auto fetch_step = [&](int k0) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int key = min(k0 + t * 16 + li, Lk - 1);
            kn[t] = *(const uint4*)(kvg + (long)key * ldkv + h * 32 + g * 8);
        }
        const int key = min(k0 + (lane >> 1), Lk - 1);      // (rows beyond Lk: a finite copy of the last row; their probabilities are zero)
        const uint4* pv = (const uint4*)(kvg + (long)key * ldkv + 256 + h * 32 + (lane & 1) * 16);
        vn[0] = pv[0]; vn[1] = pv[1];
    };
    if (!SELF) fetch_step(0);
    for (int k0 = 0; k0 < Lk; k0 += kstep) {
        f32x4 s[2];
        if (SELF) {
            const bf16x8 kf = as_bf16x8(*(const uint4*)(kvimg + (long)li * PQ + 256 + h * 32 + g * 8));
            s[0] = mfma(kf, qf, f32x4{0.f, 0.f, 0.f, 0.f});
            s[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
            // park the V tile [32 keys][32 dims]: lane -> key lane >> 1, 32-byte half lane & 1
            uint4* d = (uint4*)(vtile + (long)(lane >> 1) * VP + (lane & 1) * 16);
            d[0] = vn[0]; d[1] = vn[1];
            const uint4 kc0 = kn[0], kc1 = kn[1];
            if (k0 + kstep < Lk) fetch_step(k0 + kstep);
            s[0] = mfma(as_bf16x8(kc0), qf, f32x4{0.f, 0.f, 0.f, 0.f});
            s[1] = mfma(as_bf16x8(kc1), qf, f32x4{0.f, 0.f, 0.f, 0.f});
            __builtin_amdgcn_wave_barrier();
        }
        float p[8];
        float cmax = -INFINITY;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = k0 + t * 16 + g * 4 + r;
                const bool dead = (SELF && t == 1) || key >= Lk || (kpm_b && kpm_b[min(key, Lk - 1)]);
                p[t * 4 + r] = dead ? -INFINITY : s[t][r] * scale;
                cmax = fmaxf(cmax, p[t * 4 + r]);
            }
        cmax = group_max(cmax);
        const float mnew = fmaxf(mx, cmax);
        const float alpha = mnew == -INFINITY ? 1.f : __expf(mx - mnew);
        float ls = 0.f;
        bf16x8 pf;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float pe = mnew == -INFINITY ? 0.f : __expf(p[e] - mnew);
            ls += pe;
            if (DROP) {
                const int kk = k0 + (e >> 2) * 16 + g * 4 + (e & 3);
                pe = keep32(key, rbase + kk, a.th_a) ? pe * a.inv_keep_a : 0.f;
            }
            pf[e] = f2bf(pe);
        }
        l = l * alpha + group_sum(ls);
        mx = mnew;
#pragma unroll
        for (int r = 0; r < 4; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        if (SELF) {      // (the upper four slots carry p = 0 against the same, finite, rows)
            o0 = mfma(frag_tr(kvimg + 512 + h * 32, PQ, 0, 0, 0, li, g), pf, o0);
            o1 = mfma(frag_tr(kvimg + 512 + h * 32, PQ, 16, 0, 0, li, g), pf, o1);
        } else {
            o0 = mfma(frag_tr(vtile, VP, 0, 0, 16, li, g), pf, o0);
            o1 = mfma(frag_tr(vtile, VP, 16, 0, 16, li, g), pf, o1);
            __builtin_amdgcn_wave_barrier();
        }
    }
    const float inv = l > 0.f ? 1.f / l : 0.f;
    bf16x4 y0, y1;
#pragma unroll
    for (int r = 0; r < 4; ++r) { y0[r] = f2bf(o0[r] * inv); y1[r] = f2bf(o1[r] * inv); }
    bf16* orow = out + (long)li * PA + h * 32;
    *(uint2*)(orow + g * 4) = as_uint2(y0);
    *(uint2*)(orow + 16 + g * 4) = as_uint2(y1);
}

template <bool DROP>
__global__ __launch_bounds__(NT) void decoder_fwd_kernel(DecArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* X32 = (float*)smem_raw;                                  // [R][256] residual stream
    bf16* QP = (bf16*)(X32 + R * E);                                // [16][256] query_pos
    bf16* A0 = QP + 16 * E;                                         // [R][PA]   operand with query_pos
    bf16* A1 = A0 + R * PA;                                         // [R][PA]   operand without / attention output
    bf16* BIG = A1 + R * PA;                                        // union: QKV [R][PQ] | { QC [R][PA], V tiles [NW][32][VP] } | HID [R][PH]
    bf16* QKV = BIG;
    bf16* QC = BIG;
    bf16* VT = BIG + R * PA;                                        // [NW][32][VP]
    bf16* HID = BIG;
    float* PRM = (float*)(BIG + R * PH);                            // this layer's biases and LayerNorm parameters (fp32, layout below)
    // PRM: b_in 768 | b_o1 256 | g1 256 | e1 256 | b_q 256 | b_o2 256 | g2 256 | e2 256 | b_f1 2048 | b_f2 256 | g3 256 | e3 256
    constexpr int P_BIN = 0, P_BO1 = 768, P_G1 = 1024, P_E1 = 1280, P_BQ = 1536, P_BO2 = 1792, P_G2 = 2048, P_E2 = 2304, P_BF1 = 2560,
                  P_BF2 = 4608, P_G3 = 4864, P_E3 = 5120, P_TOTAL = 5376;
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = tid & 63, li = lane & 15, g = lane >> 4;
    const unsigned offE = (unsigned)(li * E + g * 8) * 2u, offF = (unsigned)(li * FF + g * 8) * 2u;      // lane offsets inside a weight unit
    const int b0 = blockIdx.x * RT;                                 // first clip of this workgroup
    const int Q = a.Q;

    // ---- start: tgt = 0 (transformer.py:60), query_pos rows (zero beyond Q), operand images ----
    for (int i = tid; i < 16 * E; i += NT) QP[i] = (i / E) < Q ? a.qpos[i] : f2bf(0.f);
    for (int i = tid; i < R * E; i += NT) X32[i] = 0.f;
    __syncthreads();

    // (x + query_pos | x) as bf16 operand images from the fp32 rows
    auto build_operands = [&]() {
        for (int i = tid; i < R * (E / 4); i += NT) {
            const int row = i / (E / 4), c4 = (i % (E / 4)) * 4;
            const float4 x = *(const float4*)(X32 + row * E + c4);
            const bf16x4 q = as_bf16x4(*(const uint2*)(QP + (row & 15) * E + c4));
            bf16x4 u, v;
            u[0] = f2bf(x.x + bf2f(q[0])); u[1] = f2bf(x.y + bf2f(q[1])); u[2] = f2bf(x.z + bf2f(q[2])); u[3] = f2bf(x.w + bf2f(q[3]));
            v[0] = f2bf(x.x); v[1] = f2bf(x.y); v[2] = f2bf(x.z); v[3] = f2bf(x.w);
            *(uint2*)(A0 + row * PA + c4) = as_uint2(u);
            *(uint2*)(A1 + row * PA + c4) = as_uint2(v);
        }
    };
    // X32[row][f] += Dropout(acc + bias[f])  for this wave's column tile (the LayerNorm pass that follows reads the whole row)
    auto residual_epilogue = [&](const f32x4 (&acc)[RT], const float* bias, int ct, uint32_t key) {
        const float4 bv = *(const float4*)(bias + ct * 16 + g * 4);
        const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            float* xr = X32 + (rt * 16 + li) * E + ct * 16 + g * 4;
            float4 x = *(float4*)xr;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = acc[rt][r] + bb[r];
                if (DROP) v[r] = keep32(key, (uint32_t)(((b0 + rt) * 16 + li) * E + ct * 16 + g * 4 + r), a.th) ? v[r] * a.inv_keep : 0.f;
            }
            x.x += v[0]; x.y += v[1]; x.z += v[2]; x.w += v[3];
            *(float4*)xr = x;
        }
    };
    // in-place LayerNorm of the fp32 rows (one wave per row, 4 features per lane)
    auto layer_norm = [&](const float* gamma, const float* beta) {
        const float4 gv = *(const float4*)(gamma + lane * 4), bv = *(const float4*)(beta + lane * 4);
        for (int row = wave; row < R; row += NW) {
            float4 x = *(float4*)(X32 + row * E + lane * 4);
            const float mean = wave_sum(x.x + x.y + x.z + x.w) * (1.f / E);
            const float d0 = x.x - mean, d1 = x.y - mean, d2 = x.z - mean, d3 = x.w - mean;
            const float rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.f / E) + LN_EPS);
            x.x = fmaf(d0 * rstd, gv.x, bv.x); x.y = fmaf(d1 * rstd, gv.y, bv.y); x.z = fmaf(d2 * rstd, gv.z, bv.z); x.w = fmaf(d3 * rstd, gv.w, bv.w);
            *(float4*)(X32 + row * E + lane * 4) = x;
        }
    };

#ifdef DEC_TIMING
    unsigned long long* tstamp = (unsigned long long*)a.hs32;      // phase timestamps of layer 0 .. (timing builds only: hs32 is not written)
    int tsi = 0;
#define TS() do { if (tid == 0 && blockIdx.x == 0) tstamp[tsi++] = __builtin_readcyclecounter(); } while (0)
#else
#define TS() do { } while (0)
#endif
    build_operands();
    // the first four units of layer 0 go out before anything is computed
    Ring ring;
    ring_issue<0>(ring, unit_base(a.L[0], 0, wave), offE);
    ring_issue<1>(ring, unit_base(a.L[0], 1, wave), offE);
    ring_issue<2>(ring, unit_base(a.L[0], 2, wave), offE);
    ring_issue<3>(ring, unit_base(a.L[0], 3, wave), offE);
    __syncthreads();

#pragma unroll 1
    for (int l = 0; l < a.nl; ++l) {
        const LayerW& W = a.L[l];
        const LayerW& Wn = a.L[min(l + 1, a.nl - 1)];
        const bool more = l + 1 < a.nl;
        // biases / LayerNorm parameters of the layer -> LDS (one drain of the load queue per layer instead of one per bias vector:
        // any vector load issued behind the ring's loads has to wait for all of them)
        {
            const float* src[12] = {W.b_in, W.b_o1, W.g1, W.e1, W.b_q, W.b_o2, W.g2, W.e2, W.b_f1, W.b_f2, W.g3, W.e3};
            const int beg[13] = {P_BIN, P_BO1, P_G1, P_E1, P_BQ, P_BO2, P_G2, P_E2, P_BF1, P_BF2, P_G3, P_E3, P_TOTAL};
#pragma unroll
            for (int s_ = 0; s_ < 12; ++s_)
                for (int i = tid * 4; i < beg[s_ + 1] - beg[s_]; i += NT * 4) *(float4*)(PRM + beg[s_] + i) = *(const float4*)(src[s_] + i);
        }
        __syncthreads();
        // refill the ring slot of unit U (just consumed) with unit U + RING: of this layer, or of the next one past the end
#define REFILL(U)                                                                                        \
        do {                                                                                             \
            constexpr int nu__ = (U) + RING;                                                             \
            if (nu__ < UNITS) ring_issue<(U) % RING>(ring, unit_base(W, nu__, wave), unit_is_f2(nu__) ? offF : offE);              \
            else if (more) ring_issue<(U) % RING>(ring, unit_base(Wn, nu__ - UNITS, wave), offE);        \
        } while (0)
        // ---- 1. self-attention in-projection: q | k see x + query_pos, v sees x (48 column tiles: units 0..5 of every wave) ----
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const int ct = wave + NW * u;
            f32x4 acc[RT] = {};
            if (u == 0) ring_consume<0>(ring, ct < 32 ? A0 : A1, PA, 0, li, g, acc);
            if (u == 1) ring_consume<1>(ring, ct < 32 ? A0 : A1, PA, 0, li, g, acc);
            if (u == 2) ring_consume<2>(ring, ct < 32 ? A0 : A1, PA, 0, li, g, acc);
            if (u == 3) ring_consume<3>(ring, ct < 32 ? A0 : A1, PA, 0, li, g, acc);
            if (u == 4) ring_consume<0>(ring, ct < 32 ? A0 : A1, PA, 0, li, g, acc);
            if (u == 5) ring_consume<1>(ring, ct < 32 ? A0 : A1, PA, 0, li, g, acc);
            if (u == 0) REFILL(0);
            if (u == 1) REFILL(1);
            if (u == 2) REFILL(2);
            if (u == 3) REFILL(3);
            if (u == 4) REFILL(4);
            if (u == 5) REFILL(5);
            const float4 bv = *(const float4*)(PRM + P_BIN + ct * 16 + g * 4);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                bf16x4 o;
                o[0] = f2bf(acc[rt][0] + bv.x); o[1] = f2bf(acc[rt][1] + bv.y); o[2] = f2bf(acc[rt][2] + bv.z); o[3] = f2bf(acc[rt][3] + bv.w);
                *(uint2*)(QKV + (rt * 16 + li) * PQ + ct * 16 + g * 4) = as_uint2(o);
            }
        }
        __syncthreads();
        TS();
        // ---- 2. self-attention, two (clip, head) pairs per wave; output into A1 ----
#pragma unroll 1
        for (int pr = wave; pr < RT * NH; pr += NW) {
            const int c = pr >> 3, h = pr & 7;
            attention_head<true, DROP>(a, QKV + c * 16 * PQ, PQ, QKV + c * 16 * PQ, nullptr, 0, nullptr, Q, nullptr, A1 + c * 16 * PA, h, lane,
                                 DROP ? site_key(a, l, 0) : 0u, 16u, b0 + c);
        }
        __syncthreads();
        TS();
        // ---- 3. out-projection + residual (units 6, 7), 4. norm1 ----
        {
            f32x4 acc[RT] = {};
            ring_consume<2>(ring, A1, PA, 0, li, g, acc);
            REFILL(6);
            residual_epilogue(acc, PRM + P_BO1, wave, (DROP ? site_key(a, l, 1) : 0u));
            f32x4 acc1[RT] = {};
            ring_consume<3>(ring, A1, PA, 0, li, g, acc1);
            REFILL(7);
            residual_epilogue(acc1, PRM + P_BO1, wave + NW, (DROP ? site_key(a, l, 1) : 0u));
        }
        __syncthreads();
        TS();
        layer_norm(PRM + P_G1, PRM + P_E1);
        __syncthreads();
        TS();
        build_operands();
        __syncthreads();
        TS();
        // ---- 5. cross-attention query projection (tgt + query_pos) W_q -> QC (units 8, 9) ----
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int ct = wave + NW * t;
            f32x4 acc[RT] = {};
            if (t == 0) { ring_consume<0>(ring, A0, PA, 0, li, g, acc); REFILL(8); }
            else { ring_consume<1>(ring, A0, PA, 0, li, g, acc); REFILL(9); }
            const float4 bv = *(const float4*)(PRM + P_BQ + ct * 16 + g * 4);
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                bf16x4 o;
                o[0] = f2bf(acc[rt][0] + bv.x); o[1] = f2bf(acc[rt][1] + bv.y); o[2] = f2bf(acc[rt][2] + bv.z); o[3] = f2bf(acc[rt][3] + bv.w);
                *(uint2*)(QC + (rt * 16 + li) * PA + ct * 16 + g * 4) = as_uint2(o);
            }
        }
        __syncthreads();
        TS();
        // ---- 6. cross-attention over the encoder memory of the clip -> A1 ----
#pragma unroll 1
        for (int pr = wave; pr < RT * NH; pr += NW) {
            const int c = pr >> 3, h = pr & 7;
            const int bg = min(b0 + c, a.B - 1);
            attention_head<false, DROP>(a, QC + c * 16 * PA, PA, nullptr, W.kv + (long)bg * a.Lm * a.ldkv, a.ldkv, a.kpm ? a.kpm + (long)bg * a.Lm : nullptr,
                                  a.Lm, VT + wave * 32 * VP, A1 + c * 16 * PA, h, lane, DROP ? site_key(a, l, 2) : 0u, (uint32_t)a.Lm, b0 + c);
        }
        __syncthreads();
        TS();
        // ---- 7. out-projection + residual (units 10, 11), norm2 ----
        {
            f32x4 acc[RT] = {};
            ring_consume<2>(ring, A1, PA, 0, li, g, acc);
            REFILL(10);
            residual_epilogue(acc, PRM + P_BO2, wave, (DROP ? site_key(a, l, 3) : 0u));
            f32x4 acc1[RT] = {};
            ring_consume<3>(ring, A1, PA, 0, li, g, acc1);
            REFILL(11);
            residual_epilogue(acc1, PRM + P_BO2, wave + NW, (DROP ? site_key(a, l, 3) : 0u));
        }
        __syncthreads();
        TS();
        layer_norm(PRM + P_G2, PRM + P_E2);
        __syncthreads();
        TS();
        build_operands();               // A1 = bf16(x): the FFN input
        __syncthreads();
        TS();
        // ---- 8. FFN in two 1024-wide halves of the hidden layer: linear1 + ReLU (+ Dropout) -> HID (8 units), linear2 accumulated in registers
        // (2 column tiles x 4 k-chunks = 8 units) ----
        {
            f32x4 acc2[2][RT] = {};
            const uint32_t key4 = DROP ? site_key(a, l, 4) : 0u;
#define FFN_HALF(HALF, U0)                                                                                                      \
            {                                                                                                                   \
                _Pragma("unroll") for (int v = 0; v < 8; ++v) {                                                                 \
                    const int ct = wave + NW * v;                                                                               \
                    f32x4 acc[RT] = {};                                                                                         \
                    if (v % 4 == 0) ring_consume<((U0) + 0) % RING>(ring, A1, PA, 0, li, g, acc);                                \
                    if (v % 4 == 1) ring_consume<((U0) + 1) % RING>(ring, A1, PA, 0, li, g, acc);                                \
                    if (v % 4 == 2) ring_consume<((U0) + 2) % RING>(ring, A1, PA, 0, li, g, acc);                                \
                    if (v % 4 == 3) ring_consume<((U0) + 3) % RING>(ring, A1, PA, 0, li, g, acc);                                \
                    if (v == 0) REFILL((U0) + 0); if (v == 1) REFILL((U0) + 1); if (v == 2) REFILL((U0) + 2); if (v == 3) REFILL((U0) + 3); \
                    if (v == 4) REFILL((U0) + 4); if (v == 5) REFILL((U0) + 5); if (v == 6) REFILL((U0) + 6); if (v == 7) REFILL((U0) + 7); \
                    const float4 bv = *(const float4*)(PRM + P_BF1 + (HALF) * 1024 + ct * 16 + g * 4);                                \
                    const float bb[4] = {bv.x, bv.y, bv.z, bv.w};                                                               \
                    _Pragma("unroll") for (int rt = 0; rt < RT; ++rt) {                                                         \
                        bf16x4 o;                                                                                               \
                        _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                                         \
                            float x = fmaxf(acc[rt][r] + bb[r], 0.f);                                                           \
                            if (DROP) x = keep32(key4, (uint32_t)(((b0 + rt) * 16 + li) * FF + (HALF) * 1024 + ct * 16 + g * 4 + r), a.th) ? x * a.inv_keep : 0.f; \
                            o[r] = f2bf(x);                                                                                     \
                        }                                                                                                       \
                        *(uint2*)(HID + (rt * 16 + li) * PH + ct * 16 + g * 4) = as_uint2(o);                                   \
                    }                                                                                                           \
                }                                                                                                               \
                __syncthreads(); TS();                                                                                          \
                _Pragma("unroll") for (int v = 0; v < 8; ++v) {                                                                 \
                    if (v % 4 == 0) ring_consume<((U0) + 8) % RING>(ring, HID, PH, (v & 3) * 256, li, g, acc2[v >> 2]);          \
                    if (v % 4 == 1) ring_consume<((U0) + 9) % RING>(ring, HID, PH, (v & 3) * 256, li, g, acc2[v >> 2]);          \
                    if (v % 4 == 2) ring_consume<((U0) + 10) % RING>(ring, HID, PH, (v & 3) * 256, li, g, acc2[v >> 2]);         \
                    if (v % 4 == 3) ring_consume<((U0) + 11) % RING>(ring, HID, PH, (v & 3) * 256, li, g, acc2[v >> 2]);         \
                    if (v == 0) REFILL((U0) + 8); if (v == 1) REFILL((U0) + 9); if (v == 2) REFILL((U0) + 10); if (v == 3) REFILL((U0) + 11); \
                    if (v == 4) REFILL((U0) + 12); if (v == 5) REFILL((U0) + 13); if (v == 6) REFILL((U0) + 14); if (v == 7) REFILL((U0) + 15); \
                }                                                                                                               \
                __syncthreads(); TS();                                                                                          \
            }
            FFN_HALF(0, 12)
            FFN_HALF(1, 28)
#undef FFN_HALF
            residual_epilogue(acc2[0], PRM + P_BF2, wave, (DROP ? site_key(a, l, 5) : 0u));
            residual_epilogue(acc2[1], PRM + P_BF2, wave + NW, (DROP ? site_key(a, l, 5) : 0u));
        }
#undef REFILL
        __syncthreads();
        TS();
        layer_norm(PRM + P_G3, PRM + P_E3);
        __syncthreads();
        TS();
        // ---- 9. decoder.norm of this layer's output -> hs[l] (the residual stream itself stays un-normalised, transformer.py:116-126) ----
        {
            const float4 gv = *(const float4*)(a.gN + lane * 4), bv = *(const float4*)(a.eN + lane * 4);
            for (int row = wave; row < R; row += NW) {
                const int c = row >> 4, q = row & 15;
                if (b0 + c >= a.B || q >= Q) continue;
                const float4 x = *(const float4*)(X32 + row * E + lane * 4);
                const float mean = wave_sum(x.x + x.y + x.z + x.w) * (1.f / E);
                const float d0 = x.x - mean, d1 = x.y - mean, d2 = x.z - mean, d3 = x.w - mean;
                const float rstd = rsqrtf(wave_sum(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3) * (1.f / E) + LN_EPS);
                const float y0 = fmaf(d0 * rstd, gv.x, bv.x), y1 = fmaf(d1 * rstd, gv.y, bv.y), y2 = fmaf(d2 * rstd, gv.z, bv.z), y3 = fmaf(d3 * rstd, gv.w, bv.w);
                const long orow = (((long)l * a.B + b0 + c) * Q + q) * E + lane * 4;
                bf16x4 o;
                o[0] = f2bf(y0); o[1] = f2bf(y1); o[2] = f2bf(y2); o[3] = f2bf(y3);
                *(uint2*)(a.hs + orow) = as_uint2(o);
#ifndef DEC_TIMING
                if (a.hs32) *(float4*)(a.hs32 + orow) = make_float4(y0, y1, y2, y3);
#endif
            }
        }
        build_operands();               // next layer's in-projection operands (X32 is not modified by the pass above)
        __syncthreads();
        TS();
    }
}

constexpr size_t kLds = (size_t)R * E * 4 + 16 * E * 2 + 2 * (size_t)R * PA * 2 + (size_t)R * PH * 2 + 5376 * 4;

}  // namespace

extern "C" {

// 1 when tuber_decoder_fwd takes this decoder: 256-wide model, 8 heads, 2048-wide FFN, at most 16 queries per clip and 6 layers
int tuber_decoder_fused_supported(int d_model, int nhead, int dim_ff, int num_queries, int num_layers) {
    return d_model == E && nhead == NH && dim_ff == FF && num_queries >= 1 && num_queries <= 16 && num_layers >= 1 && num_layers <= MAXL;
}

// The whole post-norm decoder stack in one launch (eval / inference form: no activations are saved).
//   layer_ptrs: HOST array of 19 device pointers per layer, in the order of LayerW (weights bf16 row-major, biases / LayerNorm fp32, then the
//               layer's packed encoder-memory projection [B*Lm][ldkv] = [(memory + pos) W_k | memory W_v]);
//   qpos: query_embed.weight as bf16 [Q][256]; kpm: [B][Lm] key-padding mask or NULL; hs: bf16 [nl][B][Q][256]; hs32: the same in fp32 or NULL;
//   pdrop / pattn: Dropout of the sublayer outputs / of the attention weights (0 in eval), masks from (seed, salt, layer, site, index).
int tuber_decoder_fwd(const void* const* layer_ptrs, int num_layers, const float* norm_weight, const float* norm_bias, const void* qpos,
                      long ldkv, const void* kpm, int B, int Q, int Lm, void* hs, float* hs32, float pdrop, float pattn,
                      const void* seed_ptr, unsigned long long salt, hipStream_t stream) {
    if (!layer_ptrs || num_layers < 1 || num_layers > MAXL || Q < 1 || Q > 16 || B < 1 || Lm < 1 || (ldkv & 7) || ldkv < 2 * E || !hs
        || pdrop < 0.f || pdrop >= 1.f || pattn < 0.f || pattn >= 1.f) return TUBER_EINVAL;
    DecArgs a{};
    for (int l = 0; l < num_layers; ++l) {
        const void* const* p = layer_ptrs + l * 19;
        LayerW& w = a.L[l];
        w.w_in = (const bf16*)p[0]; w.b_in = (const float*)p[1]; w.w_o1 = (const bf16*)p[2]; w.b_o1 = (const float*)p[3];
        w.g1 = (const float*)p[4]; w.e1 = (const float*)p[5]; w.w_q = (const bf16*)p[6]; w.b_q = (const float*)p[7];
        w.w_o2 = (const bf16*)p[8]; w.b_o2 = (const float*)p[9]; w.g2 = (const float*)p[10]; w.e2 = (const float*)p[11];
        w.w_f1 = (const bf16*)p[12]; w.b_f1 = (const float*)p[13]; w.w_f2 = (const bf16*)p[14]; w.b_f2 = (const float*)p[15];
        w.g3 = (const float*)p[16]; w.e3 = (const float*)p[17]; w.kv = (const bf16*)p[18];
        for (int i = 0; i < 19; ++i) if (!p[i]) return TUBER_EINVAL;
    }
    a.nl = num_layers; a.gN = norm_weight; a.eN = norm_bias; a.qpos = (const bf16*)qpos; a.ldkv = ldkv; a.kpm = (const uint8_t*)kpm;
    a.B = B; a.Q = Q; a.Lm = Lm; a.hs = (bf16*)hs; a.hs32 = hs32;
    a.th = (uint32_t)((double)pdrop * 4294967296.0); a.th_a = (uint32_t)((double)pattn * 4294967296.0);
    a.inv_keep = 1.f / (1.f - pdrop); a.inv_keep_a = 1.f / (1.f - pattn);
    a.seed_ptr = (const uint64_t*)seed_ptr; a.salt = salt;
    static LdsOptIn opt[2];
    if (a.th || a.th_a) {
        TUBER_LDS_OPT_IN(opt[1], decoder_fwd_kernel<true>, kLds);
        hipLaunchKernelGGL(decoder_fwd_kernel<true>, dim3((B + RT - 1) / RT), dim3(NT), kLds, stream, a);
    } else {
        TUBER_LDS_OPT_IN(opt[0], decoder_fwd_kernel<false>, kLds);
        hipLaunchKernelGGL(decoder_fwd_kernel<false>, dim3((B + RT - 1) / RT), dim3(NT), kLds, stream, a);
    }
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
