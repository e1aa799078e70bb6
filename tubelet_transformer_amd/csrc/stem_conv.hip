// CSN stem Conv3d(3,64,k=(3,7,7),s=(1,2,2),p=(1,3,3)) as an IMPLICIT GEMM on MFMA (gfx950) -- forward and weight gradient.
// reference: models/backbones/ir_CSN_152.py:109-115,172-174.  No patch matrix ever reaches HBM (the explicit 1.25 GB
// im2col of the first round cost ~7 GB/step of traffic, profiles/r01_d_pmc_hbm_traffic_per_kernel.txt).
//
// Work unit ("tile"): 64 consecutive output columns wo of one (clip, t, ho) output row, all 64 channels.  Its receptive
// field is 63 "runs" (c, kt, kh) of 2*64+5 = 133 consecutive input pixels; the fp32 NCDHW clip is read run-wise
// (coalesced along w), converted to bf16 and staged in LDS as P[run][x] (17 KB, double buffered).
// Patch loader: a run is 34 quads of 4 pixels.  One load instruction = two runs x quads 0..31 (half a wave each, one dwordx4
// per lane); quads 32 and 33 of all runs are one extra instruction in waves 0 and 1 (lane = run).  Everything about a run (row
// offset, temporal / vertical padding) is wave-uniform and lives in SGPRs; the horizontal border is handled without branches:
// the quad is loaded from the column clamped into the row and the four bf16 are shifted (zero fill) by the clamp distance.
// (The first version computed (run, column) per element in VALU -- ~100 instructions per load; the kernels were bound by that,
// not by memory: scripts/gemm_bench.py stem)
// K is laid out as k' = run*8 + kw (kw = 7 and run = 63 are zero padding -> K' = 512), so that an MFMA operand fragment
// (8 consecutive k') of output column w is the 8 consecutive pixels P[run][2w .. 2w+7] = 4 dwords from dword w of the row.
//   forward : wave = 16 output channels; its 16 weight fragments (all of K') stay in 64 VGPRs for the whole persistent
//             kernel; per k-step 4 patch fragments (one per 16-column m-tile) + 4 MFMAs.  BN partial statistics fused.
//   dW      : D[k'][n] += sum_m P^T[k'][m] G[m][n]; wave = 8 k'-tiles x 4 n-tiles (128 accumulator VGPRs, kept across
//             the persistent loop); G tile transposed into LDS with 4x4 register transposes; patch^T fragments are
//             8 strided 2-byte LDS reads.  fp32 partials per workgroup, reduced + scattered to [64][441] afterwards.
#include "common.h"

#define SW 64
#define PXW 160         // row pitch 80 dwords = 16 mod 32: the four run rows of a fragment read (ds_read_b32: 32 banks, 32-lane groups) fall on disjoint banks (136 pixels used)
#define KP 512

struct StemGeom { int B, T, H, W, Ho, Wo, tilesW, ntiles; };

// Tile order: column tile fastest, then t, then ho, then clip -- consecutive tiles are the same output row of consecutive
// frames, which share their 7 input rows across the 3-frame temporal window; stepping ho re-uses 5 of the 7 rows.
struct StemTile { int wt, t, ho, b; long row0; };       // row0 = first output position (of [B*T*Ho*Wo]) of the tile
__device__ __forceinline__ StemTile stem_tile(const StemGeom& g, int tile) {
    StemTile s;
    s.wt = tile % g.tilesW; int r = tile / g.tilesW;
    s.t = r % g.T; r /= g.T;
    s.ho = r % g.Ho; s.b = r / g.Ho;
    s.row0 = (((long)s.b * g.T + s.t) * g.Ho + s.ho) * g.Wo + s.wt * SW;
    return s;
}
// Persistent walk, XCD-aware: workgroups are dealt round-robin to the 8 XCDs, whose L2s do not share lines.  Each XCD gets one
// contiguous eighth of the tile sequence (a band of output rows, all frames), so that the input halo is fetched by ONE L2 instead
// of all eight (the plain grid-stride walk read the clip 8x: profiles/r02_n_pmc_hbm_traffic_per_kernel.txt, 976 MB/launch).
struct StemWalk { int tile, end, step; };
__device__ __forceinline__ StemWalk stem_walk(int ntiles) {
    const int G = gridDim.x;
    if (G & 7) return StemWalk{(int)blockIdx.x, ntiles, G};
    const int chunk = (ntiles + 7) >> 3, x = blockIdx.x & 7, lo = x * chunk;
    return StemWalk{lo + (int)(blockIdx.x >> 3), min(ntiles, lo + chunk), G >> 3};
}
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));      // dword-aligned vector loads (the run start is 128*wt - 3)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));

struct StemLane { int l32, half, toff, tkt, tkh; };     // tile-invariant per-lane constants (tail: lane = run)
__device__ __forceinline__ StemLane stem_lane(const StemGeom& g) {
    StemLane L;
    const int lane = threadIdx.x & 63;
    L.l32 = lane & 31; L.half = lane >> 5;
    const int c = lane / 21, kt = (lane / 7) % 3, kh = lane % 7;
    L.tkt = kt - 1; L.tkh = kh - 3;
    L.toff = ((c * g.T + kt - 1) * g.H + kh - 3) * g.W;
    return L;
}
// what stem_fetch leaves for stem_park_*: per-lane zero-fill shifts (bits) of the main / tail quads and the kill mask
// (bit p = pair p's quad is all padding, bit 31 = the tail quad is)
struct StemMeta { uint32_t kill; int sr, sl, tsr, tsl; };
__device__ __forceinline__ void stem_shift_of(int x0, int W, int& xc, int& sr, int& sl, bool& dead) {
    xc = min(max(x0, 0), W - 4);
    const int d = x0 - xc;                      // > 0: right border, < 0: left border
    dead = d >= 4 || d <= -4;
    sr = d > 0 ? 16 * min(d, 3) : 0;
    sl = d < 0 ? 16 * min(-d, 3) : 0;
}
// NW waves per workgroup; wave wv owns runs wv*64/NW ... (PAIRS = 32/NW pairs).  Their tile-invariant constants (SGPRs):
template <int NW> struct StemRuns { int off[64 / NW], dt[64 / NW], dh[64 / NW]; };
template <int NW>
__device__ __forceinline__ StemRuns<NW> stem_runs(const StemGeom& g, int wv) {
    StemRuns<NW> R;
#pragma unroll
    for (int i = 0; i < 64 / NW; ++i) {
        const int r = wv * (64 / NW) + i;
        const int c = r / 21, kt = (r / 7) % 3, kh = r % 7;
        R.dt[i] = r < 63 ? kt - 1 : (1 << 24);              // run 63 is the zero padding of K': never valid
        R.dh[i] = kh - 3;
        R.off[i] = ((c * g.T + kt - 1) * g.H + kh - 3) * g.W;
    }
    return R;
}
template <int NW>
__device__ __forceinline__ void stem_fetch(const float* __restrict__ clip, const StemGeom& g, int tile, const StemLane& L,
                                           const StemRuns<NW>& R, int wv, float4 (&v)[32 / NW + 1], StemMeta& m) {
    constexpr int PAIRS = 32 / NW;
    const StemTile st = stem_tile(g, tile);
    const float* base = clip + (((long)st.b * 3 * g.T + st.t) * g.H + 2 * st.ho) * (long)g.W;   // (c = 0, t, 2 ho) row: always inside the clip
    int xc; bool dead;
    stem_shift_of(128 * st.wt - 3 + 4 * L.l32, g.W, xc, m.sr, m.sl, dead);
    m.kill = dead ? 0x7fffffffu : 0u;
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
        int off[2]; bool ok[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {                                  // scalar: the run is wave-uniform
            ok[h] = (unsigned)(st.t + R.dt[2 * p + h]) < (unsigned)g.T && (unsigned)(2 * st.ho + R.dh[2 * p + h]) < (unsigned)g.H;
            off[h] = ok[h] ? R.off[2 * p + h] : 0;
        }
        const f32x4u q = *(const f32x4u*)(base + (L.half ? off[1] : off[0]) + xc);
        v[p] = make_float4(q.x, q.y, q.z, q.w);
        if (!(L.half ? ok[1] : ok[0])) m.kill |= 1u << p;
    }
    m.tsr = m.tsl = 0;
    if (wv < 2) {                                                       // tail quads 32 (wave 0) and 33 (wave 1): lane = run
        const int lane = threadIdx.x & 63;
        int txc; bool tdead;
        stem_shift_of(128 * st.wt - 3 + 4 * (32 + wv), g.W, txc, m.tsr, m.tsl, tdead);
        const bool ok = lane < 63 && (unsigned)(st.t + L.tkt) < (unsigned)g.T && (unsigned)(2 * st.ho + L.tkh) < (unsigned)g.H;
        const f32x4u q = *(const f32x4u*)(base + (ok ? L.toff : 0) + txc);
        v[PAIRS] = make_float4(q.x, q.y, q.z, q.w);
        if (!ok || tdead) m.kill |= 1u << 31;
    }
}
// four pixels -> 4 bf16 in a 64-bit word, shifted into place with zero fill
__device__ __forceinline__ uint64_t stem_pack4(const float4& f, int sr, int sl, bool kill) {
    const bf16x4 b = {f2bf(f.x), f2bf(f.y), f2bf(f.z), f2bf(f.w)};
    const uint2 u = as_uint2(b);
    uint64_t w = ((uint64_t)u.y << 32) | u.x;
    w = (w >> sr) << sl;
    return kill ? 0ull : w;
}
template <int NW>
__device__ __forceinline__ void stem_park(bf16 (*P)[PXW], const float4 (&v)[32 / NW + 1], const StemMeta& m, const StemLane& L, int wv) {
    constexpr int PAIRS = 32 / NW;
#pragma unroll
    for (int p = 0; p < PAIRS; ++p)
        *(uint64_t*)&P[wv * (2 * PAIRS) + 2 * p + L.half][4 * L.l32] = stem_pack4(v[p], m.sr, m.sl, (m.kill >> p) & 1);
    if (wv < 2) *(uint64_t*)&P[threadIdx.x & 63][4 * (32 + wv)] = stem_pack4(v[PAIRS], m.tsr, m.tsl, m.kill >> 31);
}
__global__ __launch_bounds__(256, 3) void stem_conv_fwd_kernel(const float* __restrict__ clip, const bf16* __restrict__ Wp,
                                                            bf16* __restrict__ out, float* __restrict__ st0,
                                                            float* __restrict__ st1, StemGeom g) {
    __shared__ __attribute__((aligned(16))) bf16 P[2][64][PXW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, gq = lane >> 4;
    // this wave's 16 channels: weight fragments for all 16 k-steps (A operand: row i = channel, 8 consecutive k')
    bf16x8 wf[16];
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) wf[ks] = as_bf16x8(*(const uint4*)(Wp + (long)(wave * 16 + li) * KP + ks * 32 + gq * 8));
    // pin the weights in VGPRs HERE: otherwise their (one-time) load is waited for inside the k-loop, with vmcnt counts that also
    // drain the patch prefetch of the next tile before the MFMAs are through
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) asm volatile("" : "+v"(wf[ks]));
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    float4 pre[9];
    StemMeta meta;
    const StemLane L = stem_lane(g);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const StemRuns<4> R = stem_runs<4>(g, wv);
    const StemWalk wk = stem_walk(g.ntiles);
    int tile = wk.tile, buf = 0;
    if (tile < wk.end) {
        stem_fetch<4>(clip, g, tile, L, R, wv, pre, meta);
        stem_park<4>(P[0], pre, meta, L, wv);
    }
    __syncthreads();
    for (; tile < wk.end; tile += wk.step) {
        const int next = tile + wk.step;
        if (next < wk.end) stem_fetch<4>(clip, g, next, L, R, wv, pre, meta);     // in flight during the MFMAs below
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            const int run = ks * 4 + gq;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const uint32_t* p = (const uint32_t*)&P[buf][run][2 * (mt * 16 + li)];         // P[run][2w .. 2w+7], w = mt*16 + li
                acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ks], as_bf16x8(make_uint4(p[0], p[1], p[2], p[3])), acc[mt], 0, 0, 0);
            }
        }
        // park the next patch BEFORE the output stores: vmcnt counts loads and stores in order, so waiting for a load that was issued
        // before a store never waits for the store's acknowledgement, the other order does
        if (next < wk.end) stem_park<4>(P[buf ^ 1], pre, meta, L, wv);
        // D[i = channel][j = column]: lane holds column mt*16 + li, channels wave*16 + gq*4 + 0..3
        const StemTile st = stem_tile(g, tile);
        const int wt = st.wt;
        const long row0 = st.row0;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int w = wt * SW + mt * 16 + li;
            if (w < g.Wo) {
                bf16x4 o;
#pragma unroll
                for (int r = 0; r < 4; ++r) { o[r] = f2bf(acc[mt][r]); s0[r] += acc[mt][r]; s1[r] += acc[mt][r] * acc[mt][r]; }
                *(uint2*)(out + (row0 + mt * 16 + li) * 64 + wave * 16 + gq * 4) = as_uint2(o);
            }
        }
        __syncthreads();
        buf ^= 1;
    }
    if (st0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float a = quad16_sum(s0[r]), b2 = quad16_sum(s1[r]);
            if (li == 0) {
                st0[(long)blockIdx.x * 64 + wave * 16 + gq * 4 + r] = a;
                st1[(long)blockIdx.x * 64 + wave * 16 + gq * 4 + r] = b2;
            }
        }
    }
}

// ---- weight gradient ----
// D[k'][n] += sum_m patch^T[k'][m] G[m][n].  The MFMA A fragment needs 8 consecutive m of one k' = (run, kw), i.e. the pixels
// P[run][2m + kw]: stride 2.  The patch is therefore staged DE-INTERLEAVED (PE = even, PO = odd pixel columns), which makes
// the fragment 8 consecutive bf16 of PE/PO[run] starting at m + kw/2; and the 16 rows of an MFMA are 16 RUNS with one common
// kw, so the sub-dword start offset is wave-uniform: one 16-byte + one 8-byte LDS read and (for odd offsets) 4 v_alignbit.
#define EOW 72
#define WG_THREADS 512   // weight-gradient workgroup: 8 waves
// weight-gradient staging of the same quads: de-interleaved into even / odd pixel columns (see below)
__device__ __forceinline__ void stem_park_eo(bf16 (*PE)[EOW], bf16 (*PO)[EOW], const float4 (&v)[5], const StemMeta& m, const StemLane& L, int wv) {
    auto put = [&](int r, int q, uint64_t w) {
        const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);                 // lo = (p0, p1), hi = (p2, p3)
        *(uint32_t*)&PE[r][2 * q] = __builtin_amdgcn_perm(hi, lo, 0x05040100);    // (p0, p2): pixels 4q, 4q+2 -> even columns 2q, 2q+1
        *(uint32_t*)&PO[r][2 * q] = __builtin_amdgcn_perm(hi, lo, 0x07060302);    // (p1, p3)
    };
#pragma unroll
    for (int p = 0; p < 4; ++p) put(wv * 8 + 2 * p + L.half, L.l32, stem_pack4(v[p], m.sr, m.sl, (m.kill >> p) & 1));
    if (wv < 2) put(threadIdx.x & 63, 32 + wv, stem_pack4(v[4], m.tsr, m.tsl, m.kill >> 31));
}
// 8 consecutive bf16 of row[] starting at element m0 + s (m0 % 8 == 0, s = 0..3 wave-uniform)
__device__ __forceinline__ bf16x8 stem_frag_shift(const bf16* row, int m0, int s) {
    const uint4 a = *(const uint4*)(row + m0);
    const uint2 b = *(const uint2*)(row + m0 + 8);
    uint32_t w[6] = {a.x, a.y, a.z, a.w, b.x, b.y};
    uint4 o;
    if (s & 2) { w[0] = w[1]; w[1] = w[2]; w[2] = w[3]; w[3] = w[4]; w[4] = w[5]; }
    if (s & 1) {
        o.x = __builtin_amdgcn_alignbit(w[1], w[0], 16); o.y = __builtin_amdgcn_alignbit(w[2], w[1], 16);
        o.z = __builtin_amdgcn_alignbit(w[3], w[2], 16); o.w = __builtin_amdgcn_alignbit(w[4], w[3], 16);
    } else {
        o = make_uint4(w[0], w[1], w[2], w[3]);
    }
    return as_bf16x8(o);
}

// dW'[k'][n] partials: one [512][64] fp32 slab per workgroup.  8 waves: wave = nh*4 + rb; nh picks the channel half (n-tiles
// 2nh, 2nh+1), rb the run block (tile row i <-> run = rb*16 + i), and the wave's 7 MFMA k'-tiles are the 7 kw of that block --
// so the fragment shift (kw/2) and the even/odd plane (kw&1) are compile-time.  56 accumulator VGPRs per lane.
// BNG: the stem BatchNorm's backward apply folded into the G operand (tuber_stem_conv_bwd_weight_bn): G = the masked gradient dz0 of
// bn's output, X = the raw conv output c0, and the gradient tile is formed as bf16(cA*dz0 + cB*c0 + cC) -- the arithmetic of
// tuber_bn_bwd_apply -- while it is parked in LDS: the [M, 64] gradient of the raw conv output never exists in HBM.
template <int MINW, bool BNG>
__global__ __launch_bounds__(WG_THREADS, MINW) void stem_conv_bwd_w_kernel(const float* __restrict__ clip, const bf16* __restrict__ G,
                                                                       const bf16* __restrict__ X, const float* __restrict__ cA,
                                                                       const float* __restrict__ cB, const float* __restrict__ cC,
                                                                       float* __restrict__ partial, StemGeom g) {
    __shared__ __attribute__((aligned(16))) bf16 PE[64][EOW];
    __shared__ __attribute__((aligned(16))) bf16 PO[64][EOW];
    __shared__ __attribute__((aligned(16))) bf16 GT[64][72];            // [n][m], m contiguous, 144-byte rows (16 B aligned)
    __shared__ __attribute__((aligned(16))) float COEF[3][64];          // BNG: cA | cB | cC (read per tile: 12 registers less across the MFMAs)
    if (BNG && threadIdx.x < 64) { COEF[0][threadIdx.x] = cA[threadIdx.x]; COEF[1][threadIdx.x] = cB[threadIdx.x]; COEF[2][threadIdx.x] = cC[threadIdx.x]; }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, gq = lane >> 4;
    f32x4 acc[7][2];
#pragma unroll
    for (int a = 0; a < 7; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    // G tile staging: 64 m x 64 n in 2(m) x 4(n) blocks, one per thread: mi = 0..31 (x2 rows), ci = 0..15 (x4 channels)
    const int ci = (lane & 3) | ((lane >> 4) << 2), mi = ((lane >> 2) & 3) + 4 * wave;
    // software pipeline over the tiles: the patch / gradient loads of tile t+1 are issued right after tile t has been parked
    // in LDS, so that they are in flight during tile t's MFMAs
    float4 pre[5];
    StemMeta meta;
    uint2 gr[2], gx[2];
    const StemLane L = stem_lane(g);
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const int nh = wv >> 2, rb = wv & 3;
    const StemRuns<8> R = stem_runs<8>(g, wv);
    auto fetch = [&](int tile) {
        stem_fetch<8>(clip, g, tile, L, R, wv, pre, meta);
        const StemTile st = stem_tile(g, tile);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int w = st.wt * SW + mi * 2 + j;
            gr[j] = w < g.Wo ? *(const uint2*)(G + (st.row0 + mi * 2 + j) * 64 + ci * 4) : make_uint2(0, 0);
            if (BNG) gx[j] = w < g.Wo ? *(const uint2*)(X + (st.row0 + mi * 2 + j) * 64 + ci * 4) : make_uint2(0, 0);
        }
    };
    const StemWalk wk = stem_walk(g.ntiles);
    if (wk.tile < wk.end) fetch(wk.tile);
    for (int tile = wk.tile; tile < wk.end; tile += wk.step) {
        __syncthreads();                                     // previous tile's MFMAs are done with PE / PO / GT
        stem_park_eo(PE, PO, pre, meta, L, wv);
        {
            bf16x4 x0 = as_bf16x4(gr[0]), x1 = as_bf16x4(gr[1]);
            if (BNG) {      // (columns beyond Wo: dz0 = c0 = 0 were substituted, but cC is not zero -- those rows must stay zero)
                const StemTile st = stem_tile(g, tile);
                const bool ok0 = st.wt * SW + mi * 2 < g.Wo, ok1 = st.wt * SW + mi * 2 + 1 < g.Wo;
                const float4 a = *(const float4*)&COEF[0][ci * 4], b = *(const float4*)&COEF[1][ci * 4], c = *(const float4*)&COEF[2][ci * 4];
                const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w}, cv[4] = {c.x, c.y, c.z, c.w};
                const bf16x4 y0 = as_bf16x4(gx[0]), y1 = as_bf16x4(gx[1]);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    x0[q] = ok0 ? f2bf(fmaf(av[q], bf2f(x0[q]), fmaf(bv[q], bf2f(y0[q]), cv[q]))) : (bf16)0.f;
                    x1[q] = ok1 ? f2bf(fmaf(av[q], bf2f(x1[q]), fmaf(bv[q], bf2f(y1[q]), cv[q]))) : (bf16)0.f;
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) *(bf16x2*)&GT[ci * 4 + c][mi * 2] = bf16x2{x0[c], x1[c]};
        }
        if (tile + wk.step < wk.end) fetch(tile + wk.step);
        __syncthreads();
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) {
            bf16x8 gb[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) gb[nt] = as_bf16x8(*(const uint4*)&GT[(nh * 2 + nt) * 16 + li][ms * 32 + gq * 8]);
#pragma unroll
            for (int kw = 0; kw < 7; ++kw) {
                const int kt = kw;
                const bf16* row = (kw & 1) ? &PO[rb * 16 + li][0] : &PE[rb * 16 + li][0];
                const bf16x8 pa = stem_frag_shift(row, ms * 32 + gq * 8, kw >> 1);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) acc[kt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pa, gb[nt], acc[kt][nt], 0, 0, 0);
            }
        }
    }
    // D[i = run][j = n]: lane holds n = nt*16 + li, run = rb*16 + gq*4 + r; slab layout [k' = run*8 + kw][n] (kw = 7 rows zero)
    float* o = partial + (long)blockIdx.x * KP * 64;
#pragma unroll
    for (int kt = 0; kt < 7; ++kt) {
        const int kw = kt;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                o[(long)((rb * 16 + gq * 4 + r) * 8 + kw) * 64 + (nh * 2 + nt) * 16 + li] = acc[kt][nt][r];
    }
}

// W[64][441] fp32 -> Wp[64][512] bf16 with k' = run*8 + kw (zero padding)
__global__ void stem_pack_w_kernel(const float* __restrict__ W, bf16* __restrict__ Wp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * KP) return;
    const int n = i / KP, kp = i % KP, run = kp >> 3, kw = kp & 7;
    Wp[i] = (run < 63 && kw < 7) ? f2bf(W[n * 441 + run * 7 + kw]) : (bf16)0.f;
}
// dW[n][run*7+kw] (+)= sum_wg partial[wg][k'][n]
__global__ __launch_bounds__(1024) void stem_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ dW, int nwg, int accumulate) {
    __shared__ float red[32][33];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + cl;                   // over k'*64 + n
    float a = 0.f;
    for (int s = rg; s < nwg; s += 32) a += partial[(long)s * KP * 64 + i];
    red[rg][cl] = a;
    __syncthreads();
    if (rg == 0) {
        a = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) a += red[k][cl];
        const int kp = i / 64, n = i % 64, run = kp >> 3, kw = kp & 7;
        if (run < 63 && kw < 7) {
            float* o = dW + n * 441 + run * 7 + kw;
            *o = accumulate ? *o + a : a;
        }
    }
}

static StemGeom stem_geom(int B, int T, int H, int W) {
    StemGeom g;
    g.B = B; g.T = T; g.H = H; g.W = W;
    g.Ho = (H + 6 - 7) / 2 + 1; g.Wo = (W + 6 - 7) / 2 + 1;
    g.tilesW = (g.Wo + SW - 1) / SW;
    g.ntiles = B * T * g.Ho * g.tilesW;
    return g;
}

extern "C" {

// persistent grid of both kernels (= partial-statistics rows of the forward, slabs of the weight gradient)
int tuber_stem_conv_blocks(int B, int T, int H, int W) {
    constexpr int cap = 768;      // 159 VGPRs: three workgroups per CU
    const StemGeom g = stem_geom(B, T, H, W);
    return g.ntiles < cap ? g.ntiles : cap;
}
// slabs of the weight gradient (two 8-wave workgroups per CU)
int tuber_stem_conv_wgrad_blocks(int B, int T, int H, int W) {
    constexpr int cap = 256;
    const StemGeom g = stem_geom(B, T, H, W);
    return g.ntiles < cap ? g.ntiles : cap;
}

int tuber_stem_pack_weight(const float* W, void* Wp, hipStream_t stream) {
    hipLaunchKernelGGL(stem_pack_w_kernel, dim3(64 * KP / 256), dim3(256), 0, stream, W, (bf16*)Wp);
    TUBER_RETURN_LAUNCH();
}

// out [B*T*Ho*Wo, 64] bf16 raw conv output (NDHWC); st0/st1 [blocks][64] partial (sum, sum^2) or NULL
int tuber_stem_conv_fwd(const float* clip, const void* Wp, void* out, float* st0, float* st1, int B, int T, int H, int W,
                        hipStream_t stream) {
    if (B <= 0 || T <= 0 || H < 7 || W < 7) return TUBER_EINVAL;
    const StemGeom g = stem_geom(B, T, H, W);
    hipLaunchKernelGGL(stem_conv_fwd_kernel, dim3(tuber_stem_conv_blocks(B, T, H, W)), dim3(256), 0, stream, clip, (const bf16*)Wp,
                       (bf16*)out, st0, st1, g);
    TUBER_RETURN_LAUNCH();
}

// dW [64][441] fp32 (+)= conv weight gradient from G = d(loss)/d(raw conv output) [M,64] bf16;
// partial must hold tuber_stem_conv_wgrad_blocks() * 512 * 64 floats
int tuber_stem_conv_bwd_weight(const float* clip, const void* G, float* partial, float* dW, int accumulate, int B, int T, int H, int W,
                               hipStream_t stream) {
    if (B <= 0 || T <= 0 || H < 7 || W < 7) return TUBER_EINVAL;
    const StemGeom g = stem_geom(B, T, H, W);
    const int nwg = tuber_stem_conv_wgrad_blocks(B, T, H, W);
    if (nwg > 256) hipLaunchKernelGGL((stem_conv_bwd_w_kernel<4, false>), dim3(nwg), dim3(WG_THREADS), 0, stream, clip, (const bf16*)G, nullptr, nullptr, nullptr, nullptr, partial, g);
    else hipLaunchKernelGGL((stem_conv_bwd_w_kernel<2, false>), dim3(nwg), dim3(WG_THREADS), 0, stream, clip, (const bf16*)G, nullptr, nullptr, nullptr, nullptr, partial, g);
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(KP * 64 / 32), dim3(1024), 0, stream, partial, dW, nwg, accumulate);
    TUBER_RETURN_LAUNCH();
}

// The same with the stem BatchNorm's backward apply folded in: dz0 = gradient of bn's output after the ReLU / pool backward
// (tuber_stem_pool_bwd), c0 = raw conv output, cA / cB / cC [64] from tuber_bn_bwd_finalize; G = bf16(cA*dz0 + cB*c0 + cC) is formed on
// load (bit-identical to tuber_bn_bwd_apply followed by tuber_stem_conv_bwd_weight, without the [M, 64] tensor in between).
int tuber_stem_conv_bwd_weight_bn(const float* clip, const void* dz0, const void* c0, const float* cA, const float* cB, const float* cC,
                                  float* partial, float* dW, int accumulate, int B, int T, int H, int W, hipStream_t stream) {
    if (B <= 0 || T <= 0 || H < 7 || W < 7 || !dz0 || !c0 || !cA || !cB || !cC) return TUBER_EINVAL;
    const StemGeom g = stem_geom(B, T, H, W);
    const int nwg = tuber_stem_conv_wgrad_blocks(B, T, H, W);
    if (nwg > 256) hipLaunchKernelGGL((stem_conv_bwd_w_kernel<4, true>), dim3(nwg), dim3(WG_THREADS), 0, stream, clip, (const bf16*)dz0, (const bf16*)c0, cA, cB, cC, partial, g);
    else hipLaunchKernelGGL((stem_conv_bwd_w_kernel<2, true>), dim3(nwg), dim3(WG_THREADS), 0, stream, clip, (const bf16*)dz0, (const bf16*)c0, cA, cB, cC, partial, g);
    hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3(KP * 64 / 32), dim3(1024), 0, stream, partial, dW, nwg, accumulate);
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
