// MFMA bf16 GEMMs for the TubeR hot path (gfx950, wave64).
//
//   gemm_nt : C[M,N] = f(A)[M,K] . B[N,K]^T      forward of every 1x1x1 conv / nn.Linear,
//                                                 and the data-gradient (with W^T as B)
//   gemm_tn : C[N,K] = sum_m G[m,N]^T . f(A)[m,K] weight gradient (split over M, fp32 partials)
//
// Activations are NDHWC / token-major, i.e. row-major [M, C] bf16 with the channel dimension
// contiguous, so a pointwise conv IS a row-major GEMM (reference: nn.Conv3d(k=1) in
// models/backbones/ir_CSN_152.py:41,58,155-161 and nn.Linear throughout models/transformer/*).
//
// Tiling: 256 threads = 4 waves; BK = 64 (128-byte LDS rows, one full cache line per row per
// k-tile); LDS rows XOR-swizzled in 16-byte chunks so every ds_read_b128 lane group is
// conflict-free; double-buffered LDS with register prefetch of the next k-tile.
// The MFMA is issued "swapped" (weights as the A operand, activations as the B operand) with a
// permuted weight-row -> MFMA-row assignment, so each lane ends up holding 4*NT consecutive
// output columns of one output row and stores them with 16-byte stores.
//
// Fused prologue (A operand): BatchNorm apply + ReLU of the producer's raw conv output
// (a = relu(x*scale[k]+shift[k])), optional strided row gather (down_sample convs).
// Fused epilogues: bias / ReLU / residual add; per-column partial statistics (sum, sum of
// squares) for training-mode BatchNorm; ReLU-mask + BN-backward partial statistics.
#include <cstdlib>
#include <cstring>

#include "common.h"

enum { A_PLAIN = 0, A_BN_RELU = 1, A_BN_BWD = 2, A_ADD = 3 };
// A_BN_BWD: a = cA[k]*A + cB[k]*A2 + cC[k] (BatchNorm backward apply; measured slower than the apply kernel, not instantiated)
// A_ADD:    a = A + A2 for the output-column tiles below add_ncols, a = A for the rest: a packed attention in-projection whose
//           q / k rows see x + pos (with_pos_embed) and whose v rows see x -- one GEMM instead of an add kernel and two GEMMs
enum { EPI_PLAIN = 0, EPI_STATS = 1, EPI_BWD = 2, EPI_JOIN = 3, EPI_JOIN_SR = 4, EPI_JOIN_DS = 5, EPI_EVAL = 6, EPI_JOIN_M = 7, EPI_JOIN_SR_M = 8, EPI_JOIN_DS_M = 9 };
// EPI_JOIN_M (round 6): EPI_JOIN whose ReLU mask [Ym > 0] comes as a BIT FIELD (Ymask [M][N / 8] bytes, written by tuber_block_out_fwd_mask) instead of the
// bf16 tensor y itself: a lane's 8 output columns of a row are one byte.  The join GEMMs run at the bandwidth of their side operands (layer3: 49 MB per launch in
// 17 us); a timing-only build without the y loads returned 0.135 ms per step, the bit field keeps 15 / 16 of that.
// EPI_EVAL (round 6, eval forward only): conv4 + the EVAL-mode bn4 + the residual join + ReLU in the GEMM epilogue --
// y = relu(acc * m_scale[n] + m_shift[n] + R32[m][n]) written as the bf16 GEMM operand of the next block AND as the fp32 residual stream (C32);
// an eval-mode BatchNorm is a constant affine map, so unlike in training nothing has to wait for the conv output's statistics: c4 never reaches HBM (it is still rounded to bf16 in
// registers, so y / C32 are bit-identical to tuber_gemm_nt + tuber_block_out_fwd_f32: the eval precision mode's validated rounding points stay as they are).
#define IS_JOIN(E) ((E) == EPI_JOIN || (E) == EPI_JOIN_SR || (E) == EPI_JOIN_DS)
#define IS_JOIN_M(E) ((E) == EPI_JOIN_M || (E) == EPI_JOIN_SR_M || (E) == EPI_JOIN_DS_M)      /* the bit-field forms of the three */
// EPI_JOIN_DS: EPI_JOIN below a stage's FIRST block: a second statistics operand Dm (the raw output of its projection shortcut) and a third
// row sum dz*Dm (stat2) for the shortcut BatchNorm's backward -- what tuber_block_out_bwd writes for such a block
// EPI_JOIN_SR: EPI_JOIN whose residual R is the gradient of a STRIDED projection shortcut (rows = the sampled positions only; its own
// instantiation: the row decode cost the hot EPI_JOIN kernel 8 spilled registers when it was a run-time branch in the same epilogue)
// EPI_JOIN: the data-gradient GEMM of one bottleneck's conv1 fused with the join backward of the bottleneck below it:
//   dz = (acc + R) * [Ym > 0]   (R = identity-shortcut gradient, Ym = the lower block's output y = relu(bn4(c4) + x))
//   + BatchNorm-backward partial statistics (sum dz, sum dz * Cm) with Cm = the lower block's raw conv4 output.

struct GemmNT {
    const bf16* A; long lda;
    const bf16* B; long ldb;
    void* C; long ldc;
    int M, N, K;
    const float* a_scale; const float* a_shift;   // A_BN_RELU (scale, shift) / A_BN_BWD (cA, cB)
    const bf16* A2; long lda2; const float* a_coef2;   // A_BN_BWD: second operand (the BN input x) and cC;  A_ADD: the addend
    int add_ncols;                                     // A_ADD: output columns [0, add_ncols) use A + A2 (multiple of the tile width)
    int gather; int To, Ho, Wo, Ti, Hi, Wi, st, ss; // row gather (strided 1x1x1 conv)
    const float* bias; const bf16* R; long ldr; int relu; int out_f32;   // EPI_PLAIN
    float* stat0; float* stat1;                   // EPI_STATS / EPI_BWD partials [tiles_m*WM][N]
    const bf16* Cm; long ldcm; const float* m_scale; const float* m_shift;  // EPI_BWD mask source
    const bf16* Ym; long ldym;                    // EPI_JOIN: mask source (dz = v * [Ym > 0]); Cm is the statistics operand
    const uint8_t* Ymask;                         // EPI_JOIN_M: the same mask as a bit field [M][N / 8]
    const bf16* Dm; long lddm; float* stat2;      // EPI_JOIN_DS
                                                  // EPI_JOIN_SR: R holds one row per STRIDED sample (n, t/st, h/ss, w/ss) of the M = n*Ti*Hi*Wi output
                                                  // rows (To..ss above): the data gradient of a stage's strided projection shortcut, added where it belongs
    const float* R32; long ldr32; float* C32; long ldc32;   // EPI_EVAL: fp32 residual stream in / out (m_scale / m_shift: the BatchNorm affine of the output columns)
    float alpha;                                  // accumulators are scaled by alpha before the epilogue
    uint32_t drop_thresh; float drop_inv_keep; const uint64_t* seed_ptr; uint64_t salt;   // EPI_PLAIN: Dropout after bias/residual/ReLU
};

__device__ __forceinline__ int swz_act(int row) { return (row >> 1) & 7; }
template <int NT>
__device__ __forceinline__ int swz_wgt(int row) { return (((row / (4 * NT)) & 3) << 1) | ((row >> 1) & 1); }

// bytes of one wave's private LDS image of the wave split-K form: the staged k-tile (A BM x 128 B | B 64 x 128 B) or, afterwards, its four
// partial sub-tiles (4 x (BM/32) x 2 accumulator blocks x 64 lanes x 16 B)
__host__ __device__ constexpr int wsk_image_bytes(int bm) { return (bm + 64) * 128 > 4 * (bm / 32) * 2 * 1024 ? (bm + 64) * 128 : 4 * (bm / 32) * 2 * 1024; }

// OCC = waves per SIMD the register allocation is held to (= workgroups per CU: a workgroup is one wave per SIMD).  It must not
// exceed what the LDS allows anyway -- 2*(BM+BN)*128 B per workgroup of the 160 KB: 64x64 -> 4-5, 64x128 -> 3, 128x128 -> 2 --
// or the compiler spills the prefetch registers to scratch inside the k-loop for occupancy the kernel can never reach.
// WSK = 1 ("wave split-K", 64x64 tiles, plain A only): the four waves do NOT tile the output -- each computes the WHOLE 64x64 tile for
// every fourth k-tile, staged in a wave-private LDS image (no workgroup barrier per k-tile), and the four partial tiles are summed
// through LDS in wave order before the common epilogue.  For the shapes whose time is their k-loop (layer3 / layer4 1x1x1 convs with
// K >= 1024 on 352 workgroups, the FFN GEMMs with K = 2048 on 11 .. 44): a k-tile costs ~0.24 us of barriers and exposed latency in
// the shared-tile loop whatever it computes.
// FULL: N is a multiple of the tile width and every leading dimension of the epilogue's tensors a multiple of 8 (all conv / FFN shapes of
// the model): the per-column bounds checks (an exec-mask branch per output column and row block) and the scalar fall-back paths of the
// epilogue are compiled out.
template <int BM, int BN, int WM, int WN, int G, int AMODE, int EPI, int OCC, int WSK = 0, bool FULL = false>
__global__ __launch_bounds__(256, OCC) void gemm_nt_kernel(GemmNT p) {
    // the *_M epilogues are their base epilogue with the ReLU mask read as a bit field (YM): everything below switches on the base (E0)
    constexpr int E0 = EPI == EPI_JOIN_M ? EPI_JOIN : EPI == EPI_JOIN_SR_M ? EPI_JOIN_SR : EPI == EPI_JOIN_DS_M ? EPI_JOIN_DS : EPI;
    constexpr bool YM = IS_JOIN_M(EPI);
    static_assert(!WSK || ((BM == 64 || BM == 96) && BN == 64 && WM == 2 && WN == 2 && AMODE == A_PLAIN), "wave split-K: 64x64 / 96x64 tiles, plain A");
    constexpr int KS = 1, kg = 0;               // (the in-workgroup k-split of round 1 was measured and dropped; the index math keeps its shape)
    constexpr int TM = BM / WM, TN = BN / WN, MT = TM / 16, NT = TN / 16;
    constexpr int CA = BM / 32, CB = BN / 32;            // 16-byte chunks per thread per k-tile
    constexpr int STAGE = (BM + BN) * 128;
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;      // thread / wave id INSIDE the group
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int L = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_n = L % tiles_n, tile_m = L / tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nkt = p.K / 64;                                   // k-tiles of the whole problem
    const int nk = (nkt - kg + KS - 1) / KS;                    // ... of this group (local tile j <-> global tile j*KS + kg)
    const int nk_max = (nkt + KS - 1) / KS;                     // barrier count is uniform over the groups
    char* const gsm = smem + kg * 2 * STAGE;                    // this group's double buffer

    // ---- staging assignment: chunk c = tid + 256*i -> (row = c>>3, q = c&7) ----
    const int q = tid & 7;
    const bf16* a_ptr[CA];
    constexpr bool TWO = AMODE == A_BN_BWD || AMODE == A_ADD;      // second A operand
    const bf16* a2_ptr[TWO ? CA : 1];
    bool a_ok[CA];
#pragma unroll
    for (int i = 0; i < CA; ++i) {
        const int row = (tid >> 3) + 32 * i;
        // rows beyond M / N are loaded from the last valid row instead of being predicated to zero: an exec-masked 16-byte load
        // costs ~8 instructions (zero fill, saveexec, branch) per k-tile; their products only reach output rows / columns that are
        // never stored, and the statistics epilogues skip them
        const int m = min(m0 + row, p.M - 1);
        a_ok[i] = m0 + row < p.M;
        long src = m;
        if (p.gather) {
            int w = m % p.Wo; int r = m / p.Wo;
            int h = r % p.Ho; r /= p.Ho;
            int t = r % p.To; int n = r / p.To;
            src = (((long)n * p.Ti + (long)t * p.st) * p.Hi + (long)h * p.ss) * p.Wi + (long)w * p.ss;
        }
        a_ptr[i] = p.A + src * p.lda + q * 8;
        if (TWO) a2_ptr[i] = p.A2 + (long)m * p.lda2 + q * 8;
    }
    const bf16* b_ptr[CB];
    bool b_ok[CB];
#pragma unroll
    for (int i = 0; i < CB; ++i) {
        const int row = (tid >> 3) + 32 * i;
        b_ok[i] = (n0 + row) < p.N;
        b_ptr[i] = p.B + (long)min(n0 + row, p.N - 1) * p.ldb + q * 8;
    }

    // k-tiles are fetched in groups of G: all global loads of a group are in flight together (one HBM latency per
    // group instead of one per tile -- these GEMMs have K <= 2048, often only 1-4 tiles), then each tile goes
    // registers -> (BN prologue) -> LDS -> MFMA.
    uint4 ra[G][CA], rb[G][CB];
    uint4 ra2[TWO ? G : 1][CA];
    const bool add_on = AMODE == A_ADD && n0 < p.add_ncols;         // uniform per workgroup
#pragma unroll
    for (int j = 0; j < G; ++j) {                 // fully defined on every path, so the arrays stay in registers
#pragma unroll
        for (int i = 0; i < CA; ++i) { ra[j][i] = make_uint4(0, 0, 0, 0); if (TWO) ra2[j][i] = make_uint4(0, 0, 0, 0); }
#pragma unroll
        for (int i = 0; i < CB; ++i) rb[j][i] = make_uint4(0, 0, 0, 0);
    }
    float* lsc = (float*)(smem + KS * 2 * STAGE);  // A_BN_RELU: scale[K] | shift[K] staged once;  A_BN_BWD: cA | cB | cC
    float* lsh = lsc + p.K;
    float* lsc2 = lsh + p.K;
    auto load_tile = [&](int kl, uint4 (&xa)[CA], uint4 (&xb)[CB], uint4 (&xa2)[CA]) {
        const int k0 = (kl * KS + kg) * 64;
#pragma unroll
        for (int i = 0; i < CA; ++i) xa[i] = *(const uint4*)(a_ptr[i] + k0);
        if (AMODE == A_BN_BWD || (AMODE == A_ADD && add_on)) {
#pragma unroll
            for (int i = 0; i < CA; ++i) xa2[i] = *(const uint4*)(a2_ptr[i] + k0);
        }
#pragma unroll
        for (int i = 0; i < CB; ++i) xb[i] = *(const uint4*)(b_ptr[i] + k0);
    };
    auto store_tile = [&](int kl, int buf, const uint4 (&xa)[CA], const uint4 (&xb)[CB], const uint4 (&xa2)[CA]) {
        const int kt = kl * KS + kg;
        char* sa = gsm + buf * STAGE;
        char* sb = sa + BM * 128;
        float sc[8], sh[8], s2[8];
        if (AMODE == A_BN_BWD) {
            const float4* c4 = (const float4*)(lsc2 + kt * 64 + q * 8);
            const float4 c0 = c4[0], c1 = c4[1];
            s2[0] = c0.x; s2[1] = c0.y; s2[2] = c0.z; s2[3] = c0.w; s2[4] = c1.x; s2[5] = c1.y; s2[6] = c1.z; s2[7] = c1.w;
        }
        if (AMODE == A_BN_RELU || AMODE == A_BN_BWD) {
            const float4* s4 = (const float4*)(lsc + kt * 64 + q * 8);
            const float4* h4 = (const float4*)(lsh + kt * 64 + q * 8);
            const float4 s0 = s4[0], s1 = s4[1], h0 = h4[0], h1 = h4[1];
            sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
            sh[0] = h0.x; sh[1] = h0.y; sh[2] = h0.z; sh[3] = h0.w; sh[4] = h1.x; sh[5] = h1.y; sh[6] = h1.z; sh[7] = h1.w;
        }
#pragma unroll
        for (int i = 0; i < CA; ++i) {
            const int row = (tid >> 3) + 32 * i;
            uint4 v = xa[i];
            if (AMODE == A_BN_RELU) {
                bf16x8 x = as_bf16x8(v), y;
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = f2bf(fmaxf(fmaf(bf2f(x[e]), sc[e], sh[e]), 0.f));
                v = as_uint4(y);
            }
            if (AMODE == A_BN_BWD) {
                const bf16x8 x = as_bf16x8(v), x2 = as_bf16x8(xa2[i]);
                bf16x8 y;
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = f2bf(fmaf(bf2f(x[e]), sc[e], fmaf(bf2f(x2[e]), sh[e], s2[e])));
                v = as_uint4(y);
            }
            if (AMODE == A_ADD && add_on) {
                const bf16x8 x = as_bf16x8(v), x2 = as_bf16x8(xa2[i]);
                bf16x8 y;
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = f2bf(bf2f(x[e]) + bf2f(x2[e]));       // = the bf16 sum the stand-alone add kernel stored
                v = as_uint4(y);
            }
            *(uint4*)(sa + row * 128 + ((q ^ swz_act(row)) << 4)) = v;
        }
#pragma unroll
        for (int i = 0; i < CB; ++i) {
            const int row = (tid >> 3) + 32 * i;
            *(uint4*)(sb + row * 128 + ((q ^ swz_wgt<NT>(row)) << 4)) = xb[i];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, li = lane & 15;
    int a_row[MT], w_row[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a_row[i] = wm * TM + i * 16 + li;
#pragma unroll
    for (int j = 0; j < NT; ++j) w_row[j] = wn * TN + (li >> 2) * (4 * NT) + j * 4 + (li & 3);

    auto compute = [&](int buf) {
        const char* sa = gsm + buf * STAGE;
        const char* sb = sa + BM * 128;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int qq = ks * 4 + g;
            bf16x8 xa[MT], wb[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i)
                xa[i] = as_bf16x8(*(const uint4*)(sa + a_row[i] * 128 + ((qq ^ swz_act(a_row[i])) << 4)));
#pragma unroll
            for (int j = 0; j < NT; ++j)
                wb[j] = as_bf16x8(*(const uint4*)(sb + w_row[j] * 128 + ((qq ^ swz_wgt<NT>(w_row[j])) << 4)));
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[j], xa[i], acc[i][j], 0, 0, 0);
        }
    };
    // ---- first k-group and the epilogue's side inputs go out before anything waits ----
    constexpr int NC = 4 * NT;
    const int nb = n0 + wn * TN + g * NC;
    const bool vec_ok = FULL || ((nb + NC <= p.N) && ((p.ldc & 7) == 0) && ((p.N & 7) == 0));
    if constexpr (!WSK) {
#pragma unroll
        for (int j = 0; j < G; ++j)
            if (j < nk) load_tile(j, ra[j], rb[j], ra2[TWO ? j : 0]);
    }
    constexpr bool SIDE = E0 == EPI_BWD || IS_JOIN(E0);
    uint4 side[SIDE ? MT : 1][NC / 8];         // EPI_BWD / EPI_JOIN: the statistics operand c, fetched behind the k-loop
    uint4 sidey[IS_JOIN(E0) && !YM ? MT : 1][NC / 8];       // EPI_JOIN: the mask source y
    uint32_t ybits[YM ? MT : 1][NC / 8];              // EPI_JOIN_M: its bit field, one byte per 8 columns
    uint4 sided[E0 == EPI_JOIN_DS ? MT : 1][NC / 8];  // EPI_JOIN_DS: the projection shortcut's raw output
    const bool side_vec = SIDE && (FULL || (vec_ok && ((p.ldcm & 7) == 0) && (!IS_JOIN(E0) || (p.ldym & 7) == 0) && (E0 != EPI_JOIN_DS || (p.lddm & 7) == 0)));
    if (SIDE && side_vec) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = m0 + wm * TM + i * 16 + li;
#pragma unroll
            for (int c8 = 0; c8 < NC / 8; ++c8) {
                side[SIDE ? i : 0][c8] = m < p.M ? *(const uint4*)(p.Cm + (long)m * p.ldcm + nb + c8 * 8) : make_uint4(0, 0, 0, 0);
                if (IS_JOIN(E0) && !YM)
                    sidey[IS_JOIN(E0) && !YM ? i : 0][c8] = m < p.M ? *(const uint4*)(p.Ym + (long)m * p.ldym + nb + c8 * 8) : make_uint4(0, 0, 0, 0);
                if (YM) ybits[YM ? i : 0][c8] = m < p.M ? p.Ymask[(long)m * (p.N >> 3) + ((nb + c8 * 8) >> 3)] : 0u;
                if (E0 == EPI_JOIN_DS)
                    sided[E0 == EPI_JOIN_DS ? i : 0][c8] = m < p.M ? *(const uint4*)(p.Dm + (long)m * p.lddm + nb + c8 * 8) : make_uint4(0, 0, 0, 0);
            }
        }
    }
    if (AMODE == A_BN_RELU || AMODE == A_BN_BWD) {
        for (int i = threadIdx.x; i < p.K; i += 256 * KS) {
            lsc[i] = p.a_scale[i]; lsh[i] = p.a_shift[i];
            if (AMODE == A_BN_BWD) lsc2[i] = p.a_coef2[i];
        }
        __syncthreads();
    }
    if constexpr (WSK) {
        // ---- wave split-K: this wave owns k-tiles wave, wave+4, ...; image = A 64 x 128 B | B 64 x 128 B, private to the wave ----
        // (BM = 96, round 5: the layer3 / layer4 shapes have 352 tiles of 64 x 64 for 256 CUs -- 96 CUs carry two workgroups and set the
        // launch's duration; 96-row tiles are 236 resp. 240 workgroups, at most one per CU, each 1.5 x the MFMA work behind the same
        // latency chain)
        constexpr int AR = BM / 8;                               // staged A rows per lane
        constexpr int IMG = wsk_image_bytes(BM);                 // per-wave image: staging (A BM x 128 B | B 64 x 128 B) or its four partial sub-tiles
        char* const sa = smem + wave * IMG;
        char* const sb = sa + BM * 128;
        const int lq = lane & 7, lr = lane >> 3;                 // chunk c = lane + 64 i -> row lr + 8 i, 16-byte chunk lq
        const bf16* pa[AR];
        const bf16* pb[8];
#pragma unroll
        for (int i = 0; i < AR; ++i) pa[i] = p.A + (long)min(m0 + lr + 8 * i, p.M - 1) * p.lda + lq * 8;
#pragma unroll
        for (int i = 0; i < 8; ++i) pb[i] = p.B + (long)min(n0 + lr + 8 * i, p.N - 1) * p.ldb + lq * 8;
        uint4 xa[AR], xb[8];
#pragma unroll
        for (int i = 0; i < AR; ++i) xa[i] = make_uint4(0, 0, 0, 0);     // defined on every path: stays in VGPRs
#pragma unroll
        for (int i = 0; i < 8; ++i) xb[i] = make_uint4(0, 0, 0, 0);
        auto fetch = [&](int kt, uint4 (&ya)[AR], uint4 (&yb)[8]) {
#pragma unroll
            for (int i = 0; i < AR; ++i) ya[i] = *(const uint4*)(pa[i] + kt * 64);
#pragma unroll
            for (int i = 0; i < 8; ++i) yb[i] = *(const uint4*)(pb[i] + kt * 64);
        };
        f32x4 acc4[2][2][MT][NT];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc4[a][b][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (wave < nkt) fetch(wave, xa, xb);
        for (int kt = wave; kt < nkt; kt += 4) {
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int row = lr + 8 * i;
                *(uint4*)(sa + row * 128 + ((lq ^ swz_act(row)) << 4)) = xa[i];
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = lr + 8 * i;
                *(uint4*)(sb + row * 128 + ((lq ^ swz_wgt<NT>(row)) << 4)) = xb[i];
            }
            if (kt + 4 < nkt) fetch(kt + 4, xa, xb);
            __builtin_amdgcn_wave_barrier();                      // one wave's LDS writes and reads stay in order: no workgroup barrier
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int qq = ks * 4 + g;
                bf16x8 fa[2][MT], fb[2][NT];
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int i = 0; i < MT; ++i) {
                        const int r = a * TM + i * 16 + li;
                        fa[a][i] = as_bf16x8(*(const uint4*)(sa + r * 128 + ((qq ^ swz_act(r)) << 4)));
                    }
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < NT; ++j) {
                        const int r = b * TN + (li >> 2) * (4 * NT) + j * 4 + (li & 3);
                        fb[b][j] = as_bf16x8(*(const uint4*)(sb + r * 128 + ((qq ^ swz_wgt<NT>(r)) << 4)));
                    }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
#pragma unroll
                        for (int i = 0; i < MT; ++i)
#pragma unroll
                            for (int j = 0; j < NT; ++j)
                                acc4[a][b][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[b][j], fa[a][i], acc4[a][b][i][j], 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---- sum the four partial tiles: every wave parks all four sub-tiles in ITS OWN image (16 KB: nobody else reads or writes it
        // before the barrier), then wave (wm, wn) adds the four copies of its sub-tile in wave order ----
        f32x4* mine = (f32x4*)sa;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) mine[(((a * 2 + b) * MT + i) * NT + j) * 64 + lane] = acc4[a][b][i][j];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int w = 0; w < 4; ++w) t += ((const f32x4*)(smem + w * IMG))[(((wm * 2 + wn) * MT + i) * NT + j) * 64 + lane];
                acc[i][j] = t;
            }
    } else {
    // software pipeline over k-tiles: register set j holds tile g0+j; as soon as it has been written to LDS the same
        // registers are re-armed with tile g0+G+j, so G tiles of global loads stay in flight behind the MFMA work.
        int buf = 0;
        for (int g0 = 0; g0 < nk_max; g0 += G) {
#pragma unroll
            for (int j = 0; j < G; ++j) {
                if (g0 + j < nk_max) {
                    const bool have = g0 + j < nk;           // (odd tile counts: the second group idles through its last barrier)
                    if (have) store_tile(g0 + j, buf, ra[j], rb[j], ra2[TWO ? j : 0]);
                    if (g0 + G + j < nk) load_tile(g0 + G + j, ra[j], rb[j], ra2[TWO ? j : 0]);
                    __syncthreads();
                    if (have) compute(buf);
                    buf ^= 1;
                }
            }
        }
    }
    const bool epi_on = true;

    // ---- epilogue: lane holds, for each mt, columns nb .. nb+4*NT-1 of row m ----
    float s0[NC], s1[NC], s2[E0 == EPI_JOIN_DS ? NC : 1];
    if (E0 != EPI_PLAIN) {
#pragma unroll
        for (int c = 0; c < NC; ++c) { s0[c] = 0.f; s1[c] = 0.f; if (E0 == EPI_JOIN_DS) s2[E0 == EPI_JOIN_DS ? c : 0] = 0.f; }
    }
    float msc[NC], msh[NC];
    if (E0 == EPI_BWD || E0 == EPI_EVAL) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const bool ok = FULL || nb + c < p.N;
            msc[c] = (p.m_scale && ok) ? p.m_scale[nb + c] : 1.f;
            msh[c] = (p.m_shift && ok) ? p.m_shift[nb + c] : 0.f;
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + wm * TM + i * 16 + li;
        const bool mok = m < p.M && epi_on;
        float v[NC];
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[j * 4 + r] = acc[i][j][r] * p.alpha;
        if (E0 == EPI_PLAIN) {
            if (p.bias) {
#pragma unroll
                for (int c = 0; c < NC; ++c) if (FULL || nb + c < p.N) v[c] += p.bias[nb + c];
            }
            if (p.R && mok) {
                if (FULL || (vec_ok && (p.ldr & 7) == 0)) {
#pragma unroll
                    for (int c8 = 0; c8 < NC / 8; ++c8) {
                        const bf16x8 rv = as_bf16x8(*(const uint4*)(p.R + (long)m * p.ldr + nb + c8 * 8));
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[c8 * 8 + e] += bf2f(rv[e]);
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < NC; ++c) if (FULL || nb + c < p.N) v[c] += bf2f(p.R[(long)m * p.ldr + nb + c]);
                }
            }
            if (p.relu) {
#pragma unroll
                for (int c = 0; c < NC; ++c) v[c] = fmaxf(v[c], 0.f);
            }
            if (p.drop_thresh) {
                const uint64_t seed = (p.seed_ptr ? *p.seed_ptr : 0ull) * 0x9E3779B97F4A7C15ull + p.salt;
                bool keep[NC];
                dropout_keep_run<NC>(seed, (uint64_t)m * p.N + nb, p.drop_thresh, keep);
#pragma unroll
                for (int c = 0; c < NC; ++c) v[c] = keep[c] ? v[c] * p.drop_inv_keep : 0.f;
            }
        } else if (E0 == EPI_EVAL) {
#pragma unroll
            for (int c = 0; c < NC; ++c) v[c] = fmaf(bf2f(f2bf(v[c])), msc[c], msh[c]);      // c4 passes through bf16 as in the two-launch path: bit-identical outputs
            if (mok) {
                const float* r32 = p.R32 + (long)m * p.ldr32 + nb;
                float* o32 = p.C32 + (long)m * p.ldc32 + nb;
#pragma unroll
                for (int c4 = 0; c4 < NC / 4; ++c4) {
                    const float4 r = *(const float4*)(r32 + c4 * 4);
                    float4 o;
                    o.x = v[c4 * 4] = fmaxf(v[c4 * 4] + r.x, 0.f);
                    o.y = v[c4 * 4 + 1] = fmaxf(v[c4 * 4 + 1] + r.y, 0.f);
                    o.z = v[c4 * 4 + 2] = fmaxf(v[c4 * 4 + 2] + r.z, 0.f);
                    o.w = v[c4 * 4 + 3] = fmaxf(v[c4 * 4 + 3] + r.w, 0.f);
                    *(float4*)(o32 + c4 * 4) = o;
                }
            }
        } else if (E0 == EPI_STATS) {
            if (mok) {                                   // rows beyond M hold a copy of row M-1
#pragma unroll
                for (int c = 0; c < NC; ++c) { s0[c] += v[c]; s1[c] += v[c] * v[c]; }
            }
        } else if (IS_JOIN(E0)) {
            if (mok) {
                long rrow = m;
                bool rok = p.R != nullptr;
                if constexpr (E0 == EPI_JOIN_SR) {      // the strided projection shortcut's gradient lives at the sampled positions only
                    const int w = m % p.Wi; int q = m / p.Wi;
                    const int h = q % p.Hi; q /= p.Hi;
                    const int t = q % p.Ti; const int n = q / p.Ti;
                    rok = rok && t % p.st == 0 && h % p.ss == 0 && w % p.ss == 0;
                    rrow = ((long)(n * p.To + t / p.st) * p.Ho + h / p.ss) * p.Wo + w / p.ss;
                }
                if (rok) {
                    if (FULL || (vec_ok && (p.ldr & 7) == 0)) {
#pragma unroll
                        for (int c8 = 0; c8 < NC / 8; ++c8) {
                            const bf16x8 rv = as_bf16x8(*(const uint4*)(p.R + rrow * p.ldr + nb + c8 * 8));
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[c8 * 8 + e] += bf2f(rv[e]);
                        }
                    } else {
#pragma unroll
                        for (int c = 0; c < NC; ++c) if (FULL || nb + c < p.N) v[c] += bf2f(p.R[rrow * p.ldr + nb + c]);
                    }
                }
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    if (FULL || nb + c < p.N) {
                        const float cv = side_vec ? bf2f(as_bf16x8(side[SIDE ? i : 0][c >> 3])[c & 7]) : bf2f(p.Cm[(long)m * p.ldcm + nb + c]);
                        bool pos;
                        if constexpr (YM) pos = (ybits[YM ? i : 0][c >> 3] >> (c & 7)) & 1u;
                        else pos = (side_vec ? bf2f(as_bf16x8(sidey[IS_JOIN(E0) && !YM ? i : 0][c >> 3])[c & 7]) : bf2f(p.Ym[(long)m * p.ldym + nb + c])) > 0.f;
                        // the stored dz is bf16: the statistics are taken of the ROUNDED value, like the stand-alone join kernel does
                        v[c] = pos ? bf2f(f2bf(v[c])) : 0.f;
                        s0[c] += v[c]; s1[c] += v[c] * cv;
                        if constexpr (E0 == EPI_JOIN_DS) {
                            const float dv = side_vec ? bf2f(as_bf16x8(sided[i][c >> 3])[c & 7]) : bf2f(p.Dm[(long)m * p.lddm + nb + c]);
                            s2[c] += v[c] * dv;
                        }
                    }
                }
            }
        } else {  // EPI_BWD: dz = acc * [relu'(bn(c))];  stats: sum dz, sum dz*c
            if (mok) {
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    if (FULL || nb + c < p.N) {
                        const float cv = side_vec ? bf2f(as_bf16x8(side[SIDE ? i : 0][c >> 3])[c & 7]) : bf2f(p.Cm[(long)m * p.ldcm + nb + c]);
                        const float z = fmaf(cv, msc[c], msh[c]);
                        v[c] = z > 0.f ? v[c] : 0.f;
                        s0[c] += v[c]; s1[c] += v[c] * cv;
                    }
                }
            }
        }
        if (mok) {
            if (E0 == EPI_PLAIN && p.out_f32) {
                float* o = (float*)p.C + (long)m * p.ldc + nb;
#pragma unroll
                for (int c = 0; c < NC; ++c) if (FULL || nb + c < p.N) o[c] = v[c];
            } else {
                bf16* o = (bf16*)p.C + (long)m * p.ldc + nb;
                if (vec_ok) {
#pragma unroll
                    for (int c8 = 0; c8 < NC / 8; ++c8) {
                        bf16x8 y;
#pragma unroll
                        for (int e = 0; e < 8; ++e) y[e] = f2bf(v[c8 * 8 + e]);
                        *(uint4*)(o + c8 * 8) = as_uint4(y);
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < NC; ++c) if (FULL || nb + c < p.N) o[c] = f2bf(v[c]);
                }
            }
        }
    }
    if (E0 != EPI_PLAIN && E0 != EPI_EVAL && p.stat0) {
        // one partial row per workgroup tile: reduce the 16 lanes sharing g, then the WM waves via LDS
        constexpr int NS = E0 == EPI_JOIN_DS ? 3 : 2;
        __syncthreads();
        float* red = (float*)smem;                       // [WM][BN][NS]
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float a = quad16_sum(s0[c]);
            const float b = quad16_sum(s1[c]);
            float d = 0.f;
            if constexpr (E0 == EPI_JOIN_DS) d = quad16_sum(s2[c]);
            if (li == 0 && epi_on) {
                const int col = wn * TN + g * NC + c;
                red[(wm * BN + col) * NS + 0] = a;
                red[(wm * BN + col) * NS + 1] = b;
                if constexpr (E0 == EPI_JOIN_DS) red[(wm * BN + col) * NS + 2] = d;
            }
        }
        __syncthreads();
        if (epi_on && tid < BN && (FULL || n0 + tid < p.N)) {
            float a = 0.f, b = 0.f, d = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) {
                a += red[(w * BN + tid) * NS]; b += red[(w * BN + tid) * NS + 1];
                if constexpr (E0 == EPI_JOIN_DS) d += red[(w * BN + tid) * NS + 2];
            }
            p.stat0[(long)tile_m * p.N + n0 + tid] = a;
            p.stat1[(long)tile_m * p.N + n0 + tid] = b;
            if constexpr (E0 == EPI_JOIN_DS) p.stat2[(long)tile_m * p.N + n0 + tid] = d;
            if constexpr (BM == 96) {
                // the consumers read tuber_gemm_nt_stat_rows(M, N) = ceil(M / 64) rows: the rows this tiling does not produce are zero
                const int tiles_m = (p.M + BM - 1) / BM, rows64 = (p.M + 63) / 64;
                for (int r = tiles_m + tile_m; r < rows64; r += tiles_m) { p.stat0[(long)r * p.N + n0 + tid] = 0.f; p.stat1[(long)r * p.N + n0 + tid] = 0.f; }
            }
        }
    }
}

// every output column of every tile exists and every epilogue tensor can be read / written in 16-byte pieces
static bool nt_full(const GemmNT& p, int bn) {
    return p.N % bn == 0 && !(p.ldc & 7) && !p.out_f32 && (!p.Cm || !(p.ldcm & 7)) && (!p.Ym || !(p.ldym & 7)) && (!p.R || !(p.ldr & 7));
}

template <int BM, int BN, int WM, int WN, int G, int OCC>
static int launch_nt_cfg(const GemmNT& p, int amode, int epi, hipStream_t s) {
    const int tiles = ceil_div(p.M, BM) * ceil_div(p.N, BN);
    const bool full = nt_full(p, BN);
    const size_t lds = 2 * (BM + BN) * 128 + (amode == A_BN_RELU ? (size_t)p.K * 8 : (amode == A_BN_BWD ? (size_t)p.K * 12 : 0));
    dim3 grid(tiles), block(256);
#define LNT(AM, EP)                                                                                                   \
    do {                                                                                                              \
        if (lds > 65536) { /* more than 64 KB of dynamic LDS needs a one-time opt-in per kernel */                    \
            static LdsOptIn opt[2];                                                                                   \
            TUBER_LDS_OPT_IN(opt[0], (gemm_nt_kernel<BM, BN, WM, WN, G, AM, EP, OCC>), 160 * 1024);                   \
            TUBER_LDS_OPT_IN(opt[1], (gemm_nt_kernel<BM, BN, WM, WN, G, AM, EP, OCC, 0, true>), 160 * 1024);          \
        }                                                                                                             \
        if (full) hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WM, WN, G, AM, EP, OCC, 0, true>), grid, block, lds, s, p);         \
        else hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WM, WN, G, AM, EP, OCC>), grid, block, lds, s, p);                \
    } while (0)
    if (amode == A_PLAIN) {
        if (epi == EPI_EVAL) return TUBER_EINVAL;        // the eval output epilogue comes with the BatchNorm + ReLU operand prologue only (conv4)
        if (epi == EPI_PLAIN) LNT(A_PLAIN, EPI_PLAIN);
        else if (epi == EPI_STATS) LNT(A_PLAIN, EPI_STATS);
        else if (epi == EPI_JOIN) LNT(A_PLAIN, EPI_JOIN);
        else if (epi == EPI_JOIN_SR) LNT(A_PLAIN, EPI_JOIN_SR);
        else if (epi == EPI_JOIN_DS) LNT(A_PLAIN, EPI_JOIN_DS);
        else if (IS_JOIN_M(epi)) {                      // the bit-field forms: full tiles only
            if (!full) return TUBER_EINVAL;
            if (epi == EPI_JOIN_M) hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WM, WN, G, A_PLAIN, EPI_JOIN_M, OCC, 0, true>), grid, block, lds, s, p);
            else if (epi == EPI_JOIN_SR_M) hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WM, WN, G, A_PLAIN, EPI_JOIN_SR_M, OCC, 0, true>), grid, block, lds, s, p);
            else hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WM, WN, G, A_PLAIN, EPI_JOIN_DS_M, OCC, 0, true>), grid, block, lds, s, p);
        }
        else LNT(A_PLAIN, EPI_BWD);
    } else {
        if (amode == A_BN_RELU) {
            if (epi == EPI_PLAIN) LNT(A_BN_RELU, EPI_PLAIN);
            else if (epi == EPI_STATS) LNT(A_BN_RELU, EPI_STATS);
            else if (IS_JOIN(epi) || IS_JOIN_M(epi)) return TUBER_EINVAL;
            else if (epi == EPI_EVAL) {
                if (!full) return TUBER_EINVAL;
                hipLaunchKernelGGL((gemm_nt_kernel<BM, BN, WM, WN, G, A_BN_RELU, EPI_EVAL, OCC, 0, true>), grid, block, lds, s, p);
            }
            else LNT(A_BN_RELU, EPI_BWD);
        } else if (amode == A_ADD) {
            if (epi != EPI_PLAIN) return TUBER_EINVAL;
            LNT(A_ADD, EPI_PLAIN);
        } else {
            return TUBER_EINVAL;                          // A_BN_BWD prologue: measured slower than the separate apply kernel, not built
        }
    }
#undef LNT
    TUBER_RETURN_LAUNCH();
}

static int g_nt_wsk96 = 1;        // EXPERIMENT hook (round 5): 0 = 64-row tiles everywhere (tuber_gemm_nt_wsk96_set)
// wave split-K launch (64x64 tiles, plain A): 64 KB of LDS (four private 16 KB images), two workgroups per CU
static bool nt_wsk_96(const GemmNT& p, int epi) {
    // 96-row tiles where the 64-row tiling needs more workgroups than there are CUs and the 96-row one does not (layer3: 352 -> 236,
    // layer4: 352 -> 240); full tiles only, the three epilogues those convs use
    const long t64 = (long)ceil_div(p.M, 64) * ceil_div(p.N, 64), t96 = (long)ceil_div(p.M, 96) * ceil_div(p.N, 64);
    return g_nt_wsk96 && t64 > 256 && t96 <= 256 && nt_full(p, 64) && (epi == EPI_PLAIN || epi == EPI_STATS || epi == EPI_BWD);
}
static int launch_nt_wsk(const GemmNT& p, int epi, hipStream_t s) {
    if (nt_wsk_96(p, epi)) {
        dim3 grid96(ceil_div(p.M, 96) * ceil_div(p.N, 64)), block96(256);
        constexpr size_t lds96 = 4 * (size_t)wsk_image_bytes(96);
#define LWSK96(EP)                                                                                                            \
    do {                                                                                                                      \
        static LdsOptIn opt;                                                                                                  \
        TUBER_LDS_OPT_IN(opt, (gemm_nt_kernel<96, 64, 2, 2, 2, A_PLAIN, EP, 1, 1, true>), lds96);                              \
        hipLaunchKernelGGL((gemm_nt_kernel<96, 64, 2, 2, 2, A_PLAIN, EP, 1, 1, true>), grid96, block96, lds96, s, p);          \
    } while (0)
        if (epi == EPI_PLAIN) LWSK96(EPI_PLAIN);
        else if (epi == EPI_STATS) LWSK96(EPI_STATS);
        else LWSK96(EPI_BWD);
#undef LWSK96
        TUBER_RETURN_LAUNCH();
    }
    const int tiles = ceil_div(p.M, 64) * ceil_div(p.N, 64);
    dim3 grid(tiles), block(256);
    const size_t lds = 4 * 16384;
    const bool full = nt_full(p, 64);
#define LWSK(EP)                                                                                                               \
    do {                                                                                                                       \
        if (full) hipLaunchKernelGGL((gemm_nt_kernel<64, 64, 2, 2, 2, A_PLAIN, EP, 2, 1, true>), grid, block, lds, s, p);       \
        else hipLaunchKernelGGL((gemm_nt_kernel<64, 64, 2, 2, 2, A_PLAIN, EP, 2, 1>), grid, block, lds, s, p);                  \
    } while (0)
    if (epi == EPI_PLAIN) LWSK(EPI_PLAIN);
    else if (epi == EPI_STATS) LWSK(EPI_STATS);
    else if (epi == EPI_JOIN) LWSK(EPI_JOIN);
    else if (epi == EPI_JOIN_SR) LWSK(EPI_JOIN_SR);
    else if (epi == EPI_JOIN_DS) LWSK(EPI_JOIN_DS);
    else LWSK(EPI_BWD);
#undef LWSK
    TUBER_RETURN_LAUNCH();
}
// 96-row tiles on the REGULAR pipeline (round 6): every plain-A shape with many rows.  `scripts/gemm_bench.py nt 0,7,13,23` on the model's shapes with
// M >= 16 896 (profiles/r06_gemm_nt_96_row_tiles.txt): 96 x 64 is at or ahead of the 64 x 64 / 64 x 128 / 128 x 128 choice on every one -- the class branch's
// 16 896-row GEMMs by 5 - 18 % (2048 x 512: 60.2 -> 57.4 us against 128 x 128, 256 x 2048: 35.8 -> 29.2 against 64 x 64, 2048 x 256: 35.7 -> 31.7 against 64 x 128),
// layer2's conv1 forward / conv4 data gradient (44 032 x 128 x 512: 1 376 tiles of 64 rows = a full round of the chip at four per CU and a third of one; 918 at 96 rows)
// by 1 - 5 %.  40 KB of LDS per workgroup: four per CU fill the 160 KB exactly; 115 - 128 VGPRs, no scratch.  Outputs bit-identical (same k order), statistics rows per
// 96 rows (the unused rows of the 64-row layout are written as zero, as the wave-split-K form does).  The layer3 / layer4 long-K shapes (M = 5 632 / 2 816) stay on the
// wave-split-K 96-row form: the regular 96 x 64 kernel wins their isolated microbenchmark (9.1 vs 10.2 us) and LOSES in the step (13.80 vs 13.62 ms, three pairs).
static int g_nt_96 = 1;           // EXPERIMENT hook: 0 = the round-5 tile choice (tuber_gemm_nt_96_set)
static bool nt_use_96(const GemmNT& p, int amode, int epi) {
    return g_nt_96 && amode == A_PLAIN && !p.gather && p.M >= 8192 && nt_full(p, 64) && (epi == EPI_PLAIN || epi == EPI_STATS || epi == EPI_BWD);
}
static int launch_nt_96(const GemmNT& p, int epi, hipStream_t s) {
    dim3 grid(ceil_div(p.M, 96) * ceil_div(p.N, 64)), block(256);
    constexpr size_t lds = 2 * (96 + 64) * 128;
#define L96(EP) hipLaunchKernelGGL((gemm_nt_kernel<96, 64, 2, 2, 2, A_PLAIN, EP, 4, 0, true>), grid, block, lds, s, p)
    if (epi == EPI_PLAIN) L96(EPI_PLAIN);
    else if (epi == EPI_STATS) L96(EPI_STATS);
    else L96(EPI_BWD);
#undef L96
    TUBER_RETURN_LAUNCH();
}
// shapes that take it: plain A, no row gather, at least `min_kt` k-tiles, and few enough 64x64 tiles that they are all resident at once
static bool nt_use_wsk(const GemmNT& p, int amode) {
    constexpr int min_kt = 16;
    if (min_kt <= 0 || amode != A_PLAIN || p.gather || p.out_f32) return false;
    const long tiles = (long)ceil_div(p.M, 64) * ceil_div(p.N, 64);
    return p.K / 64 >= min_kt && tiles <= 512;
}

// tile choice: (cfg 0) 128x128, (1) 128x64, (2) 64x64
static int g_nt_force = -1;      // tuber_gemm_nt_set_cfg (tuning runs / kernel tests only)
static int nt_force_cfg() {
    return g_nt_force;
}
static int nt_pick_cfg(int M, int N, int K) {
    if (nt_force_cfg() >= 0) return nt_force_cfg();       // tuning / experiments only
    // measured on MI355X in isolation (scripts/gemm_bench.py, profiles/r02_gemm_nt_tile_ab.txt): 64-row tiles everywhere except the two
    // FLOP-dense class-branch FFN GEMMs.  64x128 (3 workgroups per CU by LDS, registers budgeted for exactly that) wins the wide
    // short-K shapes (conv4 / dgrad1 of layer1-3, class-branch projections) where re-reading A per 64 columns is what costs;
    // 64x64 with a TWO-tile prefetch (cfg 13: 93-109 VGPRs, no spills at 4 workgroups per CU) the small-N and long-K ones -- the
    // four-tile prefetch of round 1 (cfg 2) spilled in the BN-prologue / masked-epilogue variants and is 5-15 % slower everywhere;
    // 128x128 only where M*N is large enough to fill the chip with 2 workgroups per CU.
    if ((long)M * N >= (1L << 25) && N >= 1024 && K >= 256) return 0;
    return (N >= 256 && K <= 512 && M >= 2048) ? 7 : 13;
}
static int nt_dispatch(const GemmNT& p, int amode, int epi, hipStream_t stream);
static void nt_cfg_dims(int cfg, int* bm, int* wm) {
    if (cfg == 0 || cfg == 21) { *bm = 128; *wm = 2; }
    else if (cfg == 22) { *bm = 128; *wm = 4; }
    else if (cfg == 7 || cfg == 17) { *bm = 64; *wm = 1; }
    else { *bm = 64; *wm = 2; }
}

extern "C" {

// number of partial-statistics rows gemm_nt writes for (M, N): the caller sizes stat0/stat1 as
// [rows][N] floats and hands the same row count to the finalize kernels.
// tile configuration tuber_gemm_nt picks for (M, N): 0 = 128x128, 1 = 128x64, 2 = 64x64 (profiling / tests)
int tuber_gemm_nt_cfg(int M, int N, int K) { return nt_pick_cfg(M, N, K); }

int tuber_gemm_nt_wsk96_set(int on) { g_nt_wsk96 = on; return 0; }
int tuber_gemm_nt_96_set(int on) { g_nt_96 = on; return 0; }
// rows per tile of the wave-split-K form tuber_gemm_nt takes for a plain-A (M, N, K) with a 16-byte addressable output: 0 = not taken, 64 or 96
int tuber_gemm_nt_wsk_tile_rows(int M, int N, int K) {
    GemmNT p{};
    p.M = M; p.N = N; p.K = K; p.ldc = N;
    if (nt_force_cfg() >= 0 || nt_pick_cfg(M, N, K) != 13 || !nt_use_wsk(p, A_PLAIN)) return 0;
    return nt_wsk_96(p, EPI_STATS) ? 96 : 64;
}
int tuber_gemm_nt_set_cfg(int cfg) { g_nt_force = cfg < 0 ? -1 : cfg; return 0; }

int tuber_gemm_nt_stat_rows(int M, int N) {
    int bm, wm;
    nt_cfg_dims(nt_pick_cfg(M, N, 64), &bm, &wm);     // every automatic choice has 64-row tiles
    return ceil_div(M, bm);
}

int tuber_gemm_nt(const void* A, long lda, const void* B, long ldb, void* C, long ldc, int M, int N, int K,
                  int amode, const float* a_scale, const float* a_shift,
                  int gather, int To, int Ho, int Wo, int Ti, int Hi, int Wi, int st, int ss,
                  int epi, const float* bias, const void* R, long ldr, int relu, int out_f32,
                  float* stat0, float* stat1,
                  const void* Cm, long ldcm, const float* m_scale, const float* m_shift,
                  float alpha, float drop_p, const void* seed_ptr, unsigned long long salt,
                  const void* A2, long lda2, const float* a_coef2,
                  hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (K & 63) || (lda & 7) || (ldb & 7)) return TUBER_EINVAL;
    if (amode != A_PLAIN && (!a_scale || !a_shift)) return TUBER_EINVAL;
    if (amode == A_BN_BWD && (!A2 || !a_coef2 || (lda2 & 7) || gather || epi == EPI_STATS)) return TUBER_EINVAL;
    if (amode < 0 || amode > 2) return TUBER_EINVAL;                  // (A_ADD has its own entry point: tuber_gemm_nt_addproj)
    if (epi == EPI_BWD && !Cm) return TUBER_EINVAL;
    if (epi != EPI_PLAIN && out_f32) return TUBER_EINVAL;
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && epi != EPI_PLAIN)) return TUBER_EINVAL;
    GemmNT p;
    p.alpha = alpha;
    p.A2 = (const bf16*)A2; p.lda2 = lda2; p.a_coef2 = a_coef2;
    p.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0); p.drop_inv_keep = dropout_inv_keep(drop_p);
    p.seed_ptr = (const uint64_t*)seed_ptr; p.salt = (uint64_t)salt;
    p.A = (const bf16*)A; p.lda = lda; p.B = (const bf16*)B; p.ldb = ldb; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.a_scale = a_scale; p.a_shift = a_shift;
    p.gather = gather; p.To = To; p.Ho = Ho; p.Wo = Wo; p.Ti = Ti; p.Hi = Hi; p.Wi = Wi; p.st = st; p.ss = ss;
    p.bias = bias; p.R = (const bf16*)R; p.ldr = ldr; p.relu = relu; p.out_f32 = out_f32;
    p.stat0 = stat0; p.stat1 = stat1; p.Cm = (const bf16*)Cm; p.ldcm = ldcm; p.m_scale = m_scale; p.m_shift = m_shift;
    p.Ym = nullptr; p.ldym = 0; p.add_ncols = 0;
    p.Dm = nullptr; p.lddm = 0; p.stat2 = nullptr; p.R32 = nullptr; p.ldr32 = 0; p.C32 = nullptr; p.ldc32 = 0;
    return nt_dispatch(p, amode, epi, stream);
}

// EVAL forward of a bottleneck's tail in ONE launch: conv4 on relu(bn3(c3)) + bn4 + the residual join + ReLU
//   y[M,N] = relu((relu(A * a_scale + a_shift) . B^T) * out_scale[n] + out_shift[n] + R32[m][n])
// written twice: as bf16 (y: the GEMM operand of the next bottleneck) and as fp32 (y32: the residual stream of the eval precision mode).
// Under model.eval() a BatchNorm is a constant affine map (tuber_bn_eval_affine), so -- unlike in training, where bn4 needs the statistics
// of the whole conv output first -- the join can ride in the GEMM epilogue and c4 never exists in HBM; bit-identical to tuber_gemm_nt(amode 1) +
// tuber_block_out_fwd_f32 of an identity block (ir_CSN_152.py:62-64,84-90).  N % 128 == 0 (64 for few rows), 16-byte addressable rows.
int tuber_gemm_nt_bn_out(const void* A, long lda, const float* a_scale, const float* a_shift, const void* B, long ldb,
                         const float* out_scale, const float* out_shift, const float* R32, long ldr32, void* y, long ldy, float* y32, long ldy32,
                         int M, int N, int K, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (K & 63) || (lda & 7) || (ldb & 7) || (ldy & 7) || (ldr32 & 3) || (ldy32 & 3) || (N & 63) ||
        !a_scale || !a_shift || !out_scale || !out_shift || !R32 || !y || !y32)
        return TUBER_EINVAL;
    GemmNT p;
    memset(&p, 0, sizeof p);
    p.alpha = 1.f; p.drop_inv_keep = 1.f;
    p.A = (const bf16*)A; p.lda = lda; p.B = (const bf16*)B; p.ldb = ldb; p.C = y; p.ldc = ldy;
    p.M = M; p.N = N; p.K = K; p.a_scale = a_scale; p.a_shift = a_shift;
    p.m_scale = out_scale; p.m_shift = out_shift; p.R32 = R32; p.ldr32 = ldr32; p.C32 = y32; p.ldc32 = ldy32;
    return nt_dispatch(p, A_BN_RELU, EPI_EVAL, stream);
}

// Packed attention in-projection with the positional embedding folded in (nn.MultiheadAttention's in_proj on with_pos_embed(x, pos),
// models/transformer/transformer.py:150-159,215-240):  C[M,N] = f(A)[M,K] . B[N,K]^T + bias,  f(A) = A + A2 for the output columns
// [0, add_ncols) (the q / k rows of in_proj_weight) and f(A) = A for the rest (the v rows).  add_ncols % 128 == 0.  Replaces one add
// kernel and two GEMM launches; the bf16 sum A + A2 is exactly what the add kernel would have stored.
int tuber_gemm_nt_addproj(const void* A, long lda, const void* A2, long lda2, int add_ncols, const void* B, long ldb, void* C, long ldc,
                          int M, int N, int K, const float* bias, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (K & 63) || (lda & 7) || (lda2 & 7) || (ldb & 7) || !A2 || add_ncols < 0 || (add_ncols & 127)) return TUBER_EINVAL;   // any tile width (64 / 128) divides it
    GemmNT p;
    memset(&p, 0, sizeof p);
    p.alpha = 1.f; p.drop_inv_keep = 1.f;
    p.A = (const bf16*)A; p.lda = lda; p.B = (const bf16*)B; p.ldb = ldb; p.C = C; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K; p.bias = bias;
    p.A2 = (const bf16*)A2; p.lda2 = lda2; p.add_ncols = add_ncols;
    return nt_dispatch(p, A_ADD, EPI_PLAIN, stream);
}

// Conv1 data gradient of one bottleneck FUSED with the join backward of the bottleneck below it (whose output y is this conv's input):
//   dz[M,N] = (A[M,K] . B[N,K]^T + R) * [Y > 0],  stat0 / stat1 rows = partial (sum dz, sum dz * Cm) per 64 output rows
// i.e. tuber_gemm_nt(epi 0, +R) followed by tuber_block_out_bwd on its output, without dx ever reaching HBM
// (autograd of ir_CSN_152.py:72,86-89 at a block boundary).  R may be NULL (projection-shortcut blocks add their own gradient first).
int tuber_gemm_nt_join(const void* A, long lda, const void* B, long ldb, void* dz, long ldc, int M, int N, int K,
                       const void* R, long ldr, const void* Y, long ldy, const void* Cm, long ldcm, float* stat0, float* stat1,
                       hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (K & 63) || (lda & 7) || (ldb & 7) || !Y || !Cm || !stat0 || !stat1) return TUBER_EINVAL;
    GemmNT p;
    memset(&p, 0, sizeof p);
    p.alpha = 1.f; p.drop_inv_keep = 1.f;
    p.A = (const bf16*)A; p.lda = lda; p.B = (const bf16*)B; p.ldb = ldb; p.C = dz; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.R = (const bf16*)R; p.ldr = ldr;
    p.stat0 = stat0; p.stat1 = stat1; p.Cm = (const bf16*)Cm; p.ldcm = ldcm;
    p.Ym = (const bf16*)Y; p.ldym = ldy;
    return nt_dispatch(p, A_PLAIN, EPI_JOIN, stream);
}

// tuber_gemm_nt_join with the ReLU mask of the lower block's output as the bit field tuber_block_out_fwd_mask wrote ([M][N / 8] bytes) instead of y:
// dz = (A . B^T + R) * [bit], statistics as above.  Identical results (the bit IS y > 0); the launch reads M*N/8 bytes instead of 2*M*N.
// N % 128 == 0 (64 for few rows), 16-byte addressable rows.
int tuber_gemm_nt_join_mask(const void* A, long lda, const void* B, long ldb, void* dz, long ldc, int M, int N, int K,
                            const void* R, long ldr, const void* Ymask, const void* Cm, long ldcm, float* stat0, float* stat1,
                            hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (K & 63) || (lda & 7) || (ldb & 7) || (N & 63) || !Ymask || !Cm || !stat0 || !stat1) return TUBER_EINVAL;
    GemmNT p;
    memset(&p, 0, sizeof p);
    p.alpha = 1.f; p.drop_inv_keep = 1.f;
    p.A = (const bf16*)A; p.lda = lda; p.B = (const bf16*)B; p.ldb = ldb; p.C = dz; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.R = (const bf16*)R; p.ldr = ldr;
    p.stat0 = stat0; p.stat1 = stat1; p.Cm = (const bf16*)Cm; p.ldcm = ldcm;
    p.Ymask = (const uint8_t*)Ymask;
    return nt_dispatch(p, A_PLAIN, EPI_JOIN_M, stream);
}

// tuber_gemm_nt_join below a stage's FIRST block (layer2 / layer3 / layer4; layer1's runs inside tuber_conv1_bwd_fused): Cd = the raw output of that
// block's projection shortcut, stat2 receives the rows sum dz*cd of the shortcut BatchNorm's backward (tuber_block_out_bwd's third buffer).
int tuber_gemm_nt_join_ds(const void* A, long lda, const void* B, long ldb, void* dz, long ldc, int M, int N, int K,
                          const void* R, long ldr, const void* Y, long ldy, const void* Cm, long ldcm, const void* Cd, long ldcd,
                          float* stat0, float* stat1, float* stat2, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (K & 63) || (lda & 7) || (ldb & 7) || !Y || !Cm || !Cd || !stat0 || !stat1 || !stat2) return TUBER_EINVAL;
    GemmNT p;
    memset(&p, 0, sizeof p);
    p.alpha = 1.f; p.drop_inv_keep = 1.f;
    p.A = (const bf16*)A; p.lda = lda; p.B = (const bf16*)B; p.ldb = ldb; p.C = dz; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.R = (const bf16*)R; p.ldr = ldr;
    p.stat0 = stat0; p.stat1 = stat1; p.stat2 = stat2; p.Cm = (const bf16*)Cm; p.ldcm = ldcm; p.Dm = (const bf16*)Cd; p.lddm = ldcd;
    p.Ym = (const bf16*)Y; p.ldym = ldy;
    return nt_dispatch(p, A_PLAIN, EPI_JOIN_DS, stream);
}

// tuber_gemm_nt_join where R is the data gradient of a STRIDED projection shortcut (a stage's first block, ir_CSN_152.py:155-161): R has
// one row per sampled position, Rrows = n * To * Ho * Wo, and is added to the output rows (n, t, h, w) with t % st == h % ss == w % ss == 0
// (M = n * Ti * Hi * Wi).  Replaces tuber_gemm_nt + tuber_rows_scatter_add + tuber_block_out_bwd at the stage boundaries (layer1 | layer2: a
// 4-pass elementwise kernel over 178 MB tensors).
int tuber_gemm_nt_join_strided(const void* A, long lda, const void* B, long ldb, void* dz, long ldc, int M, int N, int K,
                               const void* R, long ldr, int To, int Ho, int Wo, int Ti, int Hi, int Wi, int st, int ss,
                               const void* Y, long ldy, const void* Cm, long ldcm, float* stat0, float* stat1, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (K & 63) || (lda & 7) || (ldb & 7) || !Y || !Cm || !stat0 || !stat1 || !R || st < 1 || ss < 1
        || To != (Ti - 1) / st + 1 || Ho != (Hi - 1) / ss + 1 || Wo != (Wi - 1) / ss + 1 || M % (Ti * Hi * Wi)) return TUBER_EINVAL;
    GemmNT p;
    memset(&p, 0, sizeof p);
    p.alpha = 1.f; p.drop_inv_keep = 1.f;
    p.A = (const bf16*)A; p.lda = lda; p.B = (const bf16*)B; p.ldb = ldb; p.C = dz; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.R = (const bf16*)R; p.ldr = ldr;
    p.To = To; p.Ho = Ho; p.Wo = Wo; p.Ti = Ti; p.Hi = Hi; p.Wi = Wi; p.st = st; p.ss = ss;
    p.stat0 = stat0; p.stat1 = stat1; p.Cm = (const bf16*)Cm; p.ldcm = ldcm;
    p.Ym = (const bf16*)Y; p.ldym = ldy;
    return nt_dispatch(p, A_PLAIN, EPI_JOIN_SR, stream);
}

// tuber_gemm_nt_join_ds / tuber_gemm_nt_join_strided with the ReLU mask [y > 0] of the lower block's output read from the bit field of
// tuber_block_out_fwd_mask / tuber_blockout_conv1_fwd ([M][N / 8] bytes) instead of y: identical results.  Full tiles only (N % 128 == 0; 64 for the ds form).
int tuber_gemm_nt_join_ds_mask(const void* A, long lda, const void* B, long ldb, void* dz, long ldc, int M, int N, int K,
                               const void* R, long ldr, const void* Ymask, const void* Cm, long ldcm, const void* Cd, long ldcd,
                               float* stat0, float* stat1, float* stat2, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (K & 63) || (lda & 7) || (ldb & 7) || (N & 63) || !Ymask || !Cm || !Cd || !stat0 || !stat1 || !stat2) return TUBER_EINVAL;
    GemmNT p;
    memset(&p, 0, sizeof p);
    p.alpha = 1.f; p.drop_inv_keep = 1.f;
    p.A = (const bf16*)A; p.lda = lda; p.B = (const bf16*)B; p.ldb = ldb; p.C = dz; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.R = (const bf16*)R; p.ldr = ldr;
    p.stat0 = stat0; p.stat1 = stat1; p.stat2 = stat2; p.Cm = (const bf16*)Cm; p.ldcm = ldcm; p.Dm = (const bf16*)Cd; p.lddm = ldcd;
    p.Ymask = (const uint8_t*)Ymask;
    return nt_dispatch(p, A_PLAIN, EPI_JOIN_DS_M, stream);
}
int tuber_gemm_nt_join_strided_mask(const void* A, long lda, const void* B, long ldb, void* dz, long ldc, int M, int N, int K,
                                    const void* R, long ldr, int To, int Ho, int Wo, int Ti, int Hi, int Wi, int st, int ss,
                                    const void* Ymask, const void* Cm, long ldcm, float* stat0, float* stat1, hipStream_t stream) {
    if (M <= 0 || N <= 0 || K <= 0 || (K & 63) || (lda & 7) || (ldb & 7) || (N & 63) || !Ymask || !Cm || !stat0 || !stat1 || !R || st < 1 || ss < 1
        || To != (Ti - 1) / st + 1 || Ho != (Hi - 1) / ss + 1 || Wo != (Wi - 1) / ss + 1 || M % (Ti * Hi * Wi)) return TUBER_EINVAL;
    GemmNT p;
    memset(&p, 0, sizeof p);
    p.alpha = 1.f; p.drop_inv_keep = 1.f;
    p.A = (const bf16*)A; p.lda = lda; p.B = (const bf16*)B; p.ldb = ldb; p.C = dz; p.ldc = ldc;
    p.M = M; p.N = N; p.K = K;
    p.R = (const bf16*)R; p.ldr = ldr;
    p.To = To; p.Ho = Ho; p.Wo = Wo; p.Ti = Ti; p.Hi = Hi; p.Wi = Wi; p.st = st; p.ss = ss;
    p.stat0 = stat0; p.stat1 = stat1; p.Cm = (const bf16*)Cm; p.ldcm = ldcm;
    p.Ymask = (const uint8_t*)Ymask;
    return nt_dispatch(p, A_PLAIN, EPI_JOIN_SR_M, stream);
}

static int nt_dispatch(const GemmNT& p, int amode, int epi, hipStream_t stream) {
    const int M = p.M, N = p.N, K = p.K;
    int cfg = nt_pick_cfg(M, N, K);
    if (nt_force_cfg() < 0 && cfg == 13 && !IS_JOIN_M(epi) && nt_use_wsk(p, amode)) return launch_nt_wsk(p, epi, stream);
    if ((nt_force_cfg() < 0 && nt_use_96(p, amode, epi)) || (nt_force_cfg() == 23 && amode == A_PLAIN && nt_full(p, 64) && (epi == EPI_PLAIN || epi == EPI_STATS || epi == EPI_BWD))) return launch_nt_96(p, epi, stream);
    if (nt_force_cfg() == 23) cfg = 13;
    if (cfg == 0 && epi != EPI_PLAIN && nt_force_cfg() < 0) cfg = 7;     // statistics rows are per 64 output rows (tuber_gemm_nt_stat_rows)
    if ((epi == EPI_JOIN_DS || epi == EPI_JOIN_DS_M) && cfg == 7 && nt_force_cfg() < 0) cfg = 13;   // three side operands spill the 64x128 tile (21 registers at 3 workgroups / CU)
    switch (cfg) {
        case 0: return launch_nt_cfg<128, 128, 2, 2, 2, 2>(p, amode, epi, stream);    // class-branch FFN (plain epilogue)
        case 7: return launch_nt_cfg<64, 128, 1, 4, 2, 3>(p, amode, epi, stream);
        case 13: return launch_nt_cfg<64, 64, 2, 2, 2, 4>(p, amode, epi, stream);
#ifdef TUBER_AB_VARIANTS
        // the measured-and-rejected tile variants of profiles/r02_gemm_nt_tile_ab.txt (deeper prefetch, round-1 register cap: they
        // spill to scratch) are built only with -DTUBER_AB_VARIANTS (TUBER_AB_VARIANTS=1 python -m tubelet_transformer_amd.build)
        case 12: return launch_nt_cfg<64, 64, 2, 2, 4, 3>(p, amode, epi, stream);     // 64x64, four-tile prefetch, 3 workgroups / CU
        case 17: return launch_nt_cfg<64, 128, 1, 4, 2, 4>(p, amode, epi, stream);    // round-1 register cap (spills)
        case 2: return launch_nt_cfg<64, 64, 2, 2, 4, 4>(p, amode, epi, stream);      // round-1 default (spills in the fused variants)
        case 21: return launch_nt_cfg<128, 64, 2, 2, 2, 3>(p, amode, epi, stream);    // round-4 experiment: half the weight re-reads of the long-K layer3 convs
        case 22: return launch_nt_cfg<128, 64, 4, 1, 2, 3>(p, amode, epi, stream);
#endif
        default: return TUBER_EINVAL;
    }
}

// 1 when tuber_gemm_nt_set_cfg(cfg) names a tile configuration this library was built with
int tuber_gemm_nt_has_cfg(int cfg) {
#ifdef TUBER_AB_VARIANTS
    return cfg == 23 || cfg == 0 || cfg == 7 || cfg == 13 || cfg == 12 || cfg == 17 || cfg == 2 || cfg == 21 || cfg == 22;
#else
    return cfg == 0 || cfg == 7 || cfg == 13 || cfg == 23;
#endif
}

}  // extern "C"

// =====================================================================================
// gemm_tn: dW[N,K] = sum_m G[m,N]^T . f(A)[m,K]   (both operands are m-major in HBM)
// =====================================================================================
// Each workgroup owns one 128(N) x 128(K) output tile and one slab of M; per step it stages
// 64 rows of G and A, transposing 4x4 bf16 blocks in registers so the LDS image is
// [col][m] with m contiguous (the k-contiguous layout the MFMA fragments need), then runs the
// same fragment reads / MFMA as gemm_nt.  Slab partials are fp32 [S][N][K]; a second kernel
// sums the slabs (deterministic; no atomics).
struct GemmTN {
    const bf16* G; long ldg;     // [M, N]
    const bf16* A; long lda;     // [M, K]
    float* P;                    // partials [S][N][K]
    int M, N, K, S, rows_per_slab, accumulate;
    const float* a_scale; const float* a_shift;   // A_BN_RELU on A (per k column)
    const bf16* G2; long ldg2; const float* gA; const float* gB; const float* gC;   // GMODE 1: G := gA[n]*G + gB[n]*G2 + gC[n]
    float* bias_grad;            // gemm_tn2, single slab: dbias[n] += sum_m G[m][n] from the LDS image (no separate column-sum launch)
    int gather; int To, Ho, Wo, Ti, Hi, Wi, st, ss;  // row gather on A (G is dense over output rows)
    int amode;                   // grouped launches: A_PLAIN / A_BN_RELU per problem
    const bf16* A2; long lda2;   // transpose-read kernel: A := A + A2 (bf16 sum) -- the with_pos_embed operand of a packed in-projection
};

// transposed staging: the 64 x 128 tile (m x col) is cut in 4x4 blocks; thread -> block
// (mi = 0..15 along m, ci = 0..31 along col).  Lane mapping keeps 128-byte global segments
// (16 consecutive ci per row) and spreads the transposed 8-byte LDS writes over banks.
template <int AMODE, int NTW>
__device__ __forceinline__ void tn_stage_store(char* dst, const uint2 (&r)[4], int mi, int ci, bool is_a, bool ok,
                                               const float (&sc)[4], const float (&sh)[4]) {
    // r[j] = row (mi*4 + j), cols ci*4 .. ci*4+3
    bf16x4 x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        x[j] = as_bf16x4(r[j]);
        if (AMODE == A_BN_RELU && is_a) {
#pragma unroll
            for (int c = 0; c < 4; ++c) x[j][c] = f2bf(fmaxf(fmaf(bf2f(x[j][c]), sc[c], sh[c]), 0.f));
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        bf16x4 y = {x[0][c], x[1][c], x[2][c], x[3][c]};   // 4 consecutive m for column ci*4+c
        if (!ok) y = bf16x4{(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
        const int col = ci * 4 + c;
        const int chunk = mi >> 1;                      // 16-byte chunk along m (8 per 128-byte row)
        // rows are 128 bytes (64 m); swizzle chunk with the same functions the fragment reads use
        const int sw = is_a ? swz_act(col) : swz_wgt<NTW>(col);
        *(uint2*)(dst + col * 128 + ((chunk ^ sw) << 4) + ((mi & 1) << 3)) = as_uint2(y);
    }
}

// T = output tile edge (128: 2x2 waves of 64x64; 64: 2x2 waves of 32x32 -- 4x more workgroups for small N*K)
template <int AMODE, int T, int GMODE>
__global__ __launch_bounds__(256) void gemm_tn_kernel(GemmTN p) {
    constexpr int BN = T, BKo = T, TM = T / 2, TN = T / 2, MT = TM / 16, NT = TN / 16;
    constexpr int NB = T / 64;                 // 4x4 blocks per thread per operand per 64-row step
    constexpr int STAGE = (BN + BKo) * 128;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_k = (p.K + BKo - 1) / BKo;
    // XCD-aware order: the tiles of one M-slab read the same G / A rows, so they get consecutive logical ids (= one XCD's L2)
    int b = xcd_remap(blockIdx.x, gridDim.x);
    const int slab = b / (tiles_n * tiles_k);
    b -= slab * tiles_n * tiles_k;
    const int tile_n = b % tiles_n, tile_k = b / tiles_n;
    const int n0 = tile_n * BN, k0 = tile_k * BKo;
    const int m_begin = slab * p.rows_per_slab;
    const int m_end = min(p.M, m_begin + p.rows_per_slab);

    // staging: the 64 x T tile (m x col) of each operand is cut in 4x4 blocks (16 along m, T/4 along col); a wave covers
    // 4 mi x 16 ci with lane -> (ci_lo = (l&3) | ((l>>4)<<2), mi_lo = (l>>2)&3): 128-byte global segments per row and
    // 2-way-only LDS write conflicts for the transposed 8-byte stores.
    const int ci_lo = (lane & 3) | ((lane >> 4) << 2);      // 0..15
    const int mi_lo = (lane >> 2) & 3;                       // 0..3
    int mi[NB], ci[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int gidx = wave * NB + i;                      // 0 .. 4*NB-1 block groups of (4 mi x 16 ci)
        mi[i] = (gidx & 3) * 4 + mi_lo;
        ci[i] = (gidx >> 2) * 16 + ci_lo;
    }
    float asc[NB][4], ash[NB][4];
    bool g_col_ok[NB], a_col_ok[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int k = k0 + ci[i] * 4 + c;
            asc[i][c] = (AMODE == A_BN_RELU && k < p.K) ? p.a_scale[k] : 1.f;
            ash[i][c] = (AMODE == A_BN_RELU && k < p.K) ? p.a_shift[k] : 0.f;
        }
        g_col_ok[i] = n0 + ci[i] * 4 < p.N;
        a_col_ok[i] = k0 + ci[i] * 4 < p.K;
    }
    float gca[GMODE ? NB : 1][4], gcb[GMODE ? NB : 1][4], gcc[GMODE ? NB : 1][4];      // BatchNorm-backward apply on the G operand
    if (GMODE) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int n = n0 + ci[i] * 4 + c;
                gca[i][c] = n < p.N ? p.gA[n] : 0.f; gcb[i][c] = n < p.N ? p.gB[n] : 0.f; gcc[i][c] = n < p.N ? p.gC[n] : 0.f;
            }
    }

    // 64-row steps are fetched in groups of GS so their global loads overlap (one HBM latency per group)
    constexpr int GS = 4;
    uint2 rgs[GS][NB][4], ravs[GS][NB][4];
    uint2 rg2s[GMODE ? GS : 1][NB][4];
    bool roks[GS][NB][4];
    auto load_step = [&](int ms, uint2 (&rg)[NB][4], uint2 (&rav)[NB][4], bool (&rok)[NB][4], uint2 (&rg2)[NB][4]) {
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = ms + mi[i] * 4 + j;
                const bool ok = m < m_end;
                rok[i][j] = ok;
                long arow = m;
                if (p.gather && ok) {
                    int w = m % p.Wo; int r = m / p.Wo;
                    int h = r % p.Ho; r /= p.Ho;
                    int t = r % p.To; int n = r / p.To;
                    arow = (((long)n * p.Ti + (long)t * p.st) * p.Hi + (long)h * p.ss) * p.Wi + (long)w * p.ss;
                }
                rg[i][j] = (ok && g_col_ok[i]) ? *(const uint2*)(p.G + (long)m * p.ldg + n0 + ci[i] * 4) : make_uint2(0, 0);
                if (GMODE) rg2[i][j] = (ok && g_col_ok[i]) ? *(const uint2*)(p.G2 + (long)m * p.ldg2 + n0 + ci[i] * 4) : make_uint2(0, 0);
                rav[i][j] = (ok && a_col_ok[i]) ? *(const uint2*)(p.A + arow * p.lda + k0 + ci[i] * 4) : make_uint2(0, 0);
            }
    };
    auto store_step = [&](int buf, const uint2 (&rg)[NB][4], const uint2 (&rav)[NB][4], const bool (&rok)[NB][4], const uint2 (&rg2)[NB][4]) {
        char* sg = smem + buf * STAGE;       // G^T tile: [n][m]  (MFMA A operand -> "weight" swizzle)
        char* sa = sg + BN * 128;            // A^T tile: [k][m]  (MFMA B operand -> "act" swizzle)
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            uint2 g4[4], a4[4];
            bool any_bad = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) { g4[j] = rg[i][j]; a4[j] = rav[i][j]; any_bad |= !rok[i][j]; }
            if (GMODE) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bf16x4 x = as_bf16x4(g4[j]), x2 = as_bf16x4(rg2[i][j]);
                    bf16x4 y;
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        y[c] = rok[i][j] ? f2bf(fmaf(bf2f(x[c]), gca[i][c], fmaf(bf2f(x2[c]), gcb[i][c], gcc[i][c]))) : (bf16)0.f;
                    g4[j] = as_uint2(y);
                }
            }
            const float one[4] = {1.f, 1.f, 1.f, 1.f}, zero[4] = {0.f, 0.f, 0.f, 0.f};
            tn_stage_store<A_PLAIN, NT>(sg, g4, mi[i], ci[i], false, true, one, zero);
            if (AMODE == A_BN_RELU && any_bad) {
                // slab tail: rows beyond m_end must contribute zero (the prologue of a zero row is not zero)
                bf16x4 x[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    x[j] = as_bf16x4(a4[j]);
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        x[j][c] = rok[i][j] ? f2bf(fmaxf(fmaf(bf2f(x[j][c]), asc[i][c], ash[i][c]), 0.f)) : (bf16)0.f;
                    a4[j] = as_uint2(x[j]);
                }
                tn_stage_store<A_PLAIN, NT>(sa, a4, mi[i], ci[i], true, true, one, zero);
            } else {
                tn_stage_store<AMODE, NT>(sa, a4, mi[i], ci[i], true, true, asc[i], ash[i]);
            }
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int g = lane >> 4, li = lane & 15;
    int a_row[MT], w_row[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) a_row[i] = wm * TM + i * 16 + li;                              // k index (A^T rows)
#pragma unroll
    for (int j = 0; j < NT; ++j) w_row[j] = wn * TN + (li >> 2) * (4 * NT) + j * 4 + (li & 3);  // n index (G^T rows)

    // rolling software pipeline over the 64-row steps (like gemm_nt): register set j is re-armed with step +GS as soon as
    // it has been written to LDS, so GS steps of global loads stay in flight behind the MFMA work
    int buf = 0;
#pragma unroll
    for (int j = 0; j < GS; ++j)
        if (m_begin + 64 * j < m_end) load_step(m_begin + 64 * j, rgs[j], ravs[j], roks[j], rg2s[GMODE ? j : 0]);
    for (int ms = m_begin; ms < m_end; ms += 64 * GS) {
#pragma unroll
        for (int j = 0; j < GS; ++j) {
            if (ms + 64 * j >= m_end) continue;
            store_step(buf, rgs[j], ravs[j], roks[j], rg2s[GMODE ? j : 0]);
            if (ms + 64 * (GS + j) < m_end) load_step(ms + 64 * (GS + j), rgs[j], ravs[j], roks[j], rg2s[GMODE ? j : 0]);
            __syncthreads();
            const char* sg = smem + buf * STAGE;
            const char* sa = sg + BN * 128;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int qq = ks * 4 + g;
                bf16x8 xa[MT], wb[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i)
                    xa[i] = as_bf16x8(*(const uint4*)(sa + a_row[i] * 128 + ((qq ^ swz_act(a_row[i])) << 4)));
#pragma unroll
                for (int jj = 0; jj < NT; ++jj)
                    wb[jj] = as_bf16x8(*(const uint4*)(sg + w_row[jj] * 128 + ((qq ^ swz_wgt<NT>(w_row[jj])) << 4)));
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int jj = 0; jj < NT; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb[jj], xa[i], acc[i][jj], 0, 0, 0);
            }
            buf ^= 1;
        }
    }
    // D[i = n_local][j = k_local]: lane holds k = k0 + wm*TM + mt*16 + li, n = n0 + wn*TN + g*(4NT) + (nt*4 + r)
    float* P = p.P + (long)slab * p.N * p.K;
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int k = k0 + wm * TM + i * 16 + li;
        if (k >= p.K) continue;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * TN + g * (4 * NT) + j * 4 + r;
                if (n < p.N) {
                    float* o = P + (long)n * p.K + k;
                    *o = (p.S == 1 && p.accumulate) ? *o + acc[i][j][r] : acc[i][j][r];
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// gemm_tn via the gfx950 LDS transpose read (ds_read_b64_tr_b16): the 64-row steps of G and A are parked in LDS ROW-MAJOR with
// plain 16-byte stores (exactly as they arrive from HBM), and the m-contiguous MFMA fragments are read back transposed by the
// hardware: for a 16-lane group, lane i supplies the address of chunk i (row i/4, 4 columns (i%4)*4..) of a [4 m][16 cols] block
// and receives column i of it, i.e. 4 consecutive m for its own output column.  Two such reads make one MFMA operand.
// The register-transposing kernel above issues 17 VALU instructions per MFMA (PMC SQ_INSTS_VALU / SQ_INSTS_MFMA); this one ~4.
// Used for 64 x 64 output tiles with 8-element aligned N, K, ld; otherwise (heads with N = 3/4/80, GMODE, 128 tiles) the kernel
// above runs.
// ---------------------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
#define TNP 64               // LDS row pitch in bf16: 128 B, no padding; 32-byte units XOR-swizzled by tn2_key(row)
// Bank map (64 banks x 4 B = 256 B per cycle).  16-byte stores: 16 lanes cover rows 2j, 2j+1 = one full 256 B line for any permutation
// inside a row.  Transposed reads (ds_read_b64_tr_b16): a 32-lane pass reads rows {r..r+3} and {r+8..r+11} (r % 8 in {0, 4}), one 32-B
// unit each; rows r, r+1 sit in different halves of a line, and the unit index is XORed with a 2-bit key that differs between the row
// pairs (r, r+1), (r+2, r+3), (r+8, r+9), (r+10, r+11), so the eight 32-B pieces cover all 64 banks exactly once.
__device__ __forceinline__ int tn2_key(int row) { return ((row >> 1) & 1) | ((row >> 2) & 2); }
__device__ __forceinline__ int tn2_off(int row, int col) { return row * TNP + ((((col >> 4) ^ tn2_key(row)) << 4) | (col & 15)); }

__device__ __forceinline__ bf16x8 tn2_frag(const bf16* img, int m0, int col0, int li) {
    // rows m0 .. m0+7 (two [4][16] blocks), columns col0 .. col0+15; lane li of the group gets 8 consecutive m of column col0+li
    const bf16* p = img + tn2_off(m0 + (li >> 2), col0 + (li & 3) * 4);          // row + 4 has the same key (m0 % 8 == 0)
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * TNP));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// AMODE = A_PLAIN / A_BN_RELU at compile time, or -1: read p.amode at run time (grouped launches mix both; the branch is uniform).
// bid / nblocks: this workgroup's index inside the problem's own grid (a grouped launch concatenates several grids).
template <int AMODE>
__device__ __forceinline__ void gemm_tn2_body(const GemmTN& p, int bid, int nblocks) {
    constexpr int T = 64, TM = 32, TN = 32, MT = 2, NT = 2;
    constexpr int IMG = 64 * TNP;                       // one operand image (elements)
    __shared__ __attribute__((aligned(16))) bf16 smem[2][2][IMG];     // [buf][G | A][64 m][TNP]
    const bool bn_relu = AMODE < 0 ? p.amode == A_BN_RELU : AMODE == A_BN_RELU;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, g = lane >> 4;
    const int tiles_n = (p.N + T - 1) / T, tiles_k = (p.K + T - 1) / T;
    int b = xcd_remap(bid, nblocks);
    const int slab = b / (tiles_n * tiles_k);
    b -= slab * tiles_n * tiles_k;
    const int tile_n = b % tiles_n, tile_k = b / tiles_n;
    const int n0 = tile_n * T, k0 = tile_k * T;
    const int m_begin = slab * p.rows_per_slab;
    const int m_end = min(p.M, m_begin + p.rows_per_slab);

    // staging: thread -> 16-byte chunk c (8 columns) of rows r and r + 32 of the 64-row step
    const int c = tid & 7, r = tid >> 3;
    const bool g_col_ok = n0 + c * 8 < p.N, a_col_ok = k0 + c * 8 < p.K;
    float asc[8], ash[8];
    if (bn_relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + c * 8 + e;
            asc[e] = k < p.K ? p.a_scale[k] : 1.f;
            ash[e] = k < p.K ? p.a_shift[k] : 0.f;
        }
    }
    const bool do_bias = p.bias_grad != nullptr && tile_k == 0;
    float bsum = 0.f;
    constexpr int GS = 4;
    uint4 rg[GS][2], ra[GS][2];
    bool rok[GS][2];
    // full 64-row steps of dense operands (every step of the backbone shapes: M and the slab length are multiples of 64) skip the
    // per-row predicates and form their addresses from per-thread base pointers + a step offset that is uniform (scalar unit)
    const bool dense = !p.gather && g_col_ok && a_col_ok && !p.A2;
    const bf16* gb0 = p.G + (long)(m_begin + r) * p.ldg + n0 + c * 8;
    const bf16* ab0 = p.A + (long)(m_begin + r) * p.lda + k0 + c * 8;
    const long g32 = 32 * p.ldg, a32 = 32 * p.lda;
    auto add8 = [](uint4 a, uint4 b) {                     // bf16(a + b) per element: the sum the stand-alone add kernel would store
        const bf16x8 x = as_bf16x8(a), y = as_bf16x8(b);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(x[e]) + bf2f(y[e]));
        return as_uint4(o);
    };
    auto load_step = [&](int ms, uint4 (&xg)[2], uint4 (&xa)[2], bool (&ok)[2]) {
        if (dense && ms + 64 <= m_end) {
            const long so = (long)(ms - m_begin);
            const bf16* gq = gb0 + so * p.ldg;
            const bf16* aq = ab0 + so * p.lda;
            xg[0] = *(const uint4*)gq; xg[1] = *(const uint4*)(gq + g32);
            xa[0] = *(const uint4*)aq; xa[1] = *(const uint4*)(aq + a32);
            ok[0] = ok[1] = true;
            return;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int m = ms + r + 32 * h;
            ok[h] = m < m_end;
            long arow = m;
            if (p.gather && ok[h]) {
                int w = m % p.Wo; int q = m / p.Wo;
                int hh = q % p.Ho; q /= p.Ho;
                int t = q % p.To; int n = q / p.To;
                arow = (((long)n * p.Ti + (long)t * p.st) * p.Hi + (long)hh * p.ss) * p.Wi + (long)w * p.ss;
            }
            xg[h] = (ok[h] && g_col_ok) ? *(const uint4*)(p.G + (long)m * p.ldg + n0 + c * 8) : make_uint4(0, 0, 0, 0);
            xa[h] = (ok[h] && a_col_ok) ? *(const uint4*)(p.A + arow * p.lda + k0 + c * 8) : make_uint4(0, 0, 0, 0);
            if (p.A2 && ok[h] && a_col_ok) xa[h] = add8(xa[h], *(const uint4*)(p.A2 + (long)m * p.lda2 + k0 + c * 8));
        }
    };
    auto store_step = [&](int buf, const uint4 (&xg)[2], const uint4 (&xa)[2], const bool (&ok)[2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            uint4 av = xa[h];
            if (bn_relu) {
                const bf16x8 x = as_bf16x8(av);
                bf16x8 y;
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = f2bf(fmaxf(fmaf(bf2f(x[e]), asc[e], ash[e]), 0.f));
                av = (ok[h] && a_col_ok) ? as_uint4(y) : make_uint4(0, 0, 0, 0);      // rows / columns outside contribute zero
            }
            *(uint4*)&smem[buf][0][tn2_off(r + 32 * h, c * 8)] = xg[h];
            *(uint4*)&smem[buf][1][tn2_off(r + 32 * h, c * 8)] = av;
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int buf = 0;
#pragma unroll
    for (int j = 0; j < GS; ++j)
        if (m_begin + 64 * j < m_end) load_step(m_begin + 64 * j, rg[j], ra[j], rok[j]);
    for (int ms = m_begin; ms < m_end; ms += 64 * GS) {
#pragma unroll
        for (int j = 0; j < GS; ++j) {
            if (ms + 64 * j >= m_end) continue;
            store_step(buf, rg[j], ra[j], rok[j]);
            if (ms + 64 * (GS + j) < m_end) load_step(ms + 64 * (GS + j), rg[j], ra[j], rok[j]);
            __syncthreads();
            const bf16* gi = smem[buf][0];
            const bf16* ai = smem[buf][1];
            if (do_bias) {                          // column sums of this step's G rows: thread = (column, 16-row part)
#pragma unroll
                for (int mm = 0; mm < 16; ++mm) bsum += bf2f(gi[tn2_off((tid >> 6) * 16 + mm, tid & 63)]);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int m0 = ks * 32 + g * 8;
                bf16x8 fn[NT], fk[MT];
#pragma unroll
                for (int j2 = 0; j2 < NT; ++j2) fn[j2] = tn2_frag(gi, m0, wn * TN + j2 * 16, li);      // rows of D: n
#pragma unroll
                for (int i2 = 0; i2 < MT; ++i2) fk[i2] = tn2_frag(ai, m0, wm * TM + i2 * 16, li);      // cols of D: k
#pragma unroll
                for (int i2 = 0; i2 < MT; ++i2)
#pragma unroll
                    for (int j2 = 0; j2 < NT; ++j2)
                        acc[i2][j2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[j2], fk[i2], acc[i2][j2], 0, 0, 0);
            }
            buf ^= 1;
        }
    }
    // D[row n][col k]: lane holds k = wm*TM + i*16 + li, n = wn*TN + j*16 + g*4 + r.  The tile goes through LDS so that
    // it leaves as 16-byte stores, 256 contiguous bytes per 16 lanes (direct stores would be 64-byte fragments)
    float* P = p.P + (long)slab * p.N * p.K;
    __syncthreads();
    float* ot = (float*)&smem[0][0][0];                 // [64 n][64 k + 4] fp32 = 17 KB of the 36 KB staging area
    if (do_bias) {
        float* br = (float*)&smem[1][1][0];             // far end of the staging area (the tile below uses the first 17 KB)
        br[tid] = bsum;
        __syncthreads();
        if (tid < 64 && n0 + tid < p.N) {
            const float v = (br[tid] + br[64 + tid]) + (br[128 + tid] + br[192 + tid]);
            if (p.S == 1) p.bias_grad[n0 + tid] += v;                       // single slab: straight into the gradient
            else p.bias_grad[(long)slab * p.N + n0 + tid] = v;              // bias_grad is a [S][N] partial, reduced by the caller
        }
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
                ot[(wn * TN + j * 16 + g * 4 + rr) * 68 + wm * TM + i * 16 + li] = acc[i][j][rr];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = tid + 256 * q;
        const int n = n0 + (idx >> 4), k = k0 + (idx & 15) * 4;
        if (n < p.N && k < p.K) {                       // K % 8 == 0: whole float4 inside
            float4 v = *(const float4*)&ot[(idx >> 4) * 68 + (idx & 15) * 4];
            float4* o = (float4*)(P + (long)n * p.K + k);
            if (p.S == 1 && p.accumulate) { const float4 c = *o; v.x += c.x; v.y += c.y; v.z += c.z; v.w += c.w; }
            *o = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// gemm_tn3: the same transpose-read weight-gradient GEMM on 128 x 128 output tiles (round 3).  The 64 x 64 kernel moves
// 16 KB of operands through L2 / LDS per 64-row step for 64 x 64 x 64 MACs; over a grouped launch of eight layer3 problems that is
// 720 MB of L2 -> LDS traffic for 128 MB of operands (every A panel read 16x, every G panel 4x) at 11 % MFMA utilisation -- it
// is bound by operand movement, not by HBM or MFMA.  Here a wave owns 64 x 64 of the tile (16 accumulator blocks): one LDS
// fragment feeds 4 MFMAs instead of 2, a 64-row step carries 4x the MACs for 2x the bytes.  LDS: 2 buffers x 2 operands x
// [64 m][128 cols] bf16 = 64 KB, two workgroups per CU.  Rows are 256 B = one full sweep of the 64 banks, so the 32-byte units are
// XOR-swizzled by a 3-bit row key that differs over the eight rows {r..r+3, r+8..r+11} a 32-lane pass of ds_read_b64_tr_b16
// touches (conflict-free); 16-byte stores cover whole rows.  Taken for N, K multiples of 128 with N*K >= 2^17.
// ---------------------------------------------------------------------------------------------------------------------
#define TNP3 128
__device__ __forceinline__ int tn3_key(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
__device__ __forceinline__ int tn3_off(int row, int col) { return row * TNP3 + ((((col >> 4) ^ tn3_key(row)) << 4) | (col & 15)); }
__device__ __forceinline__ bf16x8 tn3_frag(const bf16* img, int m0, int col0, int li) {
    const bf16* p = img + tn3_off(m0 + (li >> 2), col0 + (li & 3) * 4);          // row + 4 has the same key (m0 % 8 == 0, li >> 2 < 4)
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p + 4 * TNP3));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int AMODE>
__device__ __forceinline__ void gemm_tn3_body(const GemmTN& p, int bid, int nblocks) {
    constexpr int T = 128, TW = 64, FT = 4;             // wave tile 64 x 64 = 4 x 4 MFMA blocks
    constexpr int IMG = 64 * TNP3;
    __shared__ __attribute__((aligned(16))) bf16 smem[2][2][IMG];     // [buf][G | A][64 m][128]
    const bool bn_relu = AMODE < 0 ? p.amode == A_BN_RELU : AMODE == A_BN_RELU;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, g = lane >> 4;
    const int tiles_n = (p.N + T - 1) / T, tiles_k = (p.K + T - 1) / T;
    int b = xcd_remap(bid, nblocks);
    const int slab = b / (tiles_n * tiles_k);
    b -= slab * tiles_n * tiles_k;
    const int tile_n = b % tiles_n, tile_k = b / tiles_n;
    const int n0 = tile_n * T, k0 = tile_k * T;
    const int m_begin = slab * p.rows_per_slab;
    const int m_end = min(p.M, m_begin + p.rows_per_slab);

    // staging: thread -> 16-byte chunk c (8 columns) of rows r, r + 16, r + 32, r + 48 of the 64-row step
    const int c = tid & 15, r = tid >> 4;
    const bool g_col_ok = n0 + c * 8 < p.N, a_col_ok = k0 + c * 8 < p.K;
    float asc[8], ash[8];
    if (bn_relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = k0 + c * 8 + e;
            asc[e] = k < p.K ? p.a_scale[k] : 1.f;
            ash[e] = k < p.K ? p.a_shift[k] : 0.f;
        }
    }
    const bool do_bias = p.bias_grad != nullptr && tile_k == 0;
    float bsum = 0.f;
    constexpr int GS = 2;
    uint4 rg[GS][4], ra[GS][4];
    bool rok[GS][4];
    const bool dense = !p.gather && g_col_ok && a_col_ok && !p.A2;
    const bf16* gb0 = p.G + (long)(m_begin + r) * p.ldg + n0 + c * 8;
    const bf16* ab0 = p.A + (long)(m_begin + r) * p.lda + k0 + c * 8;
    const long g16 = 16 * p.ldg, a16 = 16 * p.lda;
    auto add8 = [](uint4 a, uint4 b) {
        const bf16x8 x = as_bf16x8(a), y = as_bf16x8(b);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = f2bf(bf2f(x[e]) + bf2f(y[e]));
        return as_uint4(o);
    };
    auto load_step = [&](int ms, uint4 (&xg)[4], uint4 (&xa)[4], bool (&ok)[4]) {
        if (dense && ms + 64 <= m_end) {
            const long so = (long)(ms - m_begin);
            const bf16* gq = gb0 + so * p.ldg;
            const bf16* aq = ab0 + so * p.lda;
#pragma unroll
            for (int h = 0; h < 4; ++h) { xg[h] = *(const uint4*)(gq + h * g16); xa[h] = *(const uint4*)(aq + h * a16); ok[h] = true; }
            return;
        }
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const int m = ms + r + 16 * h;
            ok[h] = m < m_end;
            long arow = m;
            if (p.gather && ok[h]) {
                int w = m % p.Wo; int q = m / p.Wo;
                int hh = q % p.Ho; q /= p.Ho;
                int t = q % p.To; int n = q / p.To;
                arow = (((long)n * p.Ti + (long)t * p.st) * p.Hi + (long)hh * p.ss) * p.Wi + (long)w * p.ss;
            }
            xg[h] = (ok[h] && g_col_ok) ? *(const uint4*)(p.G + (long)m * p.ldg + n0 + c * 8) : make_uint4(0, 0, 0, 0);
            xa[h] = (ok[h] && a_col_ok) ? *(const uint4*)(p.A + arow * p.lda + k0 + c * 8) : make_uint4(0, 0, 0, 0);
            if (p.A2 && ok[h] && a_col_ok) xa[h] = add8(xa[h], *(const uint4*)(p.A2 + (long)m * p.lda2 + k0 + c * 8));
        }
    };
    auto store_step = [&](int buf, const uint4 (&xg)[4], const uint4 (&xa)[4], const bool (&ok)[4]) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            uint4 av = xa[h];
            if (bn_relu) {
                const bf16x8 x = as_bf16x8(av);
                bf16x8 y;
#pragma unroll
                for (int e = 0; e < 8; ++e) y[e] = f2bf(fmaxf(fmaf(bf2f(x[e]), asc[e], ash[e]), 0.f));
                av = (ok[h] && a_col_ok) ? as_uint4(y) : make_uint4(0, 0, 0, 0);
            }
            *(uint4*)&smem[buf][0][tn3_off(r + 16 * h, c * 8)] = xg[h];
            *(uint4*)&smem[buf][1][tn3_off(r + 16 * h, c * 8)] = av;
        }
    };

    f32x4 acc[FT][FT];
#pragma unroll
    for (int i = 0; i < FT; ++i)
#pragma unroll
        for (int j = 0; j < FT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    int buf = 0;
#pragma unroll
    for (int j = 0; j < GS; ++j)
        if (m_begin + 64 * j < m_end) load_step(m_begin + 64 * j, rg[j], ra[j], rok[j]);
    for (int ms = m_begin; ms < m_end; ms += 64 * GS) {
#pragma unroll
        for (int j = 0; j < GS; ++j) {
            if (ms + 64 * j >= m_end) continue;
            store_step(buf, rg[j], ra[j], rok[j]);
            if (ms + 64 * (GS + j) < m_end) load_step(ms + 64 * (GS + j), rg[j], ra[j], rok[j]);
            __syncthreads();
            const bf16* gi = smem[buf][0];
            const bf16* ai = smem[buf][1];
            if (do_bias) {                          // column sums of this step's G rows: thread = (column, 32-row half)
#pragma unroll
                for (int mm = 0; mm < 32; ++mm) bsum += bf2f(gi[tn3_off((tid >> 7) * 32 + mm, tid & 127)]);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int m0 = ks * 32 + g * 8;
                bf16x8 fn[FT], fk[FT];
#pragma unroll
                for (int j2 = 0; j2 < FT; ++j2) fn[j2] = tn3_frag(gi, m0, wn * TW + j2 * 16, li);      // rows of D: n
#pragma unroll
                for (int i2 = 0; i2 < FT; ++i2) fk[i2] = tn3_frag(ai, m0, wm * TW + i2 * 16, li);      // cols of D: k
#pragma unroll
                for (int i2 = 0; i2 < FT; ++i2)
#pragma unroll
                    for (int j2 = 0; j2 < FT; ++j2)
                        acc[i2][j2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[j2], fk[i2], acc[i2][j2], 0, 0, 0);
            }
            buf ^= 1;
        }
    }
    // D[row n][col k]: lane holds k = wm*64 + i*16 + li, n = wn*64 + j*16 + g*4 + r.  The tile leaves through LDS in two halves of
    // 64 n-rows (16-byte stores, 512 contiguous bytes per output row)
    float* P = p.P + (long)slab * p.N * p.K;
    __syncthreads();
    float* ot = (float*)&smem[0][0][0];                 // [64 n][128 k + 4] fp32 = 33 KB of the 64 KB staging area
    if (do_bias) {
        float* br = ot + 64 * 132;                      // behind the output half-tile
        br[tid] = bsum;
        __syncthreads();
        if (tid < 128 && n0 + tid < p.N) {
            const float v = br[tid] + br[128 + tid];
            if (p.S == 1) p.bias_grad[n0 + tid] += v;
            else p.bias_grad[(long)slab * p.N + n0 + tid] = v;
        }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (h) __syncthreads();
        if (wn == h) {
#pragma unroll
            for (int i = 0; i < FT; ++i)
#pragma unroll
                for (int j = 0; j < FT; ++j)
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
                        ot[(j * 16 + g * 4 + rr) * 132 + wm * TW + i * 16 + li] = acc[i][j][rr];
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int idx = tid + 256 * q;
            const int n = n0 + h * 64 + (idx >> 5), k = k0 + (idx & 31) * 4;
            if (n < p.N && k < p.K) {
                float4 v = *(const float4*)&ot[(idx >> 5) * 132 + (idx & 31) * 4];
                float4* o = (float4*)(P + (long)n * p.K + k);
                if (p.S == 1 && p.accumulate) { const float4 cc = *o; v.x += cc.x; v.y += cc.y; v.z += cc.z; v.w += cc.w; }
                *o = v;
            }
        }
    }
}

template <int AMODE>
__global__ __launch_bounds__(256, 2) void gemm_tn3_kernel(GemmTN p) { gemm_tn3_body<AMODE>(p, blockIdx.x, gridDim.x); }

template <int AMODE>
__global__ __launch_bounds__(256, 3) void gemm_tn2_kernel(GemmTN p) { gemm_tn2_body<AMODE>(p, blockIdx.x, gridDim.x); }

// Several weight-gradient GEMMs in ONE launch: the conv weight gradients of a bottleneck (conv4, conv1, down_sample) feed nothing
// until the optimizer, so they are queued and launched together -- one launch gap instead of 2-3 per bottleneck, and for the
// short-M layer3 / layer4 shapes (512 workgroups of a few 64-row steps each) enough independent workgroups to fill 256 CUs.
// The argument blocks travel BY VALUE in the kernel argument segment (a captured hipGraph bakes them in like any other launch).
#define TN_GROUP_MAX 16
struct GemmTNGroup { GemmTN p[TN_GROUP_MAX]; int begin[TN_GROUP_MAX + 1]; int n; };
__global__ __launch_bounds__(256, 3) void gemm_tn2_group_kernel(GemmTNGroup g) {
    int e = 0;
#pragma unroll
    for (int i = 1; i < TN_GROUP_MAX; ++i)
        if (i < g.n && (int)blockIdx.x >= g.begin[i]) e = i;
    gemm_tn2_body<-1>(g.p[e], (int)blockIdx.x - g.begin[e], g.begin[e + 1] - g.begin[e]);
}

__global__ __launch_bounds__(256, 2) void gemm_tn3_group_kernel(GemmTNGroup g) {
    int e = 0;
#pragma unroll
    for (int i = 1; i < TN_GROUP_MAX; ++i)
        if (i < g.n && (int)blockIdx.x >= g.begin[i]) e = i;
    gemm_tn3_body<-1>(g.p[e], (int)blockIdx.x - g.begin[e], g.begin[e + 1] - g.begin[e]);
}

// out[j] (+)= sum_s P[s][j], few slabs: one thread per element
__global__ void reduce_slabs_flat_kernel(const float* __restrict__ P, float* __restrict__ out, long n, int S, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float a = 0.f;
    for (int s = 0; s < S; ++s) a += P[(long)s * n + i];
    out[i] = accumulate ? out[i] + a : a;
}

// out[j] (+)= sum_s P[s][j]: 32 slab-groups x 32 columns per block, LDS tree over the slab groups
__global__ __launch_bounds__(1024) void reduce_slabs_kernel(const float* __restrict__ P, float* __restrict__ out, long n, int S, int accumulate) {
    __shared__ float red[32][33];
    const int cl = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const long j = (long)blockIdx.x * 32 + cl;
    float a = 0.f;
    if (j < n) for (int s = rg; s < S; s += 32) a += P[(long)s * n + j];
    red[rg][cl] = a;
    __syncthreads();
    if (rg == 0 && j < n) {
        a = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) a += red[i][cl];
        out[j] = accumulate ? out[j] + a : a;
    }
}

extern "C" {

// Tile / slab choice of gemm_tn (shapes with odd N / K): 128x128 tiles when they alone give >= 128 workgroups, else 64x64; slabs over M fill the
// chip to ~256 workgroups but never more than 8 (bounds the fp32 slab traffic to <= the size of the operands) and
// at least 256 rows each.
// (shapes the transpose-read kernels take -- N and K multiples of 8 -- always walk 64 x 64 tiles here, or 128 x 128 ones by tn_big below:
// layer4's down-sample weight gradient, 2048 x 1024 over M = 704, took 71.6 us on the in-register-transpose 128-tile kernel against
// ~14 us for the same work as 512 tiles of gemm_tn2)
static int tn_tile(int N, int K) { return !((N | K) & 7) ? 64 : (long)ceil_div(N, 128) * ceil_div(K, 128) >= 128 ? 128 : 64; }
// 128 x 128 transpose-read tiles (gemm_tn3): layer3 / layer4 convs, FFN and class-branch linears, packed in-projections
static const int g_tn_big = 1;
// Measured (scripts/gemm_bench.py tngroup, round 3): eight layer3 problems 68.2 -> 50.9 us, six layer4 ones 87.5 -> 82.6 us; but the
// long-M class-branch pair (M = 16896) 132 -> 224 us WITH THE SLAB COUNT OF THE SMALL TILES (2 and 1: few tiles x few slabs underfill
// the chip at two workgroups per CU; see below) and the short-M encoder FFNs (M = 704: 11 steps) 13.5 -> 20.8 us.
static bool tn_big(int M, int N, int K) {
    // long M (the class-branch FFN pair, M = 16896, 256 x 2048 and 2048 x 512): with enough slabs to fill the chip (2112 rows = 33 steps
    // each: 8 slabs) the big tiles win there too -- 130.8 -> 81.8 us for the pair (650 TFLOP/s), 86.8 / 92.6 / 87.4 us with 12 / 6 / 4 slabs
    constexpr int max_m = 32768;
    if (M > 8192 && (long)N * K < (1L << 19)) return false;
    return g_tn_big && M >= 2048 && M <= max_m && !((N | K) & 127) && (long)N * K >= (1L << 17) && tn_tile(N, K) == 64;
}
static int tn_slabs_wanted(int M, int N, int K) {
    const int T = tn_big(M, N, K) ? 128 : tn_tile(N, K);
    const long tiles = (long)ceil_div(N, T) * ceil_div(K, T);
    constexpr int target = 256;
    // workgroups aimed for per GEMM: 256 since the weight gradients travel in grouped launches (tuber_gemm_tn_group: 2-8 GEMMs share the
    // chip, so each needs fewer slabs to fill it: 18.87 -> 18.65 ms/step against 512, and half the slab traffic; 128 loses again)
    // big tiles: ~22 steps of 64 rows per workgroup measured best on both backbone shapes (layer3 M = 5632: 4 slabs 51 us per eight
    // problems against 56 us with 8 and 75 with 15; layer4 M = 2816: 2 slabs 72 us per six against 82 us with 1) -- fewer steps do not
    // amortise the 64 KB prologue / 64 KB fp32 epilogue of a tile, more leave the chip underfilled
    long S;
    if (tn_big(M, N, K)) {
        // round 5, groups of 16 problems: layer3 (M = 5632) as 2 slabs of 2816 rows -- 86.1 us per sixteen against 94.1 with 4 slabs
        // (112.6 with 1), half the fp32 partials written and re-read; step 14.255 -> 14.22 ms (two same-box pairs).  layer4 (M = 2816)
        // stays at 2 slabs of 1408 (71 us per six against 80 with 1)
        const int big_rows = M > 4096 ? 2816 : 1408; constexpr int long_rows = 2112;
        const int rows = M > 8192 ? long_rows : big_rows;
        S = (M + rows / 2) / rows;
    } else {
        S = (target + tiles - 1) / tiles;
    }
    // bound the fp32 slab traffic (S*N*K*4 B written + read) by the size of the operands (2*M*(N+K) B);
    // tiny outputs (<= 16 tiles: <= 256 KB per slab) may split deeply
    long cap = tiles <= 16 ? 256 : (long)M * (N + K) / (2L * N * K);
    if (cap < 1) cap = 1;
    if (S > cap) S = cap;
    const long maxS = ceil_div(M, 256);
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    return (int)S;
}
static int tn_rows_per_slab(int M, int N, int K) { return ceil_div(ceil_div(M, tn_slabs_wanted(M, N, K)), 64) * 64; }
// output tile edge the transpose-read kernels use for (M, N, K): 128 (gemm_tn3) or 64 (gemm_tn2); profiling / bench bookkeeping
int tuber_gemm_tn_tile(int M, int N, int K) { return tn_big(M, N, K) ? 128 : 64; }
// slabs actually written (rows per slab are rounded up to 64, so this can be fewer than the split aimed for)
int tuber_gemm_tn_slabs(int M, int N, int K) { return ceil_div(M, tn_rows_per_slab(M, N, K)); }

// Can tuber_gemm_tn also produce the bias gradient sum_m G[m][n] (transpose-read kernel, from the LDS image of G)?
//   0: no;  1: yes, single slab: bias_grad[n] is accumulated directly;
//   2: yes, S = tuber_gemm_tn_slabs > 1: bias_grad must point to S*N floats and receives one partial row per slab (reduce them with
//      tuber_reduce_rows(bias_grad, dbias, S, N, 1) or a tuber_multi_reduce entry).
int tuber_gemm_tn_fuses_bias(int M, int N, int K, long ldg, long lda) {
    if (!(tn_tile(N, K) == 64 && !((N | K | ldg | lda) & 7))) return 0;
    return tuber_gemm_tn_slabs(M, N, K) == 1 ? 1 : 2;
}

int tuber_gemm_tn(const void* G, long ldg, const void* A, long lda, float* partial, float* out, int accumulate,
                  int M, int N, int K, int amode, const float* a_scale, const float* a_shift,
                  int gather, int To, int Ho, int Wo, int Ti, int Hi, int Wi, int st, int ss,
                  const void* G2, long ldg2, const float* gA, const float* gB, const float* gC,
                  float* bias_grad, hipStream_t stream) {
    // N / K need not be multiples of 4, but G / A must be readable up to ceil4(N) / ceil4(K) columns (padded ld)
    if (M <= 0 || N <= 0 || K <= 0 || (ldg & 3) || (lda & 3) || ldg < ((N + 3) & ~3) || lda < ((K + 3) & ~3)) return TUBER_EINVAL;
    GemmTN p;
    p.G = (const bf16*)G; p.ldg = ldg; p.A = (const bf16*)A; p.lda = lda;
    p.M = M; p.N = N; p.K = K;
    p.rows_per_slab = tn_rows_per_slab(M, N, K);
    p.S = tuber_gemm_tn_slabs(M, N, K);
    p.accumulate = accumulate;
    p.P = p.S == 1 ? out : partial;            // a single slab writes (or accumulates into) the gradient directly
    p.a_scale = a_scale; p.a_shift = a_shift;
    const int gmode = 0;
    if (G2) return TUBER_EINVAL;               // BatchNorm-backward apply on the G operand (GMODE 1): measured slower than the apply kernel, not built
    p.G2 = (const bf16*)G2; p.ldg2 = ldg2; p.gA = gA; p.gB = gB; p.gC = gC;
    p.gather = gather; p.To = To; p.Ho = Ho; p.Wo = Wo; p.Ti = Ti; p.Hi = Hi; p.Wi = Wi; p.st = st; p.ss = ss;
    p.amode = amode;
    p.A2 = nullptr; p.lda2 = 0;
    const int T = tn_tile(N, K);
    const int tiles = ceil_div(N, T) * ceil_div(K, T);
    dim3 grid(tiles * p.S), block(256);
    const size_t lds = 2 * 2 * T * 128;
#define LTN(AM, TT, GM) hipLaunchKernelGGL((gemm_tn_kernel<AM, TT, GM>), grid, block, lds, stream, p)
    const int use_tr = 1;
    p.bias_grad = nullptr;
    if (bias_grad && !tuber_gemm_tn_fuses_bias(M, N, K, ldg, lda)) return TUBER_EINVAL;
    if (T == 64 && !gmode && use_tr && !((N | K | ldg | lda) & 7) && tn_big(M, N, K)) {     // ... on 128 x 128 tiles
        p.bias_grad = bias_grad;
        const dim3 grid3(ceil_div(N, 128) * ceil_div(K, 128) * p.S);
        if (amode == A_BN_RELU) hipLaunchKernelGGL(gemm_tn3_kernel<A_BN_RELU>, grid3, block, 0, stream, p);
        else hipLaunchKernelGGL(gemm_tn3_kernel<A_PLAIN>, grid3, block, 0, stream, p);
    } else if (T == 64 && !gmode && use_tr && !((N | K | ldg | lda) & 7)) {     // LDS transpose-read kernel
        p.bias_grad = bias_grad;
        if (amode == A_BN_RELU) hipLaunchKernelGGL(gemm_tn2_kernel<A_BN_RELU>, grid, block, 0, stream, p);
        else hipLaunchKernelGGL(gemm_tn2_kernel<A_PLAIN>, grid, block, 0, stream, p);
    } else if (T == 128) {
        if (amode == A_BN_RELU) LTN(A_BN_RELU, 128, 0); else LTN(A_PLAIN, 128, 0);
    } else {
        if (amode == A_BN_RELU) LTN(A_BN_RELU, 64, 0); else LTN(A_PLAIN, 64, 0);
    }
#undef LTN
    if (p.S > 1 && accumulate != 2) {          // accumulate == 2: the caller reduces the slabs later (tuber_multi_reduce)
        const long n = (long)N * K;
        if (p.S <= 16) hipLaunchKernelGGL(reduce_slabs_flat_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, stream, partial, out, n, p.S, accumulate);
        else hipLaunchKernelGGL(reduce_slabs_kernel, dim3(ceil_div(n, 32)), dim3(1024), 0, stream, partial, out, n, p.S, accumulate);
    }
    TUBER_RETURN_LAUNCH();
}


// one entry of tuber_gemm_tn_group (HOST memory): the tuber_gemm_tn arguments of one weight-gradient GEMM
struct TuberGemmTNArgs {
    const void* G; long ldg; const void* A; long lda; float* partial; float* out;
    int accumulate, M, N, K, amode, gather, To, Ho, Wo, Ti, Hi, Wi, st, ss;
    const float* a_scale; const float* a_shift;
    float* bias_grad;            // optional, as in tuber_gemm_tn: dbias[N] (single slab, accumulated) or [slabs][N] partial rows
    const void* A2; long lda2;   // optional: A := A + A2 (no gather then)
};
int tuber_gemm_tn_args_bytes(void) { return (int)sizeof(TuberGemmTNArgs); }
int tuber_gemm_tn_group_max(void) { return TN_GROUP_MAX; }

// n <= tuber_gemm_tn_group_max() weight-gradient GEMMs (each exactly what tuber_gemm_tn would compute, transpose-read kernel
// shapes only: 64 x 64 tiles, N / K / ld multiples of 8) in ONE launch.  Slab partials / accumulate flags per entry as in
// tuber_gemm_tn; with several slabs the caller reduces them (accumulate must be 2 then: the second stage is not launched here).
int tuber_gemm_tn_group(const void* args_host, int n, hipStream_t stream) {
    if (!args_host || n <= 0 || n > TN_GROUP_MAX) return TUBER_EINVAL;
    const TuberGemmTNArgs* a = (const TuberGemmTNArgs*)args_host;
    // problems on 128 x 128 tiles and on 64 x 64 tiles are different kernels (64 KB vs 36 KB of LDS): one launch per kind present
    GemmTNGroup gs[2];
    int total[2] = {0, 0}, cnt[2] = {0, 0};
    memset(gs, 0, sizeof gs);
    for (int i = 0; i < n; ++i) {
        const TuberGemmTNArgs& x = a[i];
        if (x.M <= 0 || x.N <= 0 || x.K <= 0 || ((x.N | x.K | x.ldg | x.lda) & 7) || x.ldg < x.N || x.lda < x.K) return TUBER_EINVAL;
        if (tn_tile(x.N, x.K) != 64 || (x.amode != A_PLAIN && x.amode != A_BN_RELU)) return TUBER_EINVAL;
        if (x.amode == A_BN_RELU && (!x.a_scale || !x.a_shift)) return TUBER_EINVAL;
        const int big = tn_big(x.M, x.N, x.K) ? 1 : 0;
        GemmTNGroup& g = gs[big];
        GemmTN& p = g.p[cnt[big]];
        p.G = (const bf16*)x.G; p.ldg = x.ldg; p.A = (const bf16*)x.A; p.lda = x.lda;
        p.M = x.M; p.N = x.N; p.K = x.K;
        p.rows_per_slab = tn_rows_per_slab(x.M, x.N, x.K);
        p.S = tuber_gemm_tn_slabs(x.M, x.N, x.K);
        if (p.S > 1 && x.accumulate != 2) return TUBER_EINVAL;
        p.accumulate = x.accumulate;
        p.P = p.S == 1 ? x.out : x.partial;
        if (!p.P) return TUBER_EINVAL;
        p.a_scale = x.a_scale; p.a_shift = x.a_shift;
        p.G2 = nullptr; p.ldg2 = 0; p.gA = p.gB = p.gC = nullptr; p.bias_grad = x.bias_grad;
        if (x.A2 && (x.gather || (x.lda2 & 7) || x.lda2 < x.K || x.amode != A_PLAIN)) return TUBER_EINVAL;
        p.A2 = (const bf16*)x.A2; p.lda2 = x.lda2;
        p.gather = x.gather; p.To = x.To; p.Ho = x.Ho; p.Wo = x.Wo; p.Ti = x.Ti; p.Hi = x.Hi; p.Wi = x.Wi; p.st = x.st; p.ss = x.ss;
        p.amode = x.amode;
        const int T = big ? 128 : 64;
        g.begin[cnt[big]] = total[big];
        total[big] += ceil_div(x.N, T) * ceil_div(x.K, T) * p.S;
        ++cnt[big];
    }
    for (int k = 0; k < 2; ++k) {
        if (!cnt[k]) continue;
        for (int i = cnt[k]; i <= TN_GROUP_MAX; ++i) gs[k].begin[i] = total[k];
        gs[k].n = cnt[k];
        if (k) hipLaunchKernelGGL(gemm_tn3_group_kernel, dim3(total[k]), dim3(256), 0, stream, gs[k]);
        else hipLaunchKernelGGL(gemm_tn2_group_kernel, dim3(total[k]), dim3(256), 0, stream, gs[k]);
    }
    TUBER_RETURN_LAUNCH();
}

}  // extern "C"
