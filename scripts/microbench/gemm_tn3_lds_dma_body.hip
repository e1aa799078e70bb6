// EXPERIMENT RECORD (round 6) -- NOT compiled into the product library.
// gemm_tn3 (the 128 x 128 weight-gradient tile of csrc/gemm.hip) with its operands parked by LDS-DMA instead of through VGPRs.
// Built behind a hook (tuber_gemm_tn_glds_set), bit-identical to the register-staged body on every shape of
// scripts/gemm_bench.py tngroup, and measured (profiles/r06_microbench_tn_group_lds_dma.txt, profiles/r06_ab_tn_glds_step.txt):
//   layer3 x8 60.1 -> 62.3 us, layer3 x16 83.3 -> 77.6 us (the register line itself spreads 80.3 - 86.3), layer4 x6 67.5 -> 66.9,
//   class-branch pair 78.3 -> 77.7, encoder FFN 13.3 -> 13.3; whole step 13.882 -> 13.911 ms (three same-box pairs).
// With NO ds_write left for the plain operands the launches take the same time: the k-loop is not bound by the VGPR -> LDS store path
// (the premise of the item); go / no-go was 15 %: NO-GO, removed from gemm.hip.  Kept here so the next attempt starts from code that
// is known to be correct (swizzle on the per-lane source address, key = lrow | ((piece >> 1) << 2)).  It plugs in as a second body
// behind `if (p.glds)` in gemm_tn3_kernel / gemm_tn3_group_kernel with the shared 64 KB LDS array declared by the kernel.
// ---------------------------------------------------------------------------------------------------------------------
// gemm_tn3 with LDS-DMA operand parking (round 6).  The register-staged body above moves every operand byte
// global -> VGPR -> LDS: 8 ds_write_b128 per thread and 64-row step, and the VGPR -> LDS store path runs at ~79 B/clk/CU against 256 B/clk
// for the transposed fragment reads (MI355X_MICROARCH.md, LDS table) -- per step the parking costs the LDS pipe about as much as the
// reads that feed the MFMAs.  Here global_load_lds_dwordx4 writes the tile straight into the image ds_read_b64_tr_b16 reads: one wave
// instruction = 64 lanes x 16 B = FOUR 256-byte rows at LDS base + lane * 16 (the destination is lane-linear), so the XOR swizzle of
// the image moves into the per-lane SOURCE address: LDS position (row r, 32-byte unit u) holds source unit u ^ tn3_key(r).  A wave
// parks rows [16 w, 16 w + 16) of a step with 4 instructions per operand; no staging registers, no ds_write.  An operand with the
// BatchNorm + ReLU prologue (conv4's weight gradient: A = relu(bn3(c3))) cannot come by DMA -- it keeps the register path (issued
// under the step's MFMAs, converted and stored behind them); the gradient operand G is always plain.  Double-buffered, one step ahead:
//     wait vmcnt(0) | barrier | issue step s+1 -> the other buffer | MFMAs of step s | (bn_relu: park A of step s+1)
// Taken for dense problems whose slabs are whole 64-row steps (every backbone shape); everything else runs the body above.
// ---------------------------------------------------------------------------------------------------------------------
template <int AMODE>
__device__ __forceinline__ void gemm_tn3g_body(const GemmTN& p, int bid, int nblocks, Tn3Smem smem) {
    constexpr int T = 128, TW = 64, FT = 4;
    const bool bn_relu = AMODE < 0 ? p.amode == A_BN_RELU : AMODE == A_BN_RELU;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, g = lane >> 4;
    const int tiles_n = p.N / T, tiles_k = p.K / T;
    int b = xcd_remap(bid, nblocks);
    const int slab = b / (tiles_n * tiles_k);
    b -= slab * tiles_n * tiles_k;
    const int tile_n = b % tiles_n, tile_k = b / tiles_n;
    const int n0 = tile_n * T, k0 = tile_k * T;
    const int m_begin = slab * p.rows_per_slab;
    const int m_end = min(p.M, m_begin + p.rows_per_slab);
    const int nsteps = (m_end - m_begin) >> 6;

    // LDS-DMA geometry: lane -> (row lrow of a 4-row piece, 16-byte position cpos of the 256-byte LDS row).  Piece i of a wave covers rows
    // 16 w + 4 i + lrow; tn3_key(row) = (row & 3) | (((row >> 3) & 1) << 2) = lrow | ((i >> 1) << 2): two source columns per lane
    const int lrow = lane >> 4, cpos = lane & 15;
    const int scol0 = ((((cpos >> 1) ^ lrow) << 4) | ((cpos & 1) << 3));
    const int scol1 = ((((cpos >> 1) ^ (lrow | 4)) << 4) | ((cpos & 1) << 3));
    const bf16* gsrc = p.G + (long)(m_begin + wave * 16 + lrow) * p.ldg + n0;
    const bf16* asrc = p.A + (long)(m_begin + wave * 16 + lrow) * p.lda + k0;
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto issue = [&](int step, int buf) {
        const long mo = (long)step * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int sc = (i >> 1) ? scol1 : scol0;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + (mo + 4 * i) * p.ldg + sc),
                                             (lds_ptr)&smem[buf][0][(wave * 16 + 4 * i) * TNP3], 16, 0, 0);
        }
        if (!bn_relu) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int sc = (i >> 1) ? scol1 : scol0;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(asrc + (mo + 4 * i) * p.lda + sc),
                                                 (lds_ptr)&smem[buf][1][(wave * 16 + 4 * i) * TNP3], 16, 0, 0);
            }
        }
    };
    // register path of a BatchNorm + ReLU operand: thread -> 16-byte chunk c of rows r, r + 16, r + 32, r + 48 (as in gemm_tn3_body)
    const int c = tid & 15, r = tid >> 4;
    float asc[8], ash[8];
    if (bn_relu) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { asc[e] = p.a_scale[k0 + c * 8 + e]; ash[e] = p.a_shift[k0 + c * 8 + e]; }
    }
    const bf16* ab0 = p.A + (long)(m_begin + r) * p.lda + k0 + c * 8;
    const long a16 = 16 * p.lda;
    uint4 ra[4];
    auto load_a = [&](int step) {
        const bf16* aq = ab0 + (long)step * 64 * p.lda;
#pragma unroll
        for (int h = 0; h < 4; ++h) ra[h] = *(const uint4*)(aq + h * a16);
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            const bf16x8 x = as_bf16x8(ra[h]);
            bf16x8 y;
#pragma unroll
            for (int e = 0; e < 8; ++e) y[e] = f2bf(fmaxf(fmaf(bf2f(x[e]), asc[e], ash[e]), 0.f));
            *(uint4*)&smem[buf][1][tn3_off(r + 16 * h, c * 8)] = as_uint4(y);
        }
    };
    const bool do_bias = p.bias_grad != nullptr && tile_k == 0;
    float bsum = 0.f;

    f32x4 acc[FT][FT];
#pragma unroll
    for (int i = 0; i < FT; ++i)
#pragma unroll
        for (int j = 0; j < FT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue(0, 0);
    if (bn_relu) { load_a(0); store_a(0); }
    for (int s = 0; s < nsteps; ++s) {
        const int buf = s & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of step s have landed ...
        __syncthreads();                                        // ... everybody's have, and nobody reads the other buffer any more
        const bool more = s + 1 < nsteps;
        if (more) {
            issue(s + 1, buf ^ 1);
            if (bn_relu) load_a(s + 1);
        }
        const bf16* gi = smem[buf][0];
        const bf16* ai = smem[buf][1];
        if (do_bias) {
#pragma unroll
            for (int mm = 0; mm < 32; ++mm) bsum += bf2f(gi[tn3_off((tid >> 7) * 32 + mm, tid & 127)]);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int m0 = ks * 32 + g * 8;
            bf16x8 fn[FT], fk[FT];
#pragma unroll
            for (int j2 = 0; j2 < FT; ++j2) fn[j2] = tn3_frag(gi, m0, wn * TW + j2 * 16, li);
#pragma unroll
            for (int i2 = 0; i2 < FT; ++i2) fk[i2] = tn3_frag(ai, m0, wm * TW + i2 * 16, li);
#pragma unroll
            for (int i2 = 0; i2 < FT; ++i2)
#pragma unroll
                for (int j2 = 0; j2 < FT; ++j2)
                    acc[i2][j2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[j2], fk[i2], acc[i2][j2], 0, 0, 0);
        }
        if (bn_relu && more) store_a(buf ^ 1);
    }
    // epilogue: as gemm_tn3_body
    float* P = p.P + (long)slab * p.N * p.K;
    __syncthreads();
    float* ot = (float*)&smem[0][0][0];
    if (do_bias) {
        float* br = ot + 64 * 132;
        br[tid] = bsum;
        __syncthreads();
        if (tid < 128) {
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
            float4 v = *(const float4*)&ot[(idx >> 5) * 132 + (idx & 31) * 4];
            float4* o = (float4*)(P + (long)n * p.K + k);
            if (p.S == 1 && p.accumulate) { const float4 cc = *o; v.x += cc.x; v.y += cc.y; v.z += cc.z; v.w += cc.w; }
            *o = v;
        }
    }
}

