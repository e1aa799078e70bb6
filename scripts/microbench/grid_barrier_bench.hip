// Prices a software grid barrier on MI355X ("go if <= 2 us").  Stand-alone tuning program, not product code.
//   hipcc --offload-arch=gfx950 -O3 -o grid_barrier_bench grid_barrier_bench.hip && ./grid_barrier_bench
// A persistent kernel (one 256-thread workgroup per CU, all co-resident) runs K rounds of
//   publish (each workgroup stores ROW bytes) -> barrier -> consume (each workgroup reads the row of another workgroup and checks it)
// Variants:
//   dev_fence   : device-wide, publication by plain stores + __threadfence() (release) / acquire fence after the barrier
//   dev_nofence : device-wide, publication by write-through device-scope stores (sc1) + s_waitcnt, counter = relaxed agent-scope
//                 atomic, consumption by device-scope (sc1) loads; no fence instruction anywhere
//   dev_bar_only: the counter round trip alone (no data)
//   xcd_*       : the same among the workgroups of ONE XCD (blockIdx % 8), everything at L2 scope (atomics without sc1, sc0 loads):
//                 the XCD's L2 is the coherence point, nothing leaves the die
// Time per round = kernel duration / K (HIP events), reported with the kernel launch + drain amortised over K = 2000 rounds.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>


#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

enum { DEV_FENCE = 0, DEV_NOFENCE = 1, DEV_BAR_ONLY = 2, XCD_NOFENCE = 3, XCD_BAR_ONLY = 4, DEV_TWO_LEVEL = 5 };

__device__ __forceinline__ void store_sc1(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ float load_sc1(const float* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// how the XCD-local variant reads another workgroup's row (published by plain stores + s_waitcnt vmcnt(0), i.e. resident in the XCD's L2):
// 0 = sc0 load, 1 = sc1 load, 2 = plain load behind ONE "buffer_inv sc1" per round, 3 = sc0 sc1 load, 4 = nt load
__device__ int g_data_mode;
__device__ __forceinline__ float load_sc0(const float* p) {
    float v;
    if (g_data_mode == 1) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (g_data_mode == 2) v = *(const volatile float*)p;
    else if (g_data_mode == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else if (g_data_mode == 4) asm volatile("global_load_dword %0, %1, off nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    else asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ unsigned g_timeouts;
// bounded spin: a barrier that cannot complete (wrong assumption about the workgroup -> XCD map, stale reads) is reported, not hung on
#define SPIN_UNTIL(cond)                                                                   \
    do {                                                                                   \
        int spins__ = 0;                                                                   \
        while (!(cond)) {                                                                  \
            __builtin_amdgcn_s_sleep(1);                                                   \
            if (++spins__ > 100000) { atomicAdd(&g_timeouts, 1u); break; }                \
        }                                                                                  \
    } while (0)
__device__ __forceinline__ int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15; }      // HW_REG_XCC_ID[3:0]
__global__ void xcc_probe_kernel(int* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }
// how a workgroup looks at a counter that lives in its XCD's L2: 0 = load with sc0 (group scope), 1 = an atomic read-modify-write of
// zero (always executes in the L2), 2 = load with sc1 (device scope)
__device__ int g_spin_mode;
__device__ __forceinline__ unsigned load_u32_sc0(const unsigned* p) {
    unsigned v;
    if (g_spin_mode == 1) return __hip_atomic_fetch_add((unsigned*)p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (g_spin_mode == 2) { asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory"); return v; }
    asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

template <int MODE>
__global__ __launch_bounds__(256) void bench_kernel(unsigned* counters, float* rows, int row_floats, int rounds, unsigned* errors) {
    const int nb = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
    constexpr bool XCD = MODE == XCD_NOFENCE || MODE == XCD_BAR_ONLY;
    const int group = XCD ? (b & 7) : 0;                 // barrier group: the XCD, or the whole grid
    const int gsize = XCD ? (nb - group + 7) / 8 : nb;
    unsigned* ctr = counters + group * 64;               // one cache line per counter
    const int peer = XCD ? ((b + 8 < nb) ? b + 8 : group) : (b + 1) % nb;      // whose row this workgroup consumes (same group)
    unsigned bad = 0;
    for (int r = 0; r < rounds; ++r) {
        if (__hip_atomic_load(&g_timeouts, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;      // a barrier failed: stop, the host reports it
        float* mine = rows + ((size_t)(r & 1) * nb + b) * row_floats;
        const float* theirs = rows + ((size_t)(r & 1) * nb + peer) * row_floats;
        if (MODE != DEV_BAR_ONLY && MODE != XCD_BAR_ONLY) {
            for (int i = tid; i < row_floats; i += 256) {
                const float v = (float)(r * 7 + b + i);
                if (MODE == DEV_FENCE) mine[i] = v;
                else if (MODE == XCD_NOFENCE) mine[i] = v;        // TCP is write-through: a plain store lands in the XCD's L2
                else store_sc1(mine + i, v);                      // write-through to the device coherence point
            }
            if (MODE == DEV_FENCE) __threadfence();
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (tid == 0) {
            const unsigned target = (unsigned)(r + 1) * gsize;
            if (XCD) {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // executes in the XCD's L2
                SPIN_UNTIL(load_u32_sc0(ctr) >= target);
            } else if (MODE == DEV_TWO_LEVEL) {
                // level 1 inside the XCD (L2 atomics), level 2 among the 8 XCD leaders at device scope, release back through the XCD counter
                unsigned* xc = counters + 64 * (1 + (b & 7));
                const int xsize = (nb - (b & 7) + 7) / 8;
                const unsigned old = __hip_atomic_fetch_add(xc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                // the XCD counter takes xsize arrivals + ONE release bump per round: the last arriver of the XCD goes to the device counter,
                // waits for the 8 leaders there and then bumps the XCD counter once more
                if (old % (xsize + 1) == (unsigned)xsize - 1) {
                    __hip_atomic_fetch_add(counters, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    SPIN_UNTIL(__hip_atomic_load(counters, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)(r + 1) * 8);
                    __hip_atomic_fetch_add(xc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                SPIN_UNTIL(load_u32_sc0(xc) >= (unsigned)(r + 1) * (xsize + 1));
            } else {
                __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                SPIN_UNTIL(__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target);
            }
        }
        __syncthreads();
        if (MODE == DEV_FENCE) __threadfence();          // acquire side (invalidates)
        if (MODE == XCD_NOFENCE && g_data_mode == 2) asm volatile("buffer_inv sc1" ::: "memory");
        if (MODE != DEV_BAR_ONLY && MODE != XCD_BAR_ONLY) {
            for (int i = tid; i < row_floats; i += 256) {
                const float want = (float)(r * 7 + peer + i);
                const float got = MODE == DEV_FENCE ? theirs[i] : MODE == XCD_NOFENCE ? load_sc0(theirs + i) : load_sc1(theirs + i);
                bad += got != want;
            }
        }
    }
    if (bad) atomicAdd(errors, bad);
}

template <int MODE>
static void run(const char* name, int nb, int row_floats, int rounds) {
    unsigned *counters, *errors;
    float* rows;
    CHECK(hipMalloc(&counters, 64 * 16 * sizeof(unsigned)));
    CHECK(hipMalloc(&errors, sizeof(unsigned)));
    CHECK(hipMalloc(&rows, (size_t)2 * nb * (row_floats > 0 ? row_floats : 1) * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    float best = 1e30f;
    unsigned err = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipMemset(counters, 0, 64 * 16 * sizeof(unsigned)));
        CHECK(hipMemset(errors, 0, sizeof(unsigned)));
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench_kernel<MODE>, dim3(nb), dim3(256), 0, 0, counters, rows, row_floats, rounds, errors);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        unsigned e;
        CHECK(hipMemcpy(&e, errors, sizeof e, hipMemcpyDeviceToHost));
        err += e;
        unsigned to = 0;
        CHECK(hipMemcpyFromSymbol(&to, HIP_SYMBOL(g_timeouts), sizeof to));
        if (to) { printf("%-14s workgroups %4d: BARRIER DID NOT COMPLETE (%u spin time-outs) -- variant abandoned\n", name, nb, to); to = 0; CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_timeouts), &to, sizeof to)); CHECK(hipFree(counters)); CHECK(hipFree(errors)); CHECK(hipFree(rows)); return; }
    }
    printf("%-14s workgroups %4d  row %5d B  rounds %d : %7.3f us per round   (stale / wrong values read: %u)\n", name, nb, row_floats * 4, rounds, 1e3f * best / rounds, err);
    fflush(stdout);
    CHECK(hipFree(counters)); CHECK(hipFree(errors)); CHECK(hipFree(rows));
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("%s, %d CUs\n", p.name, cus);
    const int K = 2000;
    {   // where do workgroups land?  (the XCD-local variants assume blockIdx % 8 == XCD)
        int* d; CHECK(hipMalloc(&d, 4096 * sizeof(int)));
        hipLaunchKernelGGL(xcc_probe_kernel, dim3(cus), dim3(64), 0, 0, d);
        std::vector<int> h(cus);
        CHECK(hipMemcpy(h.data(), d, cus * sizeof(int), hipMemcpyDeviceToHost));
        int match = 0;
        for (int i = 0; i < cus; ++i) match += h[i] == (i & 7);
        printf("XCC_ID of workgroup b == b %% 8 for %d of %d workgroups; first 16:", match, cus);
        for (int i = 0; i < 16; ++i) printf(" %d", h[i]);
        printf("\n");
        CHECK(hipFree(d));
    }
    for (int nb : {cus, 64}) {
        run<DEV_BAR_ONLY>("dev_bar_only", nb, 0, K);
        run<DEV_NOFENCE>("dev_nofence", nb, 512, K);      // 2 KB per workgroup = a [2][256] fp32 statistics row
        run<DEV_FENCE>("dev_fence", nb, 512, K);
        for (int sm = 0; sm < 3; sm += 2) {
            CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_spin_mode), &sm, sizeof sm));
            printf("  [XCD-local counter read by: %s]\n", sm == 0 ? "sc0 load" : "sc1 load");
            run<XCD_BAR_ONLY>("xcd_bar_only", nb, 0, K);
            run<DEV_TWO_LEVEL>("dev_two_level", nb, 0, K);
        }
        for (int dm = 0; dm < 5; ++dm) {
            CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_data_mode), &dm, sizeof dm));
            const char* names[] = {"sc0 loads", "sc1 loads", "buffer_inv sc1 + plain loads", "sc0 sc1 loads", "nt loads"};
            printf("  [XCD-local rows (plain stores + s_waitcnt) read by: %s]\n", names[dm]);
            run<XCD_NOFENCE>("xcd_nofence", nb, 512, K);
        }
    }
    // kernel-boundary reference: K dependent empty launches
    {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        unsigned *counters, *errors; float* rows;
        CHECK(hipMalloc(&counters, 4096)); CHECK(hipMalloc(&errors, 4)); CHECK(hipMalloc(&rows, 1 << 20));
        CHECK(hipMemset(counters, 0, 4096));
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            for (int i = 0; i < K; ++i) hipLaunchKernelGGL(bench_kernel<DEV_BAR_ONLY>, dim3(cus), dim3(256), 0, 0, counters, rows, 0, 0, errors);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("reference: %d back-to-back empty launches of %d workgroups on one stream: %7.3f us per launch (host-issue bound when > ~4 us; inside a hipGraph a dependent launch costs ~2.5 us, DESIGN.md)\n", K, cus, 1e3f * ms / K);
        }
    }
    return 0;
}
