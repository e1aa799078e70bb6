#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int pitch, int mode) {
    extern __shared__ __attribute__((aligned(16))) short s[];
    for (int i = threadIdx.x; i < 64 * pitch; i += 64) s[i] = (short)i;      // value = linear index = row*pitch + col
    __syncthreads();
    const int l = threadIdx.x, li = l & 15, g = l >> 4;
    const short* p;
    if (mode == 0) p = &s[(g * 4 + (li >> 2)) * pitch + (li & 3) * 4];        // chunk i of a [4][16] block at rows g*4..
    else p = &s[(g * 4) * pitch + li * 4];                                     // each lane its own row-chunk: row g*4, cols li*4..
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)p);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = v[e];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    short h[256];
    for (int mode = 0; mode < 2; ++mode) {
        const int pitch = 64;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 64 * pitch * 2, 0, d, pitch, mode);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d (pitch %d): value = row*pitch+col\n", mode, pitch);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", h[l * 4 + e] / pitch, h[l * 4 + e] % pitch);
            printf("\n");
        }
    }
    return 0;
}
