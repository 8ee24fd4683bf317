// Probe of ds_read_b64_tr_b16's lane <-> address mapping (gfx950): LDS holds its own element indices; every lane passes the address of
// "its" 4 contiguous elements of a [4 rows][16 cols] block with row pitch P; prints what each lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int pitch) {
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    const int l = threadIdx.x, i = l & 15, g = l >> 4;
    const short* p = lds + g * 1024 + (i >> 2) * pitch + (i & 3) * 4;   // lane i: row i >> 2, column quad i & 3 of its group's block
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 64 * 4 * 2);
    for (int pitch : {16, 64, 136}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, pitch);
        short h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        printf("pitch %d\n", pitch);
        for (int l = 0; l < 64; l += 1) {
            if (l % 16 < 6 || l % 16 == 15) {
                printf("  lane %2d:", l);
                for (int j = 0; j < 4; ++j) { int e = h[l * 4 + j] - (l >> 4) * 1024; printf(" (r%d,c%d)", e / pitch, e % pitch); }
                printf("\n");
            }
        }
    }
    return 0;
}
