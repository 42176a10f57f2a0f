// Prints which XCD (XCC_ID) each workgroup of a 256- / 512-block grid lands on, and the CU id.
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(unsigned* out) {
    if (threadIdx.x == 0) {
        unsigned xcc = __builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | ((4 - 1) << 11));
        unsigned hwid = __builtin_amdgcn_s_getreg((4 /*HW_REG_HW_ID*/) | (0 << 6) | ((32 - 1) << 11));
        out[blockIdx.x * 2] = xcc;
        out[blockIdx.x * 2 + 1] = hwid;
    }
}
int main() {
    for (int grid : {256, 512}) {
        unsigned* d; hipMalloc(&d, grid * 8);
        hipLaunchKernelGGL(probe, dim3(grid), dim3(512), 135168, 0, d);
        unsigned* h = (unsigned*)malloc(grid * 8);
        hipMemcpy(h, d, grid * 8, hipMemcpyDeviceToHost);
        printf("grid %d: xcc of blocks 0..31:", grid);
        for (int b = 0; b < 32; ++b) printf(" %u", h[b * 2]);
        int ok = 0; for (int b = 0; b < grid; ++b) ok += (h[b * 2] == (unsigned)(b % 8));
        int same8 = 0; for (int b = 8; b < grid; ++b) same8 += (h[b * 2] == h[(b - 8) * 2]);
        printf("\n  blocks with xcc == b%%8: %d / %d ; blocks sharing xcc with b-8: %d / %d\n", ok, grid, same8, grid - 8);
        hipFree(d); free(h);
    }
    return 0;
}
