// Which lane does lane i read from under row_mirror followed by row_ror:1 (and the other orders)?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(int *out) {
    const int lane = threadIdx.x;
    int v = lane;
    int m = __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, false);      // row_mirror
    int a = __builtin_amdgcn_update_dpp(0, m, 0x121, 0xf, 0xf, false);      // row_ror:1 of mirror
    int r = __builtin_amdgcn_update_dpp(0, v, 0x121, 0xf, 0xf, false);      // row_ror:1
    int b = __builtin_amdgcn_update_dpp(0, r, 0x140, 0xf, 0xf, false);      // mirror of ror:1
    int c = __builtin_amdgcn_update_dpp(0, m, 0x12f, 0xf, 0xf, false);      // row_ror:15 of mirror
    out[lane] = m; out[64 + lane] = a; out[128 + lane] = b; out[192 + lane] = c; out[256 + lane] = r;
}
int main() {
    int *d; hipMalloc(&d, 320 * 4);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    int h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char *names[5] = {"mirror", "ror1(mirror)", "mirror(ror1)", "ror15(mirror)", "ror1"};
    for (int t = 0; t < 5; t++) { printf("%-14s", names[t]); for (int i = 0; i < 20; i++) printf(" %2d", h[64 * t + i]); printf("\n"); }
    return 0;
}
