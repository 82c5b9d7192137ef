// Diagnostic (GPU box only): what the 256 MB memory-side cache (Infinity Cache / MALL) gives a
// producer -> consumer pair of HBM-bound sweeps.  DESIGN.md section 8 (slab-ordered line passes).
//   1. read-only pass over a buffer of S MB: right after a kernel WROTE it, right after a kernel
//      READ it, and cold (2 GB of other traffic in between);
//   2. two dependent passes B = f(A), C = g(B) over 2 GB arrays, whole-array launches against
//      chunked launches (pass 1 of chunk c, then pass 2 of chunk c) for several chunk sizes.
// build: hipcc --offload-arch=gfx950 -O3 tools/mall_probe2.hip -o /tmp/mall_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); fflush(stdout); return 1; } } while (0)

__global__ void __launch_bounds__(256) readk(const float4 *__restrict__ s, float *out, size_t n) {
    float acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v = s[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1.2345f) out[0] = acc;
}
__global__ void __launch_bounds__(256) writek(float4 *__restrict__ d, size_t n, float x) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        d[i] = make_float4(x, 2.f, 3.f, 4.f);
}
__global__ void __launch_bounds__(256) copyk(const float4 *__restrict__ s, float4 *__restrict__ d, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v = s[i];
        v.x += 1.f;
        d[i] = v;
    }
}

int main() {
    const size_t GB = (size_t)1 << 30, MB = (size_t)1 << 20;
    float4 *a, *b, *c, *big;
    float *o;
    CK(hipMalloc(&a, 2 * GB));
    CK(hipMalloc(&b, 2 * GB));
    CK(hipMalloc(&c, 2 * GB));
    CK(hipMalloc(&big, 2 * GB));
    CK(hipMalloc(&o, 4));
    CK(hipDeviceSynchronize());
    printf("allocated\n"); fflush(stdout);
    CK(hipMemset(a, 0, 2 * GB));
    CK(hipMemset(b, 0, 2 * GB));
    CK(hipMemset(c, 0, 2 * GB));
    CK(hipMemset(big, 0, 2 * GB));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int G = 256 * 8;
    auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("== 1. read-only pass over S MB (GB/s)\n%8s %12s %12s %12s\n", "S", "after_write", "after_read", "cold");
    for (size_t mb : {16, 32, 64, 96, 128, 160, 192, 256, 320, 384, 512, 1024}) {
        const size_t n = mb * MB / 16;
        float r[3];
        for (int mode = 0; mode < 3; mode++) {
            std::vector<float> ts;
            for (int rep = 0; rep < 7; rep++) {
                hipLaunchKernelGGL(writek, dim3(G), dim3(256), 0, 0, a, n, (float)rep);
                if (mode == 2) hipLaunchKernelGGL(writek, dim3(G), dim3(256), 0, 0, big, 2 * GB / 16, 1.f);
                if (mode == 1) {
                    hipLaunchKernelGGL(writek, dim3(G), dim3(256), 0, 0, big, 2 * GB / 16, 1.f);
                    hipLaunchKernelGGL(readk, dim3(G), dim3(256), 0, 0, a, o, n);
                }
                hipEventRecord(e0);
                hipLaunchKernelGGL(readk, dim3(G), dim3(256), 0, 0, a, o, n);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                ts.push_back(ms);
            }
            r[mode] = (float)(mb / 1024.0 / (med(ts) * 1e-3));
        }
        printf("%6zuMB %12.0f %12.0f %12.0f\n", mb, r[0], r[1], r[2]); fflush(stdout); CK(hipGetLastError());
    }
    printf("== 2. B = f(A), C = g(B) over 2 GB: total ms (algorithmic 8 GB moved)\n");
    {
        const size_t n = 2 * GB / 16;
        for (size_t chunk_mb : {0, 32, 64, 96, 128, 192, 256, 512}) {
            std::vector<float> ts;
            for (int rep = 0; rep < 5; rep++) {
                hipEventRecord(e0);
                if (chunk_mb == 0) {
                    hipLaunchKernelGGL(copyk, dim3(G), dim3(256), 0, 0, a, b, n);
                    hipLaunchKernelGGL(copyk, dim3(G), dim3(256), 0, 0, b, c, n);
                } else {
                    const size_t cn = chunk_mb * MB / 16;
                    for (size_t off = 0; off < n; off += cn) {
                        const size_t m = std::min(cn, n - off);
                        hipLaunchKernelGGL(copyk, dim3(G), dim3(256), 0, 0, a + off, b + off, m);
                        hipLaunchKernelGGL(copyk, dim3(G), dim3(256), 0, 0, b + off, c + off, m);
                    }
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                ts.push_back(ms);
            }
            const float t = med(ts);
            printf("chunk %4zu MB: %7.3f ms  %7.0f GB/s algorithmic\n", chunk_mb, t, 8.589934592 / (t * 1e-3)); fflush(stdout);
        }
    }
    return 0;
}
