// LDS atomic-add throughput per CU by operand type (conflict-free addresses), MI355X.
#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T>
__global__ void __launch_bounds__(256) k(T *out, int iters) {
    __shared__ T tile[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) tile[i] = 0;
    __syncthreads();
    const T v = (T)1;
    int idx = threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
            __hip_atomic_fetch_add(&tile[(idx + 263 * u) & 4095], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        idx = (idx + 17) & 4095;
    }
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = tile[5];
}
template <typename T>
void run(const char *name) {
    T *o; hipMalloc(&o, 4096 * sizeof(T));
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int iters = 4000, blocks = 256 * 4;
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, o, 10);
    hipEventRecord(a);
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, o, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double n = (double)blocks * 256 * iters * 8;
    printf("%-6s %8.3f ms  %7.1f G atomics/s  = %.2f per clock per CU (2.4 GHz, 256 CUs)\n", name, ms, n / ms / 1e6,
           n / (ms * 1e-3) / 2.4e9 / 256);
}
int main() { run<double>("f64"); run<unsigned long long>("u64"); run<float>("f32"); run<unsigned>("u32"); run<int>("i32"); return 0; }
