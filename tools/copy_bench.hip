// Micro-benchmark (diagnostic, GPU box only): what a plain HBM copy / read / write reaches.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) copy1(const float4 *__restrict__ s, float4 *__restrict__ d, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) d[i] = s[i];
}
template <int U>
__global__ void __launch_bounds__(256) copyU(const float4 *__restrict__ s, float4 *__restrict__ d, size_t n) {
    size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) v[u] = s[base + (size_t)u * 256];
#pragma unroll
    for (int u = 0; u < U; u++) d[base + (size_t)u * 256] = v[u];
}
template <int U>
__global__ void __launch_bounds__(256) copyU_nt(const float4 *__restrict__ s, float4 *__restrict__ d, size_t n) {
    size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    typedef float v4 __attribute__((ext_vector_type(4)));
    const v4 *ss = reinterpret_cast<const v4 *>(s);
    v4 *dd = reinterpret_cast<v4 *>(d);
    v4 w[U];
    for (int u = 0; u < U; u++) w[u] = __builtin_nontemporal_load(&ss[base + (size_t)u * 256]);
#pragma unroll
    for (int u = 0; u < U; u++) __builtin_nontemporal_store(w[u], &dd[base + (size_t)u * 256]);
}
__global__ void __launch_bounds__(256) readk(const float4 *__restrict__ s, float *out, size_t n) {
    size_t base = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
    float acc = 0;
#pragma unroll
    for (int u = 0; u < 8; u++) { float4 v = s[base + (size_t)u * 256]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 1.2345f) out[0] = acc;
}
__global__ void __launch_bounds__(256) writek(float4 *__restrict__ d, size_t n) {
    size_t base = (size_t)blockIdx.x * 256 * 8 + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 8; u++) d[base + (size_t)u * 256] = make_float4(1.f, 2.f, 3.f, (float)u);
}
int main() {
    const size_t bytes = (size_t)1 << 30;  // 1 GiB
    const size_t n = bytes / 16;
    float4 *a, *b; float *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char *name, double moved, auto launch) {
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            for (int k = 0; k < 10; k++) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        ms /= 10;
        printf("%-34s %8.1f us  %7.1f GB/s\n", name, ms * 1e3, moved / ms / 1e6);
    };
    timeit("hipMemcpyAsync D2D (r+w)", 2.0 * bytes, [&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); });
    timeit("copy 1 float4/thread (r+w)", 2.0 * bytes, [&] { hipLaunchKernelGGL(copy1, dim3(n / 256), dim3(256), 0, 0, a, b, n); });
    timeit("copy 4 float4/thread (r+w)", 2.0 * bytes, [&] { hipLaunchKernelGGL(copyU<4>, dim3(n / 1024), dim3(256), 0, 0, a, b, n); });
    timeit("copy 8 float4/thread (r+w)", 2.0 * bytes, [&] { hipLaunchKernelGGL(copyU<8>, dim3(n / 2048), dim3(256), 0, 0, a, b, n); });
    timeit("copy 8 float4/thread nt (r+w)", 2.0 * bytes, [&] { hipLaunchKernelGGL(copyU_nt<8>, dim3(n / 2048), dim3(256), 0, 0, a, b, n); });
    timeit("read only", 1.0 * bytes, [&] { hipLaunchKernelGGL(readk, dim3(n / 2048), dim3(256), 0, 0, a, o, n); });
    timeit("write only", 1.0 * bytes, [&] { hipLaunchKernelGGL(writek, dim3(n / 2048), dim3(256), 0, 0, b, n); });
    // ---- does a copy care where source and destination sit?  K buffers of 1 GiB with 24 GiB spacers between
    // them (held), matrix of copy1 rates source i -> destination j (round 5: the regions of DESIGN 4.1)
    {
        const int K = 8;
        float4 *buf[K];
        void *spacer[K];
        int got = 0;
        for (int i = 0; i < K; i++) {
            if (hipMalloc(&buf[i], bytes) != hipSuccess) break;
            hipMemset(buf[i], i, bytes);
            got++;
            if (hipMalloc(&spacer[i], (size_t)24 << 30) != hipSuccess) { spacer[i] = nullptr; }
        }
        printf("copy 1 float4/thread, source i (row) -> destination j (column), GB/s; buffers 25 GiB apart in allocation order\n");
        for (int i = 0; i < got; i++) {
            for (int j = 0; j < got; j++) {
                if (i == j) { printf("    -  "); continue; }
                float ms = 0;
                for (int rep = 0; rep < 2; rep++) {
                    hipEventRecord(e0);
                    for (int k = 0; k < 10; k++) hipLaunchKernelGGL(copy1, dim3(n / 256), dim3(256), 0, 0, buf[i], buf[j], n);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                printf(" %6.0f", 2.0 * bytes / (ms / 10) / 1e6);
            }
            printf("\n");
        }
    }
    return 0;
}
