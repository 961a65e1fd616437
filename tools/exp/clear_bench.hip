// How fast can 16.6 MB be filled? grid size / store width sweep for k_clear.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) fill16(ulonglong2 *p, size_t pairs)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const ulonglong2 ones = make_ulonglong2(~0ull, ~0ull);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += stride) p[i] = ones;
}
__global__ void __launch_bounds__(256) fill16nt(ulonglong2 *p, size_t pairs)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < pairs; i += stride) {
        __builtin_nontemporal_store(~0ull, &p[i].x); __builtin_nontemporal_store(~0ull, &p[i].y);
    }
}
int main()
{
    const size_t n = 1920 * 1081 + 1, pairs = n / 2;
    ulonglong2 *p; hipMalloc(&p, n * 8 + 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int nt = 0; nt < 2; ++nt)
        for (int g : {256, 512, 1024, 2048, 4096, 8192}) {
            float best = 1e9;
            for (int it = 0; it < 20; ++it) {
                hipEventRecord(a, 0);
                if (nt) hipLaunchKernelGGL(fill16nt, dim3(g), dim3(256), 0, 0, p, pairs);
                else    hipLaunchKernelGGL(fill16, dim3(g), dim3(256), 0, 0, p, pairs);
                hipEventRecord(b, 0); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            printf("%s grid %5d: %.1f us  (%.2f TB/s)\n", nt ? "nontemporal" : "plain      ", g, best * 1000, n * 8 / (best * 1e-3) / 1e12);
        }
    float best = 1e9;
    for (int it = 0; it < 20; ++it) { hipEventRecord(a, 0); hipMemsetAsync(p, 0xFF, n * 8, 0); hipEventRecord(b, 0); hipEventSynchronize(b); float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms; }
    printf("hipMemsetAsync: %.1f us\n", best * 1000);
    return 0;
}
