// Micro-benchmark: shader clock while packed multiply-adds run (s_memtime ticks per 100 MHz wall tick).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(64) void k(float* out, long long* clk, const float* __restrict__ w, int iters) {
    const long long c0 = clock64(), w0 = wall_clock64();
    f2 acc[8]; f2 x[16];
    for (int i = 0; i < 8; ++i) acc[i] = f2{(float)threadIdx.x + i, 1.0f};
    for (int i = 0; i < 16; ++i) x[i] = f2{(float)threadIdx.x * 0.001f + i, 0.5f + i};
    for (int it = 0; it < iters; ++it) {
        f2 ws[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) ws[i] = f2{w[(it & 7) * 32 + 2 * i], w[(it & 7) * 32 + 2 * i + 1]};
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_elementwise_fma(ws[(u * 8 + i) & 15], x[(u + i) & 15], acc[i]);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
int main() {
    const int blocks = 256 * 16;
    float *out, *w; long long* clk; CK(hipMalloc(&out, blocks * 64 * 4)); CK(hipMalloc(&w, 4096)); CK(hipMalloc(&clk, blocks * 16));
    CK(hipMemset(w, 0, 4096));
    long long* h = (long long*)malloc(blocks * 16);
    for (int iters : {60, 600, 6000, 60000, 600, 60}) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float ms;
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, out, clk, w, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h, clk, blocks * 16, hipMemcpyDeviceToHost));
        double c = 0, wl = 0; for (int i = 0; i < blocks; ++i) { c += h[2 * i]; wl += h[2 * i + 1]; }
        const double fl = (double)blocks * 64 * iters * 64 * 4;
        printf("iters %6d  %.3f ms  %.1f TFLOP/s  shader ticks / wall tick %.2f (x100 MHz)  wave cycles %.0f  per pk_fma per wave %.2f\n", iters, ms, fl / ms / 1e9, c / wl, c / blocks, c / blocks / (iters * 64.0));
    }
    return 0;
}
