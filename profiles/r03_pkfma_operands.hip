// Micro-benchmark: v_pk_fma_f32 rate by operand kind (VGPR / SGPR pair multiplier) and by number of interleaved chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

template <int CH, bool SG>
__global__ __launch_bounds__(64) void k(float* out, const float* __restrict__ w, int iters) {
    f2 acc[CH];
    f2 x[16];
    for (int i = 0; i < CH; ++i) acc[i] = f2{(float)threadIdx.x + i, 1.0f};
    for (int i = 0; i < 16; ++i) x[i] = f2{(float)threadIdx.x * 0.001f + i, 0.5f + i};
    for (int it = 0; it < iters; ++it) {
        f2 ws[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (SG) ws[i] = f2{w[(it & 7) * 32 + 2 * i], w[(it & 7) * 32 + 2 * i + 1]};            // uniform address: scalar loads
            else { ws[i] = x[(i + 5) & 15]; asm volatile("" : "+v"(ws[i])); }
        }
#pragma unroll
        for (int u = 0; u < 64 / CH; ++u) {
#pragma unroll
            for (int i = 0; i < CH; ++i) acc[i] = __builtin_elementwise_fma(ws[(u * CH + i) & 15], x[(u + i) & 15], acc[i]);
        }
    }
    float s = 0;
    for (int i = 0; i < CH; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int CH, bool SG>
void run(const char* name, float* out, float* w, int waves_per_simd) {
    const int blocks = 256 * 4 * waves_per_simd * 4, iters = 4000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        CK(hipEventRecord(e0)); hipLaunchKernelGGL((k<CH, SG>), dim3(blocks), dim3(64), 0, 0, out, w, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    const double fl = (double)blocks * 64 * iters * 64 * 4;
    printf("%-28s chains %d  %.3f ms  %.1f TFLOP/s\n", name, CH, best, fl / best / 1e9);
}

int main() {
    float *out, *w; CK(hipMalloc(&out, 256 * 4 * 8 * 4 * 64 * 4)); CK(hipMalloc(&w, 4096));
    CK(hipMemset(w, 0, 4096));
    run<8, false>("vgpr multiplier", out, w, 4);
    run<2, false>("vgpr multiplier", out, w, 4);
    run<1, false>("vgpr multiplier", out, w, 4);
    run<8, true>("sgpr-pair multiplier", out, w, 4);
    run<2, true>("sgpr-pair multiplier", out, w, 4);
    run<1, true>("sgpr-pair multiplier", out, w, 4);
    run<2, true>("sgpr-pair, 1 wave/SIMD", out, w, 1);
    run<2, false>("vgpr, 1 wave/SIMD", out, w, 1);
    return 0;
}
