// Micro-benchmark: peak fp32 rate of packed vector multiply-adds against the fp32 matrix instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_pk(float* out, int iters, float a, float b) {
    f2 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f2{(float)threadIdx.x + i, 1.0f};
    const f2 x = {a, b}, y = {b, a};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_elementwise_fma(acc[i], x, y);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_fma(float* out, int iters, float a, float b) {
    float acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (float)threadIdx.x + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = fmaf(acc[i], a, b);
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_mfma(float* out, int iters, float a, float b) {
    f4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f4{(float)threadIdx.x, 1.0f, 2.0f, (float)i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    const int blocks = 256 * 8, iters = 2000;
    float* out; CK(hipMalloc(&out, blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_pk, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double f_pk = (double)blocks * 256 * iters * 64 * 2 * 2;      // 64 packed instr per iteration, 2 lanes of the pair, 2 flop
        printf("v_pk_fma_f32            %.3f ms  %.1f TFLOP/s\n", ms, f_pk / ms / 1e9);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double f_f = (double)blocks * 256 * iters * 64 * 2;
        printf("v_fma_f32               %.3f ms  %.1f TFLOP/s\n", ms, f_f / ms / 1e9);
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double f_m = (double)blocks * 4 * iters * 16 * (16.0 * 16 * 4 * 2);   // per wavefront: 16 instr per iteration, 2048 flop each
        printf("v_mfma_f32_16x16x4_f32  %.3f ms  %.1f TFLOP/s\n", ms, f_m / ms / 1e9);
    }
    return 0;
}
