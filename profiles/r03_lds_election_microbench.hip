// Micro-benchmark: one LDS exchange-with-return against a byte write followed by a byte read (the election of the gather).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned* out, int iters) {
    __shared__ unsigned own32[4][128];
    __shared__ uint8_t own8[4][256];
    __shared__ float win[4][1024];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < 128; i += 64) own32[wv][i] = 0;
    for (int i = lane; i < 256; i += 64) own8[wv][i] = 0;
    for (int i = lane; i < 1024; i += 64) win[wv][i] = 0;
    __syncthreads();
    volatile unsigned* o32 = own32[wv];
    volatile uint8_t* o8 = own8[wv];
    volatile float* w = win[wv];
    unsigned wins = 0;
    int c = lane + (blockIdx.x & 7);
    for (int it = 1; it <= iters; ++it) {
        c = (c + 73) % 900;
        bool won;
        if (MODE == 0) {
            o8[c & 255] = (uint8_t)lane;
            won = o8[c & 255] == (uint8_t)lane;
        } else if (MODE == 1) {
            const unsigned prev = __hip_atomic_exchange((unsigned*)&own32[wv][c & 127], ((unsigned)it << 8) | lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            won = (prev >> 8) != (unsigned)it;
        } else {
            won = true;
        }
        if (won) {
            w[c] = w[c] + 1.0f; w[c + 1] = w[c + 1] + 2.0f; w[c + 72] = w[c + 72] + 3.0f; w[c + 73] = w[c + 73] + 4.0f;
            ++wins;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = wins + (unsigned)w[lane];
}
template <int MODE>
void run(const char* name, unsigned* out) {
    const int blocks = 256 * 6, iters = 4000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        float ms;
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    printf("%-44s %.3f ms  (%.1f ns per step and wavefront slot)\n", name, best, best * 1e6 / iters);
}
int main() {
    unsigned* out; CK(hipMalloc(&out, 256 * 6 * 256 * 4));
    run<2>("four read-modify-writes, no election", out);
    run<0>("byte write + byte read election, then the four", out);
    run<1>("one exchange-with-return election, then the four", out);
    return 0;
}
