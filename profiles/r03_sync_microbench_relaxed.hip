// Micro-benchmark: cost of a per-image arrive / wait / depart protocol between the workgroups of one launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct Sync { unsigned ticket; unsigned pad[15]; };
// per image: 64-byte line {key u64, arrived u32, departed u32}
struct ImgSync { unsigned long long key; unsigned arrived; unsigned departed; unsigned long long result; unsigned flag; unsigned pad[9]; };

template <int MODE>   // 0 no sync; 1 ticket + arrive/wait/depart; 2 blockIdx order (no ticket) + arrive/wait/depart
__global__ __launch_bounds__(256) void k(Sync* s, ImgSync* im, int per_img, int total, const float* __restrict__ in, float* __restrict__ out, int work) {
    __shared__ unsigned s_t;
    __shared__ unsigned long long s_key[4];
    unsigned t = blockIdx.x;
    if (MODE == 1) {
        if (threadIdx.x == 0) s_t = __hip_atomic_fetch_add(&s->ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        t = s_t;
    }
    const int b = t / per_img, tile = t % per_img;
    // dummy work: read 8 floats per thread, some flops
    float v[8];
    const size_t base = ((size_t)t * 256 + threadIdx.x) * 8;
    for (int i = 0; i < 8; ++i) v[i] = in[base + i];
    float best = -1e30f;
    for (int w = 0; w < work; ++w)
        for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], 1.0001f, 0.001f);
    for (int i = 0; i < 8; ++i) best = fmaxf(best, v[i]);
    float rden = 1.0f;
    if (MODE == 3) {
        unsigned old = 0;
        if (threadIdx.x == 0) old = __hip_atomic_load(&im[b].flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o, 64));
        unsigned long long key = ((unsigned long long)__float_as_uint(fabsf(best)) << 32) | (unsigned)tile;
        if ((threadIdx.x & 63) == 0) s_key[threadIdx.x >> 6] = key;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 1; i < 4; ++i) key = s_key[i] > key ? s_key[i] : key;
            __hip_atomic_fetch_max(&im[b].key, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned n = __hip_atomic_fetch_add(&im[b].arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (n == (unsigned)per_img - 1) {
                const unsigned long long kk = __hip_atomic_load(&im[b].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&im[b].result, kk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&im[b].key, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&im[b].arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&im[b].flag, old ^ 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_key[0] = kk;
            } else {
                while (__hip_atomic_load(&im[b].flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == old) __builtin_amdgcn_s_sleep(2);
                s_key[0] = __hip_atomic_load(&im[b].result, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        __syncthreads();
        rden = 1.0f / (__uint_as_float((unsigned)(s_key[0] >> 32)) + 1e-10f);
    } else if (MODE != 0) {
        for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o, 64));
        unsigned long long key = ((unsigned long long)__float_as_uint(fabsf(best)) << 32) | (unsigned)tile;
        if ((threadIdx.x & 63) == 0) s_key[threadIdx.x >> 6] = key;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int i = 1; i < 4; ++i) key = s_key[i] > key ? s_key[i] : key;
            __hip_atomic_fetch_max(&im[b].key, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(&im[b].arrived, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(&im[b].arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)per_img) __builtin_amdgcn_s_sleep(2);
            const unsigned long long kk = __hip_atomic_load(&im[b].key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_key[0] = kk;
            const unsigned d = __hip_atomic_fetch_add(&im[b].departed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (d == (unsigned)per_img - 1) {       // last one out resets the image's line for the next launch
                __hip_atomic_store(&im[b].key, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&im[b].arrived, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&im[b].departed, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (MODE == 1 && t == (unsigned)total - 1) __hip_atomic_store(&s->ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        rden = 1.0f / (__uint_as_float((unsigned)(s_key[0] >> 32)) + 1e-10f);
    }
    for (int i = 0; i < 8; ++i) out[base + i] = v[i] * rden;
}

int main(int argc, char** argv) {
    const int B = 32, per_img = argc > 1 ? atoi(argv[1]) : 144, work = argc > 2 ? atoi(argv[2]) : 100;
    const int total = B * per_img;
    const size_t n = (size_t)total * 256 * 8;
    float *in, *out; Sync* s; ImgSync* im;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&s, sizeof(Sync))); CK(hipMalloc(&im, sizeof(ImgSync) * B));
    CK(hipMemset(in, 0, n * 4)); CK(hipMemset(s, 0, sizeof(Sync))); CK(hipMemset(im, 0, sizeof(ImgSync) * B));
    hipEvent_t a, b2; CK(hipEventCreate(&a)); CK(hipEventCreate(&b2));
    const int modes = argc > 3 ? atoi(argv[3]) : 3;
    for (int mode = 0; mode < 4; ++mode) {
        if (!((modes >> mode) & 1)) continue;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(a));
            for (int it = 0; it < 20; ++it) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(total), dim3(256), 0, 0, s, im, per_img, total, in, out, work);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(total), dim3(256), 0, 0, s, im, per_img, total, in, out, work);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(total), dim3(256), 0, 0, s, im, per_img, total, in, out, work);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(total), dim3(256), 0, 0, s, im, per_img, total, in, out, work);
            }
            CK(hipEventRecord(b2)); CK(hipEventSynchronize(b2));
            float ms; CK(hipEventElapsedTime(&ms, a, b2));
            printf("mode %d per_img %d work %d: %.2f us per launch\n", mode, per_img, work, 1e3 * ms / 20);
            fflush(stdout);
        }
    }
    std::vector<unsigned> h(16 * B + 16);
    CK(hipMemcpy(h.data(), im, sizeof(ImgSync) * B, hipMemcpyDeviceToHost));
    unsigned nz = 0; for (int i = 0; i < B; ++i) for (int j = 0; j < 4; ++j) nz |= h[16 * i + j];
    printf("sync state after runs: %s\n", nz ? "NOT CLEAR" : "clear");
    return 0;
}
