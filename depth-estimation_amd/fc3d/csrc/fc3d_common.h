// Shared helpers for the gfx950 kernels of libfc3d_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "fc3d.h"

#define FC3D_LAUNCH_CHECK()                         \
    do {                                            \
        hipError_t e__ = hipGetLastError();         \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

#define FC3D_HIP(call)                              \
    do {                                            \
        hipError_t e__ = (call);                    \
        if (e__ != hipSuccess) return (int)e__;     \
    } while (0)

// Index and mask decisions must reproduce the reference's separately rounded fp32 ops:
// never let the compiler fuse a*b+c on its own (fmaf is written out where a fused op is meant).
#pragma clang fp contract(off)

namespace fc3d {

bool prof_enabled();
void prof_before(const char* name, hipStream_t s);
void prof_after(hipStream_t s);

// Launch + (optionally) time one kernel under a stable display name.
#define FC3D_LAUNCH(NAME, KERNEL, GRID, BLOCK, LDS, STREAM, ...)                        \
    do {                                                                               \
        const bool prof__ = ::fc3d::prof_enabled();                                    \
        if (prof__) ::fc3d::prof_before(NAME, STREAM);                                 \
        hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__);             \
        if (prof__) ::fc3d::prof_after(STREAM);                                        \
        FC3D_LAUNCH_CHECK();                                                           \
    } while (0)

constexpr int kWave = 64;

// Pointers into LDS that say so in their type.  A plain (generic) pointer to a __shared__ array that
// the compiler cannot trace back to the array (volatile accesses, pointers held across lambdas) is
// accessed with flat_load / flat_store: slower than ds_read / ds_write and counted in the vector
// memory counter too, so that every wait for an LDS value also waits for all global loads in flight.
#define FC3D_LDS __attribute__((address_space(3)))
typedef volatile FC3D_LDS float* lds_vfloat;
typedef volatile FC3D_LDS uint8_t* lds_vu8;
typedef volatile FC3D_LDS int* lds_vint;
template <typename T>
__device__ __forceinline__ volatile FC3D_LDS T* lds_ptr(T* p) {
    return (volatile FC3D_LDS T*)p;
}

// Workgroups of a launch go to the eight XCDs round robin in launch order, and each XCD has its own
// L2.  block_x() renumbers blockIdx.x so that XCD k works through the k-th contiguous eighth of the
// x range: neighbouring tiles (which share halo rows or gather the same source lines) then meet in
// one L2 instead of being fetched once per XCD.
#ifndef FC3D_XCD_REMAP
#define FC3D_XCD_REMAP 1
#endif
__device__ __forceinline__ unsigned block_x() {
#if FC3D_XCD_REMAP
    const unsigned x = blockIdx.x, q = gridDim.x >> 3;
    return x < (q << 3) ? (x & 7u) * q + (x >> 3) : x;
#else
    return blockIdx.x;
#endif
}

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline bool aligned4(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 3u) == 0; }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// The lanes of the wavefront for which `p` holds, as a mask (taken from the comparison that made `p`;
// __ballot goes through an integer and costs two vector instructions more).
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

// Order-preserving map float -> uint32 (larger float <=> larger key); -0 is folded onto +0
// so that ties between them are broken by position like torch.max does.
__device__ __forceinline__ uint32_t float_to_key(float v) {
    v = v + 0.0f;
    uint32_t u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}

// Uniform 64-bit base + 32-bit per-lane *byte* offset: the form the global load/store
// instructions take directly (scalar base register pair + one offset VGPR), no 64-bit lane math.
template <typename T>
__device__ __forceinline__ T ld_at(const T* base, unsigned elem) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + elem * (unsigned)sizeof(T));
}
template <typename T>
__device__ __forceinline__ void st_at(T* base, unsigned elem, T v) {
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + elem * (unsigned)sizeof(T)) = v;
}

// A pointer argument of a function that is not inlined arrives as a generic pointer in vector
// registers: its loads become flat_load (which also wait on the LDS counter) with 64-bit per-lane
// address arithmetic.  uniform_global() restates what the caller knows -- the pointer is the same
// in every lane and points to global memory -- so ld_global() is a global_load from a scalar base.
#define FC3D_GLOBAL __attribute__((address_space(1)))
template <typename T>
__device__ __forceinline__ const FC3D_GLOBAL T* uniform_global(const T* p) {
    const uint64_t a = reinterpret_cast<uint64_t>(p);
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)a), hi = __builtin_amdgcn_readfirstlane((uint32_t)(a >> 32));
    return reinterpret_cast<const FC3D_GLOBAL T*>(((uint64_t)hi << 32) | lo);
}
template <bool NT = false>
__device__ __forceinline__ float ld_global(const FC3D_GLOBAL float* base, unsigned elem) {
    const FC3D_GLOBAL float* p = reinterpret_cast<const FC3D_GLOBAL float*>(reinterpret_cast<const FC3D_GLOBAL char*>(base) + elem * 4u);
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT = false>
__device__ __forceinline__ float2 ld_global(const FC3D_GLOBAL float2* base, unsigned elem) {
    typedef float f2v __attribute__((ext_vector_type(2)));     // a plain vector: float2's copy constructor takes generic references only
    const FC3D_GLOBAL f2v* p = reinterpret_cast<const FC3D_GLOBAL f2v*>(reinterpret_cast<const FC3D_GLOBAL char*>(base) + elem * 8u);
    f2v v;
    if constexpr (NT) v = __builtin_nontemporal_load(p);
    else v = *p;
    return make_float2(v.x, v.y);
}

// two neighbouring elements as one load: elements 2 * pair and 2 * pair + 1 (base aligned to the pair)
template <bool NT = false>
__device__ __forceinline__ float2 ld_global_pair(const FC3D_GLOBAL float* base, unsigned pair) {
    typedef float f2v __attribute__((ext_vector_type(2)));
    const FC3D_GLOBAL f2v* p = reinterpret_cast<const FC3D_GLOBAL f2v*>(reinterpret_cast<const FC3D_GLOBAL char*>(base) + pair * 8u);
    f2v v;
    if constexpr (NT) v = __builtin_nontemporal_load(p);
    else v = *p;
    return make_float2(v.x, v.y);
}
template <bool NT = false>
__device__ __forceinline__ float4 ld_global_pair(const FC3D_GLOBAL float2* base, unsigned pair) {
    typedef float f4v __attribute__((ext_vector_type(4)));
    const FC3D_GLOBAL f4v* p = reinterpret_cast<const FC3D_GLOBAL f4v*>(reinterpret_cast<const FC3D_GLOBAL char*>(base) + pair * 16u);
    f4v v;
    if constexpr (NT) v = __builtin_nontemporal_load(p);
    else v = *p;
    return make_float4(v.x, v.y, v.z, v.w);
}

// Cache policy of streaming accesses.  A step moves about a gigabyte through a 32 MB L2 and a 256 MB
// memory-side cache: what is read once (scene points, upstream gradients) or written and not read
// again by the step (grids, the capture, the pattern gradient) is marked non-temporal, so that it does
// not push out what a later kernel of the step reads again (the coded pattern, the layer indices, the
// sampling grid).  FC3D_NT is a bit mask of tuning switches, one per stream (see FC3D_NT_* below).
#ifndef FC3D_NT
#define FC3D_NT 0x251f
#endif
#define FC3D_NT_BIT(b) ((((FC3D_NT) >> (b)) & 1) != 0)
constexpr bool kNtProjLoad = FC3D_NT_BIT(0);      // projection: scene points of the projector view
constexpr bool kNtProjStore = FC3D_NT_BIT(1);     // projection: grid_p2c
constexpr bool kNtSampLoad = FC3D_NT_BIT(2);      // sampling: scene points of the camera view
constexpr bool kNtSampGrid = FC3D_NT_BIT(3);      // sampling: grid_c2p (read again by the image-gradient pass)
constexpr bool kNtSampOut = FC3D_NT_BIT(4);       // sampling: the masked capture
constexpr bool kNtGatherLoad = FC3D_NT_BIT(5);    // image gradient: upstream gradient and grid_c2p (their last use)
constexpr bool kNtGatherStore = FC3D_NT_BIT(6);   // image gradient: d/dIp (read by the PSF backward next)
constexpr bool kNtPsfBwdLoad = FC3D_NT_BIT(7);    // PSF backward: d/dIp and layer indices (last use)
constexpr bool kNtPsfBwdStore = FC3D_NT_BIT(8);   // PSF backward: the pattern gradient
constexpr bool kNtPsfFwdLoad = FC3D_NT_BIT(9);    // PSF forward: pattern and depth
constexpr bool kNtPsfFwdStore = FC3D_NT_BIT(10);  // PSF forward: un-normalised image (read by the normalisation next)
constexpr bool kNtNormLoad = FC3D_NT_BIT(11);     // normalisation: un-normalised image (last use)
constexpr bool kNtNormStore = FC3D_NT_BIT(12);    // normalisation: the coded pattern (sampled two launches later)
constexpr bool kNtGatherImg = FC3D_NT_BIT(13);    // image gradient: the sampled image, for the by-product sum (its last use)
constexpr bool kNtCapDepth = FC3D_NT_BIT(14);     // capture, inverse warp: the camera's depth map (read once)
constexpr bool kNtCapXyz = FC3D_NT_BIT(15);       // capture, inverse warp: the camera-view scene coordinates
constexpr bool kNtCapNoise = FC3D_NT_BIT(16);     // capture, sensor model: the read-noise plane (its last use)
constexpr bool kNtCapOut = FC3D_NT_BIT(17);       // capture, sensor model: the scaled capture and its validity mask

typedef float fc3d_f4v __attribute__((ext_vector_type(4)));
typedef float fc3d_f2v __attribute__((ext_vector_type(2)));
template <bool NT>
__device__ __forceinline__ float ld_pol(const float* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT>
__device__ __forceinline__ uint8_t ld_pol(const uint8_t* p) {
    if constexpr (NT) return __builtin_nontemporal_load(p);
    else return *p;
}
template <bool NT>
__device__ __forceinline__ float2 ld_pol(const float2* p) {
    if constexpr (NT) {
        const fc3d_f2v v = __builtin_nontemporal_load(reinterpret_cast<const fc3d_f2v*>(p));
        return make_float2(v.x, v.y);
    } else {
        return *p;
    }
}
template <bool NT>
__device__ __forceinline__ float4 ld_pol(const float4* p) {
    if constexpr (NT) {
        const fc3d_f4v v = __builtin_nontemporal_load(reinterpret_cast<const fc3d_f4v*>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        return *p;
    }
}
template <bool NT>
__device__ __forceinline__ void st_pol(float* p, float v) {
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
}
template <bool NT>
__device__ __forceinline__ void st_pol(float2* p, float2 v) {
    if constexpr (NT) {
        const fc3d_f2v w = {v.x, v.y};
        __builtin_nontemporal_store(w, reinterpret_cast<fc3d_f2v*>(p));
    } else {
        *p = v;
    }
}
template <bool NT>
__device__ __forceinline__ void st_pol(float4* p, float4 v) {
    if constexpr (NT) {
        const fc3d_f4v w = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(w, reinterpret_cast<fc3d_f4v*>(p));
    } else {
        *p = v;
    }
}
// ld_at / st_at (uniform base + 32-bit element offset) with a cache policy
template <bool NT, typename T>
__device__ __forceinline__ T ld_at_pol(const T* base, unsigned elem) {
    return ld_pol<NT>(reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + elem * (unsigned)sizeof(T)));
}
template <bool NT, typename T>
__device__ __forceinline__ void st_at_pol(T* base, unsigned elem, T v) {
    st_pol<NT>(reinterpret_cast<T*>(reinterpret_cast<char*>(base) + elem * (unsigned)sizeof(T)), v);
}

template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T w = __shfl_xor(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T w = __shfl_xor(v, o, 64);
        v = w < v ? w : v;
    }
    return v;
}
// Fixed-shape butterfly: every lane ends with the same sum, and the association order
// depends only on lane ids, so the result is reproducible run to run.
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Surface-slope shading of the pattern (withReflectance, utils_pytorch.py:251-280):
// I * (0.8 cos(atan(|grad z| / 100 um)) + 0.2) with forward differences that are zero on the last
// row / column.  cos(atan(t)) is evaluated as 1/sqrt(1 + t^2): the same function without the two
// transcendental roundings, t^2 formed directly as |grad z|^2 * 1e8 and the reciprocal square root taken
// with the hardware instruction (one ulp): a handful of instructions instead of two correctly rounded
// square roots and two divisions, result within 1e-6 relative of the composed form.  (The kernels that
// fold the shading in evaluate it for every staged row, halo included.)
__device__ __forceinline__ float shade(float I, float z0, float zr, float zb, bool has_r, bool has_b) {
    const float dx = has_r ? __fsub_rn(zr, z0) : 0.0f;
    const float dy = has_b ? __fsub_rn(zb, z0) : 0.0f;
    const float t2 = __fmul_rn(__fadd_rn(__fmul_rn(dy, dy), __fmul_rn(dx, dx)), 1.0e8f);
    const float c = __builtin_amdgcn_rsqf(__fadd_rn(1.0f, t2));
    return __fmul_rn(I, __fadd_rn(__fmul_rn(c, 0.8f), 0.2f));
}

// Clears `bytes` bytes at p with an ordinary kernel launch.  (A memset call is avoided on purpose:
// replayed from a captured stream graph it was seen to leave part of its range untouched, while a
// kernel node is ordered like every other launch of the step.)
static __global__ __launch_bounds__(256) void zero_fill_kernel(uint4* __restrict__ body, long n16,
                                                               uint8_t* __restrict__ head, int nhead,
                                                               uint8_t* __restrict__ tail, int ntail) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) body[i] = make_uint4(0u, 0u, 0u, 0u);
    if (blockIdx.x == 0) {
        if ((int)threadIdx.x < nhead) head[threadIdx.x] = 0;
        if ((int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
    }
}

static inline int zero_fill(hipStream_t s, void* p, size_t bytes) {
    if (bytes == 0) return FC3D_OK;
    uint8_t* head = static_cast<uint8_t*>(p);
    size_t nhead = (16 - (reinterpret_cast<uintptr_t>(p) & 15u)) & 15u;
    if (nhead > bytes) nhead = bytes;
    uint4* body = reinterpret_cast<uint4*>(head + nhead);
    const long n16 = (long)((bytes - nhead) / 16);
    uint8_t* tail = reinterpret_cast<uint8_t*>(body + n16);
    const int ntail = (int)((bytes - nhead) % 16);
    long blocks = (n16 + 256 * 4 - 1) / (256 * 4);          // about four 16-byte stores per lane
    blocks = blocks < 1 ? 1 : (blocks > 4096 ? 4096 : blocks);
    FC3D_LAUNCH("zero_fill", zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, body, n16, head, (int)nhead, tail,
                ntail);
    return FC3D_OK;
}

}  // namespace fc3d
