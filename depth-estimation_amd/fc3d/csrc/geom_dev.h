// Device helpers shared by the geometry and capture kernels: pixel-to-lane mapping, the projection
// of a scene point, the visibility splat and the bilinear taps.  Arithmetic that decides an integer
// follows the operation order of the reference's tensor ops (see geometry.hip).
#pragma once
#include "fc3d_common.h"

namespace fc3d {

constexpr int NT = 256;

// Every lane owns PPL pixels that are 64 apart inside its wavefront's run of 64*PPL consecutive
// pixels, so each load/store/gather instruction of a wavefront touches 64 neighbouring pixels.
// Two per lane measured best at 32x512x512 (4: 45/54 us for projection/sampling, 2: 41/50, 1: 43/52).
constexpr int PPL = 2;
constexpr int PX_PER_BLOCK = NT * PPL;

__device__ __forceinline__ long lane_first_pixel() {
    return (long)blockIdx.x * PX_PER_BLOCK + (threadIdx.x >> 6) * (64 * PPL) + (threadIdx.x & 63);
}

// Row of T times a column.  torch.matmul([B,4,4],[B,4,HW]) runs a k-ordered fused
// multiply-add chain (vendor sgemm) once 16*HW >= 400, and an unfused loop below that.
template <bool FMA>
__device__ __forceinline__ float row_dot(const float* t, float c0, float c1, float c2, float c3) {
    if (FMA) {
        float a = __fmul_rn(t[0], c0);
        a = fmaf(t[1], c1, a);
        a = fmaf(t[2], c2, a);
        a = fmaf(t[3], c3, a);
        return a;
    } else {
        float a = __fmul_rn(t[0], c0);
        a = __fadd_rn(a, __fmul_rn(t[1], c1));
        a = __fadd_rn(a, __fmul_rn(t[2], c2));
        a = __fadd_rn(a, __fmul_rn(t[3], c3));
        return a;
    }
}

template <bool FMA>
__device__ __forceinline__ void project_px(const float* T, float c0, float c1, float c2, float c3, float& gx,
                                           float& gy) {
    const float xu = row_dot<FMA>(T + 0, c0, c1, c2, c3);
    const float yu = row_dot<FMA>(T + 4, c0, c1, c2, c3);
    const float zu = row_dot<FMA>(T + 8, c0, c1, c2, c3);
    const float den = __fadd_rn(zu, 1e-10f);
    gx = xu / den;
    gy = yu / den;
}

// Marks the pixels a projected point touches (gen_visible_mask).  The flat index is formed
// in fp32 and only the flat index is range-checked, exactly like the reference, so an x
// outside the row wraps into the neighbouring row.  Idempotent byte/float stores of "1"
// need no atomics and give a bit-exact, order-independent mask.
template <typename MaskT>
__device__ __forceinline__ void mark(MaskT* __restrict__ m, float xi, float yW, long HW, MaskT one) {
    const float f = __fadd_rn(xi, yW);           // integer-valued (or NaN/Inf): a sum of floor/ceil/round results
    if (HW <= (1L << 24)) {
        // every valid index is exactly representable, so the range test can stay in fp32 and the
        // index in 32 bits (NaN fails both comparisons)
        if (f >= 0.0f && f < (float)HW) m[(int)f] = one;
    } else {
        if (!(fabsf(f) < 9.0e18f)) return;       // NaN / Inf / beyond int64: the reference's cast yields a negative index
        const long idx = (long)f;
        if (idx >= 0 && idx < HW) m[idx] = one;
    }
}

template <typename MaskT>
__device__ __forceinline__ void splat_px(MaskT* __restrict__ m, float gx, float gy, int W, long HW, int method,
                                         MaskT one) {
    const float Wf = (float)W;
    if (method == 0) {
        const float fx = floorf(gx), cx = ceilf(gx);
        const float fyW = __fmul_rn(floorf(gy), Wf), cyW = __fmul_rn(ceilf(gy), Wf);
        mark(m, fx, fyW, HW, one);
        mark(m, fx, cyW, HW, one);
        mark(m, cx, fyW, HW, one);
        mark(m, cx, cyW, HW, one);
    } else {
        mark(m, rintf(gx), __fmul_rn(rintf(gy), Wf), HW, one);
    }
}

struct Taps {
    int x0, x1, y0, y1;
    float wx0, wx1, wy0, wy1;
};
// Corners are clamped first and the weights are taken from the CLAMPED corners
// (geometry_pytorch.py:136-145): out-of-image samples cancel to ~0, and an exact integer
// coordinate on the last row/column gets weight 0.  fmaxf/fminf drop a NaN so indices stay
// in range (the reference would raise on a NaN coordinate); the weights keep the NaN.
__device__ __forceinline__ Taps make_taps(float gx, float gy, int Hs, int Ws) {
    const float xm = (float)(Ws - 1), ym = (float)(Hs - 1);
    const float fx = floorf(gx), fy = floorf(gy);
    const float x0 = fminf(fmaxf(fx, 0.0f), xm), x1 = fminf(fmaxf(__fadd_rn(fx, 1.0f), 0.0f), xm);
    const float y0 = fminf(fmaxf(fy, 0.0f), ym), y1 = fminf(fmaxf(__fadd_rn(fy, 1.0f), 0.0f), ym);
    Taps t;
    t.wx0 = __fsub_rn(x1, gx);
    t.wx1 = __fsub_rn(gx, x0);
    t.wy0 = __fsub_rn(y1, gy);
    t.wy1 = __fsub_rn(gy, y0);
    t.x0 = (int)x0; t.x1 = (int)x1; t.y0 = (int)y0; t.y1 = (int)y1;
    return t;
}

__device__ __forceinline__ float blend(const Taps& t, float i00, float i01, float i10, float i11) {
    // iab = image at (x_a, y_b); same products and left-to-right sum as geometry_pytorch.py:171-176
    float o = __fmul_rn(__fmul_rn(t.wx0, t.wy0), i00);
    o = __fadd_rn(o, __fmul_rn(__fmul_rn(t.wx0, t.wy1), i01));
    o = __fadd_rn(o, __fmul_rn(__fmul_rn(t.wx1, t.wy0), i10));
    o = __fadd_rn(o, __fmul_rn(__fmul_rn(t.wx1, t.wy1), i11));
    return o;
}

static inline int check_img(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return FC3D_EINVAL;
    if ((long)B * H * W >= (1L << 31) || B > 65535 || H > 65535) return FC3D_ERANGE;
    return FC3D_OK;
}

// 16*H*W >= 400 selects torch's sgemm (fused multiply-add) path for [B,4,4]x[B,4,HW].
static inline bool use_fma(int H, int W) { return 16L * H * W >= 400; }

}  // namespace fc3d
