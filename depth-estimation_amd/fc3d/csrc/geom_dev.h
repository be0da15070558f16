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
#ifndef FC3D_PPL
#define FC3D_PPL 2
#endif
constexpr int PPL = FC3D_PPL;
constexpr int PX_PER_BLOCK = NT * PPL;

// XCD: renumber the workgroups so that each XCD walks through a contiguous part of the image.  It pays
// where neighbouring workgroups re-read each other's lines (the sampler's gather: 329 -> 279 MB per
// launch at 32x512x512, -1.2 us); the splat only streams, and measured 0.9 us slower with it.
template <bool XCD = true>
__device__ __forceinline__ long lane_first_pixel() {
    return (long)(XCD ? block_x() : blockIdx.x) * PX_PER_BLOCK + (threadIdx.x >> 6) * (64 * PPL) + (threadIdx.x & 63);
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

// ---------------------------------------------------------------------------------------
// Tap index: where the samples of a grid land in the source image, kept at three levels so that the
// image-gradient pass of the sampler can find, for a tile of the SOURCE image, every target pixel
// that may touch it:
//   segment  SEG consecutive target pixels (one row of 16 lanes),
//   run      64*PPL consecutive target pixels: what one wavefront of the streaming kernels owns,
//   group    NT*PPL consecutive target pixels: what one workgroup of them owns.
// Every level stores one box per unit: the range of the samples' base cells (x0, y0) = (floor x,
// floor y) after clamping, over the samples that contribute at all (inside the target, mask set,
// corners not clamped onto each other), packed as two signed 16-bit pairs: lo = (xmin | ymin << 16),
// hi = (xmax | ymax << 16).  An empty box has min = 32767 and max = -32768 in every field, which
// fails every overlap test.  The reductions run in the data-parallel-primitive path on packed
// 16-bit min/max.  A segment's entry is (xmin | ymin << 16, width | height << 8 | live << 16): the
// box as corner and extent (255 = that much or more) and one bit per pixel of the segment that says
// whether its sample contributes, so that the gradient pass needs neither the mask nor the clamp
// tests again.  Layout of the index of one batch: [B][groups] group boxes, [B][groups*RUN_PER_GROUP]
// run boxes, [B][groups*RUN_PER_GROUP*SEG_PER_RUN] segment boxes.
// ---------------------------------------------------------------------------------------
constexpr int SEG = 16;
constexpr int SEG_PER_RUN = 64 * PPL / SEG;
constexpr int RUN_PER_GROUP = NT / 64;
constexpr int kBoxEmptyMin = 0x7fff7fff;
constexpr int kBoxEmptyMax = (int)0x80008000;
constexpr int kTapIndexMaxDim = 32767;               // source sizes the 16-bit fields can hold

typedef short short2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int pk_min(int a, int b) {
    return __builtin_bit_cast(int, __builtin_elementwise_min(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b)));
}
__device__ __forceinline__ int pk_max(int a, int b) {
    return __builtin_bit_cast(int, __builtin_elementwise_max(__builtin_bit_cast(short2v, a), __builtin_bit_cast(short2v, b)));
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_keep(int old, int src) {   // lanes without a source lane keep `old`
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}
// lane 15 of every row of 16 lanes ends with the row's result
__device__ __forceinline__ int row_pk_min(int v) {
    v = pk_min(v, dpp_keep<0x111, 0xf>(kBoxEmptyMin, v));   // row_shr:1
    v = pk_min(v, dpp_keep<0x112, 0xf>(kBoxEmptyMin, v));   // row_shr:2
    v = pk_min(v, dpp_keep<0x114, 0xf>(kBoxEmptyMin, v));   // row_shr:4
    v = pk_min(v, dpp_keep<0x118, 0xf>(kBoxEmptyMin, v));   // row_shr:8
    return v;
}
__device__ __forceinline__ int row_pk_max(int v) {
    v = pk_max(v, dpp_keep<0x111, 0xf>(kBoxEmptyMax, v));
    v = pk_max(v, dpp_keep<0x112, 0xf>(kBoxEmptyMax, v));
    v = pk_max(v, dpp_keep<0x114, 0xf>(kBoxEmptyMax, v));
    v = pk_max(v, dpp_keep<0x118, 0xf>(kBoxEmptyMax, v));
    return v;
}
// from per-row results in lanes 15, 31, 47, 63 to the wavefront's result in lane 63
__device__ __forceinline__ int rows_pk_min(int v) {
    v = pk_min(v, dpp_keep<0x142, 0xa>(kBoxEmptyMin, v));   // row_bcast:15 into rows 1, 3
    v = pk_min(v, dpp_keep<0x143, 0xc>(kBoxEmptyMin, v));   // row_bcast:31 into rows 2, 3
    return v;
}
__device__ __forceinline__ int rows_pk_max(int v) {
    v = pk_max(v, dpp_keep<0x142, 0xa>(kBoxEmptyMax, v));
    v = pk_max(v, dpp_keep<0x143, 0xc>(kBoxEmptyMax, v));
    return v;
}

__host__ __device__ static inline long tap_index_groups(int Ht, int Wt) {
    return ((long)Ht * Wt + PX_PER_BLOCK - 1) / PX_PER_BLOCK;
}
struct TapIndex {            // the three levels of one image (or, with b = 0, the start of a batch's arrays)
    int2 *group, *run, *seg;
};
__host__ __device__ static inline TapIndex tap_index_at(void* base, int B, long groups, int b) {
    TapIndex t;
    int2* p = static_cast<int2*>(base);
    t.group = p + (long)b * groups;
    t.run = p + (long)B * groups + (long)b * groups * RUN_PER_GROUP;
    t.seg = p + (long)B * groups * (1 + RUN_PER_GROUP) + (long)b * groups * RUN_PER_GROUP * SEG_PER_RUN;
    return t;
}

// Every thread of a workgroup of the streaming kernels calls this with its PPL samples (pixel
// lane_first_pixel() + 64*i for slot i; on[i] false for samples that contribute nothing, pixels
// past the image included).  Writes the boxes of the workgroup's group, runs and segments.
// Contains a workgroup barrier.
__device__ __forceinline__ void emit_tap_index(const TapIndex& ix, const int x0[PPL], const int y0[PPL],
                                               const bool on[PPL]) {
    __shared__ int2 s_runbox[RUN_PER_GROUP];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long run = (long)block_x() * RUN_PER_GROUP + wv;
    int wmin = kBoxEmptyMin, wmax = kBoxEmptyMax;
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const int pk = x0[i] | (y0[i] << 16);
        const int mn = row_pk_min(on[i] ? pk : kBoxEmptyMin);
        const int mx = row_pk_max(on[i] ? pk : kBoxEmptyMax);
        const unsigned long long live = __ballot(on[i]);
        if ((lane & 15) == 15) {
            const unsigned bits = (unsigned)(live >> (lane & 48)) & 0xffffu;
            const int w = min((short)(mx & 0xffff) - (short)(mn & 0xffff), 255), h = min((mx >> 16) - (mn >> 16), 255);
            ix.seg[run * SEG_PER_RUN + i * (64 / SEG) + (lane >> 4)] =
                make_int2(mn, bits ? (int)((unsigned)w | ((unsigned)h << 8) | (bits << 16)) : 0);
        }
        wmin = pk_min(wmin, mn);
        wmax = pk_max(wmax, mx);
    }
    wmin = rows_pk_min(wmin);
    wmax = rows_pk_max(wmax);
    if (lane == 63) {
        ix.run[run] = make_int2(wmin, wmax);
        s_runbox[wv] = make_int2(wmin, wmax);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int gmin = s_runbox[0].x, gmax = s_runbox[0].y;
#pragma unroll
        for (int w = 1; w < RUN_PER_GROUP; ++w) {
            gmin = pk_min(gmin, s_runbox[w].x);
            gmax = pk_max(gmax, s_runbox[w].y);
        }
        ix.group[block_x()] = make_int2(gmin, gmax);
    }
}

// does a segment's entry meet the base-cell range [bx0, bx1] x [by0, by1]?
__device__ __forceinline__ bool seg_meets(int2 m, int bx0, int bx1, int by0, int by1) {
    const int xmin = (short)(m.x & 0xffff), ymin = m.x >> 16, w = m.y & 0xff, h = (m.y >> 8) & 0xff;
    return (m.y != 0) && xmin <= bx1 && ymin <= by1 && (w == 255 || xmin + w >= bx0) && (h == 255 || ymin + h >= by0);
}

// does box (lo, hi) meet the base-cell range [bx0, bx1] x [by0, by1]?
__device__ __forceinline__ bool box_meets(int2 m, int bx0, int bx1, int by0, int by1) {
    const int xmin = (short)(m.x & 0xffff), ymin = m.x >> 16, xmax = (short)(m.y & 0xffff), ymax = m.y >> 16;
    return xmin <= bx1 && xmax >= bx0 && ymin <= by1 && ymax >= by0;
}

// invert4 spread over the lanes of one wavefront: lane r*8 + j (upper half: a copy of the lower)
// holds tableau entry m[r][j].  Every entry sees the operations invert4 applies to it, in the same
// order, so the rounded inverse is the same; a lane needs one double instead of thirty-two, which
// lets a streaming kernel invert its sample's pose on the side without giving up resident wavefronts.
__device__ __forceinline__ double invert4_lanes(const float* __restrict__ a, int lane) {
    const int l = lane & 31, r = l >> 3, j = l & 7;
    double m = j < 4 ? (double)a[r * 4 + j] : (j - 4 == r ? 1.0 : 0.0);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int r2 = c + 1; r2 < 4; ++r2) {        // the largest |m[r][c]|, r >= c, ends up in row c (first one on ties)
            const double pc = __shfl(m, c * 8 + c, 64), pr = __shfl(m, r2 * 8 + c, 64);
            const int partner = (r == c ? r2 : (r == r2 ? c : r)) * 8 + j;
            const double other = __shfl(m, partner, 64);
            if (fabs(pr) > fabs(pc)) m = other;
        }
        const double d = 1.0 / __shfl(m, c * 8 + c, 64);
        if (r == c) m *= d;
        const double f = __shfl(m, r * 8 + c, 64);
        const double mc = __shfl(m, c * 8 + j, 64);
        if (r != c) m -= f * mc;
    }
    return m;
}

// Element (i, j) = lane & 15 of K*A, the 4x4 product as torch's small-matrix loop rounds it
// (intrinsics_times), with A's element (k, j) supplied by the caller's accessor.
template <typename Acol>
__device__ __forceinline__ float intrinsics_times_lane(const float* __restrict__ K, int i, Acol a_kj) {
    float s = __fmul_rn(K[i * 4 + 0], a_kj(0));
    s = __fadd_rn(s, __fmul_rn(K[i * 4 + 1], a_kj(1)));
    s = __fadd_rn(s, __fmul_rn(K[i * 4 + 2], a_kj(2)));
    s = __fadd_rn(s, __fmul_rn(K[i * 4 + 3], a_kj(3)));
    return s;
}

// What pose_transforms_kernel computes for one sample, by one wavefront, on the side of another kernel.
struct PoseArgs {
    const float* pose;      // [B][16]
    const float* K;         // K[b*k_sb + 4*i + j]
    long k_sb;
    float* t_fwd;           // [B][16] K*pose
    float* t_inv;           // [B][16] K*inverse(pose)
    float* inv;             // [B][16] inverse(pose), may be null
};
__device__ __forceinline__ void pose_outputs_wave(const PoseArgs& pa, int b, int lane, float t_fwd_e) {
    const float* A = pa.pose + (long)b * 16;
    const float* Kb = pa.K + (long)b * pa.k_sb;
    const int e = lane & 15, i = e >> 2, j = e & 3;
    const double m = invert4_lanes(A, lane);
    float af[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) af[k] = (float)__shfl(m, k * 8 + 4 + j, 64);
    const float own = (float)__shfl(m, i * 8 + 4 + j, 64);
    const float ti = intrinsics_times_lane(Kb, i, [&](int k) { return af[k]; });
    if (lane < 16) {
        pa.t_fwd[(long)b * 16 + e] = t_fwd_e;
        pa.t_inv[(long)b * 16 + e] = ti;
        if (pa.inv) pa.inv[(long)b * 16 + e] = own;
    }
}

// T = K*pose of sample b in every lane (uniform values), formed by the wavefront itself: two loads per lane
// (element lane & 15 of either matrix), the row of K and the column of pose a lane needs come from its
// neighbours, the sixteen results are handed round as scalars.  The first wavefront of the workgroup with
// blockIdx.x == 0 also leaves the transforms in memory (pose_outputs_wave).
__device__ __forceinline__ void form_transform(const PoseArgs& pa, int b, float T[16]) {
    const int lane = threadIdx.x & 63, e = lane & 15, i = e >> 2, j = e & 3;
    const float kv = pa.K[(long)b * pa.k_sb + e], av = pa.pose[(long)b * 16 + e];
    float kr[4], ac[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        kr[k] = __shfl(kv, i * 4 + k, 64);
        ac[k] = __shfl(av, k * 4 + j, 64);
    }
    const float te = intrinsics_times_lane(kr, 0, [&](int k) { return ac[k]; });
#pragma unroll
    for (int k = 0; k < 16; ++k) T[k] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(te), k));
    if (blockIdx.x == 0 && threadIdx.x < 64) pose_outputs_wave(pa, b, lane, te);
}

static inline int check_img(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return FC3D_EINVAL;
    if ((long)B * H * W >= (1L << 31) || B > 65535 || H > 65535) return FC3D_ERANGE;
    return FC3D_OK;
}

// 16*H*W >= 400 selects torch's sgemm (fused multiply-add) path for [B,4,4]x[B,4,HW].
static inline bool use_fma(int H, int W) { return 16L * H * W >= 400; }

}  // namespace fc3d
