// Pose reprojection, bilinear warp, forward-splat visibility and compositing
// (geometry_pytorch.py) for gfx950.  All kernels are HBM-streaming: every lane owns four
// neighbouring pixels of a row so that dense planes move as 16-byte accesses, the tiny
// per-sample matrices sit in scalar registers, and the only irregular traffic is the
// 4-tap gather (served by L1/L2 because warps are locally smooth) and the 1-byte splat.
//
// Arithmetic that decides an integer (floor/ceil/round of a projected coordinate) follows
// the operation order of the reference's tensor ops exactly; explicit __fmul_rn/__fadd_rn
// and fmaf keep the compiler from re-associating or contracting it.
#include "geom_dev.h"
#include <atomic>

namespace fc3d {

// ---------------------------------------------------------------------------------------
// per-sample matrices
// ---------------------------------------------------------------------------------------
// Gauss-Jordan with partial pivoting in fp64, rounded to fp32 once by the caller.  Every loop
// bound and every index is a compile-time constant (the pivot row is brought up by compare-and-swap
// of whole rows instead of an indexed exchange), so the 4 x 8 tableau lives in registers: an
// indexed private array would sit in scratch memory and give every lane of the launch a scratch
// allocation for the sake of the B lanes that invert.
__device__ inline bool invert4(const float* a, double* inv) {
    double m[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            m[i][j] = (double)a[i * 4 + j];
            m[i][4 + j] = (i == j) ? 1.0 : 0.0;
        }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int r = c + 1; r < 4; ++r) {           // the largest |m[r][c]|, r >= c, ends up in row c (first one on ties)
            const bool up = fabs(m[r][c]) > fabs(m[c][c]);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const double t = m[c][j];
                m[c][j] = up ? m[r][j] : t;
                m[r][j] = up ? t : m[r][j];
            }
        }
        const double d = 1.0 / m[c][c];
#pragma unroll
        for (int j = 0; j < 8; ++j) m[c][j] *= d;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r != c) {
                const double f = m[r][c];
#pragma unroll
                for (int j = 0; j < 8; ++j) m[r][j] -= f * m[c][j];
            }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) inv[i * 4 + j] = m[i][4 + j];
    return true;
}

// K*A for the 4x4 product torch evaluates with its small-matrix loop: products rounded
// separately and summed left to right (no fused multiply-add).
__device__ inline void intrinsics_times(const float K[16], const float* A, float* out) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = __fmul_rn(K[i * 4 + 0], A[0 * 4 + j]);
            s = __fadd_rn(s, __fmul_rn(K[i * 4 + 1], A[1 * 4 + j]));
            s = __fadd_rn(s, __fmul_rn(K[i * 4 + 2], A[2 * 4 + j]));
            s = __fadd_rn(s, __fmul_rn(K[i * 4 + 3], A[3 * 4 + j]));
            out[i * 4 + j] = s;
        }
}

// The launch can also zero-fill a buffer (grid-stride, 16 bytes per lane and step): warp_p2c needs
// its visibility mask cleared before the splat, and a separate fill would cost one more launch
// than the few microseconds of work either of the two is.
__global__ void pose_transforms_kernel(const float* __restrict__ pose, const float* __restrict__ Kin, long k_sb,
                                       int B, float* __restrict__ t_fwd, float* __restrict__ t_inv,
                                       float* __restrict__ inv_out, uint4* __restrict__ clear16, long n16,
                                       uint8_t* __restrict__ clear_tail, int ntail) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    for (long i = b; i < n16; i += (long)gridDim.x * blockDim.x) clear16[i] = make_uint4(0u, 0u, 0u, 0u);
    if (b < ntail) clear_tail[b] = 0;
    if (b >= B) return;
    float K[16];
    for (int i = 0; i < 16; ++i) K[i] = Kin ? Kin[(long)b * k_sb + i] : 0.0f;
    float A[16];
    for (int i = 0; i < 16; ++i) A[i] = pose[(long)b * 16 + i];
    float o[16];
    if (t_fwd) {
        intrinsics_times(K, A, o);
        for (int i = 0; i < 16; ++i) t_fwd[(long)b * 16 + i] = o[i];
    }
    if (t_inv || inv_out) {
        double inv[16];
        invert4(A, inv);
        float Af[16];
        for (int i = 0; i < 16; ++i) Af[i] = (float)inv[i];
        if (inv_out)
            for (int i = 0; i < 16; ++i) inv_out[(long)b * 16 + i] = Af[i];
        if (t_inv) {
            intrinsics_times(K, Af, o);
            for (int i = 0; i < 16; ++i) t_inv[(long)b * 16 + i] = o[i];
        }
    }
}

// ---------------------------------------------------------------------------------------
// projection
// ---------------------------------------------------------------------------------------
#ifndef FC3D_SAMP_DBG
#define FC3D_SAMP_DBG 0       // timing-only builds of the sampling kernel (1: no tap index, 2: no grid store, 4: mask not cleared)
#endif
enum CoordLayout { kStrided = 0, kPlanarFlat = 1, kPixel4Flat = 2, kFromDepth = 3 };
// kPlanarFlat : dense planes, component k of flat pixel p at base[k*sc + p]
// kPixel4Flat : dense [H][W][4] (what z2pointcloud's permuted view points at): one aligned float4 per pixel
// kStrided    : any other view; (y, x) are recovered from p
// kFromDepth  : no stored points at all: base is the depth plane the points were made from and the point of
//               pixel (x, y) is formed here as points_in_camera_coords stores it, (z rx[x], z ry[y], z 1, 1)
//               with separately rounded products -- 4 bytes per pixel read instead of 16

// P = unsigned: the caller vouches for 16 * H * W < 2^32, and the flat layouts then address with a uniform base
// and a 32-bit lane offset (one instruction instead of three for a 64-bit lane address).
template <int LAYOUT, bool NT = false, typename P = long>
__device__ __forceinline__ void load_coord(const float* __restrict__ base, long sc, long sh, long sw, P p, int W,
                                           float c[4], const float* __restrict__ ray_x = nullptr,
                                           const float* __restrict__ ray_y = nullptr) {
    constexpr bool SMALL = sizeof(P) == 4;
    if (LAYOUT == kFromDepth) {
        // loads only (ray_x[x], ray_y[y], z): the products wait in finish_coord, so that the loads of all of a
        // lane's pixels are in flight together
        const unsigned y = (unsigned)p / (unsigned)W, x = (unsigned)p - y * (unsigned)W;
        c[0] = ld_at(ray_x, x);
        c[1] = ld_at(ray_y, y);
        c[2] = SMALL ? ld_at_pol<NT>(base, (unsigned)p) : ld_pol<NT>(base + p);
        c[3] = 1.0f;
    } else if (LAYOUT == kPixel4Flat) {
        const float4 v = SMALL ? ld_at_pol<NT>(reinterpret_cast<const float4*>(base), (unsigned)p)
                               : ld_pol<NT>(reinterpret_cast<const float4*>(base + p * 4));
        c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
    } else if (LAYOUT == kPlanarFlat) {
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = base[k * sc + p];
    } else {
        const long y = p / W, x = p - y * W;
#pragma unroll
        for (int k = 0; k < 4; ++k) c[k] = base[k * sc + y * sh + x * sw];
    }
}

// what load_coord left to do once the loads have arrived
template <int LAYOUT>
__device__ __forceinline__ void finish_coord(float c[4]) {
    if (LAYOUT == kFromDepth) {
        const float z = c[2];
        c[0] = __fmul_rn(z, c[0]);
        c[1] = __fmul_rn(z, c[1]);
        c[2] = __fmul_rn(z, 1.0f);
    }
}

// coord_transform; with SPLAT also the 'all' visibility splat of warp_p2c's forward warp.
// With POSE the transform is not read but formed here, T = K*pose (every wavefront for itself: one
// element per lane, handed round as scalars), and the first wavefront of every sample also leaves
// K*pose, K*inverse(pose) and inverse(pose) in memory for the kernels behind this one -- what
// pose_transforms_kernel would do in a launch of its own.
template <int LAYOUT, bool FMA, bool SPLAT, bool POSE = false>
__global__ __launch_bounds__(NT) void project_kernel(const float* __restrict__ coord, long sb, long sc, long sh,
                                                     long sw, const float* __restrict__ Tm, int H, int W,
                                                     float* __restrict__ grid, uint8_t* __restrict__ mask_u8,
                                                     PoseArgs pa, const float* __restrict__ ray_x,
                                                     const float* __restrict__ ray_y) {
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    const long p0 = lane_first_pixel<false>();
    const float* cb = coord + (long)b * sb;
    float c[PPL][4];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const long p = p0 + 64 * i;
        if (p < HW) load_coord<LAYOUT, kNtProjLoad>(cb, sc, sh, sw, p, W, c[i], ray_x, ray_y);
    }
    // after the coordinate loads were issued: the few words of K and pose arrive under their latency
    float T[16];
    if (POSE) {
        form_transform(pa, b, T);
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) T[k] = Tm[(long)b * 16 + k];
    }
    float2* go = reinterpret_cast<float2*>(grid) + (long)b * HW;
    uint8_t* m = SPLAT ? mask_u8 + (long)b * HW : nullptr;
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const long p = p0 + 64 * i;
        if (p < HW) {
            float gx, gy;
            finish_coord<LAYOUT>(c[i]);
            project_px<FMA>(T, c[i][0], c[i][1], c[i][2], c[i][3], gx, gy);
            st_pol<kNtProjStore>(go + p, make_float2(gx, gy));
            if (SPLAT) splat_px<uint8_t>(m, gx, gy, W, HW, 0, (uint8_t)1);
        }
    }
}

// ---------------------------------------------------------------------------------------
// bilinear sampling
// ---------------------------------------------------------------------------------------
// warp_p2c's inverse warp: project coord_c, sample the single-channel pattern, composite
// with the visibility mask.
template <int LAYOUT, bool FMA>
__global__ __launch_bounds__(NT) void warp_sample_kernel(const float* __restrict__ coord, long sb, long sc, long sh,
                                                         long sw, const float* __restrict__ Tm,
                                                         const float* __restrict__ Ip,
                                                         uint8_t* mask_u8, int rezero, int H, int W,
                                                         float* __restrict__ grid, float* __restrict__ out,
                                                         void* __restrict__ tap_index, int B,
                                                         const float* __restrict__ ray_x,
                                                         const float* __restrict__ ray_y) {
    const int b = blockIdx.y;
    // offsets inside an image are 32-bit (the entry point refuses H*W >= 2^29): every access below is a uniform
    // 64-bit base plus a 32-bit lane offset, which costs one instruction where 64-bit lane addresses cost three
    const unsigned HW = (unsigned)H * (unsigned)W;
    const unsigned p0 = (unsigned)lane_first_pixel();
    const float* T = Tm + (long)b * 16;
    const float* cb = coord + (long)b * sb;
    const float* img = Ip + (long)b * HW;
    uint8_t* mk = mask_u8 + (long)b * HW;
    float c[PPL][4];
    float mv[PPL];
    uint8_t mb[PPL];          // the mask bytes as loaded: converted where they are used, so that no load waits for a conversion
    // No branch around a load: a pixel past the image reads the last pixel's operands instead (and stores
    // nothing).  Behind a branch the number of loads in flight is unknown at the join, and the next pixel's
    // taps would be issued only after this pixel's had returned -- one memory latency per pixel.
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const unsigned pc = min(p0 + 64u * i, HW - 1u);
        load_coord<LAYOUT, kNtSampLoad>(cb, sc, sh, sw, pc, W, c[i], ray_x, ray_y);
        mb[i] = ld_at(mk, pc);
    }
    float2* go = reinterpret_cast<float2*>(grid) + (long)b * HW;
    float* oo = out + (long)b * HW;
    float gx[PPL], gy[PPL], i00[PPL], i01[PPL], i10[PPL], i11[PPL];
    Taps t[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        mv[i] = (float)mb[i];
        finish_coord<LAYOUT>(c[i]);
        project_px<FMA>(T, c[i][0], c[i][1], c[i][2], c[i][3], gx[i], gy[i]);
        t[i] = make_taps(gx[i], gy[i], H, W);
        const unsigned r0 = (unsigned)t[i].y0 * (unsigned)W, r1 = (unsigned)t[i].y1 * (unsigned)W;
        i00[i] = ld_at(img, r0 + (unsigned)t[i].x0);
        i01[i] = ld_at(img, r1 + (unsigned)t[i].x0);
        i10[i] = ld_at(img, r0 + (unsigned)t[i].x1);
        i11[i] = ld_at(img, r1 + (unsigned)t[i].x1);
    }
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const unsigned p = p0 + 64u * i;
        if (p < HW) {
#if !(FC3D_SAMP_DBG & 2)
            st_at_pol<kNtSampGrid>(go, p, make_float2(gx[i], gy[i]));
#endif
            st_at_pol<kNtSampOut>(oo, p, blend(t[i], i00[i], i01[i], i10[i], i11[i]) * mv[i]);
        }
    }
    if (rezero && !(FC3D_SAMP_DBG & 4)) {
        // The mask leaves the launch cleared for its next use.  A wavefront's 64*PPL bytes are read by
        // its own lanes only, and those loads have returned (their values went into the stores above),
        // so a few lanes can clear the run with 16-byte stores; byte stores where the run is ragged.
        const unsigned lane = threadIdx.x & 63;
        const unsigned run0 = p0 - lane;
        if (rezero == 2 && run0 + 64u * PPL <= HW) {
            if (lane < 64 * PPL / 16) *reinterpret_cast<uint4*>(mk + run0 + lane * 16) = make_uint4(0u, 0u, 0u, 0u);
        } else {
#pragma unroll
            for (int i = 0; i < PPL; ++i)
                if (p0 + 64u * i < HW) st_at(mk, p0 + 64u * i, (uint8_t)0);
        }
    }
    if (tap_index && !(FC3D_SAMP_DBG & 1)) {   // for the gradient w.r.t. Ip: where this workgroup's pixels sampled the pattern
        int bx[PPL], by[PPL];
        bool on[PPL];
#pragma unroll
        for (int i = 0; i < PPL; ++i) {
            const bool in = p0 + 64 * i < HW;
            bx[i] = in ? t[i].x0 : 0;
            by[i] = in ? t[i].y0 : 0;
            on[i] = in && mv[i] != 0.0f && t[i].x0 != t[i].x1 && t[i].y0 != t[i].y1;
        }
        emit_tap_index(tap_index_at(tap_index, B, gridDim.x, b), bx, by, on);
    }
}

// bilinear_sampler for any channel count; one lane per target pixel.
__global__ __launch_bounds__(NT) void bilinear_fwd_kernel(const float* __restrict__ img,
                                                          const float* __restrict__ grid,
                                                          const uint8_t* __restrict__ mask_u8, int Hs, int Ws, int C,
                                                          long npix_t, float* __restrict__ out) {
    const int b = blockIdx.y;
    const long p = (long)blockIdx.x * NT + threadIdx.x;
    if (p >= npix_t) return;
    const long q = (long)b * npix_t + p;
    const float2 g = *reinterpret_cast<const float2*>(grid + q * 2);
    const Taps t = make_taps(g.x, g.y, Hs, Ws);
    const float* im = img + (long)b * Hs * Ws * C;
    const long a00 = ((long)t.y0 * Ws + t.x0) * C, a01 = ((long)t.y1 * Ws + t.x0) * C;
    const long a10 = ((long)t.y0 * Ws + t.x1) * C, a11 = ((long)t.y1 * Ws + t.x1) * C;
    const float mk = mask_u8 ? (float)mask_u8[q] : 1.0f;
    for (int c = 0; c < C; ++c) {
        float o = blend(t, im[a00 + c], im[a01 + c], im[a10 + c], im[a11 + c]);
        if (mask_u8) o = o * mk;
        out[q * C + c] = o;
    }
}

// d out / d grid.  floor and clamp carry no gradient, so only the four weights depend on the
// coordinate: d/dx = wy0*(i10-i00) + wy1*(i11-i01), d/dy = wx0*(i01-i00) + wx1*(i11-i10).
__global__ __launch_bounds__(NT) void bilinear_bwd_grid_kernel(const float* __restrict__ g_out,
                                                               const float* __restrict__ img,
                                                               const float* __restrict__ grid,
                                                               const uint8_t* __restrict__ mask_u8, int Hs, int Ws,
                                                               int C, long npix_t, float* __restrict__ g_grid) {
    const int b = blockIdx.y;
    const long p = (long)blockIdx.x * NT + threadIdx.x;
    if (p >= npix_t) return;
    const long q = (long)b * npix_t + p;
    const float2 g = *reinterpret_cast<const float2*>(grid + q * 2);
    const Taps t = make_taps(g.x, g.y, Hs, Ws);
    const float* im = img + (long)b * Hs * Ws * C;
    const long a00 = ((long)t.y0 * Ws + t.x0) * C, a01 = ((long)t.y1 * Ws + t.x0) * C;
    const long a10 = ((long)t.y0 * Ws + t.x1) * C, a11 = ((long)t.y1 * Ws + t.x1) * C;
    const float mk = mask_u8 ? (float)mask_u8[q] : 1.0f;
    float dx = 0.0f, dy = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float go = g_out[q * C + c] * mk;
        const float i00 = im[a00 + c], i01 = im[a01 + c], i10 = im[a10 + c], i11 = im[a11 + c];
        dx = fmaf(go, t.wy0 * (i10 - i00) + t.wy1 * (i11 - i01), dx);
        dy = fmaf(go, t.wx0 * (i01 - i00) + t.wx1 * (i11 - i10), dy);
    }
    *reinterpret_cast<float2*>(g_grid + q * 2) = make_float2(dx, dy);
}

// ---------------------------------------------------------------------------------------
// d out / d img for C == 1.
//
// Default (tap index given): gather form, no atomics, no zero fill, fixed summation order.
//   Every wavefront owns a GX x GY tile of the SOURCE image and keeps its sums in a private LDS
//   window (the tile plus a one-cell apron, so that the four taps of a sample whose base cell lies
//   in [x0-1, x0+GX-1] x [y0-1, y0+GY-1] all fall inside).  It looks up in the tap index which
//   segments of SEG target pixels can touch its tile, loads those segments (four per step, one
//   sample per lane), and adds the taps of the samples whose base cell is in range; at the end it
//   stores the tile (not the apron) with plain 16-byte stores.  A tap that falls into the apron of
//   one tile is added again by the tile that owns the cell, so every cell is written exactly once,
//   by one wavefront, with every contribution in it.  A segment near a tile border is loaded by two
//   or four wavefronts: at 64 x 12 tiles about 1.5 loads per sample, served by L2.
//   LDS float atomics retire about one lane every three cycles on this chip, so the window is
//   updated with plain read-add-write; lanes of one instruction that hit the same cell are
//   serialised by an election through a byte-sized owner array.
// Without an index (workspace == NULL): scatter form.  Each wavefront owns a 32x8 tile of target
//   pixels, sums their taps in an LDS window of the tile's bounding box and adds the window to the
//   zero-filled gradient image with global float atomics (about one per pixel instead of four;
//   last-bit differences from run to run).
//
// A sample whose two x corners (or two y corners) were clamped onto the same source pixel
// contributes +w and -w to that pixel, i.e. exactly nothing; such samples are skipped.
// ---------------------------------------------------------------------------------------
constexpr int WT_X = 32, WT_Y = 8;       // scatter form: wavefront tile of target pixels
constexpr int WT_LX = WT_X / 4;          // lanes along x
constexpr int WIN_CAP = 1024;            // cells per window

struct PxTaps {
    Taps t;
    float v;
    bool on;
};

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_move(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, ROW_MASK, 0xf, false);
}
// Wave-wide min/max through the data-parallel-primitive path (no LDS traffic): prefix-doubling
// inside each row of 16 lanes, then row broadcasts; lane 63 ends up with the full result.
__device__ __forceinline__ int wave_min_dpp(int v) {
    const int id = 0x7fffffff;
    v = min(v, dpp_move<0x111, 0xf>(id, v));   // row_shr:1
    v = min(v, dpp_move<0x112, 0xf>(id, v));   // row_shr:2
    v = min(v, dpp_move<0x114, 0xf>(id, v));   // row_shr:4
    v = min(v, dpp_move<0x118, 0xf>(id, v));   // row_shr:8
    v = min(v, dpp_move<0x142, 0xa>(id, v));   // row_bcast:15 into rows 1,3
    v = min(v, dpp_move<0x143, 0xc>(id, v));   // row_bcast:31 into rows 2,3
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_max_dpp(int v) {
    const int id = (int)0x80000000;
    v = max(v, dpp_move<0x111, 0xf>(id, v));
    v = max(v, dpp_move<0x112, 0xf>(id, v));
    v = max(v, dpp_move<0x114, 0xf>(id, v));
    v = max(v, dpp_move<0x118, 0xf>(id, v));
    v = max(v, dpp_move<0x142, 0xa>(id, v));
    v = max(v, dpp_move<0x143, 0xc>(id, v));
    return __builtin_amdgcn_readlane(v, 63);
}

__device__ __forceinline__ void load_tile_taps(const float* __restrict__ g_out, const float* __restrict__ grid,
                                               const uint8_t* __restrict__ mask_u8, long tb, int Ht, int Wt, int Hs,
                                               int Ws, int tx0, int ty0, int lane, PxTaps px[4]) {
    // all loads up front, unconditionally, from clamped (always valid) positions: one memory
    // latency per tile; a load inside an `if` would cost one each
    const float2* gb = reinterpret_cast<const float2*>(grid) + tb;
    const float* vb = g_out + tb;
    const uint8_t* mb = mask_u8 ? mask_u8 + tb : nullptr;
    float2 g[4];
    float gv[4];
    uint8_t mk[4];
    bool vis[4];
    {
        // a lane owns four neighbouring pixels of one row: one 16-byte load of g, two of the grid, one
        // 4-byte load of the mask
        const int x = tx0 + 4 * (lane % WT_LX), y = ty0 + lane / WT_LX;
        const bool vec = (Wt % 4 == 0) && ((reinterpret_cast<uintptr_t>(g_out) | reinterpret_cast<uintptr_t>(grid)) % 16 == 0) &&
                         (!mask_u8 || reinterpret_cast<uintptr_t>(mask_u8) % 4 == 0);
        if (vec) {
            const unsigned q = (unsigned)(min(y, Ht - 1) * Wt + min(x, Wt - 4));
            const float4 a = ld_at(reinterpret_cast<const float4*>(vb), q >> 2);
            const float4 g0 = ld_at(reinterpret_cast<const float4*>(gb), q >> 1);
            const float4 g1 = ld_at(reinterpret_cast<const float4*>(gb), (q >> 1) + 1);
            uchar4 m4 = make_uchar4(1, 1, 1, 1);
            if (mb) m4 = ld_at(reinterpret_cast<const uchar4*>(mb), q >> 2);
            gv[0] = a.x; gv[1] = a.y; gv[2] = a.z; gv[3] = a.w;
            g[0] = make_float2(g0.x, g0.y); g[1] = make_float2(g0.z, g0.w);
            g[2] = make_float2(g1.x, g1.y); g[3] = make_float2(g1.z, g1.w);
            mk[0] = m4.x; mk[1] = m4.y; mk[2] = m4.z; mk[3] = m4.w;
#pragma unroll
            for (int k = 0; k < 4; ++k) vis[k] = (x + k < Wt) && (y < Ht);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                vis[k] = (x + k < Wt) && (y < Ht);
                const unsigned q = (unsigned)(min(y, Ht - 1) * Wt + min(x + k, Wt - 1));
                g[k] = ld_at(gb, q);
                gv[k] = ld_at(vb, q);
                mk[k] = mb ? ld_at(mb, q) : (uint8_t)1;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        vis[k] = vis[k] && mk[k] != 0;
        px[k].v = vis[k] ? gv[k] : 0.0f;
        px[k].t = make_taps(g[k].x, g[k].y, Hs, Ws);
        px[k].on = vis[k] && (px[k].t.x0 != px[k].t.x1) && (px[k].t.y0 != px[k].t.y1);
    }
}

// Adds the four taps of one sample per lane into the window; lanes that aim at the same cells
// take turns.  A lane announces itself by writing its id into a byte array at its base cell and
// reads it back: LDS executes a wavefront's instructions in order and the last writer of one
// store instruction wins, so exactly one lane per contested cell sees its own id, adds, and the
// others retry (straight-line code in the common case of no collision).  A sample that is kept
// has x1 = x0 + 1 and y1 = y0 + 1, so its cells are c, c+1, c+ww, c+ww+1, and lanes with distinct
// base cells c have distinct cells in each of the four add instructions: one election on c
// covers all four.  The adds stay separate instructions in program order because lane i's c+1
// may be lane j's c.
__device__ __forceinline__ void window_add4(lds_vfloat win, lds_vu8 own, int c, int ww, float v00,
                                            float v10, float v01, float v11, bool active, int lane) {
    bool pend = active;
    do {
        if (pend) {
            own[c] = (uint8_t)lane;
            if (own[c] == (uint8_t)lane) {
                win[c] = win[c] + v00;
                win[c + 1] = win[c + 1] + v10;
                win[c + ww] = win[c + ww] + v01;
                win[c + ww + 1] = win[c + ww + 1] + v11;
                pend = false;
            }
        }
    } while (__builtin_expect(__ballot(pend) != 0, 0));
}

__global__ __launch_bounds__(NT) void bilinear_bwd_img_scatter_kernel(
    const float* __restrict__ g_out, const float* __restrict__ grid, const uint8_t* __restrict__ mask_u8, int Hs,
    int Ws, int Ht, int Wt, int tiles_x, int ntiles, float* __restrict__ g_img) {
    __shared__ __align__(16) float s_win[NT / 64][WIN_CAP];
    __shared__ uint8_t s_own[NT / 64][WIN_CAP];
    const int b = blockIdx.y;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * (NT / 64) + wv;
    if (t >= ntiles) return;                       // no workgroup barrier below: wavefronts are independent
    const int tx0 = (t % tiles_x) * WT_X, ty0 = (t / tiles_x) * WT_Y;
    PxTaps px[4];
    load_tile_taps(g_out, grid, mask_u8, (long)b * Ht * Wt, Ht, Wt, Hs, Ws, tx0, ty0, lane, px);
    int xmin = 0x7fffffff, ymin = 0x7fffffff, xmax = -1, ymax = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (px[k].on) {
            xmin = min(xmin, px[k].t.x0); xmax = max(xmax, px[k].t.x1);
            ymin = min(ymin, px[k].t.y0); ymax = max(ymax, px[k].t.y1);
        }
    xmin = wave_min_dpp(xmin); ymin = wave_min_dpp(ymin);
    xmax = wave_max_dpp(xmax); ymax = wave_max_dpp(ymax);
    const bool empty = xmax < 0;
    const int ww = empty ? 0 : xmax - xmin + 1, wh = empty ? 0 : ymax - ymin + 1;
    const bool fits = (long)ww * wh <= WIN_CAP;
    const int n = fits ? ww * wh : 0;
    lds_vfloat w = lds_ptr(s_win[wv]);
    lds_vu8 own = lds_ptr(s_own[wv]);
    {   // clear the window, 16 bytes per lane and step (a few cells past n do no harm: n <= WIN_CAP)
        float4* w4 = reinterpret_cast<float4*>(s_win[wv]);
        for (int i = lane; 4 * i < n; i += 64) w4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    if (n > 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const Taps& tp = px[k].t;
            const bool on = px[k].on;
            const int c00 = on ? (tp.y0 - ymin) * ww + (tp.x0 - xmin) : 0;
            window_add4(w, own, c00, ww, (tp.wx0 * tp.wy0) * px[k].v, (tp.wx1 * tp.wy0) * px[k].v,
                        (tp.wx0 * tp.wy1) * px[k].v, (tp.wx1 * tp.wy1) * px[k].v, on, lane);
        }
    }
    float* o = g_img + (long)b * Hs * Ws;
    if (n > 0) {
        const unsigned o0 = (unsigned)(ymin * Ws + xmin);
        const unsigned magic = ((1u << 20) + (unsigned)ww - 1u) / (unsigned)ww;   // i / ww == (i * magic) >> 20 while i * ww < 2^20
        for (int i = lane; i < n; i += 64) {
            const float v = w[i];
            const unsigned r = ((unsigned)i * magic) >> 20;
            const unsigned c = (unsigned)i - r * (unsigned)ww;
            if (v != 0.0f)
                atomicAdd(reinterpret_cast<float*>(reinterpret_cast<char*>(o) + (o0 + r * (unsigned)Ws + c) * 4u), v);
        }
    } else if (!empty) {   // the tile's taps spread over more cells than a window holds (a depth discontinuity)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (px[k].on) {
                const Taps& tp = px[k].t;
                atomicAdd(o + (long)tp.y0 * Ws + tp.x0, (tp.wx0 * tp.wy0) * px[k].v);
                atomicAdd(o + (long)tp.y1 * Ws + tp.x0, (tp.wx0 * tp.wy1) * px[k].v);
                atomicAdd(o + (long)tp.y0 * Ws + tp.x1, (tp.wx1 * tp.wy0) * px[k].v);
                atomicAdd(o + (long)tp.y1 * Ws + tp.x1, (tp.wx1 * tp.wy1) * px[k].v);
            }
    }
}

// The tap index of a grid on its own (the sampler's forward kernel inside warp_p2c writes the same
// index while it samples): same pixel-to-lane mapping as the streaming kernels.
__global__ __launch_bounds__(NT) void bilinear_tap_index_kernel(const float* __restrict__ grid,
                                                               const uint8_t* __restrict__ mask_u8, int Hs, int Ws,
                                                               long HWt, void* __restrict__ tap_index, int B) {
    const int b = blockIdx.y;
    const long p0 = lane_first_pixel();
    const float2* gb = reinterpret_cast<const float2*>(grid) + (long)b * HWt;
    const uint8_t* mb = mask_u8 ? mask_u8 + (long)b * HWt : nullptr;
    float2 g[PPL];
    uint8_t mk[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const long p = min(p0 + 64 * i, HWt - 1);
        g[i] = gb[p];
        mk[i] = mb ? mb[p] : (uint8_t)1;
    }
    int bx[PPL], by[PPL];
    bool on[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const Taps t = make_taps(g[i].x, g[i].y, Hs, Ws);
        bx[i] = t.x0;
        by[i] = t.y0;
        on[i] = (p0 + 64 * i < HWt) && mk[i] != 0 && t.x0 != t.x1 && t.y0 != t.y1;
    }
    emit_tap_index(tap_index_at(tap_index, B, gridDim.x, b), bx, by, on);
}

#ifndef FC3D_GDBG
#define FC3D_GDBG 0          // timing experiments only: 1 no window adds, 2 no sampling at all, 4 no run / segment scans, 8 no election, 16 one election round
#endif
#ifndef FC3D_GATHER_GY
#define FC3D_GATHER_GY 0     // 0: the tile height is chosen per launch (gather_tile_rows); otherwise fixed (experiments)
#endif
constexpr int GX = 64;                        // source tile of one wavefront: GX columns x GY rows (GY a template parameter)
constexpr int GWS = GX + 8;                   // window row stride: tile column 0 at window column 4 (16-byte aligned rows)
// window rows: tile row 0 at window row 1
constexpr int gather_window_rows(int gy) { return gy + 2; }
// Tile height of a launch.  Twelve rows are the optimum when the grid fills the chip several times over (fewer
// rows: more index scanning per sample; more: more candidates per tile).  A small problem -- one 256 x 256
// image is 22 workgroups of twelve-row tiles -- is bound by the length of one wavefront's chain of scans and
// sample steps instead, which shorter tiles cut: the tallest of 12 / 6 / 4 rows that still gives 1,024
// workgroups, else 4 (measured at 1 ... 64 images of 256^2 and 1 ... 8 of 512^2: 19.7 -> 13.7 us for one
// 256^2 image, break-even at about two million pixels).
static int gather_tile_rows(int B, int Hs, int Ws) {
    if (FC3D_GATHER_GY) return FC3D_GATHER_GY;
    const long cols = (Ws + (NT / 64) * GX - 1) / ((NT / 64) * GX);
    if ((long)B * cols * ((Hs + 11) / 12) >= 1024) return 12;
    if ((long)B * cols * ((Hs + 5) / 6) >= 1024) return 6;
    return 4;
}
constexpr int G_GROUPS = NT * 2;              // group boxes the workgroup scans per round
constexpr int G_HITCAP = 2 * G_GROUPS / (NT / 64);   // group hits a scanning wavefront can hold: two rounds' worth
constexpr int G_UNROLL = 2;                   // run-level steps (64 boxes each) whose loads are in flight together
constexpr int G_RUNS = 64 + 64 * G_UNROLL;    // run list: handed on above 64 entries, one round adds at most 64 * G_UNROLL
constexpr int G_CAND = 128;                   // candidate segments: drained above 64, one step adds at most 64
constexpr int G_SEGU = 4;                     // segment-level steps whose loads are in flight together
#ifndef FC3D_GATHER_DEPTH
#define FC3D_GATHER_DEPTH 4
#endif
constexpr int G_DEPTH = FC3D_GATHER_DEPTH;    // sample steps whose loads are in flight together
constexpr int G_OWN = 256;                    // election slots (cells hash onto them)

struct GatherSample {
    float2 g;
    float v;
    int live;          // non-zero: the pixel lies inside the target image and is not masked out (an int: flags
                       // of several samples in flight get packed into bytes, at a dozen instructions per step)
};

// appends the lanes with `hit` to list[n..] in lane order; returns the new length
template <typename L, typename V>
__device__ __forceinline__ int list_append(L list, int n, bool hit, V value, int lane) {
    const unsigned long long bal = ballot64(hit);
    if (hit) list[n + __popcll(bal & ((1ull << lane) - 1ull))] = value;
    return n + __popcll(bal);
}

constexpr int gather_window_cells(int gy) { return gather_window_rows(gy) * GWS; }
// window_add4 with the election slots shared by all cells that agree in the low bits of their
// index: two lanes with different cells may now meet in a slot, in which case one of them simply
// waits a turn.
__device__ __forceinline__ void window_add4_hashed(lds_vfloat win, lds_vu8 own, int c, int ww, float v00, float v10,
                                                   float v01, float v11, bool active, int lane) {
#if FC3D_GDBG & 8
    if (active) {               // timing only: no election (wrong where two lanes share a base cell)
        win[c] = win[c] + v00;
        win[c + 1] = win[c + 1] + v10;
        win[c + ww] = win[c + ww] + v01;
        win[c + ww + 1] = win[c + ww + 1] + v11;
    }
    return;
#endif
    bool pend = active;
    const int slot = c & (G_OWN - 1);
    do {
        if (pend) {
            own[slot] = (uint8_t)lane;
            if (own[slot] == (uint8_t)lane) {
                win[c] = win[c] + v00;
                win[c + 1] = win[c + 1] + v10;
                win[c + ww] = win[c + ww] + v01;
                win[c + ww + 1] = win[c + ww + 1] + v11;
                pend = false;
            }
        }
    } while (!(FC3D_GDBG & 16) && __builtin_expect(ballot64(pend) != 0, 0));   // 16, timing only: one round, losers dropped
}

// The first n candidate segments (ids in cand[], live bits in cbits[]) are sampled and their taps added
// to the window, 64 / SEG segments per step; the loads of G_DEPTH steps are issued together so that
// one memory latency covers them all.  A real function (not inlined): the scans call it from
// several places.
// A live sample's corners were not clamped: x0 = floor x, x1 = x0 + 1, and the weights are those
// of make_taps (x1 - x, x - x0; the corners are exact in fp32).
struct GatherTile {
    float cx, cy;                     // centre of the base cells whose taps reach the tile (GX + 1 columns, GY + 1 rows of them)
};
static_assert(GX % 2 == 0, "the base cells of a tile need a centre cell");
// its window cell: tile row 0 is window row 1, tile column 0 window column 4
constexpr int gather_centre_cell(int gy) { return (gy / 2) * GWS + GX / 2 + 3; }
#ifndef FC3D_GATHER_PIPE
#define FC3D_GATHER_PIPE 0        // 1: two batches of sample loads in flight, the second issued before the first batch is added (measured slower: 46.6-53 against 45.5 us, registers)
#endif
template <bool PAIRS, int GY>
__device__ __noinline__ void gather_drain(const float2* __restrict__ gb, const float* __restrict__ vb, unsigned HWt,
                                          lds_vfloat win, lds_vu8 own, lds_vint cand, lds_vint cbits, int n,
                                          GatherTile tl) {
    constexpr int STEP = 64 / SEG;
    const int lane = threadIdx.x & 63;
    const auto gbu = uniform_global(gb);
    const auto vbu = uniform_global(vb);
    static_assert(GY % 2 == 0, "the base cells of a tile need a centre cell");
    lds_vfloat winc = win + gather_centre_cell(GY);
    // the loads of one batch: G_DEPTH steps of STEP segments each
    auto issue = [&](GatherSample (&sm)[G_DEPTH], int e0) {
        if constexpr (PAIRS) {
            // Two steps share eight segments: a lane loads two neighbouring pixels of one of them (16 + 8 bytes)
            // and the first step takes the even pixel, the second the odd one.  Neighbouring pixels, which
            // share a base cell wherever the warp magnifies, then never meet in one election.
            static_assert(G_DEPTH % 2 == 0, "steps come in pairs");
#pragma unroll
            for (int pr = 0; pr < G_DEPTH / 2; ++pr) {
                // no branch around the loads (a pair of steps past the end re-reads the last segment and is not
                // consumed): values that come out of a branch get copied, and the copies wait for the loads
                const int e = e0 + pr * (2 * STEP);
                const int sl = lane >> 3, px = 2 * (lane & 7);
                const int j = min(e + sl, n - 1);
                const int seg = cand[j], cb = cbits[j];
                const int bits = (e + sl < n) ? cb : 0;
                const unsigned pair = min(((unsigned)seg * SEG + px) >> 1, (HWt >> 1) - 1u);
                const float4 g4 = ld_global_pair<kNtGatherLoad>(gbu, pair);
                const float2 v2 = ld_global_pair<kNtGatherLoad>(vbu, pair);
                sm[2 * pr].g = make_float2(g4.x, g4.y);
                sm[2 * pr].v = v2.x;
                sm[2 * pr].live = bits & (1 << px);
                sm[2 * pr + 1].g = make_float2(g4.z, g4.w);
                sm[2 * pr + 1].v = v2.y;
                sm[2 * pr + 1].live = bits & (2 << px);
            }
        } else {
#pragma unroll
            for (int d = 0; d < G_DEPTH; ++d) {
                const int e = e0 + d * STEP;
                if (e < n) {
                    const int j = min(e + (lane >> 4), n - 1);
                    const int seg = cand[j], bits = cbits[j];
                    const unsigned p = (unsigned)seg * SEG + (lane & 15);
                    sm[d].live = (e + (lane >> 4) < n) ? bits & (1 << (lane & 15)) : 0;
                    const unsigned q = min(p, HWt - 1u);
                    sm[d].g = ld_global<kNtGatherLoad>(gbu, q);
                    sm[d].v = ld_global<kNtGatherLoad>(vbu, q);
                }
            }
        }
    };
    // ... and their taps into the window
    auto consume = [&](GatherSample (&sm)[G_DEPTH], int e0) {
#pragma unroll
        for (int d = 0; d < G_DEPTH; ++d) {
            if (e0 + (PAIRS ? (d >> 1) * (2 * STEP) : d * STEP) < n) {
                // Relative to the centre of the tile's base cells, a sample is the tile's if it lies within half
                // a tile either way, and its window cell (counted from the centre cell) is one multiply-add.
                // x1 - x as 1 - (x - x0): x - x0 is exact for x >= 0, so both are the one rounding of x0 + 1 - x.
                const float gx = sm[d].g.x, gy = sm[d].g.y, v = sm[d].v;
                const float fx = floorf(gx), fy = floorf(gy);
                const float dx = __fsub_rn(fx, tl.cx), dy = __fsub_rn(fy, tl.cy);
                const bool on = sm[d].live != 0 && fabsf(dx) <= (float)(GX / 2) && fabsf(dy) <= (float)(GY / 2);
                const int c = on ? (int)__fadd_rn(__fmul_rn(dy, (float)GWS), dx) : 0;
                const float wx1 = __fsub_rn(gx, fx), wy1 = __fsub_rn(gy, fy);
                const fc3d_f2v wx = {__fsub_rn(1.0f, wx1), wx1};
                const float wy0 = __fsub_rn(1.0f, wy1);
                const fc3d_f2v r0 = (wx * wy0) * v, r1 = (wx * wy1) * v;      // {v00, v10}, {v01, v11}
#if FC3D_GDBG & 1
                if (on && v == 123.456f) winc[c] = r0.x + r1.y;
#else
                window_add4_hashed(winc, own, c, GWS, r0.x, r0.y, r1.x, r1.y, on, lane);
#endif
            }
        }
    };
    constexpr int BATCH = STEP * G_DEPTH;
#if FC3D_GATHER_PIPE
    GatherSample a[G_DEPTH], b[G_DEPTH];
    issue(a, 0);
    for (int e0 = 0; e0 < n; e0 += 2 * BATCH) {
        issue(b, e0 + BATCH);
        consume(a, e0);
        issue(a, e0 + 2 * BATCH);
        consume(b, e0 + BATCH);
    }
#else
    for (int e0 = 0; e0 < n; e0 += BATCH) {
        GatherSample sm[G_DEPTH];
        issue(sm, e0);
        consume(sm, e0);
    }
#endif
}

#ifdef FC3D_GATHER_WAVES
__attribute__((amdgpu_waves_per_eu(FC3D_GATHER_WAVES, FC3D_GATHER_WAVES)))
#endif
template <bool PAIRS, int GY>
__global__ __launch_bounds__(NT) void bilinear_bwd_img_gather_kernel(
    const float* __restrict__ g_out, const float* __restrict__ grid, void* __restrict__ tap_index, int B, int groups,
    int Hs, int Ws, unsigned HWt, float* __restrict__ g_img, const float* __restrict__ img,
    float* __restrict__ dot_partial) {
    constexpr int GWR = gather_window_rows(GY), G_WIN = gather_window_cells(GY);
    __shared__ __align__(16) float s_win[NT / 64][G_WIN];
    __shared__ uint8_t s_own[NT / 64][G_OWN];
    __shared__ uint16_t s_ghit[NT / 64][G_HITCAP];          // groups that meet the workgroup's row of tiles, per scanning wavefront, counted from `cbase`
    __shared__ int s_ng[NT / 64];
    __shared__ int s_rhit[NT / 64][G_RUNS];
    __shared__ int s_cand[NT / 64][2 * G_CAND];             // segment ids, then their live bits
    const int b = blockIdx.z;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rx0 = blockIdx.x * (NT / 64) * GX;            // the workgroup's NT/64 tiles lie side by side
    const int sx0 = rx0 + wv * GX, sy0 = blockIdx.y * GY;
    const bool tile_live = sx0 < Ws;
    // base cells whose taps reach the workgroup's tiles / this wavefront's tile
    const int rbx0 = rx0 - 1, rbx1 = rx0 + (NT / 64) * GX - 1, by0 = sy0 - 1, by1 = sy0 + GY - 1;
    const int bx0 = sx0 - 1, bx1 = sx0 + GX - 1;
    const TapIndex ix = tap_index_at(tap_index, B, groups, b);
    const float2* gb = reinterpret_cast<const float2*>(grid) + (long)b * HWt;
    const float* vb = g_out + (long)b * HWt;
    lds_vfloat win = lds_ptr(s_win[wv]);
    lds_vu8 own = lds_ptr(s_own[wv]);
    lds_vint rhit = lds_ptr(s_rhit[wv]);
    lds_vint cand = lds_ptr(s_cand[wv]), cbits = cand + G_CAND;
    {
        float4* w4 = reinterpret_cast<float4*>(s_win[wv]);
        for (int i = lane; i < GWR * GWS / 4; i += 64) w4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    GatherTile tl;
    tl.cx = (float)(bx0 + GX / 2); tl.cy = (float)(by0 + GY / 2);
    // hands the first m candidates to gather_drain and moves the rest (fewer than 64) to the front
    auto drain_front = [&](int& ncand, int m) {
#if !(FC3D_GDBG & 2)
        if (m > 0) gather_drain<PAIRS, GY>(gb, vb, HWt, win, own, cand, cbits, m, tl);
#endif
        const int rest = ncand - m;
        int keep_s = 0, keep_b = 0;
        if (lane < rest) { keep_s = cand[m + lane]; keep_b = cbits[m + lane]; }
        if (lane < rest) { cand[lane] = keep_s; cbits[lane] = keep_b; }
        ncand = rest;
    };

    // Three nested scans, each filling a list that the next level consumes: groups (scanned by the
    // whole workgroup against its row of tiles) -> runs -> segments -> samples (per wavefront, against
    // its own tile).  A list is handed on when it holds more than 64 entries or when its producer is
    // done; every loop runs one extra pass that only flushes, so that each level's code exists once.
    constexpr int GPS = 64 / RUN_PER_GROUP, RPS = 64 / SEG_PER_RUN;   // group hits / run hits whose children fill one step
    constexpr int GPW = G_GROUPS / NT;                                // group steps per wavefront and round
    // Group hits pile up over scan rounds and are worked off (runs -> segments -> samples) when a wavefront's list
    // might not hold another round, or at the end: an image of many groups (2,048 at 1024 x 1024) costs every
    // workgroup one chain of dependent loads per level, not one per 512 groups.
    int nr = 0, ncand = 0, cnt = 0, cbase = 0;
    for (int c0 = 0; c0 < groups; c0 += G_GROUPS) {
        const bool fin_g = c0 + G_GROUPS >= groups;
        int2 gbox[GPW];
#pragma unroll
        for (int u = 0; u < GPW; ++u) gbox[u] = ix.group[min(c0 + (wv * GPW + u) * 64 + lane, groups - 1)];
#pragma unroll
        for (int u = 0; u < GPW; ++u) {
            const int g = c0 + (wv * GPW + u) * 64 + lane;
            cnt = list_append(lds_ptr(s_ghit[wv]), cnt, g < groups && box_meets(gbox[u], rbx0, rbx1, by0, by1),
                              (uint16_t)(g - cbase), lane);
        }
        if (lane == 0) s_ng[wv] = cnt;
        __syncthreads();
        int off[NT / 64 + 1];
        off[0] = 0;
        int most = 0;
#pragma unroll
        for (int w2 = 0; w2 < NT / 64; ++w2) {
            const int k = s_ng[w2];
            off[w2 + 1] = off[w2] + k;
            most = max(most, k);
        }
        // the same in every wavefront of the workgroup: all read the same counts
        const bool flush = fin_g || most > G_HITCAP - GPW * 64 || c0 + 2 * G_GROUPS - cbase > 65536;
        if (flush && tile_live && !(FC3D_GDBG & 4)) {
            const int ng = off[NT / 64];
            const int rounds_r = (ng + GPS * G_UNROLL - 1) / (GPS * G_UNROLL);
            for (int rr = 0; rr <= rounds_r; ++rr) {
                const bool fin_r = rr == rounds_r;
                if (!fin_r) {
                    int run[G_UNROLL];
                    int2 rb[G_UNROLL];
#pragma unroll
                    for (int u = 0; u < G_UNROLL; ++u) {
                        const int j = min((rr * G_UNROLL + u) * GPS + lane / RUN_PER_GROUP, ng - 1);
                        int w2 = 0;
#pragma unroll
                        for (int q = 1; q < NT / 64; ++q) w2 += (j >= off[q]) ? 1 : 0;
                        int base = 0;
#pragma unroll
                        for (int q = 1; q < NT / 64; ++q) base = (w2 == q) ? off[q] : base;
                        run[u] = (cbase + (int)lds_ptr(s_ghit[0])[w2 * G_HITCAP + j - base]) * RUN_PER_GROUP + lane % RUN_PER_GROUP;
                        rb[u] = ix.run[run[u]];
                    }
#pragma unroll
                    for (int u = 0; u < G_UNROLL; ++u) {
                        const int j = (rr * G_UNROLL + u) * GPS + lane / RUN_PER_GROUP;
                        nr = list_append(rhit, nr, j < ng && box_meets(rb[u], bx0, bx1, by0, by1), run[u], lane);
                    }
                }
                if (nr <= 64 && !fin_r) continue;
                const int rounds_s = (nr + RPS * G_SEGU - 1) / (RPS * G_SEGU);
                for (int rs = 0; rs < rounds_s; ++rs) {
                    int seg[G_SEGU];
                    int2 sb[G_SEGU];
#pragma unroll
                    for (int u = 0; u < G_SEGU; ++u) {
                        const int j = (rs * G_SEGU + u) * RPS + lane / SEG_PER_RUN;
                        seg[u] = rhit[min(j, nr - 1)] * SEG_PER_RUN + lane % SEG_PER_RUN;
                        sb[u] = ix.seg[seg[u]];
                    }
#pragma unroll
                    for (int u = 0; u < G_SEGU; ++u) {
                        const int j = (rs * G_SEGU + u) * RPS + lane / SEG_PER_RUN;
                        const bool hit = j < nr && seg_meets(sb[u], bx0, bx1, by0, by1);
                        list_append(cbits, ncand, hit, (int)((unsigned)sb[u].y >> 16), lane);
                        ncand = list_append(cand, ncand, hit, seg[u], lane);
                        if (ncand > 64) drain_front(ncand, 64);
                    }
                }
                if (fin_g && fin_r) drain_front(ncand, ncand);
                nr = 0;
            }
        }
        if (flush) {
            cnt = 0;
            cbase = c0 + G_GROUPS;
        }
        __syncthreads();                              // s_ng (and after a flush s_ghit) is rewritten by the next round
    }
    // dot_partial (with img, the image that was sampled): one value per tile, the sum over the tile of
    // (gradient just formed) * img -- what a normalisation of img by one of its own statistics needs in
    // its backward pass; tile t of image b at dot_partial[b * tiles + t], tiles = all wavefronts of a grid z-slice
    const long dslot = ((long)b * gridDim.y + blockIdx.y) * (gridDim.x * (NT / 64)) + blockIdx.x * (NT / 64) + wv;
    if (!tile_live) {
        if (dot_partial && lane == 0) dot_partial[dslot] = 0.0f;
        return;
    }

    // the tile goes out with plain stores: four rows of sixteen 16-byte groups per step
    float* o = g_img + (long)b * Hs * Ws;
    const float* im = img + (long)b * Hs * Ws;
    const bool vec = (Ws % 4 == 0) && (reinterpret_cast<uintptr_t>(g_img) % 16 == 0) &&
                     (!dot_partial || reinterpret_cast<uintptr_t>(img) % 16 == 0);
    float dot = 0.0f;
    float4 iv[(GY + 3) / 4];
    if (dot_partial && vec) {      // the image tile's loads go first: they are under way while the window is read back
#pragma unroll
        for (int r0 = 0; r0 < GY; r0 += 4) {
            const int r = min(r0 + (lane >> 4), GY - 1), x = min(sx0 + 4 * (lane & 15), Ws - 4), y = min(sy0 + r, Hs - 1);
            iv[r0 / 4] = ld_at_pol<kNtGatherImg>(reinterpret_cast<const float4*>(im), ((unsigned)y * (unsigned)Ws + (unsigned)x) >> 2);
        }
    }
#pragma unroll
    for (int r0 = 0; r0 < GY; r0 += 4) {
        const int r = r0 + (lane >> 4), x = sx0 + 4 * (lane & 15), y = sy0 + r;
        if (r < GY && y < Hs && x < Ws) {
            const float4 v = *reinterpret_cast<const float4*>(&s_win[wv][(r + 1) * GWS + 4 + 4 * (lane & 15)]);
            if (vec) {
                st_at_pol<kNtGatherStore>(reinterpret_cast<float4*>(o), ((unsigned)y * (unsigned)Ws + (unsigned)x) >> 2, v);
                if (dot_partial) {
                    const float4 w = iv[r0 / 4];
                    dot = fmaf(v.x, w.x, dot); dot = fmaf(v.y, w.y, dot);
                    dot = fmaf(v.z, w.z, dot); dot = fmaf(v.w, w.w, dot);
                }
            } else {
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (x + k < Ws) {
                        o[(long)y * Ws + x + k] = vv[k];
                        if (dot_partial) dot = fmaf(vv[k], im[(long)y * Ws + x + k], dot);
                    }
            }
        }
    }
    if (dot_partial) {
        dot = wave_sum(dot);
        if (lane == 0) dot_partial[dslot] = dot;
    }
}

// Any channel count: plain float atomics into a zeroed gradient image (the xyz sampling of
// the capture simulator never needs this gradient; kept for completeness of the operator).
__global__ __launch_bounds__(NT) void bilinear_bwd_img_atomic_kernel(const float* __restrict__ g_out,
                                                                     const float* __restrict__ grid,
                                                                     const uint8_t* __restrict__ mask_u8, int Hs,
                                                                     int Ws, int C, long npix_t,
                                                                     float* __restrict__ g_img) {
    const int b = blockIdx.y;
    const long p = (long)blockIdx.x * NT + threadIdx.x;
    if (p >= npix_t) return;
    const long q = (long)b * npix_t + p;
    if (mask_u8 && mask_u8[q] == 0) return;
    const float2 g = *reinterpret_cast<const float2*>(grid + q * 2);
    const Taps t = make_taps(g.x, g.y, Hs, Ws);
    float* o = g_img + (long)b * Hs * Ws * C;
    const long a00 = ((long)t.y0 * Ws + t.x0) * C, a01 = ((long)t.y1 * Ws + t.x0) * C;
    const long a10 = ((long)t.y0 * Ws + t.x1) * C, a11 = ((long)t.y1 * Ws + t.x1) * C;
    for (int c = 0; c < C; ++c) {
        const float v = g_out[q * C + c];
        atomicAdd(o + a00 + c, (t.wx0 * t.wy0) * v);
        atomicAdd(o + a01 + c, (t.wx0 * t.wy1) * v);
        atomicAdd(o + a10 + c, (t.wx1 * t.wy0) * v);
        atomicAdd(o + a11 + c, (t.wx1 * t.wy1) * v);
    }
}

// ---------------------------------------------------------------------------------------
// standalone visibility mask
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void visible_mask_kernel(const float* __restrict__ grid, int H, int W, int method,
                                                          float* __restrict__ mask_f32,
                                                          uint8_t* __restrict__ mask_u8) {
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    const long p = (long)blockIdx.x * NT + threadIdx.x;
    if (p >= HW) return;
    const float2 g = *reinterpret_cast<const float2*>(grid + ((long)b * HW + p) * 2);
    if (mask_f32) splat_px<float>(mask_f32 + (long)b * HW, g.x, g.y, W, HW, method, 1.0f);
    if (mask_u8) splat_px<uint8_t>(mask_u8 + (long)b * HW, g.x, g.y, W, HW, method, (uint8_t)1);
}

// ---------------------------------------------------------------------------------------
// back-projection and the projector-depth -> camera-depth conversion
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void points_fwd_kernel(const float* __restrict__ z, const float* __restrict__ ray_x,
                                                        const float* __restrict__ ray_y,
                                                        const float* __restrict__ ray_dense, long ray_sb, int H, int W,
                                                        float* __restrict__ out) {
    const int b = blockIdx.z, y = blockIdx.y;
    const int x = blockIdx.x * NT + threadIdx.x;
    if (x >= W) return;
    const long p = (long)y * W + x, q = (long)b * H * W + p;
    float rx, ry, rz;
    if (ray_dense) {
        const float* r = ray_dense + (long)b * ray_sb + p * 3;
        rx = r[0]; ry = r[1]; rz = r[2];
    } else {
        rx = ray_x[x]; ry = ray_y[y]; rz = 1.0f;
    }
    const float zz = z[q];
    *reinterpret_cast<float4*>(out + q * 4) = make_float4(zz * rx, zz * ry, zz * rz, 1.0f);
}

__global__ __launch_bounds__(NT) void points_bwd_kernel(const float* __restrict__ g_out,
                                                        const float* __restrict__ ray_x,
                                                        const float* __restrict__ ray_y,
                                                        const float* __restrict__ ray_dense, long ray_sb, int H, int W,
                                                        float* __restrict__ g_z) {
    const int b = blockIdx.z, y = blockIdx.y;
    const int x = blockIdx.x * NT + threadIdx.x;
    if (x >= W) return;
    const long p = (long)y * W + x, q = (long)b * H * W + p;
    float rx, ry, rz;
    if (ray_dense) {
        const float* r = ray_dense + (long)b * ray_sb + p * 3;
        rx = r[0]; ry = r[1]; rz = r[2];
    } else {
        rx = ray_x[x]; ry = ray_y[y]; rz = 1.0f;
    }
    const float4 g = *reinterpret_cast<const float4*>(g_out + q * 4);
    g_z[q] = (g.x * rx + g.y * ry) + g.z * rz;
}

__device__ __forceinline__ float zc_den(const float* a, float x, float y) {
    return __fadd_rn(__fadd_rn(__fmul_rn(a[8], x), __fmul_rn(a[9], y)), a[10]);
}

__global__ __launch_bounds__(NT) void zp_to_zc_fwd_kernel(const float* __restrict__ zp, const float* __restrict__ xy,
                                                          const float* __restrict__ inv, long HW,
                                                          float* __restrict__ zc) {
    const int b = blockIdx.y;
    const long p = (long)blockIdx.x * NT + threadIdx.x;
    if (p >= HW) return;
    const long q = (long)b * HW + p;
    const float* a = inv + (long)b * 16;
    const float2 r = *reinterpret_cast<const float2*>(xy + q * 2);
    zc[q] = __fsub_rn(zp[q], a[11]) / zc_den(a, r.x, r.y);
}

__global__ __launch_bounds__(NT) void zp_to_zc_bwd_kernel(const float* __restrict__ g_zc,
                                                          const float* __restrict__ xy,
                                                          const float* __restrict__ inv, long HW,
                                                          float* __restrict__ g_zp) {
    const int b = blockIdx.y;
    const long p = (long)blockIdx.x * NT + threadIdx.x;
    if (p >= HW) return;
    const long q = (long)b * HW + p;
    const float* a = inv + (long)b * 16;
    const float2 r = *reinterpret_cast<const float2*>(xy + q * 2);
    g_zp[q] = g_zc[q] / zc_den(a, r.x, r.y);
}

// d zc / d (zp, xy, inverse pose): with q = g / den and t = -q * zc,
//   g_zp = q,   g_x = t * a20,   g_y = t * a21,
//   g_A[2][0..3] = sum over the sample's pixels of (t x, t y, t, -q); every other entry of g_A is zero.
// The four sums leave the kernel as one partial per workgroup; zc_grad_sum_kernel adds them in a fixed order.
__global__ __launch_bounds__(NT) void zp_to_zc_bwd_full_kernel(const float* __restrict__ g_zc, const float* __restrict__ zc,
                                                               const float* __restrict__ xy, const float* __restrict__ inv,
                                                               long HW, float* __restrict__ g_zp, float* __restrict__ g_xy,
                                                               float4* __restrict__ partial) {
    __shared__ float4 sh[NT / 64];
    const int b = blockIdx.y;
    const long p = (long)blockIdx.x * NT + threadIdx.x;
    const float* a = inv + (long)b * 16;
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (p < HW) {
        const long q = (long)b * HW + p;
        const float2 r = *reinterpret_cast<const float2*>(xy + q * 2);
        const float qq = g_zc[q] / zc_den(a, r.x, r.y);
        const float t = -(qq * zc[q]);
        if (g_zp) g_zp[q] = qq;
        if (g_xy) *reinterpret_cast<float2*>(g_xy + q * 2) = make_float2(t * a[8], t * a[9]);
        s = make_float4(t * r.x, t * r.y, t, -qq);
    }
    if (partial) {
        s.x = wave_sum(s.x); s.y = wave_sum(s.y); s.z = wave_sum(s.z); s.w = wave_sum(s.w);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            float4 t4 = sh[0];
#pragma unroll
            for (int w = 1; w < NT / 64; ++w) { t4.x += sh[w].x; t4.y += sh[w].y; t4.z += sh[w].z; t4.w += sh[w].w; }
            partial[(long)b * gridDim.x + blockIdx.x] = t4;
        }
    }
}

// g_A [B][16]: row 2 = the sums of the partials (fixed order, double accumulators), the rest zero.
__global__ __launch_bounds__(NT) void zc_grad_sum_kernel(const float4* __restrict__ partial, int nblk, float* __restrict__ g_inv) {
    __shared__ double s_red[NT / 64][4];
    const int b = blockIdx.x;
    const float4* pb = partial + (long)b * nblk;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < nblk; i += NT) {
        const float4 v = pb[i];
        s[0] += (double)v.x; s[1] += (double)v.y; s[2] += (double)v.z; s[3] += (double)v.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o, 64);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][k] = s[k];
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        float v = 0.0f;
        if (threadIdx.x >= 8 && threadIdx.x < 12) {
            double t = s_red[0][threadIdx.x - 8];
            for (int w = 1; w < NT / 64; ++w) t += s_red[w][threadIdx.x - 8];
            v = (float)t;
        }
        g_inv[(long)b * 16 + threadIdx.x] = v;
    }
}

// d points / d rays for per-sample ray maps (points_in_camera_coords with allBatch): g_ray = g_out[..., 0:3] * z
__global__ __launch_bounds__(NT) void points_bwd_rays_kernel(const float* __restrict__ g_out, const float* __restrict__ z,
                                                             long n, float* __restrict__ g_ray) {
    const long q = (long)blockIdx.x * NT + threadIdx.x;
    if (q >= n) return;
    const float4 g = *reinterpret_cast<const float4*>(g_out + q * 4);
    const float zz = z[q];
    g_ray[q * 3 + 0] = g.x * zz;
    g_ray[q * 3 + 1] = g.y * zz;
    g_ray[q * 3 + 2] = g.z * zz;
}

// ---------------------------------------------------------------------------------------
// Fused re-projection (the differentiated chain of reProj_IpGT, train_pytorch.py:194-202):
//   zc    = (zp - A[2][3]) / (A[2][0] x + A[2][1] y + A[2][2])           zp_cView_to_zc, A = inverse(pose)
//   coord = (zc x, zc y, zc, 1)                                          points_in_camera_coords(allBatch)
//   grid  = project(T, coord), T = K A                                   coord0TOgrid1
//   Ic    = bilinear(IpGT, grid)                                         bilinear_sampler
// in one pass per direction, with the arithmetic of the separate operators (same device
// functions, same rounding sequence), so the forward results are identical to composing them.
// The composed chain moves two [B,H,W,4] point clouds (and their gradients) through HBM; this
// one reads 12 B and writes 8 B per pixel forward, reads 20 B and writes 4 B backward, plus the
// four-tap gathers.
// ---------------------------------------------------------------------------------------
template <bool FMA>
__global__ __launch_bounds__(NT) void reproject_fwd_kernel(const float* __restrict__ zp, const float* __restrict__ xy,
                                                           const float* __restrict__ inv, const float* __restrict__ Tm,
                                                           const float* __restrict__ img, int H, int W, int Hs,
                                                           int Ws, float* __restrict__ zc, float* __restrict__ out,
                                                           float* __restrict__ grid) {
    const int b = blockIdx.y;
    const unsigned HW = (unsigned)(H * W);
    const unsigned p = blockIdx.x * NT + threadIdx.x;
    if (p >= HW) return;
    const long base = (long)b * HW;
    const float* a = inv + (long)b * 16;
    const float* T = Tm + (long)b * 16;
    const float2 r = ld_at(reinterpret_cast<const float2*>(xy) + base, p);
    const float z = __fsub_rn(ld_at(zp + base, p), a[11]) / zc_den(a, r.x, r.y);
    float gx, gy;
    project_px<FMA>(T, __fmul_rn(z, r.x), __fmul_rn(z, r.y), __fmul_rn(z, 1.0f), 1.0f, gx, gy);
    const Taps t = make_taps(gx, gy, Hs, Ws);
    const float* im = img + (long)b * Hs * Ws;
    const float i00 = ld_at(im, (unsigned)(t.y0 * Ws + t.x0)), i01 = ld_at(im, (unsigned)(t.y1 * Ws + t.x0));
    const float i10 = ld_at(im, (unsigned)(t.y0 * Ws + t.x1)), i11 = ld_at(im, (unsigned)(t.y1 * Ws + t.x1));
    st_at(zc + base, p, z);
    st_at(out + base, p, blend(t, i00, i01, i10, i11));
    if (grid) st_at(reinterpret_cast<float2*>(grid) + base, p, make_float2(gx, gy));
}

// g_zp = [ g_zc + (d Ic / d grid)(d grid / d coord)(d coord / d zc) g_Ic ] / zden; either incoming
// gradient may be absent.
template <bool FMA>
__global__ __launch_bounds__(NT) void reproject_bwd_kernel(const float* __restrict__ g_zc, const float* __restrict__ g_out,
                                                           const float* __restrict__ zc, const float* __restrict__ xy,
                                                           const float* __restrict__ inv, const float* __restrict__ Tm,
                                                           const float* __restrict__ img, int H, int W, int Hs,
                                                           int Ws, float* __restrict__ g_zp) {
    const int b = blockIdx.y;
    const unsigned HW = (unsigned)(H * W);
    const unsigned p = blockIdx.x * NT + threadIdx.x;
    if (p >= HW) return;
    const long base = (long)b * HW;
    const float* a = inv + (long)b * 16;
    const float* T = Tm + (long)b * 16;
    const float2 r = ld_at(reinterpret_cast<const float2*>(xy) + base, p);
    float gz = g_zc ? ld_at(g_zc + base, p) : 0.0f;
    if (g_out) {
        const float z = ld_at(zc + base, p);
        const float c0 = __fmul_rn(z, r.x), c1 = __fmul_rn(z, r.y), c2 = __fmul_rn(z, 1.0f);
        const float xu = row_dot<FMA>(T + 0, c0, c1, c2, 1.0f);
        const float yu = row_dot<FMA>(T + 4, c0, c1, c2, 1.0f);
        const float zu = row_dot<FMA>(T + 8, c0, c1, c2, 1.0f);
        const float den = __fadd_rn(zu, 1e-10f);
        const Taps t = make_taps(xu / den, yu / den, Hs, Ws);
        const float* im = img + (long)b * Hs * Ws;
        const float i00 = ld_at(im, (unsigned)(t.y0 * Ws + t.x0)), i01 = ld_at(im, (unsigned)(t.y1 * Ws + t.x0));
        const float i10 = ld_at(im, (unsigned)(t.y0 * Ws + t.x1)), i11 = ld_at(im, (unsigned)(t.y1 * Ws + t.x1));
        const float go = ld_at(g_out + base, p);
        const float ggx = go * (t.wy0 * (i10 - i00) + t.wy1 * (i11 - i01));      // as bilinear_bwd_grid
        const float ggy = go * (t.wx0 * (i01 - i00) + t.wx1 * (i11 - i10));
        const float gxu = ggx / den, gyu = ggy / den;                            // as project_bwd
        const float gzu = -(ggx * xu + ggy * yu) / (den * den);
        const float gc0 = T[0] * gxu + T[4] * gyu + T[8] * gzu;
        const float gc1 = T[1] * gxu + T[5] * gyu + T[9] * gzu;
        const float gc2 = T[2] * gxu + T[6] * gyu + T[10] * gzu;
        gz += (gc0 * r.x + gc1 * r.y) + gc2;                                     // as points_bwd
    }
    st_at(g_zp + base, p, gz / zc_den(a, r.x, r.y));
}

// ---------------------------------------------------------------------------------------
// The re-projection loss of reProj_IpGT (train_pytorch.py:188-207) on top of the chain above: the
// predicted depth is masked, re-projected and compared with the capture and the true camera depth,
// and only four sums leave the kernel,
//   S0 = sum (m (zcGT - zc))^2   S1 = sum (m (Ic - Ic'))^2   S2 = sum m |Ic - Ic'|   S3 = sum m
// (m = the eroded mask, Ic' = the re-projected pattern): neither zc nor Ic' is ever stored.  One
// partial per workgroup, added in a fixed order by a second launch, so the sums are reproducible.
// The backward pass recomputes the chain (it needs the taps anyway) and turns the gradients of the
// three sums straight into the gradient of the predicted depth.
// ---------------------------------------------------------------------------------------
struct ReprojPx {
    float me, zc, zden, xu, yu, den, icr, ic, zgt;
    Taps t;
    float i00, i01, i10, i11;
};

template <bool FMA>
__device__ __forceinline__ ReprojPx reproj_px(const float* __restrict__ zhat, const float* __restrict__ mask,
                                              const float* __restrict__ xy, const float* a, const float* T,
                                              const float* __restrict__ im, const float* __restrict__ Ic,
                                              const float* __restrict__ zcGT, long base, unsigned p, int Hs, int Ws) {
    ReprojPx q;
    const float2 r = ld_at(reinterpret_cast<const float2*>(xy) + base, p);
    q.me = ld_at(mask + base, p);
    q.ic = ld_at(Ic + base, p);
    q.zgt = ld_at(zcGT + base, p);
    const float zp = __fmul_rn(ld_at(zhat + base, p), q.me);
    q.zden = zc_den(a, r.x, r.y);
    q.zc = __fsub_rn(zp, a[11]) / q.zden;
    const float c0 = __fmul_rn(q.zc, r.x), c1 = __fmul_rn(q.zc, r.y), c2 = __fmul_rn(q.zc, 1.0f);
    q.xu = row_dot<FMA>(T + 0, c0, c1, c2, 1.0f);
    q.yu = row_dot<FMA>(T + 4, c0, c1, c2, 1.0f);
    q.den = __fadd_rn(row_dot<FMA>(T + 8, c0, c1, c2, 1.0f), 1e-10f);
    q.t = make_taps(q.xu / q.den, q.yu / q.den, Hs, Ws);
    q.i00 = ld_at(im, (unsigned)(q.t.y0 * Ws + q.t.x0)); q.i01 = ld_at(im, (unsigned)(q.t.y1 * Ws + q.t.x0));
    q.i10 = ld_at(im, (unsigned)(q.t.y0 * Ws + q.t.x1)); q.i11 = ld_at(im, (unsigned)(q.t.y1 * Ws + q.t.x1));
    q.icr = blend(q.t, q.i00, q.i01, q.i10, q.i11);
    return q;
}

template <bool FMA>
__global__ __launch_bounds__(NT) void reproj_loss_fwd_kernel(const float* __restrict__ zhat, const float* __restrict__ mask,
                                                             const float* __restrict__ xy, const float* __restrict__ inv,
                                                             const float* __restrict__ Tm, const float* __restrict__ img,
                                                             const float* __restrict__ Ic, const float* __restrict__ zcGT,
                                                             int H, int W, int Hs, int Ws, float4* __restrict__ partial) {
    __shared__ float4 sh[NT / 64];
    const int b = blockIdx.y;
    const unsigned HW = (unsigned)(H * W);
    const unsigned p = blockIdx.x * NT + threadIdx.x;
    float4 s = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (p < HW) {
        const ReprojPx q = reproj_px<FMA>(zhat, mask, xy, inv + (long)b * 16, Tm + (long)b * 16,
                                          img + (long)b * Hs * Ws, Ic, zcGT, (long)b * HW, p, Hs, Ws);
        const float dz = __fmul_rn(q.me, __fsub_rn(q.zgt, q.zc));
        const float di = __fsub_rn(q.ic, q.icr);
        const float mi = __fmul_rn(q.me, di);
        s = make_float4(__fmul_rn(dz, dz), __fmul_rn(mi, mi), __fmul_rn(q.me, fabsf(di)), q.me);
    }
    s.x = wave_sum(s.x); s.y = wave_sum(s.y); s.z = wave_sum(s.z); s.w = wave_sum(s.w);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float4 t = sh[0];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) { t.x += sh[w].x; t.y += sh[w].y; t.z += sh[w].z; t.w += sh[w].w; }
        partial[(long)b * gridDim.x + blockIdx.x] = t;
    }
}

// sums[0..3] = the partials added in index order (double accumulators, one workgroup), and the
// losses made of them with the reference's fp32 expressions (utils_pytorch.py:161-163, :177;
// train_pytorch.py:209): losses = [L1 + zc_rms, zc_rms, Ic_rms, L1],
//   rms = sqrt(S / max(n, 1e-6) + 1e-6),  L1 = S2 / (n + 1).
__global__ __launch_bounds__(NT) void reproj_loss_sum_kernel(const float4* __restrict__ partial, long n,
                                                             float* __restrict__ sums, float* __restrict__ losses) {
    __shared__ double sh[NT][4];
    double a[4] = {0.0, 0.0, 0.0, 0.0};
    for (long i = threadIdx.x; i < n; i += NT) {
        const float4 v = partial[i];
        a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) sh[threadIdx.x][j] = a[j];
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
#pragma unroll
            for (int j = 0; j < 4; ++j) sh[threadIdx.x][j] += sh[threadIdx.x + o][j];
        }
        __syncthreads();
    }
    if (threadIdx.x < 4) sums[threadIdx.x] = (float)sh[0][threadIdx.x];
    if (threadIdx.x == 0) {
        const float S0 = (float)sh[0][0], S1 = (float)sh[0][1], S2 = (float)sh[0][2], cnt = (float)sh[0][3];
        const float nc = fmaxf(cnt, 1e-6f);
        const float zc_rms = sqrtf(__fadd_rn(S0 / nc, 1e-6f));
        const float ic_rms = sqrtf(__fadd_rn(S1 / nc, 1e-6f));
        const float l1 = S2 / __fadd_rn(cnt, 1.0f);
        losses[0] = __fadd_rn(l1, zc_rms);
        losses[1] = zc_rms;
        losses[2] = ic_rms;
        losses[3] = l1;
    }
}

template <bool FMA>
__global__ __launch_bounds__(NT) void reproj_loss_bwd_kernel(const float* __restrict__ g_losses, const float* __restrict__ sums,
                                                             const float* __restrict__ zhat,
                                                             const float* __restrict__ mask, const float* __restrict__ xy,
                                                             const float* __restrict__ inv, const float* __restrict__ Tm,
                                                             const float* __restrict__ img, const float* __restrict__ Ic,
                                                             const float* __restrict__ zcGT, int H, int W, int Hs, int Ws,
                                                             float* __restrict__ g_zhat) {
    const int b = blockIdx.y;
    const unsigned HW = (unsigned)(H * W);
    const unsigned p = blockIdx.x * NT + threadIdx.x;
    if (p >= HW) return;
    const long base = (long)b * HW;
    const float* T = Tm + (long)b * 16;
    // gradients of the three sums from those of the four losses:
    //   d rms / d S = 1 / (2 rms max(n, 1e-6)),  d L1 / d S2 = 1 / (n + 1)
    const float nc = fmaxf(sums[3], 1e-6f);
    const float zc_rms = sqrtf(sums[0] / nc + 1e-6f), ic_rms = sqrtf(sums[1] / nc + 1e-6f);
    const float g0 = (g_losses[0] + g_losses[1]) / (2.0f * zc_rms * nc);
    const float g1 = g_losses[2] / (2.0f * ic_rms * nc);
    const float g2 = (g_losses[0] + g_losses[3]) / (sums[3] + 1.0f);
    const ReprojPx q = reproj_px<FMA>(zhat, mask, xy, inv + (long)b * 16, T, img + (long)b * Hs * Ws, Ic, zcGT, base, p,
                                      Hs, Ws);
    // d S0 / d zc = -2 m^2 (zcGT - zc);  d S1 / d Ic' = -2 m^2 (Ic - Ic');  d S2 / d Ic' = -m sign(Ic - Ic')
    const float m2 = q.me * q.me;
    const float di = q.ic - q.icr;
    const float sgn = di > 0.0f ? 1.0f : (di < 0.0f ? -1.0f : 0.0f);
    float gz = -2.0f * g0 * m2 * (q.zgt - q.zc);
    const float go = -2.0f * g1 * m2 * di - g2 * q.me * sgn;
    const float2 r = ld_at(reinterpret_cast<const float2*>(xy) + base, p);
    const float ggx = go * (q.t.wy0 * (q.i10 - q.i00) + q.t.wy1 * (q.i11 - q.i01));   // as reproject_bwd
    const float ggy = go * (q.t.wx0 * (q.i01 - q.i00) + q.t.wx1 * (q.i11 - q.i10));
    const float gxu = ggx / q.den, gyu = ggy / q.den;
    const float gzu = -(ggx * q.xu + ggy * q.yu) / (q.den * q.den);
    const float gc0 = T[0] * gxu + T[4] * gyu + T[8] * gzu;
    const float gc1 = T[1] * gxu + T[5] * gyu + T[9] * gzu;
    const float gc2 = T[2] * gxu + T[6] * gyu + T[10] * gzu;
    gz += (gc0 * r.x + gc1 * r.y) + gc2;
    st_at(g_zhat + base, p, (gz / q.zden) * q.me);
}

// Backward of coord_transform w.r.t. coord: with den = zu + 1e-10,
//   g_xu = g_x/den, g_yu = g_y/den, g_zu = -(g_x*xu + g_y*yu)/den^2, g_coord = T^T g_u.
template <int LAYOUT, bool FMA>
__global__ __launch_bounds__(NT) void project_bwd_kernel(const float* __restrict__ g_grid,
                                                         const float* __restrict__ coord, long sb, long sc, long sh,
                                                         long sw, const float* __restrict__ Tm, int H, int W,
                                                         float* __restrict__ g_coord) {
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    const long p0 = lane_first_pixel();
    const float* T = Tm + (long)b * 16;
    const float* cb = coord + (long)b * sb;
    const float2* gg = reinterpret_cast<const float2*>(g_grid) + (long)b * HW;
    float* o = g_coord + (long)b * 4 * HW;
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const long p = p0 + 64 * i;
        if (p >= HW) continue;
        float c[4];
        load_coord<LAYOUT>(cb, sc, sh, sw, p, W, c);
        const float xu = row_dot<FMA>(T + 0, c[0], c[1], c[2], c[3]);
        const float yu = row_dot<FMA>(T + 4, c[0], c[1], c[2], c[3]);
        const float zu = row_dot<FMA>(T + 8, c[0], c[1], c[2], c[3]);
        const float den = zu + 1e-10f;
        const float2 g = gg[p];
        const float gxu = g.x / den, gyu = g.y / den;
        const float gzu = -(g.x * xu + g.y * yu) / (den * den);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k * HW + p] = T[0 + k] * gxu + T[4 + k] * gyu + T[8 + k] * gzu;
    }
}

// Backward of coord_transform w.r.t. the transform (torch.matmul(transform, coord) is differentiable
// in both arguments, geometry_pytorch.py:79): g_T[i][k] = sum over pixels of g_u[i] * coord[k] for
// the three rows that reach the output (row 3 gets zero).  One partial of the twelve sums per
// workgroup (wavefront shuffles, then the workgroup's wavefronts in order), added up in index order
// with double accumulators by project_bwd_transform_sum_kernel: reproducible, no atomics.
template <int LAYOUT, bool FMA>
__global__ __launch_bounds__(NT) void project_bwd_transform_kernel(const float* __restrict__ g_grid,
                                                                   const float* __restrict__ coord, long sb, long sc,
                                                                   long sh, long sw, const float* __restrict__ Tm, int H,
                                                                   int W, float* __restrict__ partial) {
    __shared__ float s_part[NT / 64][12];
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    const long p0 = lane_first_pixel();
    const float* T = Tm + (long)b * 16;
    const float* cb = coord + (long)b * sb;
    const float2* gg = reinterpret_cast<const float2*>(g_grid) + (long)b * HW;
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0.0f;
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const long p = p0 + 64 * i;
        if (p >= HW) continue;
        float c[4];
        load_coord<LAYOUT>(cb, sc, sh, sw, p, W, c);
        const float xu = row_dot<FMA>(T + 0, c[0], c[1], c[2], c[3]);
        const float yu = row_dot<FMA>(T + 4, c[0], c[1], c[2], c[3]);
        const float zu = row_dot<FMA>(T + 8, c[0], c[1], c[2], c[3]);
        const float den = zu + 1e-10f;
        const float2 g = gg[p];
        const float gu[3] = {g.x / den, g.y / den, -(g.x * xu + g.y * yu) / (den * den)};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) acc[r * 4 + k] = fmaf(gu[r], c[k], acc[r * 4 + k]);
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = wave_sum(acc[k]);
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int k = 0; k < 12; ++k) s_part[threadIdx.x >> 6][k] = acc[k];
    }
    __syncthreads();
    if (threadIdx.x < 12) {
        float t = s_part[0][threadIdx.x];
#pragma unroll
        for (int w = 1; w < NT / 64; ++w) t += s_part[w][threadIdx.x];
        partial[((long)b * gridDim.x + blockIdx.x) * 12 + threadIdx.x] = t;
    }
}

__global__ __launch_bounds__(NT) void project_bwd_transform_sum_kernel(const float* __restrict__ partial, int nblk,
                                                                       float* __restrict__ g_T) {
    __shared__ double s_red[NT / 64];
    const int b = blockIdx.x;
    const float* pb = partial + (long)b * nblk * 12;
    for (int k = 0; k < 12; ++k) {
        double s = 0.0;
        for (int i = threadIdx.x; i < nblk; i += NT) s += (double)pb[(long)i * 12 + k];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = s_red[0];
            for (int w = 1; w < NT / 64; ++w) t += s_red[w];
            g_T[(long)b * 16 + k] = (float)t;
        }
    }
    if (threadIdx.x < 4) g_T[(long)b * 16 + 12 + threadIdx.x] = 0.0f;
}

// ---------------------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------------------
static int pick_layout(const float* coord, int64_t sc, int64_t sh, int64_t sw, int W) {
    if (sw == 1 && sh == W) return kPlanarFlat;
    if (sc == 1 && sw == 4 && sh == 4L * W && aligned16(coord)) return kPixel4Flat;
    return kStrided;
}

#define FC3D_DISPATCH_LAYOUT(NAME, KERNEL, layout, fma, ...)                                                     \
    do {                                                                                                   \
        if (fma) {                                                                                         \
            if (layout == kPlanarFlat) FC3D_LAUNCH(NAME, (KERNEL<kPlanarFlat, true, ##__VA_ARGS__>), g, blk, 0, s, ARGS); \
            else if (layout == kPixel4Flat) FC3D_LAUNCH(NAME, (KERNEL<kPixel4Flat, true, ##__VA_ARGS__>), g, blk, 0, s, ARGS); \
            else FC3D_LAUNCH(NAME, (KERNEL<kStrided, true, ##__VA_ARGS__>), g, blk, 0, s, ARGS);           \
        } else {                                                                                           \
            if (layout == kPlanarFlat) FC3D_LAUNCH(NAME, (KERNEL<kPlanarFlat, false, ##__VA_ARGS__>), g, blk, 0, s, ARGS); \
            else if (layout == kPixel4Flat) FC3D_LAUNCH(NAME, (KERNEL<kPixel4Flat, false, ##__VA_ARGS__>), g, blk, 0, s, ARGS); \
            else FC3D_LAUNCH(NAME, (KERNEL<kStrided, false, ##__VA_ARGS__>), g, blk, 0, s, ARGS);          \
        }                                                                                                  \
    } while (0)

// the two forward kernels of warp_p2c also exist for points formed from the depth plane
#define FC3D_DISPATCH_LAYOUT_FWD(NAME, KERNEL, layout, fma, ...)                                                 \
    do {                                                                                                   \
        if (layout == kFromDepth) {                                                                        \
            if (fma) FC3D_LAUNCH(NAME, (KERNEL<kFromDepth, true, ##__VA_ARGS__>), g, blk, 0, s, ARGS);     \
            else FC3D_LAUNCH(NAME, (KERNEL<kFromDepth, false, ##__VA_ARGS__>), g, blk, 0, s, ARGS);        \
        } else {                                                                                           \
            FC3D_DISPATCH_LAYOUT(NAME, KERNEL, layout, fma, ##__VA_ARGS__);                                \
        }                                                                                                  \
    } while (0)

}  // namespace fc3d

using namespace fc3d;

extern "C" int fc3d_pose_transforms(void* stream, const float* pose, const float* K, int64_t k_sb, int B,
                                    float* t_fwd_out, float* t_inv_out, float* inv_out, void* clear,
                                    size_t clear_bytes) {
    if (!pose || B <= 0) return FC3D_EINVAL;
    if (!K && (t_fwd_out || t_inv_out)) return FC3D_EINVAL;
    if (clear_bytes && !clear) return FC3D_EINVAL;
    if (clear_bytes && !aligned16(clear)) return FC3D_EALIGN;
    const long n16 = (long)(clear_bytes / 16);
    const int ntail = (int)(clear_bytes % 16);
    long blocks = cdiv(B, NT);
    const long want = cdiv(n16, (long)NT * 4);          // about four 16-byte stores per lane
    if (want > blocks) blocks = want < 4096 ? want : 4096;
    FC3D_LAUNCH("pose_transforms", pose_transforms_kernel, dim3((unsigned)blocks), dim3(NT), 0, (hipStream_t)stream,
                pose, K, (long)k_sb, B, t_fwd_out, t_inv_out, inv_out, (uint4*)clear, n16,
                (uint8_t*)clear + n16 * 16, ntail);
    FC3D_LAUNCH_CHECK();
    return FC3D_OK;
}

extern "C" int fc3d_project_fwd(void* stream, const float* coord, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                                const float* T, int B, int H, int W, float* grid_out) {
    if (!coord || !T || !grid_out) return FC3D_EINVAL;
    if (int e = check_img(B, H, W)) return e;
    if (!aligned4(coord) || (reinterpret_cast<uintptr_t>(grid_out) & 7u)) return FC3D_EALIGN;
    const int layout = pick_layout(coord, sc, sh, sw, W);
    const bool fma = use_fma(H, W);
    dim3 g(cdiv((long)H * W, PX_PER_BLOCK), B), blk(NT);
    hipStream_t s = (hipStream_t)stream;
    uint8_t* nomask = nullptr;
    const PoseArgs nopose = {};
    const float* noray = nullptr;
#define ARGS coord, (long)sb, (long)sc, (long)sh, (long)sw, T, H, W, grid_out, nomask, nopose, noray, noray
    FC3D_DISPATCH_LAYOUT("project_fwd", project_kernel, layout, fma, false);
#undef ARGS
    FC3D_LAUNCH_CHECK();
    return FC3D_OK;
}

extern "C" int fc3d_project_bwd(void* stream, const float* g_grid, const float* coord, int64_t sb, int64_t sc,
                                int64_t sh, int64_t sw, const float* T, int B, int H, int W, float* g_coord) {
    if (!g_grid || !coord || !T || !g_coord) return FC3D_EINVAL;
    if (int e = check_img(B, H, W)) return e;
    if (!aligned4(coord) || !aligned4(g_coord) || (reinterpret_cast<uintptr_t>(g_grid) & 7u)) return FC3D_EALIGN;
    const int layout = pick_layout(coord, sc, sh, sw, W);
    const bool fma = use_fma(H, W);
    dim3 g(cdiv((long)H * W, PX_PER_BLOCK), B), blk(NT);
    hipStream_t s = (hipStream_t)stream;
#define ARGS g_grid, coord, (long)sb, (long)sc, (long)sh, (long)sw, T, H, W, g_coord
    FC3D_DISPATCH_LAYOUT("project_bwd", project_bwd_kernel, layout, fma);
#undef ARGS
    FC3D_LAUNCH_CHECK();
    return FC3D_OK;
}

extern "C" size_t fc3d_project_bwd_transform_workspace(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * cdiv((long)H * W, PX_PER_BLOCK) * 12 * sizeof(float);
}

extern "C" int fc3d_project_bwd_transform(void* stream, const float* g_grid, const float* coord, int64_t sb, int64_t sc,
                                          int64_t sh, int64_t sw, const float* T, int B, int H, int W, void* workspace,
                                          float* g_T) {
    if (!g_grid || !coord || !T || !workspace || !g_T) return FC3D_EINVAL;
    if (int e = check_img(B, H, W)) return e;
    if (!aligned4(coord) || !aligned4(workspace) || (reinterpret_cast<uintptr_t>(g_grid) & 7u)) return FC3D_EALIGN;
    const int layout = pick_layout(coord, sc, sh, sw, W);
    const bool fma = use_fma(H, W);
    dim3 g(cdiv((long)H * W, PX_PER_BLOCK), B), blk(NT);
    hipStream_t s = (hipStream_t)stream;
    float* partial = (float*)workspace;
#define ARGS g_grid, coord, (long)sb, (long)sc, (long)sh, (long)sw, T, H, W, partial
    FC3D_DISPATCH_LAYOUT("project_bwd_transform", project_bwd_transform_kernel, layout, fma);
#undef ARGS
    FC3D_LAUNCH("project_bwd_transform_sum", project_bwd_transform_sum_kernel, dim3(B), blk, 0, s, (const float*)partial,
                (int)g.x, g_T);
    return FC3D_OK;
}

extern "C" int fc3d_bilinear_fwd(void* stream, const float* img, const float* grid, const uint8_t* mask_u8, int B,
                                 int Hs, int Ws, int C, int Ht, int Wt, float* out) {
    if (!img || !grid || !out || C <= 0) return FC3D_EINVAL;
    if (int e = check_img(B, Hs, Ws)) return e;
    if (int e = check_img(B, Ht, Wt)) return e;
    if ((long)B * Hs * Ws * C >= (1L << 31) || (long)B * Ht * Wt * C >= (1L << 31)) return FC3D_ERANGE;
    if ((long)Ht * Wt >= (1L << 29) || (long)Hs * Ws >= (1L << 30)) return FC3D_ERANGE;   // per-image byte offsets are 32-bit
    if (!aligned4(img) || (reinterpret_cast<uintptr_t>(grid) & 7u)) return FC3D_EALIGN;
    const long np = (long)Ht * Wt;
    FC3D_LAUNCH("bilinear_fwd", bilinear_fwd_kernel, dim3(cdiv(np, NT), B), dim3(NT), 0, (hipStream_t)stream, img, grid, mask_u8,
                       Hs, Ws, C, np, out);
    FC3D_LAUNCH_CHECK();
    return FC3D_OK;
}

extern "C" int fc3d_bilinear_bwd_grid(void* stream, const float* g_out, const float* img, const float* grid,
                                      const uint8_t* mask_u8, int B, int Hs, int Ws, int C, int Ht, int Wt,
                                      float* g_grid) {
    if (!g_out || !img || !grid || !g_grid || C <= 0) return FC3D_EINVAL;
    if (int e = check_img(B, Hs, Ws)) return e;
    if (int e = check_img(B, Ht, Wt)) return e;
    if ((long)B * Hs * Ws * C >= (1L << 31) || (long)B * Ht * Wt * C >= (1L << 31)) return FC3D_ERANGE;
    if ((reinterpret_cast<uintptr_t>(grid) & 7u) || (reinterpret_cast<uintptr_t>(g_grid) & 7u)) return FC3D_EALIGN;
    const long np = (long)Ht * Wt;
    FC3D_LAUNCH("bilinear_bwd_grid", bilinear_bwd_grid_kernel, dim3(cdiv(np, NT), B), dim3(NT), 0, (hipStream_t)stream, g_out, img,
                       grid, mask_u8, Hs, Ws, C, np, g_grid);
    FC3D_LAUNCH_CHECK();
    return FC3D_OK;
}

static inline int wave_tiles(int Ht, int Wt, int* tiles_x) {
    *tiles_x = cdiv(Wt, WT_X);
    return *tiles_x * cdiv(Ht, WT_Y);
}

// the gather form needs 32-bit per-image byte offsets and 16-bit box fields
static inline bool gather_fits(int Hs, int Ws, int Ht, int Wt) {
    return Hs <= kTapIndexMaxDim && Ws <= kTapIndexMaxDim && (long)Ht * Wt < (1L << 29) && (long)Hs * Ws < (1L << 30);
}

extern "C" size_t fc3d_bilinear_bwd_img_workspace(int B, int Hs, int Ws, int C, int Ht, int Wt) {
    if (B <= 0 || Hs <= 0 || Ws <= 0 || Ht <= 0 || Wt <= 0 || C != 1 || !gather_fits(Hs, Ws, Ht, Wt)) return 0;
    return (size_t)B * tap_index_groups(Ht, Wt) * (1 + RUN_PER_GROUP * (1 + SEG_PER_RUN)) * sizeof(int2);
}

static int launch_tap_index(hipStream_t s, const float* grid, const uint8_t* mask_u8, int B, int Hs, int Ws, int Ht,
                            int Wt, void* index) {
    FC3D_LAUNCH("bilinear_tap_index", bilinear_tap_index_kernel, dim3((unsigned)tap_index_groups(Ht, Wt), B), dim3(NT), 0, s,
                grid, mask_u8, Hs, Ws, (long)Ht * Wt, index, B);
    return FC3D_OK;
}

extern "C" int fc3d_bilinear_tap_index(void* stream, const float* grid, const uint8_t* mask_u8, int B, int Hs, int Ws,
                                       int Ht, int Wt, void* index) {
    if (!grid || !index) return FC3D_EINVAL;
    if (int e = check_img(B, Hs, Ws)) return e;
    if (int e = check_img(B, Ht, Wt)) return e;
    if (!gather_fits(Hs, Ws, Ht, Wt)) return FC3D_ERANGE;
    if ((reinterpret_cast<uintptr_t>(grid) & 7u) || (reinterpret_cast<uintptr_t>(index) & 7u)) return FC3D_EALIGN;
    return launch_tap_index((hipStream_t)stream, grid, mask_u8, B, Hs, Ws, Ht, Wt, index);
}

#ifndef FC3D_GATHER_PAIRS
#define FC3D_GATHER_PAIRS 1
#endif
// The image-gradient gather: samples are loaded two neighbouring pixels at a time where the target images hold
// an even number of pixels and the planes are aligned for it (every practical size), one at a time otherwise.
static int launch_gather(hipStream_t s, const float* g_out, const float* grid, void* index, int B, int Hs, int Ws, int Ht,
                         int Wt, float* g_img, const float* img, float* dot_partial) {
    const long HWt = (long)Ht * Wt;
    const int gy = gather_tile_rows(B, Hs, Ws);
    const dim3 gridDim3(cdiv(Ws, (NT / 64) * GX), cdiv(Hs, gy), B);
    const bool pairs = FC3D_GATHER_PAIRS && HWt % 2 == 0 && aligned16(grid) && (reinterpret_cast<uintptr_t>(g_out) & 7u) == 0;
#define FC3D_GATHER_CASE(PAIRS_, GY_)                                                                                   \
    if (pairs == PAIRS_ && gy == GY_) {                                                                                 \
        FC3D_LAUNCH("bilinear_bwd_img_gather", (bilinear_bwd_img_gather_kernel<PAIRS_, GY_>), gridDim3, dim3(NT), 0, s, \
                    g_out, grid, index, B, (int)tap_index_groups(Ht, Wt), Hs, Ws, (unsigned)HWt, g_img, img,           \
                    dot_partial);                                                                                       \
        return FC3D_OK;                                                                                                 \
    }
#if FC3D_GATHER_GY
    FC3D_GATHER_CASE(true, FC3D_GATHER_GY) FC3D_GATHER_CASE(false, FC3D_GATHER_GY)
#else
    FC3D_GATHER_CASE(true, 12) FC3D_GATHER_CASE(false, 12) FC3D_GATHER_CASE(true, 6) FC3D_GATHER_CASE(false, 6)
    FC3D_GATHER_CASE(true, 4) FC3D_GATHER_CASE(false, 4)
#endif
#undef FC3D_GATHER_CASE
    return FC3D_EINVAL;
}

extern "C" int fc3d_bilinear_bwd_img(void* stream, const float* g_out, const float* grid, const uint8_t* mask_u8,
                                     int B, int Hs, int Ws, int C, int Ht, int Wt, void* index, int index_ready,
                                     float* g_img) {
    if (!g_out || !grid || !g_img || C <= 0) return FC3D_EINVAL;
    if (int e = check_img(B, Hs, Ws)) return e;
    if (int e = check_img(B, Ht, Wt)) return e;
    if ((long)B * Hs * Ws * C >= (1L << 31) || (long)B * Ht * Wt * C >= (1L << 31)) return FC3D_ERANGE;
    if ((long)Ht * Wt >= (1L << 29) || (long)Hs * Ws >= (1L << 30)) return FC3D_ERANGE;   // per-image byte offsets are 32-bit
    if (reinterpret_cast<uintptr_t>(grid) & 7u) return FC3D_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    if (C != 1) {
        if (int e = zero_fill(s, g_img, (size_t)B * Hs * Ws * C * sizeof(float))) return e;
        const long np = (long)Ht * Wt;
        FC3D_LAUNCH("bilinear_bwd_img_atomic", bilinear_bwd_img_atomic_kernel, dim3(cdiv(np, NT), B), dim3(NT), 0, s, g_out, grid, mask_u8,
                           Hs, Ws, C, np, g_img);
        return FC3D_OK;
    }
    if (!index) {   // window sums go straight into the zeroed gradient image with float atomics
        int tiles_x;
        const int ntiles = wave_tiles(Ht, Wt, &tiles_x);
        if (int e = zero_fill(s, g_img, (size_t)B * Hs * Ws * sizeof(float))) return e;
        FC3D_LAUNCH("bilinear_bwd_img_scatter", bilinear_bwd_img_scatter_kernel, dim3(cdiv(ntiles, NT / 64), B), dim3(NT), 0, s,
                    g_out, grid, mask_u8, Hs, Ws, Ht, Wt, tiles_x, ntiles, g_img);
        return FC3D_OK;
    }
    if (!gather_fits(Hs, Ws, Ht, Wt)) return FC3D_ERANGE;
    if (reinterpret_cast<uintptr_t>(index) & 7u) return FC3D_EALIGN;
    if (!index_ready) {
        if (int e = launch_tap_index(s, grid, mask_u8, B, Hs, Ws, Ht, Wt, index)) return e;
    }
    return launch_gather(s, g_out, grid, index, B, Hs, Ws, Ht, Wt, g_img, nullptr, nullptr);
}

extern "C" int fc3d_bilinear_bwd_img_dot_count(int B, int Hs, int Ws) {
    if (B <= 0 || Hs <= 0 || Ws <= 0) return 0;
    return cdiv(Ws, (NT / 64) * GX) * (NT / 64) * cdiv(Hs, gather_tile_rows(B, Hs, Ws));
}

extern "C" int fc3d_bilinear_bwd_img_dot(void* stream, const float* g_out, const float* grid, int B, int Hs, int Ws,
                                         int Ht, int Wt, void* index, const float* img, float* g_img,
                                         float* dot_partial) {
    if (!g_out || !grid || !g_img || !index || !img || !dot_partial) return FC3D_EINVAL;
    if (int e = check_img(B, Hs, Ws)) return e;
    if (int e = check_img(B, Ht, Wt)) return e;
    if (!gather_fits(Hs, Ws, Ht, Wt)) return FC3D_ERANGE;
    if ((reinterpret_cast<uintptr_t>(grid) & 7u) || (reinterpret_cast<uintptr_t>(index) & 7u)) return FC3D_EALIGN;
    if (!aligned4(img) || !aligned4(g_img) || !aligned4(dot_partial)) return FC3D_EALIGN;
    return launch_gather((hipStream_t)stream, g_out, grid, index, B, Hs, Ws, Ht, Wt, g_img, img, dot_partial);
}

extern "C" int fc3d_visible_mask(void* stream, const float* grid, int B, int H, int W, int method, float* mask_f32,
                                 uint8_t* mask_u8) {
    if (!grid || (!mask_f32 && !mask_u8) || (method != 0 && method != 1)) return FC3D_EINVAL;
    if (int e = check_img(B, H, W)) return e;
    if (reinterpret_cast<uintptr_t>(grid) & 7u) return FC3D_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const size_t n = (size_t)B * H * W;
    if (mask_f32) {
        if (int e = zero_fill(s, mask_f32, n * sizeof(float))) return e;
    }
    if (mask_u8) {
        if (int e = zero_fill(s, mask_u8, n)) return e;
    }
    const long HW = (long)H * W;
    FC3D_LAUNCH("visible_mask", visible_mask_kernel, dim3(cdiv(HW, NT), B), dim3(NT), 0, s, grid, H, W, method, mask_f32,
                       mask_u8);
    FC3D_LAUNCH_CHECK();
    return FC3D_OK;
}

// warp_p2c forward in its two launches.
// 1. projection of coord_p + visibility splat.  pose == NULL: the transform T_p2c is an input.  Otherwise
//    T_p2c / T_c2p / inv are outputs: the kernel forms them from pose and K (no separate pose_transforms launch).
static int launch_project_splat(hipStream_t s, const float* coord_p, int64_t p_sb, int64_t p_sc, int64_t p_sh,
                                int64_t p_sw, float* T_p2c, float* T_c2p, const float* pose, const float* K,
                                int64_t k_sb, float* inv_out, int B, int H, int W, float* grid_p2c, uint8_t* mask_u8,
                                int mask_is_clear, const float* ray_x = nullptr, const float* ray_y = nullptr) {
    // ray_x / ray_y given: coord_p is the depth plane [B][H][W] the points are formed from (kFromDepth)
    if (!coord_p || !T_p2c || !grid_p2c || !mask_u8 || (pose && (!K || !T_c2p)) || (!ray_x != !ray_y)) return FC3D_EINVAL;
    if (int e = check_img(B, H, W)) return e;
    if ((long)H * W >= (1L << 28)) return FC3D_ERANGE;   // the same limit as the sampling launch behind this one
    if (!aligned4(coord_p) || (reinterpret_cast<uintptr_t>(grid_p2c) & 7u)) return FC3D_EALIGN;
    const bool fma = use_fma(H, W);
    dim3 g(cdiv((long)H * W, PX_PER_BLOCK), B), blk(NT);
    if (!mask_is_clear) {
        if (int e = zero_fill(s, mask_u8, (size_t)B * H * W)) return e;
    }
    const int layout = ray_x ? (int)kFromDepth : pick_layout(coord_p, p_sc, p_sh, p_sw, W);
    const PoseArgs pa = {pose, K, (long)k_sb, T_p2c, T_c2p, inv_out};
#define ARGS coord_p, (long)p_sb, (long)p_sc, (long)p_sh, (long)p_sw, (const float*)T_p2c, H, W, grid_p2c, mask_u8, pa, ray_x, ray_y
    if (pose) FC3D_DISPATCH_LAYOUT_FWD("warp_project_splat", project_kernel, layout, fma, true, true);
    else FC3D_DISPATCH_LAYOUT_FWD("warp_project_splat", project_kernel, layout, fma, true);
#undef ARGS
    FC3D_LAUNCH_CHECK();
    return FC3D_OK;
}

// 2. projection of coord_c, sampling of Ip, compositing with the mask (and the tap index on the side)
static int launch_warp_sample(hipStream_t s, const float* Ip, const float* coord_c, int64_t c_sb, int64_t c_sc,
                              int64_t c_sh, int64_t c_sw, const float* T_c2p, int B, int H, int W, uint8_t* mask_u8,
                              int mask_rezero, float* Ic_masked, float* grid_c2p, void* tap_index,
                              const float* ray_x = nullptr, const float* ray_y = nullptr) {
    if (!Ip || !coord_c || !T_c2p || !Ic_masked || !grid_c2p || !mask_u8 || (!ray_x != !ray_y)) return FC3D_EINVAL;
    if (int e = check_img(B, H, W)) return e;
    if ((long)H * W >= (1L << 28)) return FC3D_ERANGE;   // per-image byte offsets (16 B per pixel for the points) are 32-bit
    if (tap_index && (!gather_fits(H, W, H, W) || (reinterpret_cast<uintptr_t>(tap_index) & 7u))) return FC3D_ERANGE;
    if (!aligned4(coord_c) || !aligned4(Ip) || !aligned4(Ic_masked) || (reinterpret_cast<uintptr_t>(grid_c2p) & 7u))
        return FC3D_EALIGN;
    const bool fma = use_fma(H, W);
    dim3 g(cdiv((long)H * W, PX_PER_BLOCK), B), blk(NT);
    const int layout = ray_x ? (int)kFromDepth : pick_layout(coord_c, c_sc, c_sh, c_sw, W);
    // 2: runs of 64*PPL bytes start 16-byte aligned in every image, so they can be cleared with 16-byte stores
    const int rezero = !mask_rezero ? 0 : ((aligned16(mask_u8) && ((long)H * W) % 16 == 0) ? 2 : 1);
#define ARGS coord_c, (long)c_sb, (long)c_sc, (long)c_sh, (long)c_sw, T_c2p, Ip, mask_u8, rezero, H, W, grid_c2p, Ic_masked, tap_index, B, ray_x, ray_y
    FC3D_DISPATCH_LAYOUT_FWD("warp_sample", warp_sample_kernel, layout, fma);
#undef ARGS
    FC3D_LAUNCH_CHECK();
    return FC3D_OK;
}

extern "C" int fc3d_warp_p2c_fwd(void* stream, const float* Ip, const float* coord_p, int64_t p_sb, int64_t p_sc,
                                 int64_t p_sh, int64_t p_sw, const float* coord_c, int64_t c_sb, int64_t c_sc,
                                 int64_t c_sh, int64_t c_sw, const float* T_p2c, const float* T_c2p, int B, int H,
                                 int W, float* Ic_masked, float* grid_p2c, float* grid_c2p, uint8_t* mask_u8,
                                 int mask_is_clear, void* tap_index) {
    if (!Ip || !coord_c || !T_c2p || !Ic_masked || !grid_c2p) return FC3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (int e = launch_project_splat(s, coord_p, p_sb, p_sc, p_sh, p_sw, const_cast<float*>(T_p2c), nullptr, nullptr,
                                     nullptr, 0, nullptr, B, H, W, grid_p2c, mask_u8, mask_is_clear))
        return e;
    return launch_warp_sample(s, Ip, coord_c, c_sb, c_sc, c_sh, c_sw, T_c2p, B, H, W, mask_u8, 0, Ic_masked, grid_c2p,
                              tap_index);
}

extern "C" int fc3d_warp_project_splat(void* stream, const float* coord_p, int64_t p_sb, int64_t p_sc, int64_t p_sh,
                                       int64_t p_sw, const float* pose, const float* K, int64_t k_sb, int B, int H,
                                       int W, float* grid_p2c, uint8_t* mask_u8, int mask_is_clear, float* T_p2c_out,
                                       float* T_c2p_out, float* inv_out) {
    if (!pose || !K) return FC3D_EINVAL;
    return launch_project_splat((hipStream_t)stream, coord_p, p_sb, p_sc, p_sh, p_sw, T_p2c_out, T_c2p_out, pose, K,
                                k_sb, inv_out, B, H, W, grid_p2c, mask_u8, mask_is_clear);
}

extern "C" int fc3d_warp_sample(void* stream, const float* Ip, const float* coord_c, int64_t c_sb, int64_t c_sc,
                                int64_t c_sh, int64_t c_sw, const float* T_c2p, int B, int H, int W, uint8_t* mask_u8,
                                int mask_rezero, float* Ic_masked, float* grid_c2p, void* tap_index) {
    return launch_warp_sample((hipStream_t)stream, Ip, coord_c, c_sb, c_sc, c_sh, c_sw, T_c2p, B, H, W, mask_u8,
                              mask_rezero, Ic_masked, grid_c2p, tap_index);
}

extern "C" int fc3d_warp_p2c_pose_fwd(void* stream, const float* Ip, const float* coord_p, int64_t p_sb, int64_t p_sc,
                                      int64_t p_sh, int64_t p_sw, const float* coord_c, int64_t c_sb, int64_t c_sc,
                                      int64_t c_sh, int64_t c_sw, const float* pose, const float* K, int64_t k_sb,
                                      int B, int H, int W, float* Ic_masked, float* grid_p2c, float* grid_c2p,
                                      uint8_t* mask_u8, int mask_is_clear, int mask_rezero, void* tap_index,
                                      float* T_p2c_out, float* T_c2p_out, float* inv_out) {
    if (!pose || !K || !Ip || !coord_c || !Ic_masked || !grid_c2p) return FC3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (int e = launch_project_splat(s, coord_p, p_sb, p_sc, p_sh, p_sw, T_p2c_out, T_c2p_out, pose, K, k_sb, inv_out, B,
                                     H, W, grid_p2c, mask_u8, mask_is_clear))
        return e;
    return launch_warp_sample(s, Ip, coord_c, c_sb, c_sc, c_sh, c_sw, T_c2p_out, B, H, W, mask_u8, mask_rezero, Ic_masked,
                              grid_c2p, tap_index);
}

extern "C" int fc3d_warp_p2c_pose_depth_fwd(void* stream, const float* Ip, const float* z_p, const float* z_c,
                                            const float* ray_x, const float* ray_y, const float* pose, const float* K,
                                            int64_t k_sb, int B, int H, int W, float* Ic_masked, float* grid_p2c,
                                            float* grid_c2p, uint8_t* mask_u8, int mask_is_clear, int mask_rezero,
                                            void* tap_index, float* T_p2c_out, float* T_c2p_out, float* inv_out) {
    if (!pose || !K || !Ip || !z_p || !z_c || !ray_x || !ray_y || !Ic_masked || !grid_c2p) return FC3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    const int64_t HW = (int64_t)H * W;
    if (int e = launch_project_splat(s, z_p, HW, 0, 0, 0, T_p2c_out, T_c2p_out, pose, K, k_sb, inv_out, B, H, W, grid_p2c,
                                     mask_u8, mask_is_clear, ray_x, ray_y))
        return e;
    return launch_warp_sample(s, Ip, z_c, HW, 0, 0, 0, T_c2p_out, B, H, W, mask_u8, mask_rezero, Ic_masked, grid_c2p,
                              tap_index, ray_x, ray_y);
}

extern "C" int fc3d_points_fwd(void* stream, const float* z, const float* ray_x, const float* ray_y,
                               const float* ray_dense, int64_t ray_sb, int B, int H, int W, float* out) {
    if (!z || !out || (!ray_dense && (!ray_x || !ray_y))) return FC3D_EINVAL;
    if (int e = check_img(B, H, W)) return e;
    if (!aligned16(out)) return FC3D_EALIGN;
    FC3D_LAUNCH("points_fwd", points_fwd_kernel, dim3(cdiv(W, NT), H, B), dim3(NT), 0, (hipStream_t)stream, z, ray_x, ray_y,
                       ray_dense, (long)ray_sb, H, W, out);
    FC3D_LAUNCH_CHECK();
    return FC3D_OK;
}

extern "C" int fc3d_points_bwd(void* stream, const float* g_out, const float* ray_x, const float* ray_y,
                               const float* ray_dense, int64_t ray_sb, int B, int H, int W, float* g_z) {
    if (!g_out || !g_z || (!ray_dense && (!ray_x || !ray_y))) return FC3D_EINVAL;
    if (int e = check_img(B, H, W)) return e;
    if (!aligned16(g_out)) return FC3D_EALIGN;
    FC3D_LAUNCH("points_bwd", points_bwd_kernel, dim3(cdiv(W, NT), H, B), dim3(NT), 0, (hipStream_t)stream, g_out, ray_x,
                       ray_y, ray_dense, (long)ray_sb, H, W, g_z);
    FC3D_LAUNCH_CHECK();
    return FC3D_OK;
}

extern "C" int fc3d_zp_to_zc_fwd(void* stream, const float* zp, const float* xy, const float* inv, int B, int H, int W,
                                 float* zc) {
    if (!zp || !xy || !inv || !zc) return FC3D_EINVAL;
    if (int e = check_img(B, H, W)) return e;
    if (reinterpret_cast<uintptr_t>(xy) & 7u) return FC3D_EALIGN;
    const long HW = (long)H * W;
    FC3D_LAUNCH("zp_to_zc_fwd", zp_to_zc_fwd_kernel, dim3(cdiv(HW, NT), B), dim3(NT), 0, (hipStream_t)stream, zp, xy, inv, HW,
                       zc);
    FC3D_LAUNCH_CHECK();
    return FC3D_OK;
}

extern "C" int fc3d_zp_to_zc_bwd(void* stream, const float* g_zc, const float* xy, const float* inv, int B, int H,
                                 int W, float* g_zp) {
    if (!g_zc || !xy || !inv || !g_zp) return FC3D_EINVAL;
    if (int e = check_img(B, H, W)) return e;
    if (reinterpret_cast<uintptr_t>(xy) & 7u) return FC3D_EALIGN;
    const long HW = (long)H * W;
    FC3D_LAUNCH("zp_to_zc_bwd", zp_to_zc_bwd_kernel, dim3(cdiv(HW, NT), B), dim3(NT), 0, (hipStream_t)stream, g_zc, xy, inv, HW,
                       g_zp);
    FC3D_LAUNCH_CHECK();
    return FC3D_OK;
}

extern "C" size_t fc3d_zp_to_zc_bwd_full_workspace(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * cdiv((long)H * W, NT) * sizeof(float4);
}

extern "C" int fc3d_zp_to_zc_bwd_full(void* stream, const float* g_zc, const float* zc, const float* xy, const float* inv,
                                      int B, int H, int W, float* g_zp, float* g_xy, void* workspace, float* g_inv) {
    if (!g_zc || !zc || !xy || !inv || (!g_zp && !g_xy && !g_inv) || (g_inv && !workspace)) return FC3D_EINVAL;
    if (int e = check_img(B, H, W)) return e;
    if ((reinterpret_cast<uintptr_t>(xy) & 7u) || (g_xy && (reinterpret_cast<uintptr_t>(g_xy) & 7u)) ||
        (workspace && !aligned16(workspace)))
        return FC3D_EALIGN;
    const long HW = (long)H * W;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(cdiv(HW, NT), B);
    float4* partial = g_inv ? (float4*)workspace : nullptr;
    FC3D_LAUNCH("zp_to_zc_bwd_full", zp_to_zc_bwd_full_kernel, g, dim3(NT), 0, s, g_zc, zc, xy, inv, HW, g_zp, g_xy, partial);
    if (g_inv)
        FC3D_LAUNCH("zc_grad_sum", zc_grad_sum_kernel, dim3(B), dim3(NT), 0, s, (const float4*)partial, (int)g.x, g_inv);
    return FC3D_OK;
}

extern "C" int fc3d_points_bwd_rays(void* stream, const float* g_out, const float* z, int B, int H, int W, float* g_ray) {
    if (!g_out || !z || !g_ray) return FC3D_EINVAL;
    if (int e = check_img(B, H, W)) return e;
    if (!aligned16(g_out)) return FC3D_EALIGN;
    const long n = (long)B * H * W;
    FC3D_LAUNCH("points_bwd_rays", points_bwd_rays_kernel, dim3(cdiv(n, NT)), dim3(NT), 0, (hipStream_t)stream, g_out, z, n,
                g_ray);
    return FC3D_OK;
}

static int check_reproject(const void* xy, int B, int H, int W, int Hs, int Ws) {
    if (int e = check_img(B, H, W)) return e;
    if (int e = check_img(B, Hs, Ws)) return e;
    if ((long)H * W >= (1L << 29) || (long)Hs * Ws >= (1L << 30)) return FC3D_ERANGE;   // per-image byte offsets are 32-bit
    if (reinterpret_cast<uintptr_t>(xy) & 7u) return FC3D_EALIGN;
    return FC3D_OK;
}

extern "C" int fc3d_reproject_fwd(void* stream, const float* zp, const float* xy, const float* inv, const float* T,
                                  const float* img, int B, int H, int W, int Hs, int Ws, float* zc, float* out,
                                  float* grid) {
    if (!zp || !xy || !inv || !T || !img || !zc || !out) return FC3D_EINVAL;
    if (int e = check_reproject(xy, B, H, W, Hs, Ws)) return e;
    if (grid && (reinterpret_cast<uintptr_t>(grid) & 7u)) return FC3D_EALIGN;
    const dim3 g(cdiv((long)H * W, NT), B), blk(NT);
    hipStream_t s = (hipStream_t)stream;
    if (use_fma(H, W)) FC3D_LAUNCH("reproject_fwd", reproject_fwd_kernel<true>, g, blk, 0, s, zp, xy, inv, T, img, H, W, Hs, Ws, zc, out, grid);
    else FC3D_LAUNCH("reproject_fwd", reproject_fwd_kernel<false>, g, blk, 0, s, zp, xy, inv, T, img, H, W, Hs, Ws, zc, out, grid);
    return FC3D_OK;
}

extern "C" int fc3d_reproject_bwd(void* stream, const float* g_zc, const float* g_out, const float* zc, const float* xy,
                                  const float* inv, const float* T, const float* img, int B, int H, int W, int Hs,
                                  int Ws, float* g_zp) {
    if (!xy || !inv || !g_zp || (!g_zc && !g_out)) return FC3D_EINVAL;
    if (g_out && (!zc || !T || !img)) return FC3D_EINVAL;
    if (int e = check_reproject(xy, B, H, W, Hs, Ws)) return e;
    const dim3 g(cdiv((long)H * W, NT), B), blk(NT);
    hipStream_t s = (hipStream_t)stream;
    if (use_fma(H, W)) FC3D_LAUNCH("reproject_bwd", reproject_bwd_kernel<true>, g, blk, 0, s, g_zc, g_out, zc, xy, inv, T, img, H, W, Hs, Ws, g_zp);
    else FC3D_LAUNCH("reproject_bwd", reproject_bwd_kernel<false>, g, blk, 0, s, g_zc, g_out, zc, xy, inv, T, img, H, W, Hs, Ws, g_zp);
    return FC3D_OK;
}

extern "C" size_t fc3d_reproj_loss_workspace(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * cdiv((long)H * W, NT) * sizeof(float4);
}

extern "C" int fc3d_reproj_loss_fwd(void* stream, const float* zp_hat, const float* mask, const float* xy,
                                    const float* inv, const float* T, const float* img, const float* Ic,
                                    const float* zcGT, int B, int H, int W, int Hs, int Ws, void* workspace,
                                    float* sums, float* losses) {
    if (!zp_hat || !mask || !xy || !inv || !T || !img || !Ic || !zcGT || !workspace || !sums || !losses)
        return FC3D_EINVAL;
    if (int e = check_reproject(xy, B, H, W, Hs, Ws)) return e;
    if (!aligned16(workspace)) return FC3D_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(cdiv((long)H * W, NT), B), blk(NT);
    float4* partial = (float4*)workspace;
    if (use_fma(H, W)) FC3D_LAUNCH("reproj_loss_fwd", reproj_loss_fwd_kernel<true>, g, blk, 0, s, zp_hat, mask, xy, inv, T, img, Ic, zcGT, H, W, Hs, Ws, partial);
    else FC3D_LAUNCH("reproj_loss_fwd", reproj_loss_fwd_kernel<false>, g, blk, 0, s, zp_hat, mask, xy, inv, T, img, Ic, zcGT, H, W, Hs, Ws, partial);
    FC3D_LAUNCH("reproj_loss_sum", reproj_loss_sum_kernel, dim3(1), blk, 0, s, (const float4*)partial, (long)g.x * B, sums, losses);
    return FC3D_OK;
}

// ---------------------------------------------------------------------------------------
// fc3d_reproj_loss_fwd without a workspace argument: the per-workgroup partial sums live in a
// file-scope device array instead, so the call needs no device memory beyond its inputs and
// outputs.  Every GPU holds its own copy of the array at its own address; the address is looked
// up once per device (with that device current) and kept.  `scratch` below is only the host-side
// handle of the symbol: kernels and copies always get the looked-up device address.
// ---------------------------------------------------------------------------------------
constexpr long kScratchPartials = 65536;           // float4 partial sums: 1 MiB per GPU
static __device__ __align__(256) float4 scratch[kScratchPartials];
constexpr int kScratchMaxDevices = 64;
static std::atomic<float4*> s_scratch_ptr[kScratchMaxDevices];

static int scratch_for_current_device(float4** out) {
    int dev = -1;
    FC3D_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kScratchMaxDevices) return FC3D_ERANGE;
    float4* p = s_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {
        FC3D_HIP(hipGetSymbolAddress(reinterpret_cast<void**>(&p), HIP_SYMBOL(scratch)));
        if (!p) return FC3D_EINVAL;
        s_scratch_ptr[dev].store(p, std::memory_order_release);   // racing first calls look up the same address
    }
    *out = p;
    return FC3D_OK;
}

extern "C" int fc3d_reproj_loss_fwd_static(void* stream, const float* zp_hat, const float* mask, const float* xy,
                                           const float* inv, const float* T, const float* img, const float* Ic,
                                           const float* zcGT, int B, int H, int W, int Hs, int Ws, float* sums,
                                           float* losses) {
    if (!zp_hat || !mask || !xy || !inv || !T || !img || !Ic || !zcGT || !sums || !losses) return FC3D_EINVAL;
    if (int e = check_reproject(xy, B, H, W, Hs, Ws)) return e;
    const dim3 g(cdiv((long)H * W, NT), B), blk(NT);
    if ((long)g.x * B > kScratchPartials) return FC3D_ESCRATCH;
    float4* partial = nullptr;
    if (int e = scratch_for_current_device(&partial)) return e;
    hipStream_t s = (hipStream_t)stream;
    // every partial the sum reads is written by the first launch: nothing has to start at zero
    if (use_fma(H, W)) FC3D_LAUNCH("reproj_loss_fwd", reproj_loss_fwd_kernel<true>, g, blk, 0, s, zp_hat, mask, xy, inv, T, img, Ic, zcGT, H, W, Hs, Ws, partial);
    else FC3D_LAUNCH("reproj_loss_fwd", reproj_loss_fwd_kernel<false>, g, blk, 0, s, zp_hat, mask, xy, inv, T, img, Ic, zcGT, H, W, Hs, Ws, partial);
    FC3D_LAUNCH("reproj_loss_sum", reproj_loss_sum_kernel, dim3(1), blk, 0, s, (const float4*)partial, (long)g.x * B, sums, losses);
    return FC3D_OK;
}

extern "C" int fc3d_reproj_loss_bwd(void* stream, const float* g_losses, const float* sums, const float* zp_hat,
                                    const float* mask,
                                    const float* xy, const float* inv, const float* T, const float* img,
                                    const float* Ic, const float* zcGT, int B, int H, int W, int Hs, int Ws,
                                    float* g_zp_hat) {
    if (!g_losses || !sums || !zp_hat || !mask || !xy || !inv || !T || !img || !Ic || !zcGT || !g_zp_hat) return FC3D_EINVAL;
    if (int e = check_reproject(xy, B, H, W, Hs, Ws)) return e;
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(cdiv((long)H * W, NT), B), blk(NT);
    if (use_fma(H, W)) FC3D_LAUNCH("reproj_loss_bwd", reproj_loss_bwd_kernel<true>, g, blk, 0, s, g_losses, sums, zp_hat, mask, xy, inv, T, img, Ic, zcGT, H, W, Hs, Ws, g_zp_hat);
    else FC3D_LAUNCH("reproj_loss_bwd", reproj_loss_bwd_kernel<false>, g, blk, 0, s, g_losses, sums, zp_hat, mask, xy, inv, T, img, Ic, zcGT, H, W, Hs, Ws, g_zp_hat);
    return FC3D_OK;
}

extern "C" int fc3d_version(void) { return FC3D_VERSION; }

extern "C" const char* fc3d_status_string(int status) {
    switch (status) {
        case FC3D_OK: return "ok";
        case FC3D_EINVAL: return "invalid argument (null pointer or non-positive size)";
        case FC3D_ERANGE: return "size out of the range the kernels index";
        case FC3D_EALIGN: return "pointer not aligned to its element type";
        case FC3D_ESCRATCH: return "problem larger than the library's static scratch buffer (fc3d_reproj_loss_fwd_static: B*ceil(H*W/256) <= 65536)";
        default: break;
    }
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "unknown status";
}
