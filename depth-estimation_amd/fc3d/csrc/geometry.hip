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
__device__ inline bool invert4(const float* a, double* inv) {
    double m[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            m[i][j] = (double)a[i * 4 + j];
            m[i][4 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; ++c) {
        int p = c;
        double best = fabs(m[c][c]);
        for (int r = c + 1; r < 4; ++r)
            if (fabs(m[r][c]) > best) {
                best = fabs(m[r][c]);
                p = r;
            }
        if (p != c)
            for (int j = 0; j < 8; ++j) {
                double t = m[c][j];
                m[c][j] = m[p][j];
                m[p][j] = t;
            }
        const double d = 1.0 / m[c][c];
        for (int j = 0; j < 8; ++j) m[c][j] *= d;
        for (int r = 0; r < 4; ++r)
            if (r != c) {
                const double f = m[r][c];
                for (int j = 0; j < 8; ++j) m[r][j] -= f * m[c][j];
            }
    }
    for (int i = 0; i < 4; ++i)
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
enum CoordLayout { kStrided = 0, kPlanarFlat = 1, kPixel4Flat = 2 };
// kPlanarFlat : dense planes, component k of flat pixel p at base[k*sc + p]
// kPixel4Flat : dense [H][W][4] (what z2pointcloud's permuted view points at): one aligned float4 per pixel
// kStrided    : any other view; (y, x) are recovered from p

template <int LAYOUT>
__device__ __forceinline__ void load_coord(const float* __restrict__ base, long sc, long sh, long sw, long p, int W,
                                           float c[4]) {
    if (LAYOUT == kPixel4Flat) {
        const float4 v = *reinterpret_cast<const float4*>(base + p * 4);
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

// coord_transform; with SPLAT also the 'all' visibility splat of warp_p2c's forward warp.
template <int LAYOUT, bool FMA, bool SPLAT>
__global__ __launch_bounds__(NT) void project_kernel(const float* __restrict__ coord, long sb, long sc, long sh,
                                                     long sw, const float* __restrict__ Tm, int H, int W,
                                                     float* __restrict__ grid, uint8_t* __restrict__ mask_u8) {
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    const long p0 = lane_first_pixel();
    const float* T = Tm + (long)b * 16;
    const float* cb = coord + (long)b * sb;
    float c[PPL][4];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const long p = p0 + 64 * i;
        if (p < HW) load_coord<LAYOUT>(cb, sc, sh, sw, p, W, c[i]);
    }
    float2* go = reinterpret_cast<float2*>(grid) + (long)b * HW;
    uint8_t* m = SPLAT ? mask_u8 + (long)b * HW : nullptr;
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const long p = p0 + 64 * i;
        if (p < HW) {
            float gx, gy;
            project_px<FMA>(T, c[i][0], c[i][1], c[i][2], c[i][3], gx, gy);
            go[p] = make_float2(gx, gy);
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
                                                         const uint8_t* __restrict__ mask_u8, int H, int W,
                                                         float* __restrict__ grid, float* __restrict__ out) {
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    const long p0 = lane_first_pixel();
    const float* T = Tm + (long)b * 16;
    const float* cb = coord + (long)b * sb;
    const float* img = Ip + (long)b * HW;
    const uint8_t* mk = mask_u8 + (long)b * HW;
    float c[PPL][4];
    float mv[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const long p = p0 + 64 * i;
        if (p < HW) {
            load_coord<LAYOUT>(cb, sc, sh, sw, p, W, c[i]);
            mv[i] = (float)mk[p];
        }
    }
    float2* go = reinterpret_cast<float2*>(grid) + (long)b * HW;
    float* oo = out + (long)b * HW;
    float gx[PPL], gy[PPL], i00[PPL], i01[PPL], i10[PPL], i11[PPL];
    Taps t[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const long p = p0 + 64 * i;
        if (p < HW) {
            project_px<FMA>(T, c[i][0], c[i][1], c[i][2], c[i][3], gx[i], gy[i]);
            t[i] = make_taps(gx[i], gy[i], H, W);
            i00[i] = img[(long)t[i].y0 * W + t[i].x0];
            i01[i] = img[(long)t[i].y1 * W + t[i].x0];
            i10[i] = img[(long)t[i].y0 * W + t[i].x1];
            i11[i] = img[(long)t[i].y1 * W + t[i].x1];
        }
    }
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const long p = p0 + 64 * i;
        if (p < HW) {
            go[p] = make_float2(gx[i], gy[i]);
            oo[p] = blend(t[i], i00[i], i01[i], i10[i], i11[i]) * mv[i];
        }
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
// d out / d img for C == 1: scatter-add through per-wavefront LDS windows
//
// Scatter: each wavefront owns a 32x8 tile of target pixels (four neighbouring pixels of a row
//   per lane).  It finds the bounding box of the
//   source pixels its taps touch (DPP min/max) and sums all taps into a private LDS window of
//   that box.  LDS float atomics retire about one lane every three cycles on this chip, so the
//   window is updated with plain read-add-write; lanes of one instruction that hit the same cell
//   are serialised by an election through a byte-sized owner array.
// Then either (no workspace; the default of the Python wrappers)
//   the window is added to the zero-filled gradient image with global float atomics: about one
//   atomic per pixel instead of four, but last-bit differences from run to run,
// or (workspace given; torch's deterministic mode)
//   the window is stored compactly ("slab") with its box, and a merge pass lets each workgroup
//   own a 64x16 tile of the source image and add, in ascending tile order, every slab whose box
//   overlaps it -> fixed summation order, reproducible bits; tiles whose box did not fit a window
//   (discontinuous warps) are added with global float atomics in an overflow pass.
//
// A sample whose two x corners (or two y corners) were clamped onto the same source pixel
// contributes +w and -w to that pixel, i.e. exactly nothing; such samples are skipped.
// ---------------------------------------------------------------------------------------
// Wavefront tile: 8 lanes x 4 neighbouring pixels in each of 8 rows.  (64 x 4 loads a little faster
// still -- 56.8 against 58 us at 32x512x512 -- but its box under a 20 degree rotation no longer fits
// the window and the kernel falls from 39 to 85 us there.)
constexpr int WT_X = 32, WT_Y = 8;
constexpr int WT_LX = WT_X / 4;          // lanes along x
constexpr int WIN_CAP = 1024;            // cells per window / floats per slab

struct TileMeta { int xmin, ymin, ww, wh; };   // ww = 0: empty, ww = -1: overflow

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
        // 4-byte load of the mask (measured against one pixel per lane and load: 58 instead of 64 us
        // for the whole kernel at 32x512x512)
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
__device__ __forceinline__ void window_add4(volatile float* win, volatile uint8_t* own, int c, int ww, float v00,
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

template <bool DIRECT>
__global__ __launch_bounds__(NT) void bilinear_bwd_img_scatter_kernel(
    const float* __restrict__ g_out, const float* __restrict__ grid, const uint8_t* __restrict__ mask_u8, int Hs,
    int Ws, int Ht, int Wt, int tiles_x, int ntiles, TileMeta* __restrict__ meta, float* __restrict__ slab,
    float* __restrict__ g_img) {
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
    volatile float* w = s_win[wv];
    volatile uint8_t* own = s_own[wv];
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
    if (DIRECT) {
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
        } else if (!empty) {
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
        return;
    }
    const long tg = (long)b * ntiles + t;
    float* s = slab + tg * WIN_CAP;
    for (int i = lane; i < n; i += 64) s[i] = w[i];
    if (lane == 0) {
        TileMeta m;
        m.xmin = empty ? 0 : xmin; m.ymin = empty ? 0 : ymin;
        m.ww = empty ? 0 : (fits ? ww : -1); m.wh = wh;
        meta[tg] = m;
    }
}

constexpr int MT_X = 64, MT_Y = 16;      // merge tile of the source image
constexpr int LIST_CAP = 512;

__global__ __launch_bounds__(NT) void bilinear_bwd_img_merge_kernel(const TileMeta* __restrict__ meta,
                                                                    const float* __restrict__ slab, int ntiles,
                                                                    int Hs, int Ws, float* __restrict__ g_img) {
    __shared__ TileMeta s_list[LIST_CAP];
    __shared__ int s_tile[LIST_CAP];
    __shared__ int s_wcnt[NT / 64];
    const int b = blockIdx.z;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const int bx0 = blockIdx.x * MT_X, by0 = blockIdx.y * MT_Y;
    const int x = bx0 + lane;                       // a wavefront reads 64 neighbouring pixels of a row
    float acc[4] = {0.f, 0.f, 0.f, 0.f};            // rows by0 + wv + 4*i
    const TileMeta* mb = meta + (long)b * ntiles;
    const float* sb = slab + (long)b * ntiles * WIN_CAP;
    int count = 0;

    auto flush = [&](int total) {
        for (int j0 = 0; j0 < total; j0 += 4) {
            float v[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = j0 + u;
                TileMeta e = {0, 0, 0, 0};
                const float* base = sb;
                if (j < total) {
                    e = s_list[j];
                    base = sb + (long)s_tile[j] * WIN_CAP;
                }
                const int cx = x - e.xmin;
                const bool inx = (cx >= 0) && (cx < e.ww);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ry = by0 + wv + 4 * i - e.ymin;
                    v[u][i] = (inx && ry >= 0 && ry < e.wh) ? base[ry * e.ww + cx] : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] += v[u][i];   // ascending tile order, u after u
        }
    };

    for (int base = 0; base < ntiles; base += NT) {
        const int t = base + tid;
        TileMeta m = {0, 0, 0, 0};
        bool hit = false;
        if (t < ntiles) {
            m = mb[t];
            hit = (m.ww > 0) && (m.xmin < bx0 + MT_X) && (m.xmin + m.ww > bx0) && (m.ymin < by0 + MT_Y) &&
                  (m.ymin + m.wh > by0);
        }
        const unsigned long long bal = __ballot(hit);
        if (lane == 0) s_wcnt[wv] = __popcll(bal);
        __syncthreads();
        int off = count, total = count;
#pragma unroll
        for (int i = 0; i < NT / 64; ++i) {
            if (i < wv) off += s_wcnt[i];
            total += s_wcnt[i];
        }
        if (hit) {
            const int pos = off + __popcll(bal & ((1ull << lane) - 1ull));
            s_list[pos] = m;
            s_tile[pos] = t;
        }
        __syncthreads();
        count = total;
        if (count > LIST_CAP - NT || base + NT >= ntiles) {
            flush(count);
            count = 0;
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int y = by0 + wv + 4 * i;
        if (y < Hs && x < Ws) g_img[(long)b * Hs * Ws + (long)y * Ws + x] = acc[i];
    }
}

__global__ __launch_bounds__(NT) void bilinear_bwd_img_overflow_kernel(
    const float* __restrict__ g_out, const float* __restrict__ grid, const uint8_t* __restrict__ mask_u8, int Hs,
    int Ws, int Ht, int Wt, int tiles_x, int ntiles, const TileMeta* __restrict__ meta, float* __restrict__ g_img) {
    const int b = blockIdx.y;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int t = blockIdx.x * (NT / 64) + wv;
    if (t >= ntiles) return;
    if (meta[(long)b * ntiles + t].ww != -1) return;
    PxTaps px[4];
    load_tile_taps(g_out, grid, mask_u8, (long)b * Ht * Wt, Ht, Wt, Hs, Ws, (t % tiles_x) * WT_X,
                   (t / tiles_x) * WT_Y, lane, px);
    float* o = g_img + (long)b * Hs * Ws;
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
#define ARGS coord, (long)sb, (long)sc, (long)sh, (long)sw, T, H, W, grid_out, nomask
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

extern "C" size_t fc3d_bilinear_bwd_img_workspace(int B, int Hs, int Ws, int C, int Ht, int Wt) {
    if (B <= 0 || Hs <= 0 || Ws <= 0 || Ht <= 0 || Wt <= 0 || C != 1) return 0;
    int tx;
    const size_t nt = (size_t)B * wave_tiles(Ht, Wt, &tx);
    return nt * sizeof(TileMeta) + nt * WIN_CAP * sizeof(float);
}

extern "C" int fc3d_bilinear_bwd_img(void* stream, const float* g_out, const float* grid, const uint8_t* mask_u8,
                                     int B, int Hs, int Ws, int C, int Ht, int Wt, void* workspace, float* g_img) {
    if (!g_out || !grid || !g_img || C <= 0) return FC3D_EINVAL;
    if (int e = check_img(B, Hs, Ws)) return e;
    if (int e = check_img(B, Ht, Wt)) return e;
    if ((long)B * Hs * Ws * C >= (1L << 31) || (long)B * Ht * Wt * C >= (1L << 31)) return FC3D_ERANGE;
    if (reinterpret_cast<uintptr_t>(grid) & 7u) return FC3D_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    if (C != 1) {
        if (int e = zero_fill(s, g_img, (size_t)B * Hs * Ws * C * sizeof(float))) return e;
        const long np = (long)Ht * Wt;
        FC3D_LAUNCH("bilinear_bwd_img_atomic", bilinear_bwd_img_atomic_kernel, dim3(cdiv(np, NT), B), dim3(NT), 0, s, g_out, grid, mask_u8,
                           Hs, Ws, C, np, g_img);
        FC3D_LAUNCH_CHECK();
        return FC3D_OK;
    }
    int tiles_x;
    const int ntiles = wave_tiles(Ht, Wt, &tiles_x);
    const dim3 gs(cdiv(ntiles, NT / 64), B), blk(NT);
    if (!workspace) {   // window sums go straight into the zeroed gradient image with float atomics
        if (int e = zero_fill(s, g_img, (size_t)B * Hs * Ws * sizeof(float))) return e;
        FC3D_LAUNCH("bilinear_bwd_img_scatter", bilinear_bwd_img_scatter_kernel<true>, gs, blk, 0, s, g_out, grid, mask_u8, Hs, Ws, Ht, Wt, tiles_x,
                       ntiles, (TileMeta*)nullptr, (float*)nullptr, g_img);
        return FC3D_OK;
    }
    if (!aligned16(workspace)) return FC3D_EALIGN;
    TileMeta* meta = (TileMeta*)workspace;
    float* slab = (float*)(meta + (size_t)B * ntiles);
    FC3D_LAUNCH("bilinear_bwd_img_scatter", bilinear_bwd_img_scatter_kernel<false>, gs, blk, 0, s, g_out, grid, mask_u8, Hs, Ws, Ht, Wt, tiles_x,
                       ntiles, meta, slab, g_img);
    FC3D_LAUNCH_CHECK();
    FC3D_LAUNCH("bilinear_bwd_img_merge", bilinear_bwd_img_merge_kernel, dim3(cdiv(Ws, MT_X), cdiv(Hs, MT_Y), B), blk, 0, s,
                       (const TileMeta*)meta, (const float*)slab, ntiles, Hs, Ws, g_img);
    FC3D_LAUNCH_CHECK();
    FC3D_LAUNCH("bilinear_bwd_img_overflow", bilinear_bwd_img_overflow_kernel, gs, blk, 0, s, g_out, grid, mask_u8, Hs, Ws, Ht, Wt, tiles_x,
                       ntiles, (const TileMeta*)meta, g_img);
    FC3D_LAUNCH_CHECK();
    return FC3D_OK;
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

extern "C" int fc3d_warp_p2c_fwd(void* stream, const float* Ip, const float* coord_p, int64_t p_sb, int64_t p_sc,
                                 int64_t p_sh, int64_t p_sw, const float* coord_c, int64_t c_sb, int64_t c_sc,
                                 int64_t c_sh, int64_t c_sw, const float* T_p2c, const float* T_c2p, int B, int H,
                                 int W, float* Ic_masked, float* grid_p2c, float* grid_c2p, uint8_t* mask_u8,
                                 int mask_is_clear) {
    if (!Ip || !coord_p || !coord_c || !T_p2c || !T_c2p || !Ic_masked || !grid_p2c || !grid_c2p || !mask_u8)
        return FC3D_EINVAL;
    if (int e = check_img(B, H, W)) return e;
    if (!aligned4(coord_p) || !aligned4(coord_c) || !aligned4(Ip) || !aligned4(Ic_masked) ||
        (reinterpret_cast<uintptr_t>(grid_p2c) & 7u) || (reinterpret_cast<uintptr_t>(grid_c2p) & 7u))
        return FC3D_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const bool fma = use_fma(H, W);
    dim3 g(cdiv((long)H * W, PX_PER_BLOCK), B), blk(NT);
    if (!mask_is_clear) {
        if (int e = zero_fill(s, mask_u8, (size_t)B * H * W)) return e;
    }
    {
        const int layout = pick_layout(coord_p, p_sc, p_sh, p_sw, W);
#define ARGS coord_p, (long)p_sb, (long)p_sc, (long)p_sh, (long)p_sw, T_p2c, H, W, grid_p2c, mask_u8
        FC3D_DISPATCH_LAYOUT("warp_project_splat", project_kernel, layout, fma, true);
#undef ARGS
        FC3D_LAUNCH_CHECK();
    }
    {
        const int layout = pick_layout(coord_c, c_sc, c_sh, c_sw, W);
#define ARGS coord_c, (long)c_sb, (long)c_sc, (long)c_sh, (long)c_sw, T_c2p, Ip, (const uint8_t*)mask_u8, H, W, grid_c2p, Ic_masked
        FC3D_DISPATCH_LAYOUT("warp_sample", warp_sample_kernel, layout, fma);
#undef ARGS
        FC3D_LAUNCH_CHECK();
    }
    return FC3D_OK;
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
