// The capture simulator around the image-formation path (forward only; the reference runs it
// without gradients): surface-slope shading of the pattern (utils_pytorch.py:251-280), the warp of
// the coded pattern taken straight from the two depth maps (train_pytorch.py:86-87, 97, 107-116)
// and the sensor model (train_pytorch.py:100-104, 118).
//
// The composed operators move the scene points ([B,H,W,4] twice), both sampling grids and an
// [B,H,W,3] concatenation through HBM only to consume them once.  Here a scene point is formed
// in registers from its depth and the separable ray table, the grids are written only if asked
// for, and the camera-view scene coordinates are blended from the taps the pattern already uses:
// 4 B/px in and marks out for the forward splat, 9 B/px in and 16 B/px out (plus gathers that hit
// the caches) for the inverse warp.  Every value that decides an index, a mask bit or a threshold
// is computed with the operations and the order of the composed operators, so the outputs are
// bit-identical to theirs.
#include "geom_dev.h"

namespace fc3d {

// VEC = 4 (W divisible by 4, 16-byte aligned planes): four pixels of a row per lane, the row below as a
// second 16-byte load and the right neighbour of the last pixel from the next lane (one extra load
// only where the next lane is not the next group of the same row).
template <int VEC>
__global__ __launch_bounds__(NT) void reflectance_kernel(const float* __restrict__ I, long i_sb,
                                                         const float* __restrict__ z, int H, int W,
                                                         float* __restrict__ out) {
    const int b = blockIdx.y;
    const unsigned HW = (unsigned)H * (unsigned)W;
    const float* zb = z + (long)b * HW;
    const float* Ib = I + (long)b * i_sb;
    float* ob = out + (long)b * HW;
    if (VEC == 4) {
        const unsigned p = (blockIdx.x * NT + threadIdx.x) * 4u;
        const bool live = p < HW;
        const unsigned q = live ? p : 0u;
        const unsigned y = q / (unsigned)W, x = q - y * (unsigned)W;
        const bool has_b = y + 1 < (unsigned)H;
        const float4 z0 = ld_at(reinterpret_cast<const float4*>(zb), q >> 2);
        const float4 z1 = has_b ? ld_at(reinterpret_cast<const float4*>(zb), (q + W) >> 2) : z0;
        const float4 Iv = ld_at(reinterpret_cast<const float4*>(Ib), q >> 2);
        // the next lane holds pixels q+4..q+7; valid as a right neighbour when it exists and is in the same row
        float zn = __shfl_down(z0.x, 1, 64);
        const bool has_r = x + 4 < (unsigned)W;
        if (has_r && (threadIdx.x & 63) == 63) zn = zb[q + 4];
        if (!live) return;
        float4 o;
        o.x = shade(Iv.x, z0.x, z0.y, z1.x, true, has_b);
        o.y = shade(Iv.y, z0.y, z0.z, z1.y, true, has_b);
        o.z = shade(Iv.z, z0.z, z0.w, z1.z, true, has_b);
        o.w = shade(Iv.w, z0.w, zn, z1.w, has_r, has_b);
        st_at(reinterpret_cast<float4*>(ob), q >> 2, o);
    } else {
        const unsigned p = blockIdx.x * NT + threadIdx.x;
        if (p >= HW) return;
        const unsigned y = p / (unsigned)W, x = p - y * (unsigned)W;
        const bool has_r = x + 1 < (unsigned)W, has_b = y + 1 < (unsigned)H;
        const float z0 = zb[p];
        ob[p] = shade(Ib[p], z0, has_r ? zb[p + 1] : z0, has_b ? zb[p + W] : z0, has_r, has_b);
    }
}

// Scene point of pixel p at depth z: (z * ray_x[x], z * ray_y[y], z * 1, 1), the products rounded
// separately like the stored point cloud of the composed path.
__device__ __forceinline__ void point_of(float z, const float* __restrict__ ray_x, const float* __restrict__ ray_y,
                                         unsigned p, unsigned W, float c[4]) {
    const unsigned y = p / W, x = p - y * W;
    c[0] = __fmul_rn(z, ray_x[x]);
    c[1] = __fmul_rn(z, ray_y[y]);
    c[2] = __fmul_rn(z, 1.0f);
    c[3] = 1.0f;
}

// Forward warp: projector pixel -> camera pixel, 'all' visibility splat; the grid is optional.
// CPX pixels per lane, 64 apart like the other streaming kernels (2, 4 and 8 measured the same
// 25 us at 32x512x512, 14 us of it without the four one-byte splat stores per pixel).
constexpr int CPX = 4;

// POSE: K*pose is formed here (and K*inverse(pose) left in memory for the inverse warp) instead of read.
template <bool FMA, bool POSE = false>
__global__ __launch_bounds__(NT) void capture_project_kernel(const float* __restrict__ z_p,
                                                             const float* __restrict__ ray_x,
                                                             const float* __restrict__ ray_y,
                                                             const float* __restrict__ Tm, int H, int W,
                                                             float* __restrict__ grid, uint8_t* __restrict__ mask_u8,
                                                             PoseArgs pa) {
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    const long p0 = (long)blockIdx.x * (NT * CPX) + (threadIdx.x >> 6) * (64 * CPX) + (threadIdx.x & 63);
    const float* zb = z_p + (long)b * HW;
    float z[CPX];
#pragma unroll
    for (int i = 0; i < CPX; ++i) {
        const long p = p0 + 64 * i;
        z[i] = zb[p < HW ? p : HW - 1];
    }
    float T[16];
    if (POSE) {
        form_transform(pa, b, T);
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) T[k] = Tm[(long)b * 16 + k];
    }
    float2* go = grid ? reinterpret_cast<float2*>(grid) + (long)b * HW : nullptr;
    uint8_t* m = mask_u8 + (long)b * HW;
#pragma unroll
    for (int i = 0; i < CPX; ++i) {
        const long p = p0 + 64 * i;
        if (p < HW) {
            float c[4], gx, gy;
            point_of(z[i], ray_x, ray_y, (unsigned)p, (unsigned)W, c);
            project_px<FMA>(T, c[0], c[1], c[2], c[3], gx, gy);
            if (go) go[p] = make_float2(gx, gy);
            splat_px<uint8_t>(m, gx, gy, W, HW, 0, (uint8_t)1);
        }
    }
}

// torch.max's ordering: a NaN wins and then stays.
__device__ __forceinline__ float max_nan(float m, float e) { return (e > m || e != e) ? e : m; }

__device__ __forceinline__ float exposed(float ic, float off005, float g) {
    return __fadd_rn(__fadd_rn(ic, off005), __fmul_rn(g, 0.005f));
}

__device__ __forceinline__ float block_max_nan(float m, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max_nan(m, __shfl_xor(m, o, 64));
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) sh[wave] = m;
    __syncthreads();
    float r = sh[0];
#pragma unroll
    for (int w = 1; w < NT / 64; ++w) r = max_nan(r, sh[w]);
    return r;
}

// Inverse warp: camera pixel -> projector pixel; samples the coded pattern and the projector-frame
// scene coordinates (ray_x, ray_y, z_p) with the same four taps and composites both with the
// visibility mask.
// LDSRAY: the two ray tables (W + H floats) are staged in LDS first, which turns six of the fifteen
// per-pixel gathers into LDS reads; used while the tables are small enough not to cost occupancy.
constexpr int RAY_LDS_MAX = 4096;         // floats

template <bool FMA, bool LDSRAY>
__global__ __launch_bounds__(NT) void capture_sample_kernel(const float* __restrict__ z_c,
                                                            const float* __restrict__ ray_x_g,
                                                            const float* __restrict__ ray_y_g,
                                                            const float* __restrict__ Tm, const float* __restrict__ Ip,
                                                            const float* __restrict__ z_p,
                                                            const uint8_t* __restrict__ mask_u8, int H, int W,
                                                            float* __restrict__ grid, float* __restrict__ Ic,
                                                            float* __restrict__ xyz, const float* __restrict__ offset,
                                                            const float* __restrict__ gauss,
                                                            float* __restrict__ peak_partial) {
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    const long p0 = lane_first_pixel();
    const float* T = Tm + (long)b * 16;
    const float* zb = z_c + (long)b * HW;
    const float* img = Ip + (long)b * HW;
    const float* zp = z_p + (long)b * HW;
    const uint8_t* mk = mask_u8 + (long)b * HW;
    // No branch around a load (a pixel past the image reads the last pixel's operands and stores nothing):
    // behind a branch the loads in flight are unknown at the join, and the next pixel's taps would wait for
    // this pixel's to return.
    float z[PPL], mv[PPL], gn[PPL];
    uint8_t mb[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const long pc = min(p0 + 64 * i, HW - 1);
        z[i] = ld_pol<kNtCapDepth>(zb + pc);
        mb[i] = mk[pc];
        gn[i] = peak_partial ? gauss[(long)b * HW + pc] : 0.0f;
    }
    extern __shared__ float ray_lds[];    // [W] ray_x, then [H] ray_y
    const float* ray_x = ray_x_g;
    const float* ray_y = ray_y_g;
    if (LDSRAY) {
        for (int i = threadIdx.x; i < W + H; i += NT) ray_lds[i] = i < W ? ray_x_g[i] : ray_y_g[i - W];
        __syncthreads();
        ray_x = ray_lds;
        ray_y = ray_lds + W;
    }
    float gx[PPL], gy[PPL], i00[PPL], i01[PPL], i10[PPL], i11[PPL], z00[PPL], z01[PPL], z10[PPL], z11[PPL];
    float rx0[PPL], rx1[PPL], ry0[PPL], ry1[PPL];
    Taps t[PPL];
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const long pc = min(p0 + 64 * i, HW - 1);
        mv[i] = (float)mb[i];
        float c[4];
        point_of(z[i], ray_x, ray_y, (unsigned)pc, (unsigned)W, c);
        project_px<FMA>(T, c[0], c[1], c[2], c[3], gx[i], gy[i]);
        t[i] = make_taps(gx[i], gy[i], H, W);
        const unsigned a00 = (unsigned)t[i].y0 * W + t[i].x0, a01 = (unsigned)t[i].y1 * W + t[i].x0;
        const unsigned a10 = (unsigned)t[i].y0 * W + t[i].x1, a11 = (unsigned)t[i].y1 * W + t[i].x1;
        i00[i] = ld_at(img, a00); i01[i] = ld_at(img, a01); i10[i] = ld_at(img, a10); i11[i] = ld_at(img, a11);
        z00[i] = ld_at(zp, a00); z01[i] = ld_at(zp, a01); z10[i] = ld_at(zp, a10); z11[i] = ld_at(zp, a11);
        rx0[i] = ray_x[t[i].x0]; rx1[i] = ray_x[t[i].x1];
        ry0[i] = ray_y[t[i].y0]; ry1[i] = ray_y[t[i].y1];
    }
    float2* go = grid ? reinterpret_cast<float2*>(grid) + (long)b * HW : nullptr;
    float* oo = Ic + (long)b * HW;
    float* xo = xyz + (long)b * HW * 3;
    // the sensor model's per-image peak (max of Ic + exposure offset + read noise) on the side: one partial
    // maximum per workgroup, so that no separate pass over Ic and the noise is needed for it
    float peak = -INFINITY;
    const float off005 = peak_partial ? __fmul_rn(offset[b], 0.05f) : 0.0f;
#pragma unroll
    for (int i = 0; i < PPL; ++i) {
        const long p = p0 + 64 * i;
        if (p < HW) {
            if (go) go[p] = make_float2(gx[i], gy[i]);
            const float icv = __fmul_rn(blend(t[i], i00[i], i01[i], i10[i], i11[i]), mv[i]);
            peak = max_nan(peak, exposed(icv, off005, gn[i]));
            oo[p] = icv;
            st_pol<kNtCapXyz>(xo + p * 3 + 0, __fmul_rn(blend(t[i], rx0[i], rx0[i], rx1[i], rx1[i]), mv[i]));
            st_pol<kNtCapXyz>(xo + p * 3 + 1, __fmul_rn(blend(t[i], ry0[i], ry1[i], ry0[i], ry1[i]), mv[i]));
            st_pol<kNtCapXyz>(xo + p * 3 + 2, __fmul_rn(blend(t[i], z00[i], z01[i], z10[i], z11[i]), mv[i]));
        }
    }
    if (peak_partial) {
        __shared__ float sh_peak[NT / 64];
        peak = block_max_nan(peak, sh_peak);
        if (threadIdx.x == 0) peak_partial[(long)b * gridDim.x + blockIdx.x] = peak;
    }
}

// ---------------------------------------------------------------------------------------
// sensor model: exposure offset + read noise, per-image peak normalisation, gain, clipping
// ---------------------------------------------------------------------------------------
constexpr int POST_MAX_CHUNKS = 256;      // partial maxima per image
constexpr int POST_PX_PER_CHUNK = 4096;

static inline int post_chunks(long HW) {
    const long n = (HW + POST_PX_PER_CHUNK - 1) / POST_PX_PER_CHUNK;
    return (int)(n < 1 ? 1 : (n > POST_MAX_CHUNKS ? POST_MAX_CHUNKS : n));
}
// pixels per chunk, a multiple of 4 so that chunks of a 4-divisible image start on 16 bytes
static inline long post_span(long HW, int chunks) { return ((HW + 4L * chunks - 1) / (4L * chunks)) * 4; }

template <int VEC>
__global__ __launch_bounds__(NT) void capture_peak_kernel(const float* __restrict__ Ic, const float* __restrict__ offset,
                                                          const float* __restrict__ gauss, long HW, long span,
                                                          float* __restrict__ partial) {
    __shared__ float sh[NT / 64];
    const int b = blockIdx.y;
    const long lo = (long)blockIdx.x * span;
    const long hi = lo + span < HW ? lo + span : HW;
    const float off005 = __fmul_rn(offset[b], 0.05f);
    const float* ic = Ic + (long)b * HW;
    const float* g = gauss + (long)b * HW;
    float m = -INFINITY;
    if (VEC == 4) {
        for (long p = lo + 4L * threadIdx.x; p < hi; p += 4L * NT) {
            const float4 a = *reinterpret_cast<const float4*>(ic + p);
            const float4 n = *reinterpret_cast<const float4*>(g + p);
            m = max_nan(m, exposed(a.x, off005, n.x));
            m = max_nan(m, exposed(a.y, off005, n.y));
            m = max_nan(m, exposed(a.z, off005, n.z));
            m = max_nan(m, exposed(a.w, off005, n.w));
        }
    } else {
        for (long p = lo + threadIdx.x; p < hi; p += NT) m = max_nan(m, exposed(ic[p], off005, g[p]));
    }
    m = block_max_nan(m, sh);
    if (threadIdx.x == 0) partial[(long)b * gridDim.x + blockIdx.x] = m;
}

__device__ __forceinline__ void sensor_px(float ic, float off005, float g, float peak, float gain, float vis,
                                          float& scaled, float& valid) {
    const float e = exposed(ic, off005, g);
    float s = __fmul_rn(e / peak, gain);
    s = s < 0.0f ? 0.0f : s;            // clamp keeps a NaN, like torch.clamp
    s = s > 1.0f ? 1.0f : s;
    scaled = s;
    valid = __fmul_rn(s > 0.05f ? 1.0f : 0.0f, vis);
}

template <int VEC>
__global__ __launch_bounds__(NT) void capture_sensor_kernel(const float* __restrict__ Ic, const float* __restrict__ offset,
                                                            const float* __restrict__ gauss,
                                                            const float* __restrict__ scale,
                                                            uint8_t* mask_u8,
                                                            const float* __restrict__ partial, int nparts, long HW,
                                                            long span, float* __restrict__ Ic_scaled,
                                                            float* __restrict__ Ic_mask, int rezero) {
    __shared__ float sh[NT / 64];
    const int b = blockIdx.y;
    float m = -INFINITY;
    for (int c = threadIdx.x; c < nparts; c += NT) m = max_nan(m, partial[(long)b * nparts + c]);
    const float peak = block_max_nan(m, sh);
    const float off005 = __fmul_rn(offset[b], 0.05f);
    const float gain = __fadd_rn(__fmul_rn(scale[b], 0.3f), 0.7f);
    const long lo = (long)blockIdx.x * span;
    const long hi = lo + span < HW ? lo + span : HW;
    const float* ic = Ic + (long)b * HW;
    const float* g = gauss + (long)b * HW;
    uint8_t* mk = mask_u8 + (long)b * HW;
    float* so = Ic_scaled + (long)b * HW;
    float* mo = Ic_mask + (long)b * HW;
    // rezero: this kernel is the last reader of the visibility mask and hands it back cleared (every byte
    // is read and then cleared by the same lane), so the caller's next capture needs no clear
    if (VEC == 4) {
        for (long p = lo + 4L * threadIdx.x; p < hi; p += 4L * NT) {
            const float4 a = *reinterpret_cast<const float4*>(ic + p);
            const float4 n = ld_pol<kNtCapNoise>(reinterpret_cast<const float4*>(g + p));
            const uchar4 v = *reinterpret_cast<const uchar4*>(mk + p);
            if (rezero) *reinterpret_cast<uchar4*>(mk + p) = make_uchar4(0, 0, 0, 0);
            float4 s, w;
            sensor_px(a.x, off005, n.x, peak, gain, (float)v.x, s.x, w.x);
            sensor_px(a.y, off005, n.y, peak, gain, (float)v.y, s.y, w.y);
            sensor_px(a.z, off005, n.z, peak, gain, (float)v.z, s.z, w.z);
            sensor_px(a.w, off005, n.w, peak, gain, (float)v.w, s.w, w.w);
            st_pol<kNtCapOut>(reinterpret_cast<float4*>(so + p), s);
            st_pol<kNtCapOut>(reinterpret_cast<float4*>(mo + p), w);
        }
    } else {
        for (long p = lo + threadIdx.x; p < hi; p += NT) {
            const float v = (float)mk[p];
            if (rezero) mk[p] = 0;
            sensor_px(ic[p], off005, g[p], peak, gain, v, so[p], mo[p]);
        }
    }
}

// ---------------------------------------------------------------------------------------
// k x k box mean with zero padding (avg_pool2d, stride 1, padding k/2, padded cells counted) and
// the two things the training loop does with it: local normalisation of the capture that feeds
// the networks, and erosion of a 0/1 mask
// ---------------------------------------------------------------------------------------
constexpr int BOX_TW = 64, BOX_TH = 32, BOX_MAX_K = 33;
enum BoxMode { kBoxMean = 0, kBoxLocalNorm = 1, kBoxErode = 2 };

// One 64 x 32 output tile per workgroup: the tile and its halo go to LDS once (zeros outside the
// image).  Row sums over k columns are formed into a second LDS array by lanes that each walk 16
// pixels of one row with a running sum (add the entering value, subtract the leaving one; lanes sit
// on different rows and the row pitch is odd, so the reads spread over the banks), then every
// lane walks 8 rows down its column the same way.  Sums of a 0/1 mask are exact, so the erosion
// is bit-exact; the mean differs from a row-major window sum by rounding only (the running sums
// restart every 16 columns / 8 rows).
constexpr int BOX_SEG = 16;               // pixels a lane walks in the row pass

__host__ __device__ inline int box_pitch(int n) { return n | 1; }

template <int MODE>
__global__ __launch_bounds__(NT) void box_kernel(const float* __restrict__ in, int H, int W, int r, float c,
                                                 float* __restrict__ out) {
    extern __shared__ float box_lds[];
    const int k = 2 * r + 1;
    const int LW = BOX_TW + 2 * r, LH = BOX_TH + 2 * r;
    const int TP = box_pitch(LW), HP = box_pitch(BOX_TW);
    float* tile = box_lds;                 // [LH][TP]
    float* hs = box_lds + LH * TP;         // [LH][HP]
    const int b = blockIdx.z;
    const int x0 = blockIdx.x * BOX_TW, y0 = blockIdx.y * BOX_TH;
    const float* img = in + (long)b * H * W;
    {
        // six rows per wavefront at a time, every load from a clamped (always valid) address and
        // issued before the first use: one memory latency per batch instead of one per row
        constexpr int BATCH = 6, WAVES = NT / 64;
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const int gxa = x0 - r + lane, gxb = gxa + 64;          // LW <= 128: two columns per lane cover a row
        const bool oka = gxa >= 0 && gxa < W, okb = gxb >= 0 && gxb < W;
        const unsigned ca = (unsigned)min(max(gxa, 0), W - 1), cb = (unsigned)min(max(gxb, 0), W - 1);
        for (int tb = wv; tb < LH; tb += WAVES * BATCH) {
            float va[BATCH], vb[BATCH];
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const unsigned row = (unsigned)min(max(y0 - r + tb + WAVES * i, 0), H - 1) * (unsigned)W;
                va[i] = ld_at(img, row + ca);
                vb[i] = ld_at(img, row + cb);
            }
#pragma unroll
            for (int i = 0; i < BATCH; ++i) {
                const int ty = tb + WAVES * i, gy = y0 - r + ty;
                if (ty < LH) {
                    const bool rowok = gy >= 0 && gy < H;
                    tile[ty * TP + lane] = (rowok && oka) ? va[i] : 0.0f;
                    if (lane + 64 < LW) tile[ty * TP + lane + 64] = (rowok && okb) ? vb[i] : 0.0f;
                }
            }
        }
    }
    __syncthreads();
    for (int job = threadIdx.x; job < LH * (BOX_TW / BOX_SEG); job += NT) {
        const int ty = job % LH, seg = job / LH;           // neighbouring lanes: neighbouring rows
        const float* row = tile + ty * TP + seg * BOX_SEG;
        float* dst = hs + ty * HP + seg * BOX_SEG;
        float a = 0.0f;
        for (int d = 0; d < k; ++d) a += row[d];
        dst[0] = a;
#pragma unroll
        for (int i = 1; i < BOX_SEG; ++i) {
            a = (a + row[i + k - 1]) - row[i - 1];
            dst[i] = a;
        }
    }
    __syncthreads();
    const int x = threadIdx.x & 63;
    const int gx = x0 + x;
    if (gx >= W) return;
    const float area = (float)(k * k);
    float* o = out + (long)b * H * W;
    constexpr int ROWS = BOX_TH / (NT / 64);
    const int ty0 = (threadIdx.x >> 6) * ROWS;
    const float* col = hs + ty0 * HP + x;
    float a = 0.0f;
    for (int d = 0; d < k; ++d) a += col[d * HP];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const int gy = y0 + ty0 + j;
        if (j > 0) a = (a + col[(j + k - 1) * HP]) - col[(j - 1) * HP];
        if (gy >= H) continue;
        const float mean = a / area;
        float v;
        if (MODE == kBoxMean) v = mean;
        else if (MODE == kBoxLocalNorm) v = tile[(ty0 + j + r) * TP + x + r] / (mean + c);
        else v = mean > 0.999f ? 1.0f : 0.0f;
        o[(long)gy * W + gx] = v;
    }
}

}  // namespace fc3d

using namespace fc3d;

extern "C" int fc3d_box_filter(void* stream, const float* in, int B, int H, int W, int k, int mode, float c,
                               float* out) {
    if (!in || !out || in == out || k < 1 || k > BOX_MAX_K || (k & 1) == 0 || mode < 0 || mode > 2) return FC3D_EINVAL;
    if (int e = check_img(B, H, W)) return e;
    const int r = k / 2;
    const size_t lds = (size_t)(BOX_TH + 2 * r) * (box_pitch(BOX_TW + 2 * r) + box_pitch(BOX_TW)) * sizeof(float);
    const dim3 g(cdiv(W, BOX_TW), cdiv(H, BOX_TH), B), blk(NT);
    hipStream_t s = (hipStream_t)stream;
    if (mode == kBoxMean) FC3D_LAUNCH("box_mean", box_kernel<kBoxMean>, g, blk, lds, s, in, H, W, r, c, out);
    else if (mode == kBoxLocalNorm) FC3D_LAUNCH("box_local_norm", box_kernel<kBoxLocalNorm>, g, blk, lds, s, in, H, W, r, c, out);
    else FC3D_LAUNCH("box_erode", box_kernel<kBoxErode>, g, blk, lds, s, in, H, W, r, c, out);
    return FC3D_OK;
}

extern "C" int fc3d_reflectance_fwd(void* stream, const float* I, int64_t i_sb, const float* z, int B, int H, int W,
                                    float* out) {
    if (!I || !z || !out || i_sb < 0) return FC3D_EINVAL;
    if (int e = check_img(B, H, W)) return e;
    const long HW = (long)H * W;
    if (W % 4 == 0 && i_sb % 4 == 0 && aligned16(I) && aligned16(z) && aligned16(out))
        FC3D_LAUNCH("reflectance", reflectance_kernel<4>, dim3(cdiv(HW, NT * 4), B), dim3(NT), 0, (hipStream_t)stream, I,
                    (long)i_sb, z, H, W, out);
    else
        FC3D_LAUNCH("reflectance", reflectance_kernel<1>, dim3(cdiv(HW, NT), B), dim3(NT), 0, (hipStream_t)stream, I,
                    (long)i_sb, z, H, W, out);
    return FC3D_OK;
}

// forward warp + splat, then inverse warp.  pose != NULL: the transforms are outputs of the first launch.
// peak_partial != NULL: the inverse warp also leaves cdiv(H*W, 512) partial maxima of the exposed image per sample.
static int capture_warp_launch(hipStream_t s, const float* Ip, const float* z_p, const float* z_c, const float* ray_x,
                               const float* ray_y, float* T_p2c, float* T_c2p, const float* pose, const float* K,
                               int64_t k_sb, int B, int H, int W, float* Ic_masked, float* xyz_cview, uint8_t* mask_u8,
                               int mask_is_clear, float* grid_p2c, float* grid_c2p, const float* offset,
                               const float* gauss, float* peak_partial) {
    if (!Ip || !z_p || !z_c || !ray_x || !ray_y || !T_p2c || !T_c2p || !Ic_masked || !xyz_cview || !mask_u8)
        return FC3D_EINVAL;
    if (peak_partial && (!offset || !gauss)) return FC3D_EINVAL;
    if (int e = check_img(B, H, W)) return e;
    if ((long)H * W >= (1L << 29)) return FC3D_ERANGE;   // per-image byte offsets of the taps are 32-bit
    if ((reinterpret_cast<uintptr_t>(grid_p2c) & 7u) || (reinterpret_cast<uintptr_t>(grid_c2p) & 7u)) return FC3D_EALIGN;
    dim3 g(cdiv((long)H * W, PX_PER_BLOCK), B), gp(cdiv((long)H * W, NT * CPX), B), blk(NT);
    if (!mask_is_clear) {
        if (int e = zero_fill(s, mask_u8, (size_t)B * H * W)) return e;
    }
    const bool lds = W + H <= RAY_LDS_MAX;
    const size_t nlds = lds ? (size_t)(W + H) * sizeof(float) : 0;
    const PoseArgs pa = {pose, K, (long)k_sb, T_p2c, T_c2p, nullptr};
#define PROJECT_ARGS z_p, ray_x, ray_y, (const float*)T_p2c, H, W, grid_p2c, mask_u8, pa
#define SAMPLE_ARGS z_c, ray_x, ray_y, (const float*)T_c2p, Ip, z_p, (const uint8_t*)mask_u8, H, W, grid_c2p, Ic_masked, xyz_cview, offset, gauss, peak_partial
#define WARP_LAUNCH(FMA)                                                                                       \
    do {                                                                                                       \
        if (pose) FC3D_LAUNCH("capture_project_splat", (capture_project_kernel<FMA, true>), gp, blk, 0, s, PROJECT_ARGS); \
        else FC3D_LAUNCH("capture_project_splat", (capture_project_kernel<FMA, false>), gp, blk, 0, s, PROJECT_ARGS); \
        if (lds) FC3D_LAUNCH("capture_sample", (capture_sample_kernel<FMA, true>), g, blk, nlds, s, SAMPLE_ARGS); \
        else FC3D_LAUNCH("capture_sample", (capture_sample_kernel<FMA, false>), g, blk, 0, s, SAMPLE_ARGS);    \
    } while (0)
    if (use_fma(H, W)) WARP_LAUNCH(true);
    else WARP_LAUNCH(false);
#undef WARP_LAUNCH
#undef SAMPLE_ARGS
#undef PROJECT_ARGS
    return FC3D_OK;
}

extern "C" int fc3d_capture_warp_fwd(void* stream, const float* Ip, const float* z_p, const float* z_c,
                                     const float* ray_x, const float* ray_y, const float* T_p2c, const float* T_c2p,
                                     int B, int H, int W, float* Ic_masked, float* xyz_cview, uint8_t* mask_u8,
                                     int mask_is_clear, float* grid_p2c, float* grid_c2p) {
    return capture_warp_launch((hipStream_t)stream, Ip, z_p, z_c, ray_x, ray_y, const_cast<float*>(T_p2c),
                               const_cast<float*>(T_c2p), nullptr, nullptr, 0, B, H, W, Ic_masked, xyz_cview, mask_u8,
                               mask_is_clear, grid_p2c, grid_c2p, nullptr, nullptr, nullptr);
}

extern "C" size_t fc3d_capture_post_workspace(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * post_chunks((long)H * W) * sizeof(float);
}

static int sensor_launch(hipStream_t s, const float* Ic, const float* offset, const float* gauss, const float* scale,
                         uint8_t* mask_u8, int B, long HW, const float* partial, int nparts, float* Ic_scaled,
                         float* Ic_mask, int rezero) {
    const int chunks = post_chunks(HW);
    const long span = post_span(HW, chunks);
    const bool vec = HW % 4 == 0 && aligned16(Ic) && aligned16(gauss) && aligned16(Ic_scaled) && aligned16(Ic_mask) &&
                     aligned4(mask_u8);
    dim3 g(chunks, B), blk(NT);
    if (vec)
        FC3D_LAUNCH("capture_sensor", capture_sensor_kernel<4>, g, blk, 0, s, Ic, offset, gauss, scale, mask_u8, partial,
                    nparts, HW, span, Ic_scaled, Ic_mask, rezero);
    else
        FC3D_LAUNCH("capture_sensor", capture_sensor_kernel<1>, g, blk, 0, s, Ic, offset, gauss, scale, mask_u8, partial,
                    nparts, HW, span, Ic_scaled, Ic_mask, rezero);
    return FC3D_OK;
}

extern "C" int fc3d_capture_post(void* stream, const float* Ic, const float* offset, const float* gauss,
                                 const float* scale, const uint8_t* mask_u8, int B, int H, int W, float* Ic_scaled,
                                 float* Ic_mask, void* workspace) {
    if (!Ic || !offset || !gauss || !scale || !mask_u8 || !Ic_scaled || !Ic_mask || !workspace) return FC3D_EINVAL;
    if (int e = check_img(B, H, W)) return e;
    if (!aligned4(workspace)) return FC3D_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const long HW = (long)H * W;
    const int chunks = post_chunks(HW);
    const long span = post_span(HW, chunks);
    const bool vec = HW % 4 == 0 && aligned16(Ic) && aligned16(gauss);
    float* partial = (float*)workspace;
    dim3 g(chunks, B), blk(NT);
    if (vec) FC3D_LAUNCH("capture_peak", capture_peak_kernel<4>, g, blk, 0, s, Ic, offset, gauss, HW, span, partial);
    else FC3D_LAUNCH("capture_peak", capture_peak_kernel<1>, g, blk, 0, s, Ic, offset, gauss, HW, span, partial);
    return sensor_launch(s, Ic, offset, gauss, scale, const_cast<uint8_t*>(mask_u8), B, HW, partial, chunks, Ic_scaled,
                         Ic_mask, 0);
}

extern "C" size_t fc3d_capture_fwd_workspace(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return ((size_t)B * 32 + (size_t)B * cdiv((long)H * W, PX_PER_BLOCK)) * sizeof(float);
}

extern "C" int fc3d_capture_fwd(void* stream, const float* Ip, const float* z_p, const float* z_c, const float* ray_x,
                                const float* ray_y, const float* pose, const float* K, int64_t k_sb, const float* offset,
                                const float* gauss, const float* scale, int B, int H, int W, float* Ic_masked,
                                float* xyz_cview, float* Ic_scaled, float* Ic_mask, uint8_t* mask_u8, int mask_is_clear,
                                int mask_rezero, void* workspace, float* grid_p2c, float* grid_c2p) {
    if (!pose || !K || !offset || !gauss || !scale || !Ic_scaled || !Ic_mask || !workspace) return FC3D_EINVAL;
    if (!aligned4(workspace)) return FC3D_EALIGN;
    if (B <= 0 || H <= 0 || W <= 0) return FC3D_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* T_p2c = (float*)workspace;
    float* T_c2p = T_p2c + (size_t)B * 16;
    float* partial = T_c2p + (size_t)B * 16;
    if (int e = capture_warp_launch(s, Ip, z_p, z_c, ray_x, ray_y, T_p2c, T_c2p, pose, K, k_sb, B, H, W, Ic_masked,
                                    xyz_cview, mask_u8, mask_is_clear, grid_p2c, grid_c2p, offset, gauss, partial))
        return e;
    return sensor_launch(s, Ic_masked, offset, gauss, scale, mask_u8, B, (long)H * W, partial,
                         cdiv((long)H * W, PX_PER_BLOCK), Ic_scaled, Ic_mask, mask_rezero);
}
