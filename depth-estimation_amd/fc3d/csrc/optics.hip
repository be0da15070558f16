// Depth-layered PSF blur (codePattern, optics_pytorch.py:71-121) for gfx950.
//
// Every pixel's depth layer is derived from z on the fly (no [B,H,W,L] one-hot mask ever
// exists).  The per-image maximum is found without atomics: each tile / strip writes one packed
// 64-bit key (ordered value | position priority) and the normalisation pass reduces an image's
// keys.  The gradient flows to channel 0 of the pattern only.
//
// Four kernel families, picked by mode and PSF size:
//   rolling-row (3x3, 5x5, 7x7): one pixel column per lane, column neighbours through wavefront
//       shifts, PSF weights of the lane's layer in registers;
//   flat (is_conv_psf=False; after the rolling-row kernels): no blur, 1024 consecutive pixels per
//       wavefront, pure streaming;
//   layer sweep (9x9 ... 33x33 odd; after the flat kernels): 16 x 16 pixels per wavefront, input
//       tile in LDS, one depth layer at a time with the PSF in scalar registers;
//   generic 64x16 LDS tiles (everything else: even or non-square PSFs, 1x1):
//       the backward pass first sums g*out per image (deterministic two-level sum), then
//       evaluates the transposed convolution in gather form: every output pixel walks its K x K
//       neighbours p and uses the PSF of *p's* layer.
#include "fc3d_common.h"
#include <type_traits>

namespace fc3d {

constexpr int TX = 64;    // tile width  (16 lanes x 4 pixels)
constexpr int TY = 16;    // tile height
constexpr int NT = 256;   // threads per workgroup
constexpr int kLdsBudget = 60 * 1024;

// All fp32, one rounding per line, matching torch's tensor ops on the CPU:
// reciprocal(z) - 1/z0, * coeff, / step (a true division), round-half-even.
__device__ __forceinline__ int layer_of(float z, const fc3d_layer_grid& g) {
    float r = 1.0f / z;
    float d = r - g.inv_z0;
    float wm = g.coeff * d;
    float n = wm / g.step;
    float rr = rintf(n);
    float lf = rintf(rr - g.centre0);
    bool ok = (lf >= 0.0f) && (lf < (float)g.n_layers) && (fabsf(rr - (g.centre0 + lf)) < 0.01f);
    return ok ? (int)lf : -1;
}

// layer_of with the two true divisions replaced by v_rcp_f32 and a multiplication by 1/step,
// plus a flag that says when the shortcut could have changed the result.  With n the exact
// chain's quotient and n' the shortcut's, |n' - n| <= eps * (|coeff/step| * |1/z| * 1.5 + 3|n|)
// (eps = 2^-23: one ulp of the reciprocal, carried through the subtraction, and the roundings of
// the two products); the layer depends on n only through rint(n), so whenever n' is further
// than that bound from every half-integer both chains round to the same integer.  The bound is
// evaluated, with a fourfold margin, for the largest |n| and |1/z| a pixel with an in-range
// layer can have; pixels flagged ambiguous (a few in 10^4) are redone with layer_of.  Out-of-range,
// infinite and NaN depths come out as -1 on both chains.
struct LayerFast {
    float rstep, thr;
    float a, nb;      // n' = a / z - b as one fused multiply-add: a = coeff / step, nb = -a * inv_z0
};
__device__ __forceinline__ LayerFast layer_fast_setup(const fc3d_layer_grid& g) {
    const float eps = 1.1920929e-7f;
    const float nmax = fabsf(g.centre0) + (float)g.n_layers + 2.0f;
    const float gain = fabsf(g.coeff / g.step);
    const float rmax = fabsf(g.inv_z0) + nmax / gain;
    LayerFast f;
    f.rstep = 1.0f / g.step;
    f.thr = 4.0f * eps * (1.5f * gain * rmax + 3.0f * nmax);   // NaN/inf for a degenerate grid: everything is redone exactly
    f.a = g.coeff * f.rstep;
    f.nb = -(f.a * g.inv_z0);
    return f;
}
// The host's promise behind INTC0: centre0 is an integer (|centre0| < 2^22).  Then rr - centre0 is exact, the
// layer is that difference itself, and the third test of layer_of, |rr - (centre0 + l)| < 0.01, always holds.
static inline bool layer_grid_centre_is_integer(const fc3d_layer_grid& g) {
    return g.centre0 == rintf(g.centre0) && fabsf(g.centre0) < 4194304.0f;
}
template <bool INTC0 = false>
__device__ __forceinline__ int layer_fast(float z, const fc3d_layer_grid& g, const LayerFast& f, bool& ambiguous) {
    const float r = __builtin_amdgcn_rcpf(z);
    float n;
    if (INTC0) {
        // One fused multiply-add for subtraction and both scalings.  With N the real-number value and G = coeff/step:
        // |n - N| <= eps (1.5 G |1/z| + 1.5 |N|) (one ulp of the reciprocal, the roundings of a, nb and of the fused
        // operation itself), and the exact chain's quotient is within eps (0.5 G |1/z| + 1.5 |N|) of N: together
        // inside f.thr, which allows eps (6 G |1/z| + 12 |N|).
        n = fmaf(r, f.a, f.nb);
    } else {
        const float d = r - g.inv_z0;
        const float wm = g.coeff * d;
        n = wm * f.rstep;
    }
    const float rr = rintf(n);
    ambiguous = !(fabsf(n - rr) <= 0.5f - f.thr) && (fabsf(n) < 1.0e9f);
    if (INTC0) {
        const float lf = rr - g.centre0;
        const bool ok = (lf >= 0.0f) && (lf < (float)g.n_layers);
        return ok ? (int)lf : -1;
    }
    const float lf = rintf(rr - g.centre0);
    const bool ok = (lf >= 0.0f) && (lf < (float)g.n_layers) && (fabsf(rr - (g.centre0 + lf)) < 0.01f);
    return ok ? (int)lf : -1;
}

struct TileGeom {
    int th, tw, S;   // staged rows, staged columns, LDS row stride (floats)
};
__host__ __device__ inline TileGeom tile_geom(int Kh, int Kw) {
    TileGeom t;
    t.th = TY + Kh - 1;
    t.tw = TX + Kw - 1;
    t.S = (t.tw + 1 + 3) & ~3;   // +1: the sliding window pre-loads one column past the tile
    return t;
}

__device__ __forceinline__ void block_minmax(int lmin, int lmax, int* s_min, int* s_max) {
    lmin = wave_min(lmin);
    lmax = wave_max(lmax);
    if (threadIdx.x == 0) {
        *s_min = 1 << 30;
        *s_max = -1;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        atomicMin(s_min, lmin);
        atomicMax(s_max, lmax);
    }
    __syncthreads();
}

template <bool VEC>
__global__ __launch_bounds__(NT) void code_pattern_fwd_kernel(
    const float* __restrict__ pat0, long sb, long sh, long sw, const float* __restrict__ z,
    const float* __restrict__ psf, int H, int W, int Kh, int Kw, int conv, int cap_layers,
    fc3d_layer_grid lg, float* __restrict__ raw, uint8_t* __restrict__ layer_u8,
    unsigned long long* __restrict__ tilekey) {
    extern __shared__ float smem[];
    __shared__ int s_lmin, s_lmax;
    __shared__ unsigned long long s_key[NT / 64];

    const int b = blockIdx.z;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x = x0 + 4 * tx, y = y0 + ty;
    const long img = (long)b * H * W;
    const int KK = Kh * Kw;

    // -- 1. depth layer of my four pixels
    float zv[4] = {1.0f, 1.0f, 1.0f, 1.0f};
    bool inb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) inb[i] = (y < H) && (x + i < W);
    if (VEC) {
        if (inb[0]) {
            float4 v = *reinterpret_cast<const float4*>(z + img + (long)y * W + x);
            zv[0] = v.x; zv[1] = v.y; zv[2] = v.z; zv[3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (inb[i]) zv[i] = z[img + (long)y * W + x + i];
    }
    int lay[4];
    int lmin = 1 << 30, lmax = -1;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        lay[i] = inb[i] ? layer_of(zv[i], lg) : -1;
        if (lay[i] >= 0) {
            lmin = min(lmin, lay[i]);
            lmax = max(lmax, lay[i]);
        }
    }

    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (conv) {
        const TileGeom tg = tile_geom(Kh, Kw);
        float* tile = smem;
        float* wts = smem + tg.th * tg.S;
        block_minmax(lmin, lmax, &s_lmin, &s_lmax);
        const int bl0 = s_lmin;
        const int nl = (s_lmax >= bl0) ? min(s_lmax - bl0 + 1, cap_layers) : 0;

        // -- 2. stage pattern tile + halo (zero 'same' padding) and the PSFs of layers bl0..bl0+nl-1
        const int padh = (Kh - 1) / 2, padw = (Kw - 1) / 2;
        const float* pimg = pat0 + (long)b * sb;
        for (int r = tid / 64; r < tg.th; r += NT / 64) {
            const int gy = y0 - padh + r;
            const bool rowok = (gy >= 0) && (gy < H);
            for (int c = tid & 63; c < tg.S; c += 64) {
                const int gx = x0 - padw + c;
                float v = 0.0f;
                if (rowok && c < tg.tw && gx >= 0 && gx < W) v = pimg[(long)gy * sh + (long)gx * sw];
                tile[r * tg.S + c] = v;
            }
        }
        const float* wsrc = psf + (long)bl0 * KK;
        for (int i = tid; i < nl * KK; i += NT) wts[i] = wsrc[i];
        __syncthreads();

        // -- 3. accumulate
        const bool any = (lay[0] >= 0) | (lay[1] >= 0) | (lay[2] >= 0) | (lay[3] >= 0);
        const int l0 = lay[0];
        const bool uniform = (l0 >= 0) && (lay[1] == l0) && (lay[2] == l0) && (lay[3] == l0) && (l0 - bl0 < nl);
        if (uniform) {
            const float* wl = wts + (l0 - bl0) * KK;
            for (int dy = 0; dy < Kh; ++dy) {
                const float* trow = tile + (ty + dy) * tg.S + 4 * tx;
                const float* wrow = wl + dy * Kw;
                float s0 = trow[0], s1 = trow[1], s2 = trow[2], s3 = trow[3];
                for (int dx = 0; dx < Kw; ++dx) {
                    const float w = wrow[dx];
                    const float nx = trow[dx + 4];
                    acc[0] = fmaf(w, s0, acc[0]);
                    acc[1] = fmaf(w, s1, acc[1]);
                    acc[2] = fmaf(w, s2, acc[2]);
                    acc[3] = fmaf(w, s3, acc[3]);
                    s0 = s1; s1 = s2; s2 = s3; s3 = nx;
                }
            }
        } else if (any) {
            const float* wp[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int l = max(lay[i], 0);
                // layers beyond the staged range (tile spans more layers than fit in LDS) come from global
                wp[i] = (l >= bl0 && l - bl0 < nl) ? (wts + (l - bl0) * KK) : (psf + (long)l * KK);
            }
            for (int dy = 0; dy < Kh; ++dy) {
                const float* trow = tile + (ty + dy) * tg.S + 4 * tx;
                for (int dx = 0; dx < Kw; ++dx) {
                    const int k = dy * Kw + dx;
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = fmaf(wp[i][k], trow[dx + i], acc[i]);
                }
            }
        }
    } else {
        // is_conv_psf=False: the pattern itself, gated by the in-range mask
        const float* pimg = pat0 + (long)b * sb;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (inb[i]) acc[i] = pimg[(long)y * sh + (long)(x + i) * sw];
    }

    // -- 4. gate by the layer mask (multiply, so a non-finite blur still poisons the pixel
    //       as conv*0 does in the reference), store, and publish the tile's max key
    unsigned long long key = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        acc[i] = acc[i] * (lay[i] >= 0 ? 1.0f : 0.0f);
        if (inb[i]) {
            const uint32_t pos = 0xffffffffu - (uint32_t)((x + i) * H + y);   // first w, then first h
            const unsigned long long k = ((unsigned long long)float_to_key(acc[i]) << 32) | pos;
            key = k > key ? k : key;
        }
    }
    if (VEC) {
        if (inb[0]) {
            *reinterpret_cast<float4*>(raw + img + (long)y * W + x) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            if (layer_u8)
                *reinterpret_cast<uchar4*>(layer_u8 + img + (long)y * W + x) =
                    make_uchar4((uint8_t)lay[0], (uint8_t)lay[1], (uint8_t)lay[2], (uint8_t)lay[3]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (inb[i]) {
                raw[img + (long)y * W + x + i] = acc[i];
                if (layer_u8) layer_u8[img + (long)y * W + x + i] = (uint8_t)lay[i];
            }
    }
    key = wave_max(key);
    if ((tid & 63) == 0) s_key[tid >> 6] = key;
    __syncthreads();
    if (tid == 0) {
        unsigned long long k = s_key[0];
#pragma unroll
        for (int i = 1; i < NT / 64; ++i) k = s_key[i] > k ? s_key[i] : k;
        tilekey[(long)b * gridDim.x * gridDim.y + blockIdx.y * gridDim.x + blockIdx.x] = k;
    }
}

template <int AHEAD>
__device__ __forceinline__ unsigned long long block_max_key(const unsigned long long* keys, int n,
                                                            unsigned long long* s_red) {
    // AHEAD loads in flight per thread (indices past the end re-read the last key): one round trip per
    // AHEAD x blockDim keys.  Pays where an image has thousands of keys; with a few hundred the extra loads
    // cost more than the two round trips they save.
    unsigned long long k = 0;
    for (int i = threadIdx.x; i < n; i += AHEAD * (int)blockDim.x) {
        unsigned long long v[AHEAD];
#pragma unroll
        for (int j = 0; j < AHEAD; ++j) v[j] = keys[min(i + j * (int)blockDim.x, n - 1)];
#pragma unroll
        for (int j = 0; j < AHEAD; ++j) k = v[j] > k ? v[j] : k;
    }
    k = wave_max(k);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = k;
    __syncthreads();
    k = s_red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) k = s_red[i] > k ? s_red[i] : k;
    return k;
}

constexpr int NORM_V = 4;                 // float4s per thread of the normalisation pass
// Every workgroup reduces all max keys of its image.  With thousands of them (one per 16x16 tile: 4,096 at
// 1024 x 1024) that is a chain of dependent round trips per thread unless several loads are in flight.
// Measured: 576 keys (three trips) are faster left alone (eight loads instead of three: +1.9 us at 32 x 512^2).
constexpr int NORM_KEYS_PLAIN = 3 * NT;   // up to here: the plain loop
constexpr int NORM_KEYS_FOUR = 4 * NT;    // up to here: four loads in flight (one trip); beyond: eight

template <bool VEC, int AHEAD>
__global__ __launch_bounds__(NT) void code_pattern_norm_kernel(
    const float* __restrict__ raw, const unsigned long long* __restrict__ tilekey, int ntiles, int H, int W,
    float* __restrict__ out, float* __restrict__ max_out, int* __restrict__ argmax_out) {
    __shared__ unsigned long long s_red[NT / 64];
    const int b = blockIdx.y;
    const long HW = (long)H * W;
    const long base = (long)b * HW;
    // the image data is requested before the key reduction so that the two latencies overlap
    float4 v[NORM_V];
    if (VEC) {
#pragma unroll
        for (int j = 0; j < NORM_V; ++j) {
            const long i = (((long)blockIdx.x * NORM_V + j) * NT + threadIdx.x) * 4;
            v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < HW) v[j] = ld_pol<kNtNormLoad>(reinterpret_cast<const float4*>(raw + base + i));
        }
    }
    const unsigned long long key = block_max_key<AHEAD>(tilekey + (long)b * ntiles, ntiles, s_red);
    const float m = key_to_float((uint32_t)(key >> 32));
    const float den = m + 1e-10f;
    const float rden = 1.0f / den;      // one rounding more than raw / den: far inside the fp tolerance
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint32_t pos = 0xffffffffu - (uint32_t)(key & 0xffffffffu);
        max_out[b] = m;
        argmax_out[b] = (int)(pos % (uint32_t)H) * W + (int)(pos / (uint32_t)H);
    }
    if (VEC) {
#pragma unroll
        for (int j = 0; j < NORM_V; ++j) {
            const long i = (((long)blockIdx.x * NORM_V + j) * NT + threadIdx.x) * 4;
            if (i < HW)
                st_pol<kNtNormStore>(reinterpret_cast<float4*>(out + base + i),
                                     make_float4(v[j].x * rden, v[j].y * rden, v[j].z * rden, v[j].w * rden));
        }
    } else {
        const long i0 = (long)blockIdx.x * NORM_V * NT * 4;
        for (long i = i0 + threadIdx.x; i < i0 + (long)NORM_V * NT * 4 && i < HW; i += NT)
            out[base + i] = raw[base + i] * rden;
    }
}

// ---------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------
constexpr int RED_PX = NT * 4;   // pixels summed by one workgroup of the first-level reduction

__device__ __forceinline__ float block_sum(float v, float* s_red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = s_red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) t += s_red[i];
    return t;
}

template <bool VEC>
__global__ __launch_bounds__(NT) void code_pattern_bwd_reduce_kernel(
    const float* __restrict__ g, const float* __restrict__ out, long HW, float* __restrict__ partial) {
    __shared__ float s_red[NT / 64];
    const int b = blockIdx.y;
    const long base = (long)b * HW;
    const long i0 = ((long)blockIdx.x * NT + threadIdx.x) * 4;
    float s = 0.0f;
    if (VEC) {
        if (i0 < HW) {
            const float4 a = *reinterpret_cast<const float4*>(g + base + i0);
            const float4 c = *reinterpret_cast<const float4*>(out + base + i0);
            s = fmaf(a.x, c.x, s); s = fmaf(a.y, c.y, s); s = fmaf(a.z, c.z, s); s = fmaf(a.w, c.w, s);
        }
    } else {
        for (int k = 0; k < 4; ++k)
            if (i0 + k < HW) s = fmaf(g[base + i0 + k], out[base + i0 + k], s);
    }
    s = block_sum(s, s_red);
    if (threadIdx.x == 0) partial[(long)b * gridDim.x + blockIdx.x] = s;
}

template <bool VEC>
__global__ __launch_bounds__(NT) void code_pattern_bwd_kernel(
    const float* __restrict__ g, const uint8_t* __restrict__ layer_u8, const float* __restrict__ max_in,
    const int* __restrict__ argmax_in, const float* __restrict__ partial, int npartial,
    const float* __restrict__ psf, int H, int W, int L, int Kh, int Kw, int conv, int cap_layers,
    float* __restrict__ g_pat0) {
    extern __shared__ float smem[];
    __shared__ int s_lmin, s_lmax;
    __shared__ float s_red[NT / 64];

    const int b = blockIdx.z;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int x = x0 + 4 * tx, y = y0 + ty;
    const long img = (long)b * H * W;
    const int KK = Kh * Kw;

    // sum(g*out) over the image: fixed strided order + fixed tree => reproducible
    float s = 0.0f;
    for (int i = tid; i < npartial; i += NT) s += partial[(long)b * npartial + i];
    const float S = block_sum(s, s_red);
    const float den = max_in[b] + 1e-10f;
    const float corr = S / den;       // d/d(max) of sum(g * raw/den) = -sum(g*out)/den
    const int am = argmax_in[b];

    auto graw = [&](int gy, int gx) -> float {
        const long p = (long)gy * W + gx;
        const uint8_t l = layer_u8[img + p];
        if (l == 255) return 0.0f;
        float v = g[img + p] / den;
        if ((int)p == am) v -= corr;
        return v;
    };

    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if (!conv) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (y < H && x + i < W) acc[i] = graw(y, x + i);
    } else {
        const TileGeom tg = tile_geom(Kh, Kw);
        float* tile = smem;                                   // graw with halo
        float* wts = smem + tg.th * tg.S;                     // PSFs of the staged layer range
        uint8_t* ltile = reinterpret_cast<uint8_t*>(wts + (long)cap_layers * KK);
        const int padh = (Kh - 1) / 2, padw = (Kw - 1) / 2;
        const int hb = Kh - 1 - padh, wb = Kw - 1 - padw;     // halo before the tile (rows / columns)
        int lmin = 1 << 30, lmax = -1;
        for (int r = tid / 64; r < tg.th; r += NT / 64) {
            const int gy = y0 - hb + r;
            const bool rowok = (gy >= 0) && (gy < H);
            for (int c = tid & 63; c < tg.S; c += 64) {
                const int gx = x0 - wb + c;
                float v = 0.0f;
                uint8_t l = 255;
                if (rowok && c < tg.tw && gx >= 0 && gx < W) {
                    l = layer_u8[img + (long)gy * W + gx];
                    v = graw(gy, gx);
                    if (l != 255) {
                        lmin = min(lmin, (int)l);
                        lmax = max(lmax, (int)l);
                    }
                }
                tile[r * tg.S + c] = v;
                ltile[r * tg.S + c] = l;
            }
        }
        block_minmax(lmin, lmax, &s_lmin, &s_lmax);
        const int bl0 = s_lmin;
        const int nl = (s_lmax >= bl0) ? min(s_lmax - bl0 + 1, cap_layers) : 0;
        const float* wsrc = psf + (long)bl0 * KK;
        for (int i = tid; i < nl * KK; i += NT) wts[i] = wsrc[i];
        __syncthreads();

        // g_pat0[q] = sum_d PSF[layer(p)][d] * graw[p],  p = q - d + pad.
        // Staged row of p: ty + (Kh-1-dy); staged column: 4*tx + i + (Kw-1-dx).
        for (int dy = 0; dy < Kh; ++dy) {
            const int r = ty + (Kh - 1 - dy);
            const float* trow = tile + r * tg.S + 4 * tx;
            const uint8_t* lrow = ltile + r * tg.S + 4 * tx;
            for (int e = 0; e < Kw + 3; ++e) {
                const float gv = trow[e];
                const int l = lrow[e];
                if (l == 255) continue;
                const float* wl = (l - bl0 < nl) ? (wts + (l - bl0) * KK) : (psf + (long)l * KK);
                wl += dy * Kw;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int ee = e - i;
                    if (ee >= 0 && ee < Kw) acc[i] = fmaf(wl[Kw - 1 - ee], gv, acc[i]);
                }
            }
        }
    }
    if (VEC) {
        if (y < H && x < W)
            *reinterpret_cast<float4*>(g_pat0 + img + (long)y * W + x) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (y < H && x + i < W) g_pat0[img + (long)y * W + x + i] = acc[i];
    }
}

// N floats (a multiple of 4) from a 16-byte aligned LDS address, as 16-byte reads.
template <int N>
__device__ __forceinline__ void lds_read4(const float* p, float (&dst)[N]) {
    const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
    for (int j = 0; j < N / 4; ++j) {
        const float4 v = q[j];
        dst[4 * j] = v.x; dst[4 * j + 1] = v.y; dst[4 * j + 2] = v.z; dst[4 * j + 3] = v.w;
    }
}

// ---------------------------------------------------------------------------------------
// Rolling-row kernels for small odd square PSFs (3x3, 5x5, 7x7).
//
// A wavefront owns a strip 64 lanes wide (one pixel column per lane) and walks down R output
// rows.  Neighbouring columns are reached with wavefront shifts (DPP), not through LDS, so the
// strips overlap by K-1 columns and only the inner 64-(K-1) lanes store results.  LDS holds
// nothing but the PSF stack; there is no barrier after the prologue.
//
// Backward (scatter form): source pixel p = (y, x) of layer l adds PSF[l][dy][dx] * graw[p] to
// output (y + dy - pad, x + dx - pad).  Each lane keeps the K*K weights of its own pixel's layer
// in registers (reloaded from LDS only on rows where some lane's layer differs from the row
// above) and adds its K*K products into rolling accumulators, one per output row and column
// offset; when an output row is complete the K column sums are handed sideways Horner fashion,
// one shift per column step.
// ---------------------------------------------------------------------------------------
#ifndef FC3D_SHADE_WAVES
#define FC3D_SHADE_WAVES 5
#endif
#ifndef FC3D_ROLL_PK
#define FC3D_ROLL_PK 1       // packed multiply-adds in the rolling-row kernels (tuning switch)
#endif
#ifndef FC3D_ROLL_ORDER
#define FC3D_ROLL_ORDER 1    // PSF loads ahead of the image loads, image loads in the order the rows are consumed (tuning switch)
#endif
typedef float f2 __attribute__((ext_vector_type(2)));
// acc + w * (v.x, v.x): a packed multiply-add whose second operand takes its low half for both lanes of the pair
// (op_sel_hi), so that a value shared by both products needs no copy into a register pair of its own per use.
__device__ __forceinline__ f2 pk_fma_lo(f2 w, f2 v, f2 acc) {
    f2 r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(w), "v"(v), "v"(acc));
    return r;
}
template <int K>
struct Roll {
    static constexpr int PAD = (K - 1) / 2;
    static constexpr int SW = 64 - (K - 1);      // output columns per strip
    static constexpr int KK = K * K;
    static constexpr int KS = K + 1;             // floats per PSF row in LDS and registers: even, so that the pairs a
                                                 // packed multiply-add takes (columns 0-1, 2-3, ...) start on even registers
    static constexpr int KKP = (K * KS + 3) & ~3;   // one layer's PSF, padded to whole float4s
    static constexpr int FWD_WAVES = K <= 5 ? 6 : 3;   // wavefronts per SIMD the forward kernel is compiled for
    static constexpr int FWD_WAVES_SHADE = K <= 5 ? FC3D_SHADE_WAVES : 3;   // ... with the shading folded in (a depth value per pattern row more)
};

// lane i receives lane i-1's value (lane 0: zero) / lane i+1's value (lane 63: zero)
__device__ __forceinline__ float from_lane_below(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138, 0xf, 0xf, true));   // wave_shr:1
}
__device__ __forceinline__ float from_lane_above(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130, 0xf, 0xf, true));   // wave_shl:1
}

// The PSF stack in two halves: its (cache-resident) global loads are issued FIRST, ahead of a strip's image
// loads -- loads return in issue order, so behind them the weights would only be usable once every image
// row had arrived, and every wavefront would sit through the whole memory latency before its first
// multiply-add -- and the values go to LDS after the image loads have been issued.
constexpr int kPsfStageMax = 8;      // staged floats per thread: L * KKP <= 8 * NT, else the plain loop below
template <int K>
struct PsfStage {
    float v[kPsfStageMax];
    unsigned pad;        // bit u: this thread's slot of round u is padding (of a PSF row or of the layer)
};
template <int K>
__device__ __forceinline__ bool psf_stage_fits(int L) { return L * Roll<K>::KKP <= kPsfStageMax * NT; }
// (unsigned index arithmetic -- a signed division by a constant costs three times the instructions -- and only
// the rounds the stack has slots for: at 21 layers of 5x5 three of the eight.  This runs once per workgroup, but a
// workgroup's strips are short: done the plain way it was a quarter of the kernel's vector instructions.)
template <int K>
__device__ __forceinline__ void psf_stage_load(PsfStage<K>& st, const float* __restrict__ psf, int L) {
    constexpr unsigned KK = Roll<K>::KK, KKP = Roll<K>::KKP, KS = Roll<K>::KS;
    const unsigned n = (unsigned)L * KKP;
    st.pad = 0u;
    // loads from clamped (always valid) addresses, inside wave-uniform branches only
#pragma unroll
    for (int u = 0; u < kPsfStageMax; ++u) {
        st.v[u] = 0.0f;
        if ((unsigned)(u * NT) < n) {
            const unsigned i = min((unsigned)(u * NT) + threadIdx.x, n - 1u);
            const unsigned l = i / KKP, j = i - l * KKP;
            const unsigned dy = j / KS, dx = j - dy * KS;
            st.pad |= (unsigned)((dy >= (unsigned)K) | (dx >= (unsigned)K)) << u;
            st.v[u] = ld_at(psf, l * KK + min(dy, (unsigned)(K - 1)) * K + min(dx, (unsigned)(K - 1)));
        }
    }
}
template <int K>
__device__ __forceinline__ void psf_stage_store(const PsfStage<K>& st, float* __restrict__ wts, int L) {
    constexpr unsigned KKP = Roll<K>::KKP;
    const unsigned n = (unsigned)L * KKP;
#pragma unroll
    for (int u = 0; u < kPsfStageMax; ++u) {
        if ((unsigned)(u * NT) < n) {
            const unsigned i = (unsigned)(u * NT) + threadIdx.x;
            if (i < n) wts[i] = ((st.pad >> u) & 1u) ? 0.0f : st.v[u];
        }
    }
}
// a * b for wave-uniform a on the scalar unit, opaque to the compiler: folded into the lane arithmetic that
// follows, the product became a 64-bit vector multiply-add whose unused upper input register was the target of
// a load in flight, and the wavefront waited for that load before it could issue its own
__device__ __forceinline__ int scalar_mul(int a, int b) {
    int r;
    asm("s_mul_i32 %0, %1, %2" : "=s"(r) : "s"(a), "s"(b));
    return r;
}
template <int K>
__device__ __forceinline__ void stage_psf_stack(float* __restrict__ wts, const float* __restrict__ psf, int L) {
    constexpr int KK = Roll<K>::KK, KKP = Roll<K>::KKP, KS = Roll<K>::KS;
    for (int i = threadIdx.x; i < L * KKP; i += NT) {
        const int l = i / KKP, j = i - l * KKP;
        const int dy = j / KS, dx = j - dy * KS;
        wts[i] = (dy < K && dx < K) ? psf[l * KK + dy * K + dx] : 0.0f;
    }
}

//
// The gradient that reaches the per-image maximum (-sum(g*out)/den, added to the one pixel that
// holds the maximum) is not known before g and out have been read once.  Each strip therefore
// also sums g*out over the pixels it owns into one partial per wavefront, and a follow-up kernel
// (one workgroup per image) adds the partials in a fixed order and applies the K x K footprint
// of that single pixel.  This saves a separate pass over g and out.
//
// DOT: sum(g*out) is known already -- the caller hands over partial sums of it (dot_partial, ndot per
// image; the sampler's backward forms them while it writes g).  Then `out` is not read at all and the
// maximum's share goes in at the source: the pixel that holds the maximum scatters (g - S)/den
// instead of g/den, so there is no follow-up kernel either.  Only a wavefront whose source rows and
// columns contain that pixel adds up the partials (fixed order).
template <int K, int R, bool DOT = false>
__global__ __launch_bounds__(NT) void code_pattern_bwd_roll_kernel(
    const float* __restrict__ g, const float* __restrict__ out, const uint8_t* __restrict__ layer_u8,
    const float* __restrict__ max_in, const float* __restrict__ psf, int H, int W, int L, int nstrips,
    float* __restrict__ partial, float* __restrict__ g_pat0, const int* __restrict__ argmax_in,
    const float* __restrict__ dot_partial, int ndot) {
    constexpr int PAD = Roll<K>::PAD, SW = Roll<K>::SW, KKP = Roll<K>::KKP, KS = Roll<K>::KS;
    constexpr int NR = R + K - 1;                 // source rows a strip reads
    extern __shared__ float smem[];               // [L][KKP]
    const int b = blockIdx.y;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const long img = (long)b * H * W;

    // (row block, strip) pairs are numbered along one grid axis, so that only the last workgroup
    // of an image can hold wavefronts without work
    const int wid = __builtin_amdgcn_readfirstlane(block_x() * (NT / 64) + wv);   // the same for all lanes: keep it scalar
    const long pidx = (long)b * (gridDim.x * (NT / 64)) + wid;
    const int strip = wid % nstrips;
    const int xs = scalar_mul(strip, SW);         // first output column of this wavefront's strip
    const int x = xs - PAD + lane;
    const int y0 = (wid / nstrips) * R;
    const bool colok = x >= 0 && x < W;
    const int xc = min(max(x, 0), W - 1);
    const float* gi = g + img;              // uniform 64-bit bases + 32-bit per-lane offsets
    const uint8_t* li = layer_u8 + img;
    const float* oi = out + img;
    float* gp = g_pat0 + img;

    // every source row up front, from clamped (always valid) addresses, in the order the rows are
    // consumed and behind the (cache-resident) PSF stack: one memory latency per strip
    float gv[NR], ov[R];
    int lv[NR];
    const bool staged = FC3D_ROLL_ORDER && psf_stage_fits<K>(L);     // the same in every thread of the launch
    PsfStage<K> pst;
    __builtin_amdgcn_sched_barrier(0);      // see the forward kernel
    if (staged) psf_stage_load<K>(pst, psf, L);
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const unsigned q = (unsigned)(min(max(y0 - PAD + r, 0), H - 1) * W + xc);
        gv[r] = ld_at_pol<kNtPsfBwdLoad>(gi, q);
        lv[r] = ld_at_pol<kNtPsfBwdLoad>(li, q);
    }
    if (!DOT) {
#pragma unroll
        for (int o = 0; o < R; ++o) ov[o] = ld_at(oi, (unsigned)(min(y0 + o, H - 1) * W + xc));
    }
    const float den = max_in[b] + 1e-10f;
    const float rden = 1.0f / den;
    // DOT: the maximum's pixel (py, px); sfix = S in the lane that holds column px of a wavefront
    // whose source rows contain py, zero everywhere else
    int py = -1;
    float sfix = 0.0f;
    if (DOT) {
        const int am = argmax_in[b];
        py = am / W;
        const int px = am - py * W;
        if (y0 < H && py >= y0 - PAD && py < y0 - PAD + NR && px >= xs - PAD && px < xs - PAD + 64) {   // the same for all lanes
            float sp = 0.0f;
            for (int i = lane; i < ndot; i += 64) sp += dot_partial[(long)b * ndot + i];
            sp = wave_sum(sp);
            sfix = (colok && x == px) ? sp : 0.0f;
        }
    }
    if (staged) psf_stage_store<K>(pst, smem, L);
    else stage_psf_stack<K>(smem, psf, L);
    __syncthreads();
    if (y0 >= H) {
        if (!DOT && lane == 0) partial[pidx] = 0.0f;
        return;
    }
    const bool mine = lane >= PAD && lane < 64 - PAD && x < W;   // columns this lane writes
    if (!DOT) {
        float sgo = 0.0f;
#pragma unroll
        for (int o = 0; o < R; ++o) sgo = fmaf((mine && y0 + o < H) ? gv[o + PAD] : 0.0f, ov[o], sgo);
        sgo = wave_sum(sgo);
        if (lane == 0) partial[pidx] = sgo;
    }

    float w[KKP];
    int lcur = 0;
    lds_read4<KKP>(smem, w);
    // acc[o][dx]: what this lane's pixels send to output row o, column x + dx - pad.  The sideways
    // hand-over waits until the row is complete: K-1 shifts per output row, not per source row.
    // Columns 0-1, 2-3, ... are pairs: one packed multiply-add (two per lane and instruction) serves both.
#if FC3D_ROLL_PK
    f2 accp[R][K / 2];
    float accl[R];
#pragma unroll
    for (int o = 0; o < R; ++o) {
#pragma unroll
        for (int j = 0; j < K / 2; ++j) accp[o][j] = f2{0.0f, 0.0f};
        accl[o] = 0.0f;
    }
    auto acc_at = [&](int o, int dx) -> float {
        return dx == K - 1 ? accl[o] : ((dx & 1) ? accp[o][dx >> 1].y : accp[o][dx >> 1].x);
    };
#else
    float acc[R][K];
#pragma unroll
    for (int o = 0; o < R; ++o)
#pragma unroll
        for (int dx = 0; dx < K; ++dx) acc[o][dx] = 0.0f;
    auto acc_at = [&](int o, int dx) -> float { return acc[o][dx]; };
#endif

#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int y = y0 - PAD + r;
        const bool ok = colok && y >= 0 && y < H && lv[r] != 255;
        const float gsrc = DOT ? gv[r] - (y == py ? sfix : 0.0f) : gv[r];
        const float v = ok ? gsrc * rden : 0.0f;
#if FC3D_ROLL_PK
        const f2 vv = {v, 0.0f};                  // only the low half is read (pk_fma_lo)
#endif
        const int l = ok ? lv[r] : lcur;          // pixels that contribute nothing keep the weights they have
        if (__ballot(l != lcur) != 0ull) {
            lds_read4<KKP>(smem + l * KKP, w);
            lcur = l;
        }
#pragma unroll
        for (int dy = 0; dy < K; ++dy) {
            const int o = r + dy - 2 * PAD;       // output row (from y0) fed by source row r through PSF row dy
            if (o < 0 || o >= R) continue;
#if FC3D_ROLL_PK
#pragma unroll
            for (int j = 0; j < K / 2; ++j)
                accp[o][j] = pk_fma_lo(f2{w[dy * KS + 2 * j], w[dy * KS + 2 * j + 1]}, vv, accp[o][j]);
            accl[o] = fmaf(w[dy * KS + K - 1], v, accl[o]);      // K is odd
#else
#pragma unroll
            for (int dx = 0; dx < K; ++dx) acc[o][dx] = fmaf(w[dy * KS + dx], v, acc[o][dx]);
#endif
        }
        if (r >= 2 * PAD) {
            const int o = r - 2 * PAD;            // complete: its last source row was this one
            const int yo = y0 + o;
            float up = acc_at(o, K - 1);          // travels to higher lanes, one column per step
#pragma unroll
            for (int dx = K - 2; dx > PAD; --dx) up = acc_at(o, dx) + from_lane_below(up);
            up = from_lane_below(up);
            float dn = acc_at(o, 0);              // travels to lower lanes
#pragma unroll
            for (int dx = 1; dx < PAD; ++dx) dn = acc_at(o, dx) + from_lane_above(dn);
            dn = from_lane_above(dn);
            const float res = (acc_at(o, PAD) + up) + dn;
            if (mine && yo < H) st_at_pol<kNtPsfBwdStore>(gp, (unsigned)(yo * W + x), res);
        }
    }
}

// One workgroup per image, after the kernel above: S = sum of the partials (fixed order), then the
// footprint of the maximum's pixel p: g_pat0[p + d - pad] -= S/den * PSF[layer(p)][d].
__global__ __launch_bounds__(NT) void code_pattern_bwd_max_kernel(
    const float* __restrict__ partial, int npartial, const float* __restrict__ max_in,
    const int* __restrict__ argmax_in, const uint8_t* __restrict__ layer_u8, const float* __restrict__ psf, int H,
    int W, int K, float* __restrict__ g_pat0) {
    __shared__ float s_red[NT / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    float s = 0.0f;
    for (int i = tid; i < npartial; i += NT) s += partial[(long)b * npartial + i];
    const float S = block_sum(s, s_red);
    const float corr = S / (max_in[b] + 1e-10f);
    const int am = argmax_in[b];
    const long img = (long)b * H * W;
    const int l = layer_u8[img + am];
    if (l == 255) return;                         // the maximum sits on a masked pixel (all-zero image): no gradient path
    const int py = am / W, px = am - py * W, pad = (K - 1) / 2;
    for (int d = tid; d < K * K; d += NT) {
        const int dy = d / K, dx = d - dy * K;
        const int qy = py + dy - pad, qx = px + dx - pad;
        if (qy >= 0 && qy < H && qx >= 0 && qx < W) {
            float* q = g_pat0 + img + (long)qy * W + qx;
            *q = *q - corr * (psf ? psf[(long)l * K * K + d] : 1.0f);
        }
    }
}

// Forward (gather form): every arriving pattern row is spread into its K column neighbours with
// K-1 shifts and kept for K rows; a finished output row takes K*K fused multiply-adds with the
// weights of its own pixel's layer (dy, then dx ascending, like the other forward kernels).
// One max key per wavefront strip goes to the normalisation pass.
// (K <= 5: six wavefronts per SIMD -- the kernel waits on memory more than it computes, and one
// more resident wavefront is worth the couple of spilled registers the limit costs; the 7x7
// window and weights need about 100 more registers and are left to the compiler's own choice)
// SHADE: the pattern is first multiplied by the surface-slope shading of the same depth map
// (withReflectance, utils_pytorch.py:251-280, then codePattern: train_pytorch.py:91-94), so that the shaded
// pattern never goes through memory: the depth is loaded for every pattern row of the strip plus the row
// below, the right neighbour comes from the next lane (lane 63 loads its own).
template <int K, int R, bool SHADE = false, bool INTC0 = false>
__global__ __launch_bounds__(NT)
__attribute__((amdgpu_waves_per_eu(SHADE ? Roll<K>::FWD_WAVES_SHADE : Roll<K>::FWD_WAVES, SHADE ? Roll<K>::FWD_WAVES_SHADE : Roll<K>::FWD_WAVES)))
void code_pattern_fwd_roll_kernel(
    const float* __restrict__ pat0, long sb, long sh, long sw, const float* __restrict__ z,
    const float* __restrict__ psf, int H, int W, int nstrips, fc3d_layer_grid lg, float* __restrict__ raw,
    uint8_t* __restrict__ layer_u8, unsigned long long* __restrict__ tilekey) {
    constexpr int PAD = Roll<K>::PAD, SW = Roll<K>::SW, KKP = Roll<K>::KKP, KS = Roll<K>::KS;
    constexpr int NR = R + K - 1;
    extern __shared__ float smem[];               // [L][KKP]
    const int b = blockIdx.y;
    const int tid = threadIdx.x, wv = tid >> 6, lane = tid & 63;
    const long img = (long)b * H * W;
    // (row block, strip) pairs along one grid axis: no idle strips; the same for all lanes, so kept scalar
    const int wid = __builtin_amdgcn_readfirstlane(block_x() * (NT / 64) + wv);
    const long kidx = (long)b * (gridDim.x * (NT / 64)) + wid;
    const int strip = wid % nstrips;
    const int xs = scalar_mul(strip, SW);
    const int x = xs - PAD + lane;
    const int y0 = (wid / nstrips) * R;
    const bool colok = x >= 0 && x < W;
    const bool mine = lane >= PAD && lane < 64 - PAD && x < W;   // columns this lane writes
    const int xc = min(max(x, 0), W - 1);
    // uniform 64-bit bases + 32-bit per-lane offsets (the host checked that they fit)
    const float* pimg = pat0 + (long)b * sb;
    const float* zimg = z + img;
    float* rimg = raw + img;
    uint8_t* limg = layer_u8 ? layer_u8 + img : nullptr;
    const unsigned pcol = (unsigned)xc * (unsigned)sw;

    float pv[NR], zv[R];
    const bool staged = FC3D_ROLL_ORDER && psf_stage_fits<K>(lg.n_layers);     // the same in every thread of the launch
    PsfStage<K> pst;
    __builtin_amdgcn_sched_barrier(0);      // the index arithmetic above stays above: hoisted over it, the PSF loads' target
                                            // registers got reused by it and the strip's own loads waited a round trip
    if (staged) psf_stage_load<K>(pst, psf, lg.n_layers);
    auto ld_p = [&](int r) {
        return ld_at_pol<kNtPsfFwdLoad>(pimg, (unsigned)min(max(y0 - PAD + r, 0), H - 1) * (unsigned)sh + pcol);
    };
    if (!SHADE) {
        // in the order the rows are consumed: output row o needs pattern rows up to o + K - 1 and depth row o
#pragma unroll
        for (int r = 0; r < K - 1; ++r) pv[r] = ld_p(r);
#pragma unroll
        for (int o = 0; o < R; ++o) {
            pv[o + K - 1] = ld_p(o + K - 1);
            zv[o] = ld_at_pol<kNtPsfFwdLoad>(zimg, (unsigned)(min(y0 + o, H - 1) * W + xc));
        }
    } else {
#pragma unroll
        for (int r = 0; r < NR; ++r) pv[r] = ld_p(r);
        float za[NR + 1], zr63[NR];       // depth of every pattern row and the one below; lane 63's right neighbours
#pragma unroll
        for (int r = 0; r <= NR; ++r) za[r] = ld_at(zimg, (unsigned)(min(max(y0 - PAD + r, 0), H - 1) * W + xc));
#pragma unroll
        for (int r = 0; r < NR; ++r)
            zr63[r] = lane == 63 ? ld_at(zimg, (unsigned)(min(max(y0 - PAD + r, 0), H - 1) * W + min(xc + 1, W - 1))) : 0.0f;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int y = y0 - PAD + r;
            float zright = from_lane_above(za[r]);
            if (lane == 63) zright = zr63[r];
            // rows and columns outside the image are zero padding below, whatever the shading says there
            pv[r] = shade(pv[r], za[r], zright, za[r + 1], x + 1 < W, y + 1 < H);
        }
#pragma unroll
        for (int o = 0; o < R; ++o) zv[o] = za[o + PAD];
    }
    const LayerFast lfast = layer_fast_setup(lg);
    if (staged) psf_stage_store<K>(pst, smem, lg.n_layers);
    else stage_psf_stack<K>(smem, psf, lg.n_layers);
    __syncthreads();
    if (y0 >= H) {
        if (lane == 0) tilekey[kidx] = 0ull;      // below every real key
        return;
    }

    float w[KKP];
    int lcur = 0;
    lds_read4<KKP>(smem, w);
    float win[NR][K];
    // the lane's maximum so far as value + row (a 64-bit key per pixel costs twice the instructions): a later
    // row replaces it only if it is greater -- so the first row wins ties, as the position part of the key
    // would decide -- or if it is the first NaN (torch.max returns NaN once one is present)
    float best = -INFINITY;
    int besto = 0;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const int y = y0 - PAD + r;
        win[r][PAD] = (colok && y >= 0 && y < H) ? pv[r] : 0.0f;     // zero 'same' padding
#pragma unroll
        for (int d = 1; d <= PAD; ++d) {
            win[r][PAD - d] = from_lane_below(win[r][PAD - d + 1]);
            win[r][PAD + d] = from_lane_above(win[r][PAD + d - 1]);
        }
        if (r >= K - 1) {
            const int o = r - (K - 1);
            const int yo = y0 + o;
            const bool inb = mine && yo < H;
            bool amb;
            int l = layer_fast<INTC0>(zv[o], lg, lfast, amb);
            if (__ballot(amb && inb) != 0ull) l = layer_of(zv[o], lg);   // rare: the exact chain for the whole row
            l = inb ? l : -1;
            const int lw = l >= 0 ? l : lcur;
            if (__ballot(lw != lcur) != 0ull) {
                lds_read4<KKP>(smem + lw * KKP, w);
                lcur = lw;
            }
            float a = 0.0f;
#if FC3D_ROLL_PK
            // packed multiply-adds: columns 0-1, 2-3, ... of a PSF row as pairs (even and odd partial sums), the
            // last column on its own; dy ascending within each
            // (one accumulator per column pair: a packed multiply-add that reads the result of the one before
            // it waits a state)
            f2 a2[K / 2];
#pragma unroll
            for (int j = 0; j < K / 2; ++j) a2[j] = f2{0.0f, 0.0f};
#pragma unroll
            for (int dy = 0; dy < K; ++dy) {
#pragma unroll
                for (int j = 0; j < K / 2; ++j)
                    a2[j] = __builtin_elementwise_fma(f2{w[dy * KS + 2 * j], w[dy * KS + 2 * j + 1]},
                                                      f2{win[o + dy][2 * j], win[o + dy][2 * j + 1]}, a2[j]);
                a = fmaf(w[dy * KS + K - 1], win[o + dy][K - 1], a);      // K is odd
            }
#pragma unroll
            for (int j = 0; j < K / 2; ++j) a += a2[j].x + a2[j].y;
#else
#pragma unroll
            for (int dy = 0; dy < K; ++dy)
#pragma unroll
                for (int dx = 0; dx < K; ++dx) a = fmaf(w[dy * KS + dx], win[o + dy][dx], a);
#endif
            a = a * (l >= 0 ? 1.0f : 0.0f);       // a multiply: a non-finite blur still poisons the pixel
            if (inb) {
                const bool take = (a > best) || (a != a && best == best);
                best = take ? a : best;
                besto = take ? o : besto;
                const unsigned q = (unsigned)(yo * W + x);
                st_at_pol<kNtPsfFwdStore>(rimg, q, a);
                if (limg) st_at(limg, q, (uint8_t)l);
            }
        }
    }
    unsigned long long key = 0;
    if (mine) {                                   // y0 < H: the lane has stored row 0 at least
        const uint32_t pos = 0xffffffffu - (uint32_t)(x * H + y0 + besto);   // first w, then first h
        key = ((unsigned long long)float_to_key(best) << 32) | pos;
    }
    key = wave_max(key);
    if (lane == 0) tilekey[kidx] = key;
}

// ---------------------------------------------------------------------------------------
// is_conv_psf=False (optics_pytorch.py:110-112; what the reference's training loop runs): the
// pattern gated by the in-range mask, no blur.  Pure streaming: every wavefront owns 1024
// consecutive pixels of an image, 16 per lane as four 16-byte groups 256 pixels apart, and
// publishes one max key (forward) or one partial of sum(g*out) (backward).
// ---------------------------------------------------------------------------------------
constexpr int FLAT_GROUPS = 4;                    // float4 groups per lane
constexpr int FLAT_WAVE_PX = 64 * 4 * FLAT_GROUPS;

__device__ __forceinline__ int flat_layer(float z, const fc3d_layer_grid& lg, const LayerFast& lf) {
    bool amb;
    int l = layer_fast(z, lg, lf, amb);
    if (amb) l = layer_of(z, lg);
    return l;
}

// SHADE: the pattern is first multiplied by the surface-slope shading of the same depth map
// (withReflectance, utils_pytorch.py:251-280, then codePattern: what the reference's training loop
// does, train_pytorch.py:91-94) -- one more 16-byte load per group for the row below and the
// right neighbour of a group's last pixel from the next lane.
template <bool VEC, bool SHADE>
__global__ __launch_bounds__(NT) void code_pattern_flat_fwd_kernel(
    const float* __restrict__ pat0, long sb, long sh, long sw, const float* __restrict__ z, int H, int W,
    fc3d_layer_grid lg, float* __restrict__ raw, uint8_t* __restrict__ layer_u8,
    unsigned long long* __restrict__ tilekey) {
    const int b = blockIdx.y;
    const unsigned HW = (unsigned)H * (unsigned)W;
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const unsigned q0 = wave * FLAT_WAVE_PX;
    const float* pimg = pat0 + (long)b * sb;
    const float* zimg = z + (long)b * HW;
    float* rimg = raw + (long)b * HW;
    uint8_t* limg = layer_u8 ? layer_u8 + (long)b * HW : nullptr;
    const LayerFast lf = layer_fast_setup(lg);
    unsigned long long key = 0;
    if (VEC) {
        float4 pv[FLAT_GROUPS], zv[FLAT_GROUPS], zb[FLAT_GROUPS];
        float zn[FLAT_GROUPS];
#pragma unroll
        for (int g = 0; g < FLAT_GROUPS; ++g) {
            const unsigned q = min(q0 + g * 256u + lane * 4u, HW - 4u);    // clamped: always a valid group
            pv[g] = ld_at(reinterpret_cast<const float4*>(pimg), q >> 2);
            zv[g] = ld_at(reinterpret_cast<const float4*>(zimg), q >> 2);
            if (SHADE) {
                zb[g] = ld_at(reinterpret_cast<const float4*>(zimg), min(q + (unsigned)W, HW - 4u) >> 2);
                zn[g] = ld_at(zimg, min(q + 4u, HW - 1u));    // only lane 63 uses it; the others take the next lane's
            }
        }
#pragma unroll
        for (int g = 0; g < FLAT_GROUPS; ++g) {
            const unsigned q = q0 + g * 256u + lane * 4u;
            float znext = 0.0f;
            if (SHADE) {                                       // before the range test: every lane takes part
                znext = __shfl_down(zv[g].x, 1, 64);
                if (lane == 63) znext = zn[g];
            }
            if (q >= HW) continue;
            const unsigned y = q / (unsigned)W, x = q - y * (unsigned)W;   // W % 4 == 0: the group stays in one row
            const float zz[4] = {zv[g].x, zv[g].y, zv[g].z, zv[g].w};
            float pp[4] = {pv[g].x, pv[g].y, pv[g].z, pv[g].w};
            if (SHADE) {
                const bool has_b = y + 1 < (unsigned)H, has_r = x + 4 < (unsigned)W;
                pp[0] = shade(pp[0], zz[0], zz[1], zb[g].x, true, has_b);
                pp[1] = shade(pp[1], zz[1], zz[2], zb[g].y, true, has_b);
                pp[2] = shade(pp[2], zz[2], zz[3], zb[g].z, true, has_b);
                pp[3] = shade(pp[3], zz[3], znext, zb[g].w, has_r, has_b);
            }
            float a[4];
            uint8_t lb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int l = flat_layer(zz[i], lg, lf);
                a[i] = pp[i] * (l >= 0 ? 1.0f : 0.0f);       // a multiply: a non-finite pattern value still poisons the pixel
                lb[i] = (uint8_t)l;
                const uint32_t pos = 0xffffffffu - (uint32_t)((x + i) * (unsigned)H + y);   // first w, then first h
                const unsigned long long k = ((unsigned long long)float_to_key(a[i]) << 32) | pos;
                key = k > key ? k : key;
            }
            st_at(reinterpret_cast<float4*>(rimg), q >> 2, make_float4(a[0], a[1], a[2], a[3]));
            if (limg) st_at(reinterpret_cast<uchar4*>(limg), q >> 2, make_uchar4(lb[0], lb[1], lb[2], lb[3]));
        }
    } else {
        for (int i = 0; i < 4 * FLAT_GROUPS; ++i) {
            const unsigned q = q0 + i * 64u + lane;
            if (q >= HW) break;
            const unsigned y = q / (unsigned)W, x = q - y * (unsigned)W;
            const float z0 = zimg[q];
            const int l = flat_layer(z0, lg, lf);
            float pq = pimg[(long)y * sh + (long)x * sw];
            if (SHADE) {
                const bool has_r = x + 1 < (unsigned)W, has_b = y + 1 < (unsigned)H;
                pq = shade(pq, z0, has_r ? zimg[q + 1] : z0, has_b ? zimg[q + W] : z0, has_r, has_b);
            }
            const float a = pq * (l >= 0 ? 1.0f : 0.0f);
            const uint32_t pos = 0xffffffffu - (uint32_t)(x * (unsigned)H + y);
            const unsigned long long k = ((unsigned long long)float_to_key(a) << 32) | pos;
            key = k > key ? k : key;
            rimg[q] = a;
            if (limg) limg[q] = (uint8_t)l;
        }
    }
    key = wave_max(key);
    if (lane == 0) tilekey[(long)b * (gridDim.x * (NT / 64)) + wave] = key;   // 0 (below every real key) past the image
}

// g_pat0 = g / (max + 1e-10) where the pixel has a layer, 0 elsewhere, and one partial of sum(g*out)
// per wavefront for the fix-up of the maximum's pixel (code_pattern_bwd_max_kernel, 1x1 footprint).
// DOT: sum(g*out) is handed over as ndot partial sums per image (see the rolling-row backward): `out` is
// not read, no partial is written, and the pixel that holds the maximum gets (g - S)/den.
template <bool VEC, bool DOT = false>
__global__ __launch_bounds__(NT) void code_pattern_flat_bwd_kernel(
    const float* __restrict__ g, const float* __restrict__ out, const uint8_t* __restrict__ layer_u8,
    const float* __restrict__ max_in, unsigned HW, float* __restrict__ partial, float* __restrict__ g_pat0,
    const int* __restrict__ argmax_in, const float* __restrict__ dot_partial, int ndot) {
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    const unsigned q0 = wave * FLAT_WAVE_PX;
    const float* gi = g + (long)b * HW;
    const float* oi = out + (long)b * HW;
    const uint8_t* li = layer_u8 + (long)b * HW;
    float* gp = g_pat0 + (long)b * HW;
    const float rden = 1.0f / (max_in[b] + 1e-10f);
    float sgo = 0.0f;
    unsigned am = 0xffffffffu;          // DOT: the maximum's pixel if this wavefront owns it, and S
    float S = 0.0f;
    if (DOT) {
        const unsigned a = (unsigned)argmax_in[b];
        if (a >= q0 && a < q0 + FLAT_WAVE_PX) {      // the same in every lane
            for (int i = lane; i < ndot; i += 64) S += dot_partial[(long)b * ndot + i];
            S = wave_sum(S);
            am = a;
        }
    }
    if (VEC) {
        float4 gv[FLAT_GROUPS], ov[FLAT_GROUPS];
        uchar4 lv[FLAT_GROUPS];
#pragma unroll
        for (int k = 0; k < FLAT_GROUPS; ++k) {
            const unsigned q = min(q0 + k * 256u + lane * 4u, HW - 4u);
            gv[k] = ld_at(reinterpret_cast<const float4*>(gi), q >> 2);
            if (!DOT) ov[k] = ld_at(reinterpret_cast<const float4*>(oi), q >> 2);
            lv[k] = ld_at(reinterpret_cast<const uchar4*>(li), q >> 2);
        }
#pragma unroll
        for (int k = 0; k < FLAT_GROUPS; ++k) {
            const unsigned q = q0 + k * 256u + lane * 4u;
            if (q >= HW) continue;
            if (DOT) {
                gv[k].x -= q == am ? S : 0.0f; gv[k].y -= q + 1 == am ? S : 0.0f;
                gv[k].z -= q + 2 == am ? S : 0.0f; gv[k].w -= q + 3 == am ? S : 0.0f;
            } else {
                sgo = fmaf(gv[k].x, ov[k].x, sgo); sgo = fmaf(gv[k].y, ov[k].y, sgo);
                sgo = fmaf(gv[k].z, ov[k].z, sgo); sgo = fmaf(gv[k].w, ov[k].w, sgo);
            }
            float4 r;
            r.x = lv[k].x != 255 ? gv[k].x * rden : 0.0f; r.y = lv[k].y != 255 ? gv[k].y * rden : 0.0f;
            r.z = lv[k].z != 255 ? gv[k].z * rden : 0.0f; r.w = lv[k].w != 255 ? gv[k].w * rden : 0.0f;
            st_at(reinterpret_cast<float4*>(gp), q >> 2, r);
        }
    } else {
        for (int i = 0; i < 4 * FLAT_GROUPS; ++i) {
            const unsigned q = q0 + i * 64u + lane;
            if (q >= HW) break;
            float gq = gi[q];
            if (DOT) gq -= q == am ? S : 0.0f;
            else sgo = fmaf(gq, oi[q], sgo);
            gp[q] = li[q] != 255 ? gq * rden : 0.0f;
        }
    }
    if (!DOT) {
        sgo = wave_sum(sgo);
        if (lane == 0) partial[(long)b * (gridDim.x * (NT / 64)) + wave] = sgo;
    }
}

// ---------------------------------------------------------------------------------------
// Layer-sweep kernels for large odd square PSFs (9x9 ... 33x33).
//
// At these sizes the work is K*K multiply-adds per pixel and direction, so the aim is to keep the
// vector ALUs on multiply-adds and nothing else.  One wavefront (a 64-thread workgroup) owns a
// 16 x 16 output tile -- 16 columns x 4 row groups of lanes, R = 4 rows per lane -- and stages the
// (16+K-1)^2 input tile in LDS.  It then sweeps over the depth layers present, one at a time, so
// that within a sweep the PSF is the same for every lane: a PSF row is K scalar registers (s_load
// from the PSF stack, constant offsets), never a per-lane operand.  For one input row the lane
// reads its K neighbours from LDS once and reuses them for every output row that row feeds (K
// multiply-adds each), about R*K/(R+K-1) multiply-adds per LDS read.  The cost per pixel is K*K
// times the number of layers met in the tile, not times L; the square tile keeps that number
// (and the halo) small.  The LDS row stride is chosen so that the four row groups fall into
// different banks.
//
// Forward: a sweep of layer l computes the blur with PSF l for every lane and keeps it where
// the output pixel's layer is l (summation order dy, then dx ascending, as in the other forward
// kernels).  Backward: a sweep of layer l blurs the gradient image masked to the *source* pixels
// of layer l (flipped PSF) and the sweeps add up; the masked tile is a second LDS copy rebuilt
// once per sweep.
// ---------------------------------------------------------------------------------------
#ifndef FC3D_SWEEP_COLS
#define FC3D_SWEEP_COLS 16
#endif
#ifndef FC3D_SWEEP_ROWS
#define FC3D_SWEEP_ROWS 4
#endif
#ifndef FC3D_SWEEP_BWD_PSF_ROW_MAJOR
#define FC3D_SWEEP_BWD_PSF_ROW_MAJOR 19   // smallest PSF size whose backward sweep runs PSF-row-major (0: none)
#endif
#ifndef FC3D_SWEEP_BWD_RM_WAVES
#define FC3D_SWEEP_BWD_RM_WAVES 3
#endif
constexpr int kSweepCols = FC3D_SWEEP_COLS;       // output columns per wavefront (64 / kSweepCols row groups)
constexpr int kSweepRows = FC3D_SWEEP_ROWS;       // output rows per lane
constexpr int kSweepTileRows = (64 / kSweepCols) * kSweepRows;
#ifndef FC3D_SWEEP_TWO_PASS
#define FC3D_SWEEP_TWO_PASS 1
#endif
#ifndef FC3D_SWEEP_MANY_LAYERS
#define FC3D_SWEEP_MANY_LAYERS 3
#endif
constexpr int kSweepManyLayers = FC3D_SWEEP_MANY_LAYERS;   // a tile whose layers span at least this many goes first
#define FC3D_FOR_SWEEP_K(X) X(9) X(11) X(13) X(15) X(17) X(19) X(21) X(23) X(25) X(27) X(29) X(31) X(33)

// Packed multiply-adds (v_pk_fma_f32: two per lane and instruction): a PSF row is consumed in pairs
// of neighbouring columns, even and odd partial sums added at the end.  Measured per PSF size at
// 8x512x512: the forward kernel gains from 15x15 on (15 %-25 %; 11x11 and 13x13 lose to the register
// shuffling the pairs need), the backward kernel from 13x13 on (6-10 %).
template <int K, int R, int CW>
struct Sweep {
    static constexpr bool PK_FWD = K >= 15, PK_BWD = K >= 13;
    // forward loop order: PSF-row-major with the input rows rotating through registers from 15x15 on
    // (measured at 8x512x512: 15x15 43 -> 41 us, 23x23 78 -> 68, 33x33 148 -> 124; 9x9 .. 13x13 lose 1-3 us)
    static constexpr bool PSF_ROW_MAJOR_FWD = K >= 15;
    static constexpr int PAD = (K - 1) / 2;
    static constexpr int NG = 64 / CW;            // row groups of lanes
    static constexpr int TR = NG * R;             // output rows per wavefront
    static constexpr int NR = TR + K - 1;         // staged rows
    static constexpr int LW0 = CW + K - 1;        // staged columns
    // row stride: R * LW = 16 or 48 (mod 64) puts the NG groups' 16-lane reads into different banks
    static constexpr int LW = NG == 1 ? LW0 : LW0 + ((4 - LW0 % 8) + 8) % 8;
    static constexpr int LWB = (LW0 + 3) & ~3;    // bytes per row of the staged layer tile
    // wavefronts per SIMD the backward kernel is compiled for: what its LDS tile lets a compute unit hold
    static constexpr bool PSF_ROW_MAJOR_BWD = FC3D_SWEEP_BWD_PSF_ROW_MAJOR != 0 && K >= FC3D_SWEEP_BWD_PSF_ROW_MAJOR;
    static constexpr int BWD_WAVES = PSF_ROW_MAJOR_BWD ? FC3D_SWEEP_BWD_RM_WAVES : K <= 25 ? 4 : 3;
    static_assert(NG == 1 || R == 4, "the bank-spreading stride assumes four rows per lane");
};

// Stages NR x LW0 cells of a plane (top-left cell at image position (x_first, y_first)) through
// put(row, column, value, inside_image).  Loads are unconditional from clamped addresses, eight
// per lane in flight.
template <int NR, int LW0, typename T, typename F>
__device__ __forceinline__ void sweep_stage(const T* __restrict__ plane, unsigned row_stride, unsigned col_stride,
                                            int H, int W, int x_first, int y_first, int lane, F put) {
    constexpr int N = NR * LW0;
    for (int i0 = 0; i0 < N; i0 += 64 * 8) {
        T v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + u * 64 + lane, N - 1);
            const int r = i / LW0, c = i - r * LW0;
            v[u] = ld_at(plane, (unsigned)min(max(y_first + r, 0), H - 1) * row_stride +
                                    (unsigned)min(max(x_first + c, 0), W - 1) * col_stride);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + u * 64 + lane;
            if (i < N) {
                const int r = i / LW0, c = i - r * LW0;
                const int gy = y_first + r, gx = x_first + c;
                put(r, c, v[u], gy >= 0 && gy < H && gx >= 0 && gx < W);
            }
        }
    }
}

// The same for a window that lies inside the image: lane c < LW0 loads column c of every row -- the
// address is a scalar row base plus one fixed lane offset, the LDS address a constant offset from one
// lane base, so a row costs a load and a store and no vector arithmetic (the clamped form above spends
// about 25 vector instructions per cell on index splitting, clamping and bounds tests).
template <int NR, int LW0, int LW, typename T>
__device__ __forceinline__ void sweep_stage_inside(const T* __restrict__ first, unsigned row_stride, unsigned col_stride,
                                                   int lane, float* __restrict__ tile) {
    const bool on = lane < LW0;
    const unsigned voff = (unsigned)min(lane, LW0 - 1) * col_stride;
    T v[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) v[r] = ld_at(first + (size_t)r * row_stride, voff);
    if (on) {
#pragma unroll
        for (int r = 0; r < NR; ++r) tile[r * LW + lane] = (float)v[r];
    }
}

// One sweep of one layer: acc[o] += sum over PSF rows dy in [dy_lo, dy_hi] and columns e of
// w(dy, e) * tile[dy + o][e], for the R output rows o of a lane.  `mine` points at the lane's window
// in the LDS tile (row stride LW), `wl` at the K x K PSF of the layer; FLIP reads the PSF mirrored in
// both directions (the backward form).  PSF-row-major: an iteration loads ONE PSF row (K scalar
// registers) and ONE new input row (K vector registers) and uses them with the R-1 input rows of
// the previous iterations, which stay in registers -- unrolled by R so that the rotation is a
// renaming, not a copy.  (Input-row-major order needs all R PSF rows of an input row at once:
// 4 K scalar loads per row, no room to keep them across rows in 100 scalar registers, and one
// full wait for LDS and scalar loads together per row.)  Per accumulator the order of additions is
// dy ascending, then e ascending (even and odd e apart when packed), as in the other kernels.
template <int K, int R, int LW, bool PK, bool FLIP>
__device__ __forceinline__ void sweep_core(const float* mine, const float* __restrict__ wl, int dy_lo, int dy_hi,
                                           float (&acc)[R], f2 (&acc2)[R]) {
    static_assert(R == 4, "the rotation is unrolled for four rows per lane");
    constexpr int NRG = R + K - 1;
    float X[R][K];
    float Wt[1][K];
    auto load_row = [&](float (&x)[K], int r) {
#pragma unroll
        for (int e = 0; e < K; ++e) x[e] = mine[r * LW + e];
    };
    auto load_psf = [&](float (&w)[K], int dy) {
        const float* wr = wl + (FLIP ? K - 1 - dy : dy) * K;
#pragma unroll
        for (int e = 0; e < K; ++e) w[e] = FLIP ? wr[K - 1 - e] : wr[e];
    };
    // two output rows at a time, their accumulator chains interleaved (a packed multiply-add that
    // reads the result of the one before it costs a wait state)
    auto fma_pair = [&](int oa, const float (&w)[K], const float (&xa)[K], const float (&xb)[K]) {
        if (PK) {
#pragma unroll
            for (int e = 0; e + 1 < K; e += 2) {
                acc2[oa] = __builtin_elementwise_fma(f2{w[e], w[e + 1]}, f2{xa[e], xa[e + 1]}, acc2[oa]);
                acc2[oa + 1] = __builtin_elementwise_fma(f2{w[e], w[e + 1]}, f2{xb[e], xb[e + 1]}, acc2[oa + 1]);
            }
            acc2[oa].x = fmaf(w[K - 1], xa[K - 1], acc2[oa].x);          // K is odd
            acc2[oa + 1].x = fmaf(w[K - 1], xb[K - 1], acc2[oa + 1].x);
        } else {
#pragma unroll
            for (int e = 0; e < K; ++e) {
                acc[oa] = fmaf(w[e], xa[e], acc[oa]);
                acc[oa + 1] = fmaf(w[e], xb[e], acc[oa + 1]);
            }
        }
    };
    // x0..x2 hold input rows dy..dy+2; x3 is free and receives row dy+3
    auto step = [&](int dy, float (&x0)[K], float (&x1)[K], float (&x2)[K], float (&x3)[K]) {
        load_row(x3, dy + R - 1);
        load_psf(Wt[0], dy);
        fma_pair(0, Wt[0], x0, x1);
        fma_pair(2, Wt[0], x2, x3);
    };
    const int d0 = dy_lo & ~(R - 1);
    load_row(X[0], d0);
    load_row(X[1], d0 + 1);
    load_row(X[2], d0 + 2);
    for (int d4 = d0; d4 <= dy_hi; d4 += R) {
        if (d4 + R <= K) {
            step(d4, X[0], X[1], X[2], X[3]);
            step(d4 + 1, X[1], X[2], X[3], X[0]);
            step(d4 + 2, X[2], X[3], X[0], X[1]);
            step(d4 + 3, X[3], X[0], X[1], X[2]);
        } else {                                   // the last K % 4 PSF rows (K is odd: one or three)
            step(d4, X[0], X[1], X[2], X[3]);
            if (K % R == 3) {
                step(d4 + 1, X[1], X[2], X[3], X[0]);
                step(d4 + 2, X[2], X[3], X[0], X[1]);
            }
        }
    }
}

// The same sweep with whole blocks of R PSF rows left out: `live` has bit r set when window row r (of
// any row group) can hold a non-zero cell; a block of PSF rows d4 .. d4+R-1 reads rows d4 .. d4+2R-2.
// Straight-line code per block behind a uniform branch; a block that follows a skipped one reloads
// the R-1 rows the rotation would have left in registers.
template <int K, int R, int LW, bool PK, bool FLIP>
__device__ __forceinline__ void sweep_core_blocks(const float* mine, const float* __restrict__ wl,
                                                  unsigned long long live, float (&acc)[R], f2 (&acc2)[R]) {
    static_assert(R == 4, "the rotation is unrolled for four rows per lane");
    float X[R][K];
    float Wt[K];
    auto load_row = [&](float (&x)[K], int r) {
#pragma unroll
        for (int e = 0; e < K; ++e) x[e] = mine[r * LW + e];
    };
    auto load_psf = [&](int dy) {
        const float* wr = wl + (FLIP ? K - 1 - dy : dy) * K;
#pragma unroll
        for (int e = 0; e < K; ++e) Wt[e] = FLIP ? wr[K - 1 - e] : wr[e];
    };
    auto fma_pair = [&](int oa, const float (&xa)[K], const float (&xb)[K]) {
        if (PK) {
#pragma unroll
            for (int e = 0; e + 1 < K; e += 2) {
                acc2[oa] = __builtin_elementwise_fma(f2{Wt[e], Wt[e + 1]}, f2{xa[e], xa[e + 1]}, acc2[oa]);
                acc2[oa + 1] = __builtin_elementwise_fma(f2{Wt[e], Wt[e + 1]}, f2{xb[e], xb[e + 1]}, acc2[oa + 1]);
            }
            acc2[oa].x = fmaf(Wt[K - 1], xa[K - 1], acc2[oa].x);
            acc2[oa + 1].x = fmaf(Wt[K - 1], xb[K - 1], acc2[oa + 1].x);
        } else {
#pragma unroll
            for (int e = 0; e < K; ++e) {
                acc[oa] = fmaf(Wt[e], xa[e], acc[oa]);
                acc[oa + 1] = fmaf(Wt[e], xb[e], acc[oa + 1]);
            }
        }
    };
    auto step = [&](int dy, float (&x0)[K], float (&x1)[K], float (&x2)[K], float (&x3)[K]) {
        load_row(x3, dy + R - 1);
        load_psf(dy);
        fma_pair(0, x0, x1);
        fma_pair(2, x2, x3);
    };
    bool have = false;                             // X[0..2] hold rows d4 .. d4+2
#pragma unroll
    for (int d4 = 0; d4 < K; d4 += R) {
        constexpr unsigned long long kSpan = (1ull << (2 * R - 1)) - 1ull;
        if (((live >> d4) & kSpan) == 0ull) {
            have = false;
            continue;
        }
        if (!have) {
            load_row(X[0], d4);
            load_row(X[1], d4 + 1);
            load_row(X[2], d4 + 2);
        }
        have = true;
        if (d4 + R <= K) {
            step(d4, X[0], X[1], X[2], X[3]);
            step(d4 + 1, X[1], X[2], X[3], X[0]);
            step(d4 + 2, X[2], X[3], X[0], X[1]);
            step(d4 + 3, X[3], X[0], X[1], X[2]);
        } else {
            step(d4, X[0], X[1], X[2], X[3]);
            if (K % R == 3) {
                step(d4 + 1, X[1], X[2], X[3], X[0]);
                step(d4 + 2, X[2], X[3], X[0], X[1]);
            }
        }
    }
}

template <int K, int R, int CW, bool INTC0 = false>
__global__ __launch_bounds__(64) void code_pattern_fwd_sweep_kernel(
    const float* __restrict__ pat0, long sb, long sh, long sw, const float* __restrict__ z,
    const float* __restrict__ psf, int H, int W, fc3d_layer_grid lg, float* __restrict__ raw,
    uint8_t* __restrict__ layer_u8, unsigned long long* __restrict__ tilekey, int two_pass) {
    using S = Sweep<K, R, CW>;
    constexpr bool PK = S::PK_FWD;
    constexpr int PAD = S::PAD, LW = S::LW, NRG = R + K - 1;       // NRG: input rows one row group reads
    extern __shared__ float smem[];               // [NR][LW]
    const int lane = threadIdx.x, col = lane % CW, grp = lane / CW;
    // Two passes over the tiles in one grid (the upper half of gridDim.z is the second pass): the first
    // takes the tiles whose pixels span several depth layers, the second all others.  A tile costs one
    // sweep per layer, up to ten times the average along depth edges; dispatched in image order the
    // last of those decide when the launch ends, started first they run beside everything else.  A
    // workgroup that meets a tile of the other pass leaves after the layer decision.
    const int nb = two_pass ? gridDim.z / 2 : gridDim.z;
    const int pass = blockIdx.z / nb, b = blockIdx.z - pass * nb;
    const int x0 = blockIdx.x * CW, y0 = blockIdx.y * S::TR;
    const long img = (long)b * H * W;
    const int x = x0 + col, yg = y0 + grp * R;

    auto stage = [&]() {
        const bool inside = x0 - PAD >= 0 && x0 - PAD + S::LW0 <= W && y0 - PAD >= 0 && y0 - PAD + S::NR <= H;   // (uniform)
        if (inside)
            sweep_stage_inside<S::NR, S::LW0, LW>(pat0 + (long)b * sb + (long)(y0 - PAD) * sh + (long)(x0 - PAD) * sw,
                                                  (unsigned)sh, (unsigned)sw, lane, smem);
        else
            sweep_stage<S::NR, S::LW0>(pat0 + (long)b * sb, (unsigned)sh, (unsigned)sw, H, W, x0 - PAD, y0 - PAD, lane,
                                       [&](int r, int c, float v, bool ok) { smem[r * LW + c] = ok ? v : 0.0f; });
    };
    int lay[R];
    // the layers of the lane's pixels (reciprocal shortcut; the exact chain for the whole tile when any pixel
    // is flagged); returns whether the tile belongs to the first pass
    auto decide = [&]() -> bool {
        const LayerFast lfast = layer_fast_setup(lg);
        float zv[R];
        bool amb = false;
#pragma unroll
        for (int o = 0; o < R; ++o) {
            const int y = yg + o;
            zv[o] = ld_at(z + img, (unsigned)(min(y, H - 1) * W + min(x, W - 1)));
        }
#pragma unroll
        for (int o = 0; o < R; ++o) {
            bool a;
            lay[o] = layer_fast<INTC0>(zv[o], lg, lfast, a);
            amb |= a;
        }
        if (__ballot(amb) != 0ull) {
#pragma unroll
            for (int o = 0; o < R; ++o) lay[o] = layer_of(zv[o], lg);
        }
        int lo = 0x7fffffff, hi = -1;
#pragma unroll
        for (int o = 0; o < R; ++o) {
            lay[o] = (yg + o < H && x < W) ? lay[o] : -1;
            if (lay[o] >= 0) {
                lo = min(lo, lay[o]);
                hi = max(hi, lay[o]);
            }
        }
        return wave_max(hi) - wave_min(lo) >= kSweepManyLayers - 1;
    };
    if (!two_pass) {
        stage();
        decide();
    } else if (pass == 0) {
        if (!decide()) return;
        stage();
    } else {
        stage();
        if (decide()) return;
    }
    __syncthreads();

    const float* mine = smem + grp * R * LW + col;               // this lane's window: rows grp*R .., columns col ..
    float res[R];
#pragma unroll
    for (int o = 0; o < R; ++o) res[o] = 0.0f;
    int last = -1;
    for (;;) {
        int cand = 0x7fffffff;
#pragma unroll
        for (int o = 0; o < R; ++o)
            if (lay[o] > last) cand = min(cand, lay[o]);
        cand = wave_min(cand);
        if (cand == 0x7fffffff) break;
        last = cand;
        const int l = __builtin_amdgcn_readfirstlane(cand);     // the layer of this sweep, in a scalar register
        float acc[R];                                            // one of the two sets is used, per PK
        f2 acc2[R];
#pragma unroll
        for (int o = 0; o < R; ++o) {
            acc[o] = 0.0f;
            acc2[o] = f2{0.0f, 0.0f};
        }
        if constexpr (S::PSF_ROW_MAJOR_FWD) {
            sweep_core<K, R, LW, PK, false>(mine, psf + (long)l * K * K, 0, K - 1, acc, acc2);
        } else {
            unsigned rows = 0;                                       // row slots in which some lane has a pixel of layer l
#pragma unroll
            for (int o = 0; o < R; ++o)
                if (__ballot(lay[o] == l) != 0ull) rows |= 1u << o;
            // Input rows R-1 .. K-1 feed all R output rows: that part is straight-line code (R scalar
            // loads of a PSF row, R*K multiply-adds) so that the loads run ahead of the arithmetic; the
            // first and last R-1 input rows feed fewer output rows and take the branchy form.
            auto row_step = [&](int r, auto full) {
                constexpr bool FULL = decltype(full)::value;
                if (!FULL) {
                    const int lo = max(r - (K - 1), 0), hi = min(r, R - 1);   // output rows that input row r feeds
                    if ((rows & ((2u << hi) - 1u) & ~((1u << lo) - 1u)) == 0u) return;
                }
                float in[K];
#pragma unroll
                for (int dx = 0; dx < K; ++dx) in[dx] = mine[r * LW + dx];
                // PSF row dy = r - o of layer l sits (R-1-o)*K floats after wbase: constant offsets for the scalar loads
                const float* wbase = psf + ((long)l * K + (r - (R - 1))) * K;
#pragma unroll
                for (int o = 0; o < R; ++o) {
                    const int dy = r - o;
                    if (FULL || (((rows >> o) & 1u) && dy >= 0 && dy < K)) {
                        const float* wr = wbase + (R - 1 - o) * K;
                        if (PK) {
#pragma unroll
                            for (int dx = 0; dx + 1 < K; dx += 2)
                                acc2[o] = __builtin_elementwise_fma(f2{wr[dx], wr[dx + 1]}, f2{in[dx], in[dx + 1]}, acc2[o]);
                            acc2[o].x = fmaf(wr[K - 1], in[K - 1], acc2[o].x);  // K is odd
                        } else {
#pragma unroll
                            for (int dx = 0; dx < K; ++dx) acc[o] = fmaf(wr[dx], in[dx], acc[o]);
                        }
                    }
                }
            };
            for (int r = 0; r < R - 1; ++r) row_step(r, std::false_type{});
            for (int r = R - 1; r < K; ++r) row_step(r, std::true_type{});
            for (int r = K; r < NRG; ++r) row_step(r, std::false_type{});
        }
#pragma unroll
        for (int o = 0; o < R; ++o)
            if (lay[o] == l) res[o] = PK ? acc2[o].x + acc2[o].y : acc[o];
    }

    unsigned long long key = 0;
#pragma unroll
    for (int o = 0; o < R; ++o) {
        const int y = yg + o;
        const float a = res[o] * (lay[o] >= 0 ? 1.0f : 0.0f);    // a multiply: a non-finite blur still poisons the pixel
        if (y < H && x < W) {
            const uint32_t pos = 0xffffffffu - (uint32_t)(x * H + y);   // first w, then first h
            const unsigned long long k = ((unsigned long long)float_to_key(a) << 32) | pos;
            key = k > key ? k : key;
            const unsigned q = (unsigned)(y * W + x);
            st_at(raw + img, q, a);
            if (layer_u8) st_at(layer_u8 + img, q, (uint8_t)lay[o]);
        }
    }
    key = wave_max(key);
    if (lane == 0) tilekey[((long)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = key;
}

// Whether a launch runs the tiles with many layers first.  The second look at every tile costs a wavefront's
// start and one memory latency per tile; what it buys is the tail of the launch, which is a smaller share
// the more rounds of tiles the chip works through (256 compute units x 16 wavefronts hold 4,096 tiles at a
// time) and the cheaper a sweep is.  Measured on 23x23 ... 9x9 PSFs: 4 x 1024 x 1024 (16 k tiles, 41 layers)
// gains 10-25 % from 13x13 on and nothing at 9x9; 32 x 512 x 512 (33 k tiles, 21 layers) loses 10-20 % at
// every size.
static int sweep_two_pass(int K, int B, int H, int W) {
    const long tiles = (long)cdiv(W, kSweepCols) * cdiv(H, kSweepTileRows) * B;
    if (FC3D_SWEEP_TWO_PASS == 2) return 2 * B <= 65535;      // (experiments: always)
    return FC3D_SWEEP_TWO_PASS && 2 * B <= 65535 && K >= 11 && tiles <= 6L * 4096;
}

// LDS bytes of the backward sweep kernel: the tile masked to the current layer and the layer tile.
template <int K, int R, int CW>
static size_t sweep_bwd_lds() {
    using S = Sweep<K, R, CW>;
    return (size_t)((S::NR * S::LW + 63) / 64) * 64 * 5;     // tile and layer tile in whole 64-cell slices
}

// The staged cells (gradient / (max + eps) and layer id) stay in registers for the whole kernel, CPL
// per lane; a sweep writes them to the one LDS tile masked to its layer.  A second, unmasked LDS copy
// would halve the wavefronts a compute unit can hold, and that residency -- not the arithmetic --
// is what bounds this kernel.  Which staged rows hold a pixel of which layer is kept as one 64-bit
// set of layers per row (lane r: row r), so that the per-sweep row mask is a single ballot and the
// skip test of an input row is scalar arithmetic.
// DOT: sum(g*out) arrives as partial sums (see the rolling-row backward): `out` is not read, no partial
// is written, and the staged cell of the pixel that holds the maximum is lowered by S/den, so the maximum's
// share is blurred along with everything else.
template <int K, int R, int CW, bool DOT = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(Sweep<K, R, CW>::BWD_WAVES, Sweep<K, R, CW>::BWD_WAVES)))
void code_pattern_bwd_sweep_kernel(
    const float* __restrict__ g, const float* __restrict__ out, const uint8_t* __restrict__ layer_u8,
    const float* __restrict__ max_in, const float* __restrict__ psf, int H, int W, int L,
    float* __restrict__ partial, float* __restrict__ g_pat0, const int* __restrict__ argmax_in,
    const float* __restrict__ dot_partial, int ndot, int two_pass) {
    using S = Sweep<K, R, CW>;
    constexpr int PAD = S::PAD, LW = S::LW, LW0 = S::LW0, LWB = S::LW, NR = S::NR, NG = S::NG, NRG = R + K - 1;
    constexpr bool PK = S::PK_BWD;
    constexpr int HB = K - 1 - PAD;               // halo rows / columns before the tile
    constexpr int N = NR * LW, CPL = (N + 63) / 64;               // cells of the LDS tile (padding included), per lane
    static_assert(NR <= 64, "one lane per staged row");
    extern __shared__ float smem[];
    float* mt = smem;                                              // [NR][LW] the tile masked to the current layer
    uint8_t* lt = reinterpret_cast<uint8_t*>(mt + CPL * 64);       // [NR][LW] layer of every staged pixel (255: none), cell for cell like mt
    const int lane = threadIdx.x, col = lane % CW, grp = lane / CW;
    // two passes over the tiles in one grid, the tiles with many layers first (see the forward kernel)
    const int nb = two_pass ? gridDim.z / 2 : gridDim.z;
    const int pass = blockIdx.z / nb, b = blockIdx.z - pass * nb;
    const int x0 = blockIdx.x * CW, y0 = blockIdx.y * S::TR;
    const long img = (long)b * H * W;
    const int x = x0 + col, yg = y0 + grp * R;

    // Which pass the tile belongs to, from a TR x CW lattice of samples over the staged window (the same
    // samples in both passes, or a tile could be taken twice or never): loads here, the decision below.
    uint8_t smp[R];
    if (two_pass) {
#pragma unroll
        for (int o = 0; o < R; ++o) {
            const int r = ((grp * R + o) * (NR - 1)) / (S::TR - 1), c = (col * (LW0 - 1)) / (CW - 1);
            smp[o] = ld_at(layer_u8 + img, (unsigned)min(max(y0 - HB + r, 0), H - 1) * (unsigned)W +
                                               (unsigned)min(max(x0 - HB + c, 0), W - 1));
        }
    }
    auto many_layers = [&]() -> bool {
        int lo = 255, hi = -1;
#pragma unroll
        for (int o = 0; o < R; ++o)
            if (smp[o] != 255) {
                lo = min(lo, (int)smp[o]);
                hi = max(hi, (int)smp[o]);
            }
        return wave_max(hi) - wave_min(lo) >= kSweepManyLayers - 1;
    };
    if (two_pass && pass == 0 && !many_layers()) return;
    const float rden = 1.0f / (max_in[b] + 1e-10f);

    // Stage.  Lane `lane` keeps tile cells lane, lane + 64, ... (so a sweep's stores need no address
    // arithmetic); unconditional loads from clamped addresses, all of a lane's cells in flight.
    float gv[CPL];
    uint32_t lv[(CPL + 3) / 4];
#pragma unroll
    for (int v = 0; v < (CPL + 3) / 4; ++v) lv[v] = 0u;
    constexpr int SB = 16;                                         // cells of a lane loaded per batch
    // A window that lies inside the image with its padding columns and the rows that the cells past the
    // tile (the last lanes' share of the final 64-cell slice) fall into needs no clamping and no bounds
    // tests: cell i is element base + i + (i / LW) * (W - LW) of the plane.
    constexpr int RLAST = (CPL * 64 - 1) / LW;                     // last window row any lane's cell index reaches
    const bool inside = x0 - HB >= 0 && x0 - HB + LW <= W && y0 - HB >= 0 && y0 - HB + RLAST + 1 <= H;   // (uniform)
    if (inside) {
        const unsigned base = (unsigned)(y0 - HB) * (unsigned)W + (unsigned)(x0 - HB);
#pragma unroll
        for (int u0 = 0; u0 < CPL; u0 += SB) {
            uint8_t lb[SB];
#pragma unroll
            for (int u = u0; u < min(u0 + SB, CPL); ++u) {
                const unsigned i = (unsigned)(u * 64 + lane);
                const unsigned q = base + i + (i / (unsigned)LW) * (unsigned)(W - LW);
                gv[u] = ld_at(g + img, q);
                lb[u - u0] = ld_at(layer_u8 + img, q);
            }
#pragma unroll
            for (int u = u0; u < min(u0 + SB, CPL); ++u) {
                const unsigned i = (unsigned)(u * 64 + lane);
                const bool cell = i % (unsigned)LW < (unsigned)LW0 && (u * 64 + 63 < N || i < (unsigned)N);
                const uint32_t l = cell ? (uint32_t)lb[u - u0] : 255u;
                gv[u] = cell ? gv[u] * rden : 0.0f;
                lv[u / 4] |= l << (8 * (u % 4));
                lt[u * 64 + lane] = (uint8_t)l;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    } else
#pragma unroll
    for (int u0 = 0; u0 < CPL; u0 += SB) {
        uint8_t lb[SB];
#pragma unroll
        for (int u = u0; u < min(u0 + SB, CPL); ++u) {
            const int i = u * 64 + lane;
            const int r = min(i / LW, NR - 1), c = min(i % LW, LW0 - 1);
            const unsigned q = (unsigned)min(max(y0 - HB + r, 0), H - 1) * (unsigned)W + (unsigned)min(max(x0 - HB + c, 0), W - 1);
            gv[u] = ld_at(g + img, q);
            lb[u - u0] = ld_at(layer_u8 + img, q);
        }
#pragma unroll
        for (int u = u0; u < min(u0 + SB, CPL); ++u) {
            const int i = u * 64 + lane;
            const int r = i / LW, c = i % LW;
            const int gy = y0 - HB + r, gx = x0 - HB + c;
            const bool cell = r < NR && c < LW0;
            const bool ok = cell && gy >= 0 && gy < H && gx >= 0 && gx < W;
            const uint32_t l = ok ? (uint32_t)lb[u - u0] : 255u;
            gv[u] = ok ? gv[u] * rden : 0.0f;
            lv[u / 4] |= l << (8 * (u % 4));
            if (cell) lt[r * LWB + c] = (uint8_t)l;
        }
        __builtin_amdgcn_sched_barrier(0);                         // keep the batches apart: registers, not latency, are short here
    }
    if (two_pass && pass == 1 && many_layers()) return;
    if (DOT) {
        // the maximum's pixel, if this tile stages it (the same test in every lane): cell (pr, pc) of the tile
        const int am = argmax_in[b];
        const int pr = am / W - (y0 - HB), pc = am % W - (x0 - HB);
        if (pr >= 0 && pr < NR && pc >= 0 && pc < LW0) {
            float sp = 0.0f;
            for (int i = lane; i < ndot; i += 64) sp += dot_partial[(long)b * ndot + i];
            sp = wave_sum(sp);
            const int cellidx = pr * LW + pc;
#pragma unroll
            for (int u = 0; u < CPL; ++u)
                if (u == cellidx / 64 && lane == cellidx % 64) gv[u] -= sp * rden;   // (a pixel without a layer is in no sweep, so nothing comes of it there)
        }
    } else {
        // sum of g*out over the pixels this wavefront owns (for the gradient through the per-image maximum)
        float sgo = 0.0f;
#pragma unroll
        for (int o = 0; o < R; ++o) {
            const int y = yg + o;
            const unsigned q = (unsigned)(min(y, H - 1) * W + min(x, W - 1));
            const float gq = ld_at(g + img, q), ov = ld_at(out + img, q);
            sgo = fmaf((y < H && x < W) ? gq : 0.0f, ov, sgo);
        }
        sgo = wave_sum(sgo);
        if (lane == 0) partial[((long)b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x] = sgo;
    }
    __syncthreads();

    float acc[R];                                                 // one of the two sets is used, per PK
    f2 acc2[R];
#pragma unroll
    for (int o = 0; o < R; ++o) {
        acc[o] = 0.0f;
        acc2[o] = f2{0.0f, 0.0f};
    }
    constexpr unsigned long long kGroupRows = NG == 4 ? (1ull | 1ull << R | 1ull << 2 * R | 1ull << 3 * R)
                                            : NG == 2 ? (1ull | 1ull << R) : 1ull;
    static_assert(NG == 1 || NG == 2 || NG == 4, "row groups per wavefront");
    const float* mine = mt + grp * R * LW + col;                  // this lane's window: rows grp*R .., columns col ..
    for (int base = 0; base < L; base += 64) {                    // layers in sets of 64 (one set unless L > 64)
        unsigned long long rowset = 0ull;                         // lane r: layers base .. base+63 present in staged row r
        if (lane < NR) {
#pragma unroll 8
            for (int c = 0; c < LW0; ++c) {
                const unsigned d = (unsigned)lt[lane * LWB + c] - (unsigned)base;
                rowset |= d < 64u ? 1ull << d : 0ull;
            }
        }
        unsigned long long todo = rowset;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) todo |= __shfl_xor(todo, o, 64);
        todo = ((unsigned long long)__builtin_amdgcn_readfirstlane((int)(todo >> 32)) << 32) |
               (unsigned)__builtin_amdgcn_readfirstlane((int)todo);
        while (todo != 0ull) {
            const int d = __builtin_ctzll(todo);
            todo &= todo - 1ull;
            const int l = base + d;
            const unsigned long long rows = __ballot((rowset >> d) & 1ull);   // staged rows with a pixel of layer l
            __syncthreads();                      // the previous sweep's reads of the tile are done
#pragma unroll
            for (int u = 0; u < CPL; ++u) {
                const bool on = ((lv[u / 4] >> (8 * (u % 4))) & 255u) == (unsigned)l;
                mt[u * 64 + lane] = on ? gv[u] : 0.0f;
            }
            __syncthreads();
            // Row r of a group's window, column col+e is source (yg-HB+r, x-HB+e); it reaches output row o
            // through PSF row K-1-(r-o) and PSF column K-1-e.  Rows R-1 .. K-1 feed all R output rows
            // (straight-line code, see the forward kernel); the first and last R-1 rows take the branchy form.
#if FC3D_SWEEP_BWD_PSF_ROW_MAJOR
            if constexpr (K >= FC3D_SWEEP_BWD_PSF_ROW_MAJOR) {
                // window rows (of any row group) that hold a pixel of this layer, as one range
                unsigned long long any = 0ull;
#pragma unroll
                for (int gq = 0; gq < NG; ++gq) any |= rows >> (gq * R);
                any &= (1ull << NRG) - 1ull;
                sweep_core_blocks<K, R, LW, PK, true>(mine, psf + (long)l * K * K, any, acc, acc2);
                continue;
            }
#endif
            auto row_step = [&](int r, auto full) {
                constexpr bool FULL = decltype(full)::value;
                if (((rows >> r) & kGroupRows) == 0ull) return;   // no group has a pixel of this layer in its row r
                float in[K];
#pragma unroll
                for (int e = 0; e < K; ++e) in[e] = mine[r * LW + e];
                // PSF row K-1-(r-o) of layer l sits o*K floats after wbase: constant offsets for the scalar loads
                const float* wbase = psf + ((long)l * K + (K - 1 - r)) * K;
#pragma unroll
                for (int o = 0; o < R; ++o) {
                    const int dd = r - o;
                    if (FULL || (dd >= 0 && dd < K)) {
                        const float* wr = wbase + o * K;
                        if (PK) {
#pragma unroll
                            for (int e = 0; e + 1 < K; e += 2)
                                acc2[o] = __builtin_elementwise_fma(f2{wr[K - 1 - e], wr[K - 2 - e]}, f2{in[e], in[e + 1]}, acc2[o]);
                            acc2[o].x = fmaf(wr[0], in[K - 1], acc2[o].x);    // K is odd
                        } else {
#pragma unroll
                            for (int e = 0; e < K; ++e) acc[o] = fmaf(wr[K - 1 - e], in[e], acc[o]);
                        }
                    }
                }
            };
            for (int r = 0; r < R - 1; ++r) row_step(r, std::false_type{});
            for (int r = R - 1; r < K; ++r) row_step(r, std::true_type{});
            for (int r = K; r < NRG; ++r) row_step(r, std::false_type{});
        }
    }
#pragma unroll
    for (int o = 0; o < R; ++o) {
        const int y = yg + o;
        if (y < H && x < W) st_at(g_pat0 + img, (unsigned)(y * W + x), PK ? acc2[o].x + acc2[o].y : acc[o]);
    }
}

#ifndef FC3D_ROLL_ROWS
#define FC3D_ROLL_ROWS 8
#endif
#ifndef FC3D_ROLL_ROWS_BWD
#define FC3D_ROLL_ROWS_BWD 16
#endif
constexpr int kRollRows = FC3D_ROLL_ROWS;         // output rows per strip of the rolling-row kernels: forward,
constexpr int kRollRowsBwd = FC3D_ROLL_ROWS_BWD;  // backward
#ifndef FC3D_ROLL_ROWS_BWD_DOT
#define FC3D_ROLL_ROWS_BWD_DOT 16
#endif
constexpr int kRollRowsBwdDot = FC3D_ROLL_ROWS_BWD_DOT;   // backward that is handed sum(g*out): no rows of `out` in registers

constexpr int kBigLdsBudget = 96 * 1024;      // dynamic LDS a kernel may be allowed beyond the 48 KiB default
template <typename K>
static int allow_big_lds(K kernel, size_t bytes) {
    if (bytes <= 48 * 1024) return FC3D_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBigLdsBudget);
    return e == hipSuccess ? FC3D_OK : (int)e;
}

static int layer_capacity(int Kh, int Kw, int L, bool with_layer_tile, size_t* lds_bytes) {
    const TileGeom tg = tile_geom(Kh, Kw);
    const size_t tile_b = (size_t)tg.th * tg.S * (with_layer_tile ? 5 : 4);
    const size_t per_layer = (size_t)Kh * Kw * 4;
    if (tile_b + per_layer + 16 > (size_t)kLdsBudget) return -1;
    long cap = (long)((kLdsBudget - tile_b - 16) / per_layer);
    if (cap > L) cap = L;
    *lds_bytes = tile_b + (size_t)cap * per_layer + 16;
    return (int)cap;
}

}  // namespace fc3d

using namespace fc3d;

extern "C" size_t fc3d_code_pattern_fwd_workspace(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const long a = (long)cdiv(W, TX) * cdiv(H, TY);
    const long r = (long)cdiv(cdiv(W, Roll<7>::SW), NT / 64) * (NT / 64) * cdiv(H, kRollRows);
    const long sw = (long)cdiv(W, kSweepCols) * cdiv(H, kSweepTileRows);
    const long fl = (long)cdiv(cdiv((long)H * W, FLAT_WAVE_PX), NT / 64) * (NT / 64);
    long m = a > r ? a : r;
    m = sw > m ? sw : m;
    m = fl > m ? fl : m;
    return (size_t)B * m * sizeof(unsigned long long);
}

extern "C" size_t fc3d_code_pattern_bwd_workspace(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const long a = cdiv((long)H * W, RED_PX);
    const long r = (long)cdiv(cdiv(W, Roll<7>::SW), NT / 64) * (NT / 64) * cdiv(H, kRollRowsBwd);
    const long sw = (long)cdiv(W, kSweepCols) * cdiv(H, kSweepTileRows);
    const long fl = (long)cdiv(cdiv((long)H * W, FLAT_WAVE_PX), NT / 64) * (NT / 64);
    long m = a > r ? a : r;
    m = sw > m ? sw : m;
    m = fl > m ? fl : m;
    return (size_t)B * m * sizeof(float);
}

extern "C" int fc3d_code_pattern_fwd(void* stream, const float* pat0, int64_t pat_sb, int64_t pat_sh,
                                     int64_t pat_sw, const float* z, const float* psf, int B, int H, int W,
                                     int Kh, int Kw, int conv_flag, fc3d_layer_grid grid, float* raw_out,
                                     uint8_t* layer_u8, uint64_t* tilekey_u64, int* keys_per_image) {
    if (!pat0 || !z || !raw_out || !tilekey_u64 || !keys_per_image || B <= 0 || H <= 0 || W <= 0)
        return FC3D_EINVAL;
    if (grid.n_layers <= 0 || grid.n_layers > 254) return FC3D_EINVAL;
    if ((long)B * H * W >= (1L << 31) || B > 65535) return FC3D_ERANGE;
    if (!aligned4(pat0) || !aligned4(z) || !aligned4(raw_out)) return FC3D_EALIGN;
    size_t lds = 0;
    int cap = 0;
    if (conv_flag) {
        if (!psf || Kh <= 0 || Kw <= 0) return FC3D_EINVAL;
        cap = layer_capacity(Kh, Kw, grid.n_layers, false, &lds);
        if (cap < 1) return FC3D_ERANGE;   // kernel too large for the LDS tile
    }
    const bool vec = (W % 4 == 0) && aligned16(z) && aligned16(raw_out) &&
                     (!layer_u8 || aligned4(layer_u8));
    dim3 g(cdiv(W, TX), cdiv(H, TY), B), blk(NT);
    hipStream_t s = (hipStream_t)stream;
    const bool fits32 = (long)H * W < (1L << 30) && pat_sh >= 0 && pat_sw >= 0 &&
                        (long)(H - 1) * pat_sh + (long)(W - 1) * pat_sw < (1L << 30);   // byte offsets in 32 bits
    if (!conv_flag && (long)H * W < (1L << 30)) {
        const long HW = (long)H * W;
        const dim3 gf(cdiv(cdiv(HW, FLAT_WAVE_PX), NT / 64), B);
        const bool fvec = HW % 4 == 0 && W % 4 == 0 && HW >= 4 && pat_sw == 1 && pat_sh == W && pat_sb % 4 == 0 &&
                          aligned16(pat0) && aligned16(z) && aligned16(raw_out) && (!layer_u8 || aligned4(layer_u8));
        *keys_per_image = (int)(gf.x * (NT / 64));
        if (fvec) FC3D_LAUNCH("code_pattern_fwd", (code_pattern_flat_fwd_kernel<true, false>), gf, blk, 0, s, pat0, (long)pat_sb, (long)pat_sh, (long)pat_sw, z, H, W, grid, raw_out, layer_u8, (unsigned long long*)tilekey_u64);
        else FC3D_LAUNCH("code_pattern_fwd", (code_pattern_flat_fwd_kernel<false, false>), gf, blk, 0, s, pat0, (long)pat_sb, (long)pat_sh, (long)pat_sw, z, H, W, grid, raw_out, layer_u8, (unsigned long long*)tilekey_u64);
        return FC3D_OK;
    }
    if (conv_flag && Kh == Kw && fits32) {
#define FC3D_FWD_ROLL_CASE(KV)                                                                                   \
    if (Kw == KV) {                                                                                              \
        const size_t rlds = (size_t)grid.n_layers * Roll<KV>::KKP * sizeof(float);                               \
        auto kern = layer_grid_centre_is_integer(grid) ? code_pattern_fwd_roll_kernel<KV, RR, false, true>       \
                                                       : code_pattern_fwd_roll_kernel<KV, RR, false, false>;     \
        if (int e = allow_big_lds(kern, rlds)) return e;                                                         \
        const int nstrips = cdiv(W, Roll<KV>::SW);                                                               \
        const dim3 gr(cdiv((long)nstrips * cdiv(H, RR), NT / 64), B);                                            \
        *keys_per_image = (int)(gr.x * (NT / 64));                                                               \
        FC3D_LAUNCH("code_pattern_fwd", kern, gr, blk, rlds, s, pat0,                                            \
                    (long)pat_sb, (long)pat_sh, (long)pat_sw, z, psf, H, W, nstrips, grid, raw_out, layer_u8,    \
                    (unsigned long long*)tilekey_u64);                                                           \
        return FC3D_OK;                                                                                          \
    }
        constexpr int RR = kRollRows;
        FC3D_FWD_ROLL_CASE(3) FC3D_FWD_ROLL_CASE(5) FC3D_FWD_ROLL_CASE(7)
#undef FC3D_FWD_ROLL_CASE
#define FC3D_FWD_SWEEP_CASE(KV)                                                                                  \
    if (Kw == KV) {                                                                                              \
        using SW = Sweep<KV, kSweepRows, kSweepCols>;                                                            \
        const size_t slds = (size_t)SW::NR * SW::LW * sizeof(float);                                             \
        const int two_pass = sweep_two_pass(KV, B, H, W);                                                        \
        const dim3 gsw(cdiv(W, kSweepCols), cdiv(H, kSweepTileRows), two_pass ? 2 * B : B);                      \
        *keys_per_image = (int)(gsw.x * gsw.y);                                                                  \
        auto kern = layer_grid_centre_is_integer(grid) ? code_pattern_fwd_sweep_kernel<KV, kSweepRows, kSweepCols, true>  \
                                                       : code_pattern_fwd_sweep_kernel<KV, kSweepRows, kSweepCols, false>; \
        FC3D_LAUNCH("code_pattern_fwd", kern, gsw,                                                               \
                    dim3(64), slds, s, pat0, (long)pat_sb, (long)pat_sh, (long)pat_sw, z, psf, H, W, grid,       \
                    raw_out, layer_u8, (unsigned long long*)tilekey_u64, two_pass);                              \
        return FC3D_OK;                                                                                          \
    }
        FC3D_FOR_SWEEP_K(FC3D_FWD_SWEEP_CASE)
#undef FC3D_FWD_SWEEP_CASE
    }
    auto k = vec ? code_pattern_fwd_kernel<true> : code_pattern_fwd_kernel<false>;
    *keys_per_image = (int)(g.x * g.y);
    FC3D_LAUNCH("code_pattern_fwd", k, g, blk, lds, s, pat0, (long)pat_sb, (long)pat_sh, (long)pat_sw, z, psf, H, W, Kh,
                Kw, conv_flag, cap, grid, raw_out, layer_u8, (unsigned long long*)tilekey_u64);
    return FC3D_OK;
}

extern "C" int fc3d_code_pattern_shaded_fwd(void* stream, const float* pat0, int64_t pat_sb, int64_t pat_sh,
                                            int64_t pat_sw, const float* z, int B, int H, int W, fc3d_layer_grid grid,
                                            float* raw_out, uint8_t* layer_u8, uint64_t* tilekey_u64,
                                            int* keys_per_image) {
    if (!pat0 || !z || !raw_out || !tilekey_u64 || !keys_per_image || B <= 0 || H <= 0 || W <= 0) return FC3D_EINVAL;
    if (grid.n_layers <= 0 || grid.n_layers > 254) return FC3D_EINVAL;
    const long HW = (long)H * W;
    if ((long)B * HW >= (1L << 31) || B > 65535 || HW >= (1L << 30)) return FC3D_ERANGE;
    if (!aligned4(pat0) || !aligned4(z) || !aligned4(raw_out)) return FC3D_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    const dim3 gf(cdiv(cdiv(HW, FLAT_WAVE_PX), NT / 64), B), blk(NT);
    const bool fvec = HW % 4 == 0 && W % 4 == 0 && HW >= 4 && pat_sw == 1 && pat_sh == W && pat_sb % 4 == 0 &&
                      aligned16(pat0) && aligned16(z) && aligned16(raw_out) && (!layer_u8 || aligned4(layer_u8));
    *keys_per_image = (int)(gf.x * (NT / 64));
    if (fvec) FC3D_LAUNCH("code_pattern_shaded_fwd", (code_pattern_flat_fwd_kernel<true, true>), gf, blk, 0, s, pat0, (long)pat_sb, (long)pat_sh, (long)pat_sw, z, H, W, grid, raw_out, layer_u8, (unsigned long long*)tilekey_u64);
    else FC3D_LAUNCH("code_pattern_shaded_fwd", (code_pattern_flat_fwd_kernel<false, true>), gf, blk, 0, s, pat0, (long)pat_sb, (long)pat_sh, (long)pat_sw, z, H, W, grid, raw_out, layer_u8, (unsigned long long*)tilekey_u64);
    return FC3D_OK;
}

extern "C" int fc3d_code_pattern_shaded_conv_fwd(void* stream, const float* pat0, int64_t pat_sb, int64_t pat_sh,
                                                int64_t pat_sw, const float* z, const float* psf, int B, int H, int W,
                                                int Kh, int Kw, fc3d_layer_grid grid, float* raw_out, uint8_t* layer_u8,
                                                uint64_t* tilekey_u64, int* keys_per_image) {
    if (!pat0 || !z || !psf || !raw_out || !tilekey_u64 || !keys_per_image || B <= 0 || H <= 0 || W <= 0) return FC3D_EINVAL;
    if (grid.n_layers <= 0 || grid.n_layers > 254) return FC3D_EINVAL;
    if (Kh != Kw || (Kw != 3 && Kw != 5 && Kw != 7)) return FC3D_EINVAL;      // the rolling-row sizes
    const long HW = (long)H * W;
    if ((long)B * HW >= (1L << 31) || B > 65535 || HW >= (1L << 30)) return FC3D_ERANGE;
    if (pat_sh < 0 || pat_sw < 0 || (long)(H - 1) * pat_sh + (long)(W - 1) * pat_sw >= (1L << 30)) return FC3D_ERANGE;
    if (!aligned4(pat0) || !aligned4(z) || !aligned4(raw_out)) return FC3D_EALIGN;
    hipStream_t s = (hipStream_t)stream;
    dim3 blk(NT);
    constexpr int RR = kRollRows;
#define FC3D_FWD_SHADE_CASE(KV)                                                                                  \
    if (Kw == KV) {                                                                                              \
        const size_t rlds = (size_t)grid.n_layers * Roll<KV>::KKP * sizeof(float);                               \
        auto kern = layer_grid_centre_is_integer(grid) ? code_pattern_fwd_roll_kernel<KV, RR, true, true>        \
                                                       : code_pattern_fwd_roll_kernel<KV, RR, true, false>;      \
        if (int e = allow_big_lds(kern, rlds)) return e;                                                         \
        const int nstrips = cdiv(W, Roll<KV>::SW);                                                               \
        const dim3 gr(cdiv((long)nstrips * cdiv(H, RR), NT / 64), B);                                            \
        *keys_per_image = (int)(gr.x * (NT / 64));                                                               \
        FC3D_LAUNCH("code_pattern_shaded_fwd", kern, gr, blk, rlds, s,                                           \
                    pat0, (long)pat_sb, (long)pat_sh, (long)pat_sw, z, psf, H, W, nstrips, grid, raw_out,        \
                    layer_u8, (unsigned long long*)tilekey_u64);                                                 \
        return FC3D_OK;                                                                                          \
    }
    FC3D_FWD_SHADE_CASE(3) FC3D_FWD_SHADE_CASE(5) FC3D_FWD_SHADE_CASE(7)
#undef FC3D_FWD_SHADE_CASE
    return FC3D_EINVAL;
}

extern "C" int fc3d_code_pattern_norm(void* stream, const float* raw, const uint64_t* tilekey_u64,
                                      int keys_per_image, int B, int H, int W, float* out, float* max_out,
                                      int32_t* argmax_i32) {
    if (!raw || !tilekey_u64 || !out || !max_out || !argmax_i32 || B <= 0 || H <= 0 || W <= 0) return FC3D_EINVAL;
    if (keys_per_image <= 0 || (size_t)keys_per_image * B * 8 > fc3d_code_pattern_fwd_workspace(B, H, W))
        return FC3D_EINVAL;
    if ((long)B * H * W >= (1L << 31) || B > 65535) return FC3D_ERANGE;
    const int ntiles = keys_per_image;
    const bool vec = (((long)H * W) % 4 == 0) && aligned16(raw) && aligned16(out);
    const int ahead = ntiles <= NORM_KEYS_PLAIN ? 1 : ntiles <= NORM_KEYS_FOUR ? 4 : 8;
    dim3 g(cdiv((long)H * W, NT * 4 * NORM_V), B), blk(NT);
    auto k = vec ? (ahead == 8 ? code_pattern_norm_kernel<true, 8> : ahead == 4 ? code_pattern_norm_kernel<true, 4> : code_pattern_norm_kernel<true, 1>)
                 : (ahead == 8 ? code_pattern_norm_kernel<false, 8> : ahead == 4 ? code_pattern_norm_kernel<false, 4> : code_pattern_norm_kernel<false, 1>);
    FC3D_LAUNCH("code_pattern_norm", k, g, blk, 0, (hipStream_t)stream, raw, (const unsigned long long*)tilekey_u64,
                ntiles, H, W, out, max_out, argmax_i32);
    return FC3D_OK;
}

extern "C" int fc3d_code_pattern_bwd(void* stream, const float* g_out, const float* out, const uint8_t* layer_u8,
                                     const float* max_in, const int32_t* argmax_i32, const float* psf, int B, int H,
                                     int W, int L, int Kh, int Kw, int conv_flag, void* workspace, float* g_pat0) {
    if (!g_out || !out || !layer_u8 || !max_in || !argmax_i32 || !workspace || !g_pat0) return FC3D_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || L <= 0 || L > 254) return FC3D_EINVAL;
    if ((long)B * H * W >= (1L << 31) || B > 65535) return FC3D_ERANGE;
    size_t lds = 0;
    int cap = 0;
    if (conv_flag) {
        if (!psf || Kh <= 0 || Kw <= 0) return FC3D_EINVAL;
        cap = layer_capacity(Kh, Kw, L, true, &lds);
        if (cap < 1) return FC3D_ERANGE;
    }
    hipStream_t s = (hipStream_t)stream;
    const long HW = (long)H * W;
    const int nred = cdiv(HW, RED_PX);
    float* partial = (float*)workspace;
    if (!conv_flag && HW < (1L << 30)) {
        const dim3 gf(cdiv(cdiv(HW, FLAT_WAVE_PX), NT / 64), B);
        const bool fvec = HW % 4 == 0 && HW >= 4 && aligned16(g_out) && aligned16(out) && aligned16(g_pat0) && aligned4(layer_u8);
        if (fvec) FC3D_LAUNCH("code_pattern_bwd", (code_pattern_flat_bwd_kernel<true, false>), gf, dim3(NT), 0, s, g_out, out, layer_u8, max_in, (unsigned)HW, partial, g_pat0, (const int*)nullptr, (const float*)nullptr, 0);
        else FC3D_LAUNCH("code_pattern_bwd", (code_pattern_flat_bwd_kernel<false, false>), gf, dim3(NT), 0, s, g_out, out, layer_u8, max_in, (unsigned)HW, partial, g_pat0, (const int*)nullptr, (const float*)nullptr, 0);
        FC3D_LAUNCH("code_pattern_bwd_max", code_pattern_bwd_max_kernel, dim3(B), dim3(NT), 0, s, (const float*)partial,
                    (int)(gf.x * (NT / 64)), max_in, argmax_i32, layer_u8, (const float*)nullptr, H, W, 1, g_pat0);
        return FC3D_OK;
    }
    if (conv_flag && Kh == Kw && HW < (1L << 30)) {
        constexpr int R = kRollRowsBwd;
#define FC3D_BWD_ROLL_CASE(KV)                                                                                   \
    if (Kw == KV) {                                                                                              \
        const size_t rlds = (size_t)L * Roll<KV>::KKP * sizeof(float);                                           \
        if (int e = allow_big_lds(code_pattern_bwd_roll_kernel<KV, R>, rlds)) return e;                          \
        const int nstrips = cdiv(W, Roll<KV>::SW);                                                               \
        const dim3 gr(cdiv((long)nstrips * cdiv(H, R), NT / 64), B);                                             \
        FC3D_LAUNCH("code_pattern_bwd", (code_pattern_bwd_roll_kernel<KV, R>), gr, dim3(NT), rlds, s, g_out,     \
                    out, layer_u8, max_in, psf, H, W, L, nstrips, partial, g_pat0, (const int*)nullptr,          \
                    (const float*)nullptr, 0);                                                                   \
        FC3D_LAUNCH("code_pattern_bwd_max", code_pattern_bwd_max_kernel, dim3(B), dim3(NT), 0, s,                \
                    (const float*)partial, (int)(gr.x * (NT / 64)), max_in, argmax_i32, layer_u8, psf, H,        \
                    W, KV, g_pat0);                                                                              \
        return FC3D_OK;                                                                                          \
    }
        FC3D_BWD_ROLL_CASE(3) FC3D_BWD_ROLL_CASE(5) FC3D_BWD_ROLL_CASE(7)
#undef FC3D_BWD_ROLL_CASE
#define FC3D_BWD_SWEEP_CASE(KV)                                                                                  \
    if (Kw == KV) {                                                                                              \
        const size_t slds = sweep_bwd_lds<KV, kSweepRows, kSweepCols>();                                         \
        const int two_pass = sweep_two_pass(KV, B, H, W);                                                        \
        const dim3 gsw(cdiv(W, kSweepCols), cdiv(H, kSweepTileRows), two_pass ? 2 * B : B);                      \
        FC3D_LAUNCH("code_pattern_bwd", (code_pattern_bwd_sweep_kernel<KV, kSweepRows, kSweepCols>), gsw,        \
                    dim3(64), slds, s, g_out, out, layer_u8, max_in, psf, H, W, L, partial, g_pat0,              \
                    (const int*)nullptr, (const float*)nullptr, 0, two_pass);                                    \
        FC3D_LAUNCH("code_pattern_bwd_max", code_pattern_bwd_max_kernel, dim3(B), dim3(NT), 0, s,                \
                    (const float*)partial, (int)(gsw.x * gsw.y), max_in, argmax_i32, layer_u8, psf, H, W, KV,    \
                    g_pat0);                                                                                     \
        return FC3D_OK;                                                                                          \
    }
        FC3D_FOR_SWEEP_K(FC3D_BWD_SWEEP_CASE)
#undef FC3D_BWD_SWEEP_CASE
    }
    const bool rvec = (HW % 4 == 0) && aligned16(g_out) && aligned16(out);
    auto kr = rvec ? code_pattern_bwd_reduce_kernel<true> : code_pattern_bwd_reduce_kernel<false>;
    FC3D_LAUNCH("code_pattern_bwd_reduce", kr, dim3(nred, B), dim3(NT), 0, s, g_out, out, HW, partial);
    const bool vec = (W % 4 == 0) && aligned16(g_pat0);
    auto k = vec ? code_pattern_bwd_kernel<true> : code_pattern_bwd_kernel<false>;
    FC3D_LAUNCH("code_pattern_bwd", k, dim3(cdiv(W, TX), cdiv(H, TY), B), dim3(NT), lds, s, g_out, layer_u8, max_in,
                argmax_i32, (const float*)partial, nred, psf, H, W, L, Kh, Kw, conv_flag, cap, g_pat0);
    return FC3D_OK;
}

extern "C" int fc3d_code_pattern_bwd_dot_supported(int Kh, int Kw, int conv_flag) {
    return !conv_flag || (Kh == Kw && Kw >= 3 && Kw <= 33 && (Kw & 1));
}

extern "C" int fc3d_code_pattern_bwd_dot(void* stream, const float* g_out, const float* dot_partial, int n_partial,
                                         const uint8_t* layer_u8, const float* max_in, const int32_t* argmax_i32,
                                         const float* psf, int B, int H, int W, int L, int Kh, int Kw, int conv_flag,
                                         float* g_pat0) {
    if (!g_out || !dot_partial || n_partial <= 0 || !layer_u8 || !max_in || !argmax_i32 || (conv_flag && !psf) || !g_pat0)
        return FC3D_EINVAL;
    if (B <= 0 || H <= 0 || W <= 0 || L <= 0 || L > 254) return FC3D_EINVAL;
    if (!fc3d_code_pattern_bwd_dot_supported(Kh, Kw, conv_flag)) return FC3D_EINVAL;
    const long HW = (long)H * W;
    if ((long)B * HW >= (1L << 31) || B > 65535 || HW >= (1L << 30)) return FC3D_ERANGE;
    hipStream_t s = (hipStream_t)stream;
    if (!conv_flag) {       // is_conv_psf=False: the streaming kernel, 1x1 footprint
        const dim3 gf(cdiv(cdiv(HW, FLAT_WAVE_PX), NT / 64), B);
        const bool fvec = HW % 4 == 0 && HW >= 4 && aligned16(g_out) && aligned16(g_pat0) && aligned4(layer_u8);
        if (fvec) FC3D_LAUNCH("code_pattern_bwd", (code_pattern_flat_bwd_kernel<true, true>), gf, dim3(NT), 0, s, g_out, (const float*)nullptr, layer_u8, max_in, (unsigned)HW, (float*)nullptr, g_pat0, (const int*)argmax_i32, dot_partial, n_partial);
        else FC3D_LAUNCH("code_pattern_bwd", (code_pattern_flat_bwd_kernel<false, true>), gf, dim3(NT), 0, s, g_out, (const float*)nullptr, layer_u8, max_in, (unsigned)HW, (float*)nullptr, g_pat0, (const int*)argmax_i32, dot_partial, n_partial);
        return FC3D_OK;
    }
    constexpr int R = kRollRowsBwdDot;
#define FC3D_BWD_DOT_CASE(KV)                                                                                    \
    if (Kw == KV) {                                                                                              \
        const size_t rlds = (size_t)L * Roll<KV>::KKP * sizeof(float);                                           \
        if (int e = allow_big_lds(code_pattern_bwd_roll_kernel<KV, R, true>, rlds)) return e;                    \
        const int nstrips = cdiv(W, Roll<KV>::SW);                                                               \
        const dim3 gr(cdiv((long)nstrips * cdiv(H, R), NT / 64), B);                                             \
        FC3D_LAUNCH("code_pattern_bwd", (code_pattern_bwd_roll_kernel<KV, R, true>), gr, dim3(NT), rlds, s,      \
                    g_out, (const float*)nullptr, layer_u8, max_in, psf, H, W, L, nstrips, (float*)nullptr,      \
                    g_pat0, (const int*)argmax_i32, dot_partial, n_partial);                                     \
        return FC3D_OK;                                                                                          \
    }
    FC3D_BWD_DOT_CASE(3) FC3D_BWD_DOT_CASE(5) FC3D_BWD_DOT_CASE(7)
#undef FC3D_BWD_DOT_CASE
#define FC3D_BWD_DOT_SWEEP_CASE(KV)                                                                              \
    if (Kw == KV) {                                                                                              \
        const size_t slds = sweep_bwd_lds<KV, kSweepRows, kSweepCols>();                                         \
        const int two_pass = sweep_two_pass(KV, B, H, W);                                                        \
        const dim3 gsw(cdiv(W, kSweepCols), cdiv(H, kSweepTileRows), two_pass ? 2 * B : B);                      \
        FC3D_LAUNCH("code_pattern_bwd", (code_pattern_bwd_sweep_kernel<KV, kSweepRows, kSweepCols, true>), gsw,  \
                    dim3(64), slds, s, g_out, (const float*)nullptr, layer_u8, max_in, psf, H, W, L,             \
                    (float*)nullptr, g_pat0, (const int*)argmax_i32, dot_partial, n_partial, two_pass);          \
        return FC3D_OK;                                                                                          \
    }
    FC3D_FOR_SWEEP_K(FC3D_BWD_DOT_SWEEP_CASE)
#undef FC3D_BWD_DOT_SWEEP_CASE
    return FC3D_EINVAL;
}
