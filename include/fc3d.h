/* fc3d.h -- C ABI of libfc3d_hip.so: MI355X (gfx950) kernels for the FreeCam3D
 * image-formation path (depth-layered PSF blur; pose reprojection, bilinear
 * warp, forward-splat visibility and compositing), forward and backward.
 *
 * The reference reaches this path through plain Python functions on torch
 * tensors (it has no FFI).  Each entry point below names the reference function
 * whose tensor arithmetic it replaces; paths are relative to the reference
 * checkout.  The Python modules optics_pytorch.py / geometry_pytorch.py of this
 * project keep the reference's names and signatures and call these symbols
 * through ctypes with raw device pointers.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the comment says "host";
 *  - all tensors are fp32 unless the name ends in _u8 / _i32 / _u64;
 *  - images are dense row-major [B][H][W][C]; where an input may be a strided
 *    view the element strides are passed explicitly;
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *    every call only enqueues work and returns, nothing synchronises, nothing
 *    allocates: the caller owns outputs and workspaces;
 *  - return value: 0 ok, <0 invalid argument (FC3D_E*), >0 a hipError_t raised
 *    by a launch.  fc3d_status_string() explains either.
 */
#ifndef FC3D_H
#define FC3D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FC3D_OK 0
#define FC3D_EINVAL (-1)     /* null pointer / non-positive size */
#define FC3D_ERANGE (-2)     /* size beyond what the kernels index (e.g. B*H*W >= 2^31) */
#define FC3D_EALIGN (-3)     /* pointer not aligned to its element type */
#define FC3D_ESCRATCH (-4)   /* problem larger than the library's static scratch buffer */

#define FC3D_VERSION 212

int fc3d_version(void);
const char *fc3d_status_string(int status);

/* ------------------------------------------------------------------------ */
/* optics: codePattern, optics_pytorch.py:71-121 (with z2Wm :58-68 and the     */
/* one-hot layer mask gen_WmMask :30-55 evaluated per pixel on the fly)        */
/* ------------------------------------------------------------------------ */

/* Depth -> layer decision, all in fp32 like the reference's tensor ops:
 *   Wm = coeff * (1/z - inv_z0);  r = rint(Wm / step);  l = rint(r - centre0);
 *   valid iff 0 <= l < L and |r - (centre0 + l)| < 0.01.
 * The caller computes the four scalars in float64 exactly as the reference's
 * Python does (np.linspace of the layer centres) and casts them to fp32.      */
typedef struct fc3d_layer_grid {
    float coeff;     /* k*(D/2)^2/2 cast to fp32      (optics_pytorch.py:66-67) */
    float inv_z0;    /* 1/z0                          (optics_pytorch.py:67)    */
    float step;      /* layer spacing in Wm, fp32     (optics_pytorch.py:41)    */
    float centre0;   /* Wm_layers[0]/step, fp32       (optics_pytorch.py:44)    */
    int32_t n_layers;
} fc3d_layer_grid;

/* Bytes of the tile-key buffer shared by fc3d_code_pattern_fwd/_norm, and of the
 * workspace fc3d_code_pattern_bwd needs (both tiny: one word per 1024 pixels).  */
size_t fc3d_code_pattern_fwd_workspace(int B, int H, int W);
size_t fc3d_code_pattern_bwd_workspace(int B, int H, int W);

/* Forward, first half: un-normalised blurred image and the per-image maximum.
 *   pat0        channel 0 of the pattern as a strided view: element (b,y,x) at
 *               pat0[b*pat_sb + y*pat_sh + x*pat_sw]   (optics_pytorch.py:97)
 *   z           [B][H][W] depth
 *   psf         [L][Kh][Kw] cross-correlation weights (optics_pytorch.py:90,105);
 *               ignored when conv_flag == 0 (the is_conv_psf=False branch :110-112)
 *   raw_out     [B][H][W]  sum_l conv_l * mask_l                  (:115)
 *   layer_u8    [B][H][W]  layer index, 255 = no layer (may be NULL)
 *   tilekey_u64 fc3d_code_pattern_fwd_workspace() bytes: one packed key per output
 *               tile (ordered max value | first-w-then-h position priority), fully
 *               overwritten.  Zero 'same' padding: (K-1)/2 before, the rest after.  */
int fc3d_code_pattern_fwd(void *stream, const float *pat0, int64_t pat_sb, int64_t pat_sh, int64_t pat_sw,
                          const float *z, const float *psf, int B, int H, int W, int Kh, int Kw,
                          int conv_flag, fc3d_layer_grid grid,
                          float *raw_out, uint8_t *layer_u8, uint64_t *tilekey_u64,
                          int *keys_per_image /* HOST, out: keys written per image; pass to _norm */);

/* withReflectance (utils_pytorch.py:251-280) followed by the first half of codePattern without
 * blur (optics_pytorch.py:110-112) in one pass -- train_pytorch.py:91-94 as the reference's training
 * loop runs it (is_conv_psf=False, :51-52):  raw = pattern0 * (0.8 cos(atan(|grad z|/100e-6)) + 0.2)
 * where z falls into a layer, 0 elsewhere.  Same arguments and outputs as fc3d_code_pattern_fwd
 * (follow with fc3d_code_pattern_norm); identical to fc3d_reflectance_fwd + fc3d_code_pattern_fwd
 * with conv_flag 0 on the same inputs.  No backward: the reference runs it without gradients.   */
int fc3d_code_pattern_shaded_fwd(void *stream, const float *pat0, int64_t pat_sb, int64_t pat_sh, int64_t pat_sw,
                                 const float *z, int B, int H, int W, fc3d_layer_grid grid, float *raw_out,
                                 uint8_t *layer_u8, uint64_t *tilekey_u64,
                                 int *keys_per_image /* HOST, out */);

/* The same fusion with the blur (is_conv_psf=True): withReflectance (utils_pytorch.py:251-280) folded
 * into the rolling-row forward kernel of fc3d_code_pattern_fwd, for square 3x3 / 5x5 / 7x7 PSFs
 * (FC3D_EINVAL otherwise: shade with fc3d_reflectance_fwd, then fc3d_code_pattern_fwd).  Results are
 * bit-identical to those two calls.  Followed by fc3d_code_pattern_norm like every forward.     */
int fc3d_code_pattern_shaded_conv_fwd(void *stream, const float *pat0, int64_t pat_sb, int64_t pat_sh, int64_t pat_sw,
                                      const float *z, const float *psf, int B, int H, int W, int Kh, int Kw,
                                      fc3d_layer_grid grid, float *raw_out, uint8_t *layer_u8,
                                      uint64_t *tilekey_u64, int *keys_per_image);

/* Forward, second half: out = raw / (max + 1e-10) per image, in place allowed
 * (optics_pytorch.py:118-119).  Also unpacks max_out[B] and argmax_i32[B]
 * (flat y*W+x of the pixel that receives the max's gradient).                 */
int fc3d_code_pattern_norm(void *stream, const float *raw, const uint64_t *tilekey_u64, int keys_per_image,
                           int B, int H, int W, float *out, float *max_out, int32_t *argmax_i32);

/* Backward w.r.t. pattern[...,0] (autograd of optics_pytorch.py:97-119).
 *   g_out [B][H][W] upstream gradient; out = the normalised forward result.   */
int fc3d_code_pattern_bwd(void *stream, const float *g_out, const float *out, const uint8_t *layer_u8,
                          const float *max_in, const int32_t *argmax_i32, const float *psf,
                          int B, int H, int W, int L, int Kh, int Kw, int conv_flag,
                          void *workspace, float *g_pat0);

/* The same backward (autograd of optics_pytorch.py:97-119) when sum(g_out * out) per image is
 * known already, as n_partial partial sums per image (dot_partial[b * n_partial + i], added here in a
 * fixed order; fc3d_bilinear_bwd_img_dot produces them while it writes g_out): `out` is not read,
 * no workspace is needed, and the share of the per-image maximum goes in at its source pixel, so the
 * whole backward is one launch.  Square odd PSFs from 3x3 to 33x33, or conv_flag == 0 (psf may be
 * NULL then) (fc3d_code_pattern_bwd_dot_supported); FC3D_EINVAL otherwise.                    */
int fc3d_code_pattern_bwd_dot_supported(int Kh, int Kw, int conv_flag);
int fc3d_code_pattern_bwd_dot(void *stream, const float *g_out, const float *dot_partial, int n_partial,
                              const uint8_t *layer_u8, const float *max_in, const int32_t *argmax_i32,
                              const float *psf, int B, int H, int W, int L, int Kh, int Kw, int conv_flag,
                              float *g_pat0);

/* ------------------------------------------------------------------------ */
/* geometry                                                                    */
/* ------------------------------------------------------------------------ */

/* Per-sample 4x4 products used by coord0TOgrid1 / warp_p2c / zp_cView_to_zc:
 *   inv_out   [B][16] inverse of pose                      (geometry_pytorch.py:271,294)
 *   t_fwd_out [B][16] K * pose,  t_inv_out [B][16] K * inverse(pose)   (:103-104)
 * K is the intrinsic matrix get_intrinsic (:238-252) returns, read from device
 * memory at K[b*k_sb + 4*i + j] (k_sb = 0 for one matrix shared by the batch).
 * Any output may be NULL (K may be NULL if only inv_out is wanted).  The inverse
 * is computed in fp64 and rounded once; the 4x4 products use separately rounded
 * multiplies and left-to-right adds like torch's small-matrix path.
 * clear / clear_bytes (NULL / 0 for none; 16-byte aligned): a buffer the same
 * launch fills with zeros -- the visibility mask fc3d_warp_p2c_fwd would
 * otherwise clear with a launch of its own.                                   */
int fc3d_pose_transforms(void *stream, const float *pose, const float *K, int64_t k_sb, int B,
                         float *t_fwd_out, float *t_inv_out, float *inv_out,
                         void *clear, size_t clear_bytes);

/* coord_transform, geometry_pytorch.py:68-90: grid[b][y][x] = (xu, yu)/(zu+1e-10),
 * u = T[b] * coord[b][:, y, x].  coord element (b,c,y,x) at
 * coord[b*sb + c*sc + y*sh + x*sw] (the reference is handed both planar
 * [B,4,H,W] tensors and permuted [B,H,W,4] ones, :234).                       */
int fc3d_project_fwd(void *stream, const float *coord, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                     const float *T, int B, int H, int W, float *grid_out);

/* Backward of the above w.r.t. coord: g_coord is dense planar [B][4][H][W].   */
int fc3d_project_bwd(void *stream, const float *g_grid, const float *coord, int64_t sb, int64_t sc,
                     int64_t sh, int64_t sw, const float *T, int B, int H, int W, float *g_coord);

/* Backward of the same product w.r.t. the transform (torch.matmul is differentiable in both
 * arguments, geometry_pytorch.py:79): g_T [B][16], g_T[i][k] = sum over pixels of g_u[i]*coord[k]
 * for rows 0..2, row 3 zero.  workspace: fc3d_project_bwd_transform_workspace bytes (one partial of
 * the twelve sums per 512 pixels, added in a fixed order: reproducible).  The gradient w.r.t. a pose
 * behind T = K*pose is K^T g_T (coord0TOgrid1, :103-104).                                        */
size_t fc3d_project_bwd_transform_workspace(int B, int H, int W);
int fc3d_project_bwd_transform(void *stream, const float *g_grid, const float *coord, int64_t sb, int64_t sc,
                               int64_t sh, int64_t sw, const float *T, int B, int H, int W, void *workspace,
                               float *g_T);

/* bilinear_sampler, geometry_pytorch.py:109-177.  img [B][Hs][Ws][C],
 * grid [B][Ht][Wt][2] -> out [B][Ht][Wt][C].  mask_u8 [B][Ht][Wt] (may be NULL)
 * multiplies the result (the compositing step of warp_p2c, :279).             */
int fc3d_bilinear_fwd(void *stream, const float *img, const float *grid, const uint8_t *mask_u8,
                      int B, int Hs, int Ws, int C, int Ht, int Wt, float *out);

/* d out / d grid: g_grid [B][Ht][Wt][2].                                      */
int fc3d_bilinear_bwd_grid(void *stream, const float *g_out, const float *img, const float *grid,
                           const uint8_t *mask_u8, int B, int Hs, int Ws, int C, int Ht, int Wt,
                           float *g_grid);

/* d out / d img (the backward of the index gathers at geometry_pytorch.py:166-169: a scatter-add of
 * the four weighted taps), written -- not accumulated -- to g_img [B][Hs][Ws][C].
 * C == 1 with `index` (fc3d_bilinear_bwd_img_workspace() bytes, 8-byte aligned): gather form.  Every
 * wavefront owns a 64x12 tile of the SOURCE image, finds through the tap index the target pixels
 * that can touch it, sums their taps in LDS and stores the tile once: no atomics, no zero fill, a
 * fixed summation order (bit-reproducible).  The tap index (two levels of boxes that say where runs
 * of target pixels landed) is built by the call itself when index_ready == 0; fc3d_warp_p2c_fwd can
 * write it while it samples (then pass index_ready != 0), and fc3d_bilinear_tap_index builds it on
 * its own.  The index depends on grid, mask and the sizes only.  Built for locally smooth warps: a
 * grid that sends neighbouring pixels all over the image stays correct but makes every tile visit
 * every pixel.
 * C == 1 with index == NULL: scatter form -- per-wavefront LDS windows added to the zero-filled
 * g_img with float atomics (last-bit differences from run to run, like the reference's
 * index_put/scatter backward on a GPU).  C != 1 always uses atomics.
 * fc3d_bilinear_bwd_img_workspace returns 0 when the gather form does not apply (C != 1, a source
 * side beyond 32767 pixels).                                                                  */
size_t fc3d_bilinear_bwd_img_workspace(int B, int Hs, int Ws, int C, int Ht, int Wt);
int fc3d_bilinear_tap_index(void *stream, const float *grid, const uint8_t *mask_u8, int B, int Hs, int Ws,
                            int Ht, int Wt, void *index);
int fc3d_bilinear_bwd_img(void *stream, const float *g_out, const float *grid, const uint8_t *mask_u8,
                          int B, int Hs, int Ws, int C, int Ht, int Wt, void *index, int index_ready,
                          float *g_img);

/* The gather form above (C == 1, index ready) that also returns what the backward of a
 * normalisation of img by one of its own statistics needs -- the inner product of the gradient it
 * has just formed with img, the image that was sampled (codePattern divides by the per-image
 * maximum, optics_pytorch.py:118-119; its backward needs sum(g * Ip_coded)): one partial per source
 * tile, dot_partial[b * n + t], n = fc3d_bilinear_bwd_img_dot_count(B, Hs, Ws), the sum over tile t of
 * g_img * img, each tile summed in a fixed order.  fc3d_code_pattern_bwd_dot takes them.  (The
 * count depends on the batch as well: small problems are cut into shorter tiles.)               */
int fc3d_bilinear_bwd_img_dot_count(int B, int Hs, int Ws);
int fc3d_bilinear_bwd_img_dot(void *stream, const float *g_out, const float *grid, int B, int Hs, int Ws,
                              int Ht, int Wt, void *index, const float *img, float *g_img,
                              float *dot_partial);

/* gen_visible_mask, geometry_pytorch.py:180-222.  method 0 = 'all', 1 = 'round'.
 * Either output may be NULL; both are cleared by the call.                    */
int fc3d_visible_mask(void *stream, const float *grid, int B, int H, int W, int method,
                      float *mask_f32, uint8_t *mask_u8);

/* warp_p2c forward, geometry_pytorch.py:255-281, given T_p2c = K*pose and
 * T_c2p = K*inverse(pose): writes grid_p2c, grid_c2p, the visibility mask and
 * Ic_masked = sample(Ip, grid_c2p) * mask.  mask_is_clear != 0: mask_u8 already
 * holds zeros (see fc3d_pose_transforms), skip clearing it.  tap_index (may be
 * NULL): fc3d_bilinear_bwd_img_workspace(B,H,W,1,H,W) bytes that receive the tap
 * index of grid_c2p under the mask, for the backward pass w.r.t. Ip.          */
int fc3d_warp_p2c_fwd(void *stream, const float *Ip,
                      const float *coord_p, int64_t p_sb, int64_t p_sc, int64_t p_sh, int64_t p_sw,
                      const float *coord_c, int64_t c_sb, int64_t c_sc, int64_t c_sh, int64_t c_sw,
                      const float *T_p2c, const float *T_c2p, int B, int H, int W,
                      float *Ic_masked, float *grid_p2c, float *grid_c2p, uint8_t *mask_u8,
                      int mask_is_clear, void *tap_index);

/* The same forward (geometry_pytorch.py:255-281, with :271-272 and :103-104 inside) straight from
 * pose [B][16] and the intrinsics K (read at K[b*k_sb + 4*i + j] as in fc3d_pose_transforms): the
 * projection kernel forms K*pose itself and its first wavefront per sample leaves K*pose,
 * K*inverse(pose) and inverse(pose) in T_p2c_out, T_c2p_out [B][16] (both required; T_c2p_out
 * feeds the sampling kernel) and inv_out [B][16] (may be NULL) -- the values fc3d_pose_transforms
 * gives, without its launch.  mask_rezero != 0: the sampling kernel, the last reader of the
 * visibility mask, stores zeros back, so that the caller can hand the same buffer to its next
 * call with mask_is_clear != 0 and never clears it again (calls that share the buffer must be
 * ordered on one stream; the mask is not available afterwards, which only the atomic form of the
 * image gradient needs).                                                                        */
int fc3d_warp_p2c_pose_fwd(void *stream, const float *Ip,
                           const float *coord_p, int64_t p_sb, int64_t p_sc, int64_t p_sh, int64_t p_sw,
                           const float *coord_c, int64_t c_sb, int64_t c_sc, int64_t c_sh, int64_t c_sw,
                           const float *pose, const float *K, int64_t k_sb, int B, int H, int W,
                           float *Ic_masked, float *grid_p2c, float *grid_c2p, uint8_t *mask_u8,
                           int mask_is_clear, int mask_rezero, void *tap_index,
                           float *T_p2c_out, float *T_c2p_out, float *inv_out);

/* fc3d_warp_p2c_pose_fwd for scene points that are points_in_camera_coords of a depth map
 * (geometry_pytorch.py:44-65 feeding :255-281, as z2pointcloud :225-235 and every caller in
 * train_pytorch.py produce them): instead of the stored points [B][4][H][W], 16 bytes per pixel and
 * warp, the kernels read the depth planes z_p, z_c [B][H][W] they were made from and form each point
 * as it was stored, (z ray_x[x], z ray_y[y], z 1, 1) with separately rounded products (ray_x [W],
 * ray_y [H]: the separable pixel rays of geometry_pytorch.py:14-41).  Results are bit-identical to
 * the stored-points form; everything else as in fc3d_warp_p2c_pose_fwd.                         */
int fc3d_warp_p2c_pose_depth_fwd(void *stream, const float *Ip, const float *z_p, const float *z_c,
                                 const float *ray_x, const float *ray_y,
                                 const float *pose, const float *K, int64_t k_sb, int B, int H, int W,
                                 float *Ic_masked, float *grid_p2c, float *grid_c2p, uint8_t *mask_u8,
                                 int mask_is_clear, int mask_rezero, void *tap_index,
                                 float *T_p2c_out, float *T_c2p_out, float *inv_out);

/* The two launches of fc3d_warp_p2c_pose_fwd on their own, for callers that issue them apart -- the
 * first (geometry_pytorch.py:276-277 with :103-104: projection of coord_p and the visibility splat)
 * needs neither Ip nor coord_c, so it can run on another stream beside the work that produces Ip;
 * the second (:271-273, :279) needs the mask and T_c2p_out of the first.  Same arguments and
 * results as in fc3d_warp_p2c_pose_fwd.                                                         */
int fc3d_warp_project_splat(void *stream,
                            const float *coord_p, int64_t p_sb, int64_t p_sc, int64_t p_sh, int64_t p_sw,
                            const float *pose, const float *K, int64_t k_sb, int B, int H, int W,
                            float *grid_p2c, uint8_t *mask_u8, int mask_is_clear,
                            float *T_p2c_out, float *T_c2p_out, float *inv_out);
int fc3d_warp_sample(void *stream, const float *Ip,
                     const float *coord_c, int64_t c_sb, int64_t c_sc, int64_t c_sh, int64_t c_sw,
                     const float *T_c2p, int B, int H, int W, uint8_t *mask_u8, int mask_rezero,
                     float *Ic_masked, float *grid_c2p, void *tap_index);

/* points_in_camera_coords, geometry_pytorch.py:44-65: out [B][H][W][4] =
 * (z*rx, z*ry, z*rz, 1).  Rays either separable (ray_x[W], ray_y[H], rz = 1:
 * the table pixel_to_ray_array :30-41 produces) or a dense table with batch
 * stride ray_sb (0 for a shared [H][W][3] table).                             */
int fc3d_points_fwd(void *stream, const float *z, const float *ray_x, const float *ray_y,
                    const float *ray_dense, int64_t ray_sb, int B, int H, int W, float *out);
int fc3d_points_bwd(void *stream, const float *g_out, const float *ray_x, const float *ray_y,
                    const float *ray_dense, int64_t ray_sb, int B, int H, int W, float *g_z);

/* zp_cView_to_zc, geometry_pytorch.py:284-308, given inv = inverse(pose) [B][16]:
 * zc = (zp - a23) / (a20*x + a21*y + a22).  Backward w.r.t. zp.               */
int fc3d_zp_to_zc_fwd(void *stream, const float *zp, const float *xy, const float *inv, int B, int H, int W,
                      float *zc);
int fc3d_zp_to_zc_bwd(void *stream, const float *g_zc, const float *xy, const float *inv, int B, int H, int W,
                      float *g_zp);

/* The whole backward of zp_cView_to_zc (the reference's expression, geometry_pytorch.py:294-306, is
 * differentiable in all three arguments): from g_zc and the forward result zc,
 *   g_zp [B][H][W], g_xy [B][H][W][2] and g_inv [B][16], the gradient w.r.t. inverse(pose) (row 2 holds the
 * four per-sample sums, the other rows are zero; the gradient w.r.t. the pose itself follows through the
 * inverse: -A^T g_A A^T).  Any of the three may be NULL; g_inv needs workspace of
 * fc3d_zp_to_zc_bwd_full_workspace bytes, 16-byte aligned (one partial per 256 pixels, added in a fixed order). */
size_t fc3d_zp_to_zc_bwd_full_workspace(int B, int H, int W);
int fc3d_zp_to_zc_bwd_full(void *stream, const float *g_zc, const float *zc, const float *xy, const float *inv,
                           int B, int H, int W, float *g_zp, float *g_xy, void *workspace, float *g_inv);

/* Backward of points_in_camera_coords w.r.t. a per-sample ray map (allBatch, geometry_pytorch.py:57-63, the
 * form reProj_IpGT calls with rays built from xy, train_pytorch.py:196-197): g_ray [B][H][W][3] =
 * g_out[..., 0:3] * z.                                                                            */
int fc3d_points_bwd_rays(void *stream, const float *g_out, const float *z, int B, int H, int W, float *g_ray);

/* The differentiated chain of reProj_IpGT, train_pytorch.py:194-202, in one pass:
 * zc = zp_cView_to_zc(zp, pose, xy) (geometry_pytorch.py:284-308), scene points
 * (zc*x, zc*y, zc, 1) (:44-65, allBatch), grid = coord0TOgrid1(points, inverse(pose))
 * (:93-106), out = bilinear_sampler(img, grid) (:109-177).  inv = inverse(pose),
 * T = K*inverse(pose), both [B][16] (fc3d_pose_transforms); xy [B][H][W][2];
 * zp, zc, out [B][H][W]; img [B][Hs][Ws]; grid [B][H][W][2] may be NULL.  Results are those of the
 * separate entry points applied in sequence.  Backward w.r.t. zp from the gradients of
 * zc and out (either may be NULL); img, xy and the pose carry no gradient here.        */
int fc3d_reproject_fwd(void *stream, const float *zp, const float *xy, const float *inv, const float *T,
                       const float *img, int B, int H, int W, int Hs, int Ws, float *zc, float *out, float *grid);
int fc3d_reproject_bwd(void *stream, const float *g_zc, const float *g_out, const float *zc, const float *xy,
                       const float *inv, const float *T, const float *img, int B, int H, int W, int Hs, int Ws,
                       float *g_zp);

/* The re-projection loss of reProj_IpGT, train_pytorch.py:188-207, on top of that chain, for
 * a mask m that is already eroded (fc3d_box_filter mode 2, :184-186): zp = zp_hat * m (:188),
 * zc and Ic' as fc3d_reproject_fwd computes them, and
 *   sums[0] = sum (m (zcGT - zc))^2    sums[1] = sum (m (Ic - Ic'))^2      (cost_rms_mask numerators,
 *   sums[2] = sum m |Ic - Ic'|         sums[3] = sum m                      utils_pytorch.py:161-162, :177)
 * over the whole batch, and from them, with the reference's fp32 expressions (:163, :177, :209),
 *   losses = [loss_Ic_L1 + loss_zc_rms, loss_zc_rms, loss_Ic_rms, loss_Ic_L1],
 *   rms = sqrt(S / max(sums[3], 1e-6) + 1e-6),  loss_Ic_L1 = sums[2] / (sums[3] + 1)
 * -- the four values reProj_IpGT returns (:211).
 * zp_hat, mask, Ic, zcGT [B][H][W]; xy [B][H][W][2]; img [B][Hs][Ws]; sums, losses [4] device
 * floats.  zc and Ic' are not stored.  workspace: fc3d_reproj_loss_workspace bytes, 16-byte
 * aligned (one partial per 256 pixels, added in a fixed order: reproducible).  Backward:
 * g_losses [4] device floats (the gradient w.r.t. losses) and the forward's sums -> g_zp_hat
 * [B][H][W]; mask, Ic, zcGT, img, xy and the pose carry no gradient.                          */
size_t fc3d_reproj_loss_workspace(int B, int H, int W);
int fc3d_reproj_loss_fwd(void *stream, const float *zp_hat, const float *mask, const float *xy, const float *inv,
                         const float *T, const float *img, const float *Ic, const float *zcGT, int B, int H, int W,
                         int Hs, int Ws, void *workspace, float *sums, float *losses);
int fc3d_reproj_loss_bwd(void *stream, const float *g_losses, const float *sums, const float *zp_hat,
                         const float *mask, const float *xy, const float *inv, const float *T, const float *img,
                         const float *Ic, const float *zcGT, int B, int H, int W, int Hs, int Ws, float *g_zp_hat);

/* fc3d_reproj_loss_fwd for callers that cannot allocate or pass device memory beyond inputs and
 * outputs (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
 * It has no counterpart in the reference.  Same arguments, kernels and results as
 * fc3d_reproj_loss_fwd, minus the workspace: the partial sums go to a 1 MiB array that is part of
 * the library's code object on every GPU (looked up once per device, on the first call made with
 * that device current; the call never allocates or frees).  Limit: B * ceil(H*W/256) <= 65536
 * partial sums (e.g. 64 images of 512x512), otherwise FC3D_ESCRATCH and nothing is launched.
 * All calls on one GPU share that array: calls whose kernels may overlap on one GPU (different
 * streams, or several host threads) race on it and must be serialised by the caller.
 * The first call on each device looks the array's address up (hipGetSymbolAddress, which may load the
 * code object): make that call OUTSIDE any stream capture -- one eager call per device at set-up; every
 * later call, captured ones included, only enqueues the two kernels.                            */
int fc3d_reproj_loss_fwd_static(void *stream, const float *zp_hat, const float *mask, const float *xy,
                                const float *inv, const float *T, const float *img, const float *Ic,
                                const float *zcGT, int B, int H, int W, int Hs, int Ws, float *sums,
                                float *losses);

/* ------------------------------------------------------------------------ */
/* the capture simulator around the path (forward_model, train_pytorch.py:     */
/* 79-118; the reference runs it without gradients, so these are forward only)  */
/* ------------------------------------------------------------------------ */

/* withReflectance, utils_pytorch.py:251-280 (with image_gradients, :206-222):
 * out = I * (0.8*cos(atan(|grad z| / 100e-6)) + 0.2), forward differences, zero on the
 * last row / column.  I [B][H][W] with batch stride i_sb elements (0: one pattern for
 * every sample); z, out [B][H][W].  Floating-point result (1e-6 relative of the composed
 * tensor ops: cos(atan(t)) is evaluated as 1/sqrt(1+t^2)).                             */
int fc3d_reflectance_fwd(void *stream, const float *I, int64_t i_sb, const float *z, int B, int H, int W,
                         float *out);

/* train_pytorch.py:86-87, 97, 107-116 in two launches: scene points from the two depth
 * maps and the separable ray table (ray_x [W], ray_y [H]: pixel_to_ray_array's columns 0
 * and 1, geometry_pytorch.py:15-41), warp_p2c (geometry_pytorch.py:255-281), and
 * xyz_cview = bilinear_sampler(cat(xy, z_p), grid_c2p) * visible_mask.  Ip, z_p, z_c,
 * Ic_masked [B][H][W]; xyz_cview [B][H][W][3]; mask_u8 [B][H][W] (the 'all' visibility
 * mask, see fc3d_warp_p2c_fwd for mask_is_clear); grid_p2c / grid_c2p [B][H][W][2] are
 * written only when not NULL.  Bit-identical to fc3d_points_fwd + fc3d_warp_p2c_fwd +
 * fc3d_bilinear_fwd on the same inputs, without the scene points, the grids and the
 * concatenated image passing through memory.                                            */
int fc3d_capture_warp_fwd(void *stream, const float *Ip, const float *z_p, const float *z_c, const float *ray_x,
                          const float *ray_y, const float *T_p2c, const float *T_c2p, int B, int H, int W,
                          float *Ic_masked, float *xyz_cview, uint8_t *mask_u8, int mask_is_clear, float *grid_p2c,
                          float *grid_c2p);

/* The sensor model, train_pytorch.py:100-104 and :118, with the random draws passed in:
 *   e = Ic + offset[b]*0.05 + gauss*0.005;  Ic_scaled = clamp(e / max_image(e) * (scale[b]*0.3 + 0.7), 0, 1);
 *   Ic_mask = (Ic_scaled > 0.05) * visible_mask.
 * offset, scale [B] (uniform draws in [0,1)), gauss [B][H][W] (standard normal draws),
 * mask_u8 [B][H][W] from the warp; outputs [B][H][W] fp32.  Same operations in the same
 * order as the tensor expression, so the results are bit-identical to it; a NaN in an
 * image makes that image's maximum NaN, as torch.max does.  workspace: at least
 * fc3d_capture_post_workspace bytes, 4-byte aligned, need not be initialised.            */
size_t fc3d_capture_post_workspace(int B, int H, int W);
int fc3d_capture_post(void *stream, const float *Ic, const float *offset, const float *gauss, const float *scale,
                      const uint8_t *mask_u8, int B, int H, int W, float *Ic_scaled, float *Ic_mask, void *workspace);

/* The capture simulator after codePattern in three launches (train_pytorch.py:94-104, :118): the
 * forward warp forms K*pose itself (see fc3d_warp_p2c_pose_fwd), the inverse warp finds the partial
 * maxima of the exposed image while it writes Ic (no pass of its own for the sensor model's peak),
 * the sensor model, as the mask's last reader, hands it back cleared when mask_rezero != 0.
 * Results are bit-identical to fc3d_pose_transforms + fc3d_capture_warp_fwd + fc3d_capture_post.
 * workspace: fc3d_capture_fwd_workspace() bytes (the two transforms and the partial maxima);
 * grid_p2c / grid_c2p may be NULL.  No counterpart as one function in the reference: it is the
 * body of forward_model between codePattern and the crops.                                    */
size_t fc3d_capture_fwd_workspace(int B, int H, int W);
int fc3d_capture_fwd(void *stream, const float *Ip, const float *z_p, const float *z_c,
                     const float *ray_x, const float *ray_y, const float *pose, const float *K, int64_t k_sb,
                     const float *offset, const float *gauss, const float *scale, int B, int H, int W,
                     float *Ic_masked, float *xyz_cview, float *Ic_scaled, float *Ic_mask,
                     uint8_t *mask_u8, int mask_is_clear, int mask_rezero, void *workspace,
                     float *grid_p2c, float *grid_c2p);

/* k x k box mean with zero padding, F.avg_pool2d(x, k, stride=1, padding=k/2) on one-channel
 * images [B][H][W] (k odd, <= 33), and its two uses in the training loop:
 *   mode 0  out = mean
 *   mode 1  out = in / (mean + c)       ln, utils_pytorch.py:285-301 (k 17, c 0.01)
 *   mode 2  out = mean > 0.999 ? 1 : 0  mask erosion, train_pytorch.py:184-186, :222-224 (k 5),
 *                                       utils_pytorch.py:193-194 (k 9); bit-exact for 0/1 masks
 * The mean is a floating-point result (row sums, then a column sum: rounding-level difference
 * from a row-major window sum).  in and out must not alias.  No gradient: the reference applies
 * these to the simulated capture and to masks only.                                          */
int fc3d_box_filter(void *stream, const float *in, int B, int H, int W, int k, int mode, float c, float *out);

/* ------------------------------------------------------------------------ */
/* optional per-kernel timing (no counterpart in the reference, which has no   */
/* profiling hooks): while enabled every kernel launch of this library is       */
/* bracketed by a HIP event pair on the launching stream.  fc3d_profile_report  */
/* waits for those events and writes one line per kernel, "name launches        */
/* total_ms\n", into the HOST buffer buf, then forgets them.                    */
/* ------------------------------------------------------------------------ */
int fc3d_profile_enable(int on);
int fc3d_profile_report(char *buf, size_t len);

#ifdef __cplusplus
}
#endif
#endif /* FC3D_H */
