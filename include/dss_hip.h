/*
 * dss_hip.h -- C ABI of libdss_hip.so, the MI355X (gfx950) implementation of the DSS
 * differentiable EWA surface-splatting hot path.
 *
 * This is the drop-in boundary for the reference's native module `DSS._C`
 * (/root/reference/DSS/csrc/ext.cpp:5-18).  Every entry point names the reference interface it
 * replaces.  Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless it says "host";
 *   - all tensors are dense row-major float32 / int32 / int64 / uint8 exactly as the reference's
 *     torch tensors are laid out after .contiguous() (rasterize_points.cu:648-663);
 *   - the caller owns every buffer (outputs and workspace); nothing is allocated inside;
 *   - work is enqueued asynchronously on `stream` (a hipStream_t; NULL = default stream); no host
 *     synchronisation happens inside the library; re-entrant; no environment variable is read and no
 *     hidden global state is kept: the only process-wide state is the explicit option table of
 *     dss_set_option() below (tuning hints, all 0 by default) and a per-device-ordinal cache of
 *     device properties (CU count, occupancy), filled with atomics on first use;
 *   - return value: 0 on success, negative DSS_ERR_* otherwise (no exceptions cross the ABI);
 *     dss_last_error() returns a thread-local message for the last failing call on this thread;
 *   - `row0,row1` select the image row band [row0,row1) a rank renders (multi-GPU row
 *     partitioning); band-shaped tensors have `rows = row1-row0` rows.  Single GPU: 0, S.
 *     The fused training entry points (dss_render_forward / dss_render_backward) also take `row_cycle`:
 *     1 = that contiguous band; c > 1 (a power of two) = TILE-ROW-CYCLIC band: of [row0,row1) only every
 *     c-th 8-row tile row, starting at row0 (rank g of c ranks passes row0 = 8 g, row1 = S, row_cycle = c:
 *     every rank gets the same mix of dense and empty screen regions instead of one contiguous strip).
 *     Band tensors then have dss_band_rows(row0,row1,row_cycle) rows; band row l is image row
 *     row0 + (l / 8) * 8 c + l % 8.
 *
 * Image convention (rasterization_utils.cuh:8-11, rasterize_points.cu:160-164, 577-580):
 * image pixel [row r, col c] has NDC centre ( -1+(2*(S-1-c)+1)/S , -1+(2*(S-1-r)+1)/S ).
 */
#ifndef DSS_HIP_H
#define DSS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSS_HIP_VERSION 103

#if defined(__GNUC__)
#define DSS_API __attribute__((visibility("default")))
#else
#define DSS_API
#endif

#define DSS_OK 0
#define DSS_ERR_INVALID_ARGUMENT (-1) /* shape / range check failed (reference: TORCH_CHECK / checkSize) */
#define DSS_ERR_WORKSPACE (-2)        /* workspace pointer NULL or too small */
#define DSS_ERR_UNSUPPORTED (-3)      /* legal in the reference but not implemented here */
#define DSS_ERR_LAUNCH (-4)           /* hipGetLastError() after a launch (reference: AT_CUDA_CHECK) */

/* Largest points_per_pixel with a register-resident K-list.  The reference's compile-time bound
 * is kMaxPointsPerPixel = 150 (rasterization_utils.cuh:18); K in (DSS_MAX_K_FAST, 150] is served
 * by a slower scratch-memory kernel (fine_generic_kernel). */
#define DSS_MAX_K_FAST 32
#define DSS_MAX_K 150

DSS_API int dss_version(void);
DSS_API const char *dss_last_error(void);

/* Explicit tuning options (process-wide, relaxed atomics; change them between calls, not during one).  A workspace
 * size query and the launch it sizes must see the same DSS_OPT_LEAN_WORKSPACE value.
 *   DSS_OPT_LEAN_WORKSPACE 1: half the tile sub-list capacity and no packed 64-byte records in the forward workspace
 *                             (433 -> 110 MB at 4M points @2048^2 for ~+13 % step time); 0 (default): the fast layout.
 *   DSS_OPT_BACKWARD_TPW   visible points per wavefront of the backward gather: 1, 2 or 4; 0 (default) = chosen from P.
 *   DSS_OPT_BACKWARD_ADDR64 1: force the 64-bit addressing variant of the backward gather (normally only taken when a
 *                             gathered tensor exceeds 4 GB); exists so that a test can reach that variant.
 *   DSS_OPT_KNN_QUERY       query kernel of dss_knn_kth_sqdist / dss_knn_points: 0 (default) = chosen from P and K (the
 *                           cooperative kernel -- 16 lanes per query -- while the launch is latency-bound, one thread per
 *                           query above), 1 = cooperative, 2 = one thread per query (K <= 16 for 1; A/B measurements);
 *                           3 = as 0 without the skip structure of clustered clouds (from 65,536 points on, cells with more
 *                           than 64 points are ordered along a Morton curve and walked in 16-slot blocks whose boxes are
 *                           tested against the current K-th distance: knn.hip, knn_subsort_kernel).  Results do not depend
 *                           on it.
 *   DSS_OPT_BACKWARD_FUSED  launch form of dss_render_backward for short lists (P <= 262,144, whole image): 0 (default) =
 *                             automatic; 1 = the round-3 sequence (compaction | median | gather kernels); 4 = two launches
 *                             (segments + alpha plane | medians + gather: the gather's workgroups do the blend half of their
 *                             tasks while the first N of them select the medians); 5 = three (segments + alpha plane |
 *                             medians | gather).  4 and 5 use the bucket-sorted median of raster_backward.hip.  Results do
 *                             not depend on it.
 * Returns DSS_ERR_INVALID_ARGUMENT for an unknown option or value. */
#define DSS_OPT_LEAN_WORKSPACE 0
#define DSS_OPT_BACKWARD_TPW 1
#define DSS_OPT_BACKWARD_ADDR64 2
#define DSS_OPT_BACKWARD_FUSED 3
#define DSS_OPT_KNN_QUERY 4
#define DSS_OPT_COUNT 5
DSS_API int dss_band_rows(int row0, int row1, int row_cycle); /* rows of a band tensor (see `row_cycle` above) */
DSS_API int dss_set_option(int option, int value);
DSS_API int dss_get_option(int option);

/* ---------------------------------------------------------------------------------------------
 * Forward rasterizer.
 * Replaces  DSS._C.splat_points  (ext.cpp:8) = RasterizePoints (rasterize_points.h:461-525):
 * naive kernel rasterize_points.cu:131-212 for bin_size==0, coarse+fine kernels :293-432 and
 * :506-597 otherwise.  Both modes return identical results here; bin_size only selects whether
 * screen-tile lists are built (bin_size != 0) or every tile scans its whole cloud (bin_size == 0).
 *
 *   points   (P,3)  NDC x, NDC y, view-space z      ellipse (P,3)  a,b,c of Q = a dx^2 + b dx dy + c dy^2
 *   cutoff   (P,)   per-point Q threshold           radii   (P,2)  axis-aligned NDC half extents
 *   first_idx, num_pts (N,) int64                   cloud_to_packed_first_idx / num_points_per_cloud
 *   merge_thr       depth_merging_threshold         S image side, K points_per_pixel
 * outputs (band shaped, fully written, no pre-fill needed):
 *   idx int32 (N,rows,S,K)  zbuf f32 (N,rows,S,K)  qvalue f32 (N,rows,S,K)  occ f32 (N,rows,S)
 *   visible uint8 (P,) or NULL: set to 1 for every point that appears in a fragment of this band,
 *           0 otherwise (replaces get_per_point_visibility_mask, DSS/utils/__init__.py:320-340,
 *           called at rasterizer.py:639-641 and again in the backward at :854-860).
 * Per-pixel rule: hit iff pz>=0, |dx|<=rx, |dy|<=ry, Q<=cutoff; keep the K smallest (z, idx);
 * ascending; drop k with z[k]-z[0] > merge_thr; unfilled slots idx=-1, zbuf=-1, qvalue=-1.
 * ------------------------------------------------------------------------------------------- */
DSS_API size_t dss_splat_forward_workspace(int N, int64_t P, int S, int K, int bin_size);
/* Leading bytes of that workspace which binning expects zero-filled (tile counters, tile flags, queue tails and queue
 * slots): the region the DSS_WS_CLEAN contract of dss_render_forward is about. */
DSS_API size_t dss_splat_forward_clean_bytes(int N, int64_t P, int S);

DSS_API int dss_splat_forward(const float *points, const float *ellipse, const float *cutoff,
                      const float *radii, const int64_t *first_idx, const int64_t *num_pts,
                      int N, int64_t P, float merge_thr, int S, int K, int bin_size,
                      int row0, int row1,
                      int32_t *idx, float *zbuf, float *qvalue, float *occ, uint8_t *visible,
                      void *workspace, size_t workspace_bytes, void *stream);

/* The two phases of dss_splat_forward, callable separately (the binned lists in `workspace` stay
 * valid until the next dss_splat_bin on it):
 *   dss_splat_bin   tile binning: a memset + ONE kernel that appends every splat to the fixed-capacity
 *                   sub-lists of the 8x8-pixel tiles it overlaps and every tile that receives its first
 *                   splat to a queue of occupied tiles (replaces the coarse kernel
 *                   rasterize_points.cu:293-432 and its dense (N,B,B,M) bin table)
 *   dss_splat_fine  exactly ONE kernel launch: per-tile K-nearest + stores for the queued tiles, fill values
 *                   for the empty ones (replaces the fine kernel rasterize_points.cu:506-597);
 *                   workspace==NULL scans whole clouds (naive mode, :131-212).  `visible`, if given, must
 *                   have been zeroed by the caller.  zbuf may be NULL: the depth plane is then not written
 *                   (the fused backward never reads it; saves 4K of the 12K+24 bytes stored per pixel). */
DSS_API int dss_splat_bin(const float *points, const float *radii, const int64_t *first_idx,
                          const int64_t *num_pts, int N, int64_t P, int S, int row0, int row1,
                          void *workspace, size_t workspace_bytes, void *stream);
DSS_API int dss_splat_fine(const float *points, const float *ellipse, const float *cutoff,
                           const float *radii, const int64_t *first_idx, const int64_t *num_pts,
                           int N, int64_t P, float merge_thr, int S, int K, int row0, int row1,
                           int32_t *idx, float *zbuf, float *qvalue, float *occ, uint8_t *visible,
                           const void *workspace, size_t workspace_bytes, void *stream);

/* dss_splat_fine with the blend fused into its epilogue (still exactly ONE kernel launch): besides the
 * fragments it writes image (N,rows,S,C+1) and wsum (N,rows,S) exactly like dss_blend_forward would.
 * This is the kernel dss_render_forward launches; K <= DSS_MAX_K_FAST, 1 <= C <= 8. */
DSS_API int dss_splat_fine_blend(const float *points, const float *ellipse, const float *cutoff,
                                 const float *radii, const int64_t *first_idx, const int64_t *num_pts,
                                 int N, int64_t P, float merge_thr, int S, int K, int row0, int row1,
                                 int32_t *idx, float *zbuf, float *qvalue, float *occ, uint8_t *visible,
                                 const float *scaler, const float *feat, int C, float *image, float *wsum,
                                 const void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Backward of the rasterizer = EllipticalRasterizer.backward (rasterizer.py:787-977) with
 * backward_occ_fast=True (:816), in four pieces so that a multi-GPU caller can reduce between
 * them; dss_splat_backward() chains all four for the single-GPU case.
 * ------------------------------------------------------------------------------------------- */

/* rs[n] = lower_median( flattened (x,y) radii of the visible points of cloud n ) * radii_s
 * (rasterizer.py:885-888, torch.median).  Clouds without visible points get rs = 0. */
DSS_API size_t dss_backward_radius_workspace(int N, int64_t P);
DSS_API int dss_backward_radius(const float *radii, const uint8_t *visible, const int64_t *first_idx,
                        const int64_t *num_pts, int N, int64_t P, float radii_s,
                        float *rs /* (N,) */, void *workspace, size_t workspace_bytes, void *stream);

/* Occupancy surrogate gradient.  Replaces the Python grid build (rasterizer.py:889-950: FRNN
 * insert_points_cuda / prefix_sum_cuda / counting_sort_cuda) plus
 * DSS._C._splat_points_occ_fast_cuda_backward (ext.cpp:14, rasterize_points_backward.cu:30-212):
 *   for every pixel (r,c) of the band with g = grad_occ != 0 and every VISIBLE point p of the same
 *   cloud with pz>=0, |px|<=1, |py|<=1, d2 = dx^2+dy^2 <= rs[n]^2:
 *       skip if g>0 and (|dx|>rx or |dy|>ry);   grad_xy[p] += (dx,dy)/max(d2,1e-10)*g
 *   (a pair with d2 == 0 contributes 0; the reference produces NaN there).
 * Writes grad_pts[:,0:2] for ALL points (0 for invisible ones) and sets grad_pts[:,2] = 0.
 * grad_pixel_stride: elements between consecutive pixels of grad_occ (1 = dense (N,rows,S); C+1 =
 * read the alpha channel of an (N,rows,S,C+1) image gradient in place, no copy).
 * clip > 0 fuses the per-point clip hook (dss_clip_grad) into the final store; only valid when no
 * zbuf gradient and no cross-rank reduction follows (single GPU, grad_zbuf == NULL).  Pass <= 0 otherwise.
 * Gather formulation: one wavefront per point, no atomics, deterministic. */
DSS_API int dss_occ_backward(const float *points, const float *radii, const uint8_t *visible,
                     const float *rs, const float *grad_occ /* (N,rows,S) */,
                     const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P, int S,
                     int row0, int row1, int grad_pixel_stride, float clip,
                     float *grad_pts /* (P,3) */, void *stream);

/* Box-supported occupancy surrogate: DSS._C._splat_points_occ_backward on CUDA tensors (ext.cpp:10, 16;
 * RasterizePointsOccBackwardCudaKernel, rasterize_points.cu:672-757).  Not on the training path
 * (`backward_occ_fast = True`, rasterizer.py:816); kept so that every DSS._C export has a counterpart.
 *   for every pixel with g = grad_occ != 0 and every point p of the same cloud with pz>=0, |px|<=1, |py|<=1,
 *   R = radii[p]*radii_s, |dx|<=Rx, |dy|<=Ry:  skip if g>0 and (|dx|>Rx/radii_s or |dy|>Ry/radii_s);
 *   grad_xy[p] += (dx,dy)/max(d2,1e-10)*g     (d2 == 0 contributes 0; the reference yields NaN).
 * grad_occ dense (N,S,S); grad_xy (P,2) fully written.  One wavefront per point, no atomics. */
DSS_API int dss_occ_backward_box(const float *points, const float *radii, const float *grad_occ,
                                 const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P, int S,
                                 float radii_s, float *grad_xy /* (P,2) */, void *stream);

/* Replaces DSS._C._backward_zbuf (ext.cpp:17, rasterize_points.cu:823-846): accumulates IN PLACE
 * z_grad[idx[n,r,c,k]] += grad_zbuf[n,r,c,k] (zero grads skipped, stop at first idx<0), into the
 * z column of grad_pts (P,3). */
DSS_API int dss_zbuf_backward(const int32_t *idx, const float *grad_zbuf, int N, int rows, int S, int K,
                      float *grad_pts /* (P,3), in/out */, void *stream);

/* Per-point gradient clip hook (rasterizer.py:667-673, installed at :735-737), in place:
 * g <- g / max(||g||,1e-12) * min(||g||, clip).  No-op when clip <= 0. */
DSS_API int dss_clip_grad(float *grad_pts /* (P,3) */, int64_t P, float clip, void *stream);

DSS_API size_t dss_splat_backward_workspace(int N, int64_t P);
DSS_API int dss_splat_backward(const float *points, const float *radii, const uint8_t *visible,
                       const int32_t *idx, const float *grad_occ,
                       const float *grad_zbuf /* NULL = all zero */,
                       const int64_t *first_idx, const int64_t *num_pts,
                       int N, int64_t P, int S, int K, int grad_pixel_stride, float radii_s, float clip,
                       float *grad_pts /* (P,3), fully written */, float *rs_out /* (N,) or NULL */,
                       void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Blend = SurfaceSplattingRenderer.forward after rasterization (renderer.py:53-78):
 *   w_k = exp(-0.5*Q_k) * scaler[idx_k]          (renderer.py:53, rasterizer.py:631-633)
 *   img[ch] = sum_k feat[idx_k][ch] * w_k / max(sum_k w_k, 1e-4)   (pytorch3d NormWeightedCompositor,
 *                                                 renderer.py:67-72; third-party norm_weighted_sum)
 *   out[..., C] = occ                            (renderer.py:75-78)
 * feat is (P,C) row-major (Pointclouds.features_packed()); out is (N,rows,S,C+1). 1 <= C <= 8.
 * ------------------------------------------------------------------------------------------- */
DSS_API int dss_blend_forward(const int32_t *idx, const float *qvalue, const float *occ,
                              const float *scaler, const float *feat, int N, int rows, int S, int K, int C,
                              float *out, float *wsum /* (N,rows,S) or NULL: max(sum_k w_k, 1e-4) */,
                              void *stream);

/* Backward of the blend to the per-point features:
 *   grad_feat[p][ch] = sum over fragments (pixel,k) with idx == p of grad_out[pixel][ch] * w_k / wsum[pixel]
 * evaluated point-centric (one wavefront per visible point gathers over the pixels of its own
 * bounding box |dx|<=rx, |dy|<=ry): no atomics, deterministic, grad_feat (P,C) fully written.
 * wsum = the optional output of dss_blend_forward (NULL: recomputed per pixel).
 * The gradient w.r.t. occupancy is simply grad_out[..., C] (read it in place with
 * dss_occ_backward's grad_pixel_stride = C+1).  The gradient w.r.t. the weights is not produced:
 * the reference discards it (rasterizer.py:788-789 ignores qvalue_grad; EWA terms are detached,
 * :562-565). */
DSS_API int dss_blend_backward(const float *grad_out /* (N,rows,S,C+1) */, const int32_t *idx,
                               const float *qvalue, const float *wsum, const float *scaler,
                               const float *points, const float *radii, const uint8_t *visible,
                               const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P,
                               int S, int K, int C, int row0, int row1, float *grad_feat /* (P,C) */,
                               void *stream);

/* Same gradient, pixel-centric scatter with one atomicAdd per fragment and channel (the shape of
 * pytorch3d's norm_weighted_sum backward); for callers that hold fragments but not the splat
 * geometry.  grad_feat (P,C) is zeroed here first. */
DSS_API int dss_blend_backward_scatter(const float *grad_out, const int32_t *idx, const float *qvalue,
                                       const float *scaler, int N, int rows, int S, int K, int C,
                                       int64_t P, float *grad_feat, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused forward of SurfaceSplattingRenderer.forward (renderer.py:36-82 incl. the whole
 * SurfaceSplatting.forward, rasterizer.py:584-664) in one call and two launches (+ one memset unless
 * DSS_WS_CLEAN): [setup + tile binning] -> [fine + blend].  Same inputs as dss_point_setup +
 * dss_splat_forward + dss_blend_forward, same outputs (all of them are written: the per-point
 * screen-space arrays, the fragments, visibility, the (N,rows,S,C+1) image and wsum), same bits.
 * K <= DSS_MAX_K_FAST, 1 <= C <= 8.
 *
 * image_cam_stride / image_row_stride: element strides of `image` over (camera, band row); 0, 0 = dense
 * (N,rows,S,C+1).  A pixel's C+1 channels and the pixels of a row are always contiguous.  The multi-GPU layer
 * passes a (row, camera, col, channel) send buffer so that the all-gathered bands are the full image.
 *
 * workspace_state: DSS_WS_UNKNOWN (0) -- contents arbitrary: the call zeroes the tile counters itself (one
 * memset launch).  DSS_WS_CLEAN (1) -- the caller guarantees that the first dss_splat_forward_clean_bytes()
 * bytes are either zero-filled (once, after allocation) or were left by a previous SUCCESSFUL
 * DSS_WS_CLEAN call with the same (N, P, S) on this buffer, and that nothing else wrote to them since:
 * the memset launch is skipped and the fine pass restores the all-zero state on its way out (every counter,
 * flag and queue slot has exactly one owning workgroup that resets it after its last read).
 * DSS_WS_BINNED (2) -- the workspace was last used by a DSS_WS_UNKNOWN call with exactly these inputs (which leaves the
 * tile lists, the queue and the packed records in place): only the second launch ([fine + blend]) is repeated, e.g. to
 * time or profile the dominant kernel in isolation.  The per-point outputs are not rewritten.
 * ------------------------------------------------------------------------------------------- */
#define DSS_WS_UNKNOWN 0
#define DSS_WS_CLEAN 1
#define DSS_WS_BINNED 2
/* Renderer-owned cached point order (flags, OR-ed into workspace_state DSS_WS_UNKNOWN / DSS_WS_CLEAN).  Above 2,000,000
 * points the binning runs in screen-cell order, which the call creates with a counting sort of its own (the reference bins
 * in input order, rasterize_points.cu:293-432; on a randomly ordered cloud that costs returning atomics and partial-sector
 * appends).  A training loop moves its points a little per iteration, so the order of one iteration
 * serves the next ones:
 *   DSS_WS_ORDER_SAVE   sort as usual, but keep the order in the workspace as a permutation of ALL points (the culled ones
 *                       behind the others), valid until a call without either flag sorts on the same workspace again;
 *   DSS_WS_ORDER_REUSE  no sort: setup in natural order, then ONE binning pass that walks the saved order (two launches
 *                       instead of six).  Needs an earlier DSS_WS_ORDER_SAVE call with the same (N, P, S) on this workspace
 *                       (DSS_ERR_INVALID_ARGUMENT otherwise -- tracked on the host per workspace pointer; the caller must
 *                       not hand in a re-allocated buffer that happens to have the same address).
 * The outputs do not depend on the order in any bit (a pixel's K-set is independent of the order of its tile's candidates);
 * a stale order only costs locality.  Both flags are ignored where the direct binning runs (P <= 2,000,000). */
#define DSS_WS_ORDER_SAVE 0x10
#define DSS_WS_ORDER_REUSE 0x20
/* Multi-GPU row bands (flag, OR-ed into workspace_state; ignored for the whole image): every rank evaluates the setup of
 * every splat -- the backward needs the screen position and the radii of the points that are visible in OTHER ranks' bands
 * (median search radius, windows that reach across a band boundary) --, but only the splats whose pixel rectangle meets the
 * rank's own rows are binned, so only those have their ellipse, scaler and cutoff read again.
 *   DSS_WS_BAND_OUTPUTS  pts_screen, radii and valid are written for every point; ellipse, scaler and cutoff only for the
 *                        splats that meet the band (the rest of those three arrays keeps whatever the buffers held), and the
 *                        other splats are left out of the call's sort / marked in its binning records.  The fragments, the
 *                        image and the visibility flags are the same bits; dss_render_backward on the same band reads the
 *                        scaler of a point only where the point has a fragment.  (The reference has no counterpart: it
 *                        materialises every per-point tensor in PyTorch, rasterizer.py:525-565.) */
#define DSS_WS_BAND_OUTPUTS 0x40
DSS_API size_t dss_render_forward_workspace(int N, int64_t P, int S, int K);
DSS_API int dss_render_forward(const float *world, const float *normals, const float *h_point,
                               const float *h_cloud, const float *vr6, const float *frame_normals,
                               const float *M, const float *V, const float *znear,
                               const float *zfar, const int64_t *first_idx, const int64_t *num_pts,
                               int N, int64_t P, int shared_cloud, int backface_culling, int S, int K,
                               float cutoff_threshold, float antialiasing_sigma, float merge_thr,
                               int row0, int row1, int row_cycle, const float *feat /* (P,C) */, int C,
                               float *pts_screen, float *ellipse, float *radii, float *scaler,
                               float *cutoff, uint8_t *valid, int32_t *idx, float *zbuf, float *qvalue,
                               float *occ, uint8_t *visible, float *image,
                               int64_t image_cam_stride, int64_t image_row_stride, float *wsum,
                               void *workspace, size_t workspace_bytes, int workspace_state, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused single-GPU backward of renderer + rasterizer: dss_blend_backward + dss_backward_radius +
 * dss_occ_backward + dss_clip_grad in three launches for P <= 262,144 ([compaction of the visible ids and
 * radius keys + alpha plane] -> [median] -> [gathers]; six above), with the per-point work done by persistent
 * wavefronts over the compacted list of visible points (the stand-alone kernels are bound by the
 * workgroup dispatch rate at DSS sizes).  No zbuf gradient; the occupancy gradient is the alpha channel
 * of grad_out (N,rows,S,C+1); it is first copied into a dense (N,rows,S) plane in the workspace (the
 * occupancy gather reads ~30 x 30 pixel windows per visible point: at a 16-byte stride it is bound by
 * L2 bandwidth, three quarters of every cache line fetched being colour gradient it does not need).
 * With a row band (multi-GPU) `visible` must be the union
 * over all ranks, the outputs are this band's partial sums, and clip must be <= 0 (clip after the
 * all-reduce with dss_clip_grad).
 * grad_feat may be NULL (rasterizer backward only).  grad_pts (P,3) and grad_feat (P,C) are fully
 * written.  Same results as the unfused entry points (same per-point arithmetic and reduction order).
 * ------------------------------------------------------------------------------------------- */
/* `world` != NULL fuses the backward of the projection (dss_project_backward, pytorch3d transform at rasterizer.py:614)
 * into the same launch: grad_pts then receives the WORLD-space position gradients (clip applied first, like the separate
 * kernel).  For clouds that are not shared between cameras only (packed index == world index), whole image, C == 3. */
DSS_API size_t dss_render_backward_workspace(int N, int64_t P, int S);
DSS_API int dss_render_backward(const float *grad_out, const int32_t *idx, const float *qvalue,
                                const float *wsum, const float *scaler, const float *points,
                                const float *radii, const uint8_t *visible, const int64_t *first_idx,
                                const int64_t *num_pts, int N, int64_t P, int S, int K, int C,
                                int row0, int row1, int row_cycle, float radii_s, float clip,
                                float *grad_feat /* (P,C) or NULL */,
                                float *grad_pts /* (P,3) */, float *rs_out /* (N,) or NULL */,
                                const float *world /* NULL, or (P,3): see below */, const float *M /* (N,4,4) with world */,
                                void *workspace, size_t workspace_bytes, void *stream);
/* OWNER mode of a row band (multi-GPU; no counterpart in the reference, which has no distributed layer).  Same arguments as
 * dss_render_backward on the band [row0,row1) / row_cycle, plus grad_out_full (N,S,S,C+1): the image gradient of ALL rows,
 * which every rank of a row-partitioned step holds anyway (the loss is evaluated on the gathered image).  The occupancy
 * surrogate of a (camera, point) pair -- its whole search window, over all image rows -- is then computed by the ONE rank
 * whose band contains the image row of the point's centre, instead of every rank adding the rows it owns; the blend half
 * stays with the rows that hold the fragments.  grad_pts (P,3): the COMPLETE screen-space position gradient of the pairs
 * this band owns, zeros elsewhere (summing the ranks' outputs gives dss_render_backward of the whole image up to the
 * order of the additions); grad_feat (P,C): this band's partial sums, as before.  Because a pair's position gradient is
 * complete on its owner, the non-linear per-point clip and the projection backward (dss_project_backward) can run BEFORE
 * the reduction over the ranks, which then carries the world-space (P_cloud, 3 + C) sums instead of a dense (N P_cloud, 3 + C)
 * bucket.  On the whole image (row0 = 0, row1 = S, row_cycle = 1) it is dss_render_backward. */
DSS_API int dss_render_backward_owned(const float *grad_out, const float *grad_out_full, const int32_t *idx,
                                      const float *qvalue, const float *wsum, const float *scaler,
                                      const float *points, const float *radii, const uint8_t *visible,
                                      const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P, int S,
                                      int K, int C, int row0, int row1, int row_cycle, float radii_s, float clip,
                                      float *grad_feat, float *grad_pts, float *rs_out, void *workspace,
                                      size_t workspace_bytes, void *stream);
/* dss_render_backward_owned fed by the DENSE plane of the occupancy gradient, grad_occ_full (N,S,S) -- the alpha channel of
 * the image gradient of all rows.  A rank that evaluates the loss on its own band (Trainer.calc_dr_loss, trainer.py:332-376,
 * with the per-image sums all-reduced) only has the gradient of its band; the owned windows need the alpha channel of the
 * other rows too (N S^2 4 bytes over all ranks): the ranks all-gather that channel alone and put the rows in image order
 * (dss_gather_rows), the RGB gradient never leaves its rank.  grad_out (N,rows,S,C+1) is the band's gradient as before. */
DSS_API int dss_render_backward_owned_plane(const float *grad_out, const float *grad_occ_full, const int32_t *idx,
                                            const float *qvalue, const float *wsum, const float *scaler,
                                            const float *points, const float *radii, const uint8_t *visible,
                                            const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P, int S,
                                            int K, int C, int row0, int row1, int row_cycle, float radii_s, float clip,
                                            float *grad_feat, float *grad_pts, float *rs_out, void *workspace,
                                            size_t workspace_bytes, void *stream);
/* Rows of an all-gathered exchange buffer put in image order (multi-GPU row bands; the reference assembles its image on one
 * device, renderer.py:75-78):  dst[n][r][0..row_floats) = src[row_pos[r]][n][0..row_floats)  for r < rows, n < N.  src is
 * what an all-gather of (band row, camera, col, channel) send buffers leaves behind, row_pos (rows,) int32 the position of
 * image row r in it (contiguous, padded or tile-row-cyclic bands alike), dst the dense (N, rows, row_floats) image. */
DSS_API int dss_gather_rows(const float *src, const int32_t *row_pos, int N, int rows, int row_floats, float *dst,
                            void *stream);
/* Second stage of dss_render_backward alone (the persistent gather kernel = blend backward + occupancy surrogate + clip of
 * every visible point, rasterize_points_backward.cu:30-212): runs on the workspace (visible lists, alpha plane, rs) and the
 * zero-filled gradients that a preceding dss_render_backward call with the SAME arguments left behind; same result.
 * Exists so that the kernel can be timed on its own (bench.py roofline_other). */
DSS_API int dss_render_backward_gather(const float *grad_out, const int32_t *idx, const float *qvalue,
                                       const float *wsum, const float *scaler, const float *points,
                                       const float *radii, const uint8_t *visible, const int64_t *first_idx,
                                       const int64_t *num_pts, int N, int64_t P, int S, int K, int C,
                                       int row0, int row1, int row_cycle, float radii_s, float clip, float *grad_feat,
                                       float *grad_pts, float *rs_out, const float *world, const float *M,
                                       void *workspace, size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Fused per-point setup = everything SurfaceSplatting.forward does before _C.splat_points:
 *   filter_renderable (rasterizer.py:219-254: view-z in [znear,zfar], optional back-face cull),
 *   pytorch3d PointsRasterizer.transform (rasterizer.py:614: NDC x,y + view-space z),
 *   _compute_WJk (:443-496), Vrk = h (I - n n^T) (:293-402), _compute_variance_and_detMk (:404-441),
 *   _get_per_point_info / _get_ellipse_axis_aligned_radius (:525-565, :498-523).
 *   world, normals (Pw,3); h_point (Pw,) or h_cloud (N,) (one of them NULL); BOTH non-NULL: h_point is (P,), one value per
 *   PACKED point, h_cloud is not read (a shared cloud whose cameras cull differently: the reference evaluates the isotropic
 *   scale on each camera's filtered cloud, rasterizer.py:344-402);
 *   anisotropic mode (Vrk_invariant = Vrk_isotropic = False, rasterizer.py:256-291): vr6 (Pw,6) = Vrk per point
 *   (xx,xy,xz,yy,yz,zz) and frame_normals (Pw,3) = normal of the PCA frame, both from dss_local_frames; h is then
 *   unused and may be NULL;
 *   M, V (N,4,4): cameras.get_full_projection_transform().get_matrix() and
 *   get_world_to_view_transform().get_matrix() (row-vector convention, p_h @ M);
 *   shared_cloud=1: one cloud of Pw points is rendered by all N cameras (Pointclouds.extend(N),
 *   rasterizer.py:236-240) and packed point p reads world[p - first_idx[n]]; else Pw == P.
 * Outputs (P = packed points): pts_screen (P,3), ellipse (P,3), radii (P,2), scaler (P,),
 * cutoff (P,), valid uint8 (P,).  Culled points are NOT compacted away: they get valid=0 and
 * z=-1 (ignored by every kernel), so no host round trip is needed for the new sizes.
 * ------------------------------------------------------------------------------------------- */
DSS_API int dss_point_setup(const float *world, const float *normals, const float *h_point,
                            const float *h_cloud, const float *vr6 /* (Pw,6) or NULL */,
                            const float *frame_normals /* (Pw,3) or NULL */,
                            const float *M, const float *V, const float *znear,
                            const float *zfar, const int64_t *first_idx, const int64_t *num_pts,
                            int N, int64_t P, int shared_cloud, int backface_culling, int S,
                            float cutoff_threshold, float antialiasing_sigma,
                            float *pts_screen, float *ellipse, float *radii, float *scaler,
                            float *cutoff, uint8_t *valid, void *stream);

/* Backward of the projection (autograd of pytorch3d's transform, rasterizer.py:614):
 * grad_world[i] = sum over the cameras that see world point i of J^T grad_screen, J = d(ndc_x,
 * ndc_y, view_z)/d(world).  grad_world (Pw,3) is fully written; deterministic. */
/* PCA frames of the K-neighbourhoods for the anisotropic source variance (rasterizer.py:256-291 +
 * utils/mathHelper.py:34-92 estimate_pointcloud_local_coord_frames, neighborhood_size = 8): knn_idx (P,K) are the
 * cloud-local ids of dss_knn_points (self included); covariance of the K points about their mean, eigen-
 * decomposition (the reference: batched SVD of the centred (K,3) matrix, curvature = sigma^2 / K).  Outputs:
 * vr6 (P,6) = F diag(c1,c2) F^T with F the two principal tangent directions (= C - c0 e0 e0^T), frame_normals
 * (P,3) = e0, curvature (P,3) ascending (may be NULL). */
DSS_API int dss_local_frames(const float *points /* (P,3) */, const int64_t *knn_idx /* (P,K) */,
                             const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P, int K,
                             float *vr6, float *frame_normals, float *curvature, void *stream);
DSS_API int dss_project_backward(const float *world, const float *M, const float *V,
                                 const int64_t *first_idx, const int64_t *num_pts, int N, int64_t Pw,
                                 int shared_cloud, const float *grad_screen /* (P,3) */,
                                 const uint8_t *valid /* (P,) */,
                                 float clip /* > 0: apply the per-point norm clip of dss_clip_grad first */,
                                 float *grad_world /* (Pw,3) */, void *stream);

/* dss_project_backward that also reduces the per-camera feature gradients of a cloud shared by the N cameras
 * (Pointclouds.extend(N), rasterizer.py:236-240: the renderer's grad_feat is (N Pw, C), the model's colours are (Pw, C)):
 * grad_feat_world[i][ch] = sum over the cameras n of grad_feat[first_idx[n] + i][ch], in camera order -- what autograd does
 * for the extended cloud's features, here in the launch that already visits every (camera, point).  1 <= C <= 8;
 * grad_feat == NULL: exactly dss_project_backward. */
DSS_API int dss_project_backward_features(const float *world, const float *M, const float *V,
                                          const int64_t *first_idx, const int64_t *num_pts, int N, int64_t Pw,
                                          int shared_cloud, const float *grad_screen, const uint8_t *valid, float clip,
                                          float *grad_world, const float *grad_feat /* (P,C) */, int C,
                                          float *grad_feat_world /* (Pw,C) */, void *stream);

/* ---------------------------------------------------------------------------------------------
 * kNN statistic behind the source-space variance scale h (rasterizer.py:310-326, 366-388):
 * kth_sqdist[p] = K-th smallest squared distance from point p to the points of its own cloud,
 * the point itself included (= max over knn_points(..., K)[:, :, 1:] for K=7).  Exact uniform-grid
 * search; replaces the third-party CUDA calls frnn.frnn_grid_points / pytorch3d.ops.knn_points.
 * 1 <= K <= 16.  Clouds with fewer than K points report their farthest point.
 * dss_cloud_mean_clamp: out[n] = clamp(mean_i(values[i]*scale), lo, hi) per cloud, `fallback` for
 * clouds with fewer than min_points points (rasterizer.py:322-326: scale 0.5, [5e-5,1e-3], 1e-3*0.5
 * when the cloud has < 7 points); deterministic summation order.
 * ------------------------------------------------------------------------------------------- */
DSS_API size_t dss_knn_workspace(int N, int64_t P);
DSS_API int dss_knn_kth_sqdist(const float *points /* (P,3) */, const int64_t *first_idx,
                               const int64_t *num_pts, int N, int64_t P, int K,
                               float *kth_sqdist /* (P,) */, void *workspace, size_t workspace_bytes,
                               void *stream);
/* Full neighbour lists of the same search (the self query pytorch3d.ops.knn_points(p, p, lengths,
 * lengths, K) of the regularisers, losses.py:157-180): dists (P,K) squared distances ascending in (distance, id),
 * idx (P,K) int64 cloud-local ids; the point itself is entry 0; clouds with fewer than K points are zero-padded
 * like pytorch3d's padded result.  1 <= K <= 40.  Same workspace as dss_knn_kth_sqdist. */
/* dss_knn_kth_sqdist with the fixed-radius semantics of the reference's DEFAULT neighbour search: SurfaceSplatting is
 * constructed with frnn_radius = 0.2 (rasterizer.py:110) and calls frnn.frnn_grid_points(K = 7, r = frnn_radius) (:317, :373),
 * which reports a neighbour beyond r as -1 [third party lxxue/FRNN, not vendored]; `0.5 * sq_dist[:, :, 1:].max(-1)` (:320-324)
 * is then the farthest neighbour FOUND within r, or -0.5 for a point that has none.  kth_sqdist[p] = the largest of the K - 1
 * nearest non-self squared distances that is <= radius^2, -1 if there is none.  radius <= 0: exactly dss_knn_kth_sqdist (the
 * reference's pytorch3d knn_points branch, frnn_radius <= 0). */
DSS_API int dss_knn_kth_sqdist_radius(const float *points, const int64_t *first_idx, const int64_t *num_pts, int N,
                                      int64_t P, int K, float radius, float *kth_sqdist, void *workspace,
                                      size_t workspace_bytes, void *stream);
DSS_API int dss_knn_points(const float *points /* (P,3) */, const int64_t *first_idx, const int64_t *num_pts,
                           int N, int64_t P, int K, float *dists /* (P,K) */, int64_t *idx /* (P,K) */,
                           void *workspace, size_t workspace_bytes, void *stream);
DSS_API int dss_cloud_mean_clamp(const float *values /* (P,) */, const int64_t *first_idx,
                                 const int64_t *num_pts, int N, float scale, float lo, float hi,
                                 float fallback, int min_points, float *out /* (N,) */, void *stream);
/* dss_cloud_mean_clamp under the reference's depth culling (rasterizer.py:236-240: the cloud is extended to the N cameras;
 * :183-217: every camera drops the points outside [znear, zfar]; :320-326: h_n = clamp(mean over the PADDED cloud n) -- the
 * mean of `h_k.mean(dim=1)` runs over P_max = the largest kept count of the batch, padding contributing zeros).  For the
 * masked (not compacted) representation: out[n] = clamp(sum over the points camera n keeps of values[p] * scale / max_m
 * kept_m, lo, hi); `fallback` for a camera that keeps fewer than min_points points.  values (Pw,) per WORLD point (the
 * K-th-neighbour distances), world (Pw,3), V (N,4,4); shared_cloud = 1: one cloud of num_pts[0] points for all cameras, else
 * camera n sees world points [first_idx[n], first_idx[n] + num_pts[n]).  values_cam_stride: 0 = one value per world point
 * (distances searched in the whole cloud: first order only); Pw = values (N, Pw) per (camera, point) from
 * dss_knn_kth_sqdist_view (the reference's order, exact).  workspace: 512 N bytes. */
DSS_API int dss_renderable_mean_clamp(const float *values, const float *world, const float *V, const float *znear,
                                      const float *zfar, const int64_t *first_idx, const int64_t *num_pts, int N,
                                      int shared_cloud, float scale, float lo, float hi, float fallback, int min_points,
                                      int64_t values_cam_stride, float *out, void *workspace, size_t workspace_bytes,
                                      void *stream);
/* dss_knn_kth_sqdist[_radius] in the reference's ORDER under depth culling: filter_renderable extends the cloud to the
 * cameras and drops, per camera, the points outside [znear, zfar] BEFORE the neighbour search (rasterizer.py:599, 236-240,
 * 183-217, 310-326), so a point's neighbours are the ones the same camera keeps.  shared_cloud = 1: ONE cloud (N == 1) seen
 * by n_cams cameras, kth_sqdist (n_cams, P): row c = the statistic among the points camera c keeps (0 for the ones it drops);
 * shared_cloud = 0: cloud n is seen by camera n (n_cams == N), kth_sqdist (P,).  K <= 8; radius as in
 * dss_knn_kth_sqdist_radius.  Feed the result to dss_renderable_mean_clamp with values_cam_stride = P (shared) or 0. */
DSS_API int dss_knn_kth_sqdist_view(const float *points, const int64_t *first_idx, const int64_t *num_pts, int N,
                                    int64_t P, int K, float radius, const float *V, const float *znear, const float *zfar,
                                    int n_cams, int shared_cloud, float *kth_sqdist, void *workspace,
                                    size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Phong shading of the points = LightingTexture.forward (DSS/core/texture.py:65-125):
 * apply_lighting (:26-63) with lighting.py:10-77 (diffuse) and :80-172 (specular) for L PointLights
 * (point_lights = 1: light_vec = location, direction = location - point, lighting.py:239-302) or
 * DirectionalLights (point_lights = 0: light_vec = direction, :175-236) per cloud:
 *   out[p] = rgb[p] * (ambient[n] + sum_l diffuse_color[n,l] relu(n^.d^))
 *            + sum_l specular_color[n,l] (relu(v^.(-d^ + 2 (n^.d^) n^)) [n^.d^ > 0])^shininess
 * with n^, d^, v^ = normalize(normal / direction / camera - point) (F.normalize, eps 1e-6).
 *   world, normals (Pw,3); rgb, out (P,3) packed per (camera, point); shared_cloud as in dss_point_setup;
 *   ambient (N,3) (already summed over lights), diffuse_color / specular_color / light_vec (N,L,3),
 *   cam_center (N,3) = cameras.get_camera_center().
 * Backward: grad_world / grad_normals (Pw,3) (summed over the cameras of a shared cloud in camera order) and
 * grad_rgb (P,3); any of the three may be NULL.  This is the path by which an RGB loss reaches the normals.
 * ------------------------------------------------------------------------------------------- */
DSS_API int dss_phong_forward(const float *world, const float *normals, const float *rgb,
                              const int64_t *first_idx, const int64_t *num_pts, int N, int64_t Pw,
                              int shared_cloud, const float *ambient, const float *diffuse_color,
                              const float *specular_color, const float *light_vec, int L, int point_lights,
                              const float *cam_center, float shininess, float *out, void *stream);
DSS_API int dss_phong_backward(const float *grad_out, const float *world, const float *normals,
                               const float *rgb, const int64_t *first_idx, const int64_t *num_pts, int N,
                               int64_t Pw, int shared_cloud, const float *ambient,
                               const float *diffuse_color, const float *specular_color,
                               const float *light_vec, int L, int point_lights, const float *cam_center,
                               float shininess, float *grad_world, float *grad_normals, float *grad_rgb,
                               void *stream);

/* ---------------------------------------------------------------------------------------------
 * Point-cloud regularisers of the training iteration (the "both regularisers" of SURVEY 8f rank 2): ProjectionLoss
 * and RepulsionLoss, DSS/training/losses.py:145-459, built by the Trainer with knn_k = 12 (trainer.py:134-137) and
 * evaluated right after the render (trainer.py:312-330; configs/dss.yml:30 lambda_dr_proj = 0.01).  Inputs are the
 * packed self-query neighbour lists of dss_knn_points with K = knn_k: knn_d2 (P,K) squared distances, knn_idx (P,K)
 * cloud-local ids, entry 0 = the point itself (dropped like losses.py:177-179).  2 <= K <= 40.
 *   dss_mollify_normals  (_denoise_normals + get_phi, :181-222, :262-278): normals_out (P,3) = phi-weighted mean of
 *       the neighbours' normals; points with keep[p] != 0 (visibility & inmask of the points filter) keep theirs.
 *       keep may be NULL.  normals_out must not alias normals.
 *   dss_projection_loss  (:296-392): loss (P,) per point; visible (P,) uint8 or NULL (= all visible).
 *   dss_repulsion_loss   (:395-492): loss (P,3) per point; the spatial weight uses num_points / bbox_diag^2 *
 *       filter_scale per cloud (get_spatial_w :248-260; the reference only broadcasts this for a batch of one
 *       cloud, here every cloud uses its own box).  workspace: 24 bytes per cloud.
 * Gradients: every weight is a constant for autograd in the reference, so d loss_i / d p_i is closed form.  With
 * grad_points != NULL the call writes grad_points (P,3) = (d loss_i / d p_i)^T grad_loss_i, recomputed from the
 * inputs (nothing is saved between forward and backward); grad_loss (P,) resp. (P,3), NULL = ones.  loss and
 * grad_points may each be NULL (not both).
 * ------------------------------------------------------------------------------------------- */
DSS_API int dss_mollify_normals(const float *normals, const float *knn_d2, const int64_t *knn_idx,
                                const uint8_t *keep, const int64_t *first_idx, const int64_t *num_pts,
                                int N, int64_t P, int K, float *normals_out, void *stream);
DSS_API int dss_projection_loss(const float *points, const float *mollified, const float *knn_d2,
                                const int64_t *knn_idx, const uint8_t *visible, const int64_t *first_idx,
                                const int64_t *num_pts, int N, int64_t P, int K, float sharpness_sigma,
                                const float *grad_loss, float *loss, float *grad_points, void *stream);
DSS_API int dss_repulsion_loss(const float *points, const float *mollified, const int64_t *knn_idx,
                               const int64_t *first_idx, const int64_t *num_pts, int N, int64_t P, int K,
                               float sharpness_sigma, float filter_scale, const float *grad_loss,
                               float *loss, float *grad_points, void *workspace, size_t workspace_bytes,
                               void *stream);

/* In-mask filter of the regularisers (DSS/models/point_modeling.py:183-208 with utils/__init__.py:266-317):
 * inmask[p] = visible[p] && any over the N views of (bilinear grid_sample of the target mask at the point's projection,
 * position (-ndc_x, -ndc_y) clamped to [-1,1], reflection padding, align_corners=False) != 0.  points (P,3) world
 * positions of ONE cloud seen by all N cameras, M (N,4,4) full projection matrices (as dss_point_setup), mask (N,H,W)
 * float, visible (P,) uint8 or NULL (= all), inmask (P,) uint8 out. */
DSS_API int dss_points_inmask(const float *points, const float *M, const float *mask, const uint8_t *visible,
                              int N, int64_t P, int H, int W, uint8_t *inmask, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Image loss of the training iteration and its gradient w.r.t. the rendered image: Trainer.calc_dr_loss
 * (DSS/training/trainer.py:332-372) with the loss objects of Trainer.__init__ (:138-141) -- masked L1 on RGB (L1Loss,
 * losses.py:127-135; mask = target mask & predicted mask, mean over the pixels inside both of the channel sum, skipped
 * when empty), silhouette L1 mean + 0.01 * IoU loss (IouLoss :498-513, mean over the batch), weighted by
 * lambda_dr_rgb / lambda_dr_silhouette (configs/dss.yml:32-33).  Sits between dss_render_forward and
 * dss_render_backward of every iteration; replaces ~20 torch kernels (forward + autograd) over (N,H,W,4) images.
 *   rgba (N,H,W,4) contiguous, 16-byte aligned: the renderer's output (alpha = occupancy = mask_img_pred);
 *   target_rgb with element strides (n,h,w,c): (N,H,W,3) or the permuted (N,3,H,W) view of trainer.py:306;
 *   target_mask (N,H,W) float.
 * forward:  sums (N+1,5) double = per image [#inside, sum |rgb diff| inside, sum |mask - alpha|, intersection, union],
 *           last row = batch totals (deterministic: fixed-order block partials); losses (4) = total, weighted rgb
 *           term (loss_dr_rgb), weighted silhouette term (loss_dr_silhouette), IoU term.  No host synchronisation.
 * backward: grad_rgba (N,H,W,4) = d total / d rgba * grad_total[0] (device scalar, NULL = 1), from `sums`; with row
 *           bands on several GPUs, all-reduce `sums` between the two calls (or evaluate on the gathered image).
 * ------------------------------------------------------------------------------------------- */
DSS_API size_t dss_image_loss_workspace(int N, int H, int W);
DSS_API int dss_image_loss_forward(const float *rgba, const float *target_rgb, int64_t t_stride_n,
                                   int64_t t_stride_h, int64_t t_stride_w, int64_t t_stride_c,
                                   const float *target_mask, int N, int H, int W, float lambda_rgb,
                                   float lambda_silhouette, double *sums, float *losses, void *workspace,
                                   size_t workspace_bytes, void *stream);
DSS_API int dss_image_loss_backward(const float *rgba, const float *target_rgb, int64_t t_stride_n,
                                    int64_t t_stride_h, int64_t t_stride_w, int64_t t_stride_c,
                                    const float *target_mask, int N, int H, int W, float lambda_rgb,
                                    float lambda_silhouette, const double *sums, const float *grad_total,
                                    float *grad_rgba, void *stream);

/* Row bands on several GPUs ("loss scalars needing global sums -> tiny all-reduce"): rank g holds rows
 * [row0, row1) of every image.  dss_image_loss_band_sums: the five per-image sums of the band, sums (N,5) (first N rows
 * of an (N+1,5) buffer); the caller all-reduces them (SUM, 40 N bytes); dss_image_loss_from_sums: totals row + the
 * four losses from the reduced sums (H, W = the FULL image); dss_image_loss_band_backward: the band of the gradient
 * image.  rgba_band (N,rows,W,4) contiguous; target_rgb = the band of the target through its element strides (pointer
 * at row0); target_mask = pointer at row0 of a (N,H,W) float mask, mask_stride_n = H*W elements. */
DSS_API int dss_image_loss_band_sums(const float *rgba_band, const float *target_rgb, int64_t t_stride_n,
                                     int64_t t_stride_h, int64_t t_stride_w, int64_t t_stride_c,
                                     const float *target_mask, int64_t mask_stride_n, int N, int rows, int W,
                                     double *sums, void *workspace, size_t workspace_bytes, void *stream);
DSS_API int dss_image_loss_from_sums(double *sums /* (N+1,5) in/out */, int N, int H, int W, float lambda_rgb,
                                     float lambda_silhouette, float *losses /* (4) */, void *stream);
DSS_API int dss_image_loss_band_backward(const float *rgba_band, const float *target_rgb, int64_t t_stride_n,
                                         int64_t t_stride_h, int64_t t_stride_w, int64_t t_stride_c,
                                         const float *target_mask, int64_t mask_stride_n, int N, int rows, int W,
                                         int H, float lambda_rgb, float lambda_silhouette, const double *sums,
                                         const float *grad_total, float *grad_band, void *stream);

/* The same band loss in TWO launches per step: dss_image_loss_band_partials writes the block partials of the band,
 * (N, 64, 5) doubles (dss_image_loss_band_partials_count(N) of them; a rank without rows writes zeros); the caller
 * all-reduces THEM (SUM: 20 KB at 8 cameras -- latency-bound like the 40 N bytes of the sums); and
 * dss_image_loss_band_backward_partials adds them up in its prologue (every block in the same fixed order: the same bits on
 * every rank) and writes the band of the gradient image, the four losses (may be NULL) and the (N+1,5) sums (may be NULL).
 * rgba_stride_n / rgba_stride_h: element strides of rgba_band over (camera, band row), multiples of 4; 0, 0 = dense -- the
 * multi-GPU renderer's band lives in a (row, camera, col, channel) send buffer (dss_render_forward image_*_stride).
 * alpha_out (may be NULL) + its element strides over (camera, band row): the alpha channel of the gradient written a second
 * time, into the (row, camera, col) send buffer of the owner form's alpha-plane exchange. */
DSS_API size_t dss_image_loss_band_partials_count(int N);
DSS_API int dss_image_loss_band_partials(const float *rgba_band, const float *target_rgb, int64_t t_stride_n,
                                         int64_t t_stride_h, int64_t t_stride_w, int64_t t_stride_c,
                                         const float *target_mask, int64_t mask_stride_n, int N, int rows, int W,
                                         int64_t rgba_stride_n, int64_t rgba_stride_h, double *partials, void *stream);
DSS_API int dss_image_loss_band_backward_partials(const float *rgba_band, const float *target_rgb, int64_t t_stride_n,
                                                  int64_t t_stride_h, int64_t t_stride_w, int64_t t_stride_c,
                                                  const float *target_mask, int64_t mask_stride_n, int N, int rows,
                                                  int W, int H, float lambda_rgb, float lambda_silhouette,
                                                  const double *partials, const float *grad_total, float *grad_band,
                                                  float *losses /* (4) or NULL */, double *sums /* (N+1,5) or NULL */,
                                                  int64_t rgba_stride_n, int64_t rgba_stride_h, float *alpha_out,
                                                  int64_t alpha_stride_n, int64_t alpha_stride_h, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* DSS_HIP_H */
