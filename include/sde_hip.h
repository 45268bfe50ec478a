/* sde_hip.h -- C ABI of libsde_hip.so: the MI355X (gfx950) kernels under the depth-training hot path.
 *
 * The reference (zzzxxxttt/SimpleDepthEstimation) is pure Python on torch; it has no FFI.  This header is
 * the NEW boundary that sits UNDER the reference's Python plugin surface (build_model / registries /
 * model(batch)->dict): every entry point replaces the stock torch call(s) named in its
 * comment (file:line in the reference).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers + sizes, no torch types.  All buffers (outputs, workspaces, partial-sum
 *     slabs) are caller-owned; nothing here allocates, frees or synchronises, so every call is legal
 *     inside a hipGraph capture.  Work is enqueued on `stream` (a hipStream_t; NULL = default stream).
 *   - return value: 0 = SDE_OK, negative = error (sde_last_error() gives the message). Never throws.
 *   - images are planar NCHW fp32 exactly as the reference's batch dict holds them; network activations
 *     are NHWC (channels-last) in fp32 or bf16, selected by the `dtype` argument (SDE_F32 / SDE_BF16).
 *   - thread safety: stateless and re-entrant given distinct streams/buffers.
 */
#ifndef SDE_HIP_H
#define SDE_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* sde_stream_t; /* hipStream_t */

#define SDE_MAX_CTX 4
#define SDE_F32 0
#define SDE_BF16 1

#define SDE_RESIZE_BILINEAR_AC 0 /* F.interpolate(mode='bilinear', align_corners=True) */
#define SDE_RESIZE_NEAREST 1     /* F.interpolate(mode='nearest') */

const char* sde_last_error(void); /* message of the last failing call on this thread */
int sde_version(void);            /* ABI version, bumped on incompatible change */

/* ---------------------------------------------------------------------------------------------------
 * Geometry / photometric path (fp32)
 * ------------------------------------------------------------------------------------------------- */

/* detectron2/geometry/camera.py:L40-46 resize_img.  src [planes,H,W] -> dst [planes,h,w]. */
int sde_resize(const float* src, float* dst, int planes, int H, int W, int h, int w, int mode, sde_stream_t stream);

/* detectron2/geometry/pose_utils.py:L98-137 pose_vec2mat (+ its VJP).  vec [n,6] = (tx,ty,tz,rx,ry,rz) -> mat [n,4,4]. */
int sde_pose_vec2mat(const float* vec, float* mat, int n, sde_stream_t stream);
int sde_pose_vec2mat_bwd(const float* vec, const float* dmat, float* dvec, int n, sde_stream_t stream);

/* detectron2/geometry/camera.py:L166-202 view_synthesis (= scale_intrinsics L14 + inv_intrinsics L25 + img_to_points L125 +
 * points_to_img L141 + F.grid_sample L196), with a per-sample constant translation (pose[:, :3, 3]).
 * img [B,C,H,W], depth [B,1,H,W], K [B,3,3] UNscaled intrinsics (sx,sy are the scale_intrinsics factors), pose [B,4,4].
 * Outputs (any of Z/grid/valid/fx/fy may be NULL): sampled [B,C,H,W], Z [B,1,H,W] (clamped 1e-5), grid [B,H,W,2] normalised,
 * valid [B,1,H,W] u8, fx/fy [B,H,W] int32 = floor of the un-normalised sample coordinate (bit-exact with the reference). */
int sde_view_synthesis(const float* img, const float* depth, const float* K, const float* pose, float sx, float sy, int B, int C, int H,
                       int W, float* sampled, float* Z, float* grid, uint8_t* valid, int32_t* fx, int32_t* fy, sde_stream_t stream);

/* One scale of the MonoDepth2 photometric loss, detectron2/modeling/meta_arch/MonoDepth2.py:L78-101,L116-124,L130-151
 * (rgb_consistency_loss for every context, warped + identity/auto-mask, SSIM ssim_loss.py:L34-53, min or mean reduce). */
typedef struct sde_photo_desc {
    const float* A;                /* target frame at this scale   [B,3,h,w] */
    const float* ctx[SDE_MAX_CTX]; /* context frames at this scale [B,3,h,w] */
    const float* pose[SDE_MAX_CTX];/* target->context poses        [B,4,4]   */
    const float* depth;            /* predicted depth              [B,1,h,w] */
    const float* K;                /* full-resolution intrinsics   [B,3,3]   */
    int32_t B, h, w, nctx;
    int32_t automask;              /* LOSS.AUTOMASK: add the un-warped (identity) map of every context */
    int32_t reduce_mean;           /* LOSS.PHOTOMETRIC_REDUCE: 0 = 'min', 1 = 'mean' */
    float sx, sy;                  /* scale_intrinsics factors w/W, h/H */
    float ssim_w, C1, C2;          /* LOSS.SSIM_WEIGHT, LOSS.C1, LOSS.C2 */
} sde_photo_desc;

/* number of workgroups (= length of `partial`, and of pose_partial / (nctx*12) for backward) */
int sde_photo_num_blocks(int B, int h, int w, int backward);

/* Forward.  sampled[j] [B,3,h,w] (saved for backward), sel [B,h,w] u8 arg-min map index (order: warp0, id0, warp1, id1 ...),
 * maps (optional, may be NULL) [B,nmaps,h,w] the individual photometric maps, partial [sde_photo_num_blocks(...,0)] workspace.
 * loss_out[0] (+)= loss_scale * mean_over_pixels(reduced map). */
int sde_photo_fwd(const sde_photo_desc* d, float* const* sampled, uint8_t* sel, float* maps, float* partial, float* loss_out,
                  float loss_scale, int accumulate, sde_stream_t stream);

/* Backward w.r.t. depth and poses.  gout: device scalar (upstream gradient), multiplied by gscale.
 * d_depth [B,1,h,w]; pose_partial [sde_photo_num_blocks(...,1) * nctx * 12] workspace; d_pose[j] [B,4,4]. */
int sde_photo_bwd(const sde_photo_desc* d, const float* const* sampled, const uint8_t* sel, const float* gout, float gscale, float* d_depth,
                  int accumulate_depth, float* pose_partial, float* const* d_pose, int accumulate_pose, sde_stream_t stream);

/* detectron2/modeling/losses/smoothness_loss.py:L42-80.  depth [B,1,h,w], img [B,3,h,w].
 * mean_part [B*32], dn [B,h,w], loss_part/s_part [sde_smooth_num_blocks] are caller workspaces that backward re-reads.
 * loss_out[0] (+)= loss_scale * smoothness_loss. */
int sde_smooth_num_blocks(int B, int h, int w);
int sde_smooth_fwd(const float* depth, const float* img, int B, int h, int w, float* mean_part, float* dn, float* loss_part, float* s_part,
                   float* loss_out, float loss_scale, int accumulate, sde_stream_t stream);
int sde_smooth_bwd(const float* depth, const float* dn, const float* mean_part, const float* s_part, const float* gout, float gscale, int B,
                   int h, int w, float* d_depth, int accumulate, sde_stream_t stream);

/* detectron2/modeling/losses/losses.py:L5-13 silog_loss against resize_img(depth_gt, 'nearest') (Supervised.py:L44-45), fused:
 * est [B,1,h,w], gt [B,1,H,W] full resolution (nearest-sampled in-kernel), mask gt > 1.  No compaction, no host sync.
 * part [sde_silog_num_blocks*3] workspace; stats[4] = (count, E[d], E[d^2], loss). */
int sde_silog_num_blocks(int B, int h, int w);
int sde_silog_fwd(const float* est, const float* gt, int B, int h, int w, int H, int W, float variance_focus, float* part, float* stats,
                  sde_stream_t stream);
int sde_silog_bwd(const float* est, const float* gt, const float* stats, const float* gout, float gscale, float variance_focus, int B, int h,
                  int w, int H, int W, float* d_est, int accumulate, sde_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SDE_HIP_H */
