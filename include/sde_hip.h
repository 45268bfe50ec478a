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
 *     are NHWC (channels-last) in fp32, bf16 or fp16, selected by the `dtype` argument (SDE_F32 / SDE_BF16 / SDE_F16).
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
#define SDE_F16 2 /* IEEE half storage, fp32 accumulate: BASELINE.json configs[4] (needs loss scaling: sde_adam_step's scale_state) */

/* Partial-sum slabs (BatchNorm statistics, BN/bias backward reductions) must be allocated with SDE_REDUCE_ROWS extra rows:
 * slabs taller than that are first folded into their own tail by a deterministic two-level reduction. */
#define SDE_REDUCE_ROWS 32

#define SDE_RESIZE_BILINEAR_AC 0 /* F.interpolate(mode='bilinear', align_corners=True) */
#define SDE_RESIZE_NEAREST 1     /* F.interpolate(mode='nearest') */

const char* sde_last_error(void); /* message of the last failing call on this thread */
int sde_version(void);            /* ABI version, bumped on incompatible change */

/* Diagnostics: store the device's constant-rate wall clock (ticks of sde_wall_clock_khz()) into *slot, in stream order.  Captured into the step's
 * hipGraph at chosen points, the markers give the replayed step's real timeline (bench.py --marks).  Not part of the reference's surface. */
int sde_mark_time(uint64_t* slot, sde_stream_t stream);
int sde_wall_clock_khz(void); /* < 0: query failed */
/* Stream-ordered flag for the host: *dst = value (dst: host-pinned, device-visible memory; system-scope release store by one thread).  The device
 * prefetcher's hand-over between its copy stream and the training stream (no event record / wait pairs). */
int sde_store_u64(uint64_t* dst, uint64_t value, sde_stream_t stream);

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
    const float* clip_thr;         /* LOSS.CLIP > 0: device [nmaps] thresholds mean + clip*std of each UNCLIPPED map (MonoDepth2.py:L147-149:
                                      one forward with maps != NULL and clip_thr == NULL yields them); NULL = no clipping */
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
/* Every scale of the loss (MonoDepth2.py:L78-112 loops over the four decoder scales) in ONE launch per phase: d [n] descriptors (n <= 4, same batch and
 * contexts, no clip thresholds), fine scale first; sampled [n][SDE_MAX_CTX], sel [n], partial [n] (sizes as for the single-scale calls);
 * loss_out [n] = mean over pixels of each scale's reduced map -- bit-identical to n calls of sde_photo_fwd.  Backward: gout [n] device scalars (the
 * upstream gradient of each scale's loss), d_depth [n], pose_partial [n]; d_pose [SDE_MAX_CTX] receives the SUM over the scales (written, not
 * accumulated).  The coarse scales (1/4 ... 1/64 of the pixels) are launch-sized on their own; behind the fine scale's workgroups they fill its tail. */
int sde_photo_multi_fwd(const sde_photo_desc* d, int n, float* const* sampled, uint8_t* const* sel, float* const* partial, float* loss_out,
                        sde_stream_t stream);
int sde_photo_multi_bwd(const sde_photo_desc* d, int n, const float* const* sampled, const uint8_t* const* sel, const float* gout, float* const* d_depth,
                        float* const* pose_partial, float* const* d_pose, sde_stream_t stream);
/* d_pose == NULL in sde_photo_multi_bwd leaves the per-workgroup pose partials unsummed; this call sums them (same order, same result).  Two calls so
 * that the caller can enqueue the pose side on the stream PoseNet's backward runs on (it is the only consumer), behind an event of its own. */
int sde_photo_multi_pose_finalize(const sde_photo_desc* d, int n, const float* const* pose_partial, float* const* d_pose, sde_stream_t stream);

/* The photometric AND the edge-aware smoothness term of every scale of MonoDepth2's loss loop (detectron2/modeling/meta_arch/MonoDepth2.py:L78-126,
 * modeling/losses/smoothness_loss.py:L42-80) behind one call per pass.  d [n] as for sde_photo_multi_fwd (d[s].A doubles as the smoothness term's image,
 * d[s].depth as its depth).  photo_w / smooth_w: host arrays [n] of the weights the reference multiplies each scale's term with (smooth_w NULL: no
 * smoothness term, its buffers may be NULL).  Buffers per scale: sampled / sel / photo_partial as for sde_photo_multi_fwd; sm_mean_part [B*32], sm_dn [B*h*w],
 * sm_loss_part / sm_s_part [sde_smooth_num_blocks].  per_scale [2n] receives every scale's photometric mean, then every scale's smoothness value
 * (bit-identical to sde_photo_fwd / sde_smooth_fwd); totals [2] = the two weighted sums, accumulated in scale order.  ticket: one device int, zero on
 * entry, left zero.  Backward: g_rec / g_smooth = device scalars, the upstream gradients of totals[0] / totals[1] (g_smooth NULL: photometric part only);
 * d_depth [n] is written (photometric) and then accumulated (smoothness); pose side as in sde_photo_multi_bwd. */
int sde_mono_loss_fwd(const sde_photo_desc* d, int n, const float* photo_w, const float* smooth_w, float* const* sampled, uint8_t* const* sel,
                      float* const* photo_partial, float* const* sm_mean_part, float* const* sm_dn, float* const* sm_loss_part, float* const* sm_s_part,
                      float* per_scale, float* totals, int* ticket, sde_stream_t stream);
int sde_mono_loss_bwd(const sde_photo_desc* d, int n, const float* photo_w, const float* smooth_w, const float* const* sampled, const uint8_t* const* sel,
                      const float* g_rec, const float* g_smooth, const float* const* sm_mean_part, const float* const* sm_dn, const float* const* sm_s_part,
                      float* const* d_depth, float* const* pose_partial, float* const* d_pose, sde_stream_t stream);
/* The smoothness half of sde_mono_loss_bwd on its own (a caller that forks the pose side to another stream between the two halves): d_depth[s] (+)= ... */
int sde_smooth_multi_bwd(const sde_photo_desc* d, int n, const float* smooth_w, const float* g_smooth, const float* const* sm_mean_part, const float* const* sm_dn,
                         const float* const* sm_s_part, float* const* d_depth, int accumulate, sde_stream_t stream);

/* Stand-alone SSIM distance map, the callable module of detectron2/modeling/losses/ssim_loss.py:L6-53:
 * out[b,c,h,w] = clamp((1 - SSIM(x, y)) / 2, 0, 1) with ReflectionPad2d(1) + 3x3 mean; x, y, out planar [B,C,H,W] fp32 (the training path
 * evaluates SSIM inside sde_photo_fwd / sde_photo_bwd).  Backward: dx and/or dy (either may be NULL) for the upstream gradient gout [B,C,H,W];
 * coef_ws: [B*C*H*W][4] floats of workspace, 16-byte aligned. */
int sde_ssim_fwd(const float* x, const float* y, int B, int C, int H, int W, float C1, float C2, float* out, sde_stream_t stream);
int sde_ssim_bwd(const float* x, const float* y, const float* gout, int B, int C, int H, int W, float C1, float C2, float* coef_ws, float* dx, float* dy,
                 sde_stream_t stream);

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
/* The same loss summed over up to SDE_SILOG_MAX_SCALES prediction scales against ONE ground truth (Supervised.py:L42-47: the four decoder scales,
 * weight 1/4 each) in one launch per phase: total[0] = sum_k weight[k] * SILog(est[k], nearest(gt)), stats [n][4] as above per scale (bit-identical to
 * the single-scale call), part: [sde_silog_multi_num_blocks][3].  Backward writes d_est[k] = gout * gscale * weight[k] * dSILog_k/d est[k]. */
#define SDE_SILOG_MAX_SCALES 4
int sde_silog_multi_num_blocks(int B, const int* h, const int* w, int n);
int sde_silog_multi_fwd(const float* const* est, const float* gt, int B, const int* h, const int* w, const float* weight, int n, int H, int W,
                        float variance_focus, float* part, float* stats, float* total, sde_stream_t stream);
int sde_silog_multi_bwd(const float* const* est, const float* gt, const float* stats, const float* gout, float gscale, float variance_focus, int B,
                        const int* h, const int* w, const float* weight, int n, int H, int W, float* const* d_est, sde_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Convolution engine (NHWC activations, fp32 or bf16 storage, fp32 accumulate)
 * ------------------------------------------------------------------------------------------------- */
#define SDE_SRC_PLAIN 0   /* input tensor as stored */
#define SDE_SRC_UPCAT 1   /* cat(F.interpolate(x0, scale 2, nearest), x1) -- depth_decoder.py:L102-105 -- never materialised */
#define SDE_SRC_ZEROINS 2 /* x0 with zeros inserted between pixels: data-gradient of a stride-2 convolution */
#define SDE_ACT_NONE 0
#define SDE_ACT_ELU 1
#define SDE_ACT_RELU 2

/* Input side of a convolution.  The virtual input is [Bn, IH, IW, C0+C1]; channel counts are multiples of 16 bytes. */
typedef struct sde_conv_desc {
    const void* x0; /* [Bn, H0, W0, C0] */
    const void* x1; /* [Bn, IH, IW, C1] (SDE_SRC_UPCAT only) or NULL */
    int32_t dtype;  /* SDE_F32 / SDE_BF16 (storage type of x0, x1, weights and outputs) */
    int32_t C0, C1, H0, W0, IH, IW, src_mode;
    int32_t KH, KW, stride, pad;
    int32_t reflect; /* 1: ReflectionPad2d(pad) (depth_decoder.py:L40-43), 0: zero padding */
    int32_t Bn, OH, OW;
} sde_conv_desc;

/* master fp32 OIHW weight [Cout,Cin,KH,KW] -> K-major operand in `dtype`:
 *   for_dgrad = 0: [Cout_pad][KH][KW][Cin_pad]            (forward and weight-gradient layout)
 *   for_dgrad = 1: [Cin_pad][KH][KW][Cout_pad], taps flipped (data-gradient operand) */
int sde_pack_weight(const float* w, void* out, int dtype, int Cout, int Cin, int KH, int KW, int Cin_pad, int Cout_pad, int for_dgrad,
                    sde_stream_t stream);

/* Both operands of every layer of a model in ONE launch (the weights change once per optimizer step).  items: DEVICE array of n layers,
 * built once; a workgroup transposes one (32 output channels x sde-chosen input-channel block x all taps) tile through LDS;
 * `end` = exclusive prefix sum of sde_pack_item_blocks() over the layers, total_blocks = items[n-1].end. */
typedef struct sde_pack_item {
    const float* src; /* master fp32 weights: [Cout][Cin][KH][KW] (src_layout SDE_W_OIHW) or [Cout][KH][KW][Cin] (SDE_W_OHWI, channels-last) */
    void* dst_fwd;    /* [Cout_pad][KH][KW][Cin_pad] in `dtype` (or NULL) */
    void* dst_dgrad;  /* [Cin_pad][KH][KW][Cout_pad], taps flipped (or NULL) */
    int32_t Cout, Cin, KH, KW, Cin_pad, Cout_pad;
    int32_t src_layout, reserved;
    int64_t end;
} sde_pack_item;
#define SDE_W_OIHW 0 /* torch's default memory order of a conv weight */
#define SDE_W_OHWI 1 /* torch.channels_last order of the same [Cout,Cin,KH,KW] tensor: K-major like the packed operands and the gradient slabs */
int sde_pack_item_blocks(int Cout_pad, int Cin_pad, int KH, int KW);
int sde_pack_weights_batched(const sde_pack_item* items_dev, int n, long total_blocks, int dtype, sde_stream_t stream);

/* y[Bn,OH,OW,ldy] = act(conv(virtual input, w_packed) + bias); channels >= Cout of y are written as zeros.
 * Replaces nn.Conv2d (+ReflectionPad2d, +upsample/cat, +nn.ELU) forward -- resnet_encoder.py:L91-97, depth_decoder.py:L21-53,
 * PoseNet.py:L13-16 -- and, with the flipped operand of sde_pack_weight(for_dgrad=1), their data-gradient.
 * stats (optional): [sde_conv_fwd_tiles_m + SDE_REDUCE_ROWS][Cout][2] per-tile (sum, sum of squares) of the stored outputs, for BatchNorm. */
int sde_conv_fwd(const sde_conv_desc* d, const void* w_packed, const float* bias, int act, void* y, int Cout, int ldy, float* stats,
                 sde_stream_t stream);
/* The same with a caller-owned fp32 workspace of sde_conv_fwd_ws_bytes() bytes (0: not needed): small-M, long-K layers then run split-K
 * (K cut into up to 8 ranges, partial tiles summed in a fixed order by a finishing kernel that also applies bias / activation / statistics). */
size_t sde_conv_fwd_ws_bytes(const sde_conv_desc* d, int ldy);
int sde_conv_fwd_ws(const sde_conv_desc* d, const void* w_packed, const float* bias, int act, void* y, int Cout, int ldy, float* stats, void* ws,
                    size_t ws_bytes, sde_stream_t stream);
/* Data gradient of a stride-1 convolution whose INPUT was relu(BatchNorm(y_bn)) with no residual and no other consumer (torchvision Bottleneck:
 * conv2 <- bn1, conv3 <- bn2; BasicBlock: conv2 <- bn1 -- detectron2/layers/resnet_encoder.py:L88-99 wraps those blocks): the GEMM is the one
 * sde_conv_fwd runs for the same descriptor (d = the dgrad view: source dz, flipped operand), and its epilogue takes over the first pass of
 * BatchNorm's backward: gm [M][ldy] = (fma(y_bn, scale, shift) > 0) * g is stored instead of g, and part [rows][Cout][2] receives the
 * per-workgroup sums (sum gm, sum gm * xhat), xhat = (y_bn - mean) * rstd; bnp [4][Cout] = (mean, rstd, scale, shift) as sde_bn_finalize wrote it.
 * sde_conv_dgrad_bnbwd_rows: rows of that slab, or 0 when the dispatcher has no fused form for this layer (fp32, split-K, stride 2, channel
 * counts that are not multiples of 64, narrow layers): the caller then runs sde_conv_fwd and sde_bn_bwd.  Follow with sde_bn_bwd_from_part. */
int sde_conv_dgrad_bnbwd_rows(const sde_conv_desc* d, int Cout, int ldy);
int sde_conv_dgrad_bnbwd(const sde_conv_desc* d, const void* w_packed, void* gm, int Cout, int ldy, const void* bn_y, const float* bnp, float* part,
                         sde_stream_t stream);
/* The residual form (torchvision Bottleneck, resnet_encoder.py:L88-99: out = relu(bn3(y_bn) + identity) feeds conv1 of the next block AND that block's
 * skip path): the data gradient of that conv1 takes over the WHOLE reduce pass of bn3's backward -- gm = (g + res_grad) * (bn_out > 0), res_grad the
 * gradient arriving over the skip path, bn_out the block output (the mask), and the same partial slab.  Persistent GEMM and LDS-halo 3x3 kernel
 * (BasicBlock: bn2 -> conv1 of the next block); rows = 0 where neither runs the layer. */
int sde_conv_dgrad_bnbwd_res_rows(const sde_conv_desc* d, int Cout, int ldy);
int sde_conv_dgrad_bnbwd_res(const sde_conv_desc* d, const void* w_packed, void* gm, int Cout, int ldy, const void* bn_y, const float* bnp, float* part,
                             const void* res_grad, const void* bn_out, sde_stream_t stream);
int sde_conv_fwd_tiles_m(const sde_conv_desc* d, int ldy);
int sde_conv_fwd_variant(const sde_conv_desc* d, int ldy);
/* Tuning / test knob: the LDS-halo 3x3 kernel is used when a launch has at least this many workgroups (default 192; 0 = whenever it
 * applies, negative = never).  Returns the previous value.  Results do not depend on it beyond fp32 summation order. */
int sde_conv_set_halo_min_blocks(int min_blocks);
/* Dispatcher options (process-wide; A/B measurements and tests): returns the previous value, negative on a bad key / value.  Results do
 * not depend on them beyond fp32 summation order.
 *   SDE_OPT_PGEMM        1 (default): layers with 64-channel-multiple inputs run on the persistent LDS-DMA GEMM (csrc/pgemm.hip); 0: never
 *   SDE_OPT_PGEMM_DEPTH  stages of its LDS ring: 3 (default) or 4
 *   SDE_OPT_PGEMM_3X3    1: it also takes the 3x3 stride-1 layers the LDS-halo kernel would get (default 0) */
#define SDE_OPT_PGEMM 1
#define SDE_OPT_PGEMM_DEPTH 2
#define SDE_OPT_PGEMM_3X3 3
#define SDE_OPT_PGEMM_TILE 4   /* force its tile: 64064, 128064, 128128; 0 (default) = chosen per layer */
#define SDE_OPT_SPLITK 5       /* 1 (default): small-M, long-K layers cut K into up to 8 ranges (sde_conv_fwd_ws); 0: never */
#define SDE_OPT_WGRAD_BLOCKS 6 /* workgroup target of the weight-gradient GEMM's pixel splits (default 256 = one per CU) */
#define SDE_OPT_WGRAD_HALO 7   /* 1 (default): the 3x3 stride-1 layers with <= 96 input and <= 32 output channels take the LDS-halo weight-gradient
                                  kernel (csrc/wgrad_halo.hip); 0: the generic kernel */
#define SDE_OPT_CONV_SMALL 8   /* 1 (default): 3x3 stride-1 layers with 8 / 16 / 32 input channels and >= 16 K output pixels take the narrow-input halo
                                  kernel (csrc/conv_halo_small.hip) in the forward and data-gradient passes; 0: the generic kernel */
#define SDE_OPT_WGRAD_DMA 9    /* 1 (default): layers with 64-channel-multiple inputs and outputs (zero padding, no concat) take the LDS-DMA weight-gradient
                                  kernel (csrc/wgrad_dma.hip); 0: the register-staged kernel */
#define SDE_OPT_BNBWD_FUSE 10  /* 1 (default): sde_conv_dgrad_bnbwd_rows reports the layers whose data gradient can carry BatchNorm's backward reduction
                                * in its epilogue; 0: it reports none (the separate bn_bwd_reduce pass runs everywhere: A/B, tests) */
#define SDE_OPT_CU_RESERVE 11  /* compute units the persistent kernels (pgemm, chalo, whalo) leave free: their grids are sized for (CUs - n) instead of every CU.
                                * 0 (default); a multiple of 8 <= 128.  For data-parallel runs: RCCL's channel kernels otherwise wait for a resident workgroup to end */
int sde_conv_set_option(int key, int value);

/* dW (master fp32 OIHW, [Cout,Cin_real,KH,KW]) (+)= sum over output pixels of dy^T * im2col(virtual input).
 * slab: caller workspace [splits][Cout][KH*KW*(C0+C1)] fp32, splits = sde_conv_wgrad_splits(d, Cout): one fp32 partial per pixel range,
 * summed in a fixed order (stacks taller than SDE_WGRAD_FOLD_ROWS as that many chunk sums first) -- bit-reproducible, no atomics. */
#define SDE_WGRAD_FOLD_ROWS 16
int sde_conv_wgrad_splits(const sde_conv_desc* d, int Cout);
int sde_conv_wgrad(const sde_conv_desc* d, const void* dy, int Cout, int ldd, int Cin_real, float* slab, int splits, float* dw, int accumulate,
                   sde_stream_t stream);

/* Deferred form for training loops: sde_conv_wgrad_partial runs the GEMM only (slab: [splits][Cout][KH*KW*Cin], no scratch rows);
 * sde_wgrad_reduce_batched then sums the slabs of MANY layers in one launch per <= 120 items, in the same order as sde_conv_wgrad
 * (results are bit-identical).  `items` is a HOST array: it is copied into the kernel arguments, so nothing has to outlive the call. */
int sde_conv_wgrad_partial(const sde_conv_desc* d, const void* dy, int Cout, int ldd, float* slab, int splits, sde_stream_t stream);
typedef struct sde_wreduce_item {
    const float* slab; /* the slab stack sde_conv_wgrad_partial filled (device) */
    float* dw;         /* master fp32 gradient of the [Cout,Cin_real,KH,KW] weight (device), in OIHW or OHWI memory order */
    int32_t rows /* = splits */, Cout, KHW, Cin_pad, Cin_real;
    int32_t accumulate; /* bit 0: add to dw instead of overwriting; bit 1 (SDE_WREDUCE_OHWI): dw is in OHWI (channels-last) order */
} sde_wreduce_item;
#define SDE_WREDUCE_ACCUMULATE 1
#define SDE_WREDUCE_OHWI 2
int sde_wgrad_reduce_batched(const sde_wreduce_item* items, int n, sde_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Layers around the convolutions (NHWC, `dtype` storage, fp32 math)
 * ------------------------------------------------------------------------------------------------- */
/* (img - mean)/std (Supervised.py:L39, MonoDepth2.py:L60) fused with NCHW fp32 -> NHWC `dtype`, channel padding and the optional
 * torch.flip(image,[3]) of DepthResNet.py:L52-53.  mean/std may both be NULL (PoseNet input: no normalisation). */
int sde_prep_input(const float* img, const float* mean, const float* std_, int B, int C, int H, int W, int Cpad, int flip, int dtype, void* out,
                   sde_stream_t stream);

/* Training-mode nn.BatchNorm2d of torchvision's ResNet (resnet_encoder.py:L92, blocks L94-97): batch statistics from the
 * convolution's stats slab -> bnp [4][C] = (mean, rstd, scale, shift); running stats updated in place (momentum, unbiased var). */
int sde_bn_finalize(const float* part, int tiles, int C, long count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                    float momentum, float eps, float* bnp, sde_stream_t stream);
int sde_bn_eval_params(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, int C, float* bnp,
                       sde_stream_t stream);
/* out = [relu](y*scale + shift [+ residual]) */
int sde_bn_apply(const void* y, const float* bnp, const void* residual, int relu, long M, int C, int dtype, void* out, sde_stream_t stream);
/* sde_bn_finalize + sde_bn_apply in one launch (same arguments, count = M rows), for 16-bit types, C % 64 == 0 and at most 256 slab rows
 * (sde_bn_finalize_apply_ok): every workgroup reduces the slab columns of its own 64 channels, redundantly and in a fixed order. */
int sde_bn_finalize_apply_ok(int tiles, int C, int dtype);
/* sde_bn_bwd fuses its finalize and apply passes the same way; for the big layers its reduce pass runs 1024-thread workgroups so that the partial
 * slab stays within 256 rows.  sde_bn_set_fuse(0) turns the fusion off, (2) keeps it for the short slabs only (A/B, tests).  Returns the previous value. */
int sde_bn_set_fuse(int on);
int sde_bn_finalize_apply(const float* part, int tiles, int C, long count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                          float momentum, float eps, float* bnp, const void* y, const void* residual, int relu, int dtype, void* out, sde_stream_t stream);
/* BatchNorm(+ReLU, +residual) backward.  The normalised output may have up to three consumers whose gradients arrive separately
 * (dout, dout1, dout2; the latter two may be NULL): they are summed on the fly.  gm [M,C] (workspace, required when relu is set or more than
 * one gradient is given) receives dz = relu'(out) * (sum of the gradients) -- which is also the gradient of the residual input.
 * out: the saved forward output (ReLU mask).  With relu set and out == NULL the mask is re-derived from y and bnp as fma(y, scale, shift) > 0 --
 * valid only when the forward had NO residual (then it equals out > 0, and one activation-sized stream less is read).
 * part: [sde_reduce_num_blocks(M, C) + SDE_REDUCE_ROWS][C][2] workspace, coef: [2][C] workspace.  dy [M,C]; dgamma/dbeta [C] (+)=. */
int sde_reduce_num_blocks(long M, int C);
int sde_bn_bwd(const void* dout, const void* dout1, const void* dout2, const void* out, const void* y, const float* bnp, const float* gamma, int relu,
               long M, int C, int dtype, float* part, float* coef, float* dgamma, float* dbeta, int accumulate_params, void* gm, void* dy,
               sde_stream_t stream);
/* The same backward with the reduce pass already done by the data-gradient GEMM that produced the incoming gradient (sde_conv_dgrad_bnbwd):
 * gm [M,C] = relu'(bn(y)) * g as that GEMM stored it, part [rows + SDE_REDUCE_ROWS][C][2] = its per-workgroup (sum gm, sum gm * xhat).  Runs the
 * finalize and apply passes only (one launch for rows <= 256 and C % 64 == 0, as sde_bn_bwd).  dy [M,C]; dgamma/dbeta [C] (+)=; coef [2][C]. */
int sde_bn_bwd_from_part(const float* part, int rows, const void* gm, const void* y, const float* bnp, long M, int C, int dtype, float* coef,
                         float* dgamma, float* dbeta, int accumulate_params, void* dy, sde_stream_t stream);

/* nn.MaxPool2d(3, 2, 1) (resnet_encoder.py:L94).  idx: [B,OH,OW,C] u8 arg-max saved for backward. */
int sde_maxpool_fwd(const void* x, int B, int H, int W, int C, int dtype, void* out, uint8_t* idx, sde_stream_t stream);
int sde_maxpool_bwd(const void* dout, const uint8_t* idx, int B, int H, int W, int C, int dtype, void* dx, sde_stream_t stream);
/* ... with the gradients of two consumers of the pooled tensor summed on the fly (dout1 may be NULL) */
int sde_maxpool_bwd_sum(const void* dout, const void* dout1, const uint8_t* idx, int B, int H, int W, int C, int dtype, void* dx, sde_stream_t stream);

/* dz = dout * act'(out) (nn.ELU / nn.ReLU backward) fused with the bias gradient dbias[c] (+)= sum_rows dz[:, c], c < Cbias.
 * dz and/or dbias may be NULL; part: [sde_reduce_num_blocks(M, C) + SDE_REDUCE_ROWS][C] workspace (needed when dbias != NULL). */
int sde_act_bwd_bias(const void* dout, const void* out, int act, long M, int C, int dtype, void* dz, float* part, float* dbias, int Cbias, int accumulate,
                     sde_stream_t stream);
/* ... of an activation with two consumers (a decoder level feeds its disparity head and the next level): dz = act'(out) * (dout + dout1), dout1 may be NULL */
int sde_act_bwd_bias_sum(const void* dout, const void* dout1, const void* out, int act, long M, int C, int dtype, void* dz, float* part, float* dbias, int Cbias,
                         int accumulate, sde_stream_t stream);
/* part given but dbias == NULL: only the per-workgroup column partials are written ([sde_reduce_num_blocks(M, C)][C]); this call then sums the partial
 * slabs of MANY layers in one launch (out[c] (+)= sum_rows part[r][c], c < C, every item exactly as the per-layer finalize sums it).  The per-layer
 * finalize is a launch-sized kernel on the backward pass's serial chain that nothing downstream waits for.  `items` is a HOST array (copied by value). */
typedef struct sde_colsum_item {
    const float* part; /* [rows][ld] fp32 partial slab (device) */
    float* out;        /* [C] destination (device) */
    int32_t rows, ld, C, accumulate;
} sde_colsum_item;
int sde_colsum_finalize_batched(const sde_colsum_item* items, int n, sde_stream_t stream);

/* Backward of ReflectionPad2d(1) [+ nearest x2 upsample + channel concat] (depth_decoder.py:L40-47,L102-105):
 * dxp [B,H+2,W+2,C] (gradient w.r.t. the padded virtual input, from the data-gradient GEMM) ->
 * upcat=0: dx0 [B,H,W,C];  upcat=1: dx0 [B,H/2,W/2,C0], dx1 [B,H,W,C-C0]. */
int sde_refl_fold(const void* dxp, int B, int H, int W, int C, int C0, int upcat, int dtype, void* dx0, void* dx1, sde_stream_t stream);

/* nn.Softplus + disp_to_depth(min_depth, max_depth)[1] (+ torch.flip of the output) -- depth_decoder.py:L9-18,L108, DepthResNet.py:L57-60.
 * y [B,H,W,ld] (channel 0 is the disparity logit) -> depth [B,1,H,W] fp32; backward writes dy [B,H,W,ld] (channels > 0 zero). */
int sde_depth_head_fwd(const void* y, int B, int H, int W, int ld, float min_depth, float max_depth, int flip, int dtype, float* depth, sde_stream_t stream);
int sde_depth_head_bwd(const void* y, const float* ddepth, int B, int H, int W, int ld, float min_depth, float max_depth, int flip, int dtype, void* dy,
                       sde_stream_t stream);
/* ... also producing the bias gradient of the one-channel convolution in front of the head (depth_decoder.py:L95-97 `dispconv`): dbias[0] (+)= sum of the
 * logit gradients as stored in dy.  part: [sde_depth_head_bias_blocks(B, H, W)] floats; part and dbias are given together or both NULL. */
int sde_depth_head_bias_blocks(int B, int H, int W);
int sde_depth_head_bwd_bias(const void* y, const float* ddepth, int B, int H, int W, int ld, float min_depth, float max_depth, int flip, int dtype, void* dy,
                            float* part, float* dbias, int accumulate, sde_stream_t stream);

/* nn.GroupNorm(G) + activation: relu = 0 none, 1 nn.ReLU (PoseNet.py:L13-20), 2 nn.ELU (layers01.py:L33-40).
 * part: [B][SDE_GN_CHUNKS][C][2] workspace, gnp: [B][G][2] (mean, rstd) saved for backward, coef: [B][G][2] workspace. */
#define SDE_GN_CHUNKS 64
int sde_gn_relu_fwd(const void* x, const float* gamma, const float* beta, int B, int HW, int C, int G, float eps, int relu, int dtype, float* part, float* gnp,
                    void* out, sde_stream_t stream);
int sde_gn_relu_bwd(const void* dout, const void* out, const void* x, const float* gnp, const float* gamma, int B, int HW, int C, int G, int relu, int dtype,
                    float* part, float* coef, float* dgamma, float* dbeta, int accumulate_params, void* dx, sde_stream_t stream);
/* The same with a residual input: the normalised tensor is x + res (summed in fp32, never materialised) -- layers01.py:L74-76 ResidualConv:
 * normalize(x_out + shortcut).  res may be NULL (= the calls above); with res, C / (16-byte group) must divide 256.  Backward: dx is the
 * gradient of BOTH x and res. */
int sde_gn_relu_res_fwd(const void* x, const void* res, const float* gamma, const float* beta, int B, int HW, int C, int G, float eps, int relu, int dtype,
                        float* part, float* gnp, void* out, sde_stream_t stream);
int sde_gn_relu_res_bwd(const void* dout, const void* out, const void* x, const void* res, const float* gnp, const float* gamma, int B, int HW, int C, int G, int relu,
                        int dtype, float* part, float* coef, float* dgamma, float* dbeta, int accumulate_params, void* dx, sde_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * PackNet's 3-D convolution (layers01.py:L223-298): x.unsqueeze(1) -> nn.Conv3d(1, 8, 3, padding=1) -> view(b, 8*D, h, w) on NHWC data:
 * y[b,h,w,f*D+ch] = bias[f] + sum w[f][kd][kh][kw] x[b,h+kh-1,w+kw-1,ch+kd-1].  w: [8][3][3][3] fp32 (torch's [8,1,3,3,3]), D % (16 B) == 0.
 * wgrad: part = workspace [sde_conv3d_wgrad_num_blocks()][224] floats; dw [8*27], dbias [8] (may be NULL). */
int sde_conv3d_fwd(const void* x, const float* w, const float* bias, int B, int H, int W, int D, int dtype, void* y, sde_stream_t stream);
int sde_conv3d_dgrad(const void* dy, const float* w, int B, int H, int W, int D, int dtype, void* dx, sde_stream_t stream);
int sde_conv3d_wgrad_num_blocks(int B, int H, int W, int D, int dtype);
int sde_conv3d_wgrad(const void* x, const void* dy, int B, int H, int W, int D, int dtype, float* part, float* dw, float* dbias, int accumulate,
                     sde_stream_t stream);

/* PackNet01's data movement (csrc/packnet.hip), NHWC, 16 bytes per lane, each other's backward:
 *   sde_space_to_depth: layers01.py:L138-160 `packing` (r = 2): x [B,H,W,C] -> y [B,H/2,W/2,4C], y[b,i,j,c*4+di*2+dj] = x[b,2i+di,2j+dj,c]
 *   sde_depth_to_space: nn.PixelShuffle(2) of layers01.py:L262-298:       x [B,H,W,C4] -> y [B,2H,2W,C4/4], the inverse permutation
 * sde_concat_fwd: PackNet01.py:L150-199 torch.cat([unpacked, skip(, nearest_x2(inv_depth))], 1) (version A) or [unpacked + skip(, ...)] (add = 1,
 * version B: C1 == C0) as ONE pass: out [B,H,W,Ct], Ct = channels used rounded up to the 16-byte group, fill = exact zeros; inv [B,H/2,W/2] fp32
 * or NULL.  sde_concat_bwd: d0 [B,H,W,C0] (add mode: also the gradient of p1), d1 [B,H,W,C1] (concat mode), d_inv [B,H/2,W/2] fp32 (2x2 sums) or NULL. */
int sde_space_to_depth(const void* x, int B, int H, int W, int C, int dtype, void* y, sde_stream_t stream);
int sde_depth_to_space(const void* x, int B, int H, int W, int C4, int dtype, void* y, sde_stream_t stream);
int sde_concat_fwd(const void* p0, const void* p1, const float* inv, int add, int B, int H, int W, int C0, int C1, int Ct, int dtype, void* out,
                   sde_stream_t stream);
int sde_concat_bwd(const void* dout, int add, int B, int H, int W, int C0, int C1, int Ct, int dtype, void* d0, void* d1, float* d_inv, sde_stream_t stream);
/* layers01.py:L105-133 InvDepth's tail + PackNet01.py:L120-123,L199 scale_inv_depth: logit = y[...,0] ([B,H,W,ld], the one-channel convolution's padded
 * output) -> inv [B,H,W] fp32 = sigmoid(logit) / min_depth_head (fed to the next decoder level) and depth [B,1,H,W] fp32 =
 * 1 / (1/max_depth + (1/min_depth - 1/max_depth) * inv), mirrored along x when flip.  Backward: d_inv and / or d_depth (either may be NULL) -> dy. */
int sde_inv_depth_head_fwd(const void* y, int B, int H, int W, int ld, float min_depth_head, float min_depth, float max_depth, int flip, int dtype, float* inv,
                           float* depth, sde_stream_t stream);
int sde_inv_depth_head_bwd(const void* y, const float* d_inv, const float* d_depth, int B, int H, int W, int ld, float min_depth_head, float min_depth,
                           float max_depth, int flip, int dtype, void* dy, sde_stream_t stream);

/* torch.optim.Adam / AdamW step (projects/MonoDepth2/train.py:L50-57, projects/Supervised/train.py:L77-81) over ONE flat fp32 buffer.
 * The descriptor is HOST memory and is copied into the kernel arguments (nothing to upload or keep alive): segment s covers
 * [seg_end[s-1], seg_end[s]) with its own lr / weight decay; bias_corr1/2 = (1 - beta1^t, 1 - beta2^t) (the step count lives on the host);
 * grad_scale multiplies g first (1/world_size after a sum all-reduce).
 * scale_state (optional, DEVICE float[4] = {loss_scale, found_inf, growth_tracker, applied_steps}): fp16 training with dynamic loss
 * scaling, the GradScaler of the reference's AMPTrainer (detectron2/engine/train_loop.py:L294-341) without its host sync:
 * sde_grad_check raises found_inf when any gradient is inf / nan, sde_adam_step divides the gradients by loss_scale and SKIPS the whole
 * update when found_inf is set, sde_loss_scale_update then backs the scale off (x backoff_factor) or counts towards growth (x
 * growth_factor every growth_interval clean steps), counts the step in applied_steps when it was NOT skipped, and clears found_inf.
 * With scale_state the optimizer's step count lives on the device (GradScaler.step skips optimizer.step(), so Adam's `step` does not
 * advance on an overflow): the kernel ignores bias_corr1/2 and forms 1 - beta^(applied_steps + 1) itself from beta1_d / beta2_d.
 * The loss is multiplied by scale_state[0] on the device before backward. */
#define SDE_ADAM_MAX_SEG 8
typedef struct sde_adam_desc {
    long seg_end[SDE_ADAM_MAX_SEG];
    float seg_lr[SDE_ADAM_MAX_SEG], seg_wd[SDE_ADAM_MAX_SEG];
    int32_t nseg, decoupled_wd;
    float beta1, beta2, eps, bias_corr1, bias_corr2, grad_scale;
    const float* scale_state;
    double beta1_d, beta2_d;       /* the betas in double (device-side bias corrections of the scale_state path) */
} sde_adam_desc;
int sde_adam_step(float* p, const float* g, float* m, float* v, long n, const sde_adam_desc* d, sde_stream_t stream);
int sde_grad_check(const float* g, long n, float* scale_state, sde_stream_t stream);
int sde_loss_scale_update(float* scale_state, float growth_factor, float backoff_factor, int growth_interval, sde_stream_t stream);


/* ---------------------------------------------------------------------------------------------------
 * Device-side input pipeline: N uint8 frames [N][Hs][Ws][3] of ONE source size -> the resized (h x w), colour-jittered frame and
 * the resized un-jittered frame as fp32 NCHW in [0, 1] (`img` / `img_orig`, `ctx_img` / `ctx_img_orig` of the batch dict), replacing the CPU chain
 * Resize -> RandomImageAug -> ToTensor of detectron2/data/preprocess/augmentation.py:L124-166,L229-266 for the image entries, in the same integer /
 * float32 arithmetic (OpenCV's fixed-point INTER_LINEAR; Pillow's Image.blend / L / HSV conversions), i.e. bit-identical to this package's CPU chain.
 * xtab [w][4] = (x0, x1, a0, a1), ytab [h][4] = (y0, y1, b0, b1): OpenCV's source taps and 11-bit weights per output column / row (device, 16-byte
 * aligned; data/device_aug.py builds them once per size pair).  jit [N][8] (device): brightness, contrast, saturation, hue factor, then the order of
 * the four steps (0 brightness, 1 contrast, 2 saturation, 3 hue) as floats; jit[n][4] < 0: frame n is not jittered.  lsum [N]: workspace.
 * orig may be NULL.  Two launches (the contrast step needs the frame's mean luminance at that point of the chain), no host synchronisation. */
int sde_image_prep_u8(const uint8_t* src, int N, int Hs, int Ws, int h, int w, const int* xtab, const int* ytab, const float* jit, unsigned* lsum, float* img,
                      float* orig, sde_stream_t stream);
/* G <= 4 frame tensors of one shape and one parameter set per sample (a MonoDepth2 batch: target frames + context frames) in ONE pair of launches:
 * src / img / orig are HOST arrays of G device pointers (orig NULL: none), lsum [G*N]. */
int sde_image_prep_u8_multi(const uint8_t* const* src, int G, int N, int Hs, int Ws, int h, int w, const int* xtab, const int* ytab, const float* jit, unsigned* lsum,
                            float* const* img, float* const* orig, sde_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Evaluation: detectron2/evaluation/depth_evaluation.py:L74-104 kitti_evaluator.process for ONE image, with
 * compute_errors (L30-53), garg_crop / eigen_crop (L16-27, as the window [y0,y1) x [x0,x1) in gt coordinates), the 1e-3 < gt < 80 median
 * scaling of TEST.GT_SCALE (L91-93) and the postprocess.backward() chain folded into two index maps: ymap[gh], xmap[gw] give the
 * prediction row / column every ground-truth row / column reads (Resize.backward = cv2 INTER_NEAREST, augmentation.py:L163-166; the crop
 * un-pastes of L67-74, L113-120; -1 = outside the prediction: value 0).  pred [ph,pw], gt [gh,gw] fp32, maps int32, all device memory.
 * part: workspace [sde_depth_metrics_num_blocks(y1-y0, x1-x0)][SDE_EVAL_NSUM] doubles; med: workspace of 4 x 4 bytes, ZEROED by the caller;
 * keys: workspace of 2 * (y1-y0) * (x1-x0) uint32 for the median selection (needed only when gt_scale == 1).
 * gt_scale: 0 = no scaling; 1 = compute the medians into med[0..1], then scale; 2 = med[0..1] already hold the medians of this image and
 * window (the evaluators of one config share them: same mask, same crop).
 * out[12] doubles = silog, log10, abs_rel, sq_rel, rms, log_rms, d1, d2, d3 (compute_errors' return order), then the number of valid pixels
 * (0: the reference skips the image, L102), median(gt), median(pred) (0 when gt_scale == 0).  No host synchronisation. */
#define SDE_EVAL_NSUM 11
int sde_depth_metrics_num_blocks(int crop_h, int crop_w);
int sde_depth_metrics(const float* pred, int ph, int pw, const float* gt, int gh, int gw, const int* ymap, const int* xmap, int y0, int y1,
                      int x0, int x1, float min_depth, float max_depth, int gt_scale, double* part, float* med, unsigned* keys, double* out,
                      sde_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SDE_HIP_H */
