// KITTI depth metrics on the GPU: the per-image body of kitti_evaluator.process
// (detectron2/evaluation/depth_evaluation.py:L74-104) + compute_errors (L30-53) as three launches per image, no host round trip:
//
//   1. eval_median_kernel   (only with TEST.GT_SCALE) exact medians of gt[m] and pred[m], m = 1e-3 < gt < 80 inside the crop window, by a
//                           4-pass 8-bit radix select over order-preserving float keys (np.median: mean of the two middle elements);
//   2. eval_sums_kernel     one pass over the crop window: per-pixel terms in fp32 in numpy's operation order, accumulated in fp64;
//   3. eval_finalize_kernel fixed-order sum of the per-workgroup partials -> the 9 metrics.
//
// The prediction is read THROUGH the index maps ymap[gh] / xmap[gw] (gt pixel -> prediction pixel, -1 = outside: value 0), which are the
// composition of the evaluator's postprocess.backward() chain (Resize: nearest, augmentation.py:L163-166; KBCrop / CropTopTo: paste into a
// zero canvas, L67-74 / L113-120) -- the resized full-resolution prediction is never materialised.
// HBM-bound by construction: 8 B per crop pixel (gt + gathered pred), ~250 K pixels per KITTI image.
#include "common.h"
#include "sde_hip.h"

namespace {

constexpr int NSUM = SDE_EVAL_NSUM;     // 11 accumulators, see eval_sums_kernel

struct EvalP {
    const float* pred; int pw;
    const float* gt; int gw;
    const int* ymap; const int* xmap;
    int y0, y1, x0, x1;
    float lo, hi;
};

__device__ __forceinline__ float pred_at(const EvalP& p, int y, int x) {
    const int py = p.ymap[y], px = p.xmap[x];
    return (py >= 0 && px >= 0) ? p.pred[(size_t)py * p.pw + px] : 0.f;
}
// order-preserving float -> uint key (total order incl. negatives) and back
__device__ __forceinline__ unsigned fkey(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

// grid = 2 workgroups of 1024: blockIdx.x == 0 selects over gt, == 1 over the prediction; med[which], med[2] = count
__global__ void __launch_bounds__(1024) eval_median_kernel(EvalP p, float* med) {
    __shared__ unsigned hist[2][256];
    __shared__ unsigned s_prefix[2], s_rank[2], s_n;
    const int which = blockIdx.x, tid = threadIdx.x;
    const int cw = p.x1 - p.x0, npx = (p.y1 - p.y0) * cw;
    unsigned prefix[2] = {0u, 0u}, rank[2] = {0u, 0u};
    unsigned mask = 0u;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = tid; i < 512; i += 1024) hist[i >> 8][i & 255] = 0u;
        __syncthreads();
        for (int i = tid; i < npx; i += 1024) {
            const int y = p.y0 + i / cw, x = p.x0 + i % cw;
            const float g = p.gt[(size_t)y * p.gw + x];
            if (!(g > 1e-3f && g < 80.f)) continue;
            const unsigned k = fkey(which == 0 ? g : pred_at(p, y, x));
            const unsigned b = (k >> shift) & 255u;
            if ((k & mask) == prefix[0]) atomicAdd(&hist[0][b], 1u);
            if ((k & mask) == prefix[1]) atomicAdd(&hist[1][b], 1u);
        }
        __syncthreads();
        if (tid < 2) {
            if (pass == 0) {
                unsigned n = 0;
                for (int b = 0; b < 256; ++b) n += hist[0][b];
                if (tid == 0) s_n = n;
                rank[tid] = n ? (tid == 0 ? (n - 1) / 2 : n / 2) : 0u;
            }
            unsigned r = rank[tid], b = 0;
            for (; b < 255u; ++b) {
                const unsigned c = hist[tid][b];
                if (r < c) break;
                r -= c;
            }
            s_prefix[tid] = prefix[tid] | (b << shift);
            s_rank[tid] = r;
        }
        __syncthreads();
        prefix[0] = s_prefix[0]; prefix[1] = s_prefix[1];
        rank[0] = s_rank[0]; rank[1] = s_rank[1];
        mask |= 255u << shift;
        __syncthreads();
    }
    if (tid == 0) {
        const float a = fkey_inv(prefix[0]), b = fkey_inv(prefix[1]);
        med[which] = (a + b) * 0.5f;             // np.median of float32: fp32 mean of the two middle elements
        if (which == 0) med[2] = (float)s_n;
    }
}

// sums: 0 n, 1..3 [thresh < 1.25^k], 4 (gt-pred)^2, 5 (log gt - log pred)^2, 6 |gt-pred|/gt, 7 (gt-pred)^2/gt, 8 err, 9 err^2 (err = log pred - log gt),
// 10 |log10 pred - log10 gt|
__global__ void __launch_bounds__(256) eval_sums_kernel(EvalP p, const float* med, int gt_scale, double* part) {
    __shared__ double red[4][NSUM];
    const int cw = p.x1 - p.x0, npx = (p.y1 - p.y0) * cw;
    float mg = 1.f, mp = 1.f;
    if (gt_scale) { mg = med[0]; mp = med[1]; }
    double s[NSUM];
#pragma unroll
    for (int k = 0; k < NSUM; ++k) s[k] = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < npx; i += gridDim.x * 256) {
        const int y = p.y0 + i / cw, x = p.x0 + i % cw;
        const float g = p.gt[(size_t)y * p.gw + x];
        if (!(g > p.lo && g < p.hi)) continue;
        float pr = pred_at(p, y, x);
        if (gt_scale) pr = (pr * mg) / mp;       // pred * median(gt) / median(pred), left to right in fp32 (L93)
        const float t = fmaxf(g / pr, pr / g);
        const float d = g - pr, lg = logf(g), lp = logf(pr);
        const float dl = lg - lp, e = lp - lg;
        s[0] += 1.0;
        s[1] += t < 1.25f ? 1.0 : 0.0;
        s[2] += t < 1.5625f ? 1.0 : 0.0;
        s[3] += t < 1.953125f ? 1.0 : 0.0;
        s[4] += (double)(d * d);
        s[5] += (double)(dl * dl);
        s[6] += (double)(fabsf(d) / g);
        s[7] += (double)((d * d) / g);
        s[8] += (double)e;
        s[9] += (double)(e * e);
        s[10] += (double)fabsf(log10f(pr) - log10f(g));
    }
#pragma unroll
    for (int k = 0; k < NSUM; ++k)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int k = 0; k < NSUM; ++k) red[wave][k] = s[k];
    __syncthreads();
    if (threadIdx.x < NSUM) part[(size_t)blockIdx.x * NSUM + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out[0..8] = silog, log10, abs_rel, sq_rel, rms, log_rms, d1, d2, d3 (compute_errors' return order), out[9] = n, out[10..11] = medians
__global__ void eval_finalize_kernel(const double* part, int nblk, const float* med, int gt_scale, double* out) {
    __shared__ double tot[NSUM];
    if (threadIdx.x < NSUM) {
        double a = 0.0;
        for (int b = 0; b < nblk; ++b) a += part[(size_t)b * NSUM + threadIdx.x];
        tot[threadIdx.x] = a;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double n = tot[0], inv = n > 0 ? 1.0 / n : 0.0;
        const double me = tot[8] * inv, me2 = tot[9] * inv;
        out[0] = sqrt(me2 - me * me + 1e-8) * 100.0;
        out[1] = tot[10] * inv;
        out[2] = tot[6] * inv;
        out[3] = tot[7] * inv;
        out[4] = sqrt(tot[4] * inv);
        out[5] = sqrt(tot[5] * inv);
        out[6] = tot[1] * inv;
        out[7] = tot[2] * inv;
        out[8] = tot[3] * inv;
        out[9] = n;
        out[10] = gt_scale ? (double)med[0] : 0.0;
        out[11] = gt_scale ? (double)med[1] : 0.0;
    }
}

}  // namespace

extern "C" {

int sde_depth_metrics_num_blocks(int crop_h, int crop_w) {
    const long nb = ((long)crop_h * crop_w + 256 * 4 - 1) / (256 * 4);
    return (int)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
}

int sde_depth_metrics(const float* pred, int ph, int pw, const float* gt, int gh, int gw, const int* ymap, const int* xmap, int y0, int y1,
                      int x0, int x1, float min_depth, float max_depth, int gt_scale, double* part, float* med, double* out,
                      sde_stream_t stream) {
    SDE_CHECK_ARG(pred && gt && ymap && xmap && part && med && out, "sde_depth_metrics: null pointer");
    SDE_CHECK_ARG(ph > 0 && pw > 0 && gh > 0 && gw > 0, "sde_depth_metrics: bad shape");
    SDE_CHECK_ARG(0 <= y0 && y0 < y1 && y1 <= gh && 0 <= x0 && x0 < x1 && x1 <= gw, "sde_depth_metrics: crop window [%d,%d)x[%d,%d) outside %dx%d",
                  y0, y1, x0, x1, gh, gw);
    EvalP p{pred, pw, gt, gw, ymap, xmap, y0, y1, x0, x1, min_depth, max_depth};
    hipStream_t s = (hipStream_t)stream;
    if (gt_scale) {
        hipLaunchKernelGGL(eval_median_kernel, dim3(2), dim3(1024), 0, s, p, med);
        SDE_CHECK_LAUNCH("sde_depth_metrics/median");
    }
    const int nb = sde_depth_metrics_num_blocks(y1 - y0, x1 - x0);
    hipLaunchKernelGGL(eval_sums_kernel, dim3(nb), dim3(256), 0, s, p, med, gt_scale, part);
    SDE_CHECK_LAUNCH("sde_depth_metrics/sums");
    hipLaunchKernelGGL(eval_finalize_kernel, dim3(1), dim3(64), 0, s, part, nb, med, gt_scale, out);
    SDE_CHECK_LAUNCH("sde_depth_metrics/finalize");
    return SDE_OK;
}

}  // extern "C"
