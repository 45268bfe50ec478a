// KITTI depth metrics on the GPU: the per-image body of kitti_evaluator.process
// (detectron2/evaluation/depth_evaluation.py:L74-104) + compute_errors (L30-53) as two launches per image (four with median scaling), no host round trip:
//
//   1. eval_compact_kernel + eval_select_kernel (only with TEST.GT_SCALE) exact medians of gt[m] and pred[m], m = 1e-3 < gt < 80 inside the
//                           crop window: the valid pixels' order-preserving float keys are compacted by the whole grid, then a 4-pass 8-bit
//                           radix select finds the two middle ranks (np.median: mean of the two middle elements);
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

// Median step 1: every workgroup appends the order-preserving keys of the valid pixels (1e-3 < gt < 80 inside the window) of its 1024-pixel
// chunks to two dense arrays (gt keys, prediction keys).  Four independent pixels per thread, ballot prefix sums inside a wave, ONE global
// atomic per chunk reserves the slots.  The order of the dense arrays is arbitrary -- a median does not depend on it.  cnt = number of keys
// (zeroed by the caller).
__global__ void __launch_bounds__(256) eval_compact_kernel(EvalP p, unsigned* keys_g, unsigned* keys_p, unsigned* cnt) {
    __shared__ unsigned wave_tot[4], s_base;
    const int cw = p.x1 - p.x0, npx = (p.y1 - p.y0) * cw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int chunk = blockIdx.x; chunk * 1024 < npx; chunk += gridDim.x) {
        float g[4]; int yy[4], xx[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = chunk * 1024 + j * 256 + threadIdx.x;
            const int ic = i < npx ? i : npx - 1;
            yy[j] = p.y0 + ic / cw; xx[j] = p.x0 + ic % cw;
            g[j] = i < npx ? p.gt[(size_t)yy[j] * p.gw + xx[j]] : 0.f;
        }
        unsigned long long m[4];
        unsigned tot = 0u;
#pragma unroll
        for (int j = 0; j < 4; ++j) { m[j] = __ballot(g[j] > 1e-3f && g[j] < 80.f); tot += (unsigned)__popcll(m[j]); }
        if (lane == 0) wave_tot[wave] = tot;
        __syncthreads();
        if (threadIdx.x == 0) s_base = atomicAdd(cnt, wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3]);
        __syncthreads();
        unsigned at = s_base;
        for (int w = 0; w < wave; ++w) at += wave_tot[w];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if ((m[j] >> lane) & 1ull) {
                const unsigned slot = at + (unsigned)__popcll(m[j] & ((1ull << lane) - 1ull));
                keys_g[slot] = fkey(g[j]); keys_p[slot] = fkey(pred_at(p, yy[j], xx[j]));
            }
            at += (unsigned)__popcll(m[j]);
        }
        __syncthreads();
    }
}

// Median step 2: grid = 2 workgroups of 1024 (blockIdx.x == 0: gt keys, 1: prediction keys).  4-pass, 8-bit, MSB-first radix select of the two
// middle ranks over the dense keys; LDS histograms.  In the first pass nearly every key of a wave falls into the same two or three bins (sign +
// exponent bits), so lanes with equal bins are counted with one atomic per distinct bin; the lower bytes are spread and use plain LDS atomics.
__global__ void __launch_bounds__(1024) eval_select_kernel(const unsigned* keys_g, const unsigned* keys_p, const unsigned* cnt, float* med) {
    __shared__ unsigned hist[2][256];
    __shared__ unsigned s_prefix[2], s_rank[2];
    const int which = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const unsigned* keys = which == 0 ? keys_g : keys_p;
    const unsigned n = *cnt;
    if (n == 0u) {                                   // nothing valid: the evaluator skips the image (count 0 in out[9])
        if (tid == 0) { med[which] = 0.f; med[2] = 0.f; }
        return;
    }
    unsigned prefix[2] = {0u, 0u}, rank[2] = {(n - 1) / 2, n / 2};
    unsigned mask = 0u;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        for (int i = tid; i < 512; i += 1024) hist[i >> 8][i & 255] = 0u;
        __syncthreads();
        // 8 independent loads per thread and trip: a single workgroup has no other way to hide the memory latency
        for (unsigned base = 0; base < n; base += 8192u) {
            unsigned kk[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned i = base + j * 1024u + tid;
                kk[j] = i < n ? keys[i] : 0u;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool in = base + j * 1024u + tid < n;
                const unsigned k = kk[j], b = (k >> shift) & 255u;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    bool mine = in && (k & mask) == prefix[h];
                    if (pass == 0) {
                        unsigned long long todo = __ballot(mine);
                        while (todo) {
                            const int leader = __ffsll((long long)todo) - 1;
                            const unsigned lb = __shfl(b, leader, 64);
                            const unsigned long long same = __ballot(mine && b == lb);
                            if (lane == leader) atomicAdd(&hist[h][lb], (unsigned)__popcll(same));
                            todo &= ~same;
                            if (b == lb) mine = false;
                        }
                    } else if (mine) {
                        atomicAdd(&hist[h][b], 1u);
                    }
                }
            }
        }
        __syncthreads();
        if (tid < 128) {        // one wave per tracked rank: 4 bins per lane, shuffle scan over the lanes, the lane whose range holds the rank resolves it
            const int h = tid >> 6;
            const unsigned c0 = hist[h][4 * lane], c1 = hist[h][4 * lane + 1], c2 = hist[h][4 * lane + 2], c3 = hist[h][4 * lane + 3];
            const unsigned sum = c0 + c1 + c2 + c3;
            unsigned incl = sum;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const unsigned t = __shfl_up(incl, o, 64);
                if (lane >= o) incl += t;
            }
            const unsigned excl = incl - sum, r = rank[h];
            if (r >= excl && r < incl) {             // exactly one lane: the rank is below the number of keys that share the prefix
                unsigned rr = r - excl, b = 4u * lane;
                if (rr >= c0) { rr -= c0; ++b; if (rr >= c1) { rr -= c1; ++b; if (rr >= c2) { rr -= c2; ++b; } } }
                s_prefix[h] = prefix[h] | (b << shift);
                s_rank[h] = rr;
            }
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
        if (which == 0) med[2] = (float)n;
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

// out[0..8] = silog, log10, abs_rel, sq_rel, rms, log_rms, d1, d2, d3 (compute_errors' return order), out[9] = n, out[10..11] = medians.
// NSUM waves: wave k adds column k of the partials (lanes stride the rows, then a fixed-order butterfly), thread 0 forms the metrics.
__global__ void __launch_bounds__(64 * NSUM) eval_finalize_kernel(const double* part, int nblk, const float* med, int gt_scale, double* out) {
    __shared__ double tot[NSUM];
    const int k = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double a = 0.0;
    for (int b = lane; b < nblk; b += 64) a += part[(size_t)b * NSUM + k];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
    if (lane == 0) tot[k] = a;
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
                      int x0, int x1, float min_depth, float max_depth, int gt_scale, double* part, float* med, unsigned* keys, double* out,
                      sde_stream_t stream) {
    SDE_CHECK_ARG(pred && gt && ymap && xmap && part && med && out, "sde_depth_metrics: null pointer");
    SDE_CHECK_ARG(gt_scale >= 0 && gt_scale <= 2, "sde_depth_metrics: gt_scale must be 0, 1 or 2");
    SDE_CHECK_ARG(ph > 0 && pw > 0 && gh > 0 && gw > 0, "sde_depth_metrics: bad shape");
    SDE_CHECK_ARG(0 <= y0 && y0 < y1 && y1 <= gh && 0 <= x0 && x0 < x1 && x1 <= gw, "sde_depth_metrics: crop window [%d,%d)x[%d,%d) outside %dx%d",
                  y0, y1, x0, x1, gh, gw);
    EvalP p{pred, pw, gt, gw, ymap, xmap, y0, y1, x0, x1, min_depth, max_depth};
    hipStream_t s = (hipStream_t)stream;
    const int nb = sde_depth_metrics_num_blocks(y1 - y0, x1 - x0);
    if (gt_scale == 1) {        // gt_scale == 2: med[0..1] already hold this image's medians (another evaluator computed them)
        SDE_CHECK_ARG(keys, "sde_depth_metrics: gt_scale = 1 needs the key workspace");
        const size_t npx = (size_t)(y1 - y0) * (x1 - x0);
        unsigned* cnt = reinterpret_cast<unsigned*>(med) + 3;        // zeroed by the caller together with med
        hipLaunchKernelGGL(eval_compact_kernel, dim3(nb), dim3(256), 0, s, p, keys, keys + npx, cnt);
        SDE_CHECK_LAUNCH("sde_depth_metrics/compact");
        hipLaunchKernelGGL(eval_select_kernel, dim3(2), dim3(1024), 0, s, keys, keys + npx, cnt, med);
        SDE_CHECK_LAUNCH("sde_depth_metrics/select");
    }
    hipLaunchKernelGGL(eval_sums_kernel, dim3(nb), dim3(256), 0, s, p, med, gt_scale, part);
    SDE_CHECK_LAUNCH("sde_depth_metrics/sums");
    hipLaunchKernelGGL(eval_finalize_kernel, dim3(1), dim3(64 * NSUM), 0, s, part, nb, med, gt_scale, out);
    SDE_CHECK_LAUNCH("sde_depth_metrics/finalize");
    return SDE_OK;
}

}  // extern "C"
