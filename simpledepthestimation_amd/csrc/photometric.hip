// Photometric-reprojection path for gfx950 (MI355X): image pyramid, inverse warp + bilinear sampling,
// SSIM + L1 photometric map, min-reprojection / auto-mask reduce, edge-aware smoothness, masked SILog.
//
// Replaces (reference, read-only): detectron2/geometry/camera.py:L14-46,L125-202 (scale_intrinsics,
// resize_img, img_to_points, points_to_img, view_synthesis + F.grid_sample),
// detectron2/modeling/losses/ssim_loss.py:L34-53, smoothness_loss.py:L42-80, losses.py:L5-18 and the
// loss loop of detectron2/modeling/meta_arch/MonoDepth2.py:L78-151.
//
// Numerics: fp32 everywhere.  This file MUST be compiled with -ffp-contract=off: the projection chain
// reproduces the reference's fp32 operation order (explicit fmaf where torch-CPU bmm fuses, separate
// mul/add where it does not) so that floor() of the sample coordinate is bit-exact.
//
// Data layout in HBM: planar NCHW fp32 images ([B,3,h,w]) exactly as the reference's batch dict hands
// them over, depth [B,1,h,w].  One workgroup = one TWxTH pixel tile of one sample staged through LDS
// with a 1-pixel (forward) / 2-pixel (backward) halo; 64-wide waves run along x so every global
// access is a contiguous 128..256 B row segment per plane.
#include "common.h"
#include "sde_hip.h"

namespace {

constexpr int FT_W = 32, FT_H = 16;            // threads per block (positions incl. halo)
constexpr int FT_N = FT_W * FT_H;              // 512
// backward tile (2-pixel halo on each side: 28 x 12 owned pixels of 32 x 16 threads).  32 x 32 threads own 77 % instead of 66 % but measured
// no faster at 192x640 (195.8 vs 195.2 us) and 25 % slower on the coarser scales: the kernel waits on its barriers and gathers, not on lanes.
constexpr int BT_W = 32, BT_H = 16, BT_N = BT_W * BT_H;
constexpr float kEps = 1e-6f;
constexpr float kFltMax = 3.402823466e+38f;

__device__ __forceinline__ int reflect_idx(int i, int n) {   // ReflectionPad2d(1) index map
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

struct Cam {
    float ki[9];   // inverse of the scaled intrinsics (camera.py:L25-37)
    float kr[9];   // K @ R   (3x3 @ 3x3: torch naive path, no FMA)
    float kt[3];   // K @ t   (MKL path: mul, fma, fma)
    float k[9];    // scaled intrinsics (camera.py:L14-22)
};

// The camera of a (sample, context) is the same for every lane of a workgroup: moved into scalar registers (v_readfirstlane), its 30 values stop
// occupying vector registers across the projection / VJP code (the backward kernel is register-bound: 120 VGPRs = 4 waves per SIMD with them in VGPRs).
__device__ __forceinline__ float sde_uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ Cam cam_uniform(const Cam& c) {
    Cam u;
#pragma unroll
    for (int i = 0; i < 9; ++i) { u.ki[i] = sde_uniform(c.ki[i]); u.kr[i] = sde_uniform(c.kr[i]); u.k[i] = sde_uniform(c.k[i]); }
#pragma unroll
    for (int i = 0; i < 3; ++i) u.kt[i] = sde_uniform(c.kt[i]);
    return u;
}

__device__ __forceinline__ void make_cam(const float* __restrict__ K, const float* __restrict__ P, float sx, float sy,
                                         Cam& c) {
#pragma unroll
    for (int i = 0; i < 9; ++i) c.k[i] = K[i];
    c.k[0] = c.k[0] * sx; c.k[4] = c.k[4] * sy; c.k[2] = c.k[2] * sx; c.k[5] = c.k[5] * sy;
#pragma unroll
    for (int i = 0; i < 9; ++i) c.ki[i] = c.k[i];
    c.ki[0] = 1.0f / c.k[0];
    c.ki[4] = 1.0f / c.k[4];
    c.ki[2] = (-1.0f * c.k[2]) / c.k[0];
    c.ki[5] = (-1.0f * c.k[5]) / c.k[4];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            c.kr[3 * i + j] = (c.k[3 * i] * P[j] + c.k[3 * i + 1] * P[4 + j]) + c.k[3 * i + 2] * P[8 + j];
        c.kt[i] = fmaf(c.k[3 * i + 2], P[11], fmaf(c.k[3 * i + 1], P[7], c.k[3 * i] * P[3]));
    }
}

struct Proj {
    float p[3];      // back-projected point (camera A)
    float q[3];      // projected homogeneous coords (camera B)
    float X, Y;      // q0/(q2+eps), q1/(q2+eps)
    float ix, iy;    // un-normalised sample coordinate after nan_to_num/clamp/normalise round trip
    bool passx, passy;  // gradient passes nan_to_num + clamp
};

__device__ __forceinline__ float nan_to_num(float v) {
    if (v != v) return 0.f;
    if (v > kFltMax) return kFltMax;
    if (v < -kFltMax) return -kFltMax;
    return v;
}

__device__ __forceinline__ void project(const Cam& c, int x, int y, float d, int W, int H, Proj& o) {
    const float g0 = (float)x * d, g1 = (float)y * d, g2 = d;
#pragma unroll
    for (int i = 0; i < 3; ++i) o.p[i] = fmaf(c.ki[3 * i + 2], g2, fmaf(c.ki[3 * i + 1], g1, c.ki[3 * i] * g0)) + 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
        o.q[i] = fmaf(c.kr[3 * i + 2], o.p[2], fmaf(c.kr[3 * i + 1], o.p[1], c.kr[3 * i] * o.p[0])) + c.kt[i];
    const float den = o.q[2] + kEps;
    o.X = o.q[0] / den;
    o.Y = o.q[1] / den;
    const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
    float xs = nan_to_num(o.X), ys = nan_to_num(o.Y);
    o.passx = (o.X == o.X) && (fabsf(o.X) <= kFltMax) && xs >= 0.f && xs <= wm1;
    o.passy = (o.Y == o.Y) && (fabsf(o.Y) <= kFltMax) && ys >= 0.f && ys <= hm1;
    xs = fminf(fmaxf(xs, 0.f), wm1);
    ys = fminf(fmaxf(ys, 0.f), hm1);
    const float xn = (2.0f * xs) / wm1 - 1.0f;
    const float yn = (2.0f * ys) / hm1 - 1.0f;
    o.ix = (xn + 1.0f) * (wm1 / 2.0f);     // ATen CPU grid_sampler un-normalise, align_corners=True
    o.iy = (yn + 1.0f) * (hm1 / 2.0f);
}

struct Taps {
    int x0, y0;
    float wx, ex, ny, sy;   // east/west/north/south weights as in ATen's compute_interp_params
    bool okx0, okx1, oky0, oky1;
};

__device__ __forceinline__ void make_taps(float ix, float iy, int W, int H, Taps& t) {
    const float fx = floorf(ix), fy = floorf(iy);
    t.x0 = (int)fx; t.y0 = (int)fy;
    t.wx = ix - fx; t.ex = 1.0f - t.wx;
    t.ny = iy - fy; t.sy = 1.0f - t.ny;
    t.okx0 = t.x0 >= 0 && t.x0 < W;       t.okx1 = t.x0 + 1 >= 0 && t.x0 + 1 < W;
    t.oky0 = t.y0 >= 0 && t.y0 < H;       t.oky1 = t.y0 + 1 >= 0 && t.y0 + 1 < H;
}

// Bilinear sample of one plane (zeros padding).  Returns value; optionally the 4 tap values.
__device__ __forceinline__ float bilinear(const float* __restrict__ pl, int W, const Taps& t, float* v4) {
    const int x0 = t.x0, y0 = t.y0;
    const float nw = (t.okx0 && t.oky0) ? pl[(long)y0 * W + x0] : 0.f;
    const float ne = (t.okx1 && t.oky0) ? pl[(long)y0 * W + x0 + 1] : 0.f;
    const float sw = (t.okx0 && t.oky1) ? pl[(long)(y0 + 1) * W + x0] : 0.f;
    const float se = (t.okx1 && t.oky1) ? pl[(long)(y0 + 1) * W + x0 + 1] : 0.f;
    if (v4) { v4[0] = nw; v4[1] = ne; v4[2] = sw; v4[3] = se; }
    return nw * (t.sy * t.ex) + ne * (t.sy * t.wx) + sw * (t.ny * t.ex) + se * (t.ny * t.wx);
}

// ------------------------------------------------------------------------------------------------
// Standalone view synthesis (parity surface of camera.py:L166-202; also used by tests for indices)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) view_synthesis_kernel(const float* __restrict__ img, const float* __restrict__ depth,
                                                             const float* __restrict__ K, const float* __restrict__ pose,
                                                             float sx, float sy, int B, int C, int H, int W,
                                                             float* __restrict__ sampled, float* __restrict__ Zout,
                                                             float* __restrict__ grid, uint8_t* __restrict__ valid,
                                                             int* __restrict__ fxo, int* __restrict__ fyo) {
    const int b = blockIdx.z;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    Cam cam;
    make_cam(K + 9 * b, pose + 16 * b, sx, sy, cam);
    const long pix = (long)y * W + x;
    const long hw = (long)H * W;
    Proj pr;
    project(cam, x, y, depth[b * hw + pix], W, H, pr);
    Taps t;
    make_taps(pr.ix, pr.iy, W, H, t);
    for (int c = 0; c < C; ++c) sampled[((long)b * C + c) * hw + pix] = bilinear(img + ((long)b * C + c) * hw, W, t, nullptr);
    if (Zout) Zout[b * hw + pix] = fmaxf(pr.q[2], 1e-5f);
    if (grid) {
        const float wm1 = (float)(W - 1), hm1 = (float)(H - 1);
        const float xs = fminf(fmaxf(nan_to_num(pr.X), 0.f), wm1), ys = fminf(fmaxf(nan_to_num(pr.Y), 0.f), hm1);
        grid[(b * hw + pix) * 2 + 0] = (2.0f * xs) / wm1 - 1.0f;
        grid[(b * hw + pix) * 2 + 1] = (2.0f * ys) / hm1 - 1.0f;
    }
    if (valid) {
        const bool fin = (pr.X == pr.X) && fabsf(pr.X) <= kFltMax && (pr.Y == pr.Y) && fabsf(pr.Y) <= kFltMax;
        valid[b * hw + pix] = (fin && pr.X >= 0.f && pr.X < (float)(W - 1) && pr.Y >= 0.f && pr.Y < (float)(H - 1) && pr.q[2] > 0.f) ? 1 : 0;
    }
    if (fxo) fxo[b * hw + pix] = t.x0;
    if (fyo) fyo[b * hw + pix] = t.y0;
}

// ------------------------------------------------------------------------------------------------
// Resize (image pyramid): bilinear align_corners=True, and nearest (camera.py:L40-46)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) resize_bilinear_ac_kernel(const float* __restrict__ src, float* __restrict__ dst, int planes,
                                                                 int H, int W, int h, int w, float sh, float sw) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float fy = (float)y * sh, fx = (float)x * sw;
    int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    y0 = min(y0, H - 1); x0 = min(x0, W - 1);
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
    const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    for (int p = blockIdx.z; p < planes; p += gridDim.z) {
        const float* s = src + (long)p * H * W;
        const float top = lx0 * s[(long)y0 * W + x0] + lx1 * s[(long)y0 * W + x1];
        const float bot = lx0 * s[(long)y1 * W + x0] + lx1 * s[(long)y1 * W + x1];
        dst[((long)p * h + y) * w + x] = ly0 * top + ly1 * bot;
    }
}

__global__ void __launch_bounds__(256) resize_nearest_kernel(const float* __restrict__ src, float* __restrict__ dst, int planes, int H,
                                                             int W, int h, int w, float sh, float sw) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const int ys = min((int)floorf((float)y * sh), H - 1), xs = min((int)floorf((float)x * sw), W - 1);
    for (int p = blockIdx.z; p < planes; p += gridDim.z) dst[((long)p * h + y) * w + x] = src[((long)p * H + ys) * W + xs];
}

// ------------------------------------------------------------------------------------------------
// Fused forward: warp + sample + SSIM + L1 + mix, all maps of one scale, min/mean reduce, block sums
// ------------------------------------------------------------------------------------------------
struct PhotoArgs {
    const float* A;
    const float* ctx[SDE_MAX_CTX];
    const float* pose[SDE_MAX_CTX];
    float* sampled[SDE_MAX_CTX];
    const float* depth;
    const float* K;
    uint8_t* sel;
    float* partial;
    float* maps;   // optional [B, nmaps, h, w] per-map photometric values (tests / debugging), may be null
    const float* thr;   // optional [nmaps] LOSS.CLIP thresholds (mean + clip * std of each unclipped map), may be null
    int B, h, w, nctx, automask, reduce_mean;
    float sx, sy, ssim_w, C1, C2;
};

// clamp((1 - SSIM)/2, 0, 1) from the nine-tap sums (ssim_loss.py:L34-53: mu = avgpool3x3, sigma = E[x^2] - mu^2)
__device__ __forceinline__ float ssim_from_moments(float sx, float sy, float sxx, float syy, float sxy, float C1, float C2) {
#pragma clang fp contract(fast)
    const float inv9 = 1.0f / 9.0f;
    const float mx = sx * inv9, my = sy * inv9;
    const float mxy = mx * my, mxx = mx * mx, myy = my * my;
    const float vx = sxx * inv9 - mxx, vy = syy * inv9 - myy, vxy = sxy * inv9 - mxy;
    const float n = (2.0f * mxy + C1) * (2.0f * vxy + C2);
    const float d = (mxx + myy + C1) * (vx + vy + C2);
    return fminf(fmaxf((1.0f - n * __builtin_amdgcn_rcpf(d)) * 0.5f, 0.f), 1.f);     // v_rcp_f32 (1 ulp) instead of the 10-instruction IEEE division
}

// bx, by, bz / gdx, gdy: this workgroup's tile coordinates and the tile grid of ITS scale (the multi-scale launch packs the grids of all scales
// into one linear grid; the single-scale launch passes blockIdx / gridDim)
template <int NCTX>
__device__ __forceinline__ void photo_fwd_body(const PhotoArgs& a, const int bx, const int by, const int bz, const int gdx, const int gdy) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sA = lds;                        // [3][FT_N]
    float* sC = sA + 3 * FT_N;              // [NCTX][3][FT_N]  raw context frames (identity / auto-mask maps)
    float* sS = sC + NCTX * 3 * FT_N;       // [NCTX][3][FT_N]  warped samples
    float* red = sS + NCTX * 3 * FT_N;      // [16]
    const int tx = threadIdx.x, ty = threadIdx.y, lp = ty * FT_W + tx;
    const int b = bz, h = a.h, w = a.w;
    const long hw = (long)h * w;
    // position in the image of this thread (with halo); reflected when outside (ReflectionPad2d(1))
    const int gx = bx * (FT_W - 2) + tx - 1, gy = by * (FT_H - 2) + ty - 1;
    const int rx = reflect_idx(gx, w), ry = reflect_idx(gy, h);
    const bool usable = rx >= 0 && rx < w && ry >= 0 && ry < h;   // false only far outside on ragged tiles
    const bool inimg = gx >= 0 && gx < w && gy >= 0 && gy < h;
    const long pix = usable ? (long)ry * w + rx : 0;
    const float d = usable ? a.depth[b * hw + pix] : 1.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) sA[c * FT_N + lp] = usable ? a.A[((long)b * 3 + c) * hw + pix] : 0.f;
    // the camera matrices are per (sample, context): one thread each builds them, everybody reads them back (LDS broadcast)
    __shared__ Cam scam[NCTX];
#pragma unroll
    for (int j = 0; j < NCTX; ++j)
        if (lp == j) make_cam(a.K + 9 * b, a.pose[j] + 16 * b, a.sx, a.sy, scam[j]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NCTX; ++j) {
        const Cam cam = scam[j];
        Proj pr;
        project(cam, rx, ry, d, w, h, pr);
        Taps t;
        make_taps(pr.ix, pr.iy, w, h, t);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* pl = a.ctx[j] + ((long)b * 3 + c) * hw;
            const float s = usable ? bilinear(pl, w, t, nullptr) : 0.f;
            sS[(j * 3 + c) * FT_N + lp] = s;
            sC[(j * 3 + c) * FT_N + lp] = usable ? pl[pix] : 0.f;
            if (inimg && tx >= 1 && tx < FT_W - 1 && ty >= 1 && ty < FT_H - 1 && a.sampled[j]) a.sampled[j][((long)b * 3 + c) * hw + pix] = s;
        }
    }
    __syncthreads();
    float v = 0.f;
    const bool interior = tx >= 1 && tx < FT_W - 1 && ty >= 1 && ty < FT_H - 1 && inimg;
    if (interior) {
        // FMA contraction is fine here (and only here): the file is built with -ffp-contract=off for the projection chain, whose sample
        // indices must match the reference bit for bit; the SSIM / L1 arithmetic is held to 1e-5 like every other fp32 map
#pragma clang fp contract(fast)
        const int nmaps = a.automask ? 2 * NCTX : NCTX;
        // channel-outer / map-inner: the 3x3 moments of the target frame (sum y, sum y^2) and its nine values are shared by all
        // maps of a channel; every individual sum keeps the operand order of ssim_dist(), so the results are unchanged bit for bit
        float l1[2 * NCTX], ss[2 * NCTX];
#pragma unroll
        for (int m = 0; m < 2 * NCTX; ++m) { l1[m] = 0.f; ss[m] = 0.f; }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float* ys = sA + c * FT_N;
            float y[9], sy = 0.f, syy = 0.f;
#pragma unroll
            for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int k = (dy + 1) * 3 + dx + 1;
                    y[k] = ys[lp + dy * FT_W + dx];
                    sy += y[k]; syy += y[k] * y[k];
                }
#pragma unroll
            for (int m = 0; m < 2 * NCTX; ++m) {
                const int j = m >> 1;
                const bool ident = m & 1;
                if (ident && !a.automask) continue;
                const float* xs = (ident ? sC : sS) + (j * 3 + c) * FT_N;
                l1[m] += fabsf(xs[lp] - y[4]);
                if (a.ssim_w > 0.f) {
                    float sx = 0.f, sxx = 0.f, sxy = 0.f;
#pragma unroll
                    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                        for (int dx = -1; dx <= 1; ++dx) {
                            const float x = xs[lp + dy * FT_W + dx], yk = y[(dy + 1) * 3 + dx + 1];
                            sx += x; sxx += x * x; sxy += x * yk;
                        }
                    ss[m] += ssim_from_moments(sx, sy, sxx, syy, sxy, a.C1, a.C2);
                }
            }
        }
        float best = 0.f, acc = 0.f;
        int bi = 0;
        unsigned clipped = 0;
#pragma unroll
        for (int m = 0; m < 2 * NCTX; ++m) {
            const bool ident = m & 1;
            if (ident && !a.automask) continue;
            const float third = 1.0f / 3.0f;
            const float l = l1[m] * third;
            float pm = l;
            if (a.ssim_w > 0.f) pm = (ss[m] * third) * a.ssim_w + l * (1.0f - a.ssim_w);
            const int mi = a.automask ? m : (m >> 1);
            if (a.thr && pm > a.thr[mi]) { pm = a.thr[mi]; clipped |= 1u << mi; }      // torch.clamp(max=...): no gradient where it bites
            if (a.maps) a.maps[(((long)b * nmaps + mi) * h + gy) * w + gx] = pm;
            acc += pm;
            if (mi == 0 || pm < best) { best = pm; bi = mi; }
        }
        v = a.reduce_mean ? acc * (1.0f / (float)nmaps) : best;
        // sel: 'min' -> index of the arg-min map (254: it was clipped, no gradient); 'mean' -> bit mask of the clipped maps (255 without clipping)
        if (a.sel) a.sel[b * hw + (long)gy * w + gx] = a.reduce_mean ? (a.thr ? (uint8_t)clipped : 255) : (((clipped >> bi) & 1u) ? 254 : (uint8_t)bi);
    }
    const float s = sde_block_sum(v, red);
    if (lp == 0) a.partial[(bz * gdy + by) * gdx + bx] = s;
}

template <int NCTX>
__global__ void __launch_bounds__(FT_N) photo_fwd_kernel(PhotoArgs a) {
    photo_fwd_body<NCTX>(a, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}

// All scales of the loss in ONE launch: the coarse scales are launch-sized on their own (31 / 19 / 14 us for 1/4 ... 1/64 of the pixels of the
// 90 us full-resolution launch); packed behind the fine scale's workgroups they fill its tail instead.  Scale s owns the linear blocks
// [first[s], first[s+1]); fine scale first, so the long workgroups start first.
constexpr int PH_MAX_SCALES = 4;
struct PhotoMulti {
    PhotoArgs a[PH_MAX_SCALES];
    int first[PH_MAX_SCALES + 1];
    int gdx[PH_MAX_SCALES], gdy[PH_MAX_SCALES];
    int n;
};

template <int NCTX>
__global__ void __launch_bounds__(FT_N) photo_fwd_multi_kernel(const PhotoMulti m) {
    int s = 0;
    while (s + 1 < m.n && (int)blockIdx.x >= m.first[s + 1]) ++s;
    const int local = blockIdx.x - m.first[s], gdx = m.gdx[s], gdy = m.gdy[s];
    photo_fwd_body<NCTX>(m.a[s], local % gdx, (local / gdx) % gdy, local / (gdx * gdy), gdx, gdy);
}

// ------------------------------------------------------------------------------------------------
// Fused backward: d(rec_loss)/d(depth) and d/d(pose) through SSIM + L1 + bilinear sampling + projection
// ------------------------------------------------------------------------------------------------
struct PhotoBwdArgs {
    const float* A;
    const float* ctx[SDE_MAX_CTX];
    const float* pose[SDE_MAX_CTX];
    const float* sampled[SDE_MAX_CTX];
    const float* depth;
    const float* K;
    const uint8_t* sel;
    const float* gout;      // device scalar: upstream gradient of this scale's reduced loss
    float* d_depth;         // [B,1,h,w]
    float* pose_partial;    // [nblocks][NCTX][12]  (dR row-major 9, dt 3)
    int B, h, w, nctx, automask, reduce_mean, accumulate, clip;   // clip: sel carries LOSS.CLIP information (see photo_fwd_kernel)
    float sx, sy, ssim_w, C1, C2, gscale;   // gscale = 1/(B*h*w)  (mean over pixels)
};

// DENSE: the instantiation for grids that fill the GPU several times over (the two fine scales): compiled for six waves per SIMD (80 VGPRs, three
// workgroups per CU; seven values spill) -- 164 -> 145 us at 192x640, 55 -> 50 at 96x320; the coarse scales, whose grids do not fill the CUs, keep
// the spill-free 92-VGPR form (they lose 15 % with the spills: 22.7 -> 26.1, 17.5 -> 20.2 us)
template <int NCTX>
__device__ __forceinline__ void photo_bwd_body(const PhotoBwdArgs& a, const int bx, const int by, const int bz, const int gdx, const int gdy) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* sA = lds;                  // [3][BT_N]
    float* sX = sA + 3 * BT_N;        // [3][BT_N]     current context's warped sample
    float* sK = sX + 3 * BT_N;        // [3][3][BT_N]  per-window coefficients (dA, dB, dC) per channel
    float* red = sK + 9 * BT_N;       // [waves][NCTX][12]
    const int tx = threadIdx.x, ty = threadIdx.y, lp = ty * BT_W + tx;
    const int b = bz, h = a.h, w = a.w;
    const long hw = (long)h * w;
    const int gx = bx * (BT_W - 4) + tx - 2, gy = by * (BT_H - 4) + ty - 2;
    const int rx = reflect_idx(gx, w), ry = reflect_idx(gy, h);
    const bool usable = rx >= 0 && rx < w && ry >= 0 && ry < h;
    const bool inimg = gx >= 0 && gx < w && gy >= 0 && gy < h;
    const long pix = usable ? (long)ry * w + rx : 0;
    const bool win = inimg && tx >= 1 && tx < BT_W - 1 && ty >= 1 && ty < BT_H - 1;      // window centres this block evaluates
    const bool interior = inimg && tx >= 2 && tx < BT_W - 2 && ty >= 2 && ty < BT_H - 2;  // pixels this block owns
    const int nmaps = a.automask ? 2 * NCTX : NCTX;
    const float g = a.gout[0] * a.gscale;
    const float g_mean = g / (float)nmaps;        // 'mean' reduce: every map's share
    const int mysel = inimg ? a.sel[b * hw + pix] : 254;
#pragma unroll
    for (int c = 0; c < 3; ++c) sA[c * BT_N + lp] = usable ? a.A[((long)b * 3 + c) * hw + pix] : 0.f;
    const float d = usable ? a.depth[b * hw + pix] : 1.f;
    float dd = 0.f;   // d loss / d depth at this pixel
    const int blk = (bz * gdy + by) * gdx + bx;
    __shared__ Cam scam[NCTX];       // camera matrices per (sample, context): built by one thread each, read back as LDS broadcasts
#pragma unroll
    for (int j = 0; j < NCTX; ++j)
        if (lp == j) make_cam(a.K + 9 * b, a.pose[j] + 16 * b, a.sx, a.sy, scam[j]);
    for (int j = 0; j < NCTX; ++j) {
        __syncthreads();   // previous iteration's readers are done with sX / sK
#pragma unroll
        for (int c = 0; c < 3; ++c) sX[c * BT_N + lp] = usable ? a.sampled[j][((long)b * 3 + c) * hw + pix] : 0.f;
        __syncthreads();
        const int mi = a.automask ? 2 * j : j;
        // weight of this map's value at window centre (min: indicator of the arg-min; mean: 1/nmaps)
        const bool mean_on = !(a.clip && ((mysel >> mi) & 1));       // 'mean' reduce: this map's value was not clipped at this pixel
        const float gw = win ? (a.reduce_mean ? (mean_on ? g_mean : 0.f) : (mysel == mi ? g : 0.f)) : 0.f;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float cA = 0.f, cB = 0.f, cC = 0.f;
            if (gw != 0.f && a.ssim_w > 0.f) {
#pragma clang fp contract(fast)
                const float* xs = sX + c * BT_N; const float* ys = sA + c * BT_N;
                float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
                for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
                    for (int dx = -1; dx <= 1; ++dx) {
                        const float x = xs[lp + dy * BT_W + dx], y = ys[lp + dy * BT_W + dx];
                        sx += x; sy += y; sxx += x * x; syy += y * y; sxy += x * y;
                    }
                const float inv9 = 1.0f / 9.0f;
                const float mx = sx * inv9, my = sy * inv9, exx = sxx * inv9, eyy = syy * inv9, exy = sxy * inv9;
                const float n1 = 2.0f * mx * my + a.C1, n2 = 2.0f * (exy - mx * my) + a.C2;
                const float d1 = mx * mx + my * my + a.C1, d2 = (exx - mx * mx) + (eyy - my * my) + a.C2;
                const float n = n1 * n2, dn = d1 * d2;
                const float rdn = __builtin_amdgcn_rcpf(dn);                    // one v_rcp_f32 for the four quotients below
                const float l = (1.0f - n * rdn) * 0.5f;
                if (l >= 0.f && l <= 1.f) {
                    const float f = -0.5f * gw * (a.ssim_w * (1.0f / 3.0f)) * inv9;      // d loss / d ssim, folded with the 1/9 of the box filter
                    const float dn_dmx = 2.0f * my * (n2 - n1), dd_dmx = 2.0f * mx * (d2 - d1);
                    cA = f * (dn_dmx * dn - n * dd_dmx) * (rdn * rdn);
                    cB = f * (-n * d1) * (rdn * rdn);
                    cC = f * (2.0f * n1) * rdn;
                }
            }
            sK[(c * 3 + 0) * BT_N + lp] = cA; sK[(c * 3 + 1) * BT_N + lp] = cB; sK[(c * 3 + 2) * BT_N + lp] = cC;
        }
        __syncthreads();
        float acc12[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) acc12[i] = 0.f;
        const Cam cam = cam_uniform(scam[j]);
        if (interior) {
            float ds[3];
            const float l1w = a.reduce_mean ? (mean_on ? g_mean : 0.f) : (mysel == mi ? g : 0.f);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float xq = sX[c * BT_N + lp], aq = sA[c * BT_N + lp];
                float s = 0.f;
#pragma unroll
                for (int ey = -1; ey <= 1; ++ey) {
                    const int wy = gy + ey;
                    if (wy < 0 || wy >= h) continue;
                    const float my_ = ((gy == 1 && ey == -1) || (gy == h - 2 && ey == 1)) ? 2.f : 1.f;
#pragma unroll
                    for (int ex = -1; ex <= 1; ++ex) {
                        const int wx_ = gx + ex;
                        if (wx_ < 0 || wx_ >= w) continue;
                        const float mx_ = ((gx == 1 && ex == -1) || (gx == w - 2 && ex == 1)) ? 2.f : 1.f;
                        const int q = lp + ey * BT_W + ex;
                        s += (my_ * mx_) * (sK[(c * 3 + 0) * BT_N + q] + 2.0f * xq * sK[(c * 3 + 1) * BT_N + q] + aq * sK[(c * 3 + 2) * BT_N + q]);
                    }
                }
                const float df = xq - aq;
                const float sg = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
                const float l1c = (a.ssim_w > 0.f) ? (1.0f - a.ssim_w) : 1.0f;
                ds[c] = s + l1w * l1c * sg * (1.0f / 3.0f);
            }
            Proj pr;
            project(cam, gx, gy, d, w, h, pr);
            Taps t;
            make_taps(pr.ix, pr.iy, w, h, t);
            float dix = 0.f, diy = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v4[4];
                bilinear(a.ctx[j] + ((long)b * 3 + c) * hw, w, t, v4);
                dix += ds[c] * ((v4[1] - v4[0]) * t.sy + (v4[3] - v4[2]) * t.ny);
                diy += ds[c] * ((v4[2] - v4[0]) * t.ex + (v4[3] - v4[1]) * t.wx);
            }
            const float dX = pr.passx ? dix : 0.f, dY = pr.passy ? diy : 0.f;
            const float den = pr.q[2] + kEps;
            float dq[3];
            const float rden = 1.0f / den;
            dq[0] = dX * rden; dq[1] = dY * rden;
            dq[2] = -(dX * pr.X + dY * pr.Y) * rden;
            const float fxp = (float)gx, fyp = (float)gy;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float dp = dq[0] * cam.kr[k] + dq[1] * cam.kr[3 + k] + dq[2] * cam.kr[6 + k];
                dd += dp * (cam.ki[3 * k] * fxp + cam.ki[3 * k + 1] * fyp + cam.ki[3 * k + 2]);
            }
            // dKR_ik = dq_i * p_k ; dKt_i = dq_i  -> transform by K^T:  dR_mk = sum_i K_im dKR_ik ; dt_m = sum_i K_im dq_i
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const float kq = cam.k[m] * dq[0] + cam.k[3 + m] * dq[1] + cam.k[6 + m] * dq[2];
#pragma unroll
                for (int k = 0; k < 3; ++k) acc12[3 * m + k] = kq * pr.p[k];
                acc12[9 + m] = kq;
            }
        }
        // the 12 pose-gradient terms: wave sums now (every wave owns its slots of `red`: no barrier), the block sum once after the loop
#pragma unroll
        for (int i = 0; i < 12; ++i) acc12[i] = sde_wave_sum(acc12[i]);
        if ((lp & 63) == 0)
#pragma unroll
            for (int i = 0; i < 12; ++i) red[((lp >> 6) * NCTX + j) * 12 + i] = acc12[i];
    }
    __syncthreads();
    if (lp < 12 * NCTX) {
        float s = 0.f;
        for (int wv = 0; wv < BT_N / 64; ++wv) s += red[wv * NCTX * 12 + lp];
        a.pose_partial[(long)blk * NCTX * 12 + lp] = s;
    }
    if (interior) {
        float* o = a.d_depth + b * hw + pix;
        *o = a.accumulate ? (*o + dd) : dd;
    }
}

template <int NCTX, bool DENSE = false>
__global__ void __launch_bounds__(BT_N, DENSE ? 6 : 1) photo_bwd_kernel(PhotoBwdArgs a) {
    photo_bwd_body<NCTX>(a, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}

struct PhotoBwdMulti {
    PhotoBwdArgs a[PH_MAX_SCALES];
    int first[PH_MAX_SCALES + 1];
    int gdx[PH_MAX_SCALES], gdy[PH_MAX_SCALES];
    int n;
};

template <int NCTX, bool DENSE>
__global__ void __launch_bounds__(BT_N, DENSE ? 6 : 1) photo_bwd_multi_kernel(const PhotoBwdMulti m) {
    int s = 0;
    while (s + 1 < m.n && (int)blockIdx.x >= m.first[s + 1]) ++s;
    const int local = blockIdx.x - m.first[s], gdx = m.gdx[s], gdy = m.gdy[s];
    photo_bwd_body<NCTX>(m.a[s], local % gdx, (local / gdx) % gdy, local / (gdx * gdy), gdx, gdy);
}

// ------------------------------------------------------------------------------------------------
// Stand-alone SSIM distance map (the callable module of ssim_loss.py:L6-53): out = clamp((1 - SSIM(x, y)) / 2, 0, 1) per pixel and channel,
// ReflectionPad2d(1) + 3x3 mean, planar [B*C][H][W] fp32.  The training path evaluates SSIM inside photo_fwd / photo_bwd; this is the operator
// surface for callers of the module itself.  Backward: per window the four coefficients of d(out)/d(window sums) -- cAx, cAy (means), cB
// (second moments), cC (cross moment) -- then every pixel gathers the up to nine windows that contain it, with the multiplicity the
// reflection gives border taps.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) ssim_map_kernel(const float* __restrict__ x, const float* __restrict__ y, int planes, int H, int W, float C1, float C2,
                                                       float* __restrict__ out, const float* __restrict__ gout, float* __restrict__ coef) {
#pragma clang fp contract(fast)
    const int px = blockIdx.x * 64 + (threadIdx.x & 63), py = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= W || py >= H) return;
    for (int p = blockIdx.z; p < planes; p += gridDim.z) {
        const float* xs = x + (long)p * H * W;
        const float* ys = y + (long)p * H * W;
        float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const int ry = reflect_idx(py + dy, H);
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const int rx = reflect_idx(px + dx, W);
                const float a = xs[(long)ry * W + rx], b = ys[(long)ry * W + rx];
                sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
            }
        }
        const long o = ((long)p * H + py) * W + px;
        if (out) out[o] = ssim_from_moments(sx, sy, sxx, syy, sxy, C1, C2);
        if (coef) {
            const float inv9 = 1.0f / 9.0f;
            const float mx = sx * inv9, my = sy * inv9, exx = sxx * inv9, eyy = syy * inv9, exy = sxy * inv9;
            const float n1 = 2.0f * mx * my + C1, n2 = 2.0f * (exy - mx * my) + C2;
            const float d1 = mx * mx + my * my + C1, d2 = (exx - mx * mx) + (eyy - my * my) + C2;
            const float n = n1 * n2, dn = d1 * d2, rdn = __builtin_amdgcn_rcpf(dn);
            const float l = (1.0f - n * rdn) * 0.5f;
            float cAx = 0.f, cAy = 0.f, cB = 0.f, cC = 0.f;
            if (l >= 0.f && l <= 1.f) {                                  // clamp passes the gradient on [0, 1] only
                const float f = -0.5f * gout[o] * inv9;                    // d out / d SSIM, folded with the 1/9 of the box filter
                const float dd = 2.0f * (d2 - d1);
                cAx = f * (2.0f * my * (n2 - n1) * dn - n * dd * mx) * (rdn * rdn);
                cAy = f * (2.0f * mx * (n2 - n1) * dn - n * dd * my) * (rdn * rdn);
                cB = f * (-n * d1) * (rdn * rdn);
                cC = f * (2.0f * n1) * rdn;
            }
            reinterpret_cast<float4*>(coef)[o] = make_float4(cAx, cAy, cB, cC);
        }
    }
}

__global__ void __launch_bounds__(256) ssim_bwd_gather_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ coef, int planes,
                                                              int H, int W, float* __restrict__ dx, float* __restrict__ dy) {
#pragma clang fp contract(fast)
    const int px = blockIdx.x * 64 + (threadIdx.x & 63), py = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (px >= W || py >= H) return;
    for (int p = blockIdx.z; p < planes; p += gridDim.z) {
        const long o = ((long)p * H + py) * W + px;
        const float xv = x[o], yv = y[o];
        float gx = 0.f, gy = 0.f;
#pragma unroll
        for (int ey = -1; ey <= 1; ++ey) {
            const int wy = py + ey;
            if (wy < 0 || wy >= H) continue;
            const float my_ = ((py == 1 && ey == -1) || (py == H - 2 && ey == 1)) ? 2.f : 1.f;      // the window's reflected row is this row again
#pragma unroll
            for (int ex = -1; ex <= 1; ++ex) {
                const int wx = px + ex;
                if (wx < 0 || wx >= W) continue;
                const float mx_ = ((px == 1 && ex == -1) || (px == W - 2 && ex == 1)) ? 2.f : 1.f;
                const float4 c = reinterpret_cast<const float4*>(coef)[((long)p * H + wy) * W + wx];
                const float m = my_ * mx_;
                gx += m * (c.x + 2.0f * xv * c.z + yv * c.w);
                gy += m * (c.y + 2.0f * yv * c.z + xv * c.w);
            }
        }
        if (dx) dx[o] = gx;
        if (dy) dy[o] = gy;
    }
}

// Sum per-block pose partials of each sample into d_pose [NCTX][B][4][4] (last row zero).
__global__ void pose_grad_finalize_kernel(const float* __restrict__ partial, int blocks_per_sample, int nctx, int B,
                                          float* __restrict__ dpose0, float* __restrict__ dpose1, float* __restrict__ dpose2,
                                          float* __restrict__ dpose3, int accumulate) {
    // one wave per (sample, context): the lanes split the per-workgroup partials, then a fixed butterfly per output (deterministic)
    const int b = blockIdx.x, j = blockIdx.y, lane = threadIdx.x;
    float acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = 0.f;
    for (int k = lane; k < blocks_per_sample; k += 64) {
        const float* p = partial + (((long)b * blocks_per_sample + k) * nctx + j) * 12;
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] += p[i];
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = sde_wave_sum(acc[i]);
    float* dp = j == 0 ? dpose0 : (j == 1 ? dpose1 : (j == 2 ? dpose2 : dpose3));
    if (lane < 12) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) s = lane == i ? acc[i] : s;
        const int r = lane < 9 ? lane / 3 : lane - 9, c = lane < 9 ? lane % 3 : 3;
        float* o = dp + b * 16 + r * 4 + c;
        *o = accumulate ? (*o + s) : s;
        if (!accumulate && lane < 4) dp[b * 16 + 12 + lane] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// Deterministic sum of a partial-sum slab: out[k] (+)= scale * sum(partial[0..n))
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) reduce_partials_kernel(const float* __restrict__ partial, int n, float scale, float* __restrict__ out,
                                                              int accumulate) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    s = sde_block_sum(s, red);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + s * scale : s * scale;
}

// one workgroup per scale: out[s] = scale[s] * sum(partial_s[0..n_s))  (the same fixed order as reduce_partials_kernel: per-scale losses bit-identical)
struct MultiReduce { const float* partial[PH_MAX_SCALES]; int n[PH_MAX_SCALES]; float scale[PH_MAX_SCALES]; };
__global__ void __launch_bounds__(256) reduce_partials_multi_kernel(const MultiReduce r, float* __restrict__ out) {
    __shared__ float red[16];
    const int k = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < r.n[k]; i += 256) s += r.partial[k][i];
    s = sde_block_sum(s, red);
    if (threadIdx.x == 0) out[k] = s * r.scale[k];
}

// pose gradients of all scales: one wave per (sample, context) walks the scales' partial slabs in order (scale-by-scale, the order in which the
// per-scale finalize launches used to accumulate)
struct MultiPose { const float* partial[PH_MAX_SCALES]; int blocks_per_sample[PH_MAX_SCALES]; int n; };
__global__ void pose_grad_finalize_multi_kernel(const MultiPose mp, int nctx, int B, float* __restrict__ dpose0, float* __restrict__ dpose1,
                                                float* __restrict__ dpose2, float* __restrict__ dpose3) {
    const int b = blockIdx.x, j = blockIdx.y, lane = threadIdx.x;
    float tot[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) tot[i] = 0.f;
    for (int s = 0; s < mp.n; ++s) {
        float acc[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = 0.f;
        const int bps = mp.blocks_per_sample[s];
        for (int k = lane; k < bps; k += 64) {
            const float* p = mp.partial[s] + (((long)b * bps + k) * nctx + j) * 12;
#pragma unroll
            for (int i = 0; i < 12; ++i) acc[i] += p[i];
        }
#pragma unroll
        for (int i = 0; i < 12; ++i) tot[i] += sde_wave_sum(acc[i]);
    }
    float* dp = j == 0 ? dpose0 : (j == 1 ? dpose1 : (j == 2 ? dpose2 : dpose3));
    if (lane < 12) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 12; ++i) s = lane == i ? tot[i] : s;
        const int r = lane < 9 ? lane / 3 : lane - 9, c = lane < 9 ? lane % 3 : 3;
        dp[b * 16 + r * 4 + c] = s;
        if (lane < 4) dp[b * 16 + 12 + lane] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------------
// Edge-aware smoothness (smoothness_loss.py:L42-80)
// ------------------------------------------------------------------------------------------------
constexpr int SM_CHUNKS = 32;   // partial sums per image for the inverse-depth mean

__device__ __forceinline__ void smooth_mean_body(const float* __restrict__ depth, int hw, float* __restrict__ part /*[B][SM_CHUNKS]*/, const int b, const int ch) {
    __shared__ float red[16];
    const int per = (hw + SM_CHUNKS - 1) / SM_CHUNKS;
    const int lo = ch * per, hi = min(hw, lo + per);
    float s = 0.f;
    for (int i = lo + threadIdx.x; i < hi; i += 256) s += 1.0f / fmaxf(depth[(long)b * hw + i], 1e-6f);
    s = sde_block_sum(s, red);
    if (threadIdx.x == 0) part[b * SM_CHUNKS + ch] = s;
}

__global__ void __launch_bounds__(256) smooth_mean_kernel(const float* __restrict__ depth, int hw, float* __restrict__ part) {
    smooth_mean_body(depth, hw, part, blockIdx.y, blockIdx.x);
}

// forward + the upstream-independent part of the backward:
//   dn[b,y,x] = d(loss)/d(normalised inverse depth), S partials = sum_q dn[q]*inv[q]
__device__ __forceinline__ void smooth_fwd_body(const float* __restrict__ depth, const float* __restrict__ img,
                                                const float* __restrict__ mean_part, int B, int h, int w,
                                                float* __restrict__ dn, float* __restrict__ loss_part /*[nblk]*/,
                                                float* __restrict__ s_part /*[nblk]*/, const int bx, const int by, const int bz, const int gdx, const int gdy) {
    __shared__ float red[16];
    const int b = bz;
    const int x = bx * 64 + (threadIdx.x & 63);
    const int y = by * 4 + (threadIdx.x >> 6);
    const long hw = (long)h * w;
    float m = 0.f;
    for (int i = 0; i < SM_CHUNKS; ++i) m += mean_part[b * SM_CHUNKS + i];
    m = m / (float)hw;
    const float mc = fmaxf(m, 1e-6f);
    const float nx = 1.0f / ((float)B * (float)h * (float)(w - 1));
    const float nyn = 1.0f / ((float)B * (float)(h - 1) * (float)w);
    float loss = 0.f, g = 0.f, sv = 0.f;
    if (x < w && y < h) {
        const float* D = depth + b * hw;
        const float* I = img + (long)b * 3 * hw;
        const long p = (long)y * w + x;
        const float inv = 1.0f / fmaxf(D[p], 1e-6f);
        const float n0 = inv / mc;
        float i0[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) i0[c] = I[c * hw + p];
        // term anchored here (x -> x+1), and the one anchored at the left neighbour (x-1 -> x)
        if (x < w - 1) {
            const float n1 = (1.0f / fmaxf(D[p + 1], 1e-6f)) / mc;
            float e = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) e += fabsf(i0[c] - I[c * hw + p + 1]);
            const float wt = expf(-(e / 3.0f));
            const float v = (n0 - n1) * wt;
            loss += fabsf(v) * nx;
            g += (v > 0.f ? wt : (v < 0.f ? -wt : 0.f)) * nx;
        }
        if (x > 0) {
            const float n1 = (1.0f / fmaxf(D[p - 1], 1e-6f)) / mc;
            float e = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) e += fabsf(I[c * hw + p - 1] - i0[c]);
            const float wt = expf(-(e / 3.0f));
            const float v = (n1 - n0) * wt;
            g -= (v > 0.f ? wt : (v < 0.f ? -wt : 0.f)) * nx;
        }
        if (y < h - 1) {
            const float n1 = (1.0f / fmaxf(D[p + w], 1e-6f)) / mc;
            float e = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) e += fabsf(i0[c] - I[c * hw + p + w]);
            const float wt = expf(-(e / 3.0f));
            const float v = (n0 - n1) * wt;
            loss += fabsf(v) * nyn;
            g += (v > 0.f ? wt : (v < 0.f ? -wt : 0.f)) * nyn;
        }
        if (y > 0) {
            const float n1 = (1.0f / fmaxf(D[p - w], 1e-6f)) / mc;
            float e = 0.f;
#pragma unroll
            for (int c = 0; c < 3; ++c) e += fabsf(I[c * hw + p - w] - i0[c]);
            const float wt = expf(-(e / 3.0f));
            const float v = (n1 - n0) * wt;
            g -= (v > 0.f ? wt : (v < 0.f ? -wt : 0.f)) * nyn;
        }
        if (dn) dn[b * hw + p] = g;
        sv = g * inv;
    }
    const int blk = (bz * gdy + by) * gdx + bx;
    const float ls = sde_block_sum(loss, red);
    if (threadIdx.x == 0) loss_part[blk] = ls;
    const float ss = sde_block_sum(sv, red);
    if (threadIdx.x == 0 && s_part) s_part[blk] = ss;
}

__global__ void __launch_bounds__(256) smooth_fwd_kernel(const float* __restrict__ depth, const float* __restrict__ img, const float* __restrict__ mean_part,
                                                         int B, int h, int w, float* __restrict__ dn, float* __restrict__ loss_part, float* __restrict__ s_part) {
    smooth_fwd_body(depth, img, mean_part, B, h, w, dn, loss_part, s_part, blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y);
}

__device__ __forceinline__ void smooth_bwd_body(const float* __restrict__ depth, const float* __restrict__ dn,
                                                const float* __restrict__ mean_part, const float* __restrict__ s_part,
                                                int blocks_per_sample, const float* __restrict__ gout, float gscale,
                                                int h, int w, float* __restrict__ d_depth, int accumulate, const int bx, const int by, const int bz) {
    __shared__ float sh[2];
    __shared__ float red[16];
    const int b = bz;
    const long hw = (long)h * w;
    // the sample's sum of the forward partials: every workgroup needs it, and a single thread walking the up to 480 partials one dependent load at a
    // time made this pass 6x its traffic time (61 us at 192x640); all 256 threads take a strided share, in the same fixed order in every workgroup
    float s = 0.f;
    for (int i = threadIdx.x; i < blocks_per_sample; i += 256) s += s_part[(long)b * blocks_per_sample + i];
    s = sde_block_sum(s, red);
    if (threadIdx.x == 0) {
        float m = 0.f;
        for (int i = 0; i < SM_CHUNKS; ++i) m += mean_part[b * SM_CHUNKS + i];
        sh[0] = m / (float)hw;
        sh[1] = s;
    }
    __syncthreads();
    const int x = bx * 64 + (threadIdx.x & 63);
    const int y = by * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const float m = sh[0], S = sh[1];
    const long p = b * hw + (long)y * w + x;
    const float dv = depth[p];
    const float g = gout[0] * gscale;
    float dinv;
    if (m > 1e-6f) dinv = dn[p] / m - S / ((float)hw * m * m);     // nrm = inv / mean(inv)
    else dinv = dn[p] / 1e-6f;                                    // clamp active: mean treated as constant
    const float dd = (dv > 1e-6f) ? -dinv / (dv * dv) : 0.f;       // inv = 1 / clamp(depth, 1e-6)
    d_depth[p] = accumulate ? d_depth[p] + g * dd : g * dd;
}

__global__ void __launch_bounds__(256) smooth_bwd_kernel(const float* __restrict__ depth, const float* __restrict__ dn, const float* __restrict__ mean_part,
                                                         const float* __restrict__ s_part, int blocks_per_sample, const float* __restrict__ gout, float gscale,
                                                         int h, int w, float* __restrict__ d_depth, int accumulate) {
    smooth_bwd_body(depth, dn, mean_part, s_part, blocks_per_sample, gout, gscale, h, w, d_depth, accumulate, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Every scale of MonoDepth2's loss loop (MonoDepth2.py:L78-126) behind one launch per pass: the smoothness kernels over a linear grid whose block ranges
// belong to the scales (each scale keeps the block partition of its single-scale launch, so every partial sum is the one that launch writes), and ONE
// finalize for the photometric and the smoothness partial slabs that also forms the two weighted totals the reference accumulates scale by scale.
struct SmoothArgs { const float* depth; const float* img; float* mean_part; float* dn; float* loss_part; float* s_part; const float* gout; float* d_depth; float gscale; int h, w; };
struct SmoothMulti {
    SmoothArgs a[PH_MAX_SCALES];
    int first[PH_MAX_SCALES + 1];
    int gdx[PH_MAX_SCALES], gdy[PH_MAX_SCALES];
    int n, B, accumulate;
};

__global__ void __launch_bounds__(256) smooth_mean_multi_kernel(const SmoothMulti m) {
    const SmoothArgs& a = m.a[blockIdx.z];
    smooth_mean_body(a.depth, a.h * a.w, a.mean_part, blockIdx.y, blockIdx.x);
}

__global__ void __launch_bounds__(256) smooth_fwd_multi_kernel(const SmoothMulti m) {
    int s = 0;
    while (s + 1 < m.n && (int)blockIdx.x >= m.first[s + 1]) ++s;
    const int local = blockIdx.x - m.first[s], gdx = m.gdx[s], gdy = m.gdy[s];
    const SmoothArgs& a = m.a[s];
    smooth_fwd_body(a.depth, a.img, a.mean_part, m.B, a.h, a.w, a.dn, a.loss_part, a.s_part, local % gdx, (local / gdx) % gdy, local / (gdx * gdy), gdx, gdy);
}

__global__ void __launch_bounds__(256) smooth_bwd_multi_kernel(const SmoothMulti m) {
    int s = 0;
    while (s + 1 < m.n && (int)blockIdx.x >= m.first[s + 1]) ++s;
    const int local = blockIdx.x - m.first[s], gdx = m.gdx[s], gdy = m.gdy[s];
    const SmoothArgs& a = m.a[s];
    smooth_bwd_body(a.depth, a.dn, a.mean_part, a.s_part, gdx * gdy, a.gout, a.gscale, a.h, a.w, a.d_depth, m.accumulate, local % gdx, (local / gdx) % gdy, local / (gdx * gdy));
}

// One workgroup per partial slab (photometric scales first, then smoothness scales): per_scale[k] = scale[k] * sum(slab k) in the fixed order of
// reduce_partials_kernel; the workgroup that finishes last (device-scope ticket, left at zero again) adds up totals[0] = sum_k<nphoto weight[k] * per_scale[k] and
// totals[1] = the same over the smoothness slabs, in scale order -- the order of the reference's `loss += term * w` (MonoDepth2.py:L103-112, L126).
struct MonoReduce {
    const float* partial[2 * PH_MAX_SCALES];
    int n[2 * PH_MAX_SCALES];
    float scale[2 * PH_MAX_SCALES], weight[2 * PH_MAX_SCALES];
    int count, nphoto;
};
__global__ void __launch_bounds__(256) mono_loss_finalize_kernel(const MonoReduce r, float* __restrict__ per_scale, float* __restrict__ totals, int* __restrict__ ticket) {
    __shared__ float red[16];
    __shared__ int last;
    const int k = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < r.n[k]; i += 256) s += r.partial[k][i];
    s = sde_block_sum(s, red);
    if (threadIdx.x == 0) {
        __hip_atomic_store(per_scale + k, s * r.scale[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = (__hip_atomic_fetch_add(ticket, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == r.count - 1);
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __threadfence();
        float t0 = 0.f, t1 = 0.f;
        for (int j = 0; j < r.count; ++j) {
            const float v = __hip_atomic_load(per_scale + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * r.weight[j];
            if (j < r.nphoto) t0 += v; else t1 += v;
        }
        totals[0] = t0; totals[1] = t1;
        __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------------------
// Masked SILog (losses.py:L10-13) against nearest-resized ground truth, no compaction, no host sync
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) silog_fwd_kernel(const float* __restrict__ est, const float* __restrict__ gt, int B, int h, int w,
                                                        int H, int W, float sh_, float sw_, float* __restrict__ part /*[nblk][3]*/) {
    __shared__ float red[16];
    const long n = (long)B * h * w;
    float c = 0.f, s1 = 0.f, s2 = 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int x = (int)(i % w), y = (int)((i / w) % h), b = (int)(i / ((long)w * h));
        const int ys = min((int)floorf((float)y * sh_), H - 1), xs = min((int)floorf((float)x * sw_), W - 1);
        const float g = gt[((long)b * H + ys) * W + xs];
        if (g > 1.0f) {
            const float d = logf(est[i]) - logf(g);
            c += 1.f; s1 += d; s2 += d * d;
        }
    }
    c = sde_block_sum(c, red);
    if (threadIdx.x == 0) part[blockIdx.x * 3 + 0] = c;
    s1 = sde_block_sum(s1, red);
    if (threadIdx.x == 0) part[blockIdx.x * 3 + 1] = s1;
    s2 = sde_block_sum(s2, red);
    if (threadIdx.x == 0) part[blockIdx.x * 3 + 2] = s2;
}

// stats[0..3] = (count, mean d, mean d^2, loss); loss = 10 * sqrt(E[d^2] - vf * E[d]^2)
__global__ void silog_finalize_kernel(const float* __restrict__ part, int nblk, float vf, float* __restrict__ stats) {
    __shared__ double sh[3][64];
    const int t = threadIdx.x;
    double c = 0, s1 = 0, s2 = 0;
    for (int i = t; i < nblk; i += 64) { c += part[i * 3]; s1 += part[i * 3 + 1]; s2 += part[i * 3 + 2]; }
    sh[0][t] = c; sh[1][t] = s1; sh[2][t] = s2;
    __syncthreads();
    if (t == 0) {
        c = s1 = s2 = 0;
        for (int i = 0; i < 64; ++i) { c += sh[0][i]; s1 += sh[1][i]; s2 += sh[2][i]; }
        const double m1 = s1 / c, m2 = s2 / c;
        stats[0] = (float)c; stats[1] = (float)m1; stats[2] = (float)m2;
        stats[3] = (float)(sqrt(m2 - (double)vf * m1 * m1) * 10.0);
    }
}

__global__ void __launch_bounds__(256) silog_bwd_kernel(const float* __restrict__ est, const float* __restrict__ gt, const float* __restrict__ stats,
                                                        const float* __restrict__ gout, float gscale, float vf, int B, int h, int w, int H,
                                                        int W, float sh_, float sw_, float* __restrict__ d_est, int accumulate) {
    const long n = (long)B * h * w;
    const float cnt = stats[0], m1 = stats[1], loss = stats[3];
    const float g = gout[0] * gscale;
    // d loss / d d_i = (100 / loss) * (d_i - vf * m1) / cnt        [loss = 10 sqrt(v), dv/dd_i = 2 d_i/cnt - 2 vf m1/cnt]
    const float k = (loss > 0.f) ? 100.0f / (loss * cnt) : 0.f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int x = (int)(i % w), y = (int)((i / w) % h), b = (int)(i / ((long)w * h));
        const int ys = min((int)floorf((float)y * sh_), H - 1), xs = min((int)floorf((float)x * sw_), W - 1);
        const float gv = gt[((long)b * H + ys) * W + xs];
        float r = 0.f;
        if (gv > 1.0f) {
            const float e = est[i];
            const float d = logf(e) - logf(gv);
            r = g * k * (d - vf * m1) / e;
        }
        d_est[i] = accumulate ? d_est[i] + r : r;
    }
}

// The multi-scale form of the three kernels above (Supervised.py:L42-47 sums the loss of the four prediction scales against one ground truth):
// every scale keeps the block partition and the summation order of the single-scale launch (bit-identical per-scale statistics), the launches are
// one per phase instead of one per scale, and the finalize also forms sum_k weight[k] * loss[k].
struct SilogScales {
    const float* est[SDE_SILOG_MAX_SCALES];
    float* d_est[SDE_SILOG_MAX_SCALES];
    int h[SDE_SILOG_MAX_SCALES], w[SDE_SILOG_MAX_SCALES], blk_end[SDE_SILOG_MAX_SCALES];
    float sh[SDE_SILOG_MAX_SCALES], sw[SDE_SILOG_MAX_SCALES], weight[SDE_SILOG_MAX_SCALES];
    int n;
};

__device__ __forceinline__ int silog_scale_of(const SilogScales& a, int blk, int& first, int& count) {
    int k = 0;
    while (k + 1 < a.n && blk >= a.blk_end[k]) ++k;
    first = k ? a.blk_end[k - 1] : 0;
    count = a.blk_end[k] - first;
    return k;
}

__global__ void __launch_bounds__(256) silog_multi_fwd_kernel(SilogScales a, const float* __restrict__ gt, int B, int H, int W, float* __restrict__ part) {
    __shared__ float red[16];
    int first, count;
    const int k = silog_scale_of(a, blockIdx.x, first, count);
    const int h = a.h[k], w = a.w[k];
    const float* __restrict__ est = a.est[k];
    const float sh_ = a.sh[k], sw_ = a.sw[k];
    const long n = (long)B * h * w;
    float c = 0.f, s1 = 0.f, s2 = 0.f;
    for (long i = (long)(blockIdx.x - first) * 256 + threadIdx.x; i < n; i += (long)count * 256) {
        const int x = (int)(i % w), y = (int)((i / w) % h), b = (int)(i / ((long)w * h));
        const int ys = min((int)floorf((float)y * sh_), H - 1), xs = min((int)floorf((float)x * sw_), W - 1);
        const float g = gt[((long)b * H + ys) * W + xs];
        if (g > 1.0f) {
            const float d = logf(est[i]) - logf(g);
            c += 1.f; s1 += d; s2 += d * d;
        }
    }
    c = sde_block_sum(c, red);
    if (threadIdx.x == 0) part[blockIdx.x * 3 + 0] = c;
    s1 = sde_block_sum(s1, red);
    if (threadIdx.x == 0) part[blockIdx.x * 3 + 1] = s1;
    s2 = sde_block_sum(s2, red);
    if (threadIdx.x == 0) part[blockIdx.x * 3 + 2] = s2;
}

// stats[k][0..3] as silog_finalize_kernel; total[0] = sum_k weight[k] * loss[k] (scales in order, fp32).  One wave per scale, each doing exactly what the
// single-scale finalize does (same lanes, same order), side by side.
__global__ void __launch_bounds__(64 * SDE_SILOG_MAX_SCALES) silog_multi_finalize_kernel(const float* __restrict__ part, SilogScales a, float vf,
                                                                                         float* __restrict__ stats, float* __restrict__ total) {
    __shared__ double sh[SDE_SILOG_MAX_SCALES][3][64];
    __shared__ float loss[SDE_SILOG_MAX_SCALES];
    const int t = threadIdx.x & 63, k = threadIdx.x >> 6;
    if (k < a.n) {
        const int first = k ? a.blk_end[k - 1] : 0, nblk = a.blk_end[k] - first;
        const float* p = part + (long)first * 3;
        double c = 0, s1 = 0, s2 = 0;
        for (int i = t; i < nblk; i += 64) { c += p[i * 3]; s1 += p[i * 3 + 1]; s2 += p[i * 3 + 2]; }
        sh[k][0][t] = c; sh[k][1][t] = s1; sh[k][2][t] = s2;
    }
    __syncthreads();
    if (k < a.n && t == 0) {
        double c = 0, s1 = 0, s2 = 0;
        for (int i = 0; i < 64; ++i) { c += sh[k][0][i]; s1 += sh[k][1][i]; s2 += sh[k][2][i]; }
        const double m1 = s1 / c, m2 = s2 / c;
        stats[4 * k + 0] = (float)c; stats[4 * k + 1] = (float)m1; stats[4 * k + 2] = (float)m2;
        loss[k] = (float)(sqrt(m2 - (double)vf * m1 * m1) * 10.0);
        stats[4 * k + 3] = loss[k];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int j = 0; j < a.n; ++j) s += a.weight[j] * loss[j];
        total[0] = s;
    }
}

__global__ void __launch_bounds__(256) silog_multi_bwd_kernel(SilogScales a, const float* __restrict__ gt, const float* __restrict__ stats,
                                                              const float* __restrict__ gout, float gscale, float vf, int B, int H, int W) {
    int first, count;
    const int k = silog_scale_of(a, blockIdx.x, first, count);
    const int h = a.h[k], w = a.w[k];
    const float* __restrict__ est = a.est[k];
    float* __restrict__ d_est = a.d_est[k];
    const float sh_ = a.sh[k], sw_ = a.sw[k];
    const long n = (long)B * h * w;
    const float cnt = stats[4 * k], m1 = stats[4 * k + 1], loss = stats[4 * k + 3];
    const float g = gout[0] * gscale * a.weight[k];
    const float kk = (loss > 0.f) ? 100.0f / (loss * cnt) : 0.f;
    for (long i = (long)(blockIdx.x - first) * 256 + threadIdx.x; i < n; i += (long)count * 256) {
        const int x = (int)(i % w), y = (int)((i / w) % h), b = (int)(i / ((long)w * h));
        const int ys = min((int)floorf((float)y * sh_), H - 1), xs = min((int)floorf((float)x * sw_), W - 1);
        const float gv = gt[((long)b * H + ys) * W + xs];
        float r = 0.f;
        if (gv > 1.0f) {
            const float e = est[i];
            const float d = logf(e) - logf(gv);
            r = g * kk * (d - vf * m1) / e;
        }
        d_est[i] = r;
    }
}

// pose_vec2mat (pose_utils.py:L98-137): R = X(rx) Y(ry) Z(rz), T = [R t; 0 0 0 1]; also its VJP
__global__ void pose_vec2mat_kernel(const float* __restrict__ vec, float* __restrict__ mat, int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float* v = vec + 6 * i;
    const float cx = cosf(v[3]), sx = sinf(v[3]), cy = cosf(v[4]), sy = sinf(v[4]), cz = cosf(v[5]), sz = sinf(v[5]);
    // X*Y = [[cy,0,sy],[sx*sy,cx,-sx*cy],[-cx*sy,sx,cx*cy]]
    const float xy[9] = {cy, 0.f, sy, sx * sy, cx, -sx * cy, -cx * sy, sx, cx * cy};
    float* m = mat + 16 * i;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        m[4 * r + 0] = xy[3 * r] * cz + xy[3 * r + 1] * sz;
        m[4 * r + 1] = -xy[3 * r] * sz + xy[3 * r + 1] * cz;
        m[4 * r + 2] = xy[3 * r + 2];
        m[4 * r + 3] = v[r];
    }
    m[12] = 0.f; m[13] = 0.f; m[14] = 0.f; m[15] = 1.f;
}

__global__ void pose_vec2mat_bwd_kernel(const float* __restrict__ vec, const float* __restrict__ dmat, float* __restrict__ dvec, int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float* v = vec + 6 * i;
    const float* g = dmat + 16 * i;
    const float cx = cosf(v[3]), sx = sinf(v[3]), cy = cosf(v[4]), sy = sinf(v[4]), cz = cosf(v[5]), sz = sinf(v[5]);
    const float xy[9] = {cy, 0.f, sy, sx * sy, cx, -sx * cy, -cx * sy, sx, cx * cy};
    // d/drx of XY, d/dry of XY
    const float dxy_x[9] = {0.f, 0.f, 0.f, cx * sy, -sx, -cx * cy, sx * sy, cx, -sx * cy};
    const float dxy_y[9] = {-sy, 0.f, cy, sx * cy, 0.f, sx * sy, -cx * cy, 0.f, -cx * sy};
    float gx = 0.f, gy = 0.f, gz = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const float g0 = g[4 * r], g1 = g[4 * r + 1], g2 = g[4 * r + 2];
        // R[r][0] = a*cz + b*sz ; R[r][1] = -a*sz + b*cz ; R[r][2] = c   with (a,b,c) = xy row r
        const float da = g0 * cz - g1 * sz, db = g0 * sz + g1 * cz, dc = g2;
        gx += da * dxy_x[3 * r] + db * dxy_x[3 * r + 1] + dc * dxy_x[3 * r + 2];
        gy += da * dxy_y[3 * r] + db * dxy_y[3 * r + 1] + dc * dxy_y[3 * r + 2];
        gz += g0 * (-xy[3 * r] * sz + xy[3 * r + 1] * cz) + g1 * (-xy[3 * r] * cz - xy[3 * r + 1] * sz);
    }
    float* o = dvec + 6 * i;
    o[0] = g[3]; o[1] = g[7]; o[2] = g[11]; o[3] = gx; o[4] = gy; o[5] = gz;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int sde_view_synthesis(const float* img, const float* depth, const float* K, const float* pose, float sx, float sy, int B, int C,
                       int H, int W, float* sampled, float* Z, float* grid, uint8_t* valid, int32_t* fx, int32_t* fy,
                       sde_stream_t stream) {
    SDE_CHECK_ARG(img && depth && K && pose && sampled, "sde_view_synthesis: null pointer");
    SDE_CHECK_ARG(B > 0 && C > 0 && H > 1 && W > 1, "sde_view_synthesis: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    dim3 grid_(sde_cdiv(W, 64), sde_cdiv(H, 4), B);
    hipLaunchKernelGGL(view_synthesis_kernel, grid_, dim3(256), 0, (hipStream_t)stream, img, depth, K, pose, sx, sy, B, C, H, W, sampled, Z,
                       grid, valid, fx, fy);
    SDE_CHECK_LAUNCH("sde_view_synthesis");
    return SDE_OK;
}

int sde_resize(const float* src, float* dst, int planes, int H, int W, int h, int w, int mode, sde_stream_t stream) {
    SDE_CHECK_ARG(src && dst, "sde_resize: null pointer");
    SDE_CHECK_ARG(planes > 0 && H > 0 && W > 0 && h > 0 && w > 0, "sde_resize: bad shape");
    SDE_CHECK_ARG(mode == SDE_RESIZE_BILINEAR_AC || mode == SDE_RESIZE_NEAREST, "sde_resize: bad mode %d", mode);
    dim3 g(sde_cdiv(w, 64), sde_cdiv(h, 4), planes < 64 ? planes : 64);
    if (mode == SDE_RESIZE_BILINEAR_AC) {
        const float sh = h > 1 ? (float)((double)(H - 1) / (double)(h - 1)) : 0.f, sw = w > 1 ? (float)((double)(W - 1) / (double)(w - 1)) : 0.f;
        hipLaunchKernelGGL(resize_bilinear_ac_kernel, g, dim3(256), 0, (hipStream_t)stream, src, dst, planes, H, W, h, w, sh, sw);
    } else {
        const float sh = (float)((double)H / (double)h), sw = (float)((double)W / (double)w);
        hipLaunchKernelGGL(resize_nearest_kernel, g, dim3(256), 0, (hipStream_t)stream, src, dst, planes, H, W, h, w, sh, sw);
    }
    SDE_CHECK_LAUNCH("sde_resize");
    return SDE_OK;
}

static size_t photo_fwd_lds(int nctx) { return (size_t)((3 + 6 * nctx) * FT_N + 16) * sizeof(float); }
static size_t photo_bwd_lds() { return (size_t)((3 + 3 + 9) * BT_N + (BT_N / 64) * SDE_MAX_CTX * 12) * sizeof(float); }

int sde_photo_num_blocks(int B, int h, int w, int backward) {
    if (backward) return sde_cdiv(w, BT_W - 4) * sde_cdiv(h, BT_H - 4) * B;
    return sde_cdiv(w, FT_W - 2) * sde_cdiv(h, FT_H - 2) * B;
}

int sde_photo_fwd(const sde_photo_desc* d, float* const* sampled, uint8_t* sel, float* maps, float* partial, float* loss_out,
                  float loss_scale, int accumulate, sde_stream_t stream) {
    SDE_CHECK_ARG(d && d->A && d->depth && d->K && partial && loss_out, "sde_photo_fwd: null pointer");
    SDE_CHECK_ARG(d->nctx >= 1 && d->nctx <= SDE_MAX_CTX, "sde_photo_fwd: nctx=%d out of range", d->nctx);
    SDE_CHECK_ARG(d->B > 0 && d->h >= 4 && d->w >= 4, "sde_photo_fwd: bad shape B=%d h=%d w=%d", d->B, d->h, d->w);
    PhotoArgs a;
    a.A = d->A; a.depth = d->depth; a.K = d->K; a.sel = sel; a.partial = partial; a.maps = maps; a.thr = d->clip_thr;
    for (int j = 0; j < SDE_MAX_CTX; ++j) {
        a.ctx[j] = j < d->nctx ? d->ctx[j] : nullptr;
        a.pose[j] = j < d->nctx ? d->pose[j] : nullptr;
        a.sampled[j] = (j < d->nctx && sampled) ? sampled[j] : nullptr;
        SDE_CHECK_ARG(j >= d->nctx || (a.ctx[j] && a.pose[j]), "sde_photo_fwd: null ctx/pose %d", j);
    }
    a.B = d->B; a.h = d->h; a.w = d->w; a.nctx = d->nctx; a.automask = d->automask; a.reduce_mean = d->reduce_mean;
    a.sx = d->sx; a.sy = d->sy; a.ssim_w = d->ssim_w; a.C1 = d->C1; a.C2 = d->C2;
    dim3 grid(sde_cdiv(d->w, FT_W - 2), sde_cdiv(d->h, FT_H - 2), d->B), blk(FT_W, FT_H);
    const size_t lds = photo_fwd_lds(d->nctx);
    hipStream_t s = (hipStream_t)stream;
    switch (d->nctx) {
        case 1: hipLaunchKernelGGL(photo_fwd_kernel<1>, grid, blk, lds, s, a); break;
        case 2: hipLaunchKernelGGL(photo_fwd_kernel<2>, grid, blk, lds, s, a); break;
        case 3: hipLaunchKernelGGL(photo_fwd_kernel<3>, grid, blk, lds, s, a); break;
        default: hipLaunchKernelGGL(photo_fwd_kernel<4>, grid, blk, lds, s, a); break;
    }
    SDE_CHECK_LAUNCH("sde_photo_fwd");
    const int nblk = grid.x * grid.y * grid.z;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, s, partial, nblk, loss_scale / ((float)d->B * d->h * d->w), loss_out,
                       accumulate);
    SDE_CHECK_LAUNCH("sde_photo_fwd/reduce");
    return SDE_OK;
}

int sde_photo_bwd(const sde_photo_desc* d, const float* const* sampled, const uint8_t* sel, const float* gout, float gscale,
                  float* d_depth, int accumulate_depth, float* pose_partial, float* const* d_pose, int accumulate_pose,
                  sde_stream_t stream) {
    SDE_CHECK_ARG(d && d->A && d->depth && d->K && sampled && gout && d_depth && pose_partial && d_pose, "sde_photo_bwd: null pointer");
    SDE_CHECK_ARG(d->nctx >= 1 && d->nctx <= SDE_MAX_CTX, "sde_photo_bwd: nctx=%d out of range", d->nctx);
    SDE_CHECK_ARG(d->reduce_mean || sel, "sde_photo_bwd: sel required for min reduce");
    PhotoBwdArgs a;
    a.A = d->A; a.depth = d->depth; a.K = d->K; a.sel = sel; a.gout = gout; a.d_depth = d_depth; a.pose_partial = pose_partial;
    for (int j = 0; j < SDE_MAX_CTX; ++j) {
        a.ctx[j] = j < d->nctx ? d->ctx[j] : nullptr;
        a.pose[j] = j < d->nctx ? d->pose[j] : nullptr;
        a.sampled[j] = j < d->nctx ? sampled[j] : nullptr;
        SDE_CHECK_ARG(j >= d->nctx || (a.ctx[j] && a.pose[j] && a.sampled[j] && d_pose[j]), "sde_photo_bwd: null ctx/pose/sampled %d", j);
    }
    a.B = d->B; a.h = d->h; a.w = d->w; a.nctx = d->nctx; a.automask = d->automask; a.reduce_mean = d->reduce_mean;
    a.accumulate = accumulate_depth; a.clip = d->clip_thr != nullptr;
    a.sx = d->sx; a.sy = d->sy; a.ssim_w = d->ssim_w; a.C1 = d->C1; a.C2 = d->C2;
    a.gscale = gscale / ((float)d->B * d->h * d->w);
    dim3 grid(sde_cdiv(d->w, BT_W - 4), sde_cdiv(d->h, BT_H - 4), d->B), blk(BT_W, BT_H);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = photo_bwd_lds();
    const bool dense = (long)grid.x * grid.y * grid.z >= 1024;      // the grid fills three workgroups per CU (768) with a second round to spare
    switch (d->nctx) {
        case 1: hipLaunchKernelGGL(photo_bwd_kernel<1>, grid, blk, lds, s, a); break;
        case 2:
            if (dense) hipLaunchKernelGGL((photo_bwd_kernel<2, true>), grid, blk, lds, s, a);
            else hipLaunchKernelGGL(photo_bwd_kernel<2>, grid, blk, lds, s, a);
            break;
        case 3: hipLaunchKernelGGL(photo_bwd_kernel<3>, grid, blk, lds, s, a); break;
        default: hipLaunchKernelGGL(photo_bwd_kernel<4>, grid, blk, lds, s, a); break;
    }
    SDE_CHECK_LAUNCH("sde_photo_bwd");
    hipLaunchKernelGGL(pose_grad_finalize_kernel, dim3(d->B, d->nctx), dim3(64), 0, s, pose_partial, (int)(grid.x * grid.y), d->nctx, d->B,
                       d_pose[0], d->nctx > 1 ? d_pose[1] : nullptr, d->nctx > 2 ? d_pose[2] : nullptr, d->nctx > 3 ? d_pose[3] : nullptr,
                       accumulate_pose);
    SDE_CHECK_LAUNCH("sde_photo_bwd/finalize");
    return SDE_OK;
}

static int fill_fwd_args(const sde_photo_desc* d, float* const* sampled, uint8_t* sel, float* partial, PhotoArgs& a) {
    SDE_CHECK_ARG(d->A && d->depth && d->K && partial && sel, "sde_photo_multi_fwd: null pointer");
    a.A = d->A; a.depth = d->depth; a.K = d->K; a.sel = sel; a.partial = partial; a.maps = nullptr; a.thr = nullptr;
    for (int j = 0; j < SDE_MAX_CTX; ++j) {
        a.ctx[j] = j < d->nctx ? d->ctx[j] : nullptr;
        a.pose[j] = j < d->nctx ? d->pose[j] : nullptr;
        a.sampled[j] = j < d->nctx ? sampled[j] : nullptr;
        SDE_CHECK_ARG(j >= d->nctx || (a.ctx[j] && a.pose[j] && a.sampled[j]), "sde_photo_multi_fwd: null ctx / pose / sampled %d", j);
    }
    a.B = d->B; a.h = d->h; a.w = d->w; a.nctx = d->nctx; a.automask = d->automask; a.reduce_mean = d->reduce_mean;
    a.sx = d->sx; a.sy = d->sy; a.ssim_w = d->ssim_w; a.C1 = d->C1; a.C2 = d->C2;
    return SDE_OK;
}

static int photo_multi_fwd_launch(const sde_photo_desc* d, int n, float* const* sampled, uint8_t* const* sel, float* const* partial, hipStream_t st, MultiReduce& r) {
    SDE_CHECK_ARG(d && sampled && sel && partial && n >= 1 && n <= PH_MAX_SCALES, "sde_photo_multi_fwd: bad argument (n=%d)", n);
    PhotoMulti m;
    m.n = n; m.first[0] = 0;
    for (int s = 0; s < n; ++s) {
        SDE_CHECK_ARG(d[s].nctx == d[0].nctx && d[s].B == d[0].B && d[s].nctx >= 1 && d[s].nctx <= SDE_MAX_CTX && !d[s].clip_thr && d[s].B > 0 && d[s].h >= 4 && d[s].w >= 4,
                      "sde_photo_multi_fwd: scale %d: same batch / contexts as scale 0, no clip thresholds, h, w >= 4", s);
        int rc = fill_fwd_args(d + s, sampled + s * SDE_MAX_CTX, sel[s], partial[s], m.a[s]);
        if (rc) return rc;
        m.gdx[s] = sde_cdiv(d[s].w, FT_W - 2); m.gdy[s] = sde_cdiv(d[s].h, FT_H - 2);
        m.first[s + 1] = m.first[s] + m.gdx[s] * m.gdy[s] * d[s].B;
        r.partial[s] = partial[s]; r.n[s] = m.gdx[s] * m.gdy[s] * d[s].B; r.scale[s] = 1.0f / ((float)d[s].B * d[s].h * d[s].w);
    }
    for (int s = n; s < PH_MAX_SCALES; ++s) { m.gdx[s] = m.gdy[s] = 1; m.first[s + 1] = m.first[n]; r.partial[s] = nullptr; r.n[s] = 0; r.scale[s] = 0.f; m.a[s] = m.a[0]; }
    const dim3 grid(m.first[n]), blk(FT_W, FT_H);
    const size_t lds = photo_fwd_lds(d[0].nctx);
    switch (d[0].nctx) {
        case 1: hipLaunchKernelGGL(photo_fwd_multi_kernel<1>, grid, blk, lds, st, m); break;
        case 2: hipLaunchKernelGGL(photo_fwd_multi_kernel<2>, grid, blk, lds, st, m); break;
        case 3: hipLaunchKernelGGL(photo_fwd_multi_kernel<3>, grid, blk, lds, st, m); break;
        default: hipLaunchKernelGGL(photo_fwd_multi_kernel<4>, grid, blk, lds, st, m); break;
    }
    SDE_CHECK_LAUNCH("sde_photo_multi_fwd");
    return SDE_OK;
}

int sde_photo_multi_fwd(const sde_photo_desc* d, int n, float* const* sampled, uint8_t* const* sel, float* const* partial, float* loss_out,
                        sde_stream_t stream) {
    SDE_CHECK_ARG(loss_out, "sde_photo_multi_fwd: null loss_out");
    MultiReduce r;
    hipStream_t st = (hipStream_t)stream;
    const int rc = photo_multi_fwd_launch(d, n, sampled, sel, partial, st, r);
    if (rc) return rc;
    hipLaunchKernelGGL(reduce_partials_multi_kernel, dim3(n), dim3(256), 0, st, r, loss_out);
    SDE_CHECK_LAUNCH("sde_photo_multi_fwd/reduce");
    return SDE_OK;
}

// gout + s * gout_stride is scale s's upstream gradient (a device scalar); weight (host, may be null) scales it further
static int photo_multi_bwd_launch(const sde_photo_desc* d, int n, const float* const* sampled, const uint8_t* const* sel, const float* gout, int gout_stride,
                                  const float* weight, float* const* d_depth, float* const* pose_partial, float* const* d_pose, hipStream_t st) {
    // d_pose == NULL: the pose partials are left for sde_photo_multi_pose_finalize (which the caller may enqueue on another stream)
    SDE_CHECK_ARG(d && sampled && sel && gout && d_depth && pose_partial && n >= 1 && n <= PH_MAX_SCALES, "sde_photo_multi_bwd: bad argument (n=%d)", n);
    PhotoBwdMulti m;
    MultiPose mp;
    m.n = mp.n = n; m.first[0] = 0;
    for (int s = 0; s < n; ++s) {
        const sde_photo_desc& ds = d[s];
        SDE_CHECK_ARG(ds.nctx == d[0].nctx && ds.B == d[0].B && ds.nctx >= 1 && ds.nctx <= SDE_MAX_CTX && !ds.clip_thr && ds.A && ds.depth && ds.K && d_depth[s] && pose_partial[s] &&
                      (ds.reduce_mean || sel[s]), "sde_photo_multi_bwd: scale %d: bad descriptor", s);
        PhotoBwdArgs& a = m.a[s];
        a.A = ds.A; a.depth = ds.depth; a.K = ds.K; a.sel = sel[s]; a.gout = gout + s * gout_stride; a.d_depth = d_depth[s]; a.pose_partial = pose_partial[s];
        for (int j = 0; j < SDE_MAX_CTX; ++j) {
            a.ctx[j] = j < ds.nctx ? ds.ctx[j] : nullptr;
            a.pose[j] = j < ds.nctx ? ds.pose[j] : nullptr;
            a.sampled[j] = j < ds.nctx ? sampled[s * SDE_MAX_CTX + j] : nullptr;
            SDE_CHECK_ARG(j >= ds.nctx || (a.ctx[j] && a.pose[j] && a.sampled[j] && (!d_pose || d_pose[j])), "sde_photo_multi_bwd: null ctx / pose / sampled %d", j);
        }
        a.B = ds.B; a.h = ds.h; a.w = ds.w; a.nctx = ds.nctx; a.automask = ds.automask; a.reduce_mean = ds.reduce_mean; a.accumulate = 0; a.clip = 0;
        a.sx = ds.sx; a.sy = ds.sy; a.ssim_w = ds.ssim_w; a.C1 = ds.C1; a.C2 = ds.C2;
        a.gscale = (weight ? weight[s] : 1.0f) / ((float)ds.B * ds.h * ds.w);
        m.gdx[s] = sde_cdiv(ds.w, BT_W - 4); m.gdy[s] = sde_cdiv(ds.h, BT_H - 4);
        m.first[s + 1] = m.first[s] + m.gdx[s] * m.gdy[s] * ds.B;
        mp.partial[s] = pose_partial[s]; mp.blocks_per_sample[s] = m.gdx[s] * m.gdy[s];
    }
    for (int s = n; s < PH_MAX_SCALES; ++s) { m.gdx[s] = m.gdy[s] = 1; m.first[s + 1] = m.first[n]; m.a[s] = m.a[0]; mp.partial[s] = nullptr; mp.blocks_per_sample[s] = 0; }
    const dim3 grid(m.first[n]), blk(BT_W, BT_H);
    const size_t lds = photo_bwd_lds();
    switch (d[0].nctx) {
        case 1: hipLaunchKernelGGL((photo_bwd_multi_kernel<1, false>), grid, blk, lds, st, m); break;
        case 2:
            if (m.first[n] >= 1024) hipLaunchKernelGGL((photo_bwd_multi_kernel<2, true>), grid, blk, lds, st, m);
            else hipLaunchKernelGGL((photo_bwd_multi_kernel<2, false>), grid, blk, lds, st, m);
            break;
        case 3: hipLaunchKernelGGL((photo_bwd_multi_kernel<3, false>), grid, blk, lds, st, m); break;
        default: hipLaunchKernelGGL((photo_bwd_multi_kernel<4, false>), grid, blk, lds, st, m); break;
    }
    SDE_CHECK_LAUNCH("sde_photo_multi_bwd");
    if (!d_pose) return SDE_OK;
    hipLaunchKernelGGL(pose_grad_finalize_multi_kernel, dim3(d[0].B, d[0].nctx), dim3(64), 0, st, mp, d[0].nctx, d[0].B, d_pose[0], d[0].nctx > 1 ? d_pose[1] : nullptr,
                       d[0].nctx > 2 ? d_pose[2] : nullptr, d[0].nctx > 3 ? d_pose[3] : nullptr);
    SDE_CHECK_LAUNCH("sde_photo_multi_bwd/finalize");
    return SDE_OK;
}

int sde_photo_multi_bwd(const sde_photo_desc* d, int n, const float* const* sampled, const uint8_t* const* sel, const float* gout, float* const* d_depth,
                        float* const* pose_partial, float* const* d_pose, sde_stream_t stream) {
    return photo_multi_bwd_launch(d, n, sampled, sel, gout, 1, nullptr, d_depth, pose_partial, d_pose, (hipStream_t)stream);
}

// ---- photometric + smoothness terms of every scale (MonoDepth2.py:L78-126) ----
static int smooth_multi_fill(const sde_photo_desc* d, int n, SmoothMulti& m, const char* who) {
    m.n = n; m.B = d[0].B; m.first[0] = 0; m.accumulate = 0;
    for (int s = 0; s < n; ++s) {
        SDE_CHECK_ARG(d[s].depth && d[s].A && d[s].B == d[0].B && d[s].B > 0 && d[s].h > 1 && d[s].w > 1, "%s: scale %d: bad descriptor", who, s);
        SmoothArgs& a = m.a[s];
        a = SmoothArgs{};
        a.depth = d[s].depth; a.img = d[s].A; a.h = d[s].h; a.w = d[s].w;
        m.gdx[s] = sde_cdiv(d[s].w, 64); m.gdy[s] = sde_cdiv(d[s].h, 4);
        m.first[s + 1] = m.first[s] + m.gdx[s] * m.gdy[s] * d[s].B;
    }
    for (int s = n; s < PH_MAX_SCALES; ++s) { m.a[s] = m.a[0]; m.gdx[s] = m.gdy[s] = 1; m.first[s + 1] = m.first[n]; }
    return SDE_OK;
}

int sde_mono_loss_fwd(const sde_photo_desc* d, int n, const float* photo_w, const float* smooth_w, float* const* sampled, uint8_t* const* sel,
                      float* const* photo_partial, float* const* sm_mean_part, float* const* sm_dn, float* const* sm_loss_part, float* const* sm_s_part,
                      float* per_scale, float* totals, int* ticket, sde_stream_t stream) {
    SDE_CHECK_ARG(d && photo_w && per_scale && totals && ticket && n >= 1 && n <= PH_MAX_SCALES, "sde_mono_loss_fwd: bad argument (n=%d)", n);
    SDE_CHECK_ARG(!smooth_w || (sm_mean_part && sm_dn && sm_loss_part && sm_s_part), "sde_mono_loss_fwd: smoothness weights without smoothness buffers");
    hipStream_t st = (hipStream_t)stream;
    MonoReduce mr;
    mr.nphoto = n; mr.count = smooth_w ? 2 * n : n;
    for (int k = 0; k < 2 * PH_MAX_SCALES; ++k) { mr.partial[k] = nullptr; mr.n[k] = 0; mr.scale[k] = mr.weight[k] = 0.f; }
    if (smooth_w) {
        SmoothMulti m;
        int rc = smooth_multi_fill(d, n, m, "sde_mono_loss_fwd");
        if (rc) return rc;
        for (int s = 0; s < n; ++s) {
            SDE_CHECK_ARG(sm_mean_part[s] && sm_dn[s] && sm_loss_part[s] && sm_s_part[s], "sde_mono_loss_fwd: scale %d: null smoothness buffer", s);
            m.a[s].mean_part = sm_mean_part[s]; m.a[s].dn = sm_dn[s]; m.a[s].loss_part = sm_loss_part[s]; m.a[s].s_part = sm_s_part[s];
            mr.partial[n + s] = sm_loss_part[s]; mr.n[n + s] = m.first[s + 1] - m.first[s]; mr.scale[n + s] = 1.0f; mr.weight[n + s] = smooth_w[s];
        }
        hipLaunchKernelGGL(smooth_mean_multi_kernel, dim3(SM_CHUNKS, d[0].B, n), dim3(256), 0, st, m);
        SDE_CHECK_LAUNCH("sde_mono_loss_fwd/mean");
        hipLaunchKernelGGL(smooth_fwd_multi_kernel, dim3(m.first[n]), dim3(256), 0, st, m);
        SDE_CHECK_LAUNCH("sde_mono_loss_fwd/smooth");
    }
    MultiReduce r;
    const int rc = photo_multi_fwd_launch(d, n, sampled, sel, photo_partial, st, r);
    if (rc) return rc;
    for (int s = 0; s < n; ++s) { mr.partial[s] = r.partial[s]; mr.n[s] = r.n[s]; mr.scale[s] = r.scale[s]; mr.weight[s] = photo_w[s]; }
    hipLaunchKernelGGL(mono_loss_finalize_kernel, dim3(mr.count), dim3(256), 0, st, mr, per_scale, totals, ticket);
    SDE_CHECK_LAUNCH("sde_mono_loss_fwd/finalize");
    return SDE_OK;
}

int sde_mono_loss_bwd(const sde_photo_desc* d, int n, const float* photo_w, const float* smooth_w, const float* const* sampled, const uint8_t* const* sel,
                      const float* g_rec, const float* g_smooth, const float* const* sm_mean_part, const float* const* sm_dn, const float* const* sm_s_part,
                      float* const* d_depth, float* const* pose_partial, float* const* d_pose, sde_stream_t stream) {
    SDE_CHECK_ARG(d && photo_w && g_rec && d_depth && n >= 1 && n <= PH_MAX_SCALES, "sde_mono_loss_bwd: bad argument (n=%d)", n);
    hipStream_t st = (hipStream_t)stream;
    int rc = photo_multi_bwd_launch(d, n, sampled, sel, g_rec, 0, photo_w, d_depth, pose_partial, d_pose, st);      // writes d_depth
    if (rc) return rc;
    if (smooth_w && g_smooth) return sde_smooth_multi_bwd(d, n, smooth_w, g_smooth, sm_mean_part, sm_dn, sm_s_part, d_depth, 1, stream);      // ... the smoothness term adds to it
    return SDE_OK;
}

int sde_smooth_multi_bwd(const sde_photo_desc* d, int n, const float* smooth_w, const float* g_smooth, const float* const* sm_mean_part, const float* const* sm_dn,
                         const float* const* sm_s_part, float* const* d_depth, int accumulate, sde_stream_t stream) {
    SDE_CHECK_ARG(d && smooth_w && g_smooth && sm_mean_part && sm_dn && sm_s_part && d_depth && n >= 1 && n <= PH_MAX_SCALES, "sde_smooth_multi_bwd: bad argument (n=%d)", n);
    SmoothMulti m;
    const int rc = smooth_multi_fill(d, n, m, "sde_smooth_multi_bwd");
    if (rc) return rc;
    m.accumulate = accumulate ? 1 : 0;
    for (int s = 0; s < n; ++s) {
        SDE_CHECK_ARG(sm_mean_part[s] && sm_dn[s] && sm_s_part[s] && d_depth[s], "sde_smooth_multi_bwd: scale %d: null buffer", s);
        m.a[s].mean_part = (float*)sm_mean_part[s]; m.a[s].dn = (float*)sm_dn[s]; m.a[s].s_part = (float*)sm_s_part[s];
        m.a[s].gout = g_smooth; m.a[s].gscale = smooth_w[s]; m.a[s].d_depth = d_depth[s];
    }
    hipLaunchKernelGGL(smooth_bwd_multi_kernel, dim3(m.first[n]), dim3(256), 0, (hipStream_t)stream, m);
    SDE_CHECK_LAUNCH("sde_smooth_multi_bwd");
    return SDE_OK;
}

int sde_photo_multi_pose_finalize(const sde_photo_desc* d, int n, const float* const* pose_partial, float* const* d_pose, sde_stream_t stream) {
    SDE_CHECK_ARG(d && pose_partial && d_pose && n >= 1 && n <= PH_MAX_SCALES, "sde_photo_multi_pose_finalize: bad argument (n=%d)", n);
    MultiPose mp;
    mp.n = n;
    for (int s = 0; s < PH_MAX_SCALES; ++s) {
        const bool on = s < n;
        SDE_CHECK_ARG(!on || (pose_partial[s] && d[s].nctx == d[0].nctx && d[s].B == d[0].B && d[s].h > 0 && d[s].w > 0), "sde_photo_multi_pose_finalize: scale %d: bad descriptor", s);
        mp.partial[s] = on ? pose_partial[s] : nullptr;
        mp.blocks_per_sample[s] = on ? sde_cdiv(d[s].w, BT_W - 4) * sde_cdiv(d[s].h, BT_H - 4) : 0;
    }
    const int nctx = d[0].nctx;
    SDE_CHECK_ARG(nctx >= 1 && nctx <= SDE_MAX_CTX, "sde_photo_multi_pose_finalize: nctx=%d", nctx);
    for (int j = 0; j < nctx; ++j) SDE_CHECK_ARG(d_pose[j], "sde_photo_multi_pose_finalize: null d_pose[%d]", j);
    hipLaunchKernelGGL(pose_grad_finalize_multi_kernel, dim3(d[0].B, nctx), dim3(64), 0, (hipStream_t)stream, mp, nctx, d[0].B, d_pose[0], nctx > 1 ? d_pose[1] : nullptr,
                       nctx > 2 ? d_pose[2] : nullptr, nctx > 3 ? d_pose[3] : nullptr);
    SDE_CHECK_LAUNCH("sde_photo_multi_pose_finalize");
    return SDE_OK;
}

int sde_ssim_fwd(const float* x, const float* y, int B, int C, int H, int W, float C1, float C2, float* out, sde_stream_t stream) {
    SDE_CHECK_ARG(x && y && out && B > 0 && C > 0 && H >= 2 && W >= 2, "sde_ssim_fwd: bad argument (B=%d C=%d H=%d W=%d)", B, C, H, W);
    const int planes = B * C;
    dim3 grid(sde_cdiv(W, 64), sde_cdiv(H, 4), planes < 64 ? planes : 64);
    hipLaunchKernelGGL(ssim_map_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, planes, H, W, C1, C2, out, (const float*)nullptr, (float*)nullptr);
    SDE_CHECK_LAUNCH("sde_ssim_fwd");
    return SDE_OK;
}

int sde_ssim_bwd(const float* x, const float* y, const float* gout, int B, int C, int H, int W, float C1, float C2, float* coef_ws, float* dx, float* dy,
                 sde_stream_t stream) {
    SDE_CHECK_ARG(x && y && gout && coef_ws && (dx || dy) && B > 0 && C > 0 && H >= 2 && W >= 2, "sde_ssim_bwd: bad argument (B=%d C=%d H=%d W=%d)", B, C, H, W);
    SDE_CHECK_ARG(((uintptr_t)coef_ws & 15) == 0, "sde_ssim_bwd: coef_ws must be 16-byte aligned");
    const int planes = B * C;
    dim3 grid(sde_cdiv(W, 64), sde_cdiv(H, 4), planes < 64 ? planes : 64);
    hipLaunchKernelGGL(ssim_map_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, planes, H, W, C1, C2, (float*)nullptr, gout, coef_ws);
    SDE_CHECK_LAUNCH("sde_ssim_bwd/coefficients");
    hipLaunchKernelGGL(ssim_bwd_gather_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, y, coef_ws, planes, H, W, dx, dy);
    SDE_CHECK_LAUNCH("sde_ssim_bwd/gather");
    return SDE_OK;
}

int sde_smooth_num_blocks(int B, int h, int w) { return sde_cdiv(w, 64) * sde_cdiv(h, 4) * B; }

int sde_smooth_fwd(const float* depth, const float* img, int B, int h, int w, float* mean_part, float* dn, float* loss_part, float* s_part,
                   float* loss_out, float loss_scale, int accumulate, sde_stream_t stream) {
    SDE_CHECK_ARG(depth && img && mean_part && loss_part && loss_out, "sde_smooth_fwd: null pointer");
    SDE_CHECK_ARG(B > 0 && h > 1 && w > 1, "sde_smooth_fwd: bad shape");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(smooth_mean_kernel, dim3(SM_CHUNKS, B), dim3(256), 0, s, depth, h * w, mean_part);
    SDE_CHECK_LAUNCH("sde_smooth_fwd/mean");
    dim3 grid(sde_cdiv(w, 64), sde_cdiv(h, 4), B);
    hipLaunchKernelGGL(smooth_fwd_kernel, grid, dim3(256), 0, s, depth, img, mean_part, B, h, w, dn, loss_part, s_part);
    SDE_CHECK_LAUNCH("sde_smooth_fwd");
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, s, loss_part, (int)(grid.x * grid.y * grid.z), loss_scale, loss_out,
                       accumulate);
    SDE_CHECK_LAUNCH("sde_smooth_fwd/reduce");
    return SDE_OK;
}

int sde_smooth_bwd(const float* depth, const float* dn, const float* mean_part, const float* s_part, const float* gout, float gscale, int B,
                   int h, int w, float* d_depth, int accumulate, sde_stream_t stream) {
    SDE_CHECK_ARG(depth && dn && mean_part && s_part && gout && d_depth, "sde_smooth_bwd: null pointer");
    dim3 grid(sde_cdiv(w, 64), sde_cdiv(h, 4), B);
    hipLaunchKernelGGL(smooth_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, depth, dn, mean_part, s_part, (int)(grid.x * grid.y), gout,
                       gscale, h, w, d_depth, accumulate);
    SDE_CHECK_LAUNCH("sde_smooth_bwd");
    return SDE_OK;
}

int sde_silog_num_blocks(int B, int h, int w) {
    const long n = (long)B * h * w;
    const long nb = (n + 1023) / 1024;
    return (int)(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb));
}

int sde_silog_fwd(const float* est, const float* gt, int B, int h, int w, int H, int W, float variance_focus, float* part, float* stats,
                  sde_stream_t stream) {
    SDE_CHECK_ARG(est && gt && part && stats, "sde_silog_fwd: null pointer");
    SDE_CHECK_ARG(B > 0 && h > 0 && w > 0 && H >= h && W >= w, "sde_silog_fwd: bad shape");
    const int nb = sde_silog_num_blocks(B, h, w);
    const float sh = (float)((double)H / (double)h), sw = (float)((double)W / (double)w);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(silog_fwd_kernel, dim3(nb), dim3(256), 0, s, est, gt, B, h, w, H, W, sh, sw, part);
    SDE_CHECK_LAUNCH("sde_silog_fwd");
    hipLaunchKernelGGL(silog_finalize_kernel, dim3(1), dim3(64), 0, s, part, nb, variance_focus, stats);
    SDE_CHECK_LAUNCH("sde_silog_fwd/finalize");
    return SDE_OK;
}

int sde_silog_bwd(const float* est, const float* gt, const float* stats, const float* gout, float gscale, float variance_focus, int B, int h,
                  int w, int H, int W, float* d_est, int accumulate, sde_stream_t stream) {
    SDE_CHECK_ARG(est && gt && stats && gout && d_est, "sde_silog_bwd: null pointer");
    const int nb = sde_silog_num_blocks(B, h, w);
    const float sh = (float)((double)H / (double)h), sw = (float)((double)W / (double)w);
    hipLaunchKernelGGL(silog_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, est, gt, stats, gout, gscale, variance_focus, B, h, w, H,
                       W, sh, sw, d_est, accumulate);
    SDE_CHECK_LAUNCH("sde_silog_bwd");
    return SDE_OK;
}

static int silog_scales(const float* const* est, float* const* d_est, const int* h, const int* w, const float* weight, int n, int B, int H, int W,
                        SilogScales& a) {
    SDE_CHECK_ARG(est && h && w && weight && n >= 1 && n <= SDE_SILOG_MAX_SCALES && B > 0, "sde_silog_multi: bad scale table (n=%d)", n);
    a.n = n;
    int end = 0;
    for (int k = 0; k < n; ++k) {
        SDE_CHECK_ARG(est[k] && h[k] > 0 && w[k] > 0 && H >= h[k] && W >= w[k] && (!d_est || d_est[k]), "sde_silog_multi: bad scale %d", k);
        a.est[k] = est[k]; a.d_est[k] = d_est ? d_est[k] : nullptr; a.h[k] = h[k]; a.w[k] = w[k]; a.weight[k] = weight[k];
        a.sh[k] = (float)((double)H / (double)h[k]); a.sw[k] = (float)((double)W / (double)w[k]);
        end += sde_silog_num_blocks(B, h[k], w[k]);
        a.blk_end[k] = end;
    }
    return SDE_OK;
}

int sde_silog_multi_num_blocks(int B, const int* h, const int* w, int n) {
    int end = 0;
    for (int k = 0; k < n; ++k) end += sde_silog_num_blocks(B, h[k], w[k]);
    return end;
}

int sde_silog_multi_fwd(const float* const* est, const float* gt, int B, const int* h, const int* w, const float* weight, int n, int H, int W,
                        float variance_focus, float* part, float* stats, float* total, sde_stream_t stream) {
    SDE_CHECK_ARG(gt && part && stats && total, "sde_silog_multi_fwd: null pointer");
    SilogScales a;
    const int rc = silog_scales(est, nullptr, h, w, weight, n, B, H, W, a);
    if (rc != SDE_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(silog_multi_fwd_kernel, dim3(a.blk_end[n - 1]), dim3(256), 0, s, a, gt, B, H, W, part);
    SDE_CHECK_LAUNCH("sde_silog_multi_fwd");
    hipLaunchKernelGGL(silog_multi_finalize_kernel, dim3(1), dim3(64 * n), 0, s, part, a, variance_focus, stats, total);
    SDE_CHECK_LAUNCH("sde_silog_multi_fwd/finalize");
    return SDE_OK;
}

int sde_silog_multi_bwd(const float* const* est, const float* gt, const float* stats, const float* gout, float gscale, float variance_focus, int B,
                        const int* h, const int* w, const float* weight, int n, int H, int W, float* const* d_est, sde_stream_t stream) {
    SDE_CHECK_ARG(gt && stats && gout && d_est, "sde_silog_multi_bwd: null pointer");
    SilogScales a;
    const int rc = silog_scales(est, d_est, h, w, weight, n, B, H, W, a);
    if (rc != SDE_OK) return rc;
    hipLaunchKernelGGL(silog_multi_bwd_kernel, dim3(a.blk_end[n - 1]), dim3(256), 0, (hipStream_t)stream, a, gt, stats, gout, gscale, variance_focus, B, H, W);
    SDE_CHECK_LAUNCH("sde_silog_multi_bwd");
    return SDE_OK;
}

int sde_pose_vec2mat(const float* vec, float* mat, int n, sde_stream_t stream) {
    SDE_CHECK_ARG(vec && mat && n > 0, "sde_pose_vec2mat: bad argument");
    hipLaunchKernelGGL(pose_vec2mat_kernel, dim3(sde_cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, vec, mat, n);
    SDE_CHECK_LAUNCH("sde_pose_vec2mat");
    return SDE_OK;
}

int sde_pose_vec2mat_bwd(const float* vec, const float* dmat, float* dvec, int n, sde_stream_t stream) {
    SDE_CHECK_ARG(vec && dmat && dvec && n > 0, "sde_pose_vec2mat_bwd: bad argument");
    hipLaunchKernelGGL(pose_vec2mat_bwd_kernel, dim3(sde_cdiv(n, 64)), dim3(64), 0, (hipStream_t)stream, vec, dmat, dvec, n);
    SDE_CHECK_LAUNCH("sde_pose_vec2mat_bwd");
    return SDE_OK;
}

}  // extern "C"
