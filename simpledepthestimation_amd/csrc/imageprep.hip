// Device-side input pipeline for uint8 camera frames ("uint8 H2D + GPU resize / colour-jitter kernel").
//
// Replaces, for the image entries of a batch, the CPU chain of detectron2/data/preprocess/augmentation.py:L124-166 (`Resize`: cv2.resize INTER_LINEAR on
// uint8), L229-266 (`RandomImageAug`: torchvision ColorJitter on PIL images -- brightness, contrast, saturation, hue in a random order, ONE parameter set
// for a sample's target and context frames) and formating.py `ToTensor` (HWC uint8 -> CHW float / 255), in the same INTEGER arithmetic, so that the frames
// the kernels produce are the frames the CPU pipeline of this package produces, bit for bit:
//   * resize: OpenCV's published fixed-point bilinear (11-bit coefficients; data/preprocess/augmentation.py: resize_linear_u8);
//   * ImageEnhance.Brightness / Contrast / Color = Image.blend(degenerate, image, factor): out = (uint8)(d + f * (x - d)) in float32, truncating
//     (clipped first when f is outside [0, 1]); degenerate = black / the rounded mean luminance of the image AT THAT POINT of the chain /
//     the pixel's luminance, L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16;
//   * hue: Pillow's RGB -> HSV -> RGB conversions (float / double mix of Convert.c, checked exhaustively over all 2^24 triples against Pillow on
//     the CPU: tests/test_data.py) with H shifted by uint8(hue_factor * 255), wrapping.
// The contrast step needs a whole-image mean in the middle of the per-pixel chain: pass 0 runs the chain up to the contrast step and sums L
// per frame (integer atomics: exact, order-independent), pass 1 runs the whole chain.  Outputs: the jittered frame and the un-jittered frame
// (img / img_orig of the batch dict), fp32 NCHW in [0, 1].
//
// HBM-bound and tiny: 3 B in (source pixels, each read ~once through L2) + 24 B out per output pixel.
// This file is compiled with -ffp-contract=off: the blends must round like Pillow's (no FMA).
#include "common.h"
#include "sde_hip.h"

namespace {

constexpr int PREP_MAX_GROUPS = 4;
struct PrepArgs {
    // G groups of N frames each (a MonoDepth2 batch: the target frames and the two context-frame tensors) share the size pair, the tap tables and the
    // per-sample jitter parameters; blockIdx.z = group * N + sample
    const uint8_t* src[PREP_MAX_GROUPS];          // [N][Hs][Ws][3]
    const int* xtab;             // [w][4]: x0, x1, a0, a1   (OpenCV INTER_LINEAR taps / 11-bit weights, host-built once per (Ws, w))
    const int* ytab;             // [h][4]: y0, y1, b0, b1
    const float* jit;            // [N][8]: brightness, contrast, saturation, hue factors, then the order of the four steps (0..3 as floats; < 0: no jitter)
    unsigned* lsum;              // [G][N] workspace: sum of L over the frame at the contrast step
    float* img[PREP_MAX_GROUPS];                  // [N][3][h][w] jittered / 255
    float* orig[PREP_MAX_GROUPS];                 // [N][3][h][w] un-jittered / 255 (may be null)
    int N, Hs, Ws, h, w;
};

__device__ __forceinline__ int lum(int r, int g, int b) { return (r * 19595 + g * 38470 + b * 7471 + 0x8000) >> 16; }

// Image.blend(degenerate, image, f) for one channel value (Blend.c): float32 arithmetic, truncation; outside [0, 1] clip first
__device__ __forceinline__ int blend1(int d, int x, float f) {
    const float t = (float)d + f * (float)(x - d);
    if (f >= 0.f && f <= 1.0f) return (int)(uint8_t)t;
    if (t <= 0.f) return 0;
    if (t >= 255.f) return 255;
    return (int)(uint8_t)t;
}

// Convert.c rgb2hsv_row / hsv2rgb with the hue shift in between (functional_pil.adjust_hue)
__device__ __forceinline__ void hue_shift(int& r, int& g, int& b, int shift) {
    const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
    int uh = 0, us = 0;
    const int uv = maxc;
    if (minc != maxc) {
        const float cr = (float)(maxc - minc);
        const float s = cr / (float)maxc;
        const float rc = (float)(maxc - r) / cr, gc = (float)(maxc - g) / cr, bc = (float)(maxc - b) / cr;
        float hf;
        if (r == maxc) hf = (float)((double)bc - (double)gc);
        else if (g == maxc) hf = (float)(2.0 + (double)rc - (double)bc);
        else hf = (float)(4.0 + (double)gc - (double)rc);
        hf = (float)fmod((double)hf / 6.0 + 1.0, 1.0);
        uh = min(max((int)((double)hf * 255.0), 0), 255);
        us = min(max((int)((double)s * 255.0), 0), 255);
    }
    uh = (uh + shift) & 255;
    if (us == 0) { r = g = b = uv; return; }
    const double hv = (double)(float)uh * 6.0 / 255.0;
    const int i = (int)floor(hv);
    const float f = (float)(hv - (double)(float)i);
    const float fs = (float)((double)(float)us / 255.0);
    const double vf = (double)(float)uv;
    const int p = min(max((int)round(vf * (1.0 - (double)fs)), 0), 255);
    const int q = min(max((int)round(vf * (1.0 - (double)fs * (double)f)), 0), 255);
    const int t = min(max((int)round(vf * (1.0 - (double)fs * (1.0 - (double)f))), 0), 255);
    switch (i % 6) {
        case 0: r = uv; g = t; b = p; break;
        case 1: r = q; g = uv; b = p; break;
        case 2: r = p; g = uv; b = t; break;
        case 3: r = p; g = q; b = uv; break;
        case 4: r = t; g = p; b = uv; break;
        default: r = uv; g = p; b = q; break;
    }
}

// one output pixel of the fixed-point bilinear resize (resize_linear_u8)
__device__ __forceinline__ void resize_px(const PrepArgs& a, const uint8_t* frame, int y, int x, int& r, int& g, int& b) {
    const int4 xt = reinterpret_cast<const int4*>(a.xtab)[x], yt = reinterpret_cast<const int4*>(a.ytab)[y];
    const uint8_t* r0 = frame + (size_t)yt.x * a.Ws * 3;
    const uint8_t* r1 = frame + (size_t)yt.y * a.Ws * 3;
    int o[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int s0 = (int)r0[xt.x * 3 + c] * xt.z + (int)r0[xt.y * 3 + c] * xt.w;
        const int s1 = (int)r1[xt.x * 3 + c] * xt.z + (int)r1[xt.y * 3 + c] * xt.w;
        const int v = (((yt.z * (s0 >> 4)) >> 16) + ((yt.w * (s1 >> 4)) >> 16) + 2) >> 2;
        o[c] = min(max(v, 0), 255);
    }
    r = o[0]; g = o[1]; b = o[2];
}

// steps [first, last) of the frame's jitter chain; `mean` = the contrast step's degenerate value
__device__ __forceinline__ void jitter_steps(const float* jp, int first, int last, int mean, int& r, int& g, int& b) {
    for (int k = first; k < last; ++k) {
        const int fn = (int)jp[4 + k];
        if (fn == 0) { r = blend1(0, r, jp[0]); g = blend1(0, g, jp[0]); b = blend1(0, b, jp[0]); }
        else if (fn == 1) { r = blend1(mean, r, jp[1]); g = blend1(mean, g, jp[1]); b = blend1(mean, b, jp[1]); }
        else if (fn == 2) { const int l = lum(r, g, b); r = blend1(l, r, jp[2]); g = blend1(l, g, jp[2]); b = blend1(l, b, jp[2]); }
        else { const int sh = ((int)((double)jp[3] * 255.0)) & 255; hue_shift(r, g, b, sh); }     // int(hue_factor * 255) % 256: truncate towards zero, wrap
    }
}

__device__ __forceinline__ int contrast_pos(const float* jp) {
    for (int k = 0; k < 4; ++k)
        if ((int)jp[4 + k] == 1) return k;
    return 4;
}

// Image.blend's shortcuts: factor == 1.0 returns the image itself (blend1 gives the same value: d + 1 * (x - d) = x exactly), factor == 0.0 the
// degenerate image (d + 0 * (x - d) = d exactly) -- no special case needed.

// pass 0: sum of L over the frame after the steps in front of the contrast step.  One workgroup walks P0_ROWS rows of a 64-column strip, so a frame
// takes a few hundred atomics on its counter instead of one per 64 x 4 tile (7200 same-address atomics per frame serialised: 73 us for 12 frames).
constexpr int P0_ROWS = 32;
__global__ void __launch_bounds__(256) image_prep_mean_kernel(PrepArgs a) {
    __shared__ unsigned red[4];
    const int grp = blockIdx.z / a.N, n = blockIdx.z % a.N;
    const float* jp = a.jit + n * 8;
    const bool jitter = jp[4] >= 0.f;
    const int cpos = jitter ? contrast_pos(jp) : 4;
    if (!(jitter && cpos < 4)) return;                       // uniform per frame: no contrast step, nothing to sum
    const uint8_t* frame = a.src[grp] + (size_t)n * a.Hs * a.Ws * 3;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    unsigned l = 0;
    if (x < a.w)
        for (int y = blockIdx.y * P0_ROWS + (threadIdx.x >> 6); y < min(a.h, (int)(blockIdx.y + 1) * P0_ROWS); y += 4) {
            int r, g, b;
            resize_px(a, frame, y, x, r, g, b);
            jitter_steps(jp, 0, cpos, 0, r, g, b);
            l += (unsigned)lum(r, g, b);
        }
    for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(a.lsum + blockIdx.z, red[0] + red[1] + red[2] + red[3]);      // integer: exact and order-independent
}

__global__ void __launch_bounds__(256) image_prep_kernel(PrepArgs a) {
    const int grp = blockIdx.z / a.N, n = blockIdx.z % a.N;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (!(x < a.w && y < a.h)) return;
    const float* jp = a.jit + n * 8;
    const bool jitter = jp[4] >= 0.f;
    const int cpos = jitter ? contrast_pos(jp) : 4;
    const uint8_t* frame = a.src[grp] + (size_t)n * a.Hs * a.Ws * 3;
    int r, g, b;
    resize_px(a, frame, y, x, r, g, b);
    const size_t hw = (size_t)a.h * a.w, o0 = (size_t)n * 3 * hw + (size_t)y * a.w + x;
    float* __restrict__ orig = a.orig[grp];
    float* __restrict__ img = a.img[grp];
    if (orig) { orig[o0] = (float)r / 255.0f; orig[o0 + hw] = (float)g / 255.0f; orig[o0 + 2 * hw] = (float)b / 255.0f; }
    if (jitter) {
        // ImageStat.Stat(L image).mean[0] = sum / count in double; int(mean + 0.5)
        const int mean = cpos < 4 ? (int)((double)a.lsum[blockIdx.z] / (double)(a.h * a.w) + 0.5) : 0;
        jitter_steps(jp, 0, 4, mean, r, g, b);
    }
    img[o0] = (float)r / 255.0f; img[o0 + hw] = (float)g / 255.0f; img[o0 + 2 * hw] = (float)b / 255.0f;
}

}  // namespace

extern "C" {

int sde_image_prep_u8_multi(const uint8_t* const* src, int G, int N, int Hs, int Ws, int h, int w, const int* xtab, const int* ytab, const float* jit, unsigned* lsum,
                            float* const* img, float* const* orig, sde_stream_t stream) {
    SDE_CHECK_ARG(src && img && G >= 1 && G <= PREP_MAX_GROUPS && xtab && ytab && jit && lsum && N > 0 && (long)N * G <= 65535 && Hs > 0 && Ws > 0 && h > 0 && w > 0,
                  "sde_image_prep_u8: bad argument (G=%d N=%d)", G, N);
    SDE_CHECK_ARG((long)Hs * Ws * 3 * N < 0x7fffffffL && (((uintptr_t)xtab | (uintptr_t)ytab) & 15) == 0, "sde_image_prep_u8: batch too large or unaligned tap tables");
    PrepArgs a{};
    a.xtab = xtab; a.ytab = ytab; a.jit = jit; a.lsum = lsum; a.N = N; a.Hs = Hs; a.Ws = Ws; a.h = h; a.w = w;
    for (int g = 0; g < PREP_MAX_GROUPS; ++g) {
        SDE_CHECK_ARG(g >= G || (src[g] && img[g]), "sde_image_prep_u8: null source / destination of group %d", g);
        a.src[g] = g < G ? src[g] : nullptr; a.img[g] = g < G ? img[g] : nullptr; a.orig[g] = (g < G && orig) ? orig[g] : nullptr;
    }
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(lsum, 0, sizeof(unsigned) * N * G, s) != hipSuccess) { sde_set_error("sde_image_prep_u8: memset failed"); return SDE_ERR_LAUNCH; }
    hipLaunchKernelGGL(image_prep_mean_kernel, dim3((unsigned)sde_cdiv(w, 64), (unsigned)sde_cdiv(h, P0_ROWS), (unsigned)(N * G)), dim3(256), 0, s, a);
    SDE_CHECK_LAUNCH("sde_image_prep_u8/mean");
    hipLaunchKernelGGL(image_prep_kernel, dim3((unsigned)sde_cdiv(w, 64), (unsigned)sde_cdiv(h, 4), (unsigned)(N * G)), dim3(256), 0, s, a);
    SDE_CHECK_LAUNCH("sde_image_prep_u8");
    return SDE_OK;
}

int sde_image_prep_u8(const uint8_t* src, int N, int Hs, int Ws, int h, int w, const int* xtab, const int* ytab, const float* jit, unsigned* lsum, float* img,
                      float* orig, sde_stream_t stream) {
    SDE_CHECK_ARG(src && img, "sde_image_prep_u8: null source / destination");
    return sde_image_prep_u8_multi(&src, 1, N, Hs, Ws, h, w, xtab, ytab, jit, lsum, &img, orig ? &orig : nullptr, stream);
}

}  // extern "C"
