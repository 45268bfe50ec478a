// PackNet01's data movement and inverse-depth head on NHWC tensors (reference: detectron2/layers/layers01.py:L105-160 `InvDepth`, `packing`;
// L262-298 `UnpackLayerConv3d` (nn.PixelShuffle); detectron2/modeling/depth_net/PackNet01.py:L150-199 torch.cat of (unpacked, skip, up-sampled
// inverse depth), `scale_inv_depth`).  The reference does each of these with a chain of view / permute / contiguous / cat / interpolate calls;
// here each is one bandwidth-bound kernel moving 16 bytes per lane, with its exact inverse as the backward:
//
//   space_to_depth  out[b, y, x, c*4 + dy*2 + dx] = in[b, 2y+dy, 2x+dx, c]          (packing, r = 2: layers01.py:L138-160)
//   depth_to_space  out[b, 2y+dy, 2x+dx, c]       = in[b, y, x, c*4 + dy*2 + dx]    (nn.PixelShuffle(2) on NHWC)
//   concat          out[b, y, x, :] = [p0 (+ p1 | , p1)] ++ [nearest_x2(inv_depth)[b, y, x]] ++ zero fill to the 16-byte group
//   inv_depth_head  inv = sigmoid(logit) / min_depth_head ; depth = 1 / (1/max_depth + (1/min_depth - 1/max_depth) * inv)  (+ horizontal flip)
//
// HBM-bound: algorithmic bytes = one read + one write of the tensor (2 x 2 B per element in 16-bit storage).
#include "common.h"
#include "sde_hip.h"

namespace {

// 16 bytes of elements, as raw bits (these kernels only move / zero / add elements)
template <int ES> struct Bits;
template <> struct Bits<2> { typedef unsigned short E; static constexpr int V = 8; };
template <> struct Bits<4> { typedef unsigned int E; static constexpr int V = 4; };

template <int ES>
__global__ void __launch_bounds__(256) space_to_depth_kernel(const void* __restrict__ in_, void* __restrict__ out_, int B, int H, int W, int C) {
    typedef typename Bits<ES>::E E;
    constexpr int V = Bits<ES>::V;
    const E* in = (const E*)in_;
    E* out = (E*)out_;
    const int oh = H / 2, ow = W / 2, cch = C / V;
    const long total = (long)B * oh * ow * cch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % cch);
        const long pix = i / cch;
        const int x = (int)(pix % ow), y = (int)((pix / ow) % oh), b = (int)(pix / ((long)ow * oh));
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            v[k] = *reinterpret_cast<const uint4*>(in + (((long)b * H + 2 * y + (k >> 1)) * W + 2 * x + (k & 1)) * C + cg * V);
        E o[4 * V];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const E* e = reinterpret_cast<const E*>(&v[k]);
#pragma unroll
            for (int c = 0; c < V; ++c) o[c * 4 + k] = e[c];
        }
        uint4* dst = reinterpret_cast<uint4*>(out + pix * (4L * C) + (long)cg * 4 * V);
#pragma unroll
        for (int k = 0; k < 4; ++k) dst[k] = reinterpret_cast<const uint4*>(o)[k];
    }
}

// in [B,H,W,C4] -> out [B,2H,2W,C4/4]
template <int ES>
__global__ void __launch_bounds__(256) depth_to_space_kernel(const void* __restrict__ in_, void* __restrict__ out_, int B, int H, int W, int C4) {
    typedef typename Bits<ES>::E E;
    constexpr int V = Bits<ES>::V;
    const E* in = (const E*)in_;
    E* out = (E*)out_;
    const int C = C4 / 4, cch = C / V;
    const long total = (long)B * H * W * cch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % cch);
        const long pix = i / cch;
        const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
        E l[4 * V];
        const uint4* src = reinterpret_cast<const uint4*>(in + pix * (long)C4 + (long)cg * 4 * V);
#pragma unroll
        for (int k = 0; k < 4; ++k) reinterpret_cast<uint4*>(l)[k] = src[k];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            E e[V];
#pragma unroll
            for (int c = 0; c < V; ++c) e[c] = l[c * 4 + k];
            *reinterpret_cast<uint4*>(out + (((long)b * 2 * H + 2 * y + (k >> 1)) * (2L * W) + 2 * x + (k & 1)) * C + cg * V) = *reinterpret_cast<const uint4*>(e);
        }
    }
}

template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int V = 4; };
template <> struct VecOf<bf16_t> { static constexpr int V = 8; };
template <> struct VecOf<half_t> { static constexpr int V = 8; };

// out [B,H,W,Ct]: groups [0, C0/V) from p0 (+ p1 when add), then C1/V groups from p1 (concat mode), then -- when inv is given -- one group whose first
// element is inv[b, y/2, x/2] (nearest x2 of the [B,H/2,W/2] fp32 map) and whose rest is zero
template <typename T>
__global__ void __launch_bounds__(256) concat_fwd_kernel(const T* __restrict__ p0, const T* __restrict__ p1, const float* __restrict__ inv, int add, int B, int H,
                                                         int W, int C0, int C1, int Ct, T* __restrict__ out) {
    constexpr int V = VecOf<T>::V;
    const int g0 = C0 / V, g1 = add ? 0 : C1 / V, gt = Ct / V;
    const long total = (long)B * H * W * gt;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int g = (int)(i % gt);
        const long pix = i / gt;
        T* dst = out + pix * Ct + (long)g * V;
        if (g < g0) {
            if (add) {
                const uint4 a = *reinterpret_cast<const uint4*>(p0 + pix * C0 + (long)g * V), b = *reinterpret_cast<const uint4*>(p1 + pix * C0 + (long)g * V);
                const T* ea = reinterpret_cast<const T*>(&a); const T* eb = reinterpret_cast<const T*>(&b);
                T o[V];
#pragma unroll
                for (int e = 0; e < V; ++e) o[e] = (T)((float)ea[e] + (float)eb[e]);
                *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(o);
            } else *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(p0 + pix * C0 + (long)g * V);
        } else if (g < g0 + g1) {
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(p1 + pix * C1 + (long)(g - g0) * V);
        } else {
            T o[V];
#pragma unroll
            for (int e = 0; e < V; ++e) o[e] = (T)0.f;
            if (inv && g == g0 + g1) {
                const int x = (int)(pix % W), y = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
                o[0] = (T)inv[((long)b * (H / 2) + (y >> 1)) * (W / 2) + (x >> 1)];
            }
            *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(o);
        }
    }
}

// d0 [B,H,W,C0] = dout[..., :C0]; d1 [B,H,W,C1] = dout[..., C0:C0+C1] (concat mode; in add mode the caller hands d0 to both inputs)
template <typename T>
__global__ void __launch_bounds__(256) concat_bwd_kernel(const T* __restrict__ dout, int B, int H, int W, int C0, int C1, int Ct, T* __restrict__ d0,
                                                         T* __restrict__ d1) {
    constexpr int V = VecOf<T>::V;
    const int g0 = C0 / V, g1 = d1 ? C1 / V : 0, gs = g0 + g1;
    const long total = (long)B * H * W * gs;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int g = (int)(i % gs);
        const long pix = i / gs;
        const uint4 v = *reinterpret_cast<const uint4*>(dout + pix * Ct + (long)g * V);
        if (g < g0) *reinterpret_cast<uint4*>(d0 + pix * C0 + (long)g * V) = v;
        else *reinterpret_cast<uint4*>(d1 + pix * C1 + (long)(g - g0) * V) = v;
    }
}

// d_inv[b, y, x] = sum over the 2x2 cell of dout[b, 2y+dy, 2x+dx, ch]   (backward of the nearest x2 up-sampling into channel ch)
template <typename T>
__global__ void __launch_bounds__(256) concat_bwd_inv_kernel(const T* __restrict__ dout, int B, int h, int w, int Ct, int ch, float* __restrict__ d_inv) {
    const long total = (long)B * h * w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x = (int)(i % w), y = (int)((i / w) % h), b = (int)(i / ((long)w * h));
        const T* r0 = dout + (((long)b * 2 * h + 2 * y) * (2L * w) + 2 * x) * Ct + ch;
        const T* r1 = r0 + 2L * w * Ct;
        d_inv[i] = ((float)r0[0] + (float)r0[Ct]) + ((float)r1[0] + (float)r1[Ct]);
    }
}

// logit = y[..., 0] -> inv [B,H,W] = sigmoid(logit) / md_head ; depth [B,1,H,W] = 1 / (lo + (hi - lo) * inv), flipped along x when flip
template <typename T>
__global__ void __launch_bounds__(256) inv_depth_head_fwd_kernel(const T* __restrict__ y, int B, int H, int W, int ld, float md_head, float lo, float hi, int flip,
                                                                 float* __restrict__ inv, float* __restrict__ depth) {
    const long n = (long)B * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = (float)y[i * ld];
        const float sg = 1.0f / (1.0f + expf(-v));
        const float d = sg / md_head;
        inv[i] = d;
        long o = i;
        if (flip) { const int x = (int)(i % W); o = i - x + (W - 1 - x); }
        depth[o] = 1.0f / (lo + (hi - lo) * d);
    }
}

// dy[..., 0] = (d_inv + d_depth * d depth / d inv) * sigmoid'(logit) / md_head; the other channels of the padded group are zero
template <typename T>
__global__ void __launch_bounds__(256) inv_depth_head_bwd_kernel(const T* __restrict__ y, const float* __restrict__ d_inv, const float* __restrict__ d_depth, int B,
                                                                 int H, int W, int ld, float md_head, float lo, float hi, int flip, T* __restrict__ dy) {
    const long n = (long)B * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = (float)y[i * ld];
        const float sg = 1.0f / (1.0f + expf(-v));
        const float d = sg / md_head;
        float g = d_inv ? d_inv[i] : 0.f;
        if (d_depth) {
            long o = i;
            if (flip) { const int x = (int)(i % W); o = i - x + (W - 1 - x); }
            const float sd = lo + (hi - lo) * d;
            g += d_depth[o] * (-(hi - lo) / (sd * sd));
        }
        T* dst = dy + i * ld;
        dst[0] = (T)(g * (sg * (1.0f - sg) / md_head));
        for (int c = 1; c < ld; ++c) dst[c] = (T)0.f;
    }
}

int grid_for(long n) {
    long nb = (n + 255) / 256;
    if (nb > 8192) nb = 8192;
    return (int)(nb < 1 ? 1 : nb);
}

}  // namespace

#define PK_DISPATCH(dtype, F32, BF, HF)  do { if ((dtype) == SDE_F32) { F32; } else if ((dtype) == SDE_BF16) { BF; } else { HF; } } while (0)

extern "C" {

int sde_space_to_depth(const void* x, int B, int H, int W, int C, int dtype, void* y, sde_stream_t stream) {
    const int V = SDE_IS16(dtype) ? 8 : 4;
    SDE_CHECK_ARG(x && y && SDE_DTYPE_OK(dtype) && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % V == 0,
                  "sde_space_to_depth: bad argument (H=%d W=%d must be even, C=%d a multiple of %d)", H, W, C, V);
    const int nb = grid_for((long)B * (H / 2) * (W / 2) * (C / V));
    if (SDE_IS16(dtype)) hipLaunchKernelGGL(space_to_depth_kernel<2>, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C);
    else hipLaunchKernelGGL(space_to_depth_kernel<4>, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C);
    SDE_CHECK_LAUNCH("sde_space_to_depth");
    return SDE_OK;
}

int sde_depth_to_space(const void* x, int B, int H, int W, int C4, int dtype, void* y, sde_stream_t stream) {
    const int V = SDE_IS16(dtype) ? 8 : 4;
    SDE_CHECK_ARG(x && y && SDE_DTYPE_OK(dtype) && B > 0 && H > 0 && W > 0 && C4 > 0 && C4 % (4 * V) == 0,
                  "sde_depth_to_space: bad argument (C=%d must be a multiple of %d)", C4, 4 * V);
    const int nb = grid_for((long)B * H * W * (C4 / 4 / V));
    if (SDE_IS16(dtype)) hipLaunchKernelGGL(depth_to_space_kernel<2>, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C4);
    else hipLaunchKernelGGL(depth_to_space_kernel<4>, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C4);
    SDE_CHECK_LAUNCH("sde_depth_to_space");
    return SDE_OK;
}

int sde_concat_fwd(const void* p0, const void* p1, const float* inv, int add, int B, int H, int W, int C0, int C1, int Ct, int dtype, void* out,
                   sde_stream_t stream) {
    const int V = SDE_IS16(dtype) ? 8 : 4;
    const int used = C0 + (add ? 0 : C1) + (inv ? 1 : 0);
    SDE_CHECK_ARG(p0 && out && SDE_DTYPE_OK(dtype) && B > 0 && H > 0 && W > 0 && C0 > 0 && C0 % V == 0 && C1 >= 0 && C1 % V == 0 && Ct % V == 0 && Ct >= used && Ct < used + V,
                  "sde_concat_fwd: bad channel counts (C0=%d C1=%d Ct=%d)", C0, C1, Ct);
    SDE_CHECK_ARG((C1 == 0) == (p1 == nullptr) && (!add || C1 == C0), "sde_concat_fwd: second source / add mode mismatch");
    SDE_CHECK_ARG(!inv || (H % 2 == 0 && W % 2 == 0), "sde_concat_fwd: the inverse-depth map is up-sampled x2: H, W must be even");
    const int nb = grid_for((long)B * H * W * (Ct / V));
    hipStream_t s = (hipStream_t)stream;
    PK_DISPATCH(dtype,
                hipLaunchKernelGGL(concat_fwd_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)p0, (const float*)p1, inv, add, B, H, W, C0, C1, Ct, (float*)out),
                hipLaunchKernelGGL(concat_fwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)p0, (const bf16_t*)p1, inv, add, B, H, W, C0, C1, Ct, (bf16_t*)out),
                hipLaunchKernelGGL(concat_fwd_kernel<half_t>, dim3(nb), dim3(256), 0, s, (const half_t*)p0, (const half_t*)p1, inv, add, B, H, W, C0, C1, Ct, (half_t*)out));
    SDE_CHECK_LAUNCH("sde_concat_fwd");
    return SDE_OK;
}

int sde_concat_bwd(const void* dout, int add, int B, int H, int W, int C0, int C1, int Ct, int dtype, void* d0, void* d1, float* d_inv, sde_stream_t stream) {
    const int V = SDE_IS16(dtype) ? 8 : 4;
    SDE_CHECK_ARG(dout && d0 && SDE_DTYPE_OK(dtype) && B > 0 && H > 0 && W > 0 && C0 > 0 && C0 % V == 0 && C1 % V == 0 && Ct % V == 0, "sde_concat_bwd: bad argument");
    SDE_CHECK_ARG(add ? d1 == nullptr : ((C1 == 0) == (d1 == nullptr)), "sde_concat_bwd: d1 is written in concat mode only");
    hipStream_t s = (hipStream_t)stream;
    const int C1c = add ? 0 : C1;
    const int nb = grid_for((long)B * H * W * ((C0 + C1c) / V));
    PK_DISPATCH(dtype,
                hipLaunchKernelGGL(concat_bwd_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)dout, B, H, W, C0, C1c, Ct, (float*)d0, (float*)d1),
                hipLaunchKernelGGL(concat_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)dout, B, H, W, C0, C1c, Ct, (bf16_t*)d0, (bf16_t*)d1),
                hipLaunchKernelGGL(concat_bwd_kernel<half_t>, dim3(nb), dim3(256), 0, s, (const half_t*)dout, B, H, W, C0, C1c, Ct, (half_t*)d0, (half_t*)d1));
    SDE_CHECK_LAUNCH("sde_concat_bwd");
    if (d_inv) {
        SDE_CHECK_ARG(H % 2 == 0 && W % 2 == 0 && C0 + C1c < Ct, "sde_concat_bwd: no inverse-depth channel in this layout");
        const int nb2 = grid_for((long)B * (H / 2) * (W / 2));
        PK_DISPATCH(dtype,
                    hipLaunchKernelGGL(concat_bwd_inv_kernel<float>, dim3(nb2), dim3(256), 0, s, (const float*)dout, B, H / 2, W / 2, Ct, C0 + C1c, d_inv),
                    hipLaunchKernelGGL(concat_bwd_inv_kernel<bf16_t>, dim3(nb2), dim3(256), 0, s, (const bf16_t*)dout, B, H / 2, W / 2, Ct, C0 + C1c, d_inv),
                    hipLaunchKernelGGL(concat_bwd_inv_kernel<half_t>, dim3(nb2), dim3(256), 0, s, (const half_t*)dout, B, H / 2, W / 2, Ct, C0 + C1c, d_inv));
        SDE_CHECK_LAUNCH("sde_concat_bwd/inv");
    }
    return SDE_OK;
}

int sde_inv_depth_head_fwd(const void* y, int B, int H, int W, int ld, float min_depth_head, float min_depth, float max_depth, int flip, int dtype, float* inv,
                           float* depth, sde_stream_t stream) {
    SDE_CHECK_ARG(y && inv && depth && SDE_DTYPE_OK(dtype) && B > 0 && H > 0 && W > 0 && ld > 0 && min_depth_head > 0.f && min_depth > 0.f && max_depth > min_depth,
                  "sde_inv_depth_head_fwd: bad argument");
    const int nb = grid_for((long)B * H * W);
    hipStream_t s = (hipStream_t)stream;
    const float lo = 1.0f / max_depth, hi = 1.0f / min_depth;
    PK_DISPATCH(dtype,
                hipLaunchKernelGGL(inv_depth_head_fwd_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)y, B, H, W, ld, min_depth_head, lo, hi, flip, inv, depth),
                hipLaunchKernelGGL(inv_depth_head_fwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)y, B, H, W, ld, min_depth_head, lo, hi, flip, inv, depth),
                hipLaunchKernelGGL(inv_depth_head_fwd_kernel<half_t>, dim3(nb), dim3(256), 0, s, (const half_t*)y, B, H, W, ld, min_depth_head, lo, hi, flip, inv, depth));
    SDE_CHECK_LAUNCH("sde_inv_depth_head_fwd");
    return SDE_OK;
}

int sde_inv_depth_head_bwd(const void* y, const float* d_inv, const float* d_depth, int B, int H, int W, int ld, float min_depth_head, float min_depth,
                           float max_depth, int flip, int dtype, void* dy, sde_stream_t stream) {
    SDE_CHECK_ARG(y && dy && (d_inv || d_depth) && SDE_DTYPE_OK(dtype) && B > 0 && H > 0 && W > 0 && ld > 0 && min_depth_head > 0.f && min_depth > 0.f && max_depth > min_depth,
                  "sde_inv_depth_head_bwd: bad argument");
    const int nb = grid_for((long)B * H * W);
    hipStream_t s = (hipStream_t)stream;
    const float lo = 1.0f / max_depth, hi = 1.0f / min_depth;
    PK_DISPATCH(dtype,
                hipLaunchKernelGGL(inv_depth_head_bwd_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)y, d_inv, d_depth, B, H, W, ld, min_depth_head, lo, hi, flip, (float*)dy),
                hipLaunchKernelGGL(inv_depth_head_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)y, d_inv, d_depth, B, H, W, ld, min_depth_head, lo, hi, flip, (bf16_t*)dy),
                hipLaunchKernelGGL(inv_depth_head_bwd_kernel<half_t>, dim3(nb), dim3(256), 0, s, (const half_t*)y, d_inv, d_depth, B, H, W, ld, min_depth_head, lo, hi, flip, (half_t*)dy));
    SDE_CHECK_LAUNCH("sde_inv_depth_head_bwd");
    return SDE_OK;
}

}  // extern "C"
