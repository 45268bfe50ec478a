// The 3-D convolution of PackNet's packing / unpacking layers (reference: detectron2/layers/layers01.py:L223-298):
//   x.unsqueeze(1) -> nn.Conv3d(1, 8, kernel 3x3x3, stride 1, padding 1) -> view(b, 8*D, h, w)
// i.e. a 27-tap stencil over the (channel, y, x) volume of a feature map with 8 output features; output channel = f*D + ch.
//
// Layout: NHWC, so the depth axis of the 3-D convolution is the contiguous channel axis: the three depth taps of a pixel are
// neighbouring elements of one 16-byte group (plus one element either side).  The work is 216 FMAs per input element with
// K = 27: far too thin for the matrix cores (a 16x16x32 MFMA would be 42 % padding), so these are VALU kernels --
// one thread = one pixel x one 16-byte channel group x all 8 features (64 fp32 accumulators), weights through scalar loads.
// Roofline: VALU fp32 (78.6 TFLOP/s); bytes per element: 2 in + 16 out (bf16).
//
// Weight gradient: per-workgroup partial sums of the 216 + 8 outputs, then a fixed-order column sum (no atomics).
#include "common.h"
#include "sde_hip.h"

namespace {

constexpr int NF = 8;            // output features of the Conv3d (layers01.py d=8)
constexpr int NT = 27;           // taps
constexpr int WG_COLS = NF * (NT + 1);   // per-feature 27 weight gradients + 1 bias gradient

template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int V = 4; };
template <> struct VecOf<bf16_t> { static constexpr int V = 8; };
template <> struct VecOf<half_t> { static constexpr int V = 8; };

template <typename T> __device__ __forceinline__ void load_vec(const T* p, float* o);
template <> __device__ __forceinline__ void load_vec<float>(const float* p, float* o) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
template <> __device__ __forceinline__ void load_vec<bf16_t>(const bf16_t* p, float* o) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) { o[2 * i] = __uint_as_float(u[i] << 16); o[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u); }
}
template <> __device__ __forceinline__ void load_vec<half_t>(const half_t* p, float* o) {
    typedef __attribute__((ext_vector_type(8))) _Float16 h8;
    const h8 t = *reinterpret_cast<const h8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (float)t[i];
}
template <typename T> __device__ __forceinline__ void store_vec(T* p, const float* v);
template <> __device__ __forceinline__ void store_vec<float>(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store_vec<bf16_t>(bf16_t* p, const float* v) {
    bf16_t t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16_t)v[i];
    *reinterpret_cast<uint4*>(p) = *reinterpret_cast<const uint4*>(t);
}

template <> __device__ __forceinline__ void store_vec<half_t>(half_t* p, const float* v) {
    typedef __attribute__((ext_vector_type(8))) _Float16 h8;
    h8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (_Float16)v[i];
    *reinterpret_cast<h8*>(p) = t;
}

// Value held by the previous / next lane of the wave (DPP wave_shr:1 / wave_shl:1: a VALU modifier, no LDS crossbar trip).
__device__ __forceinline__ float lane_prev(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float lane_next(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// the 16-byte group at p plus its two neighbours along the channel axis (zero outside [0, D)).  Consecutive lanes hold consecutive channel groups of
// one pixel (idx = pixel * groups + group), so the neighbour elements are the next lane's first and the previous lane's last element: two DPP moves
// instead of two 2-byte global loads per group (the loads were two thirds of the load instructions of these kernels).  Only the first / last lane of a
// wave can have its neighbour in another wave (more than 64 groups per pixel, or a group count that does not divide 64): those two lanes load it.
// Lanes of one pixel take the same branches (same tap validity), so a lane that uses a neighbour's value always finds that neighbour active.
template <typename T>
__device__ __forceinline__ void load_with_edges(const T* p, int c0, int D, float* o /*[V+2]*/) {
    constexpr int V = VecOf<T>::V;
    load_vec<T>(p, o + 1);
    float l = lane_prev(o[V]), r = lane_next(o[1]);
    const int lane = threadIdx.x & 63;
    if (lane == 0 && c0 > 0) l = (float)p[-1];
    if (lane == 63 && c0 + V < D) r = (float)p[V];
    o[0] = c0 > 0 ? l : 0.f;
    o[V + 1] = c0 + V < D ? r : 0.f;
}

struct Item { int b, y, x, c0; long pix; };
template <int V> __device__ __forceinline__ Item decode(long idx, int H, int W, int D) {
    const int groups = D / V;
    Item it;
    const int g = (int)(idx % groups);
    it.pix = idx / groups;
    it.x = (int)(it.pix % W);
    it.y = (int)((it.pix / W) % H);
    it.b = (int)(it.pix / ((long)W * H));
    it.c0 = g * V;
    return it;
}

// y[b,h,w, f*D + ch] = bias[f] + sum_{kd,kh,kw} w[f][kd][kh][kw] * x[b, h+kh-1, w+kw-1, ch+kd-1]
template <typename T>
__global__ void __launch_bounds__(256) conv3d_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                         T* __restrict__ y, int B, int H, int W, int D) {
    constexpr int V = VecOf<T>::V;
    const long total = (long)B * H * W * (D / V);
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const Item it = decode<V>(idx, H, W, D);
    float acc[NF][V];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int c = 0; c < V; ++c) acc[f][c] = bias ? bias[f] : 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int iy = it.y + kh - 1;
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ix = it.x + kw - 1;
            if (ix < 0 || ix >= W) continue;
            float xv[V + 2];
            load_with_edges<T>(x + (((long)it.b * H + iy) * W + ix) * D + it.c0, it.c0, D, xv);
#pragma unroll
            for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                for (int f = 0; f < NF; ++f) {
                    const float wv = w[f * NT + kd * 9 + kh * 3 + kw];
#pragma unroll
                    for (int c = 0; c < V; ++c) acc[f][c] = fmaf(wv, xv[c + kd], acc[f][c]);
                }
        }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) store_vec<T>(y + it.pix * ((long)NF * D) + (long)f * D + it.c0, acc[f]);
}

// dx[b,h,w,ch] = sum_f sum_{kd,kh,kw} w[f][kd][kh][kw] * dy[b, h-(kh-1), w-(kw-1), f*D + ch-(kd-1)]
template <typename T>
__global__ void __launch_bounds__(256) conv3d_dgrad_kernel(const T* __restrict__ dy, const float* __restrict__ w, T* __restrict__ dx, int B, int H,
                                                           int W, int D) {
    constexpr int V = VecOf<T>::V;
    const long total = (long)B * H * W * (D / V);
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const Item it = decode<V>(idx, H, W, D);
    float acc[V];
#pragma unroll
    for (int c = 0; c < V; ++c) acc[c] = 0.f;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int iy = it.y - (kh - 1);
        if (iy < 0 || iy >= H) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int ix = it.x - (kw - 1);
            if (ix < 0 || ix >= W) continue;
            const T* base = dy + (((long)it.b * H + iy) * W + ix) * ((long)NF * D) + it.c0;
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                float gv[V + 2];
                load_with_edges<T>(base + (long)f * D, it.c0, D, gv);
#pragma unroll
                for (int kd = 0; kd < 3; ++kd) {
                    const float wv = w[f * NT + kd * 9 + kh * 3 + kw];
#pragma unroll
                    for (int c = 0; c < V; ++c) acc[c] = fmaf(wv, gv[c + 2 - kd], acc[c]);
                }
            }
        }
    }
    store_vec<T>(dx + it.pix * (long)D + it.c0, acc);
}

// (Tried in round 3 and removed: an LDS-tiled data gradient -- 4 x 8 pixels x 64 channels per workgroup, the dy halo staged once, weights in LDS, the
// chunk-edge channels as a correction pass.  Bit-for-bit the same results, but 1.74 ms against 1.55 ms for the kernel above at 12 x 96 x 320 x 256: the
// kernel is bound by its conversion / shuffle instruction mix, not by the 27-fold re-reads, which L1 / L2 absorb.  profiles/README.md, round 3.)

// part[block][f*28 + t] (t < 27: weight tap, t == 27: bias) = this workgroup's share of
//   dw[f][kd][kh][kw] = sum dy[b,h,w,f*D+ch] * x[b,h+kh-1,w+kw-1,ch+kd-1] ;  dbias[f] = sum dy[b,h,w,f*D+ch]
constexpr int WG_ITEMS = 8;      // (pixel, channel group) items per thread at most; small layers take fewer so that the grid still fills the CUs
#ifndef C3D_WG_F
#define C3D_WG_F 4
#endif
constexpr int WG_F = C3D_WG_F;   // features per pass: the 3x3 neighbourhood is re-read NF / WG_F times (registers: WG_F x 28 sums + 10 taps)
template <typename T>
__global__ void __launch_bounds__(256) conv3d_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ part, int B, int H,
                                                           int W, int D, int items) {
    constexpr int V = VecOf<T>::V;
    __shared__ float red[4][WG_F][NT + 1];
    const long total = (long)B * H * W * (D / V);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int f0 = 0; f0 < NF; f0 += WG_F) {
        float acc[WG_F][NT + 1];
#pragma unroll
        for (int f = 0; f < WG_F; ++f)
#pragma unroll
            for (int t = 0; t <= NT; ++t) acc[f][t] = 0.f;
#pragma unroll 1
        for (int i = 0; i < items; ++i) {
            const long idx = ((long)blockIdx.x * items + i) * 256 + threadIdx.x;
            if (idx >= total) break;
            const Item it = decode<V>(idx, H, W, D);
            float g[WG_F][V];
#pragma unroll
            for (int f = 0; f < WG_F; ++f) {
                load_vec<T>(dy + it.pix * ((long)NF * D) + (long)(f0 + f) * D + it.c0, g[f]);
#pragma unroll
                for (int c = 0; c < V; ++c) acc[f][NT] += g[f][c];
            }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const int iy = it.y + kh - 1;
                if (iy < 0 || iy >= H) continue;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const int ix = it.x + kw - 1;
                    if (ix < 0 || ix >= W) continue;
                    float xv[V + 2];
                    load_with_edges<T>(x + (((long)it.b * H + iy) * W + ix) * D + it.c0, it.c0, D, xv);
#pragma unroll
                    for (int kd = 0; kd < 3; ++kd)
#pragma unroll
                        for (int f = 0; f < WG_F; ++f) {           // WG_F x 3 independent chains per tap
                            float s = 0.f;
#pragma unroll
                            for (int c = 0; c < V; ++c) s = fmaf(g[f][c], xv[c + kd], s);
                            acc[f][kd * 9 + kh * 3 + kw] += s;
                        }
                }
            }
        }
#pragma unroll
        for (int f = 0; f < WG_F; ++f)
#pragma unroll
            for (int t = 0; t <= NT; ++t) acc[f][t] = sde_wave_sum(acc[f][t]);
        __syncthreads();       // previous pass's readers are done with red[]
        if (lane == 0)
#pragma unroll
            for (int f = 0; f < WG_F; ++f)
#pragma unroll
                for (int t = 0; t <= NT; ++t) red[wave][f][t] = acc[f][t];
        __syncthreads();
        if (threadIdx.x < WG_F * (NT + 1)) {
            const int f = threadIdx.x / (NT + 1), t = threadIdx.x % (NT + 1);
            part[(size_t)blockIdx.x * WG_COLS + (f0 + f) * (NT + 1) + t] = (red[0][f][t] + red[1][f][t]) + (red[2][f][t] + red[3][f][t]);
        }
    }
}

// column c of part[rows][WG_COLS] summed in a fixed order -> dw[f*27 + t] / dbias[f]
__global__ void __launch_bounds__(256) conv3d_wgrad_finalize_kernel(const float* __restrict__ part, int rows, float* __restrict__ dw, float* __restrict__ dbias,
                                                                    int accumulate) {
    __shared__ double red[256];
    const int col = blockIdx.x;
    double s = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) s += (double)part[(size_t)r * WG_COLS + col];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int f = col / (NT + 1), t = col % (NT + 1);
        const float v = (float)red[0];
        if (t < NT) { float* o = dw + f * NT + t; *o = accumulate ? *o + v : v; }
        else if (dbias) { float* o = dbias + f; *o = accumulate ? *o + v : v; }
    }
}

int check_shape(const char* who, const void* a, const void* b, int B, int H, int W, int D, int dtype) {
    if (!a || !b) { sde_set_error("%s: null pointer", who); return SDE_ERR_ARG; }
    if (!SDE_DTYPE_OK(dtype)) { sde_set_error("%s: bad dtype %d", who, dtype); return SDE_ERR_ARG; }
    const int V = SDE_IS16(dtype) ? 8 : 4;
    if (B <= 0 || H <= 0 || W <= 0 || D <= 0 || D % V) { sde_set_error("%s: bad shape B=%d H=%d W=%d D=%d (D %% %d)", who, B, H, W, D, V); return SDE_ERR_ARG; }
    if ((long)B * H * W * NF * D > 0x7fffffff0L) { sde_set_error("%s: tensor too large", who); return SDE_ERR_ARG; }
    return SDE_OK;
}

long n_items(int B, int H, int W, int D, int dtype) { return (long)B * H * W * (D / (SDE_IS16(dtype) ? 8 : 4)); }

}  // namespace

extern "C" {

int sde_conv3d_fwd(const void* x, const float* w, const float* bias, int B, int H, int W, int D, int dtype, void* y, sde_stream_t stream) {
    int rc = check_shape("sde_conv3d_fwd", x, y, B, H, W, D, dtype);
    if (rc) return rc;
    SDE_CHECK_ARG(w, "sde_conv3d_fwd: null weights");
    const unsigned nb = (unsigned)((n_items(B, H, W, D, dtype) + 255) / 256);
    if (dtype == SDE_BF16) hipLaunchKernelGGL(conv3d_fwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, w, bias, (bf16_t*)y, B, H, W, D);
    else if (dtype == SDE_F16) hipLaunchKernelGGL(conv3d_fwd_kernel<half_t>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, w, bias, (half_t*)y, B, H, W, D);
    else hipLaunchKernelGGL(conv3d_fwd_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)x, w, bias, (float*)y, B, H, W, D);
    SDE_CHECK_LAUNCH("sde_conv3d_fwd");
    return SDE_OK;
}

int sde_conv3d_dgrad(const void* dy, const float* w, int B, int H, int W, int D, int dtype, void* dx, sde_stream_t stream) {
    int rc = check_shape("sde_conv3d_dgrad", dy, dx, B, H, W, D, dtype);
    if (rc) return rc;
    SDE_CHECK_ARG(w, "sde_conv3d_dgrad: null weights");
    const unsigned nb = (unsigned)((n_items(B, H, W, D, dtype) + 255) / 256);
    if (dtype == SDE_BF16) hipLaunchKernelGGL(conv3d_dgrad_kernel<bf16_t>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, w, (bf16_t*)dx, B, H, W, D);
    else if (dtype == SDE_F16) hipLaunchKernelGGL(conv3d_dgrad_kernel<half_t>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const half_t*)dy, w, (half_t*)dx, B, H, W, D);
    else hipLaunchKernelGGL(conv3d_dgrad_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)dy, w, (float*)dx, B, H, W, D);
    SDE_CHECK_LAUNCH("sde_conv3d_dgrad");
    return SDE_OK;
}

// items per thread: 8 for the big layers (fewer partial rows), down to 1 where 8 would leave fewer than ~1024 workgroups (the 6x20 ... 24x80 levels
// ran 23-180 workgroups of serial work: 80 us launches for 0.01 GB of data)
static int wgrad_items(long n) {
    long it = n / (256L * 1024);
    return (int)(it < 1 ? 1 : (it > WG_ITEMS ? WG_ITEMS : it));
}

int sde_conv3d_wgrad_num_blocks(int B, int H, int W, int D, int dtype) {
    const int V = SDE_IS16(dtype) ? 8 : 4;
    if (B <= 0 || H <= 0 || W <= 0 || D <= 0 || D % V) return -1;
    const long n = n_items(B, H, W, D, dtype);
    const int items = wgrad_items(n);
    return (int)((n + 256L * items - 1) / (256L * items));
}

int sde_conv3d_wgrad(const void* x, const void* dy, int B, int H, int W, int D, int dtype, float* part, float* dw, float* dbias, int accumulate,
                     sde_stream_t stream) {
    int rc = check_shape("sde_conv3d_wgrad", x, dy, B, H, W, D, dtype);
    if (rc) return rc;
    SDE_CHECK_ARG(part && dw, "sde_conv3d_wgrad: null pointer");
    const int nb = sde_conv3d_wgrad_num_blocks(B, H, W, D, dtype);
    const int items = wgrad_items(n_items(B, H, W, D, dtype));
    if (dtype == SDE_BF16) hipLaunchKernelGGL(conv3d_wgrad_kernel<bf16_t>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)dy, part, B, H, W, D, items);
    else if (dtype == SDE_F16) hipLaunchKernelGGL(conv3d_wgrad_kernel<half_t>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, (const half_t*)dy, part, B, H, W, D, items);
    else hipLaunchKernelGGL(conv3d_wgrad_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float*)x, (const float*)dy, part, B, H, W, D, items);
    SDE_CHECK_LAUNCH("sde_conv3d_wgrad");
    hipLaunchKernelGGL(conv3d_wgrad_finalize_kernel, dim3(WG_COLS), dim3(256), 0, (hipStream_t)stream, part, nb, dw, dbias, accumulate);
    SDE_CHECK_LAUNCH("sde_conv3d_wgrad/finalize");
    return SDE_OK;
}

}  // extern "C"
