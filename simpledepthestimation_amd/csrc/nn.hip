// Bandwidth-bound layers around the convolution GEMMs (gfx950): input normalisation + layout change, training-mode
// BatchNorm (+ReLU, +residual) forward/backward, GroupNorm+ReLU, 3x3/2 max-pool, activation backward + bias gradient,
// reflection-pad / nearest-upsample gradient folding, the depth head (softplus + disp_to_depth) and fused Adam/AdamW.
//
// Replaces (reference, read-only): (img-mean)/std of meta_arch/Supervised.py:L39 / MonoDepth2.py:L60; torchvision's
// BatchNorm2d/ReLU/MaxPool2d/residual add inside layers/resnet_encoder.py:L91-97; nn.ELU, ReflectionPad2d backward,
// F.interpolate(nearest) backward and torch.cat backward of layers/depth_decoder.py:L21-53,L95-110; nn.Softplus +
// disp_to_depth (depth_decoder.py:L9-18, DepthResNet.py:L57) and torch.flip (DepthResNet.py:L52-60); nn.GroupNorm(16)+ReLU
// of pose_net/PoseNet.py:L13-20; torch.optim.Adam / AdamW of projects/*/train.py.
//
// Layout: activations NHWC, 16-byte channel groups; every kernel moves 16 B per lane per access with lanes running along
// the channel-fastest axis (full 1 KiB per wave instruction).  Channel reductions write per-workgroup partial slabs that a
// small second kernel sums in fixed order (fp64) -- no float atomics, bit-reproducible statistics and gradients.
#include <stdlib.h>
#include "common.h"
#include "sde_hip.h"

namespace {

template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int V = 4; };
template <> struct VecOf<bf16_t> { static constexpr int V = 8; };
template <> struct VecOf<half_t> { static constexpr int V = 8; };

template <typename T> __device__ __forceinline__ void load_vec(const T* p, float* v);
template <> __device__ __forceinline__ void load_vec<float>(const float* p, float* v) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void load_vec<bf16_t>(const bf16_t* p, float* v) {
    const uint4 t = *reinterpret_cast<const uint4*>(p);
    const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        v[2 * i] = __uint_as_float(w[i] << 16);
        v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
}
template <> __device__ __forceinline__ void load_vec<half_t>(const half_t* p, float* v) {
    typedef __attribute__((ext_vector_type(8))) _Float16 h8;
    const h8 t = *reinterpret_cast<const h8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}
template <typename T> __device__ __forceinline__ void store_vec(T* p, const float* v);
template <> __device__ __forceinline__ void store_vec<float>(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store_vec<bf16_t>(bf16_t* p, const float* v) {
    bf16_t o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (bf16_t)v[i];
    *reinterpret_cast<uint4*>(p) = *reinterpret_cast<uint4*>(o);
}

template <> __device__ __forceinline__ void store_vec<half_t>(half_t* p, const float* v) {
    typedef __attribute__((ext_vector_type(8))) _Float16 h8;
    h8 o;
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = (_Float16)v[i];
    *reinterpret_cast<h8*>(p) = o;
}

// ------------------------------------------------------------------------------------------------------------------
// Input: NCHW fp32 image -> normalised NHWC T, channels zero-padded to Cpad, optional horizontal flip
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) prep_input_kernel(const float* __restrict__ img, const float* __restrict__ mean, const float* __restrict__ std_,
                                                         int B, int C, int H, int W, int Cpad, int flip, T* __restrict__ out) {
    const long n = (long)B * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int x = (int)(i % W), y = (int)((i / W) % H), b = (int)(i / ((long)W * H));
        const int xs = flip ? W - 1 - x : x;
        T* o = out + i * Cpad;
        for (int c = 0; c < Cpad; ++c) {
            float v = 0.f;
            if (c < C) {
                v = img[(((long)b * C + c) * H + y) * W + xs];
                if (mean) v = (v - mean[c]) / std_[c];
            }
            o[c] = (T)v;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Deterministic two-level reduction of partial-sum slabs: out[ro][col] = sum of rows of chunk ro of in[rows][width].
// The small second level (<= SDE_REDUCE_ROWS rows) is summed in fp64 by the *_finalize kernels.
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024) rows_reduce_kernel(const float* __restrict__ in, int rows, int width, int chunk, float* __restrict__ out) {
    __shared__ float sh[16][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6, ro = blockIdx.y;
    const int r0 = ro * chunk, r1 = min(rows, r0 + chunk);
    float s = 0.f;
    if (col < width)
        for (int r = r0 + sub; r < r1; r += 16) s += in[(size_t)r * width + col];
    sh[sub][threadIdx.x & 63] = s;
    __syncthreads();
    if (sub == 0 && col < width) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += sh[i][threadIdx.x & 63];
        out[(size_t)ro * width + col] = t;
    }
}

// Sum over the 32 lanes of a half-wave (the finalize kernels spread the <= SDE_REDUCE_ROWS partial rows over 32 lanes, so every
// row is one independent load instead of a serial chain of dependent loads).  Fixed shuffle order: deterministic.
__device__ __forceinline__ double half_wave_sum(double v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Column sums of a partial-sum slab [rows][width] for the 16 consecutive columns starting at col0, by one 1024-thread workgroup:
// thread = (row lane tid / 16, column tid % 16); 64 rows are in flight per load and every row is one 64-byte segment, so the loads
// coalesce (the previous form walked rows with a stride of the whole slab width per lane, 32 lanes per channel); 4 independent loads per
// thread before the first add, i.e. 256 rows per round trip: a 2048-row slab is 8 round trips.  Double accumulation in a fixed order
// (row lane, then the 64-way LDS sum): deterministic.  Result in sh[SLAB_RES + 0..15] after the trailing barrier; columns >= width read 0.
constexpr int SLAB_T = 1024, SLAB_RL = SLAB_T / 16, SLAB_RES = SLAB_RL * 17, SLAB_SH = SLAB_RES + 16;
__device__ __forceinline__ void slab_colsum16(const float* __restrict__ part, int rows, int width, int col0, double* sh /* [SLAB_SH] */) {
    const int rl = threadIdx.x >> 4, e = threadIdx.x & 15;
    const bool ok = col0 + e < width;
    const float* src = part + col0 + e;
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int t = rl;
    if (ok) {
        for (; t + 7 * SLAB_RL < rows; t += 8 * SLAB_RL) {       // 512 rows per round trip
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = src[(size_t)(t + i * SLAB_RL) * width];
            a0 += (double)v[0] + (double)v[4]; a1 += (double)v[1] + (double)v[5]; a2 += (double)v[2] + (double)v[6]; a3 += (double)v[3] + (double)v[7];
        }
        for (; t + 3 * SLAB_RL < rows; t += 4 * SLAB_RL) {
            const float v0 = src[(size_t)t * width], v1 = src[(size_t)(t + SLAB_RL) * width], v2 = src[(size_t)(t + 2 * SLAB_RL) * width],
                        v3 = src[(size_t)(t + 3 * SLAB_RL) * width];
            a0 += v0; a1 += v1; a2 += v2; a3 += v3;
        }
        for (; t < rows; t += SLAB_RL) a0 += src[(size_t)t * width];
    }
    sh[rl * 17 + e] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (threadIdx.x < 16) {
        double tot = 0;
        for (int i = 0; i < SLAB_RL; ++i) tot += sh[i * 17 + threadIdx.x];
        sh[SLAB_RES + threadIdx.x] = tot;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------------------------
// BatchNorm
// ------------------------------------------------------------------------------------------------------------------
// bnp layout: [4][C] = mean, rstd, scale (= gamma*rstd), shift (= beta - mean*scale)
__global__ void __launch_bounds__(SLAB_T) bn_finalize_kernel(const float* __restrict__ part, int tiles, int C, float count, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                                          float momentum, float eps, float* __restrict__ bnp) {
    // one workgroup = 8 channels = 16 consecutive floats (sum, sum^2 interleaved) of every partial row
    __shared__ double sh[SLAB_SH];
    slab_colsum16(part, tiles, 2 * C, blockIdx.x * 16, sh);
    const int c = blockIdx.x * 8 + threadIdx.x;
    if (threadIdx.x >= 8 || c >= C) return;
    const double s1 = sh[SLAB_RES + 2 * threadIdx.x], s2 = sh[SLAB_RES + 2 * threadIdx.x + 1];
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0) var = 0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * rstd;
    bnp[c] = (float)mean; bnp[C + c] = rstd; bnp[2 * C + c] = sc; bnp[3 * C + c] = beta[c] - (float)mean * sc;
    if (rmean) {   // running statistics: unbiased variance, torch momentum convention
        const double unb = count > 1.f ? var * count / (count - 1.0) : var;
        rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
        rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
    }
}

__global__ void __launch_bounds__(64) bn_eval_params_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ rmean, const float* __restrict__ rvar, float eps, int C,
                                                            float* __restrict__ bnp) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c >= C) return;
    const float rstd = 1.0f / sqrtf(rvar[c] + eps);
    const float sc = gamma[c] * rstd;
    bnp[c] = rmean[c]; bnp[C + c] = rstd; bnp[2 * C + c] = sc; bnp[3 * C + c] = beta[c] - rmean[c] * sc;
}

// out = act(y*scale + shift (+ residual))
template <typename T>
__global__ void __launch_bounds__(256) bn_apply_kernel(const T* __restrict__ y, const float* __restrict__ bnp, const T* __restrict__ res, int relu,
                                                       long M, int C, T* __restrict__ out) {
    constexpr int V = VecOf<T>::V;
    const int cch = C / V;
    const long total = M * cch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c0 = (int)(i % cch) * V;
        float v[V], r[V];
        load_vec<T>(y + i * V, v);
        if (res) load_vec<T>(res + i * V, r);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float t = fmaf(v[e], bnp[2 * C + c0 + e], bnp[3 * C + c0 + e]);      // (bn_bwd_reduce re-derives the ReLU mask from exactly this expression)
            if (res) t += r[e];
            if (relu) t = fmaxf(t, 0.f);
            v[e] = t;
        }
        store_vec<T>(out + i * V, v);
    }
}

// BatchNorm finalize + apply in ONE launch for layers whose partial slab is short (<= BNFA_MAX_ROWS rows): a workgroup owns a strip of 64
// channels (128 bytes of every row, 16-bit types) x a range of rows.  It first reduces the slab columns of ITS 64 channels -- every workgroup of
// a strip redundantly, in the same fixed order, fp64: identical results everywhere -- into scale / shift held in LDS, then streams its rows.  The
// workgroups of row range 0 also write bnp (for backward) and update the running statistics.  Saves the separate finalize launch (4.7 us + a kernel
// boundary, 53 times per ResNet-50 forward pass) at the price of <= 100 KB of L2-resident re-reads per workgroup.
constexpr int BNFA_MAX_ROWS = 256;
template <typename T>
__global__ void __launch_bounds__(256) bn_finalize_apply_kernel(const float* __restrict__ part, int rows, int C, float count, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                                                                float momentum, float eps, float* __restrict__ bnp, const T* __restrict__ y,
                                                                const T* __restrict__ res, int relu, long M, long rows_per_wg, T* __restrict__ out) {
    constexpr int V = VecOf<T>::V;                       // 8: this kernel is instantiated for the 16-bit types only (64 channels = 8 lanes x 16 B)
    __shared__ double acc[8][128];
    __shared__ float ss[2][64];                          // scale, shift of the strip
    const int strip = blockIdx.x, c0 = strip * 64;
    const int tid = threadIdx.x;
    {   // column sums of the strip's 128 floats (sum, sum^2 interleaved) of every slab row: thread = (float4 tid & 31, row lane tid >> 5), eight
        // independent 16-byte loads per round trip (256 rows = 4 round trips)
        const int f4 = tid & 31, rl = tid >> 5;
        const float4* src = reinterpret_cast<const float4*>(part + (size_t)c0 * 2) + f4;
        const size_t ld4 = (size_t)C / 2;                // float4 per slab row
        double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
        int r = rl;
        for (; r + 56 < rows; r += 64) {
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = src[(size_t)(r + 8 * i) * ld4];
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                a[0] += v[i].x; a[1] += v[i].y; a[2] += v[i].z; a[3] += v[i].w;
                b[0] += v[i + 1].x; b[1] += v[i + 1].y; b[2] += v[i + 1].z; b[3] += v[i + 1].w;
            }
        }
        for (; r < rows; r += 8) { const float4 v = src[(size_t)r * ld4]; a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w; }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[rl][f4 * 4 + e] = a[e] + b[e];
    }
    __syncthreads();
    if (tid < 64) {
        const int c = c0 + tid;
        double s1 = 0, s2 = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { s1 += acc[i][2 * tid]; s2 += acc[i][2 * tid + 1]; }
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        if (var < 0) var = 0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float sc = gamma[c] * rstd, sf = beta[c] - (float)mean * sc;
        ss[0][tid] = sc; ss[1][tid] = sf;
        if (blockIdx.y == 0) {
            bnp[c] = (float)mean; bnp[C + c] = rstd; bnp[2 * C + c] = sc; bnp[3 * C + c] = sf;
            if (rmean) {   // running statistics: unbiased variance, torch momentum convention
                const double unb = count > 1.f ? var * count / (count - 1.0) : var;
                rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
                rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
            }
        }
    }
    __syncthreads();
    const int g8 = tid & 7, rl = tid >> 3;               // 16-byte group inside the strip, row lane (32 rows per pass)
    float sc[V], sf[V];
#pragma unroll
    for (int e = 0; e < V; ++e) { sc[e] = ss[0][g8 * V + e]; sf[e] = ss[1][g8 * V + e]; }
    const long r0 = (long)blockIdx.y * rows_per_wg, r1 = min(M, r0 + rows_per_wg);
    for (long r = r0 + rl; r < r1; r += 32) {
        const long off = r * C + c0 + g8 * V;
        float v[V], rr[V];
        load_vec<T>(y + off, v);
        if (res) load_vec<T>(res + off, rr);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            float t = fmaf(v[e], sc[e], sf[e]);          // (bn_bwd_reduce re-derives the ReLU mask from exactly this expression)
            if (res) t += rr[e];
            if (relu) t = fmaxf(t, 0.f);
            v[e] = t;
        }
        store_vec<T>(out + off, v);
    }
}

// Combine per-thread column accumulators of a 256-thread block whose threads tid, tid+cch, tid+2cch, ... share a 16-byte
// channel group (cch = groups per row, a divisor of 256).  sh: [256][NV] floats.  Thread t < cch returns the block sums of its
// group in a[]; no float atomics (deterministic, and LDS float atomics are slow).
template <int NV, int NT = 256>
__device__ __forceinline__ void block_group_reduce(float (&a)[NV], int cch, float* sh) {
    const int t = threadIdx.x;
#pragma unroll
    for (int e = 0; e < NV; ++e) sh[e * NT + t] = a[e];      // element-major: lanes hit consecutive banks
    __syncthreads();
    if (NT > 256) {
        // wide workgroups: the first 256 threads fold the other quarters onto themselves first (same column: NT and 256 are multiples of cch)
        if (t < 256) {
            for (int r = t + 256; r < NT; r += 256)
#pragma unroll
                for (int e = 0; e < NV; ++e) a[e] += sh[e * NT + r];
#pragma unroll
            for (int e = 0; e < NV; ++e) sh[e * NT + t] = a[e];
        }
        __syncthreads();
    }
    if (t < cch) {
        for (int r = t + cch; r < 256; r += cch)
#pragma unroll
            for (int e = 0; e < NV; ++e) a[e] += sh[e * NT + r];
    }
}

// Channel reduction pass of BN backward: partial[blk][C][2] = (sum dz, sum dz*xhat), dz = dout * (out > 0 if relu).
// Grid-stride over (row, 16-byte group) items; gridDim*256 is a multiple of the groups per row, so a thread's group is fixed.
// dz = relu'(out) * (dout0 [+ dout1] [+ dout2]): the block output feeds up to three consumers (next block's first convolution, its residual
// path or down-sampling convolution, a decoder skip); their gradients arrive separately and are summed here instead of by an extra add
// kernel per tensor.  When `gm` is given the masked sum is stored once (it IS the gradient of the residual input, and the apply pass reads it
// instead of dout / out); per-block partial sums of dz and dz * xhat go to `part`.
template <typename T, int V>
__device__ __forceinline__ void round_vec(float* d) {       // values as they come back from storage type T
    if (sizeof(T) == 4) return;
#pragma unroll
    for (int e = 0; e < V; ++e) d[e] = (float)(T)d[e];
}

// relu: 0 none; 1 mask from the saved output; 2 mask re-derived from y (BatchNorm + ReLU WITHOUT a residual: out > 0 <=> fma(y, scale, shift) > 0,
// the expression bn_apply_kernel evaluates) -- one activation-sized stream less to read (two of three BatchNorms of a bottleneck block)
template <typename T, int V>
__device__ __forceinline__ void bn_load_dz(const T* d0, const T* d1, const T* d2, const T* out, int relu, long off, float* d, const float* yv,
                                           const float* sc, const float* sf) {
    load_vec<T>(d0 + off, d);
    if (d1) {
        float t[V];
        load_vec<T>(d1 + off, t);
#pragma unroll
        for (int e = 0; e < V; ++e) d[e] += t[e];
    }
    if (d2) {
        float t[V];
        load_vec<T>(d2 + off, t);
#pragma unroll
        for (int e = 0; e < V; ++e) d[e] += t[e];
    }
    if (relu == 1) {
        float o[V];
        load_vec<T>(out + off, o);
#pragma unroll
        for (int e = 0; e < V; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f;
    } else if (relu == 2) {
#pragma unroll
        for (int e = 0; e < V; ++e) d[e] = fmaf(yv[e], sc[e], sf[e]) > 0.f ? d[e] : 0.f;
    }
}

// NT = 1024 (16-bit types, host-checked fast-path shapes only): the same threads in flight from a quarter of the workgroups, so that the partial
// slab of the big layers stays within the 256 rows the fused finalize + apply launch re-reduces per workgroup.
template <typename T, int NT = 256>
__global__ void __launch_bounds__(NT) bn_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ dout1, const T* __restrict__ dout2,
                                                           const T* __restrict__ out, const T* __restrict__ y,
                                                           const float* __restrict__ bnp, int relu, long M, int C, long rows_per_block,
                                                           float* __restrict__ part, T* __restrict__ gm) {
    constexpr int V = VecOf<T>::V;
    extern __shared__ float sh[];   // fast path: [NT][2V]; fallback: [2][C]
    const int cch = C / V;
    if (NT > 256 || (cch <= 256 && 256 % cch == 0)) {
        const int col = threadIdx.x % cch, c0 = col * V;
        float mean[V], rstd[V], sc[V], sf[V], acc[2 * V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            mean[e] = bnp[c0 + e]; rstd[e] = bnp[C + c0 + e]; sc[e] = bnp[2 * C + c0 + e]; sf[e] = bnp[3 * C + c0 + e];
            acc[e] = 0.f; acc[V + e] = 0.f;
        }
        const long total = M * cch;
        for (long i = (long)blockIdx.x * NT + threadIdx.x; i < total; i += (long)gridDim.x * NT) {
            float d[V], yv[V];
            load_vec<T>(y + i * V, yv);
            bn_load_dz<T, V>(dout, dout1, dout2, out, relu, i * V, d, yv, sc, sf);
            if (gm) {
                round_vec<T, V>(d);              // the sums must see what the apply pass will read back
                store_vec<T>(gm + i * V, d);
            }
#pragma unroll
            for (int e = 0; e < V; ++e) { acc[e] += d[e]; acc[V + e] += d[e] * ((yv[e] - mean[e]) * rstd[e]); }
        }
        block_group_reduce<2 * V, NT>(acc, cch, sh);
        if (threadIdx.x < cch) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                part[((size_t)blockIdx.x * C + c0 + e) * 2] = acc[e];
                part[((size_t)blockIdx.x * C + c0 + e) * 2 + 1] = acc[V + e];
            }
        }
        return;
    }
    // generic fallback (more 16-byte groups per row than threads): shared float atomics
    for (int i = threadIdx.x; i < 2 * C; i += 256) sh[i] = 0.f;
    __syncthreads();
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    const long total = (r1 - r0) * cch;
    for (long i = threadIdx.x; i < total; i += 256) {
        const int col = (int)(i % cch), c0 = col * V;
        const long off = (r0 * cch + i) * V;
        float d[V], yv[V];
        load_vec<T>(y + off, yv);
        bn_load_dz<T, V>(dout, dout1, dout2, out, relu, off, d, yv, bnp + 2 * C + c0, bnp + 3 * C + c0);
        if (gm) {
            round_vec<T, V>(d);
            store_vec<T>(gm + off, d);
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float xh = (yv[e] - bnp[c0 + e]) * bnp[C + c0 + e];
            atomicAdd(&sh[c0 + e], d[e]); atomicAdd(&sh[C + c0 + e], d[e] * xh);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256) {
        part[((size_t)blockIdx.x * C + i) * 2] = sh[i];
        part[((size_t)blockIdx.x * C + i) * 2 + 1] = sh[C + i];
    }
}

// coef layout [2][C]: mean(dz), mean(dz*xhat); also dgamma (+)=, dbeta (+)=
__global__ void __launch_bounds__(SLAB_T) bn_bwd_finalize_kernel(const float* __restrict__ part, int nblk, int C, float count, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate, float* __restrict__ coef) {
    __shared__ double sh[SLAB_SH];
    slab_colsum16(part, nblk, 2 * C, blockIdx.x * 16, sh);
    const int c = blockIdx.x * 8 + threadIdx.x;
    if (threadIdx.x >= 8 || c >= C) return;
    const double s1 = sh[SLAB_RES + 2 * threadIdx.x], s2 = sh[SLAB_RES + 2 * threadIdx.x + 1];
    coef[c] = (float)(s1 / count); coef[C + c] = (float)(s2 / count);
    dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
    dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
}

// dy = scale * (dz - mean(dz) - xhat * mean(dz*xhat)), dz read back as the reduce pass left it (gm) or, for a single un-masked gradient, dout itself
template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const T* __restrict__ dz_in, const T* __restrict__ y,
                                                           const float* __restrict__ bnp, const float* __restrict__ coef, long M, int C,
                                                           T* __restrict__ dy) {
    constexpr int V = VecOf<T>::V;
    const int cch = C / V;
    const long total = M * cch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c0 = (int)(i % cch) * V;
        float d[V], yv[V], g[V];
        load_vec<T>(dz_in + i * V, d);
        load_vec<T>(y + i * V, yv);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float xh = (yv[e] - bnp[c0 + e]) * bnp[C + c0 + e];
            g[e] = bnp[2 * C + c0 + e] * (d[e] - coef[c0 + e] - xh * coef[C + c0 + e]);
        }
        store_vec<T>(dy + i * V, g);
    }
}

// bn_bwd_finalize + bn_bwd_apply in ONE launch for short partial slabs (<= BNFA_MAX_ROWS rows: the layer3 / layer4 bottleneck BatchNorms), the backward
// twin of bn_finalize_apply_kernel: a workgroup owns a 64-channel strip x a range of rows, first reduces the slab columns (sum dz, sum dz * xhat) of its
// own 64 channels -- every workgroup of a strip redundantly, same fixed order, fp64 -- then streams its rows; row range 0 also writes dgamma / dbeta.
template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_finalize_apply_kernel(const float* __restrict__ part, int rows, int C, float count, float* __restrict__ dgamma,
                                                                    float* __restrict__ dbeta, int accumulate, const T* __restrict__ dz_in,
                                                                    const T* __restrict__ y, const float* __restrict__ bnp, long M, long rows_per_wg,
                                                                    T* __restrict__ dy) {
    constexpr int V = VecOf<T>::V;
    __shared__ double acc[8][128];
    __shared__ float cf[2][64];                          // mean(dz), mean(dz * xhat) of the strip
    const int c0 = blockIdx.x * 64, tid = threadIdx.x;
    {
        const int f4 = tid & 31, rl = tid >> 5;
        const float4* src = reinterpret_cast<const float4*>(part + (size_t)c0 * 2) + f4;
        const size_t ld4 = (size_t)C / 2;
        double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
        int r = rl;
        for (; r + 56 < rows; r += 64) {
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = src[(size_t)(r + 8 * i) * ld4];
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                a[0] += v[i].x; a[1] += v[i].y; a[2] += v[i].z; a[3] += v[i].w;
                b[0] += v[i + 1].x; b[1] += v[i + 1].y; b[2] += v[i + 1].z; b[3] += v[i + 1].w;
            }
        }
        for (; r < rows; r += 8) { const float4 v = src[(size_t)r * ld4]; a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w; }
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[rl][f4 * 4 + e] = a[e] + b[e];
    }
    __syncthreads();
    if (tid < 64) {
        const int c = c0 + tid;
        double s1 = 0, s2 = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { s1 += acc[i][2 * tid]; s2 += acc[i][2 * tid + 1]; }
        cf[0][tid] = (float)(s1 / count); cf[1][tid] = (float)(s2 / count);
        if (blockIdx.y == 0) {
            dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
            dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
        }
    }
    __syncthreads();
    const int g8 = tid & 7, rl = tid >> 3;
    float mean[V], rstd[V], sc[V], k1[V], k2[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const int c = c0 + g8 * V + e;
        mean[e] = bnp[c]; rstd[e] = bnp[C + c]; sc[e] = bnp[2 * C + c]; k1[e] = cf[0][g8 * V + e]; k2[e] = cf[1][g8 * V + e];
    }
    const long r0 = (long)blockIdx.y * rows_per_wg, r1 = min(M, r0 + rows_per_wg);
    for (long r = r0 + rl; r < r1; r += 32) {
        const long off = r * C + c0 + g8 * V;
        float d[V], yv[V], g[V];
        load_vec<T>(dz_in + off, d);
        load_vec<T>(y + off, yv);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float xh = (yv[e] - mean[e]) * rstd[e];
            g[e] = sc[e] * (d[e] - k1[e] - xh * k2[e]);
        }
        store_vec<T>(dy + off, g);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// 3x3 stride-2 pad-1 max-pool (NHWC) with saved arg-max (first maximum in row-major window order, like ATen)
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const T* __restrict__ x, int B, int H, int W, int C, int OH, int OW, T* __restrict__ out,
                                                          uint8_t* __restrict__ idx) {
    constexpr int V = VecOf<T>::V;
    const int cch = C / V;
    const long total = (long)B * OH * OW * cch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int col = (int)(i % cch);
        const long pix = i / cch;
        const int ow = (int)(pix % OW), oh = (int)((pix / OW) % OH), b = (int)(pix / ((long)OW * OH));
        float best[V]; int bi[V];
#pragma unroll
        for (int e = 0; e < V; ++e) { best[e] = -INFINITY; bi[e] = 0; }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
            if (ih < 0 || ih >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw;
                if (iw < 0 || iw >= W) continue;
                float v[V];
                load_vec<T>(x + (((long)b * H + ih) * W + iw) * C + col * V, v);
#pragma unroll
                for (int e = 0; e < V; ++e)
                    if (v[e] > best[e] || v[e] != v[e]) { best[e] = v[e]; bi[e] = kh * 3 + kw; }
            }
        }
        store_vec<T>(out + i * V, best);
        uint8_t* ip = idx + i * V;
#pragma unroll
        for (int e = 0; e < V; ++e) ip[e] = (uint8_t)bi[e];
    }
}

template <typename T>
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ dout1, const uint8_t* __restrict__ idx, int B, int H,
                                                          int W, int C, int OH, int OW, T* __restrict__ dx) {
    constexpr int V = VecOf<T>::V;
    const int cch = C / V;
    const long total = (long)B * H * W * cch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int col = (int)(i % cch);
        const long pix = i / cch;
        const int iw = (int)(pix % W), ih = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
        float g[V];
#pragma unroll
        for (int e = 0; e < V; ++e) g[e] = 0.f;
        // windows (oh, ow) with ih = 2*oh - 1 + kh  =>  oh = (ih + 1 - kh)/2
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int t = ih + 1 - kh;
            if (t < 0 || (t & 1)) continue;
            const int oh = t >> 1;
            if (oh >= OH) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int u = iw + 1 - kw;
                if (u < 0 || (u & 1)) continue;
                const int ow = u >> 1;
                if (ow >= OW) continue;
                const long o = ((((long)b * OH + oh) * OW + ow) * cch + col) * V;
                float d[V];
                load_vec<T>(dout + o, d);
                if (dout1) {          // the pooled tensor had two consumers: their gradients are summed here
                    float t[V];
                    load_vec<T>(dout1 + o, t);
#pragma unroll
                    for (int e = 0; e < V; ++e) d[e] += t[e];
                }
                const uint8_t* ip = idx + o;
#pragma unroll
                for (int e = 0; e < V; ++e)
                    if (ip[e] == kh * 3 + kw) g[e] += d[e];
            }
        }
        store_vec<T>(dx + i * V, g);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Activation backward fused with the bias-gradient column sums: dz = dout * act'(out); partial[blk][C] = sum_rows dz
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) act_bwd_bias_kernel(const T* __restrict__ dout, const T* __restrict__ dout1, const T* __restrict__ out, int act, long M,
                                                           int C, long rows_per_block, T* __restrict__ dz, float* __restrict__ part) {
    constexpr int V = VecOf<T>::V;
    extern __shared__ float sh[];   // fast path: [256][V]; fallback: [C]
    const int cch = C / V;
    if (cch <= 256 && 256 % cch == 0) {
        const int col = threadIdx.x % cch, c0 = col * V;
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        const long total = M * cch;
        const long stride = (long)gridDim.x * 256;
        long i = (long)blockIdx.x * 256 + threadIdx.x;
        // four items per round trip: narrow full-resolution tensors (the disparity heads: 8-16 channels) give a thread only a handful of items, and
        // one dependent load at a time left the pass latency-bound (26 us for 24 MB)
        for (; i + 3 * stride < total; i += 4 * stride) {
            float d[4][V], o[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) load_vec<T>(dout + (i + u * stride) * V, d[u]);
            if (dout1) {              // second consumer of the activation (decoder level -> its disparity head and the next level)
#pragma unroll
                for (int u = 0; u < 4; ++u) load_vec<T>(dout1 + (i + u * stride) * V, o[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < V; ++e) d[u][e] += o[u][e];
            }
            if (act != SDE_ACT_NONE) {
#pragma unroll
                for (int u = 0; u < 4; ++u) load_vec<T>(out + (i + u * stride) * V, o[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        if (act == SDE_ACT_ELU) d[u][e] = o[u][e] > 0.f ? d[u][e] : d[u][e] * (o[u][e] + 1.0f);     // ELU'(x) = exp(x) = out + 1 for x <= 0
                        else d[u][e] = o[u][e] > 0.f ? d[u][e] : 0.f;
                    }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (dz) store_vec<T>(dz + (i + u * stride) * V, d[u]);
#pragma unroll
                for (int e = 0; e < V; ++e) acc[e] += d[u][e];
            }
        }
        for (; i < total; i += stride) {
            float d[V], o[V];
            load_vec<T>(dout + i * V, d);
            if (dout1) {
                load_vec<T>(dout1 + i * V, o);
#pragma unroll
                for (int e = 0; e < V; ++e) d[e] += o[e];
            }
            if (act != SDE_ACT_NONE) {
                load_vec<T>(out + i * V, o);
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    if (act == SDE_ACT_ELU) d[e] = o[e] > 0.f ? d[e] : d[e] * (o[e] + 1.0f);
                    else d[e] = o[e] > 0.f ? d[e] : 0.f;
                }
            }
            if (dz) store_vec<T>(dz + i * V, d);
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] += d[e];
        }
        if (part) {
            block_group_reduce<V>(acc, cch, sh);
            if (threadIdx.x < cch)
#pragma unroll
                for (int e = 0; e < V; ++e) part[(size_t)blockIdx.x * C + c0 + e] = acc[e];
        }
        return;
    }
    if (part) {
        for (int i = threadIdx.x; i < C; i += 256) sh[i] = 0.f;
        __syncthreads();
    }
    const long r0 = (long)blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    const long total = (r1 - r0) * cch;
    for (long i = threadIdx.x; i < total; i += 256) {
        const int c0 = (int)(i % cch) * V;
        const long off = (r0 * cch + i) * V;
        float d[V], o[V];
        load_vec<T>(dout + off, d);
        if (dout1) {
            load_vec<T>(dout1 + off, o);
#pragma unroll
            for (int e = 0; e < V; ++e) d[e] += o[e];
        }
        if (act != SDE_ACT_NONE) {
            load_vec<T>(out + off, o);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                if (act == SDE_ACT_ELU) d[e] = o[e] > 0.f ? d[e] : d[e] * (o[e] + 1.0f);
                else if (act == SDE_ACT_RELU) d[e] = o[e] > 0.f ? d[e] : 0.f;
            }
        }
        if (dz) store_vec<T>(dz + off, d);
        if (part) {
#pragma unroll
            for (int e = 0; e < V; ++e) atomicAdd(&sh[c0 + e], d[e]);
        }
    }
    if (part) {
        __syncthreads();
        for (int i = threadIdx.x; i < C; i += 256) part[(size_t)blockIdx.x * C + i] = sh[i];
    }
}

__global__ void __launch_bounds__(SLAB_T) colsum_finalize_kernel(const float* __restrict__ part, int nblk, int ld, int C, float* __restrict__ out, int accumulate) {
    __shared__ double sh[SLAB_SH];
    slab_colsum16(part, nblk, ld, blockIdx.x * 16, sh);
    const int c = blockIdx.x * 16 + threadIdx.x;
    if (threadIdx.x >= 16 || c >= C) return;
    const double sv = sh[SLAB_RES + threadIdx.x];
    out[c] = accumulate ? out[c] + (float)sv : (float)sv;
}

// The bias-gradient column sums of MANY layers in one launch (the per-layer finalize is a 5 us launch on the serial chain of the backward pass that nothing
// downstream waits for): block ranges per item, every item summed exactly as colsum_finalize_kernel sums it.  The table travels by value.
constexpr int COLSUM_MAX_ITEMS = 64;
struct ColsumBatch {
    const float* part[COLSUM_MAX_ITEMS];
    float* out[COLSUM_MAX_ITEMS];
    int rows[COLSUM_MAX_ITEMS], ld[COLSUM_MAX_ITEMS], C[COLSUM_MAX_ITEMS], accumulate[COLSUM_MAX_ITEMS];
    int first[COLSUM_MAX_ITEMS + 1];
    int n;
};
__global__ void __launch_bounds__(SLAB_T) colsum_finalize_batched_kernel(const ColsumBatch b) {
    __shared__ double sh[SLAB_SH];
    int k = 0;
    while (k + 1 < b.n && (int)blockIdx.x >= b.first[k + 1]) ++k;
    const int col0 = ((int)blockIdx.x - b.first[k]) * 16;
    slab_colsum16(b.part[k], b.rows[k], b.ld[k], col0, sh);
    const int c = col0 + threadIdx.x;
    if (threadIdx.x >= 16 || c >= b.C[k]) return;
    const double sv = sh[SLAB_RES + threadIdx.x];
    float* out = b.out[k];
    out[c] = b.accumulate[k] ? out[c] + (float)sv : (float)sv;
}

// ------------------------------------------------------------------------------------------------------------------
// Gradient folding for the decoder: reflection pad backward (+ nearest x2 upsample backward + concat split)
//   dxp [B, H+2, W+2, C] is the gradient w.r.t. the reflection-PADDED virtual input (what the data-gradient GEMM emits).
//   plain : dx0 [B,H,W,C]            = fold(dxp)
//   upcat : dx0 [B,H/2,W/2,C0]       = sum over the 2x2 upsampled block of fold(dxp)[..., :C0]
//           dx1 [B,H,W,C1]           = fold(dxp)[..., C0:]
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void fold_at(const T* __restrict__ dxp, int b, int h, int w, int H, int W, int C, int c0, float* g) {
    constexpr int V = VecOf<T>::V;
#pragma unroll
    for (int e = 0; e < V; ++e) g[e] = 0.f;
    // padded rows that ReflectionPad2d(1) maps onto row h: h+1 always, 0 if h == 1, H+1 if h == H-2 (both when H == 3)
    int hs[3], ws[3], nh = 0, nw = 0;
    hs[nh++] = h + 1;
    if (h == 1) hs[nh++] = 0;
    if (h == H - 2) hs[nh++] = H + 1;
    ws[nw++] = w + 1;
    if (w == 1) ws[nw++] = 0;
    if (w == W - 2) ws[nw++] = W + 1;
    for (int a = 0; a < nh; ++a)
        for (int bb = 0; bb < nw; ++bb) {
            float v[V];
            load_vec<T>(dxp + (((long)b * (H + 2) + hs[a]) * (W + 2) + ws[bb]) * C + c0, v);
#pragma unroll
            for (int e = 0; e < V; ++e) g[e] += v[e];
        }
}

template <typename T>
__global__ void __launch_bounds__(256) refl_fold_kernel(const T* __restrict__ dxp, int B, int H, int W, int C, int C0, int upcat, T* __restrict__ dx0,
                                                        T* __restrict__ dx1) {
    constexpr int V = VecOf<T>::V;
    if (!upcat) {
        const int cch = C / V;
        const long total = (long)B * H * W * cch;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int col = (int)(i % cch);
            const long pix = i / cch;
            const int w = (int)(pix % W), h = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
            float g[V];
            fold_at<T>(dxp, b, h, w, H, W, C, col * V, g);
            store_vec<T>(dx0 + i * V, g);
        }
        return;
    }
    const int C1 = C - C0;
    const int c0ch = C0 / V, c1ch = C1 / V;
    const int H2 = H / 2, W2 = W / 2;
    const long n0 = (long)B * H2 * W2 * c0ch, n1 = (long)B * H * W * c1ch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n0 + n1; i += (long)gridDim.x * 256) {
        if (i < n0) {
            const int col = (int)(i % c0ch);
            const long pix = i / c0ch;
            const int w = (int)(pix % W2), h = (int)((pix / W2) % H2), b = (int)(pix / ((long)W2 * H2));
            float acc[V];
#pragma unroll
            for (int e = 0; e < V; ++e) acc[e] = 0.f;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int bb = 0; bb < 2; ++bb) {
                    float g[V];
                    fold_at<T>(dxp, b, 2 * h + a, 2 * w + bb, H, W, C, col * V, g);
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[e] += g[e];
                }
            store_vec<T>(dx0 + i * V, acc);
        } else {
            const long j = i - n0;
            const int col = (int)(j % c1ch);
            const long pix = j / c1ch;
            const int w = (int)(pix % W), h = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
            float g[V];
            fold_at<T>(dxp, b, h, w, H, W, C, C0 + col * V, g);
            store_vec<T>(dx1 + j * V, g);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Depth head tail: depth = 1 / (min_disp + (max_disp - min_disp) * softplus(y[..., 0])), optional horizontal flip
// ------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) depth_head_fwd_kernel(const T* __restrict__ y, int B, int H, int W, int ld, float min_disp, float max_disp, int flip,
                                                             float* __restrict__ depth) {
    const long n = (long)B * H * W;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = (float)y[i * ld];
        const float sp = v > 20.f ? v : log1pf(expf(v));
        const float sd = min_disp + (max_disp - min_disp) * sp;
        long o = i;
        if (flip) { const int x = (int)(i % W); o = i - x + (W - 1 - x); }
        depth[o] = 1.0f / sd;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) depth_head_bwd_kernel(const T* __restrict__ y, const float* __restrict__ ddepth, int B, int H, int W, int ld,
                                                             float min_disp, float max_disp, int flip, T* __restrict__ dy, float* __restrict__ bias_part) {
    __shared__ float red[16];
    const long n = (long)B * H * W;
    float bsum = 0.f;       // sum of the (storage-rounded) logit gradients of this workgroup: the bias gradient of the one-channel convolution in front
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = (float)y[i * ld];
        const float sp = v > 20.f ? v : log1pf(expf(v));
        const float sd = min_disp + (max_disp - min_disp) * sp;
        const float dsp = v > 20.f ? 1.f : 1.0f / (1.0f + expf(-v));
        long o = i;
        if (flip) { const int x = (int)(i % W); o = i - x + (W - 1 - x); }
        const float g = ddepth[o] * (-1.0f / (sd * sd)) * (max_disp - min_disp) * dsp;
        T* d = dy + i * ld;
        d[0] = (T)g;
        bsum += (float)(T)g;
        for (int c = 1; c < ld; ++c) d[c] = (T)0.f;
    }
    if (bias_part) {
        bsum = sde_block_sum(bsum, red);
        if (threadIdx.x == 0) bias_part[blockIdx.x] = bsum;
    }
}

// dbias[0] (+)= sum of the per-workgroup partials of depth_head_bwd_kernel, fixed order
__global__ void __launch_bounds__(256) head_bias_finalize_kernel(const float* __restrict__ part, int n, float* __restrict__ dbias, int accumulate) {
    __shared__ float red[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    s = sde_block_sum(s, red);
    if (threadIdx.x == 0) dbias[0] = accumulate ? dbias[0] + s : s;
}

// ------------------------------------------------------------------------------------------------------------------
// GroupNorm(G) + ReLU, NHWC.  Stage 1: per (sample, chunk-of-pixels) per-channel (sum, sum^2) partials.
// ------------------------------------------------------------------------------------------------------------------
constexpr int GN_CHUNKS = SDE_GN_CHUNKS;    // pixel chunks per sample (workgroups of the statistics kernels: B x GN_CHUNKS)

// activation fused behind the normalisation: 0 none, 1 ReLU (PoseNet.py:L13-20), 2 ELU (layers01.py:L33-40); the derivative is taken
// from the stored output, as the reference's in-place activations do
__device__ __forceinline__ float gn_act(float v, int act) { return act == 1 ? fmaxf(v, 0.f) : (act == 2 ? (v > 0.f ? v : expm1f(v)) : v); }
__device__ __forceinline__ float gn_act_grad(float d, float out, int act) {
    return act == 1 ? (out > 0.f ? d : 0.f) : (act == 2 ? (out > 0.f ? d : d * (out + 1.0f)) : d);
}

// The 256/C threads that walked channel c = tid % C combine their partial sums in thread order (no float atomics: the statistics,
// and with them every PoseNet gradient, are bit-reproducible from run to run).  All 256 threads must call it; sh = [2][C].
__device__ __forceinline__ void fixed_order_channel_sum(float a1, float a2, int C, float* sh) {
    __shared__ float red[512];
    red[threadIdx.x] = a1; red[256 + threadIdx.x] = a2;
    __syncthreads();
    if ((int)threadIdx.x < C) {
        float t1 = 0.f, t2 = 0.f;
        for (int j = threadIdx.x; j < 256; j += C) { t1 += red[j]; t2 += red[256 + j]; }
        sh[threadIdx.x] = t1; sh[C + threadIdx.x] = t2;
    }
}

// 16-byte variant: thread t walks the channel group t % cch (cch = C / V, a power of two <= 256); the 256 / cch threads of a group are
// combined by a fixed binary tree in LDS (deterministic).  a1 / a2: the thread's V partial sums each.  sh = [2][C].
template <int V>
__device__ __forceinline__ void fixed_order_group_sum(const float* a1, const float* a2, int cch, float* sh, int C) {
    __shared__ float red[256 * 2 * V];
#pragma unroll
    for (int e = 0; e < V; ++e) { red[(threadIdx.x * 2) * V + e] = a1[e]; red[(threadIdx.x * 2 + 1) * V + e] = a2[e]; }
    __syncthreads();
    // fixed binary tree over the 256 / cch threads of a channel group (threads t and t + s share a group: s is a multiple of cch)
    for (int s = 128; s >= cch; s >>= 1) {
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int e = 0; e < 2 * V; ++e) red[threadIdx.x * 2 * V + e] += red[(threadIdx.x + s) * 2 * V + e];
        __syncthreads();
    }
    if ((int)threadIdx.x < cch)
#pragma unroll
        for (int e = 0; e < V; ++e) { sh[threadIdx.x * V + e] = red[(threadIdx.x * 2) * V + e]; sh[C + threadIdx.x * V + e] = red[(threadIdx.x * 2 + 1) * V + e]; }
}

template <typename T>
__global__ void __launch_bounds__(256) gn_stats_kernel(const T* __restrict__ x, const T* __restrict__ res, int HW, int C, float* __restrict__ part /*[B][GN_CHUNKS][C][2]*/) {
    // res (optional, vector path only -- host-checked): the normalised tensor is x + res (layers01.py:L74-76 ResidualConv: x_out + shortcut), summed in fp32
    extern __shared__ float sh[];   // [2][C]
    const int b = blockIdx.y, ch = blockIdx.x;
    for (int i = threadIdx.x; i < 2 * C; i += 256) sh[i] = 0.f;
    __syncthreads();
    const int per = (HW + GN_CHUNKS - 1) / GN_CHUNKS;
    const int p0 = ch * per, p1 = min(HW, p0 + per);
    const long total = (long)(p1 - p0) * C;
    const T* base = x + ((long)b * HW + p0) * C;
    float a1 = 0.f, a2 = 0.f;
    constexpr int V = VecOf<T>::V;
    if (C % V == 0 && 256 % (C / V) == 0) {          // 16 bytes per lane, fixed channel group per thread (256 % cch == 0 => cch is a power of two)
        const int cch = C / V;
        float v1[V], v2[V];
#pragma unroll
        for (int e = 0; e < V; ++e) { v1[e] = 0.f; v2[e] = 0.f; }
        const long groups = (long)(p1 - p0) * cch;
        const T* rbase = res ? res + ((long)b * HW + p0) * C : nullptr;
        for (long i = threadIdx.x; i < groups; i += 256) {
            float v[V];
            load_vec<T>(base + i * V, v);
            if (rbase) {
                float r[V];
                load_vec<T>(rbase + i * V, r);
#pragma unroll
                for (int e = 0; e < V; ++e) v[e] += r[e];
            }
#pragma unroll
            for (int e = 0; e < V; ++e) { v1[e] += v[e]; v2[e] += v[e] * v[e]; }
        }
        fixed_order_group_sum<V>(v1, v2, cch, sh, C);
    } else
    // thread walks elements with a fixed channel when 256 % C == 0 (C <= 256), else falls back to shared atomics per element
    if (256 % C == 0) {
        const int c = threadIdx.x % C;
        for (long i = threadIdx.x; i < total; i += 256) { const float v = (float)base[i]; a1 += v; a2 += v * v; }
        (void)c;
        fixed_order_channel_sum(a1, a2, C, sh);
    } else if (C % 256 == 0 && C <= 2048) {
        // channel c is walked by exactly one thread (c % 256), in slot c / 256: no cross-thread combination at all
        float b1[8], b2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { b1[j] = 0.f; b2[j] = 0.f; }
        const int slots = C / 256;
        int slot = 0;
        for (long i = threadIdx.x; i < total; i += 256) {
            const float v = (float)base[i];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j == slot) { b1[j] += v; b2[j] += v * v; }
            if (++slot == slots) slot = 0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < slots) { sh[threadIdx.x + 256 * j] = b1[j]; sh[C + threadIdx.x + 256 * j] = b2[j]; }
    } else {
        for (long i = threadIdx.x; i < total; i += 256) {
            const int c = (int)(i % C);
            const float v = (float)base[i];
            atomicAdd(&sh[c], v); atomicAdd(&sh[C + c], v * v);
        }
    }
    __syncthreads();
    float* o = part + (((size_t)b * GN_CHUNKS + ch) * C) * 2;
    for (int i = threadIdx.x; i < C; i += 256) { o[i * 2] = sh[i]; o[i * 2 + 1] = sh[C + i]; }
}

__device__ __forceinline__ double wave_sum_f64(double v) {       // fixed butterfly order: deterministic
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// gnp [B][G][2] = mean, rstd.  One wave per (sample, group): the lanes split the GN_CHUNKS x cpg partials.
__global__ void __launch_bounds__(64) gn_finalize_kernel(const float* __restrict__ part, int B, int C, int G, int HW, float eps, float* __restrict__ gnp) {
    const int i = blockIdx.x;                    // (b, g)
    const int b = i / G, gidx = i % G, cpg = C / G;
    double s1 = 0, s2 = 0;
    for (int q = threadIdx.x; q < GN_CHUNKS * cpg; q += 64) {
        const int ch = q / cpg, c = gidx * cpg + q % cpg;
        s1 += part[(((size_t)b * GN_CHUNKS + ch) * C + c) * 2];
        s2 += part[(((size_t)b * GN_CHUNKS + ch) * C + c) * 2 + 1];
    }
    s1 = wave_sum_f64(s1); s2 = wave_sum_f64(s2);
    if (threadIdx.x == 0) {
        const double n = (double)HW * cpg, mean = s1 / n;
        double var = s2 / n - mean * mean;
        if (var < 0) var = 0;
        gnp[i * 2] = (float)mean;
        gnp[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

template <typename T>
__global__ void __launch_bounds__(256) gn_apply_kernel(const T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ gnp, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, int B, int HW, int C, int G, int relu, T* __restrict__ out) {
    constexpr int V = VecOf<T>::V;                 // 16 bytes per lane
    const int cch = C / V, cpg = C / G;
    const long total = (long)B * HW * cch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c0 = (int)(i % cch) * V, b = (int)(i / ((long)HW * cch));
        float v[V];
        load_vec<T>(x + i * V, v);
        if (res) {
            float r[V];
            load_vec<T>(res + i * V, r);
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] += r[e];
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float* st = gnp + ((size_t)b * G + (c0 + e) / cpg) * 2;
            v[e] = gn_act((v[e] - st[0]) * st[1] * gamma[c0 + e] + beta[c0 + e], relu);
        }
        store_vec<T>(out + i * V, v);
    }
}

// backward stage 1: per (sample, chunk) per-channel partial sums of dz and dz*xhat (dz = dout * relu mask)
template <typename T>
__global__ void __launch_bounds__(256) gn_bwd_stats_kernel(const T* __restrict__ dout, const T* __restrict__ out, const T* __restrict__ x, const T* __restrict__ res,
                                                           const float* __restrict__ gnp, int HW, int C, int G, int relu,
                                                           float* __restrict__ part /*[B][GN_CHUNKS][C][2]*/) {
    extern __shared__ float sh[];
    const int b = blockIdx.y, ch = blockIdx.x, cpg = C / G;
    for (int i = threadIdx.x; i < 2 * C; i += 256) sh[i] = 0.f;
    __syncthreads();
    const int per = (HW + GN_CHUNKS - 1) / GN_CHUNKS;
    const int p0 = ch * per, p1 = min(HW, p0 + per);
    const long total = (long)(p1 - p0) * C;
    const long base = ((long)b * HW + p0) * C;
    constexpr int V = VecOf<T>::V;
    if (C % V == 0 && 256 % (C / V) == 0) {
        const int cch = C / V, c0 = (threadIdx.x % cch) * V;
        float m_[V], r_[V], v1[V], v2[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float* st = gnp + ((size_t)b * G + (c0 + e) / cpg) * 2;
            m_[e] = st[0]; r_[e] = st[1]; v1[e] = 0.f; v2[e] = 0.f;
        }
        const long groups = (long)(p1 - p0) * cch;
        for (long i = threadIdx.x; i < groups; i += 256) {
            float d[V], o[V], xv[V];
            load_vec<T>(dout + base + i * V, d);
            if (relu) load_vec<T>(out + base + i * V, o);
            load_vec<T>(x + base + i * V, xv);
            if (res) {
                float rr[V];
                load_vec<T>(res + base + i * V, rr);
#pragma unroll
                for (int e = 0; e < V; ++e) xv[e] += rr[e];
            }
#pragma unroll
            for (int e = 0; e < V; ++e) {
                float de = d[e];
                if (relu) de = gn_act_grad(de, o[e], relu);
                v1[e] += de; v2[e] += de * ((xv[e] - m_[e]) * r_[e]);
            }
        }
        fixed_order_group_sum<V>(v1, v2, cch, sh, C);
    } else
    if (256 % C == 0) {
        const int c = threadIdx.x % C;
        const float* st = gnp + ((size_t)b * G + c / cpg) * 2;
        float a1 = 0.f, a2 = 0.f;
        for (long i = threadIdx.x; i < total; i += 256) {
            float d = (float)dout[base + i];
            if (relu) d = gn_act_grad(d, (float)out[base + i], relu);
            const float xh = ((float)x[base + i] - st[0]) * st[1];
            a1 += d; a2 += d * xh;
        }
        fixed_order_channel_sum(a1, a2, C, sh);
    } else if (C % 256 == 0 && C <= 2048) {
        float b1[8], b2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { b1[j] = 0.f; b2[j] = 0.f; }
        const int slots = C / 256;
        int slot = 0;
        for (long i = threadIdx.x; i < total; i += 256) {
            const int c = threadIdx.x + 256 * slot;
            const float* st = gnp + ((size_t)b * G + c / cpg) * 2;
            float d = (float)dout[base + i];
            if (relu) d = gn_act_grad(d, (float)out[base + i], relu);
            const float xh = ((float)x[base + i] - st[0]) * st[1];
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j == slot) { b1[j] += d; b2[j] += d * xh; }
            if (++slot == slots) slot = 0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (j < slots) { sh[threadIdx.x + 256 * j] = b1[j]; sh[C + threadIdx.x + 256 * j] = b2[j]; }
    } else {
        for (long i = threadIdx.x; i < total; i += 256) {
            const int c = (int)(i % C);
            const float* st = gnp + ((size_t)b * G + c / cpg) * 2;
            float d = (float)dout[base + i];
            if (relu) d = gn_act_grad(d, (float)out[base + i], relu);
            const float xh = ((float)x[base + i] - st[0]) * st[1];
            atomicAdd(&sh[c], d); atomicAdd(&sh[C + c], d * xh);
        }
    }
    __syncthreads();
    float* o = part + (((size_t)b * GN_CHUNKS + ch) * C) * 2;
    for (int i = threadIdx.x; i < C; i += 256) { o[i * 2] = sh[i]; o[i * 2 + 1] = sh[C + i]; }
}

// blocks [0, B*G): per (b, g) m1 = mean_g(gamma*dz), m2 = mean_g(gamma*dz*xhat); blocks [B*G, B*G + C): per channel
// dgamma = sum_b sum dz*xhat, dbeta = sum_b sum dz.  One wave each, lanes split the partials.
__global__ void __launch_bounds__(64) gn_bwd_finalize_kernel(const float* __restrict__ part, const float* __restrict__ gamma, int B, int C, int G, int HW,
                                                             float* __restrict__ coef /*[B][G][2]*/, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             int accumulate) {
    const int cpg = C / G;
    double s1 = 0, s2 = 0;
    if ((int)blockIdx.x < B * G) {
        const int i = blockIdx.x, b = i / G, gidx = i % G;
        for (int q = threadIdx.x; q < GN_CHUNKS * cpg; q += 64) {
            const int ch = q / cpg, c = gidx * cpg + q % cpg;
            s1 += (double)gamma[c] * part[(((size_t)b * GN_CHUNKS + ch) * C + c) * 2];
            s2 += (double)gamma[c] * part[(((size_t)b * GN_CHUNKS + ch) * C + c) * 2 + 1];
        }
        s1 = wave_sum_f64(s1); s2 = wave_sum_f64(s2);
        if (threadIdx.x == 0) {
            const double n = (double)HW * cpg;
            coef[i * 2] = (float)(s1 / n); coef[i * 2 + 1] = (float)(s2 / n);
        }
    } else {
        const int c = blockIdx.x - B * G;
        for (int q = threadIdx.x; q < B * GN_CHUNKS; q += 64) {
            s1 += part[((size_t)q * C + c) * 2];               // q = b * GN_CHUNKS + ch
            s2 += part[((size_t)q * C + c) * 2 + 1];
        }
        s1 = wave_sum_f64(s1); s2 = wave_sum_f64(s2);
        if (threadIdx.x == 0) {
            dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
            dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ out, const T* __restrict__ x, const T* __restrict__ res,
                                                           const float* __restrict__ gnp, const float* __restrict__ coef, const float* __restrict__ gamma,
                                                           int B, int HW, int C, int G, int relu, T* __restrict__ dx) {
    constexpr int V = VecOf<T>::V;
    const int cch = C / V, cpg = C / G;
    const long total = (long)B * HW * cch;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c0 = (int)(i % cch) * V, b = (int)(i / ((long)HW * cch));
        float d[V], o[V], xv[V];
        load_vec<T>(dout + i * V, d);
        if (relu) load_vec<T>(out + i * V, o);
        load_vec<T>(x + i * V, xv);
        if (res) {
            float rr[V];
            load_vec<T>(res + i * V, rr);
#pragma unroll
            for (int e = 0; e < V; ++e) xv[e] += rr[e];
        }
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const size_t gi = ((size_t)b * G + (c0 + e) / cpg) * 2;
            float de = d[e];
            if (relu) de = gn_act_grad(de, o[e], relu);
            const float xh = (xv[e] - gnp[gi]) * gnp[gi + 1];
            d[e] = gnp[gi + 1] * (gamma[c0 + e] * de - coef[gi] - xh * coef[gi + 1]);
        }
        store_vec<T>(dx + i * V, d);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Fused Adam / AdamW over a flat fp32 parameter buffer split into segments with their own lr / weight decay
// ------------------------------------------------------------------------------------------------------------------
// The whole descriptor (segment table, learning rates, bias corrections) travels BY VALUE in the kernel arguments: the host changes the
// learning rate every iteration (polynomial decay) and nothing has to be uploaded, kept alive or ordered against the launch.
// scale_state (optional, fp16 training): device {loss_scale, found_inf, growth_tracker}: gradients are divided by loss_scale and the
// whole update is skipped when found_inf != 0 (engine/train_loop.py:L294-341 AMPTrainer / GradScaler.step semantics, without a host sync).
__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, long n,
                                                   const sde_adam_desc d) {
    float gs = d.grad_scale;
    float bc1 = d.bias_corr1, bc2 = d.bias_corr2;
    if (d.scale_state) {
        if (d.scale_state[1] != 0.f) return;            // overflow in this step's gradients: skip (uniform over the grid)
        gs /= d.scale_state[0];
        // the step count of the APPLIED updates lives on the device (a skipped step must not advance Adam's `step`): state[3], incremented
        // by loss_scale_update_kernel behind this launch
        const double t = (double)d.scale_state[3] + 1.0;
        bc1 = (float)(1.0 - pow(d.beta1_d, t));
        bc2 = (float)(1.0 - pow(d.beta2_d, t));
    }
    const float beta1 = d.beta1, beta2 = d.beta2, rbc2 = 1.0f / sqrtf(bc2);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        int s = 0;
        while (s < d.nseg - 1 && i >= d.seg_end[s]) ++s;
        const float lr = d.seg_lr[s], wd = d.seg_wd[s];
        float gi = g[i] * gs, pi = p[i];
        if (d.decoupled_wd) pi *= (1.0f - lr * wd);     // AdamW: torch.optim.AdamW
        else gi += wd * pi;                             // Adam with L2 (wd = 0 on this path)
        const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) * rbc2 + d.eps;
        p[i] = pi - (lr / bc1) * (mi / denom);
    }
}

// found_inf = any non-finite gradient (GradScaler's unscale_/found_inf check): one pass over the flat gradient, 16 B per lane; a workgroup
// that sees one raises state[1] (idempotent plain store of the same value: no atomics, deterministic).
__global__ void __launch_bounds__(256) grad_check_kernel(const float4* __restrict__ g4, long n4, const float* __restrict__ tail, int ntail, float* __restrict__ state) {
    bool bad = false;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 q = g4[i];
        const float t = fabsf(q.x) + fabsf(q.y) + fabsf(q.z) + fabsf(q.w);      // inf or nan in any lane makes t non-finite
        bad |= !(t <= 3.0e38f);
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) bad |= !(fabsf(tail[threadIdx.x]) <= 3.0e38f);
    if (__any(bad) && (threadIdx.x & 63) == 0) state[1] = 1.0f;
}

// GradScaler.update(): found_inf -> scale *= backoff, tracker = 0; else tracker += 1 and at growth_interval scale *= growth, tracker = 0.
// state[3] counts the optimizer steps that were applied (not skipped).
__global__ void loss_scale_update_kernel(float* __restrict__ state, float growth, float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (state[1] != 0.f) { state[0] *= backoff; state[2] = 0.f; }
    else {
        state[3] += 1.f;                                 // one more optimizer step was applied (exact in fp32 up to 2^24 steps)
        const float t = state[2] + 1.f;
        if (t >= (float)interval) { state[0] *= growth; state[2] = 0.f; } else state[2] = t;
    }
    state[1] = 0.f;
}

// If the slab has more than SDE_REDUCE_ROWS rows, fold it into SDE_REDUCE_ROWS rows stored right behind it (the caller
// allocates rows + SDE_REDUCE_ROWS rows).  Returns the pointer / row count the finalize kernel should read.
const float* pre_reduce(const float* part, int& rows, int width, hipStream_t s) {
    // only slabs taller than this take the extra launch: the finalize kernels keep 256 rows in flight per round trip
    constexpr int thr = 4096;
    if (rows <= thr) return part;
    float* out = const_cast<float*>(part) + (size_t)rows * width;
    const int chunk = (rows + SDE_REDUCE_ROWS - 1) / SDE_REDUCE_ROWS;
    const int rows_out = (rows + chunk - 1) / chunk;
    hipLaunchKernelGGL(rows_reduce_kernel, dim3((width + 63) / 64, rows_out), dim3(1024), 0, s, part, rows, width, chunk, out);
    rows = rows_out;
    return out;
}

int grid_for(long n_items) {
    long nb = (n_items + 255) / 256;
    if (nb < 1) nb = 1;
    if (nb > 8192) nb = 8192;
    return (int)nb;
}

}  // namespace

#define DISPATCH_T(dtype, CALL_F32, CALL_BF16, CALL_F16) \
    do {                                        \
        if ((dtype) == SDE_BF16) { CALL_BF16; }  \
        else if ((dtype) == SDE_F16) { CALL_F16; } \
        else { CALL_F32; }                      \
    } while (0)

extern "C" {

int sde_prep_input(const float* img, const float* mean, const float* std_, int B, int C, int H, int W, int Cpad, int flip, int dtype, void* out,
                   sde_stream_t stream) {
    SDE_CHECK_ARG(img && out && B > 0 && C > 0 && H > 0 && W > 0 && Cpad >= C, "sde_prep_input: bad argument");
    SDE_CHECK_ARG((mean == nullptr) == (std_ == nullptr), "sde_prep_input: mean/std must both be given or both be null");
    hipStream_t s = (hipStream_t)stream;
    const int nb = grid_for((long)B * H * W);
    DISPATCH_T(dtype, hipLaunchKernelGGL(prep_input_kernel<float>, dim3(nb), dim3(256), 0, s, img, mean, std_, B, C, H, W, Cpad, flip, (float*)out),
               hipLaunchKernelGGL(prep_input_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, img, mean, std_, B, C, H, W, Cpad, flip, (bf16_t*)out),
               hipLaunchKernelGGL(prep_input_kernel<half_t>, dim3(nb), dim3(256), 0, s, img, mean, std_, B, C, H, W, Cpad, flip, (half_t*)out));
    SDE_CHECK_LAUNCH("sde_prep_input");
    return SDE_OK;
}

int sde_bn_finalize(const float* part, int tiles, int C, long count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                    float momentum, float eps, float* bnp, sde_stream_t stream) {
    SDE_CHECK_ARG(part && gamma && beta && bnp && tiles > 0 && C > 0 && count > 0, "sde_bn_finalize: bad argument");
    int rows = tiles;
    const float* src = pre_reduce(part, rows, 2 * C, (hipStream_t)stream);
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(sde_cdiv(C, 8)), dim3(SLAB_T), 0, (hipStream_t)stream, src, rows, C, (float)count, gamma, beta, running_mean,
                       running_var, momentum, eps, bnp);
    SDE_CHECK_LAUNCH("sde_bn_finalize");
    return SDE_OK;
}

static int g_bn_fuse = 1;       // sde_bn_set_fuse(0): always the separate finalize and apply launches (A/B, tests)
static int g_bn_wide = 1;       // sde_bn_set_fuse(2): fused for short slabs only, the big layers keep 256-thread reduce workgroups + three launches (A/B)
int sde_bn_set_fuse(int on) {
    const int old = g_bn_fuse ? (g_bn_wide ? 1 : 2) : 0;
    g_bn_fuse = on ? 1 : 0;
    g_bn_wide = on == 1;
    return old;
}

int sde_bn_finalize_apply_ok(int tiles, int C, int dtype) { return SDE_IS16(dtype) && C % 64 == 0 && tiles >= 1 && tiles <= BNFA_MAX_ROWS; }

int sde_bn_finalize_apply(const float* part, int tiles, int C, long count, const float* gamma, const float* beta, float* running_mean, float* running_var,
                          float momentum, float eps, float* bnp, const void* y, const void* residual, int relu, int dtype, void* out, sde_stream_t stream) {
    SDE_CHECK_ARG(part && gamma && beta && bnp && y && out && count > 0, "sde_bn_finalize_apply: bad argument");
    SDE_CHECK_ARG(sde_bn_finalize_apply_ok(tiles, C, dtype), "sde_bn_finalize_apply: needs a 16-bit type, C %% 64 == 0 and <= %d slab rows (tiles=%d C=%d)",
                  BNFA_MAX_ROWS, tiles, C);
    const long M = count;
    const int strips = C / 64;
    // enough workgroups to stream at full rate (>= ~1024), rows per workgroup a multiple of the 32-row pass
    long chunks = (1024 + strips - 1) / strips;
    long rpw = (M + chunks - 1) / chunks;
    rpw = (rpw + 31) / 32 * 32;
    if (rpw < 32) rpw = 32;
    chunks = (M + rpw - 1) / rpw;
    const dim3 grid((unsigned)strips, (unsigned)chunks);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SDE_BF16)
        hipLaunchKernelGGL(bn_finalize_apply_kernel<bf16_t>, grid, dim3(256), 0, s, part, tiles, C, (float)count, gamma, beta, running_mean, running_var, momentum, eps,
                           bnp, (const bf16_t*)y, (const bf16_t*)residual, relu, M, rpw, (bf16_t*)out);
    else
        hipLaunchKernelGGL(bn_finalize_apply_kernel<half_t>, grid, dim3(256), 0, s, part, tiles, C, (float)count, gamma, beta, running_mean, running_var, momentum, eps,
                           bnp, (const half_t*)y, (const half_t*)residual, relu, M, rpw, (half_t*)out);
    SDE_CHECK_LAUNCH("sde_bn_finalize_apply");
    return SDE_OK;
}

int sde_bn_eval_params(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps, int C, float* bnp,
                       sde_stream_t stream) {
    SDE_CHECK_ARG(gamma && beta && running_mean && running_var && bnp && C > 0, "sde_bn_eval_params: bad argument");
    hipLaunchKernelGGL(bn_eval_params_kernel, dim3(sde_cdiv(C, 64)), dim3(64), 0, (hipStream_t)stream, gamma, beta, running_mean, running_var, eps, C, bnp);
    SDE_CHECK_LAUNCH("sde_bn_eval_params");
    return SDE_OK;
}

int sde_bn_apply(const void* y, const float* bnp, const void* residual, int relu, long M, int C, int dtype, void* out, sde_stream_t stream) {
    const int V = SDE_IS16(dtype) ? 8 : 4;
    SDE_CHECK_ARG(y && bnp && out && M > 0 && C > 0 && C % V == 0, "sde_bn_apply: bad argument (C=%d)", C);
    hipStream_t s = (hipStream_t)stream;
    const int nb = grid_for(M * (C / V));
    DISPATCH_T(dtype, hipLaunchKernelGGL(bn_apply_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)y, bnp, (const float*)residual, relu, M, C, (float*)out),
               hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)y, bnp, (const bf16_t*)residual, relu, M, C, (bf16_t*)out),
               hipLaunchKernelGGL(bn_apply_kernel<half_t>, dim3(nb), dim3(256), 0, s, (const half_t*)y, bnp, (const half_t*)residual, relu, M, C, (half_t*)out));
    SDE_CHECK_LAUNCH("sde_bn_apply");
    return SDE_OK;
}

int sde_reduce_num_blocks(long M, int C) {
    // ~4 sixteen-byte groups per thread (bf16 grouping), enough workgroups to keep HBM busy: these passes are pure streaming
    constexpr long cap = 1024;      // (2048 measured the same for the streaming pass and costs the finalize kernels a second round trip)
    long nb = (M * (long)C / 8 + 1023) / 1024;
    if (nb > cap) nb = cap;
    if (nb < 1) nb = 1;
    return (int)nb;
}

int sde_bn_bwd(const void* dout, const void* dout1, const void* dout2, const void* out, const void* y, const float* bnp, const float* gamma, int relu,
               long M, int C, int dtype, float* part, float* coef, float* dgamma, float* dbeta, int accumulate_params, void* gm, void* dy,
               sde_stream_t stream) {
    const int V = SDE_IS16(dtype) ? 8 : 4;
    SDE_CHECK_ARG(dout && y && bnp && part && coef && dgamma && dbeta && dy && M > 0 && C > 0 && C % V == 0, "sde_bn_bwd: bad argument");
    if (relu && !out) relu = 2;          // BatchNorm + ReLU without a residual: the mask is re-derived from y and the BN parameters
    SDE_CHECK_ARG(gm || (!relu && !dout1 && !dout2), "sde_bn_bwd: a masked or summed gradient needs the gm buffer");
    SDE_CHECK_ARG(dout1 || !dout2, "sde_bn_bwd: dout2 without dout1");
    (void)gamma;
    hipStream_t s = (hipStream_t)stream;
    int nblk = sde_reduce_num_blocks(M, C);
    const long rpb = (M + nblk - 1) / nblk;
    const size_t lds = (2 * (size_t)C > 256 * 16 ? 2 * (size_t)C : 256 * 16) * sizeof(float);
    const int cch = C / V;
    const bool wide = g_bn_fuse && g_bn_wide && SDE_IS16(dtype) && C % 64 == 0 && nblk > BNFA_MAX_ROWS && cch <= 256 && 256 % cch == 0;
    if (wide) {
        // big layers: 1024-thread workgroups, a quarter as many of them -> at most 256 partial rows, and finalize + apply become one launch below
        nblk = (nblk + 3) / 4;
        if (dtype == SDE_BF16)
            hipLaunchKernelGGL((bn_bwd_reduce_kernel<bf16_t, 1024>), dim3(nblk), dim3(1024), 1024 * 16 * sizeof(float), s, (const bf16_t*)dout, (const bf16_t*)dout1,
                               (const bf16_t*)dout2, (const bf16_t*)out, (const bf16_t*)y, bnp, relu, M, C, rpb, part, (bf16_t*)gm);
        else
            hipLaunchKernelGGL((bn_bwd_reduce_kernel<half_t, 1024>), dim3(nblk), dim3(1024), 1024 * 16 * sizeof(float), s, (const half_t*)dout, (const half_t*)dout1,
                               (const half_t*)dout2, (const half_t*)out, (const half_t*)y, bnp, relu, M, C, rpb, part, (half_t*)gm);
    } else
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(nblk), dim3(256), lds, s, (const float*)dout, (const float*)dout1, (const float*)dout2, (const float*)out, (const float*)y, bnp, relu, M, C, rpb, part, (float*)gm),
               hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, dim3(nblk), dim3(256), lds, s, (const bf16_t*)dout, (const bf16_t*)dout1, (const bf16_t*)dout2, (const bf16_t*)out, (const bf16_t*)y, bnp, relu, M, C, rpb, part, (bf16_t*)gm),
               hipLaunchKernelGGL(bn_bwd_reduce_kernel<half_t>, dim3(nblk), dim3(256), lds, s, (const half_t*)dout, (const half_t*)dout1, (const half_t*)dout2, (const half_t*)out, (const half_t*)y, bnp, relu, M, C, rpb, part, (half_t*)gm));
    SDE_CHECK_LAUNCH("sde_bn_bwd/reduce");
    const void* dz = gm ? gm : dout;
    if (g_bn_fuse && SDE_IS16(dtype) && C % 64 == 0 && nblk <= BNFA_MAX_ROWS) {       // short slab: finalize + apply in one launch
        const int strips = C / 64;
        long chunks = (1024 + strips - 1) / strips;
        long rpw = (M + chunks - 1) / chunks;
        rpw = (rpw + 31) / 32 * 32;
        chunks = (M + rpw - 1) / rpw;
        const dim3 grid((unsigned)strips, (unsigned)chunks);
        if (dtype == SDE_BF16)
            hipLaunchKernelGGL(bn_bwd_finalize_apply_kernel<bf16_t>, grid, dim3(256), 0, s, part, nblk, C, (float)M, dgamma, dbeta, accumulate_params, (const bf16_t*)dz,
                               (const bf16_t*)y, bnp, M, rpw, (bf16_t*)dy);
        else
            hipLaunchKernelGGL(bn_bwd_finalize_apply_kernel<half_t>, grid, dim3(256), 0, s, part, nblk, C, (float)M, dgamma, dbeta, accumulate_params, (const half_t*)dz,
                               (const half_t*)y, bnp, M, rpw, (half_t*)dy);
        SDE_CHECK_LAUNCH("sde_bn_bwd/finalize+apply");
        return SDE_OK;
    }
    int rows = nblk;
    const float* src = pre_reduce(part, rows, 2 * C, s);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(sde_cdiv(C, 8)), dim3(SLAB_T), 0, s, src, rows, C, (float)M, dgamma, dbeta, accumulate_params, coef);
    SDE_CHECK_LAUNCH("sde_bn_bwd/finalize");
    const int nb = grid_for(M * (C / V));
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)dz, (const float*)y, bnp, coef, M, C, (float*)dy),
               hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)dz, (const bf16_t*)y, bnp, coef, M, C, (bf16_t*)dy),
               hipLaunchKernelGGL(bn_bwd_apply_kernel<half_t>, dim3(nb), dim3(256), 0, s, (const half_t*)dz, (const half_t*)y, bnp, coef, M, C, (half_t*)dy));
    SDE_CHECK_LAUNCH("sde_bn_bwd/apply");
    return SDE_OK;
}

int sde_bn_bwd_from_part(const float* part, int rows, const void* gm, const void* y, const float* bnp, long M, int C, int dtype, float* coef,
                         float* dgamma, float* dbeta, int accumulate_params, void* dy, sde_stream_t stream) {
    const int V = SDE_IS16(dtype) ? 8 : 4;
    SDE_CHECK_ARG(part && gm && y && bnp && coef && dgamma && dbeta && dy && rows > 0 && M > 0 && C > 0 && C % V == 0, "sde_bn_bwd_from_part: bad argument");
    hipStream_t s = (hipStream_t)stream;
    if (g_bn_fuse && SDE_IS16(dtype) && C % 64 == 0 && rows <= BNFA_MAX_ROWS) {       // short slab: finalize + apply in one launch
        const int strips = C / 64;
        long chunks = (1024 + strips - 1) / strips;
        long rpw = (M + chunks - 1) / chunks;
        rpw = (rpw + 31) / 32 * 32;
        chunks = (M + rpw - 1) / rpw;
        const dim3 grid((unsigned)strips, (unsigned)chunks);
        if (dtype == SDE_BF16)
            hipLaunchKernelGGL(bn_bwd_finalize_apply_kernel<bf16_t>, grid, dim3(256), 0, s, part, rows, C, (float)M, dgamma, dbeta, accumulate_params, (const bf16_t*)gm,
                               (const bf16_t*)y, bnp, M, rpw, (bf16_t*)dy);
        else
            hipLaunchKernelGGL(bn_bwd_finalize_apply_kernel<half_t>, grid, dim3(256), 0, s, part, rows, C, (float)M, dgamma, dbeta, accumulate_params, (const half_t*)gm,
                               (const half_t*)y, bnp, M, rpw, (half_t*)dy);
        SDE_CHECK_LAUNCH("sde_bn_bwd_from_part/finalize+apply");
        return SDE_OK;
    }
    int r = rows;
    const float* src = pre_reduce(part, r, 2 * C, s);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(sde_cdiv(C, 8)), dim3(SLAB_T), 0, s, src, r, C, (float)M, dgamma, dbeta, accumulate_params, coef);
    SDE_CHECK_LAUNCH("sde_bn_bwd_from_part/finalize");
    const int nb = grid_for(M * (C / V));
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)gm, (const float*)y, bnp, coef, M, C, (float*)dy),
               hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)gm, (const bf16_t*)y, bnp, coef, M, C, (bf16_t*)dy),
               hipLaunchKernelGGL(bn_bwd_apply_kernel<half_t>, dim3(nb), dim3(256), 0, s, (const half_t*)gm, (const half_t*)y, bnp, coef, M, C, (half_t*)dy));
    SDE_CHECK_LAUNCH("sde_bn_bwd_from_part/apply");
    return SDE_OK;
}

int sde_maxpool_fwd(const void* x, int B, int H, int W, int C, int dtype, void* out, uint8_t* idx, sde_stream_t stream) {
    const int V = SDE_IS16(dtype) ? 8 : 4;
    SDE_CHECK_ARG(x && out && idx && B > 0 && H > 1 && W > 1 && C % V == 0, "sde_maxpool_fwd: bad argument");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    hipStream_t s = (hipStream_t)stream;
    const int nb = grid_for((long)B * OH * OW * (C / V));
    DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)x, B, H, W, C, OH, OW, (float*)out, idx),
               hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)x, B, H, W, C, OH, OW, (bf16_t*)out, idx),
               hipLaunchKernelGGL(maxpool_fwd_kernel<half_t>, dim3(nb), dim3(256), 0, s, (const half_t*)x, B, H, W, C, OH, OW, (half_t*)out, idx));
    SDE_CHECK_LAUNCH("sde_maxpool_fwd");
    return SDE_OK;
}

int sde_maxpool_bwd(const void* dout, const uint8_t* idx, int B, int H, int W, int C, int dtype, void* dx, sde_stream_t stream) {
    return sde_maxpool_bwd_sum(dout, nullptr, idx, B, H, W, C, dtype, dx, stream);
}

int sde_maxpool_bwd_sum(const void* dout, const void* dout1, const uint8_t* idx, int B, int H, int W, int C, int dtype, void* dx, sde_stream_t stream) {
    const int V = SDE_IS16(dtype) ? 8 : 4;
    SDE_CHECK_ARG(dout && idx && dx && B > 0 && H > 1 && W > 1 && C % V == 0, "sde_maxpool_bwd: bad argument");
    const int OH = (H + 2 - 3) / 2 + 1, OW = (W + 2 - 3) / 2 + 1;
    hipStream_t s = (hipStream_t)stream;
    const int nb = grid_for((long)B * H * W * (C / V));
    DISPATCH_T(dtype, hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)dout, (const float*)dout1, idx, B, H, W, C, OH, OW, (float*)dx),
               hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)dout, (const bf16_t*)dout1, idx, B, H, W, C, OH, OW, (bf16_t*)dx),
               hipLaunchKernelGGL(maxpool_bwd_kernel<half_t>, dim3(nb), dim3(256), 0, s, (const half_t*)dout, (const half_t*)dout1, idx, B, H, W, C, OH, OW, (half_t*)dx));
    SDE_CHECK_LAUNCH("sde_maxpool_bwd");
    return SDE_OK;
}

int sde_act_bwd_bias(const void* dout, const void* out, int act, long M, int C, int dtype, void* dz, float* part, float* dbias, int Cbias, int accumulate,
                     sde_stream_t stream) {
    return sde_act_bwd_bias_sum(dout, nullptr, out, act, M, C, dtype, dz, part, dbias, Cbias, accumulate, stream);
}

int sde_act_bwd_bias_sum(const void* dout, const void* dout1, const void* out, int act, long M, int C, int dtype, void* dz, float* part, float* dbias, int Cbias,
                         int accumulate, sde_stream_t stream) {
    const int V = SDE_IS16(dtype) ? 8 : 4;
    SDE_CHECK_ARG(dout && M > 0 && C > 0 && C % V == 0, "sde_act_bwd_bias: bad argument");
    SDE_CHECK_ARG(act == SDE_ACT_NONE || out, "sde_act_bwd_bias: activation backward needs the saved output");
    SDE_CHECK_ARG((dbias == nullptr) || (part && Cbias > 0 && Cbias <= C), "sde_act_bwd_bias: bias gradient needs a partial slab");
    hipStream_t s = (hipStream_t)stream;
    const int nblk = sde_reduce_num_blocks(M, C);
    const long rpb = (M + nblk - 1) / nblk;
    float* p = part;                    // (part without dbias: the column partials only -- the caller sums them later, sde_colsum_finalize_batched)
    const size_t lds = ((size_t)C > 256 * 8 ? (size_t)C : 256 * 8) * sizeof(float);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(act_bwd_bias_kernel<float>, dim3(nblk), dim3(256), lds, s, (const float*)dout, (const float*)dout1, (const float*)out, act, M, C, rpb, (float*)dz, p),
               hipLaunchKernelGGL(act_bwd_bias_kernel<bf16_t>, dim3(nblk), dim3(256), lds, s, (const bf16_t*)dout, (const bf16_t*)dout1, (const bf16_t*)out, act, M, C, rpb, (bf16_t*)dz, p),
               hipLaunchKernelGGL(act_bwd_bias_kernel<half_t>, dim3(nblk), dim3(256), lds, s, (const half_t*)dout, (const half_t*)dout1, (const half_t*)out, act, M, C, rpb, (half_t*)dz, p));
    SDE_CHECK_LAUNCH("sde_act_bwd_bias");
    if (dbias) {
        int rows = nblk;
        const float* src = pre_reduce(part, rows, C, s);
        hipLaunchKernelGGL(colsum_finalize_kernel, dim3(sde_cdiv(Cbias, 16)), dim3(SLAB_T), 0, s, src, rows, C, Cbias, dbias, accumulate);
        SDE_CHECK_LAUNCH("sde_act_bwd_bias/finalize");
    }
    return SDE_OK;
}

int sde_colsum_finalize_batched(const sde_colsum_item* items, int n, sde_stream_t stream) {
    SDE_CHECK_ARG(items && n >= 0, "sde_colsum_finalize_batched: bad argument");
    for (int base = 0; base < n; base += COLSUM_MAX_ITEMS) {
        ColsumBatch b;
        b.n = n - base < COLSUM_MAX_ITEMS ? n - base : COLSUM_MAX_ITEMS;
        b.first[0] = 0;
        for (int k = 0; k < COLSUM_MAX_ITEMS; ++k) {
            const bool on = k < b.n;
            const sde_colsum_item* it = on ? items + base + k : nullptr;
            SDE_CHECK_ARG(!on || (it->part && it->out && it->rows >= 1 && it->rows <= 4096 && it->C >= 1 && it->C <= it->ld),
                          "sde_colsum_finalize_batched: item %d: bad slab (rows=%d ld=%d C=%d)", base + k, on ? it->rows : 0, on ? it->ld : 0, on ? it->C : 0);
            b.part[k] = on ? it->part : nullptr; b.out[k] = on ? it->out : nullptr;
            b.rows[k] = on ? it->rows : 0; b.ld[k] = on ? it->ld : 0; b.C[k] = on ? it->C : 0; b.accumulate[k] = on ? it->accumulate : 0;
            b.first[k + 1] = b.first[k] + (on ? sde_cdiv(it->C, 16) : 0);
        }
        if (b.first[b.n] == 0) continue;
        hipLaunchKernelGGL(colsum_finalize_batched_kernel, dim3(b.first[b.n]), dim3(SLAB_T), 0, (hipStream_t)stream, b);
        SDE_CHECK_LAUNCH("sde_colsum_finalize_batched");
    }
    return SDE_OK;
}

int sde_refl_fold(const void* dxp, int B, int H, int W, int C, int C0, int upcat, int dtype, void* dx0, void* dx1, sde_stream_t stream) {
    const int V = SDE_IS16(dtype) ? 8 : 4;
    SDE_CHECK_ARG(dxp && dx0 && B > 0 && H >= 2 && W >= 2 && C % V == 0, "sde_refl_fold: bad argument (B=%d H=%d W=%d C=%d)", B, H, W, C);
    SDE_CHECK_ARG(!upcat || (C0 > 0 && C0 <= C && C0 % V == 0 && H % 2 == 0 && W % 2 == 0 && (C0 == C || dx1)), "sde_refl_fold: bad upcat argument");
    hipStream_t s = (hipStream_t)stream;
    const long items = upcat ? (long)B * (H / 2) * (W / 2) * (C0 / V) + (long)B * H * W * ((C - C0) / V) : (long)B * H * W * (C / V);
    const int nb = grid_for(items);
    DISPATCH_T(dtype, hipLaunchKernelGGL(refl_fold_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)dxp, B, H, W, C, C0, upcat, (float*)dx0, (float*)dx1),
               hipLaunchKernelGGL(refl_fold_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)dxp, B, H, W, C, C0, upcat, (bf16_t*)dx0, (bf16_t*)dx1),
               hipLaunchKernelGGL(refl_fold_kernel<half_t>, dim3(nb), dim3(256), 0, s, (const half_t*)dxp, B, H, W, C, C0, upcat, (half_t*)dx0, (half_t*)dx1));
    SDE_CHECK_LAUNCH("sde_refl_fold");
    return SDE_OK;
}

int sde_depth_head_fwd(const void* y, int B, int H, int W, int ld, float min_depth, float max_depth, int flip, int dtype, float* depth, sde_stream_t stream) {
    SDE_CHECK_ARG(y && depth && B > 0 && H > 0 && W > 0 && ld > 0 && min_depth > 0 && max_depth > min_depth, "sde_depth_head_fwd: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const float mind = 1.0f / max_depth, maxd = 1.0f / min_depth;
    const int nb = grid_for((long)B * H * W);
    DISPATCH_T(dtype, hipLaunchKernelGGL(depth_head_fwd_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)y, B, H, W, ld, mind, maxd, flip, depth),
               hipLaunchKernelGGL(depth_head_fwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)y, B, H, W, ld, mind, maxd, flip, depth),
               hipLaunchKernelGGL(depth_head_fwd_kernel<half_t>, dim3(nb), dim3(256), 0, s, (const half_t*)y, B, H, W, ld, mind, maxd, flip, depth));
    SDE_CHECK_LAUNCH("sde_depth_head_fwd");
    return SDE_OK;
}

int sde_depth_head_bwd(const void* y, const float* ddepth, int B, int H, int W, int ld, float min_depth, float max_depth, int flip, int dtype, void* dy,
                       sde_stream_t stream) {
    return sde_depth_head_bwd_bias(y, ddepth, B, H, W, ld, min_depth, max_depth, flip, dtype, dy, nullptr, nullptr, 0, stream);
}

int sde_depth_head_bias_blocks(int B, int H, int W) { return grid_for((long)B * H * W); }

int sde_depth_head_bwd_bias(const void* y, const float* ddepth, int B, int H, int W, int ld, float min_depth, float max_depth, int flip, int dtype, void* dy,
                            float* part, float* dbias, int accumulate, sde_stream_t stream) {
    SDE_CHECK_ARG(y && ddepth && dy && B > 0 && H > 0 && W > 0 && ld > 0, "sde_depth_head_bwd: bad argument");
    SDE_CHECK_ARG((part == nullptr) == (dbias == nullptr), "sde_depth_head_bwd_bias: the bias gradient needs both the partial buffer and the output");
    hipStream_t s = (hipStream_t)stream;
    const float mind = 1.0f / max_depth, maxd = 1.0f / min_depth;
    const int nb = grid_for((long)B * H * W);
    DISPATCH_T(dtype, hipLaunchKernelGGL(depth_head_bwd_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)y, ddepth, B, H, W, ld, mind, maxd, flip, (float*)dy, part),
               hipLaunchKernelGGL(depth_head_bwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)y, ddepth, B, H, W, ld, mind, maxd, flip, (bf16_t*)dy, part),
               hipLaunchKernelGGL(depth_head_bwd_kernel<half_t>, dim3(nb), dim3(256), 0, s, (const half_t*)y, ddepth, B, H, W, ld, mind, maxd, flip, (half_t*)dy, part));
    SDE_CHECK_LAUNCH("sde_depth_head_bwd");
    if (dbias) {
        hipLaunchKernelGGL(head_bias_finalize_kernel, dim3(1), dim3(256), 0, s, part, nb, dbias, accumulate);
        SDE_CHECK_LAUNCH("sde_depth_head_bwd_bias/finalize");
    }
    return SDE_OK;
}

int sde_gn_relu_fwd(const void* x, const float* gamma, const float* beta, int B, int HW, int C, int G, float eps, int relu, int dtype, float* part, float* gnp,
                    void* out, sde_stream_t stream) {
    return sde_gn_relu_res_fwd(x, nullptr, gamma, beta, B, HW, C, G, eps, relu, dtype, part, gnp, out, stream);
}

static bool gn_vector_path(int C, int dtype) { const int V = SDE_IS16(dtype) ? 8 : 4; return C % V == 0 && 256 % (C / V) == 0; }

int sde_gn_relu_res_fwd(const void* x, const void* res, const float* gamma, const float* beta, int B, int HW, int C, int G, float eps, int relu, int dtype,
                        float* part, float* gnp, void* out, sde_stream_t stream) {
    SDE_CHECK_ARG(x && gamma && beta && part && gnp && out && B > 0 && HW > 0 && C > 0 && G > 0 && C % G == 0, "sde_gn_relu_fwd: bad argument");
    SDE_CHECK_ARG(C % (SDE_IS16(dtype) ? 8 : 4) == 0, "sde_gn_relu_fwd: C=%d must be a multiple of the 16-byte group", C);
    SDE_CHECK_ARG(!res || gn_vector_path(C, dtype), "sde_gn_relu_res_fwd: the residual form needs C / (16-byte group) to divide 256 (C=%d)", C);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = 2 * (size_t)C * sizeof(float);
    DISPATCH_T(dtype, hipLaunchKernelGGL(gn_stats_kernel<float>, dim3(GN_CHUNKS, B), dim3(256), lds, s, (const float*)x, (const float*)res, HW, C, part),
               hipLaunchKernelGGL(gn_stats_kernel<bf16_t>, dim3(GN_CHUNKS, B), dim3(256), lds, s, (const bf16_t*)x, (const bf16_t*)res, HW, C, part),
               hipLaunchKernelGGL(gn_stats_kernel<half_t>, dim3(GN_CHUNKS, B), dim3(256), lds, s, (const half_t*)x, (const half_t*)res, HW, C, part));
    SDE_CHECK_LAUNCH("sde_gn_relu_fwd/stats");
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * G), dim3(64), 0, s, part, B, C, G, HW, eps, gnp);
    SDE_CHECK_LAUNCH("sde_gn_relu_fwd/finalize");
    const int nb = grid_for((long)B * HW * (C / (SDE_IS16(dtype) ? 8 : 4)));
    DISPATCH_T(dtype, hipLaunchKernelGGL(gn_apply_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)x, (const float*)res, gnp, gamma, beta, B, HW, C, G, relu, (float*)out),
               hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)res, gnp, gamma, beta, B, HW, C, G, relu, (bf16_t*)out),
               hipLaunchKernelGGL(gn_apply_kernel<half_t>, dim3(nb), dim3(256), 0, s, (const half_t*)x, (const half_t*)res, gnp, gamma, beta, B, HW, C, G, relu, (half_t*)out));
    SDE_CHECK_LAUNCH("sde_gn_relu_fwd/apply");
    return SDE_OK;
}

int sde_gn_relu_bwd(const void* dout, const void* out, const void* x, const float* gnp, const float* gamma, int B, int HW, int C, int G, int relu, int dtype,
                    float* part, float* coef, float* dgamma, float* dbeta, int accumulate_params, void* dx, sde_stream_t stream) {
    return sde_gn_relu_res_bwd(dout, out, x, nullptr, gnp, gamma, B, HW, C, G, relu, dtype, part, coef, dgamma, dbeta, accumulate_params, dx, stream);
}

int sde_gn_relu_res_bwd(const void* dout, const void* out, const void* x, const void* res, const float* gnp, const float* gamma, int B, int HW, int C, int G, int relu,
                        int dtype, float* part, float* coef, float* dgamma, float* dbeta, int accumulate_params, void* dx, sde_stream_t stream) {
    SDE_CHECK_ARG(dout && out && x && gnp && gamma && part && coef && dgamma && dbeta && dx && C % G == 0, "sde_gn_relu_bwd: bad argument");
    SDE_CHECK_ARG(!res || gn_vector_path(C, dtype), "sde_gn_relu_res_bwd: the residual form needs C / (16-byte group) to divide 256 (C=%d)", C);
    SDE_CHECK_ARG(C % (SDE_IS16(dtype) ? 8 : 4) == 0, "sde_gn_relu_bwd: C=%d must be a multiple of the 16-byte group", C);
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = 2 * (size_t)C * sizeof(float);
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(gn_bwd_stats_kernel<float>, dim3(GN_CHUNKS, B), dim3(256), lds, s, (const float*)dout, (const float*)out, (const float*)x, (const float*)res, gnp, HW, C, G, relu, part),
               hipLaunchKernelGGL(gn_bwd_stats_kernel<bf16_t>, dim3(GN_CHUNKS, B), dim3(256), lds, s, (const bf16_t*)dout, (const bf16_t*)out, (const bf16_t*)x, (const bf16_t*)res, gnp, HW, C, G, relu, part),
               hipLaunchKernelGGL(gn_bwd_stats_kernel<half_t>, dim3(GN_CHUNKS, B), dim3(256), lds, s, (const half_t*)dout, (const half_t*)out, (const half_t*)x, (const half_t*)res, gnp, HW, C, G, relu, part));
    SDE_CHECK_LAUNCH("sde_gn_relu_bwd/stats");
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(B * G + C), dim3(64), 0, s, part, gamma, B, C, G, HW, coef, dgamma, dbeta, accumulate_params);
    SDE_CHECK_LAUNCH("sde_gn_relu_bwd/finalize");
    const int nb = grid_for((long)B * HW * (C / (SDE_IS16(dtype) ? 8 : 4)));
    DISPATCH_T(dtype,
               hipLaunchKernelGGL(gn_bwd_apply_kernel<float>, dim3(nb), dim3(256), 0, s, (const float*)dout, (const float*)out, (const float*)x, (const float*)res, gnp, coef, gamma, B, HW, C, G, relu, (float*)dx),
               hipLaunchKernelGGL(gn_bwd_apply_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, (const bf16_t*)dout, (const bf16_t*)out, (const bf16_t*)x, (const bf16_t*)res, gnp, coef, gamma, B, HW, C, G, relu, (bf16_t*)dx),
               hipLaunchKernelGGL(gn_bwd_apply_kernel<half_t>, dim3(nb), dim3(256), 0, s, (const half_t*)dout, (const half_t*)out, (const half_t*)x, (const half_t*)res, gnp, coef, gamma, B, HW, C, G, relu, (half_t*)dx));
    SDE_CHECK_LAUNCH("sde_gn_relu_bwd/apply");
    return SDE_OK;
}

int sde_adam_step(float* p, const float* g, float* m, float* v, long n, const sde_adam_desc* d, sde_stream_t stream) {
    SDE_CHECK_ARG(p && g && m && v && d && n > 0, "sde_adam_step: null pointer");
    SDE_CHECK_ARG(d->nseg > 0 && d->nseg <= SDE_ADAM_MAX_SEG && (d->scale_state ? (d->beta1_d > 0.0 && d->beta1_d < 1.0 && d->beta2_d > 0.0 && d->beta2_d < 1.0)
                                                                                  : (d->bias_corr1 > 0.f && d->bias_corr2 > 0.f)),
                  "sde_adam_step: bad descriptor (nseg %d)", d->nseg);
    for (int i = 0; i < d->nseg; ++i)
        SDE_CHECK_ARG(d->seg_end[i] > (i ? d->seg_end[i - 1] : 0) && d->seg_end[i] <= n, "sde_adam_step: segment %d ends at %ld (n = %ld)", i, d->seg_end[i], n);
    SDE_CHECK_ARG(d->seg_end[d->nseg - 1] == n, "sde_adam_step: the segments must cover the buffer");
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, *d);
    SDE_CHECK_LAUNCH("sde_adam_step");
    return SDE_OK;
}

int sde_grad_check(const float* g, long n, float* scale_state, sde_stream_t stream) {
    SDE_CHECK_ARG(g && scale_state && n > 0 && ((uintptr_t)g & 15) == 0, "sde_grad_check: bad argument");
    const long n4 = n / 4;
    hipLaunchKernelGGL(grad_check_kernel, dim3(grid_for(n4 > 0 ? n4 : 1)), dim3(256), 0, (hipStream_t)stream, (const float4*)g, n4, g + n4 * 4, (int)(n - n4 * 4), scale_state);
    SDE_CHECK_LAUNCH("sde_grad_check");
    return SDE_OK;
}

int sde_loss_scale_update(float* scale_state, float growth_factor, float backoff_factor, int growth_interval, sde_stream_t stream) {
    SDE_CHECK_ARG(scale_state && growth_factor >= 1.f && backoff_factor > 0.f && backoff_factor <= 1.f && growth_interval > 0, "sde_loss_scale_update: bad argument");
    hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, scale_state, growth_factor, backoff_factor, growth_interval);
    SDE_CHECK_LAUNCH("sde_loss_scale_update");
    return SDE_OK;
}

}  // extern "C"
