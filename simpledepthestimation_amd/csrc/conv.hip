// Implicit-GEMM convolution engine for gfx950 (MI355X): forward, data-gradient and weight-gradient of every
// convolution on the DepthNet / PoseNet path, in fp32 (parity mode) or bf16 (throughput mode), fp32 accumulate.
//
// Replaces (reference, read-only): the nn.Conv2d / ReflectionPad2d / F.interpolate(nearest) / torch.cat calls of
// detectron2/layers/resnet_encoder.py:L88-99 (torchvision ResNet convs), detectron2/layers/depth_decoder.py:L21-53,
// L95-110 (Conv3x3, ConvBlock, upsample + skip concat) and detectron2/modeling/pose_net/PoseNet.py:L13-20 -- and their
// autograd backward.
//
// Design (MI355X-first, not a cuDNN-shaped port):
//   * activations NHWC, channel count padded to 16 bytes, so every im2col element group is one 16-byte load that is
//     contiguous in HBM; weights are pre-packed [Cout][KH][KW][Cin] (K contiguous), i.e. both MFMA operands are K-major.
//   * the A operand is gathered on the fly (no im2col buffer): zero / reflection padding, stride, nearest x2 upsample +
//     channel concat of a skip tensor (decoder) and zero-insertion (data-gradient of stride-2 convs) are all address
//     arithmetic in the loader -- the padded / upsampled / concatenated tensors never exist in HBM.
//   * 256-thread workgroups (4 wave64), 16x16 MFMA fragments (v_mfma_f32_16x16x32_bf16 / v_mfma_f32_16x16x4_f32),
//     128-byte K slices staged through XOR-swizzled LDS (conflict-free ds_read_b128), double-buffered with the next
//     slice's global loads in flight over the MFMAs.
//   * epilogue through LDS: bias + ELU in fp32, 16-byte coalesced NHWC stores, and per-tile per-channel (sum, sum^2)
//     partials for training-mode BatchNorm (deterministic: no atomics).
//   * weight gradient: reduction over pixels with both operands read transposed out of LDS (ds_read_b64_tr_b16), split
//     over pixel ranges into fp32 slabs that a second kernel sums in fixed order (bit-reproducible gradients).
#include <type_traits>

#include "common.h"
#include "sde_hip.h"
#include "conv_common.h"

namespace {
using namespace sdeconv;

template <int I> using IC = std::integral_constant<int, I>;

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;

template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int V = 4; };
template <> struct VecOf<bf16_t> { static constexpr int V = 8; };
template <> struct VecOf<half_t> { static constexpr int V = 8; };

template <typename T>
__device__ __forceinline__ uint4 gather16(const Gather& g, int n, int ih, int iw, int ci) {
    uint4 z = {0u, 0u, 0u, 0u};
    if (g.reflect) { ih = reflect1(ih, g.IH); iw = reflect1(iw, g.IW); }
    else if (ih < 0 || ih >= g.IH || iw < 0 || iw >= g.IW) return z;
    const T* src;
    if (g.mode == SDE_SRC_PLAIN) {
        src = (const T*)g.x0 + ((size_t)(n * g.H0 + ih) * g.W0 + iw) * g.C0 + ci;
    } else if (g.mode == SDE_SRC_UPCAT) {
        if (ci < g.C0) src = (const T*)g.x0 + ((size_t)(n * g.H0 + (ih >> 1)) * g.W0 + (iw >> 1)) * g.C0 + ci;
        else src = (const T*)g.x1 + ((size_t)(n * g.IH + ih) * g.IW + iw) * g.C1 + (ci - g.C0);
    } else {
        if ((ih | iw) & 1) return z;
        src = (const T*)g.x0 + ((size_t)(n * g.H0 + (ih >> 1)) * g.W0 + (iw >> 1)) * g.C0 + ci;
    }
    return *reinterpret_cast<const uint4*>(src);
}

// ------------------------------------------------------------------------------------------------------------------
// MFMA wrappers: one 64-byte K sub-block (32 bf16 / 16 f32) of a 16x16 fragment pair
// ------------------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ f32x4 mma64(const uint4& a, const uint4& b, f32x4 c);
template <> __device__ __forceinline__ f32x4 mma64<bf16_t>(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mma64<half_t>(const uint4& a, const uint4& b, f32x4 c) {
    typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <> __device__ __forceinline__ f32x4 mma64<float>(const uint4& a, const uint4& b, f32x4 c) {
    // lane group g = lane>>4 holds k = 4g+j in element j: MFMA j contracts the 4 k's {4g+j}; A and B use the same map.
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), c, 0, 0, 0);
    return c;
}

template <typename T> __device__ __forceinline__ float to_f32(T v);
template <> __device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f32<bf16_t>(bf16_t v) { return (float)v; }
template <> __device__ __forceinline__ float to_f32<half_t>(half_t v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f32<bf16_t>(float v) { return (bf16_t)v; }
template <> __device__ __forceinline__ half_t from_f32<half_t>(float v) { return (half_t)v; }

// ------------------------------------------------------------------------------------------------------------------
// Forward / data-gradient implicit GEMM:  Y[M, Cout] = im2col(X)[M, K] * Wp[Cout, K]^T
// ------------------------------------------------------------------------------------------------------------------
constexpr int KSTAGE_BYTES = 128;   // K bytes per row per pipeline stage (2 MFMA sub-blocks of 64 B)
constexpr int NTHREADS = 256;

// Address of the 16-byte group (n, ih, iw, ci) of the virtual input, or nullptr when it is padding / an inserted zero.
template <typename T>
__device__ __forceinline__ const T* gather_ptr(const Gather& g, int n, int ih, int iw, int ci) {
    if (g.reflect) { ih = reflect1(ih, g.IH); iw = reflect1(iw, g.IW); }
    else if ((unsigned)ih >= (unsigned)g.IH || (unsigned)iw >= (unsigned)g.IW) return nullptr;
    if (g.mode == SDE_SRC_PLAIN) return (const T*)g.x0 + ((size_t)((n * g.H0 + ih) * g.W0 + iw)) * g.C0 + ci;
    if (g.mode == SDE_SRC_UPCAT) {
        if (ci < g.C0) return (const T*)g.x0 + ((size_t)((n * g.H0 + (ih >> 1)) * g.W0 + (iw >> 1))) * g.C0 + ci;
        return (const T*)g.x1 + ((size_t)((n * g.IH + ih) * g.IW + iw)) * g.C1 + (ci - g.C0);
    }
    if ((ih | iw) & 1) return nullptr;
    return (const T*)g.x0 + ((size_t)((n * g.H0 + (ih >> 1)) * g.W0 + (iw >> 1))) * g.C0 + ci;
}


template <int SRC> __device__ __forceinline__ int src_mode(const Gather& g) {
    if (SRC == SRC_RUNTIME) return g.mode;
    return SRC == SRC_UPCAT_REFLECT ? SDE_SRC_UPCAT : (SRC == SRC_ZEROINS_ZERO ? SDE_SRC_ZEROINS : SDE_SRC_PLAIN);
}
template <int SRC> __device__ __forceinline__ bool src_reflect(const Gather& g) {
    if (SRC == SRC_RUNTIME) return g.reflect != 0;
    return SRC == SRC_PLAIN_REFLECT || SRC == SRC_UPCAT_REFLECT;
}

// offsets (bytes) of the 16-byte group (n, ih, iw, ci) in source 0 / source 1; kOOB where it is padding / not that source
template <typename T, int SRC>
__device__ __forceinline__ void gather_off(const Gather& g, bool ok, int n, int ih, int iw, int ci, unsigned& o0, unsigned& o1) {
    const int mode = src_mode<SRC>(g);
    if (src_reflect<SRC>(g)) { ih = reflect1(ih, g.IH); iw = reflect1(iw, g.IW); }
    else ok = ok & ((unsigned)ih < (unsigned)g.IH) & ((unsigned)iw < (unsigned)g.IW);
    o1 = kOOB;
    if (mode == SDE_SRC_PLAIN) {
        const unsigned o = (unsigned)(((n * g.H0 + ih) * g.W0 + iw) * g.C0 + ci) * (unsigned)sizeof(T);
        o0 = ok ? o : kOOB;
    } else if (mode == SDE_SRC_UPCAT) {
        const bool first = ci < g.C0;
        const unsigned oa = (unsigned)(((n * g.H0 + (ih >> 1)) * g.W0 + (iw >> 1)) * g.C0 + ci) * (unsigned)sizeof(T);
        const unsigned ob = (unsigned)(((n * g.IH + ih) * g.IW + iw) * g.C1 + (ci - g.C0)) * (unsigned)sizeof(T);
        o0 = (ok & first) ? oa : kOOB;
        o1 = (ok & !first) ? ob : kOOB;
    } else {
        ok = ok & !((ih | iw) & 1);
        const unsigned o = (unsigned)(((n * g.H0 + (ih >> 1)) * g.W0 + (iw >> 1)) * g.C0 + ci) * (unsigned)sizeof(T);
        o0 = ok ? o : kOOB;
    }
}

template <typename T, int BM, int BN, int WM, int WN, int SRC, bool ONE_TAP>
__global__ void __launch_bounds__(NTHREADS, 2) igemm_kernel(IGemmP p) {
    constexpr int V = VecOf<T>::V;
    constexpr int BK = KSTAGE_BYTES / (int)sizeof(T);
    constexpr int WTM = BM / WM, WTN = BN / WN;     // wave tile
    constexpr int FM = WTM / 16, FN = WTN / 16;
    constexpr int TPR = NTHREADS / BM;              // threads per A row (2 or 4): each owns CPT consecutive 16-byte chunks
    constexpr int CPT = 8 / TPR;
    constexpr int B_ROWS_PER_PASS = NTHREADS / 8;   // B tile: 8 chunks of 16 B per row, 32 rows per pass
    constexpr int B_PASSES = (BN + B_ROWS_PER_PASS - 1) / B_ROWS_PER_PASS;
    constexpr int CPAD = 4;
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                                // [2][BM][128 B]
    unsigned char* sB = smem + 2 * BM * KSTAGE_BYTES;        // [2][BN][128 B]

    const Gather& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (p.ldy + BN - 1) / BN;
    const int logical_all = xcd_remap(blockIdx.x, gridDim.x);
    const int tiles_all = gridDim.x / p.ksplit;
    const int split = logical_all / tiles_all, logical = logical_all - split * tiles_all;     // the splits of a tile land on different XCDs: no sharing assumed
    const int tile_m = logical / tiles_n, tile_n = logical - tile_m * tiles_n;     // N tiles of one M tile are adjacent in time, on one XCD
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- A side: this thread gathers chunks [ac0, ac0+CPT) of tile row `arow` in every stage.
    // Output-pixel decomposition happens once; the (tap, channel) position of the first chunk is advanced incrementally.
    const int arow = tid / TPR, ac0 = (tid % TPR) * CPT;
    int an = -1, aih = 0, aiw = 0;
    {
        const int m = m0 + arow;
        if (m < g.M) {
            an = m / (g.OH * g.OW);
            const int rem = m - an * (g.OH * g.OW);
            const int oh = rem / g.OW, ow = rem - oh * g.OW;
            aih = oh * g.stride - g.pad; aiw = ow * g.stride - g.pad;
        }
    }
    const int nk_total = (g.Ktot + BK - 1) / BK;
    const int nk_per = (nk_total + p.ksplit - 1) / p.ksplit;
    const int s_begin = split * nk_per;                                 // this workgroup's K stages: [s_begin, s_begin + nk)
    int aci, akh, akw;     // position of chunk ac0 of the NEXT stage to load
    {
        const int k = s_begin * BK + ac0 * V;
        const int tap = k / g.Cin;
        aci = k - tap * g.Cin; akh = tap / g.KW; akw = tap - akh * g.KW;
    }
    // ---- B side (weights, plain 2-D): rows r0 + 32 i, chunk column cc
    const int cc = tid & 7, r0 = tid >> 3;
    const int nk = max(0, min(nk_total, s_begin + nk_per) - s_begin);
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(g.x0, (long)g.Bn * g.H0 * g.W0 * g.C0 * (long)sizeof(T));
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(g.x1 ? g.x1 : g.x0, g.x1 ? (long)g.Bn * g.IH * g.IW * g.C1 * (long)sizeof(T) : 0);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(p.w, (long)p.Cout * g.Ktot * (long)sizeof(T));
    // ONE_TAP: a thread's CPT chunks are 16-byte groups of one filter tap (channel counts multiples of CPT*V: every layer but the
    // image / 16-channel ones), so one offset per stage serves all of them.  SRC_1X1: no taps at all, offset = row base + k.
    const int mode = src_mode<SRC>(g);
    unsigned rowoff = kOOB;        // SRC_1X1: byte offset of this thread's input pixel
    if (SRC == SRC_1X1 && an >= 0) rowoff = (unsigned)(((an * g.H0 + aih) * g.W0 + aiw) * g.C0) * (unsigned)sizeof(T);

    // SRC_1X1 with K a whole number of stages (every real 1x1 layer: Cin % 64 == 0): the lane's byte offsets never change -- the K
    // position of a stage is a scalar offset of the buffer load, so a stage's loads cost no VALU instruction at all.
    const bool kfull = SRC == SRC_1X1 && (g.Ktot % BK) == 0 && !p.no_kfull;
    const bool kfullB = (g.Ktot % BK) == 0 && !p.no_kfull;       // the same for the weight rows of every source kind
    unsigned voffA = kOOB, voffB[B_PASSES];
    if (SRC == SRC_1X1 && an >= 0) voffA = rowoff + (unsigned)(ac0 * V) * (unsigned)sizeof(T);
#pragma unroll
    for (int i = 0; i < B_PASSES; ++i) {
        const int row = r0 + i * B_ROWS_PER_PASS, n = n0 + row;
        voffB[i] = (row < BN && n < p.Cout) ? (unsigned)(n * g.Ktot + cc * V) * (unsigned)sizeof(T) : kOOB;
    }

    // Three register stages in flight over two LDS buffers: the loop is bound by load latency, not by MFMA issue, so stage s+3
    // is requested while stage s is multiplied and stage s+1 is written to LDS (the compiler's counted vmcnt keeps s+2, s+3 flying).
    uint4 ra[3][CPT], rb[3][B_PASSES];

    auto load_stage = [&](int s, auto SET) {
        constexpr int st = decltype(SET)::value;
        if (SRC == SRC_1X1 && kfull) {
            const unsigned soff = (unsigned)(s_begin + s) * (unsigned)KSTAGE_BYTES;
#pragma unroll
            for (int c = 0; c < CPT; ++c) ra[st][c] = buf_load16s(rs0, voffA + 16u * c, soff);      // kOOB + 16c stays out of range
#pragma unroll
            for (int i = 0; i < B_PASSES; ++i) rb[st][i] = buf_load16s(rsw, voffB[i], soff);
            return;
        }
        if (SRC == SRC_1X1) {
            const int k = (s_begin + s) * BK + ac0 * V;
#pragma unroll
            for (int c = 0; c < CPT; ++c)
                ra[st][c] = buf_load16(rs0, (k + c * V < g.Ktot) ? rowoff + (unsigned)(k + c * V) * (unsigned)sizeof(T) : kOOB);   // rowoff is kOOB for rows >= M
        } else if (ONE_TAP) {
            unsigned o0, o1;
            gather_off<T, SRC>(g, an >= 0 && akh < g.KH, an, aih + akh, aiw + akw, aci, o0, o1);
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                uint4 v = buf_load16(rs0, o0 + 16u * c);       // kOOB + 16c stays out of range
                if (mode == SDE_SRC_UPCAT) {
                    const uint4 u = buf_load16(rs1, o1 + 16u * c);
                    v.x |= u.x; v.y |= u.y; v.z |= u.z; v.w |= u.w;
                }
                ra[st][c] = v;
            }
        } else {
            int ci = aci, kh = akh, kw = akw;
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                unsigned o0, o1;
                gather_off<T, SRC>(g, an >= 0 && kh < g.KH, an, aih + kh, aiw + kw, ci, o0, o1);
                uint4 v = buf_load16(rs0, o0);
                if (mode == SDE_SRC_UPCAT) {
                    const uint4 u = buf_load16(rs1, o1);
                    v.x |= u.x; v.y |= u.y; v.z |= u.z; v.w |= u.w;
                }
                ra[st][c] = v;
                ci += V;
                if (ci == g.Cin) { ci = 0; if (++kw == g.KW) { kw = 0; ++kh; } }
            }
        }
        if (SRC != SRC_1X1) {       // advance the thread's base (tap, channel) position by one stage (BK elements)
            aci += BK;
            while (aci >= g.Cin) { aci -= g.Cin; if (++akw == g.KW) { akw = 0; ++akh; } }
        }
        if (kfullB) {
            const unsigned soff = (unsigned)(s_begin + s) * (unsigned)KSTAGE_BYTES;
#pragma unroll
            for (int i = 0; i < B_PASSES; ++i) rb[st][i] = buf_load16s(rsw, voffB[i], soff);
            return;
        }
        const int k = (s_begin + s) * BK + cc * V;
        const bool kok = k < g.Ktot;
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int row = r0 + i * B_ROWS_PER_PASS;
            const int n = n0 + row;
            constexpr bool rows_fit = B_ROWS_PER_PASS * B_PASSES <= BN;        // every pass row is a tile row (BN >= 32): no per-row test
            const unsigned ow = (kok && (rows_fit || row < BN) && n < p.Cout) ? (unsigned)(n * g.Ktot + k) * (unsigned)sizeof(T) : kOOB;
            rb[st][i] = buf_load16(rsw, ow);
        }
    };
    auto store_stage = [&](int buf, auto SET) {
        constexpr int st = decltype(SET)::value;
#pragma unroll
        for (int c = 0; c < CPT; ++c)
            *reinterpret_cast<uint4*>(sA + (size_t)(buf * BM + arow) * KSTAGE_BYTES + (((ac0 + c) ^ (arow & 7)) << 4)) = ra[st][c];
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int row = r0 + i * B_ROWS_PER_PASS;
            if (B_ROWS_PER_PASS * B_PASSES <= BN || row < BN)
                *reinterpret_cast<uint4*>(sB + (size_t)(buf * BN + row) * KSTAGE_BYTES + ((cc ^ (row & 7)) << 4)) = rb[st][i];
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    auto compute_stage = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 a[FM], b[FN];
            const int ch = kk * 4 + fg;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int row = wm * WTM + i * 16 + fr;
                a[i] = *reinterpret_cast<const uint4*>(sA + (size_t)(buf * BM + row) * KSTAGE_BYTES + ((ch ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int row = wn * WTN + j * 16 + fr;
                b[j] = *reinterpret_cast<const uint4*>(sB + (size_t)(buf * BN + row) * KSTAGE_BYTES + ((ch ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = mma64<T>(a[i], b[j], acc[i][j]);
        }
    };
    // one pipeline step: request stage st+3 into the register set that stage st just vacated, multiply stage st out of LDS,
    // then publish stage st+1 (requested two steps ago) into the other LDS buffer.
    auto step = [&](int st, auto CUR, auto NXT) {
        if (st >= nk) return;
        if (st + 3 < nk) load_stage(st + 3, CUR);
        compute_stage(st & 1);
        if (st + 1 < nk) store_stage((st + 1) & 1, NXT);
        __syncthreads();
    };
    if (nk > 0) load_stage(0, IC<0>{});
    if (nk > 1) load_stage(1, IC<1>{});
    if (nk > 2) load_stage(2, IC<2>{});
    if (nk > 0) store_stage(0, IC<0>{});
    __syncthreads();
    for (int s = 0; s < nk; s += 3) {
        step(s, IC<0>{}, IC<1>{});
        step(s + 1, IC<1>{}, IC<2>{});
        step(s + 2, IC<2>{}, IC<0>{});
    }

    // ---- epilogue: accumulators -> LDS (fp32, padded rows) -> bias/act -> coalesced NHWC stores (+ BN partials) ----
    float* sC = reinterpret_cast<float*>(smem);   // [BM][BN + CPAD]
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * WTM + i * 16 + fg * 4 + r, col = wn * WTN + j * 16 + fr;
                sC[row * (BN + CPAD) + col] = acc[i][j][r];
            }
    __syncthreads();
    // float4 groups: consecutive lanes read consecutive 16 B of the staging tile (conflict-free ds_read_b128) and write
    // consecutive 4-element groups of the NHWC row (a wave instruction covers whole 128-byte lines)
    constexpr int G4 = BN / 4;
    const bool vec_ok = (p.ldy % 4) == 0;
    if (p.ksplit > 1) {       // raw fp32 partial tile; bias / activation / statistics happen in splitk_finish_kernel
        float* wsp = p.ws + (size_t)split * g.M * p.ldy;
        for (int id = tid; id < BM * G4; id += NTHREADS) {
            const int row = id / G4, c0 = (id - row * G4) * 4;
            const int m = m0 + row, n = n0 + c0;
            if (m >= g.M || n >= p.ldy) continue;
            *reinterpret_cast<float4*>(wsp + (size_t)m * p.ldy + n) = *reinterpret_cast<const float4*>(&sC[row * (BN + CPAD) + c0]);     // ldy % 4 == 0
        }
        return;
    }
    for (int id = tid; id < BM * G4; id += NTHREADS) {
        const int row = id / G4, c0 = (id - row * G4) * 4;
        const int m = m0 + row, n = n0 + c0;
        if (m >= g.M || n >= p.ldy) continue;
        const float4 q = *reinterpret_cast<const float4*>(&sC[row * (BN + CPAD) + c0]);
        float v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = v[e];
            if (p.bias && n + e < p.Cout) t += p.bias[n + e];
            if (p.act == SDE_ACT_ELU) t = t > 0.f ? t : expm1f(t);
            if (n + e >= p.Cout) t = 0.f;     // padded output channels are exact zeros
            v[e] = t;
        }
        T* dst = (T*)p.y + (size_t)m * p.ldy + n;
        T o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(v[e]);
        if (vec_ok && n + 4 <= p.ldy) {
            if (sizeof(T) == 2) *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<uint2*>(o);
            else *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<uint4*>(o);
        } else {
            for (int e = 0; e < 4 && n + e < p.ldy; ++e) dst[e] = o[e];
        }
        if (p.stats)     // keep the rounded value for the statistics pass
            *reinterpret_cast<float4*>(&sC[row * (BN + CPAD) + c0]) = make_float4(to_f32<T>(o[0]), to_f32<T>(o[1]), to_f32<T>(o[2]), to_f32<T>(o[3]));
    }
    if (p.stats) {
        __syncthreads();
        // per-tile column sums of y and y^2 over the valid rows (what BatchNorm's batch statistics need); the 256 threads split
        // each column's rows into NTHREADS/BN interleaved parts that are combined through the (now free) staging buffers' tail
        constexpr int PARTS = NTHREADS / BN >= 1 ? NTHREADS / BN : 1;
        float* red = sC + BM * (BN + CPAD);                    // [PARTS][BN][2] behind the tile (allocated by launch_igemm)
        const int c = tid % BN, part = tid / BN;
        const int rows = min(BM, g.M - m0);
        float s1 = 0.f, s2 = 0.f;
        if (part < PARTS)
            for (int r = part; r < rows; r += PARTS) { const float t = sC[r * (BN + CPAD) + c]; s1 += t; s2 += t * t; }
        if (part < PARTS) { red[(part * BN + c) * 2] = s1; red[(part * BN + c) * 2 + 1] = s2; }
        __syncthreads();
        if (tid < BN && n0 + tid < p.Cout) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int q = 0; q < PARTS; ++q) { a += red[(q * BN + tid) * 2]; b += red[(q * BN + tid) * 2 + 1]; }
            p.stats[((size_t)tile_m * p.Cout + n0 + tid) * 2 + 0] = a;
            p.stats[((size_t)tile_m * p.Cout + n0 + tid) * 2 + 1] = b;
        }
    }
}

// Split-K finish: y[m][n] = act(sum_s ws[s][m][n] + bias[n]) in a fixed order, padded channels zero, plus the per-64-row-tile column
// sums BatchNorm needs (same slab layout as the fused epilogue of the 64x64 tile).  One workgroup = 64 rows x 64 columns.
template <typename T>
__global__ void __launch_bounds__(256) splitk_finish_kernel(const float* __restrict__ ws, int S, int M, int ldy, int Cout, const float* __restrict__ bias,
                                                            int act, T* __restrict__ y, float* __restrict__ stats) {
    __shared__ float red[16][64][2];
    const int tiles_n = (ldy + 63) / 64;
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int c0 = tile_n * 64 + (threadIdx.x & 15) * 4, rg = threadIdx.x >> 4;     // 16 column groups of 4, 16 row groups
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (c0 < ldy) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = tile_m * 64 + rg + 16 * i;
            if (m >= M) continue;
            float4 a = *reinterpret_cast<const float4*>(ws + (size_t)m * ldy + c0);
            for (int sp = 1; sp < S; ++sp) {
                const float4 b = *reinterpret_cast<const float4*>(ws + ((size_t)sp * M + m) * ldy + c0);
                a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
            }
            float v[4] = {a.x, a.y, a.z, a.w};
            T o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float t = v[e];
                if (bias && c0 + e < Cout) t += bias[c0 + e];
                if (act == SDE_ACT_ELU) t = t > 0.f ? t : expm1f(t);
                if (c0 + e >= Cout) t = 0.f;
                o[e] = from_f32<T>(t);
                const float r = to_f32<T>(o[e]);         // statistics of the stored (rounded) values, as in the fused epilogue
                s1[e] += r; s2[e] += r * r;
            }
            T* dst = y + (size_t)m * ldy + c0;
            if (sizeof(T) == 2) *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<uint2*>(o);
            else *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<uint4*>(o);
        }
    }
    if (!stats) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[rg][(threadIdx.x & 15) * 4 + e][0] = s1[e]; red[rg][(threadIdx.x & 15) * 4 + e][1] = s2[e]; }
    __syncthreads();
    if (threadIdx.x < 64 && tile_n * 64 + threadIdx.x < Cout) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) { a += red[q][threadIdx.x][0]; b += red[q][threadIdx.x][1]; }
        stats[((size_t)tile_m * Cout + tile_n * 64 + threadIdx.x) * 2 + 0] = a;
        stats[((size_t)tile_m * Cout + tile_n * 64 + threadIdx.x) * 2 + 1] = b;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution with an LDS-resident halo tile ("LDS im2col"):  one workgroup = 8x16 output pixels.
//   Per 64-channel block the (8+2)x(16+2) input pixels are staged in LDS ONCE; the nine filter taps are then nine shifted
//   reads of that tile (address = base + tap offset), so the input crosses the vector-memory path 1.4x instead of 9x and the
//   K loop carries no gather arithmetic at all -- only the small weight tiles stream per tap.  Same epilogue as igemm_kernel.
//   Used for forward and data-gradient of every 3x3/1 layer whose output tiles well (see use_halo()).
// ------------------------------------------------------------------------------------------------------------------
constexpr int HT_H = 8, HT_W = 16, HALO_W = HT_W + 2, HALO_PIX = (HT_H + 2) * HALO_W;   // 10 x 18 = 180 halo pixels
constexpr int HPIX_STRIDE = 144;                                                        // 128 B of channels + 16 B pad
constexpr int HALO_PASSES = (HALO_PIX * 8 + NTHREADS - 1) / NTHREADS;                   // 16-byte chunks per thread per block

// byte offsets of halo pixel (ih, iw) of image n in source 0 / source 1 (channel 0), kOOB when it is padding
template <typename T, int SRC>
__device__ __forceinline__ void halo_pix(const Gather& g, bool ok, int n, int ih, int iw, unsigned& p0, unsigned& p1) {
    if (src_reflect<SRC>(g)) { ih = reflect1(ih, g.IH); iw = reflect1(iw, g.IW); ok = ok & ((unsigned)ih < (unsigned)g.IH) & ((unsigned)iw < (unsigned)g.IW); }
    else ok = ok & ((unsigned)ih < (unsigned)g.IH) & ((unsigned)iw < (unsigned)g.IW);
    p1 = kOOB;
    if (src_mode<SRC>(g) == SDE_SRC_UPCAT) {
        p0 = ok ? (unsigned)(((n * g.H0 + (ih >> 1)) * g.W0 + (iw >> 1)) * g.C0) * (unsigned)sizeof(T) : kOOB;
        p1 = (ok && g.C1 > 0) ? (unsigned)(((n * g.IH + ih) * g.IW + iw) * g.C1) * (unsigned)sizeof(T) : kOOB;
    } else {
        p0 = ok ? (unsigned)(((n * g.H0 + ih) * g.W0 + iw) * g.C0) * (unsigned)sizeof(T) : kOOB;
    }
}

template <typename T, int BN, int WM, int WN, int SRC>
__global__ void __launch_bounds__(NTHREADS, 2) halo3_kernel(IGemmP p) {
    constexpr int V = VecOf<T>::V;
    constexpr int BM = HT_H * HT_W;                  // 128 output pixels
    constexpr int BK = KSTAGE_BYTES / (int)sizeof(T);
    constexpr int WTM = BM / WM, WTN = BN / WN;
    constexpr int FM = WTM / 16, FN = WTN / 16;      // an M fragment = 16 consecutive x of one tile row
    constexpr int B_ROWS_PER_PASS = NTHREADS / 8;
    constexpr int B_PASSES = (BN + B_ROWS_PER_PASS - 1) / B_ROWS_PER_PASS;
    constexpr int CPAD = 4;
    static_assert(WM * WN == 4 && sizeof(T) == 2, "4 waves, 16-bit storage");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sH = smem;                                   // [HALO_PIX][144 B]
    unsigned char* sB = smem + HALO_PIX * HPIX_STRIDE;          // [2][BN][128 B]

    const Gather& g = p.g;
    const int mode = src_mode<SRC>(g);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_n = (p.ldy + BN - 1) / BN, tiles_x = (g.OW + HT_W - 1) / HT_W, tiles_y = (g.OH + HT_H - 1) / HT_H;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = logical / tiles_n, tile_n = logical - tile_m * tiles_n;
    const int img = tile_m / (tiles_x * tiles_y), trem = tile_m - img * (tiles_x * tiles_y);
    const int oy0 = (trem / tiles_x) * HT_H, ox0 = (trem % tiles_x) * HT_W;
    const int n0 = tile_n * BN;
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(g.x0, (long)g.Bn * g.H0 * g.W0 * g.C0 * (long)sizeof(T));
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(g.x1 ? g.x1 : g.x0, g.x1 ? (long)g.Bn * g.IH * g.IW * g.C1 * (long)sizeof(T) : 0);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(p.w, (long)p.Cout * g.Ktot * (long)sizeof(T));

    // ---- halo side: this thread stages 16-byte chunk (tid & 7) of halo pixels (tid >> 3) + 32 j, j < HALO_PASSES
    const int hc = tid & 7;
    unsigned hp0[HALO_PASSES], hp1[HALO_PASSES];
#pragma unroll
    for (int j = 0; j < HALO_PASSES; ++j) {
        const int hp = (tid >> 3) + 32 * j;
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        halo_pix<T, SRC>(g, hp < HALO_PIX, img, oy0 - g.pad + hy, ox0 - g.pad + hx, hp0[j], hp1[j]);
    }
    const int ncb = (g.Cin + BK - 1) / BK;          // 64-channel blocks
    const int nst = ncb * 9;                        // pipeline stages: (channel block, tap)
    uint4 rh[HALO_PASSES];
    auto load_halo = [&](int cb) {
        const int ci = cb * BK + hc * V;
#pragma unroll
        for (int j = 0; j < HALO_PASSES; ++j) {
            unsigned o0, o1 = kOOB;
            if (mode == SDE_SRC_UPCAT) {
                o0 = (ci < g.C0 && hp0[j] != kOOB) ? hp0[j] + (unsigned)ci * (unsigned)sizeof(T) : kOOB;
                o1 = (ci >= g.C0 && ci < g.Cin && hp1[j] != kOOB) ? hp1[j] + (unsigned)(ci - g.C0) * (unsigned)sizeof(T) : kOOB;
            } else {
                o0 = (ci < g.Cin && hp0[j] != kOOB) ? hp0[j] + (unsigned)ci * (unsigned)sizeof(T) : kOOB;
            }
            uint4 v = buf_load16(rs0, o0);
            if (mode == SDE_SRC_UPCAT) {
                const uint4 u = buf_load16(rs1, o1);
                v.x |= u.x; v.y |= u.y; v.z |= u.z; v.w |= u.w;
            }
            rh[j] = v;
        }
    };
    auto store_halo = [&]() {
#pragma unroll
        for (int j = 0; j < HALO_PASSES; ++j) {
            const int hp = (tid >> 3) + 32 * j;
            if (hp < HALO_PIX) *reinterpret_cast<uint4*>(sH + hp * HPIX_STRIDE + hc * 16) = rh[j];
        }
    };
    // ---- weight side: rows r0 + 32 i, chunk cc (as igemm_kernel); stage s = cb * 9 + tap reads k = tap*Cin + cb*64 + cc*8
    const int cc = tid & 7, r0 = tid >> 3;
    uint4 rb[3][B_PASSES];
    auto load_b = [&](int s, auto SET) {
        constexpr int st = decltype(SET)::value;
        const int cb = s / 9, tap = s - cb * 9;
        const int ci = cb * BK + cc * V;
        const int k = tap * g.Cin + ci;
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int row = r0 + i * B_ROWS_PER_PASS, n = n0 + row;
            rb[st][i] = buf_load16(rsw, (ci < g.Cin && row < BN && n < p.Cout) ? (unsigned)(n * g.Ktot + k) * (unsigned)sizeof(T) : kOOB);
        }
    };
    auto store_b = [&](int buf, auto SET) {
        constexpr int st = decltype(SET)::value;
#pragma unroll
        for (int i = 0; i < B_PASSES; ++i) {
            const int row = r0 + i * B_ROWS_PER_PASS;
            if (B_ROWS_PER_PASS * B_PASSES <= BN || row < BN)
                *reinterpret_cast<uint4*>(sB + (size_t)(buf * BN + row) * KSTAGE_BYTES + ((cc ^ (row & 7)) << 4)) = rb[st][i];
        }
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    int abase[FM];                       // LDS byte offset of this lane's pixel (tile row wm*FM+i, column fr), channel chunk fg, tap (0,0)
#pragma unroll
    for (int i = 0; i < FM; ++i) abase[i] = ((wm * FM + i) * HALO_W + fr) * HPIX_STRIDE + fg * 16;
    auto compute_stage = [&](int buf, int tap) {
        const int toff = ((tap / 3) * HALO_W + (tap % 3)) * HPIX_STRIDE;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            uint4 a[FM], b[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) a[i] = *reinterpret_cast<const uint4*>(sH + abase[i] + toff + kk * 64);
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int row = wn * WTN + j * 16 + fr;
                b[j] = *reinterpret_cast<const uint4*>(sB + (size_t)(buf * BN + row) * KSTAGE_BYTES + (((kk * 4 + fg) ^ (row & 7)) << 4));
            }
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = mma64<T>(a[i], b[j], acc[i][j]);
        }
    };
    // one step: request weights of stage st+3, (at the first tap of a block) request the NEXT block's halo into registers,
    // multiply stage st, publish weights of stage st+1; at a block boundary swap the halo tile (all waves are past their reads).
    auto step = [&](int st, auto CUR, auto NXT) {
        if (st >= nst) return;
        const int cb = st / 9, tap = st - cb * 9;
        if (st + 3 < nst) load_b(st + 3, CUR);
        if (tap == 0 && cb + 1 < ncb) load_halo(cb + 1);
        compute_stage(st & 1, tap);
        if (st + 1 < nst) store_b((st + 1) & 1, NXT);
        __syncthreads();
        if (tap == 8 && cb + 1 < ncb) {
            store_halo();
            __syncthreads();
        }
    };
    load_halo(0);
    load_b(0, IC<0>{});
    if (nst > 1) load_b(1, IC<1>{});
    if (nst > 2) load_b(2, IC<2>{});
    store_halo();
    store_b(0, IC<0>{});
    __syncthreads();
    for (int s = 0; s < nst; s += 3) {
        step(s, IC<0>{}, IC<1>{});
        step(s + 1, IC<1>{}, IC<2>{});
        step(s + 2, IC<2>{}, IC<0>{});
    }

    // ---- epilogue (as igemm_kernel; tile row r = ty*16 + tx -> output pixel (oy0+ty, ox0+tx))
    float* sC = reinterpret_cast<float*>(smem);   // [BM][BN + CPAD]
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * WTM + i * 16 + fg * 4 + r, col = wn * WTN + j * 16 + fr;
                sC[row * (BN + CPAD) + col] = acc[i][j][r];
            }
    __syncthreads();
    constexpr int G4 = BN / 4;
    const bool vec_ok = (p.ldy % 4) == 0;
    if (p.bn_y) {
        // BatchNorm-backward epilogue (IGemmP::bn_y): this tile of the data gradient g is masked with relu'(bn(y_bn)) -- the mask re-derived from
        // y_bn exactly as bn_bwd_reduce does -- and stored; every thread keeps (sum gm, sum gm * xhat) of its 4 channels over its rows in
        // registers, the 16 row groups are combined through the (then free) tile buffer.  Host: Cout % BN == 0, ldy == Cout.
        static_assert(NTHREADS % G4 == 0, "a thread keeps one channel group");
        const int c0 = (tid % G4) * 4, n = n0 + c0;
        const float4 mean = *reinterpret_cast<const float4*>(p.bnp + n), rstd = *reinterpret_cast<const float4*>(p.bnp + p.Cout + n);
        const float4 scl = *reinterpret_cast<const float4*>(p.bnp + 2 * p.Cout + n), shf = *reinterpret_cast<const float4*>(p.bnp + 3 * p.Cout + n);
        const float mn[4] = {mean.x, mean.y, mean.z, mean.w}, rs[4] = {rstd.x, rstd.y, rstd.z, rstd.w};
        const float sc[4] = {scl.x, scl.y, scl.z, scl.w}, sf[4] = {shf.x, shf.y, shf.z, shf.w};
        float a1[4] = {0.f, 0.f, 0.f, 0.f}, a2[4] = {0.f, 0.f, 0.f, 0.f};
        for (int id = tid; id < BM * G4; id += NTHREADS) {
            const int row = id / G4;
            const int oy = oy0 + row / HT_W, ox = ox0 + row % HT_W;
            if (!(oy < g.OH && ox < g.OW)) continue;
            const size_t off = ((size_t)(img * g.OH + oy) * g.OW + ox) * p.ldy + n;
            const uint2 yr = *reinterpret_cast<const uint2*>((const T*)p.bn_y + off);
            const T* yt = reinterpret_cast<const T*>(&yr);
            const float4 q = *reinterpret_cast<const float4*>(&sC[row * (BN + CPAD) + c0]);
            const float v[4] = {q.x, q.y, q.z, q.w};
            T o[4];
            if (p.bn_gb) {
                // residual form (IGemmP::bn_gb / bn_out): gm = (g + the other consumer's gradient) * [block output > 0], each sum rounded to the storage
                // type where bn_bwd_reduce would have read it from memory
                const uint2 gr = *reinterpret_cast<const uint2*>((const T*)p.bn_gb + off), orr = *reinterpret_cast<const uint2*>((const T*)p.bn_out + off);
                const T* gt = reinterpret_cast<const T*>(&gr);
                const T* ot = reinterpret_cast<const T*>(&orr);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float yv = to_f32<T>(yt[e]);
                    const float ga = to_f32<T>(from_f32<T>(v[e]));
                    o[e] = from_f32<T>(to_f32<T>(ot[e]) > 0.f ? ga + to_f32<T>(gt[e]) : 0.f);
                    const float t = to_f32<T>(o[e]);
                    a1[e] += t; a2[e] += t * ((yv - mn[e]) * rs[e]);
                }
            } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float yv = to_f32<T>(yt[e]);
                o[e] = from_f32<T>(fmaf(yv, sc[e], sf[e]) > 0.f ? v[e] : 0.f);
                const float t = to_f32<T>(o[e]);
                a1[e] += t; a2[e] += t * ((yv - mn[e]) * rs[e]);
            }
            }
            *reinterpret_cast<uint2*>((T*)p.y + off) = *reinterpret_cast<uint2*>(o);
        }
        __syncthreads();                               // every thread is done with the accumulator tile
        constexpr int PARTS = NTHREADS / G4;
        float* red = sC;                               // [PARTS][BN][2]
        const int part = tid / G4;
#pragma unroll
        for (int e = 0; e < 4; ++e) { red[(part * BN + c0 + e) * 2] = a1[e]; red[(part * BN + c0 + e) * 2 + 1] = a2[e]; }
        __syncthreads();
        if (tid < BN && n0 + tid < p.Cout) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int q = 0; q < PARTS; ++q) { a += red[(q * BN + tid) * 2]; b += red[(q * BN + tid) * 2 + 1]; }
            p.stats[((size_t)tile_m * p.Cout + n0 + tid) * 2 + 0] = a;
            p.stats[((size_t)tile_m * p.Cout + n0 + tid) * 2 + 1] = b;
        }
        return;
    }
    for (int id = tid; id < BM * G4; id += NTHREADS) {
        const int row = id / G4, c0 = (id - row * G4) * 4;
        const int oy = oy0 + row / HT_W, ox = ox0 + row % HT_W, n = n0 + c0;
        const bool rok = oy < g.OH && ox < g.OW;
        if (!rok || n >= p.ldy) {
            if (p.stats && !rok) *reinterpret_cast<float4*>(&sC[row * (BN + CPAD) + c0]) = make_float4(0.f, 0.f, 0.f, 0.f);   // not part of the statistics
            continue;
        }
        const float4 q = *reinterpret_cast<const float4*>(&sC[row * (BN + CPAD) + c0]);
        float v[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = v[e];
            if (p.bias && n + e < p.Cout) t += p.bias[n + e];
            if (p.act == SDE_ACT_ELU) t = t > 0.f ? t : expm1f(t);
            if (n + e >= p.Cout) t = 0.f;
            v[e] = t;
        }
        T* dst = (T*)p.y + ((size_t)(img * g.OH + oy) * g.OW + ox) * p.ldy + n;
        T o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = from_f32<T>(v[e]);
        if (vec_ok && n + 4 <= p.ldy) *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<uint2*>(o);
        else for (int e = 0; e < 4 && n + e < p.ldy; ++e) dst[e] = o[e];
        if (p.stats)
            *reinterpret_cast<float4*>(&sC[row * (BN + CPAD) + c0]) = make_float4(to_f32<T>(o[0]), to_f32<T>(o[1]), to_f32<T>(o[2]), to_f32<T>(o[3]));
    }
    if (p.stats) {
        __syncthreads();
        constexpr int PARTS = NTHREADS / BN >= 1 ? NTHREADS / BN : 1;
        float* red = sC + BM * (BN + CPAD);
        const int c = tid % BN, part = tid / BN;
        float s1 = 0.f, s2 = 0.f;
        if (part < PARTS)     // rows outside the image were zeroed above, so they add nothing
            for (int r = part; r < BM; r += PARTS) { const float t = sC[r * (BN + CPAD) + c]; s1 += t; s2 += t * t; }
        if (part < PARTS) { red[(part * BN + c) * 2] = s1; red[(part * BN + c) * 2 + 1] = s2; }
        __syncthreads();
        if (tid < BN && n0 + tid < p.Cout) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int q = 0; q < PARTS; ++q) { a += red[(q * BN + tid) * 2]; b += red[(q * BN + tid) * 2 + 1]; }
            p.stats[((size_t)tile_m * p.Cout + n0 + tid) * 2 + 0] = a;
            p.stats[((size_t)tile_m * p.Cout + n0 + tid) * 2 + 1] = b;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight gradient:  dW[Cout, K] = sum_m dY[m, Cout]^T * im2col(X)[m, K], split over pixel ranges into fp32 slabs
// ------------------------------------------------------------------------------------------------------------------
struct WGradP {
    Gather g;
    const void* dy;   // [M][ldd]
    float* slab;      // [splits][Cout][Ktot]
    int Cout, ldd, rows_per_split;
    int single_buf;   // one LDS stage buffer instead of two (see wgrad_kernel)
};

template <typename T> struct WGTraits;
template <> struct WGTraits<bf16_t> { static constexpr int BR = 64; };   // pixels per stage
template <> struct WGTraits<half_t> { static constexpr int BR = 64; };
template <> struct WGTraits<float> { static constexpr int BR = 32; };

// Transposed fragment read: rows of the LDS tile are pixels (reduction index), columns are output rows/cols.
// Returns the 64-byte-K-sub-block operand of lane `lane` for 16 consecutive columns starting at col0, pixels p0..
template <typename T> struct TrRead;
template <> struct TrRead<bf16_t> {
    // 32 pixels per sub-block; lane group g needs pixels 8g..8g+7 of column (lane&15): two ds_read_b64_tr_b16
    static __device__ __forceinline__ uint4 rd(const unsigned char* tile, int stride, int p0, int col0, int lane) {
        const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
        // rows whose bit 3 is set are stored with their two 128-byte halves swapped (see wgrad_kernel): with the 288-byte row
        // stride the eight 4-lane groups of a 32-lane half then read eight disjoint 8-bank groups (conflict-free tr-reads)
        const int row = p0 + 8 * g + q;
        const unsigned char* a0 = tile + (size_t)row * stride + (size_t)((((col0 + 4 * pp) * 2)) ^ ((row & 8) << 4));
#if defined(SDE_WGRAD_SAFE_READ)
        // reference path: element-wise transposed gather (slow, used to validate the tr-read mapping)
        const int col = col0 + i;
        s16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const short*>(tile + (size_t)(p0 + 8 * g + j) * stride + (size_t)((col * 2) ^ (((p0 + 8 * g + j) & 8) << 4)));
        (void)a0;
        return __builtin_bit_cast(uint4, v);
#else
        typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0 + 4 * (size_t)stride));
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(uint4, v);
#endif
    }
};
template <> struct TrRead<half_t> : TrRead<bf16_t> {};      // the transposed LDS read moves 16-bit elements, whatever they encode
template <> struct TrRead<float> {
    // 16 pixels per sub-block; element j of lane group g is pixel 4g+j (same map as mma64<float>)
    static __device__ __forceinline__ uint4 rd(const unsigned char* tile, int stride, int p0, int col0, int lane) {
        const int g = lane >> 4, i = lane & 15;
        const unsigned char* a = tile + (size_t)(p0 + 4 * g) * stride + (size_t)(col0 + i) * 4;
        uint4 v;
        v.x = *reinterpret_cast<const uint32_t*>(a);
        v.y = *reinterpret_cast<const uint32_t*>(a + stride);
        v.z = *reinterpret_cast<const uint32_t*>(a + 2 * (size_t)stride);
        v.w = *reinterpret_cast<const uint32_t*>(a + 3 * (size_t)stride);
        return v;
    }
};

template <typename T, int BMG, int BNG, int WM, int WN, int SRC, bool ONE_TAP>
__global__ void __launch_bounds__(NTHREADS, (BMG <= 64 && sizeof(T) == 2) ? 3 : 2) wgrad_kernel(WGradP p) {
    constexpr int V = VecOf<T>::V;
    constexpr int BR = WGTraits<T>::BR;
    constexpr int SUBP = 64 / (int)sizeof(T);          // pixels per MFMA sub-block (32 bf16 / 16 f32)
    constexpr int NSUB = BR / SUBP;
    constexpr int WTM = BMG / WM, WTN = BNG / WN;
    constexpr int FM = WTM / 16, FN = WTN / 16;
    // LDS row strides (bytes).  bf16: 288 B = 72 banks (== 8 mod 64) for every tile width, plus the half-swap XOR of rows with
    // bit 3 set -> transposed reads are bank-conflict free; fp32 (parity mode): plain padded rows.
    constexpr int STRA = sizeof(T) == 2 ? 288 : BMG * (int)sizeof(T) + 16;
    constexpr int STRB = sizeof(T) == 2 ? 288 : BNG * (int)sizeof(T) + 16;
    constexpr int SWZ = sizeof(T) == 2 ? 1 : 0;
    static_assert(sizeof(T) != 2 || (BMG <= 128 && BNG <= 128), "bf16 rows must fit the 288-byte stride with the half swap");
    constexpr int ACH = BMG / V, BCH = BNG / V;        // 16-byte chunks per pixel row
    constexpr int TPP = NTHREADS / BR;                 // threads per pixel row (4 bf16 / 8 f32)
    constexpr int CPTB = BCH / TPP;                    // B chunks per thread (4)
    constexpr int CPTA = (ACH + TPP - 1) / TPP;        // A chunks per thread (<= 4)
    static_assert(WM * WN == 4, "4 waves");
    static_assert(BCH % TPP == 0, "B tile chunking");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sA = smem;                                          // [nbuf][BR][STRA]
    unsigned char* sB = smem + (p.single_buf ? 1 : 2) * BR * STRA;     // [nbuf][BR][STRB]

    const Gather& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tiles_co = (p.Cout + BMG - 1) / BMG, tiles_k = (g.Ktot + BNG - 1) / BNG;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / (tiles_co * tiles_k), rem_t = logical - split * (tiles_co * tiles_k);
    const int co0 = (rem_t % tiles_co) * BMG, k0 = (rem_t / tiles_co) * BNG;       // all tiles of one pixel range run together on one XCD
    const int mbeg = split * p.rows_per_split, mend = min(g.M, mbeg + p.rows_per_split);

    // Each thread owns pixel row `px` of every stage and fixed chunk columns: the im2col column -> (tap, ci) map is
    // computed once; the pixel -> (n, oh, ow) map is advanced incrementally by BR pixels per stage.
    const int px = tid / TPP, tq = tid % TPP;
    int bkh[CPTB], bkw[CPTB], bci[CPTB];
#pragma unroll
    for (int c = 0; c < CPTB; ++c) {
        const int k = k0 + (tq * CPTB + c) * V;
        if (k < g.Ktot) {
            const int tap = k / g.Cin;
            bci[c] = k - tap * g.Cin; bkh[c] = tap / g.KW; bkw[c] = tap - bkh[c] * g.KW;
        } else {
            bci[c] = -1; bkh[c] = 0; bkw[c] = 0;
        }
    }
    int pn, poh, pow_;     // pixel of the NEXT stage to load
    {
        const int m = mbeg + px;
        pn = m / (g.OH * g.OW);
        const int rem = m - pn * (g.OH * g.OW);
        poh = rem / g.OW; pow_ = rem - poh * g.OW;
    }
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(g.x0, (long)g.Bn * g.H0 * g.W0 * g.C0 * (long)sizeof(T));
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(g.x1 ? g.x1 : g.x0, g.x1 ? (long)g.Bn * g.IH * g.IW * g.C1 * (long)sizeof(T) : 0);
    const __amdgpu_buffer_rsrc_t rsd = make_rsrc(p.dy, (long)g.M * p.ldd * (long)sizeof(T));
    const int mode = src_mode<SRC>(g);
    uint4 ra[3][CPTA], rb[3][CPTB];     // three register stages in flight (see igemm_kernel)
    auto load_stage = [&](int s, auto SET) {
        constexpr int st = decltype(SET)::value;
        const int m = mbeg + s * BR + px;
        const bool mok = m < mend;
#pragma unroll
        for (int c = 0; c < CPTA; ++c) {
            const int ch = tq * CPTA + c;
            const int co = co0 + ch * V;
            const unsigned od = (mok && ch < ACH && co < p.ldd) ? (unsigned)(m * p.ldd + co) * (unsigned)sizeof(T) : kOOB;
            ra[st][c] = buf_load16(rsd, od);
        }
        const int ih0 = poh * g.stride - g.pad, iw0 = pow_ * g.stride - g.pad;
        if (SRC == SRC_1X1) {
            const unsigned rowoff = (unsigned)(((pn * g.H0 + ih0) * g.W0 + iw0) * g.C0) * (unsigned)sizeof(T);
#pragma unroll
            for (int c = 0; c < CPTB; ++c)
                rb[st][c] = buf_load16(rs0, (mok && bci[c] >= 0) ? rowoff + (unsigned)bci[c] * (unsigned)sizeof(T) : kOOB);
        } else if (ONE_TAP) {
            unsigned o0, o1;
            gather_off<T, SRC>(g, mok && bci[0] >= 0, pn, ih0 + bkh[0], iw0 + bkw[0], bci[0], o0, o1);
#pragma unroll
            for (int c = 0; c < CPTB; ++c) {
                uint4 v = buf_load16(rs0, o0 + 16u * c);
                if (mode == SDE_SRC_UPCAT) {
                    const uint4 u = buf_load16(rs1, o1 + 16u * c);
                    v.x |= u.x; v.y |= u.y; v.z |= u.z; v.w |= u.w;
                }
                rb[st][c] = v;
            }
        } else {
#pragma unroll
            for (int c = 0; c < CPTB; ++c) {
                unsigned o0, o1;
                gather_off<T, SRC>(g, mok && bci[c] >= 0, pn, ih0 + bkh[c], iw0 + bkw[c], bci[c], o0, o1);
                uint4 v = buf_load16(rs0, o0);
                if (mode == SDE_SRC_UPCAT) {
                    const uint4 u = buf_load16(rs1, o1);
                    v.x |= u.x; v.y |= u.y; v.z |= u.z; v.w |= u.w;
                }
                rb[st][c] = v;
            }
        }
        pow_ += BR;
        while (pow_ >= g.OW) { pow_ -= g.OW; if (++poh == g.OH) { poh = 0; ++pn; } }
    };
    auto store_stage = [&](int buf, auto SET) {
        constexpr int st = decltype(SET)::value;
#pragma unroll
        for (int c = 0; c < CPTA; ++c) {
            const int ch = tq * CPTA + c;
            if (ch < ACH) *reinterpret_cast<uint4*>(sA + (size_t)(buf * BR + px) * STRA + ((ch * 16) ^ (SWZ * ((px & 8) << 4)))) = ra[st][c];
        }
#pragma unroll
        for (int c = 0; c < CPTB; ++c)
            *reinterpret_cast<uint4*>(sB + (size_t)(buf * BR + px) * STRB + (((tq * CPTB + c) * 16) ^ (SWZ * ((px & 8) << 4)))) = rb[st][c];
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int ns = (mend - mbeg + BR - 1) / BR;
    auto compute_stage = [&](int buf) {
        const unsigned char* tA = sA + (size_t)buf * BR * STRA;
        const unsigned char* tB = sB + (size_t)buf * BR * STRB;
#pragma unroll
        for (int kk = 0; kk < NSUB; ++kk) {
            uint4 a[FM], b[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) a[i] = TrRead<T>::rd(tA, STRA, kk * SUBP, wm * WTM + i * 16, lane);
#pragma unroll
            for (int j = 0; j < FN; ++j) b[j] = TrRead<T>::rd(tB, STRB, kk * SUBP, wn * WTN + j * 16, lane);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = mma64<T>(a[i], b[j], acc[i][j]);
        }
    };
    // p.single_buf: one LDS stage buffer instead of two (half the LDS -> one more workgroup per CU for the <= 64-row tiles) at the
    // price of a second barrier per stage; the three register stages still keep two stages of global loads in flight
    const int dbl = p.single_buf ? 0 : 1;
    auto step = [&](int st, auto CUR, auto NXT) {
        if (st >= ns) return;
        if (st + 3 < ns) load_stage(st + 3, CUR);
        compute_stage((st & 1) & dbl);
        if (!dbl) __syncthreads();
        if (st + 1 < ns) store_stage(((st + 1) & 1) & dbl, NXT);
        __syncthreads();
    };
    if (ns > 0) load_stage(0, IC<0>{});
    if (ns > 1) load_stage(1, IC<1>{});
    if (ns > 2) load_stage(2, IC<2>{});
    if (ns > 0) store_stage(0, IC<0>{});
    __syncthreads();
    for (int s = 0; s < ns; s += 3) {
        step(s, IC<0>{}, IC<1>{});
        step(s + 1, IC<1>{}, IC<2>{});
        step(s + 2, IC<2>{}, IC<0>{});
    }
    // slab[split][co][k]
    const int fr = lane & 15, fg = lane >> 4;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + wm * WTM + i * 16 + fg * 4 + r, k = k0 + wn * WTN + j * 16 + fr;
                if (co < p.Cout && k < g.Ktot) p.slab[((size_t)split * p.Cout + co) * g.Ktot + k] = acc[i][j][r];
            }
}

// Sum the slabs in a fixed order and write the master OIHW fp32 gradient (skipping padded input channels), for MANY layers in
// ONE launch: block b finds its (layer, output channel) by binary search in the prefix sums of the layers' Cout.
//   * one workgroup per output channel; the K-major sums go through LDS so that both the slab reads (float4 rows) and the OIHW
//     writes are contiguous -- the [tap][ci] -> [ci][tap] transpose happens in LDS, in chunks of input channels when K*4 B
//     exceeds the LDS budget (PackNet's 16384-channel layers have K = 147456);
//   * tall slab stacks are summed as chunks of `chunk` rows (even / odd rows in two chains), then the <= SDE_WGRAD_FOLD_ROWS
//     chunk sums four ways: fixed order, bit-reproducible;
//   * the item table travels BY VALUE in the kernel arguments (<= 4 KB): no device table whose upload / lifetime has to be
//     ordered against a launch that runs long after the host has moved on; a captured hipGraph node carries it in its parameters.
constexpr int WREDUCE_MAX = 120;
constexpr int WREDUCE_LDS_FLOATS = 16384;        // 64 KB: two workgroups per CU
struct WReduceArg { const float* slab; float* dw; int end; unsigned short splits, chunk, Cout, KHW, Cin_pad, Cin_real /* | 0x8000: accumulate */; };
static_assert(sizeof(WReduceArg) == 32, "WReduceArg packing");
struct WReduceBatch { int n, lds_floats; WReduceArg it[WREDUCE_MAX]; };

__host__ __device__ inline int wreduce_cb(int KHW, int Cin_pad, int lds_floats) {      // input channels per LDS chunk (multiple of 4)
    const int cb = (lds_floats / KHW) & ~3;
    return cb < Cin_pad ? cb : Cin_pad;
}

__global__ void __launch_bounds__(256) wgrad_reduce_batched_kernel(const WReduceBatch batch) {
    extern __shared__ __attribute__((aligned(16))) float sk[];      // [KHW][cb]
    int lo = 0, hi = batch.n - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (batch.it[mid].end > b) hi = mid; else lo = mid + 1;
    }
    const WReduceArg it = batch.it[lo];
    const int co = b - (lo ? batch.it[lo - 1].end : 0);
    const int KHW = it.KHW, Cin_pad = it.Cin_pad, K = KHW * Cin_pad, Cout = it.Cout, splits = it.splits, chunk = it.chunk & 0x7fff;
    const bool ohwi = (it.chunk & 0x8000) != 0;          // dw in channels-last order [co][tap][ci]: no transpose on the way out
    const int cin_real = it.Cin_real & 0x7fff, accumulate = it.Cin_real >> 15;
    const int CB = wreduce_cb(KHW, Cin_pad, batch.lds_floats);
    const size_t total4 = (size_t)Cout * K / 4;                     // K is a multiple of 4: rows are walked as float4 (16 B per lane)
    auto add4 = [](float4& a, const float4 v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
    float* o = it.dw + (size_t)co * cin_real * KHW;
    for (int ci0 = 0; ci0 < Cin_pad; ci0 += CB) {
        const int cb = min(CB, Cin_pad - ci0), cb4 = cb / 4;
        for (int q = threadIdx.x; q < KHW * cb4; q += 256) {
            const int tap = q / cb4, c4 = q - tap * cb4;
            const float4* src = reinterpret_cast<const float4*>(it.slab) + (size_t)co * (K / 4) + (tap * Cin_pad + ci0) / 4 + c4;
            const float4 z = {0.f, 0.f, 0.f, 0.f};
            float4 s0 = z, s1 = z, s2 = z, s3 = z;
            if (chunk == 1) {
                int sp = 0;
                for (; sp + 3 < splits; sp += 4) {
                    add4(s0, src[(size_t)sp * total4]); add4(s1, src[(size_t)(sp + 1) * total4]);
                    add4(s2, src[(size_t)(sp + 2) * total4]); add4(s3, src[(size_t)(sp + 3) * total4]);
                }
                for (; sp < splits; ++sp) add4(s0, src[(size_t)sp * total4]);
            } else {
                auto part = [&](int ro) {
                    const int r0 = ro * chunk, r1 = min(splits, r0 + chunk);
                    float4 a = z, c = z;
                    int r = r0;
                    for (; r + 1 < r1; r += 2) { add4(a, src[(size_t)r * total4]); add4(c, src[(size_t)(r + 1) * total4]); }
                    if (r < r1) add4(a, src[(size_t)r * total4]);
                    add4(a, c);
                    return a;
                };
                const int rows = (splits + chunk - 1) / chunk;
                int ro = 0;
                for (; ro + 3 < rows; ro += 4) { add4(s0, part(ro)); add4(s1, part(ro + 1)); add4(s2, part(ro + 2)); add4(s3, part(ro + 3)); }
                for (; ro < rows; ++ro) add4(s0, part(ro));
            }
            add4(s0, s1); add4(s2, s3); add4(s0, s2);
            reinterpret_cast<float4*>(sk)[q] = s0;                 // sk[tap][c4*4 ..]
        }
        __syncthreads();
        const int creal = min(cb, cin_real - ci0);                  // real channels of this chunk (<= 0: padding only)
        if (ohwi) {
            for (int j = threadIdx.x; j < creal * KHW; j += 256) {
                const int tap = j / creal, ci = j - tap * creal;
                const float v = sk[tap * cb + ci];
                float* dst = o + (size_t)tap * cin_real + ci0 + ci;
                *dst = accumulate ? *dst + v : v;
            }
        } else
        for (int j = threadIdx.x; j < creal * KHW; j += 256) {
            const int ci = j / KHW, tap = j - ci * KHW;
            const float v = sk[tap * cb + ci];
            float* dst = o + (size_t)(ci0 + ci) * KHW + tap;
            *dst = accumulate ? *dst + v : v;
        }
        __syncthreads();
    }
}

// Channels-last gradients (SDE_WREDUCE_OHWI) of layers without channel padding have exactly the slab's memory order: the reduction is then
// a plain fixed-order sum of float4 rows -- no LDS, no transpose, every access a whole 16-byte group of consecutive lanes.
// One workgroup = 1024 consecutive floats of one item; same summation order as the kernel above (chunk sums of even / odd rows, four ways).
__global__ void __launch_bounds__(256) wgrad_sum_batched_kernel(const WReduceBatch batch) {
    // A workgroup = 64 consecutive float4 columns (1 KB of every slab row) x 4 split lanes (one wave each): lane r sums slabs r, r + 4, r + 8, ...
    // with four loads in flight, the four partial sums are combined through LDS in a fixed order ((p0 + p1) + (p2 + p3)): bit-reproducible.
    // (One thread per column walking ALL slabs left the small layers with 16-64 workgroups of serial load chains: 15 us per launch on average,
    // 120 us for the 64-slab stacks of layer 1; profiles/r02g_sup50_kernel_stats.csv is the state before this form.)
    __shared__ float4 comb[3][64];
    int lo = 0, hi = batch.n - 1;
    const int b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (batch.it[mid].end > b) hi = mid; else lo = mid + 1;
    }
    const WReduceArg it = batch.it[lo];
    const int local = b - (lo ? batch.it[lo - 1].end : 0);
    const size_t total4 = (size_t)it.Cout * it.KHW * it.Cin_pad / 4;
    const int col = threadIdx.x & 63, r = threadIdx.x >> 6;
    const size_t q = (size_t)local * 64 + col;
    const bool live = q < total4;
    const int splits = it.splits, accumulate = it.Cin_real >> 15;
    auto add4 = [](float4& a, const float4 v) { a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; };
    const float4 z = {0.f, 0.f, 0.f, 0.f};
    float4 s0 = z, s1 = z, s2 = z, s3 = z;
    if (live) {
        const float4* src = reinterpret_cast<const float4*>(it.slab) + q;
        int sp = r;
        for (; sp + 12 < splits; sp += 16) {
            const float4 v0 = src[(size_t)sp * total4], v1 = src[(size_t)(sp + 4) * total4], v2 = src[(size_t)(sp + 8) * total4], v3 = src[(size_t)(sp + 12) * total4];
            add4(s0, v0); add4(s1, v1); add4(s2, v2); add4(s3, v3);
        }
        for (; sp < splits; sp += 4) add4(s0, src[(size_t)sp * total4]);
    }
    add4(s0, s1); add4(s2, s3); add4(s0, s2);
    if (r > 0) comb[r - 1][col] = s0;
    __syncthreads();
    if (r == 0 && live) {
        float4 p1 = comb[0][col];
        const float4 p2 = comb[1][col], p3 = comb[2][col];
        add4(s0, p1);
        float4 t = p2; add4(t, p3);
        add4(s0, t);
        float4* dst = reinterpret_cast<float4*>(it.dw) + q;
        if (accumulate) { const float4 o = *dst; add4(s0, o); }
        *dst = s0;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight packing (master fp32 OIHW -> K-major operands)
// ------------------------------------------------------------------------------------------------------------------
// forward operand: Wp[co][kh][kw][ci_pad]
template <typename T>
__global__ void __launch_bounds__(256) pack_w_fwd_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, int KH, int KW,
                                                         int Cin_pad, int Cout_rows) {
    const size_t total = (size_t)Cout_rows * KH * KW * Cin_pad;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ci = (int)(i % Cin_pad);
        const int tap = (int)((i / Cin_pad) % (KH * KW));
        const int co = (int)(i / ((size_t)Cin_pad * KH * KW));
        float v = 0.f;
        if (ci < Cin && co < Cout) v = w[((size_t)co * Cin + ci) * KH * KW + tap];
        out[i] = from_f32<T>(v);
    }
}
// data-gradient operand: Wd[ci][kh'][kw'][co_pad] with kh' = KH-1-kh, kw' = KW-1-kw (flipped taps)
template <typename T>
__global__ void __launch_bounds__(256) pack_w_dgrad_kernel(const float* __restrict__ w, T* __restrict__ out, int Cout, int Cin, int KH, int KW,
                                                           int Cout_pad, int Cin_rows) {
    const size_t total = (size_t)Cin_rows * KH * KW * Cout_pad;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int co = (int)(i % Cout_pad);
        const int tapf = (int)((i / Cout_pad) % (KH * KW));
        const int ci = (int)(i / ((size_t)Cout_pad * KH * KW));
        const int tap = KH * KW - 1 - tapf;
        float v = 0.f;
        if (ci < Cin && co < Cout) v = w[((size_t)co * Cin + ci) * KH * KW + tap];
        out[i] = from_f32<T>(v);
    }
}

// All layers' operands in ONE launch.  items[] (device, built once per model) describe each layer; a workgroup owns a
// (32 output channels) x (PACK_CI(khw) input channels) x (all taps) tile of one layer: the OIHW rows are read ONCE, coalesced, into
// LDS and written out twice -- as the forward operand [co][tap][ci] and as the tap-flipped data-gradient operand [ci][tap][co] --
// in contiguous runs.  (The per-element form of this kernel re-fetched every 128-byte line of the master weights ~khw times with
// 64 distinct lines per wave instruction and cost 0.41 ms per step on ResNet-50; `end` = exclusive prefix sum of the tiles.)
constexpr int PACK_CO = 32, PACK_LDS_FLOATS = 12800;
__host__ __device__ inline int pack_ci(int khw) {        // input channels per tile: multiple of 8, 32*ci*khw floats fit the LDS tile
    int c = (PACK_LDS_FLOATS / PACK_CO - 1) / khw;
    c &= ~7;
    return c < 8 ? 8 : (c > 256 ? 256 : c);
}

template <typename T>
__global__ void __launch_bounds__(256) pack_batched_kernel(const sde_pack_item* __restrict__ items, int n) {
    __shared__ float tile[PACK_LDS_FLOATS];
    int lo = 0, hi = n - 1;
    const long b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (items[mid].end > b) hi = mid; else lo = mid + 1;
    }
    const sde_pack_item it = items[lo];
    const int local = (int)(b - (lo ? items[lo - 1].end : 0));
    const int khw = it.KH * it.KW, CI = pack_ci(khw);
    const int tiles_ci = (it.Cin_pad + CI - 1) / CI;
    const int co0 = (local / tiles_ci) * PACK_CO, ci0 = (local % tiles_ci) * CI;
    const int n_ci_real = max(0, min(CI, it.Cin - ci0));          // channels that exist in the master weights
    const int n_ci = min(CI, it.Cin_pad - ci0), n_co = min(PACK_CO, it.Cout_pad - co0);
    const int seg = n_ci_real * khw, rs = CI * khw + 1;           // row stride odd: the transposed reads below are conflict-free
    // Fast path (16-bit operands of channels-last master weights without channel padding -- every layer but the stem and the 1-channel heads):
    // float4 loads, LDS tile in source order [co][tap][ci], and 16-byte stores of 8 values -- 8 consecutive ci of the forward operand, 8
    // consecutive co of the data-gradient operand -- instead of one index computation and one 2-byte store per element.
    if constexpr (sizeof(T) == 2)
    if ((it.src_layout == SDE_W_OHWI || khw == 1) && it.Cin_pad == it.Cin && it.Cout_pad == it.Cout && (it.Cin & 7) == 0 && (it.Cout & 7) == 0 && it.dst_fwd && it.dst_dgrad) {
        const int nq = n_ci >> 2;                                  // float4 per (co, tap) run
        for (int idx = threadIdx.x; idx < n_co * khw * nq; idx += 256) {
            const int q4 = idx % nq, t2 = idx / nq, tap = t2 % khw, co_l = t2 / khw;
            const float4 v = *reinterpret_cast<const float4*>(it.src + ((size_t)(co0 + co_l) * khw + tap) * it.Cin + ci0 + q4 * 4);
            float* d = tile + co_l * rs + tap * n_ci + q4 * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
        typedef __attribute__((ext_vector_type(8))) unsigned short us8;
        const int n8 = n_ci >> 3;
        T* dstf = (T*)it.dst_fwd;
        for (int idx = threadIdx.x; idx < n_co * khw * n8; idx += 256) {
            const int c8 = idx % n8, t2 = idx / n8, tap = t2 % khw, co_l = t2 / khw;
            const float* sp = tile + co_l * rs + tap * n_ci + c8 * 8;
            us8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = __builtin_bit_cast(unsigned short, from_f32<T>(sp[j]));
            *reinterpret_cast<us8*>(dstf + ((size_t)(co0 + co_l) * khw + tap) * it.Cin_pad + ci0 + c8 * 8) = o;
        }
        const int m8 = n_co >> 3;
        T* dstd = (T*)it.dst_dgrad;
        for (int idx = threadIdx.x; idx < n_ci * khw * m8; idx += 256) {
            const int o8 = idx % m8, t2 = idx / m8, tapf = t2 % khw, ci_l = t2 / khw;
            const float* sp = tile + (o8 * 8) * rs + (khw - 1 - tapf) * n_ci + ci_l;
            us8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = __builtin_bit_cast(unsigned short, from_f32<T>(sp[j * rs]));
            *reinterpret_cast<us8*>(dstd + ((size_t)(ci0 + ci_l) * khw + tapf) * it.Cout_pad + co0 + o8 * 8) = o;
        }
        return;
    }
    if (seg > 0 && it.src_layout == SDE_W_OHWI) {      // channels-last master weights [co][tap][ci]: contiguous along ci
        for (int idx = threadIdx.x; idx < PACK_CO * seg; idx += 256) {
            const int ci_l = idx % n_ci_real, t2 = idx / n_ci_real, tap = t2 % khw, co_l = t2 / khw;
            const int co = co0 + co_l;
            tile[co_l * rs + ci_l * khw + tap] = co < it.Cout ? it.src[((size_t)co * khw + tap) * it.Cin + ci0 + ci_l] : 0.f;
        }
    } else if (seg > 0)
        for (int idx = threadIdx.x; idx < PACK_CO * seg; idx += 256) {
            const int co_l = idx / seg, e = idx - co_l * seg;
            const int co = co0 + co_l;
            tile[co_l * rs + e] = co < it.Cout ? it.src[((size_t)co * it.Cin + ci0) * khw + e] : 0.f;
        }
    __syncthreads();
    if (it.dst_fwd) {             // [Cout_pad][khw][Cin_pad]
        T* dst = (T*)it.dst_fwd;
        for (int idx = threadIdx.x; idx < n_co * khw * n_ci; idx += 256) {
            const int ci_l = idx % n_ci, t2 = idx / n_ci, tap = t2 % khw, co_l = t2 / khw;
            const float v = (ci_l < n_ci_real && co0 + co_l < it.Cout) ? tile[co_l * rs + ci_l * khw + tap] : 0.f;
            dst[((size_t)(co0 + co_l) * khw + tap) * it.Cin_pad + ci0 + ci_l] = from_f32<T>(v);
        }
    }
    if (it.dst_dgrad) {           // [Cin_pad][khw][Cout_pad], taps flipped
        T* dst = (T*)it.dst_dgrad;
        for (int idx = threadIdx.x; idx < n_ci * khw * n_co; idx += 256) {
            const int co_l = idx % n_co, t2 = idx / n_co, tapf = t2 % khw, ci_l = t2 / khw;
            const float v = (ci_l < n_ci_real && co0 + co_l < it.Cout) ? tile[co_l * rs + ci_l * khw + (khw - 1 - tapf)] : 0.f;
            dst[((size_t)(ci0 + ci_l) * khw + tapf) * it.Cout_pad + co0 + co_l] = from_f32<T>(v);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int WM, int WN, int SRC, bool ONE_TAP>
int launch_igemm(const IGemmP& p, hipStream_t s) {
    constexpr int stage = 2 * (BM + BN) * KSTAGE_BYTES;
    constexpr int ctile = BM * (BN + 4) * 4 + NTHREADS * 2 * 4;      // staging tile + per-part column sums
    constexpr int lds = stage > ctile ? stage : ctile;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<T, BM, BN, WM, WN, SRC, ONE_TAP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    dim3 grid(sde_cdiv(p.g.M, BM) * sde_cdiv(p.ldy, BN) * p.ksplit);
    hipLaunchKernelGGL((igemm_kernel<T, BM, BN, WM, WN, SRC, ONE_TAP>), grid, dim3(NTHREADS), lds, s, p);
    return 0;
}

// Tile choice (one place): returns BM*1000 + BN.  Small-N layers get narrow N tiles (the GEMM is then A-bandwidth bound);
// layers whose 128x128 grid would not fill the 256 CUs fall back to 64x64 tiles.
int pick_tile(long M, int N, int Ktot) {
    // Measured end to end in round 1 (Supervised R50, bs 12): 64x64 tiles for every N > 32 layer 10.10 ms/step, 128x64 10.28, 128x128 10.47 -- the
    // register-staged loop is latency-bound, not MFMA-bound (111 VGPRs / 32 KB LDS = 4 workgroups per CU vs 222 / 70 KB = 2), so occupancy wins
    // over operand reuse.  The wide instantiations were removed with their switches.
    (void)M; (void)Ktot;
    if (N > 32) return 64064;
    if (N > 16) return 128032;
    return 128016;
}

// which specialisation of the gather a descriptor gets (bf16); fp32 (parity mode) keeps the run-time generic kernel
int src_kind(const Gather& g) {
    if (g.mode == SDE_SRC_UPCAT) return SRC_UPCAT_REFLECT;
    if (g.mode == SDE_SRC_ZEROINS) return SRC_ZEROINS_ZERO;
    if (g.KH == 1 && g.KW == 1 && g.pad == 0 && !g.reflect) return SRC_1X1;
    return g.reflect ? SRC_PLAIN_REFLECT : SRC_PLAIN_ZERO;
}
bool one_tap_ok(const Gather& g, int group) {      // group = elements covered by one thread per stage
    return (g.Cin % group == 0) && (g.mode != SDE_SRC_UPCAT || g.C0 % group == 0);
}

template <typename T, int BM, int BN, int WM, int WN>
int dispatch_src(const IGemmP& p, hipStream_t s) {
    constexpr int group = (8 / (NTHREADS / BM)) * VecOf<T>::V;
    if (sizeof(T) == 4 || (p.g.mode == SDE_SRC_UPCAT && !p.g.reflect) || (p.g.mode == SDE_SRC_ZEROINS && p.g.reflect))
        return launch_igemm<T, BM, BN, WM, WN, SRC_RUNTIME, false>(p, s);
    const bool ot = one_tap_ok(p.g, group);
    switch (src_kind(p.g)) {
        case SRC_1X1: return launch_igemm<T, BM, BN, WM, WN, SRC_1X1, true>(p, s);
        case SRC_PLAIN_ZERO: return ot ? launch_igemm<T, BM, BN, WM, WN, SRC_PLAIN_ZERO, true>(p, s) : launch_igemm<T, BM, BN, WM, WN, SRC_PLAIN_ZERO, false>(p, s);
        case SRC_PLAIN_REFLECT: return ot ? launch_igemm<T, BM, BN, WM, WN, SRC_PLAIN_REFLECT, true>(p, s) : launch_igemm<T, BM, BN, WM, WN, SRC_PLAIN_REFLECT, false>(p, s);
        case SRC_UPCAT_REFLECT: return ot ? launch_igemm<T, BM, BN, WM, WN, SRC_UPCAT_REFLECT, true>(p, s) : launch_igemm<T, BM, BN, WM, WN, SRC_UPCAT_REFLECT, false>(p, s);
        default: return ot ? launch_igemm<T, BM, BN, WM, WN, SRC_ZEROINS_ZERO, true>(p, s) : launch_igemm<T, BM, BN, WM, WN, SRC_ZEROINS_ZERO, false>(p, s);
    }
}

// ---- LDS-halo 3x3 kernel: when it applies and how it tiles (one place; sde_conv_fwd_tiles_m / _variant use it too)
int g_halo_min_blocks = 192;      // below this many workgroups the 64x64 generic tiles fill the chip better (sde_conv_set_halo_min_blocks)

long halo_tiles(const Gather& g) { return (long)g.Bn * sde_cdiv(g.OH, HT_H) * sde_cdiv(g.OW, HT_W); }
// N tile of the halo kernel: 64, or 32 / 16 for narrow outputs
int halo_bn(const Gather& g, int ldy) {
    (void)g;
    return ldy > 32 ? 64 : (ldy > 16 ? 32 : 16);      // 64 wide at most: 3 waves / SIMD (measured 10.05 vs 10.10 ms/step against a 128-wide tile)
}
bool use_halo(const Gather& g, int dtype, int ldy) {
    if (g_halo_min_blocks < 0 || !SDE_IS16(dtype) || g.KH != 3 || g.KW != 3 || g.stride != 1 || g.mode == SDE_SRC_ZEROINS) return false;
    if (g.mode == SDE_SRC_UPCAT && !g.reflect) return false;
    if (g.Cin % 8 || g.OH < HT_H || g.OW < HT_W) return false;
    if (sde_cdiv(g.Cin, 64) * 64 * 3 > g.Cin * 4) return false;     // the tile stages 64-channel blocks: narrow inputs (Cin < 48) waste MFMA and LDS
    const long tiles = (long)sde_cdiv(g.OH, HT_H) * sde_cdiv(g.OW, HT_W);
    if ((double)g.OH * g.OW < 0.75 * (double)tiles * HT_H * HT_W) return false;         // ragged tiling: the generic kernel wastes less
    if (halo_tiles(g) * sde_cdiv(ldy, halo_bn(g, ldy)) < g_halo_min_blocks) return false; // too few workgroups: prefer the 64x64 generic tiles
    return true;
}
int halo_tiles_m(const Gather& g) { return (int)halo_tiles(g); }

template <typename T, int BN, int WM, int WN, int SRC>
int launch_halo(const IGemmP& p, hipStream_t s) {
    constexpr int stage = HALO_PIX * HPIX_STRIDE + 2 * BN * KSTAGE_BYTES;
    constexpr int ctile = 128 * (BN + 4) * 4 + NTHREADS * 2 * 4;
    constexpr int lds = stage > ctile ? stage : ctile;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&halo3_kernel<T, BN, WM, WN, SRC>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    dim3 grid(halo_tiles_m(p.g) * sde_cdiv(p.ldy, BN));
    hipLaunchKernelGGL((halo3_kernel<T, BN, WM, WN, SRC>), grid, dim3(NTHREADS), lds, s, p);
    return 0;
}
template <typename T, int BN, int WM, int WN>
int dispatch_halo_src(const IGemmP& p, hipStream_t s) {
    if (p.g.mode == SDE_SRC_UPCAT) return launch_halo<T, BN, WM, WN, SRC_UPCAT_REFLECT>(p, s);
    return p.g.reflect ? launch_halo<T, BN, WM, WN, SRC_PLAIN_REFLECT>(p, s) : launch_halo<T, BN, WM, WN, SRC_PLAIN_ZERO>(p, s);
}
template <typename T>
int dispatch_halo(const IGemmP& p, hipStream_t s) {
    switch (halo_bn(p.g, p.ldy)) {
        case 64: return dispatch_halo_src<T, 64, 2, 2>(p, s);
        case 32: return dispatch_halo_src<T, 32, 4, 1>(p, s);
        default: return dispatch_halo_src<T, 16, 4, 1>(p, s);
    }
}

template <typename T>
int dispatch_igemm(const IGemmP& p, hipStream_t s) {
    switch (pick_tile(p.g.M, p.ldy, p.g.Ktot)) {
        case 128032: return dispatch_src<T, 128, 32, 4, 1>(p, s);
        case 128016: return dispatch_src<T, 128, 16, 4, 1>(p, s);
        default: return dispatch_src<T, 64, 64, 2, 2>(p, s);
    }
}

template <typename T, int BMG, int BNG, int WM, int WN, int SRC, bool ONE_TAP>
int launch_wgrad(const WGradP& p, int splits, hipStream_t s) {
    constexpr int BR = WGTraits<T>::BR;
    constexpr int lds2 = sizeof(T) == 2 ? 4 * BR * 288 : 2 * BR * (BMG * (int)sizeof(T) + 16) + 2 * BR * (BNG * (int)sizeof(T) + 16);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, BMG, BNG, WM, WN, SRC, ONE_TAP>), hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
        attr_done = true;
    }
    const int lds = p.single_buf ? lds2 / 2 : lds2;
    dim3 grid(sde_cdiv(p.Cout, BMG) * sde_cdiv(p.g.Ktot, BNG) * splits);
    hipLaunchKernelGGL((wgrad_kernel<T, BMG, BNG, WM, WN, SRC, ONE_TAP>), grid, dim3(NTHREADS), lds, s, p);
    return 0;
}

template <typename T, int BMG, int BNG, int WM, int WN>
int dispatch_wsrc(const WGradP& p, int splits, hipStream_t s) {
    constexpr int group = BNG / (NTHREADS / WGTraits<T>::BR);     // elements covered by a thread's CPTB chunks
    if (sizeof(T) == 4 || (p.g.mode == SDE_SRC_UPCAT && !p.g.reflect) || p.g.mode == SDE_SRC_ZEROINS)
        return launch_wgrad<T, BMG, BNG, WM, WN, SRC_RUNTIME, false>(p, splits, s);
    const bool ot = one_tap_ok(p.g, group);
    switch (src_kind(p.g)) {
        case SRC_1X1: return launch_wgrad<T, BMG, BNG, WM, WN, SRC_1X1, true>(p, splits, s);
        case SRC_PLAIN_ZERO: return ot ? launch_wgrad<T, BMG, BNG, WM, WN, SRC_PLAIN_ZERO, true>(p, splits, s) : launch_wgrad<T, BMG, BNG, WM, WN, SRC_PLAIN_ZERO, false>(p, splits, s);
        case SRC_PLAIN_REFLECT: return ot ? launch_wgrad<T, BMG, BNG, WM, WN, SRC_PLAIN_REFLECT, true>(p, splits, s) : launch_wgrad<T, BMG, BNG, WM, WN, SRC_PLAIN_REFLECT, false>(p, splits, s);
        default: return ot ? launch_wgrad<T, BMG, BNG, WM, WN, SRC_UPCAT_REFLECT, true>(p, splits, s) : launch_wgrad<T, BMG, BNG, WM, WN, SRC_UPCAT_REFLECT, false>(p, splits, s);
    }
}

// Weight-gradient tiles (Cout x K): 64 x 128 -- twice the tiles of 128 x 128, so half the pixel splits / fp32 slabs for the same number of
// workgroups (measured 10.03 vs 10.12 ms/step; 64 x 64 measured neutral) -- and 32 / 16 rows for narrow outputs.
int wgrad_bmg(int Cout) { return Cout > 32 ? 64 : (Cout > 16 ? 32 : 16); }

template <typename T>
int dispatch_wgrad(const WGradP& p, int splits, hipStream_t s) {
    switch (wgrad_bmg(p.Cout)) {
        case 64: return dispatch_wsrc<T, 64, 128, 1, 4>(p, splits, s);
        case 32: return dispatch_wsrc<T, 32, 128, 1, 4>(p, splits, s);
        default: return dispatch_wsrc<T, 16, 128, 1, 4>(p, splits, s);
    }
}

int fill_gather(const sde_conv_desc* d, Gather& g, const char* who) {
    const int V = SDE_IS16(d->dtype) ? 8 : 4;
    if (!(d->x0 && d->Bn > 0 && d->KH > 0 && d->KW > 0 && d->stride > 0)) { sde_set_error("%s: bad descriptor", who); return SDE_ERR_ARG; }
    if (!SDE_DTYPE_OK(d->dtype)) { sde_set_error("%s: bad dtype %d", who, d->dtype); return SDE_ERR_ARG; }
    if (d->C0 % V || d->C1 % V) { sde_set_error("%s: channel counts (%d,%d) must be multiples of %d", who, d->C0, d->C1, V); return SDE_ERR_ARG; }
    if (d->src_mode == SDE_SRC_UPCAT) {
        if (d->IH != 2 * d->H0 || d->IW != 2 * d->W0 || (d->C1 > 0 && !d->x1)) { sde_set_error("%s: upcat shape mismatch", who); return SDE_ERR_ARG; }
    } else if (d->src_mode == SDE_SRC_PLAIN) {
        if (d->IH != d->H0 || d->IW != d->W0 || d->C1 != 0) { sde_set_error("%s: plain source shape mismatch", who); return SDE_ERR_ARG; }
    } else if (d->src_mode == SDE_SRC_ZEROINS) {
        if (d->IH > 2 * d->H0 || d->IW > 2 * d->W0 || d->IH < 2 * d->H0 - 1 || d->IW < 2 * d->W0 - 1 || d->C1 != 0) { sde_set_error("%s: zero-insert shape mismatch", who); return SDE_ERR_ARG; }
    } else { sde_set_error("%s: bad src_mode %d", who, d->src_mode); return SDE_ERR_ARG; }
    if (d->reflect && (d->pad >= d->IH || d->pad >= d->IW || d->pad > 1)) { sde_set_error("%s: reflection pad must be 1 and < size", who); return SDE_ERR_ARG; }
    // every gathered coordinate must land inside [-(pad), size+pad): true by construction of OH/OW below
    const int oh_max = (d->OH - 1) * d->stride - d->pad + d->KH - 1, ow_max = (d->OW - 1) * d->stride - d->pad + d->KW - 1;
    if (d->reflect && (oh_max > d->IH || ow_max > d->IW)) { sde_set_error("%s: output extent exceeds reflected input", who); return SDE_ERR_ARG; }
    g.x0 = d->x0; g.x1 = d->x1; g.C0 = d->C0; g.C1 = d->C1; g.Cin = d->C0 + d->C1; g.H0 = d->H0; g.W0 = d->W0; g.IH = d->IH; g.IW = d->IW;
    g.mode = d->src_mode; g.KH = d->KH; g.KW = d->KW; g.stride = d->stride; g.pad = d->pad; g.reflect = d->reflect;
    g.Bn = d->Bn; g.OH = d->OH; g.OW = d->OW; g.M = d->Bn * d->OH * d->OW; g.Ktot = d->KH * d->KW * g.Cin;
    if ((long)d->Bn * d->OH * d->OW > 0x7fffffffL) { sde_set_error("%s: too many output pixels", who); return SDE_ERR_ARG; }
    return SDE_OK;
}

}  // namespace

// persistent LDS-DMA GEMM (pgemm.hip)
namespace sdeconv {
bool pgemm_applicable(const Gather& g, int dtype, int ldy);
bool pgemm_bnbwd_ok(const Gather& g, int dtype, int ldy, int Cout, int depth);
int pgemm_tile(const Gather& g, int ldy);
int pgemm_run(const IGemmP& p, int dtype, int depth, hipStream_t s);
int pgemm_stats_rows(const Gather& g, int ldy, int depth, bool bnbwd = false);
// narrow-input 3x3 layers at high resolution (conv_halo_small.hip)
bool chalo_applicable(const Gather& g, int dtype, int ldy);
int chalo_run(const IGemmP& p, int dtype, hipStream_t s);
// weight gradient on LDS-DMA for 64-channel-multiple layers (wgrad_dma.hip)
bool wgrad_dma_applicable(const Gather& g, int dtype, int Cout, int ldd);
int wgrad_dma_run(const Gather& g, int dtype, const void* dy, int Cout, int ldd, float* slab, int splits, int rows_per_split, hipStream_t s);
// halo weight gradient of the small-channel 3x3 layers (wgrad_halo.hip)
bool whalo_applicable(const Gather& g, int dtype, int Cout, int ldd);
int whalo_splits(const Gather& g, int Cout);
int whalo_run(const Gather& g, int dtype, const void* dy, int Cout, int ldd, float* slab, int splits, hipStream_t s);
extern int g_pgemm_force_tile;
}
namespace {
// Dispatcher options: the ONLY process-wide state of the convolution engine (sde_conv_set_option; every former SDE_* environment switch of
// this file either became one of these or was deleted with the code path it selected).
int g_use_pgemm = 1;        // SDE_OPT_PGEMM
int g_pgemm_depth = 3;      // SDE_OPT_PGEMM_DEPTH: ring stages (3: 48 KB -> 3 workgroups per CU; measured better than 4 on every layer)
int g_pgemm_3x3 = 0;        // SDE_OPT_PGEMM_3X3: also take the layers the LDS-halo 3x3 kernel would get
int g_splitk = 1;           // SDE_OPT_SPLITK
long g_wgrad_blocks = 256;  // SDE_OPT_WGRAD_BLOCKS
bool use_pgemm(const Gather& g, int dtype, int ldy) {
    if (!g_use_pgemm || !pgemm_applicable(g, dtype, ldy)) return false;
    return g_pgemm_3x3 || !use_halo(g, dtype, ldy);
}
}

extern "C" {

// Split-K factor of a forward / data-gradient GEMM: layers whose 64x64 tiling leaves most of the 256 CUs x 4 workgroup slots empty while
// the K loop is long (layer4 and the first decoder levels: M = 1440 ... 5760 pixels) cut K into up to 8 ranges.
static int pick_ksplit(const Gather& g, int dtype, int ldy) {
    const bool pg = use_pgemm(g, dtype, ldy);
    if (!g_splitk || ldy % 4) return 1;
    if (pg ? (pgemm_tile(g, ldy) != 64064 || g.mode == SDE_SRC_ZEROINS) : (use_halo(g, dtype, ldy) || pick_tile(g.M, ldy, g.Ktot) != 64064)) return 1;
    const long tiles = (long)sde_cdiv(g.M, 64) * sde_cdiv(ldy, 64);
    const int nk = sde_cdiv(g.Ktot, SDE_IS16(dtype) ? 64 : 32);
    // register-staged kernel (4 workgroups per CU, pipeline drained per tile): split below 384 tiles towards 768 workgroups (measured 10.12 ->
    // 9.73 ms/step, flat between 256 and 1024).  Persistent LDS-DMA kernel: only below one tile per CU (256), towards 512 -- the M = 5760
    // layers (360 tiles) measured 11-22 us unsplit against 18-28 us split three ways (profiles/r02_gemm_microbench.txt).
    const long tmax = pg ? 256 : 384, target = pg ? 512 : 768;
    // ... and between one and two tiles per CU when K is long (upconv(4,1): 360 tiles x 180 stages run as two uneven waves of workgroups): two ranges.
    // Measured (scripts/microbench_gemm.py, MBG_SPLITK = threshold): ResNet-50 upconv(4,1) 114 -> 80 us, ResNet-18's (72 stages) 50.6 -> 42.2 us; the
    // 3x3 256-channel layers (36 stages) get slower when split (23.7 -> 26.9 us), hence 48.  Option values >= 2 set the threshold (A/B).
    if (pg && tiles >= tmax && tiles < 2 * tmax && nk >= (g_splitk >= 2 ? g_splitk : 48)) return 2;
    if (tiles >= tmax || nk < 16) return 1;
    long S = sde_cdiv(target, tiles);
    if (S > 8) S = 8;
    if (S > nk / 8) S = nk / 8;
    return S < 2 ? 1 : (int)S;
}

int g_conv_small = 1;       // SDE_OPT_CONV_SMALL
static bool use_chalo(const Gather& g, int dtype, int ldy) { return g_conv_small && sdeconv::chalo_applicable(g, dtype, ldy); }

static int conv_fwd_impl(const sde_conv_desc* d, const void* w_packed, const float* bias, int act, void* y, int Cout, int ldy, float* stats,
                         float* ws, size_t ws_bytes, sde_stream_t stream) {
    SDE_CHECK_ARG(d && w_packed && y, "sde_conv_fwd: null pointer");
    IGemmP p;
    int rc = fill_gather(d, p.g, "sde_conv_fwd");
    if (rc) return rc;
    SDE_CHECK_ARG(Cout > 0 && ldy >= Cout, "sde_conv_fwd: bad Cout=%d ldy=%d", Cout, ldy);
    SDE_CHECK_ARG(act == SDE_ACT_NONE || act == SDE_ACT_ELU, "sde_conv_fwd: bad act %d", act);
    p.w = w_packed; p.bias = bias; p.y = y; p.stats = stats; p.Cout = Cout; p.ldy = ldy; p.act = act;
    p.ksplit = 1; p.ws = nullptr;
    p.no_kfull = 0;
    p.bn_y = nullptr; p.bnp = nullptr; p.bn_gb = nullptr; p.bn_out = nullptr;
    const int S = ws ? pick_ksplit(p.g, d->dtype, ldy) : 1;
    if (S > 1) {
        SDE_CHECK_ARG(ws_bytes >= (size_t)S * p.g.M * ldy * sizeof(float), "sde_conv_fwd_ws: workspace too small (%zu bytes)", ws_bytes);
        p.ksplit = S; p.ws = ws;
    }
    if (!stats && p.ksplit == 1 && use_chalo(p.g, d->dtype, ldy)) {
        sdeconv::chalo_run(p, d->dtype, (hipStream_t)stream);
    } else if (use_pgemm(p.g, d->dtype, ldy)) {
        SDE_CHECK_ARG((p.ksplit - 1) * sde_cdiv(p.g.Ktot / 64, p.ksplit) < p.g.Ktot / 64, "sde_conv_fwd: empty K split");
        // sde_conv_fwd_tiles_m does not know whether a workspace will be passed: a split layer's statistics slab has one row per 64 rows
        SDE_CHECK_ARG(p.ksplit == 1 || sdeconv::pgemm_stats_rows(p.g, ldy, g_pgemm_depth) == sde_cdiv(p.g.M, 64), "sde_conv_fwd: split-K layer with per-workgroup statistics");
        sdeconv::pgemm_run(p, d->dtype, g_pgemm_depth, (hipStream_t)stream);
    } else if (use_halo(p.g, d->dtype, ldy)) {
        if (d->dtype == SDE_BF16) dispatch_halo<bf16_t>(p, (hipStream_t)stream); else dispatch_halo<half_t>(p, (hipStream_t)stream);
    } else if (d->dtype == SDE_BF16) dispatch_igemm<bf16_t>(p, (hipStream_t)stream);
    else if (d->dtype == SDE_F16) dispatch_igemm<half_t>(p, (hipStream_t)stream);
    else dispatch_igemm<float>(p, (hipStream_t)stream);
    SDE_CHECK_LAUNCH("sde_conv_fwd");
    if (S > 1) {
        const unsigned nb = (unsigned)(sde_cdiv(p.g.M, 64) * sde_cdiv(ldy, 64));
        if (d->dtype == SDE_BF16) hipLaunchKernelGGL(splitk_finish_kernel<bf16_t>, dim3(nb), dim3(256), 0, (hipStream_t)stream, ws, S, p.g.M, ldy, Cout, bias, act, (bf16_t*)y, stats);
        else if (d->dtype == SDE_F16) hipLaunchKernelGGL(splitk_finish_kernel<half_t>, dim3(nb), dim3(256), 0, (hipStream_t)stream, ws, S, p.g.M, ldy, Cout, bias, act, (half_t*)y, stats);
        else hipLaunchKernelGGL(splitk_finish_kernel<float>, dim3(nb), dim3(256), 0, (hipStream_t)stream, ws, S, p.g.M, ldy, Cout, bias, act, (float*)y, stats);
        SDE_CHECK_LAUNCH("sde_conv_fwd/splitk_finish");
    }
    return SDE_OK;
}

int sde_conv_fwd(const sde_conv_desc* d, const void* w_packed, const float* bias, int act, void* y, int Cout, int ldy, float* stats,
                 sde_stream_t stream) {
    return conv_fwd_impl(d, w_packed, bias, act, y, Cout, ldy, stats, nullptr, 0, stream);
}

size_t sde_conv_fwd_ws_bytes(const sde_conv_desc* d, int ldy) {
    Gather g;
    if (!d || fill_gather(d, g, "sde_conv_fwd_ws_bytes") != SDE_OK) return 0;
    const int S = pick_ksplit(g, d->dtype, ldy);
    return S > 1 ? (size_t)S * g.M * ldy * sizeof(float) : 0;
}

int sde_conv_fwd_ws(const sde_conv_desc* d, const void* w_packed, const float* bias, int act, void* y, int Cout, int ldy, float* stats, void* ws,
                    size_t ws_bytes, sde_stream_t stream) {
    return conv_fwd_impl(d, w_packed, bias, act, y, Cout, ldy, stats, (float*)ws, ws_bytes, stream);
}

static int gather_of(const sde_conv_desc* d, Gather& g) { return fill_gather(d, g, "sde_conv_fwd_tiles_m"); }

int g_bnbwd_fuse = 1;       // SDE_OPT_BNBWD_FUSE

// 1 = persistent GEMM, 2 = LDS-halo 3x3 kernel, 0 = this layer's data gradient runs on a kernel without the BatchNorm-backward epilogue
static int bnbwd_kind(const Gather& g, int dtype, int ldy, int Cout) {
    if (!g_bnbwd_fuse || !SDE_IS16(dtype) || ldy != Cout || Cout % 64 || g.mode != SDE_SRC_PLAIN || g.stride != 1 || g.reflect) return 0;
    if (pick_ksplit(g, dtype, ldy) != 1) return 0;
    if (use_chalo(g, dtype, ldy)) return 0;
    if (use_pgemm(g, dtype, ldy)) return sdeconv::pgemm_bnbwd_ok(g, dtype, ldy, Cout, g_pgemm_depth) ? 1 : 0;
    if (use_halo(g, dtype, ldy)) return halo_bn(g, ldy) == 64 ? 2 : 0;
    return 0;
}

int sde_conv_dgrad_bnbwd_rows(const sde_conv_desc* d, int Cout, int ldy) {
    Gather g;
    if (!d || fill_gather(d, g, "sde_conv_dgrad_bnbwd_rows") != SDE_OK) return 0;
    const int kind = bnbwd_kind(g, d->dtype, ldy, Cout);
    if (kind == 1) return sdeconv::pgemm_stats_rows(g, ldy, g_pgemm_depth, true);
    if (kind == 2) return halo_tiles_m(g);
    return 0;
}

int sde_conv_dgrad_bnbwd(const sde_conv_desc* d, const void* w_packed, void* gm, int Cout, int ldy, const void* bn_y, const float* bnp, float* part,
                         sde_stream_t stream) {
    SDE_CHECK_ARG(d && w_packed && gm && bn_y && bnp && part, "sde_conv_dgrad_bnbwd: null pointer");
    IGemmP p;
    int rc = fill_gather(d, p.g, "sde_conv_dgrad_bnbwd");
    if (rc) return rc;
    const int kind = bnbwd_kind(p.g, d->dtype, ldy, Cout);
    SDE_CHECK_ARG(kind != 0, "sde_conv_dgrad_bnbwd: this layer has no fused form (ask sde_conv_dgrad_bnbwd_rows first)");
    p.w = w_packed; p.bias = nullptr; p.y = gm; p.stats = part; p.Cout = Cout; p.ldy = ldy; p.act = SDE_ACT_NONE;
    p.ksplit = 1; p.ws = nullptr; p.no_kfull = 0;
    p.bn_y = bn_y; p.bnp = bnp; p.bn_gb = nullptr; p.bn_out = nullptr;
    if (kind == 1) sdeconv::pgemm_run(p, d->dtype, g_pgemm_depth, (hipStream_t)stream);
    else if (d->dtype == SDE_BF16) dispatch_halo<bf16_t>(p, (hipStream_t)stream);
    else dispatch_halo<half_t>(p, (hipStream_t)stream);
    SDE_CHECK_LAUNCH("sde_conv_dgrad_bnbwd");
    return SDE_OK;
}

// The same for a BatchNorm followed by "+ identity, ReLU" whose output has a second consumer (torchvision's Bottleneck: conv1 of the next block and that
// block's skip path): gm = (g + res_grad) * [bn_out > 0], partials of (sum gm, sum gm * xhat) -- the whole bn_bwd_reduce pass of the residual BatchNorm.
// The persistent GEMM and the LDS-halo 3x3 kernel (BasicBlock's conv1) carry this form.
int sde_conv_dgrad_bnbwd_res_rows(const sde_conv_desc* d, int Cout, int ldy) {
    Gather g;
    if (!d || fill_gather(d, g, "sde_conv_dgrad_bnbwd_res_rows") != SDE_OK) return 0;
    const int kind = bnbwd_kind(g, d->dtype, ldy, Cout);
    return kind == 1 ? sdeconv::pgemm_stats_rows(g, ldy, g_pgemm_depth, true) : kind == 2 ? halo_tiles_m(g) : 0;
}

int sde_conv_dgrad_bnbwd_res(const sde_conv_desc* d, const void* w_packed, void* gm, int Cout, int ldy, const void* bn_y, const float* bnp, float* part,
                             const void* res_grad, const void* bn_out, sde_stream_t stream) {
    SDE_CHECK_ARG(d && w_packed && gm && bn_y && bnp && part && res_grad && bn_out, "sde_conv_dgrad_bnbwd_res: null pointer");
    IGemmP p;
    int rc = fill_gather(d, p.g, "sde_conv_dgrad_bnbwd_res");
    if (rc) return rc;
    const int kind = bnbwd_kind(p.g, d->dtype, ldy, Cout);
    SDE_CHECK_ARG(kind != 0, "sde_conv_dgrad_bnbwd_res: this layer has no fused form (ask sde_conv_dgrad_bnbwd_res_rows first)");
    p.w = w_packed; p.bias = nullptr; p.y = gm; p.stats = part; p.Cout = Cout; p.ldy = ldy; p.act = SDE_ACT_NONE;
    p.ksplit = 1; p.ws = nullptr; p.no_kfull = 0;
    p.bn_y = bn_y; p.bnp = bnp; p.bn_gb = res_grad; p.bn_out = bn_out;
    if (kind == 1) sdeconv::pgemm_run(p, d->dtype, g_pgemm_depth, (hipStream_t)stream);
    else if (d->dtype == SDE_BF16) dispatch_halo<bf16_t>(p, (hipStream_t)stream);
    else dispatch_halo<half_t>(p, (hipStream_t)stream);
    SDE_CHECK_LAUNCH("sde_conv_dgrad_bnbwd_res");
    return SDE_OK;
}

int sde_conv_fwd_variant(const sde_conv_desc* d, int ldy) {
    Gather g;
    if (gather_of(d, g) == SDE_OK && use_chalo(g, d->dtype, ldy)) return 5000000 + g.Cin * 1000 + ldy;            // 5<Cin><ldy>: narrow-input halo kernel (without BN statistics)
    if (gather_of(d, g) == SDE_OK && use_pgemm(g, d->dtype, ldy)) return 7000000 + pgemm_tile(g, ldy);   // 7<BM><BN>: persistent LDS-DMA GEMM
    if (gather_of(d, g) == SDE_OK && use_halo(g, d->dtype, ldy)) return 3128000 + halo_bn(g, ldy);      // 3128<BN>: LDS-halo 3x3 kernel
    return pick_tile((long)d->Bn * d->OH * d->OW, ldy, d->KH * d->KW * (d->C0 + d->C1));
}

int sde_conv_fwd_tiles_m(const sde_conv_desc* d, int ldy) {
    // number of M tiles the dispatcher will use (= rows of the BN-statistics slab)
    Gather g;
    if (gather_of(d, g) == SDE_OK && use_pgemm(g, d->dtype, ldy)) return sdeconv::pgemm_stats_rows(g, ldy, g_pgemm_depth);
    if (gather_of(d, g) == SDE_OK && use_halo(g, d->dtype, ldy)) return halo_tiles_m(g);
    const long M = (long)d->Bn * d->OH * d->OW;
    return sde_cdiv(M, pick_tile(M, ldy, d->KH * d->KW * (d->C0 + d->C1)) / 1000);
}

int g_wgrad_halo = 1;       // SDE_OPT_WGRAD_HALO
int g_wgrad_dma = 1;        // SDE_OPT_WGRAD_DMA
static bool use_whalo(const Gather& g, int dtype, int Cout, int ldd) { return g_wgrad_halo && sdeconv::whalo_applicable(g, dtype, Cout, ldd); }

int sde_conv_wgrad_splits(const sde_conv_desc* d, int Cout) {
    {
        Gather g;
        if (fill_gather(d, g, "sde_conv_wgrad_splits") == SDE_OK && use_whalo(g, d->dtype, Cout, 8)) return sdeconv::whalo_splits(g, Cout);
    }
    const long M = (long)d->Bn * d->OH * d->OW;
    const int Ktot = d->KH * d->KW * (d->C0 + d->C1);
    const long tiles = (long)sde_cdiv(Cout, wgrad_bmg(Cout)) * sde_cdiv(Ktot, 128);
    const int BR = SDE_IS16(d->dtype) ? 64 : 32;
    long tgt = g_wgrad_blocks;                  // default 256 = one workgroup per CU (sde_conv_set_option(SDE_OPT_WGRAD_BLOCKS, n))
    // a network's first layer (image input: <= 16 padded channels, huge M, a handful of output tiles) has no data gradient beside it and runs last
    // in backward with nothing else on the GPU: two workgroups per CU hide its gather latency (7x7 stem: 74 -> 51 us; the one-per-CU default is
    // the better choice only where the GEMM shares the GPU with the data-gradient chain)
    // (a fixed target, not a multiple of the option: with PackNet's 1024 the 5x5 first layer ran 2048 workgroups and took 1.9 ms instead of 0.3)
    if (d->C0 + d->C1 <= 16 && M >= 65536 && tiles <= 8) tgt = 2 * sdeconv::sde_persistent_cus();
    long want = (tgt + tiles - 1) / tiles;                  // default 256 (one workgroup per CU): measured best end to end -- every extra split is another fp32 slab through HBM
    const long max_by_rows = (M + 4 * BR - 1) / (4 * BR);   // at least 4 stages per split
    if (want > max_by_rows) want = max_by_rows;
    const long max_by_mem = (256L << 20) / ((long)Cout * Ktot * 4);   // slab <= 256 MiB
    if (want > max_by_mem) want = max_by_mem;
    if (want < 1) want = 1;
    return (int)want;
}

// The GEMM: `splits` fp32 slabs [Cout][Ktot].
static int wgrad_partial(const sde_conv_desc* d, const void* dy, int Cout, int ldd, float* slab, int splits, hipStream_t s, Gather& g) {
    WGradP p;
    int rc = fill_gather(d, p.g, "sde_conv_wgrad");
    if (rc) return rc;
    const int V = SDE_IS16(d->dtype) ? 8 : 4;
    SDE_CHECK_ARG(Cout > 0 && ldd >= Cout && ldd % V == 0, "sde_conv_wgrad: bad Cout=%d ldd=%d", Cout, ldd);
    SDE_CHECK_ARG(splits >= 1, "sde_conv_wgrad: bad splits=%d", splits);
    if (use_whalo(p.g, d->dtype, Cout, ldd)) {
        SDE_CHECK_ARG(splits == sdeconv::whalo_splits(p.g, Cout), "sde_conv_wgrad: splits=%d, this layer needs sde_conv_wgrad_splits() = %d", splits,
                      sdeconv::whalo_splits(p.g, Cout));
        sdeconv::whalo_run(p.g, d->dtype, dy, Cout, ldd, slab, splits, s);
        SDE_CHECK_LAUNCH("sde_conv_wgrad (halo)");
        g = p.g;
        return SDE_OK;
    }
    const int BR = SDE_IS16(d->dtype) ? 64 : 32;
    int rps = sde_cdiv(p.g.M, splits);
    rps = sde_cdiv(rps, BR) * BR;
    SDE_CHECK_ARG((long)rps * splits >= p.g.M, "sde_conv_wgrad: split arithmetic");
    // 1x1 layers 11-16 us against 14-21 us for the register-staged kernel, 3x3 layers 22-26 us against 27-30 us (profiles/README.md item 16)
    if (g_wgrad_dma && sdeconv::wgrad_dma_applicable(p.g, d->dtype, Cout, ldd)) {
        sdeconv::wgrad_dma_run(p.g, d->dtype, dy, Cout, ldd, slab, splits, rps, s);
        SDE_CHECK_LAUNCH("sde_conv_wgrad (LDS-DMA)");
        g = p.g;
        return SDE_OK;
    }
    p.dy = dy; p.slab = slab; p.Cout = Cout; p.ldd = ldd; p.rows_per_split = rps;
    p.single_buf = SDE_IS16(d->dtype) ? 1 : 0;      // one LDS stage buffer for the 16-bit tiles: 3 workgroups per CU, measured -0.7 % step time
    if (d->dtype == SDE_BF16) dispatch_wgrad<bf16_t>(p, splits, s);
    else if (d->dtype == SDE_F16) dispatch_wgrad<half_t>(p, splits, s);
    else dispatch_wgrad<float>(p, splits, s);
    SDE_CHECK_LAUNCH("sde_conv_wgrad");
    g = p.g;
    return SDE_OK;
}

static int launch_wreduce(const sde_wreduce_item* items_in, int n_in, hipStream_t s) {
    for (int i = 0; i < n_in; ++i) {
        const sde_wreduce_item& it = items_in[i];
        SDE_CHECK_ARG(it.slab && it.dw && it.rows >= 1 && it.rows <= 65535 && it.Cout >= 1 && it.Cout <= 65535 && it.KHW >= 1 && it.KHW <= 4096 &&
                          it.Cin_pad >= 4 && it.Cin_pad <= 32767 && it.Cin_pad % 4 == 0 && it.Cin_real >= 1 && it.Cin_real <= it.Cin_pad,
                      "sde_wgrad_reduce_batched: item %d out of range", i);
        SDE_CHECK_ARG(((uintptr_t)it.slab & 15) == 0, "sde_wgrad_reduce_batched: item %d: unaligned slab", i);
    }
    // streaming form: channels-last gradient, no channel padding, 16-byte aligned slot -> the slab rows ARE gradient rows
    auto streams = [](const sde_wreduce_item& it) {
        // (a 1x1 weight is the same memory in either order: torch reports it contiguous, so callers do not flag it channels-last -- the 36 1x1 layers
        // of ResNet-50 went through the transposing kernel until the timeline showed 0.58 ms of it per step)
        return ((it.accumulate & SDE_WREDUCE_OHWI) || it.KHW == 1) && it.Cin_pad == it.Cin_real && ((uintptr_t)it.dw & 15) == 0;
    };
    for (int pass = 0; pass < 2; ++pass) {
        sde_wreduce_item sel[WREDUCE_MAX];
        int n = 0;
        auto flush = [&]() -> int {
            if (n == 0) return SDE_OK;
            WReduceBatch batch;
            batch.n = n; batch.lds_floats = WREDUCE_LDS_FLOATS;
            int end = 0, need = 0;
            for (int i = 0; i < n; ++i) {
                const sde_wreduce_item& it = sel[i];
                end += pass == 0 ? sde_cdiv((long)it.Cout * it.KHW * it.Cin_pad / 4, 64) : it.Cout;
                const int lds = it.KHW * wreduce_cb(it.KHW, it.Cin_pad, WREDUCE_LDS_FLOATS);
                if (lds > need) need = lds;
                const int chunk = it.rows > SDE_WGRAD_FOLD_ROWS ? sde_cdiv(it.rows, SDE_WGRAD_FOLD_ROWS) : 1;
                batch.it[i] = WReduceArg{it.slab, it.dw, end, (unsigned short)it.rows, (unsigned short)(chunk | ((it.accumulate & SDE_WREDUCE_OHWI) ? 0x8000 : 0)),
                                         (unsigned short)it.Cout, (unsigned short)it.KHW, (unsigned short)it.Cin_pad,
                                         (unsigned short)(it.Cin_real | ((it.accumulate & SDE_WREDUCE_ACCUMULATE) ? 0x8000 : 0))};
            }
            for (int i = n; i < WREDUCE_MAX; ++i) batch.it[i] = WReduceArg{nullptr, nullptr, end, 0, 0, 0, 0, 0, 0};
            if (pass == 0) {
                hipLaunchKernelGGL(wgrad_sum_batched_kernel, dim3((unsigned)end), dim3(256), 0, s, batch);
            } else {
                static bool attr_done = false;
                if (!attr_done) {
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_reduce_batched_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                    attr_done = true;
                }
                hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3((unsigned)end), dim3(256), (size_t)need * sizeof(float), s, batch);
            }
            SDE_CHECK_LAUNCH("sde_wgrad_reduce_batched");
            n = 0;
            return SDE_OK;
        };
        for (int i = 0; i < n_in; ++i) {
            if (streams(items_in[i]) != (pass == 0)) continue;
            sel[n++] = items_in[i];
            if (n == WREDUCE_MAX) { const int rc = flush(); if (rc) return rc; }
        }
        const int rc = flush();
        if (rc) return rc;
    }
    return SDE_OK;
}

int sde_conv_wgrad(const sde_conv_desc* d, const void* dy, int Cout, int ldd, int Cin_real, float* slab, int splits, float* dw,
                   int accumulate, sde_stream_t stream) {
    SDE_CHECK_ARG(d && dy && slab && dw, "sde_conv_wgrad: null pointer");
    hipStream_t s = (hipStream_t)stream;
    Gather g;
    int rc = wgrad_partial(d, dy, Cout, ldd, slab, splits, s, g);
    if (rc) return rc;
    SDE_CHECK_ARG(Cin_real >= 1 && Cin_real <= g.Cin, "sde_conv_wgrad: bad Cin_real=%d", Cin_real);
    const sde_wreduce_item it = {slab, dw, splits, Cout, d->KH * d->KW, g.Cin, Cin_real, accumulate};      // accumulate: SDE_WREDUCE_* bits
    return launch_wreduce(&it, 1, s);
}

int sde_conv_wgrad_partial(const sde_conv_desc* d, const void* dy, int Cout, int ldd, float* slab, int splits, sde_stream_t stream) {
    SDE_CHECK_ARG(d && dy && slab, "sde_conv_wgrad_partial: null pointer");
    Gather g;
    return wgrad_partial(d, dy, Cout, ldd, slab, splits, (hipStream_t)stream, g);
}

int sde_wgrad_reduce_batched(const sde_wreduce_item* items, int n, sde_stream_t stream) {
    SDE_CHECK_ARG(items && n > 0, "sde_wgrad_reduce_batched: bad argument");
    return launch_wreduce(items, n, (hipStream_t)stream);
}

int sde_pack_weight(const float* w, void* out, int dtype, int Cout, int Cin, int KH, int KW, int Cin_pad, int Cout_pad, int for_dgrad,
                    sde_stream_t stream) {
    SDE_CHECK_ARG(w && out, "sde_pack_weight: null pointer");
    SDE_CHECK_ARG(Cout > 0 && Cin > 0 && KH > 0 && KW > 0 && Cin_pad >= Cin && Cout_pad >= Cout, "sde_pack_weight: bad shape");
    SDE_CHECK_ARG(SDE_DTYPE_OK(dtype), "sde_pack_weight: bad dtype");
    hipStream_t s = (hipStream_t)stream;
    const size_t total = for_dgrad ? (size_t)Cin_pad * KH * KW * Cout_pad : (size_t)Cout_pad * KH * KW * Cin_pad;
    const int nb = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
    if (for_dgrad) {
        if (dtype == SDE_BF16) hipLaunchKernelGGL(pack_w_dgrad_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, w, (bf16_t*)out, Cout, Cin, KH, KW, Cout_pad, Cin_pad);
        else if (dtype == SDE_F16) hipLaunchKernelGGL(pack_w_dgrad_kernel<half_t>, dim3(nb), dim3(256), 0, s, w, (half_t*)out, Cout, Cin, KH, KW, Cout_pad, Cin_pad);
        else hipLaunchKernelGGL(pack_w_dgrad_kernel<float>, dim3(nb), dim3(256), 0, s, w, (float*)out, Cout, Cin, KH, KW, Cout_pad, Cin_pad);
    } else {
        if (dtype == SDE_BF16) hipLaunchKernelGGL(pack_w_fwd_kernel<bf16_t>, dim3(nb), dim3(256), 0, s, w, (bf16_t*)out, Cout, Cin, KH, KW, Cin_pad, Cout_pad);
        else if (dtype == SDE_F16) hipLaunchKernelGGL(pack_w_fwd_kernel<half_t>, dim3(nb), dim3(256), 0, s, w, (half_t*)out, Cout, Cin, KH, KW, Cin_pad, Cout_pad);
        else hipLaunchKernelGGL(pack_w_fwd_kernel<float>, dim3(nb), dim3(256), 0, s, w, (float*)out, Cout, Cin, KH, KW, Cin_pad, Cout_pad);
    }
    SDE_CHECK_LAUNCH("sde_pack_weight");
    return SDE_OK;
}

int sde_conv_set_option(int key, int value) {
    if (key == SDE_OPT_WGRAD_BLOCKS) {
        SDE_CHECK_ARG(value >= 1 && value <= 65536, "sde_conv_set_option: bad workgroup target %d", value);
        const int old = (int)g_wgrad_blocks;
        g_wgrad_blocks = value;
        return old;
    }
    int* slot = key == SDE_OPT_PGEMM ? &g_use_pgemm : key == SDE_OPT_PGEMM_DEPTH ? &g_pgemm_depth : key == SDE_OPT_PGEMM_3X3 ? &g_pgemm_3x3 :
                key == SDE_OPT_PGEMM_TILE ? &sdeconv::g_pgemm_force_tile : key == SDE_OPT_SPLITK ? &g_splitk : key == SDE_OPT_WGRAD_HALO ? &g_wgrad_halo : key == SDE_OPT_CONV_SMALL ? &g_conv_small : key == SDE_OPT_WGRAD_DMA ? &g_wgrad_dma : key == SDE_OPT_BNBWD_FUSE ? &g_bnbwd_fuse : key == SDE_OPT_CU_RESERVE ? &sdeconv::g_cu_reserve : nullptr;
    SDE_CHECK_ARG(slot, "sde_conv_set_option: unknown key %d", key);
    SDE_CHECK_ARG(key != SDE_OPT_CU_RESERVE || (value >= 0 && value <= 128 && value % 8 == 0), "sde_conv_set_option: CU reserve must be a multiple of 8 in [0, 128], got %d", value);
    SDE_CHECK_ARG(key != SDE_OPT_PGEMM_DEPTH || value == 3 || value == 4, "sde_conv_set_option: ring depth must be 3 or 4");
    SDE_CHECK_ARG(key != SDE_OPT_PGEMM_TILE || value == 0 || value == 64064 || value == 128064 || value == 128128, "sde_conv_set_option: bad tile %d", value);
    const int old = *slot;
    *slot = value;
    return old;
}

int sde_conv_set_halo_min_blocks(int min_blocks) {
    const int old = g_halo_min_blocks;
    g_halo_min_blocks = min_blocks;
    return old;
}

int sde_pack_item_blocks(int Cout_pad, int Cin_pad, int KH, int KW) {
    if (Cout_pad <= 0 || Cin_pad <= 0 || KH <= 0 || KW <= 0 || (long)PACK_CO * 8 * KH * KW + PACK_CO > PACK_LDS_FLOATS) return -1;
    return sde_cdiv(Cout_pad, PACK_CO) * sde_cdiv(Cin_pad, pack_ci(KH * KW));
}

int sde_pack_weights_batched(const sde_pack_item* items_dev, int n, long total_blocks, int dtype, sde_stream_t stream) {
    SDE_CHECK_ARG(items_dev && n > 0 && total_blocks > 0 && total_blocks < 0x7fffffffL, "sde_pack_weights_batched: bad argument");
    SDE_CHECK_ARG(SDE_DTYPE_OK(dtype), "sde_pack_weights_batched: bad dtype");
    if (dtype == SDE_BF16) hipLaunchKernelGGL(pack_batched_kernel<bf16_t>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, n);
    else if (dtype == SDE_F16) hipLaunchKernelGGL(pack_batched_kernel<half_t>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, n);
    else hipLaunchKernelGGL(pack_batched_kernel<float>, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, items_dev, n);
    SDE_CHECK_LAUNCH("sde_pack_weights_batched");
    return SDE_OK;
}

}  // extern "C"
