// 3x3 stride-1 convolution of narrow inputs (8, 16 or 32 channels) at high resolution: the decoder's last two stages and its disparity
// heads (reference: depth_decoder.py:L21-53 Conv3x3 / ConvBlock, L95-110), and -- as a pad-2 "full" correlation with the flipped operand --
// the data gradients of the 16- and 32-output-channel layers.
//
// These layers are HBM-bound (32 -> 16 channels at 12 x 96 x 320: 35 MB of traffic, 3.4 GFLOP), but the generic implicit-GEMM kernel
// gathers every 16-byte piece of every tap from global memory with its own reflect / bounds computation and ran them at 4-6x their memory
// time.  Here a persistent workgroup walks 8 x 32-pixel output tiles: the (8+2) x (32+2) input halo is staged once per tile into LDS,
// pixel-major, and the nine taps are MFMA B operands read straight from it (the K index runs over (tap, channel), which IS the memory
// order of a pixel's channels, so a lane's 8 K values are one 16-byte LDS read at the tap's shifted address):
//
//   v_mfma_f32_16x16x32:  D[co][pixel] += A[co][k] * B[k][pixel],  k = 32 / Cin taps x Cin channels per step, 9 Cin / 32 (rounded up) steps
//
// The packed weights ([ldy][3][3][Cin], the layout sde_pack_weight produces) sit in LDS for the workgroup's lifetime, re-ordered so that
// every A fragment is one contiguous 1 KB read.  Output: a lane holds 4 consecutive output channels of one pixel -> 8-byte stores, 16
// pixels x 32 bytes per instruction; bias + ELU in registers.  fp32 accumulation; results differ from the generic kernel by summation order.
#include "conv_common.h"

namespace sdeconv {

typedef __attribute__((ext_vector_type(4))) float ch_f32x4;
typedef __attribute__((ext_vector_type(8))) short ch_s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 ch_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 ch_f16x8;
typedef __attribute__((ext_vector_type(2))) unsigned ch_u32x2;

constexpr int CH_TH = 8, CH_TW = 32, CH_HH = CH_TH + 2, CH_HW = CH_TW + 2, CH_THREADS = 256;

struct CHaloP {
    Gather g;
    const void* w;        // packed [ldy][9][Cin]
    const float* bias;    // [Cout] or null
    void* y;              // [M][ldy]
    int Cout, ldy, act;
    int tiles_h, tiles_w, total;
};

template <typename T16> struct ChT;
template <> struct ChT<bf16_t> {
    static __device__ __forceinline__ ch_f32x4 mma(ch_s16x8 a, ch_s16x8 b, ch_f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ch_bf16x8, a), __builtin_bit_cast(ch_bf16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        typedef __attribute__((ext_vector_type(2))) __bf16 v2;
        return __builtin_bit_cast(unsigned, v2{(__bf16)a, (__bf16)b});
    }
};
template <> struct ChT<half_t> {
    static __device__ __forceinline__ ch_f32x4 mma(ch_s16x8 a, ch_s16x8 b, ch_f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(ch_f16x8, a), __builtin_bit_cast(ch_f16x8, b), c, 0, 0, 0);
    }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        typedef __attribute__((ext_vector_type(2))) _Float16 v2;
        return __builtin_bit_cast(unsigned, v2{(_Float16)a, (_Float16)b});
    }
};

template <typename T16, int CIN, int CB>
__global__ void __launch_bounds__(CH_THREADS) chalo_kernel(const CHaloP p) {
    constexpr int TPS = 32 / CIN;                        // taps per 32-deep K step
    constexpr int STEPS = (9 + TPS - 1) / TPS;
    constexpr int PIXB = CIN * 2;                        // bytes per halo pixel
    constexpr int XCH = CIN / 8;                         // 16-byte chunks per pixel
    constexpr int XN = CH_HH * CH_HW * XCH;
    constexpr int XPT = (XN + CH_THREADS - 1) / CH_THREADS;
    constexpr int WBYTES = STEPS * CB * 1024;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sW = smem;                            // [STEPS][CB][4 k-groups][16 co][8]
    unsigned char* sX = smem + WBYTES;                   // [10 x 34 pixels][CIN]

    const Gather& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lg = lane >> 4;
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(g.x0, (long)g.Bn * g.H0 * g.W0 * g.C0 * 2L);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(g.x1 ? g.x1 : g.x0, g.x1 ? (long)g.Bn * g.IH * g.IW * g.C1 * 2L : 0);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(p.w, (long)p.ldy * 9 * CIN * 2L);
    const bool upcat = g.mode == SDE_SRC_UPCAT;

    // ---- weights -> LDS, once: chunk (s, cb, kg, co) = K elements 8 kg .. 8 kg + 7 of step s, output channel 16 cb + co
    for (int id = tid; id < STEPS * CB * 64; id += CH_THREADS) {
        const int co = id & 15, kg = (id >> 4) & 3, t2 = id >> 6, cb = t2 % CB, s = t2 / CB;
        const int tap = s * TPS + (8 * kg) / CIN, ci = (8 * kg) % CIN, row = cb * 16 + co;
        const bool ok = tap < 9 && row < p.ldy;
        *reinterpret_cast<uint4*>(sW + id * 16) = buf_load16(rsw, ok ? (unsigned)(((row * 9 + tap) * CIN + ci) * 2) : kOOB);
    }
    // ---- per-lane constants: bias of this lane's 4 output channels per block, B-operand offsets of its K group per step
    float bias4[CB][4];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int co = cb * 16 + 4 * lg + e;
            bias4[cb][e] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
        }
    int boff[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        int tap = s * TPS + (8 * lg) / CIN;
        if (tap > 8) tap = 8;                            // padding taps: the A operand is zero there, any address inside the halo will do
        const int ty = tap / 3, tx = tap - ty * 3;
        boff[s] = (ty * CH_HW + tx) * PIXB + ((8 * lg) % CIN) * 2;
    }

    uint4 rx[XPT];
    auto load_tile = [&](int t) {
        const int per_img = p.tiles_h * p.tiles_w;
        const int n = t / per_img, rem = t - n * per_img;
        const int th = rem / p.tiles_w, tw = rem - th * p.tiles_w;
        const int ih0 = th * CH_TH - g.pad, iw0 = tw * CH_TW - g.pad;
#pragma unroll
        for (int k = 0; k < XPT; ++k) {
            const int id = tid + k * CH_THREADS;
            const int hp = id / XCH, c = id - hp * XCH;
            const int hr = hp / CH_HW, hc = hp - hr * CH_HW;
            int ih = ih0 + hr, iw = iw0 + hc;
            bool ok = id < XN;
            if (g.reflect) {            // (pad 1) ragged tiles reach past the image: those outputs are never stored, any in-range pixel will do
                ih = reflect1(min(max(ih, -1), g.IH), g.IH); iw = reflect1(min(max(iw, -1), g.IW), g.IW);
            } else {
                ok = ok && (unsigned)ih < (unsigned)g.IH && (unsigned)iw < (unsigned)g.IW;
            }
            const int ch = c * 8;
            unsigned o0, o1 = kOOB;
            if (upcat) {
                o0 = ch < g.C0 ? (unsigned)((((n * g.H0 + (ih >> 1)) * g.W0 + (iw >> 1)) * g.C0 + ch) * 2) : kOOB;
                o1 = ch >= g.C0 ? (unsigned)((((n * g.IH + ih) * g.IW + iw) * g.C1 + (ch - g.C0)) * 2) : kOOB;
            } else {
                o0 = (unsigned)((((n * g.H0 + ih) * g.W0 + iw) * g.C0 + ch) * 2);
            }
            uint4 v = buf_load16(rs0, ok ? o0 : kOOB);
            if (upcat) {
                const uint4 u = buf_load16(rs1, ok ? o1 : kOOB);
                v.x |= u.x; v.y |= u.y; v.z |= u.z; v.w |= u.w;
            }
            rx[k] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int k = 0; k < XPT; ++k) {
            const int id = tid + k * CH_THREADS;
            if (id < XN) *reinterpret_cast<uint4*>(sX + id * 16) = rx[k];       // (hp, c) in order: the halo image is contiguous
        }
    };

    const int grid = gridDim.x;
    int w = blockIdx.x;
    if (w < p.total) load_tile(xcd_remap(w, p.total));
    for (; w < p.total; w += grid) {
        const int t = xcd_remap(w, p.total);
        __syncthreads();                                 // every wave is done with the previous tile (first pass: the weights are in LDS)
        store_tile();
        __syncthreads();
        if (w + grid < p.total) load_tile(xcd_remap(w + grid, p.total));

        // wave -> tile rows 2 wave, 2 wave + 1; pixel block pb = (row, 16-pixel half)
        ch_f32x4 acc[4][CB];
#pragma unroll
        for (int pb = 0; pb < 4; ++pb)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) acc[pb][cb] = ch_f32x4{0.f, 0.f, 0.f, 0.f};
        const unsigned char* xb = sX + ((2 * wave) * CH_HW + li) * PIXB;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            ch_s16x8 a[CB], b[4];
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) a[cb] = *reinterpret_cast<const ch_s16x8*>(sW + ((s * CB + cb) * 64 + lane) * 16);
#pragma unroll
            for (int pb = 0; pb < 4; ++pb) b[pb] = *reinterpret_cast<const ch_s16x8*>(xb + ((pb >> 1) * CH_HW + (pb & 1) * 16) * PIXB + boff[s]);
#pragma unroll
            for (int pb = 0; pb < 4; ++pb)
#pragma unroll
                for (int cb = 0; cb < CB; ++cb) acc[pb][cb] = ChT<T16>::mma(a[cb], b[pb], acc[pb][cb]);
        }
        // ---- epilogue: lane = (pixel li of the block, output channels 4 lg .. 4 lg + 3 of every 16-channel block)
        const int per_img = p.tiles_h * p.tiles_w;
        const int n = t / per_img, rem = t - n * per_img;
        const int th = rem / p.tiles_w, tw = rem - th * p.tiles_w;
#pragma unroll
        for (int pb = 0; pb < 4; ++pb) {
            const int oh = th * CH_TH + 2 * wave + (pb >> 1), ow = tw * CH_TW + (pb & 1) * 16 + li;
            if (oh >= g.OH || ow >= g.OW) continue;
            unsigned char* dst = (unsigned char*)p.y + ((size_t)(n * g.OH + oh) * g.OW + ow) * p.ldy * 2;
#pragma unroll
            for (int cb = 0; cb < CB; ++cb) {
                const int co = cb * 16 + 4 * lg;
                if (co >= p.ldy) continue;               // ldy % 8 == 0: a lane's four channels are all inside or all outside
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x = acc[pb][cb][e] + bias4[cb][e];
                    if (p.act == SDE_ACT_ELU) x = x > 0.f ? x : expm1f(x);
                    v[e] = x;
                }
                *reinterpret_cast<ch_u32x2*>(dst + co * 2) = ch_u32x2{ChT<T16>::pack2(v[0], v[1]), ChT<T16>::pack2(v[2], v[3])};
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
bool chalo_applicable(const Gather& g, int dtype, int ldy) {
    if (!SDE_IS16(dtype)) return false;
    if (g.KH != 3 || g.KW != 3 || g.stride != 1) return false;
    if (g.mode != SDE_SRC_PLAIN && g.mode != SDE_SRC_UPCAT) return false;
    if (g.reflect ? g.pad != 1 : (g.pad != 1 && g.pad != 2)) return false;
    if (g.mode == SDE_SRC_UPCAT && !g.reflect) return false;
    if ((g.Cin != 8 && g.Cin != 16 && g.Cin != 32) || g.C0 % 8 || ldy % 8 || ldy > 96) return false;
    if (g.IH < 2 || g.IW < 2 || g.OH != g.IH + 2 * g.pad - 2 || g.OW != g.IW + 2 * g.pad - 2) return false;
    if ((long)g.M < 64L * CH_TH * CH_TW) return false;
    if ((long)g.Bn * g.IH * g.IW * (g.C0 > g.C1 ? g.C0 : g.C1) * 2L >= 0x7fffffffL || (long)g.M * ldy * 2L >= 0x7fffffffL) return false;
    return true;
}

static int ch_cb(int ldy) { const int c = sde_cdiv(ldy, 16); return c <= 2 ? c : (c <= 4 ? 4 : 6); }

template <typename T16, int CIN, int CB>
static void ch_launch(const CHaloP& p, hipStream_t s) {
    constexpr int TPS = 32 / CIN, STEPS = (9 + TPS - 1) / TPS;
    constexpr int lds = STEPS * CB * 1024 + CH_HH * CH_HW * CIN * 2;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&chalo_kernel<T16, CIN, CB>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    int per_cu = (160 * 1024) / lds;
    per_cu = per_cu > 4 ? 4 : (per_cu < 1 ? 1 : per_cu);
    const int cap = sde_persistent_cus() * per_cu;
    const int grid = p.total < cap ? p.total : cap;
    hipLaunchKernelGGL((chalo_kernel<T16, CIN, CB>), dim3(grid), dim3(CH_THREADS), lds, s, p);
}

template <typename T16, int CIN>
static void ch_dispatch_cb(const CHaloP& p, int CB, hipStream_t s) {
    switch (CB) {
        case 1: return ch_launch<T16, CIN, 1>(p, s);
        case 2: return ch_launch<T16, CIN, 2>(p, s);
        case 4: return ch_launch<T16, CIN, 4>(p, s);
        default: return ch_launch<T16, CIN, 6>(p, s);
    }
}

template <typename T16>
static void ch_dispatch(const CHaloP& p, hipStream_t s) {
    const int CB = ch_cb(p.ldy);
    if (p.g.Cin == 8) return ch_dispatch_cb<T16, 8>(p, CB, s);
    if (p.g.Cin == 16) return ch_dispatch_cb<T16, 16>(p, CB, s);
    return ch_dispatch_cb<T16, 32>(p, CB, s);
}

int chalo_run(const IGemmP& q, int dtype, hipStream_t s) {
    CHaloP p;
    p.g = q.g; p.w = q.w; p.bias = q.bias; p.y = q.y; p.Cout = q.Cout; p.ldy = q.ldy; p.act = q.act;
    p.tiles_h = sde_cdiv(q.g.OH, CH_TH); p.tiles_w = sde_cdiv(q.g.OW, CH_TW);
    p.total = q.g.Bn * p.tiles_h * p.tiles_w;
    if (dtype == SDE_F16) ch_dispatch<half_t>(p, s); else ch_dispatch<bf16_t>(p, s);
    return 0;
}

}  // namespace sdeconv
