// Weight gradient of the small-channel 3x3 convolutions (the decoder's high-resolution layers: 16..96 input channels, <= 32 output channels,
// up to 12 x 192 x 640 pixels; reference: depth_decoder.py:L21-53 Conv3x3 / ConvBlock, L95-110 the decoder loop).
//
//   dW[co][kh][kw][ci] = sum over pixels p of dY[p][co] * X[p + (kh-1, kw-1)][ci]
//
// These layers are HBM-bound (2.3 kFLOP per 64 bytes of traffic at 16 -> 16 channels), but the generic wgrad_kernel (conv.hip) gathers
// the im2col row of every pixel from global memory -- nine 32-byte pieces with a reflect computation each -- and spends 8-10x the
// memory time on address arithmetic and LDS stores.  Here a persistent workgroup walks 8 x 32-pixel output tiles: the (8+2) x (32+2)
// input halo and the dY tile are staged ONCE per tile into LDS, planar by 16-channel block (32 bytes per pixel per plane), and all nine
// taps are MFMA operands read from that halo at shifted addresses with the transposed LDS read (pixels are the reduction index):
//
//   v_mfma_f32_16x16x32: A = dY^T (16 co x 32 pixels), B = X shifted by the tap (32 pixels x 16 ci), one K block = one tile row.
//
// A lane's K values j = 4 rd + q (read rd, row q of the 4 x 16 block its 16-lane group addresses) stand for pixel 16 h + 8 rd + 4 g0 + q
// of the row (h = lane >> 5, g0 = (lane >> 4) & 1) for BOTH operands: any bijection works for a sum, and this one makes the eight
// pixels a 32-lane half reads per instruction contiguous (256 bytes = all 64 banks once): conflict-free at the 32-byte pixel stride.
//
// The accumulators ((9 taps x Cin/16 blocks) x Cout/16 fragments) live in registers over ALL tiles of the workgroup; the 8 waves split
// them NW ways and the tile rows 8/NW ways.  At the end the row groups are summed through LDS and the workgroup writes ONE fp32 slab
// [Cout][9 Cin] -- the slab format of sde_conv_wgrad_partial, so the batched reduction (conv.hip) is unchanged; splits = workgroups.
#include "conv_common.h"

namespace sdeconv {

typedef __attribute__((ext_vector_type(4))) float wh_f32x4;
typedef __attribute__((ext_vector_type(4))) short wh_s16x4;
typedef __attribute__((ext_vector_type(8))) short wh_s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 wh_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 wh_f16x8;

constexpr int WH_TH = 8, WH_TW = 32, WH_HH = WH_TH + 2, WH_HW = WH_TW + 2;
constexpr int WH_THREADS = 512;
constexpr int WH_XPLANE = WH_HH * WH_HW * 32 + 32;      // + 32 B: consecutive planes start 8 banks apart (the 16-byte stores of one pixel's chunks)
constexpr int WH_YPLANE = WH_TH * WH_TW * 32 + 32;

struct WHaloP {
    Gather g;
    const void* dy;       // [M][ldd]
    float* slab;          // [workgroups][Cout][9 Cin]
    int Cout, ldd;
    int tiles_h, tiles_w, total;      // 8 x 32 tiles per image, over the batch
};

template <typename T16> struct WhMma;
template <> struct WhMma<bf16_t> {
    static __device__ __forceinline__ wh_f32x4 mma(wh_s16x8 a, wh_s16x8 b, wh_f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(wh_bf16x8, a), __builtin_bit_cast(wh_bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct WhMma<half_t> {
    static __device__ __forceinline__ wh_f32x4 mma(wh_s16x8 a, wh_s16x8 b, wh_f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(wh_f16x8, a), __builtin_bit_cast(wh_f16x8, b), c, 0, 0, 0);
    }
};

// 8 K values (pixels) x this lane's column: two transposed 64-bit LDS reads 8 pixels (256 bytes) apart
__device__ __forceinline__ wh_s16x8 wh_tr_read(const unsigned char* a) {
    typedef __attribute__((address_space(3))) wh_s16x4 lds_s16x4;
    const wh_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a));
    const wh_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a + 256));
    return wh_s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

template <typename T16, int NB, int CO16, int NW>
__global__ void __launch_bounds__(WH_THREADS) whalo_kernel(const WHaloP p) {
    constexpr int PW = 8 / NW;                          // row groups: wave (nw, pw) owns tile rows pw, pw + PW, ...
    constexpr int UNITS = 9 * NB;                       // (tap, 16-channel block) pairs = 16-column groups of the slab
    constexpr int UPW = (UNITS + NW - 1) / NW;          // ... per wave
    constexpr int XCH = 2 * NB, YCH = 2 * CO16;         // 16-byte chunks per pixel
    constexpr int XN = WH_HH * WH_HW * XCH, YN = WH_TH * WH_TW * YCH;
    constexpr int XPT = (XN + WH_THREADS - 1) / WH_THREADS, YPT = (YN + WH_THREADS - 1) / WH_THREADS;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sX = smem;                           // [NB][10 x 34 pixels][16 ch]
    unsigned char* sY = smem + NB * WH_XPLANE;          // [CO16][8 x 32 pixels][16 co]

    const Gather& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nw = wave % NW, pw = wave / NW;
    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(g.x0, (long)g.Bn * g.H0 * g.W0 * g.C0 * 2L);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(g.x1 ? g.x1 : g.x0, g.x1 ? (long)g.Bn * g.IH * g.IW * g.C1 * 2L : 0);
    const __amdgpu_buffer_rsrc_t rsd = make_rsrc(p.dy, (long)g.M * p.ldd * 2L);
    const bool upcat = g.mode == SDE_SRC_UPCAT;

    uint4 rx[XPT], ry[YPT];
    auto load_tile = [&](int t) {       // tile t -> registers
        const int per_img = p.tiles_h * p.tiles_w;
        const int n = t / per_img, rem = t - n * per_img;
        const int th = rem / p.tiles_w, tw = rem - th * p.tiles_w;
        const int oh0 = th * WH_TH, ow0 = tw * WH_TW;
#pragma unroll
        for (int k = 0; k < XPT; ++k) {
            const int id = tid + k * WH_THREADS;
            const int hp = id / XCH, c = id - hp * XCH;
            const int hr = hp / WH_HW, hc = hp - hr * WH_HW;
            int ih = oh0 + hr - 1, iw = ow0 + hc - 1;
            bool ok = id < XN;
            if (g.reflect) {            // ragged tiles reach past the image: any in-range pixel will do there (its dY is zero)
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
#pragma unroll
        for (int k = 0; k < YPT; ++k) {
            const int id = tid + k * WH_THREADS;
            const int px = id / YCH, c = id - px * YCH;
            const int oh = oh0 + px / WH_TW, ow = ow0 + px % WH_TW;
            const bool ok = id < YN && oh < g.OH && ow < g.OW && c * 8 < p.ldd;
            ry[k] = buf_load16(rsd, ok ? (unsigned)((((n * g.OH + oh) * g.OW + ow) * p.ldd + c * 8) * 2) : kOOB);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int k = 0; k < XPT; ++k) {
            const int id = tid + k * WH_THREADS;
            const int hp = id / XCH, c = id - hp * XCH;
            if (id < XN) *reinterpret_cast<uint4*>(sX + (c >> 1) * WH_XPLANE + hp * 32 + (c & 1) * 16) = rx[k];
        }
#pragma unroll
        for (int k = 0; k < YPT; ++k) {
            const int id = tid + k * WH_THREADS;
            const int px = id / YCH, c = id - px * YCH;
            if (id < YN) *reinterpret_cast<uint4*>(sY + (c >> 1) * WH_YPLANE + px * 32 + (c & 1) * 16) = ry[k];
        }
    };

    // this wave's units: LDS byte offset of (plane b, tap (kh, kw)) relative to the tile-row base
    const int u0 = nw * UPW;
    int uoff[UPW];
#pragma unroll
    for (int u = 0; u < UPW; ++u) {
        const int ug = min(u0 + u, UNITS - 1);          // (the last wave's surplus units recompute the final one; never written)
        const int tap = ug / NB, b = ug - tap * NB;
        const int kh = tap / 3, kw = tap - kh * 3;
        uoff[u] = b * WH_XPLANE + (kh * WH_HW + kw) * 32;
    }
    const int lq = (lane & 15) >> 2, lpp = lane & 3, lg0 = (lane >> 4) & 1, lh = lane >> 5;
    const int lane_off = (16 * lh + 4 * lg0 + lq) * 32 + lpp * 8;

    wh_f32x4 acc[UPW][CO16];
#pragma unroll
    for (int u = 0; u < UPW; ++u)
#pragma unroll
        for (int c = 0; c < CO16; ++c) acc[u][c] = wh_f32x4{0.f, 0.f, 0.f, 0.f};

    const int grid = gridDim.x;
    int w = blockIdx.x;
    if (w < p.total) load_tile(xcd_remap(w, p.total));
    for (; w < p.total; w += grid) {
        __syncthreads();                                // every wave is done with the previous tile
        store_tile();
        __syncthreads();
        if (w + grid < p.total) load_tile(xcd_remap(w + grid, p.total));      // in flight while this tile is computed
#pragma unroll 1
        for (int r = pw; r < WH_TH; r += PW) {
            wh_s16x8 a[CO16];
#pragma unroll
            for (int c = 0; c < CO16; ++c) a[c] = wh_tr_read(sY + c * WH_YPLANE + r * (WH_TW * 32) + lane_off);
            const unsigned char* xrow = sX + r * (WH_HW * 32) + lane_off;
#pragma unroll
            for (int u = 0; u < UPW; ++u) {
                const wh_s16x8 b = wh_tr_read(xrow + uoff[u]);
#pragma unroll
                for (int c = 0; c < CO16; ++c) acc[u][c] = WhMma<T16>::mma(a[c], b, acc[u][c]);
            }
        }
    }

    // sum the PW row groups through LDS (fixed order), then one coalesced slab write
    const int Ktot = 9 * NB * 16;
    float* red = reinterpret_cast<float*>(smem);        // [CO16 * 16][Ktot]
    const int fr = lane & 15, fg = lane >> 4;
#pragma unroll 1
    for (int round = 0; round < PW; ++round) {
        __syncthreads();
        if (pw == round) {
#pragma unroll
            for (int u = 0; u < UPW; ++u) {
                if (u0 + u >= UNITS) continue;
#pragma unroll
                for (int c = 0; c < CO16; ++c)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float* dst = red + (c * 16 + fg * 4 + e) * Ktot + (u0 + u) * 16 + fr;
                        *dst = round == 0 ? acc[u][c][e] : *dst + acc[u][c][e];
                    }
            }
        }
    }
    __syncthreads();
    float* out = p.slab + (size_t)blockIdx.x * p.Cout * Ktot;
    for (int i = tid; i < p.Cout * Ktot; i += WH_THREADS) out[i] = red[i];
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
static int wh_nb(const Gather& g) { return g.Cin / 16; }
static int wh_nw(int NB, int CO16) { return NB * CO16 >= 4 ? 4 : (NB * CO16 >= 3 ? 2 : 1); }      // <= 18 units x CO16 fragments of accumulators per wave

bool whalo_applicable(const Gather& g, int dtype, int Cout, int ldd) {
    if (!SDE_IS16(dtype)) return false;
    if (g.KH != 3 || g.KW != 3 || g.stride != 1 || g.pad != 1) return false;
    if (g.mode != SDE_SRC_PLAIN && g.mode != SDE_SRC_UPCAT) return false;
    if (g.mode == SDE_SRC_UPCAT && !g.reflect) return false;
    if (g.Cin % 16 || g.C0 % 8 || g.Cin > 96 || Cout > 32 || ldd % 8) return false;
    const int NB = g.Cin / 16;
    if (NB != 1 && NB != 2 && NB != 4 && NB != 6) return false;
    if (g.IH < 2 || g.IW < 2 || g.OH != g.IH || g.OW != g.IW) return false;
    if ((long)g.M < 64L * WH_TH * WH_TW) return false;             // small layers: the generic kernel's pixel splits fill the chip better
    if ((long)g.Bn * g.IH * g.IW * (g.C0 > g.C1 ? g.C0 : g.C1) * 2L >= 0x7fffffffL || (long)g.M * ldd * 2L >= 0x7fffffffL) return false;
    return true;
}

static size_t wh_lds(int NB, int CO16) {
    const size_t tile = (size_t)NB * WH_XPLANE + (size_t)CO16 * WH_YPLANE, red = (size_t)CO16 * 16 * 9 * NB * 16 * 4;
    return tile > red ? tile : red;
}

int whalo_splits(const Gather& g, int Cout) {
    const int NB = wh_nb(g), CO16 = Cout > 16 ? 2 : 1;
    const int total = g.Bn * sde_cdiv(g.OH, WH_TH) * sde_cdiv(g.OW, WH_TW);
    const int per_cu = wh_lds(NB, CO16) * 2 <= 160 * 1024 ? 2 : 1;
    const int grid = sde_persistent_cus() * per_cu;
    return total < grid ? total : grid;
}

template <typename T16, int NB, int CO16>
static void wh_launch(const WHaloP& p, int grid, hipStream_t s) {
    constexpr int NW = NB * CO16 >= 4 ? 4 : (NB * CO16 >= 3 ? 2 : 1);
    const size_t lds = wh_lds(NB, CO16);
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&whalo_kernel<T16, NB, CO16, NW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_done = true;
    }
    hipLaunchKernelGGL((whalo_kernel<T16, NB, CO16, NW>), dim3(grid), dim3(WH_THREADS), lds, s, p);
}

template <typename T16>
static void wh_dispatch(const WHaloP& p, int NB, int CO16, int grid, hipStream_t s) {
    if (CO16 == 1) {
        switch (NB) {
            case 1: return wh_launch<T16, 1, 1>(p, grid, s);
            case 2: return wh_launch<T16, 2, 1>(p, grid, s);
            case 4: return wh_launch<T16, 4, 1>(p, grid, s);
            default: return wh_launch<T16, 6, 1>(p, grid, s);
        }
    }
    switch (NB) {
        case 1: return wh_launch<T16, 1, 2>(p, grid, s);
        case 2: return wh_launch<T16, 2, 2>(p, grid, s);
        case 4: return wh_launch<T16, 4, 2>(p, grid, s);
        default: return wh_launch<T16, 6, 2>(p, grid, s);
    }
}

// slab: [splits][Cout][9 Cin] with splits == whalo_splits(g, Cout)
int whalo_run(const Gather& g, int dtype, const void* dy, int Cout, int ldd, float* slab, int splits, hipStream_t s) {
    WHaloP p;
    p.g = g; p.dy = dy; p.slab = slab; p.Cout = Cout; p.ldd = ldd;
    p.tiles_h = sde_cdiv(g.OH, WH_TH); p.tiles_w = sde_cdiv(g.OW, WH_TW);
    p.total = g.Bn * p.tiles_h * p.tiles_w;
    if (splits != whalo_splits(g, Cout)) return SDE_ERR_ARG;
    const int NB = wh_nb(g), CO16 = Cout > 16 ? 2 : 1;
    (void)wh_nw;
    if (dtype == SDE_F16) wh_dispatch<half_t>(p, NB, CO16, splits, s); else wh_dispatch<bf16_t>(p, NB, CO16, splits, s);
    return SDE_OK;
}

}  // namespace sdeconv
