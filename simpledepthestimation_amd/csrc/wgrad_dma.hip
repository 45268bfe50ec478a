// Weight gradient on LDS-DMA for the layers with 64-channel-multiple inputs and outputs: every 1x1 and 3x3 convolution of the ResNet encoders
// (reference: detectron2/layers/resnet_encoder.py:L88-99) and the DepthDecoder's reflection-padded and up-sample + concat layers down to 64
// channels (detectron2/layers/depth_decoder.py:L21-53,L95-110), i.e. the autograd backward of their nn.Conv2d modules.
//
//   dW[co][tap][ci] = sum over pixels p of dY[p][co] * X[p + tap][ci]           (one 64 x 128 tile of [Cout][KH*KW*Cin] per workgroup and pixel range)
//
// wgrad_kernel (conv.hip) stages both operands through registers and 16-byte ds_write (a third of its LDS cycles are the bank conflicts of
// those stores, profiles/r02g_sup50_sq_by_kernel.csv) with two barriers per 64-pixel stage.  Here both operands are rows of 128 bytes per
// pixel -- dY[p][64 co] and X[p + tap][64 ci] -- i.e. exactly the row shape the persistent forward GEMM (pgemm.hip) moves by LDS-DMA:
//   * a stage = 64 pixels x (64 co | 64 ci of K block 0 | 64 ci of K block 1) = 24 KB, global -> LDS directly (buffer_load_dwordx4 ... lds), a ring of
//     D stages with D-1 in flight behind a counted s_waitcnt vmcnt and ONE barrier per stage; zero padding, stride and ragged pixel ranges are
//     out-of-range source offsets (the DMA writes zeros);
//   * pixels are the reduction index, so both MFMA operands are read TRANSPOSED (ds_read_b64_tr_b16); a 32-lane half reads four consecutive
//     pixel rows x 64 bytes per instruction, and the 16-byte chunk index is XORed with 4 * ((row >> 1) & 1) -- on the SOURCE address, the LDS image
//     of a DMA piece being lane-linear -- so those four rows hit four disjoint 16-bank groups (conflict-free);
//   * v_mfma_f32_32x32x16: A = dY^T (32 co x 16 pixels), B = X (16 pixels x 32 k); wave w owns k columns 32 w .. 32 w + 31 of the tile, all 64 co.
// Output: the fp32 slab [split][Cout][Ktot] of sde_conv_wgrad_partial (same tiling and split arithmetic as wgrad_kernel: a drop-in).
#include "conv_common.h"

namespace sdeconv {

typedef __attribute__((ext_vector_type(16))) float wd_f32x16;
typedef __attribute__((ext_vector_type(4))) short wd_s16x4;
typedef __attribute__((ext_vector_type(8))) short wd_s16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 wd_bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 wd_f16x8;
typedef __attribute__((address_space(3))) void wd_lds_void;

constexpr int WD_THREADS = 256, WD_PIX = 64, WD_STAGE = 3 * WD_PIX * 128;      // 24 KB: [dY 64 rows][X block 0: 64 rows][X block 1: 64 rows]
constexpr int WD_L = 6;                                                          // DMA instructions per wave and stage
#ifndef WD_DEPTH
// Ring depth (stages); WD_DEPTH - 1 stages of DMA in flight.  Measured (scripts/microbench_conv.py, A/B builds in one call): the 1x1 layers do not care
// (3 / 4 / 6: 206-299 / 192-300 / 192-302 TFLOP/s), the 3x3 layers do: 264-305 TFLOP/s at depth 3 against 168-195 at 4 or 6 -- two stages (48 KB) of rows
// in flight still find the neighbouring taps' lines in the 32 KB vector L1 / the XCD's L2 sets, three or five do not.
#define WD_DEPTH 3
#endif

struct WDmaP {
    Gather g;
    const void* dy;       // [M][ldd]
    float* slab;          // [splits][Cout][Ktot]
    int Cout, ldd, rows_per_split;
};

template <int N> __device__ __forceinline__ void wd_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wd_dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (wd_lds_void*)lds, 16, voff, 0, 0, 0);
}

template <typename T16> struct WdMma;
template <> struct WdMma<bf16_t> {
    static __device__ __forceinline__ wd_f32x16 mma(wd_s16x8 a, wd_s16x8 b, wd_f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(wd_bf16x8, a), __builtin_bit_cast(wd_bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct WdMma<half_t> {
    static __device__ __forceinline__ wd_f32x16 mma(wd_s16x8 a, wd_s16x8 b, wd_f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(wd_f16x8, a), __builtin_bit_cast(wd_f16x8, b), c, 0, 0, 0);
    }
};

// Transposed 64-bit LDS read (4 pixels x this lane's column) through inline asm: hipcc orders the builtin form behind ALL LDS-DMA still in flight
// (s_waitcnt vmcnt(0) in front of the first read of every stage, which drains the ring); the asm form is invisible to that pass, so the
// kernel waits for its own reads (one s_waitcnt lgkmcnt(0) naming every fragment) before the MFMAs consume them.
template <int OFF>
__device__ __forceinline__ wd_s16x4 wd_tr4(unsigned lds_addr) {
    wd_s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(lds_addr), "n"(OFF) : "memory");
    return v;
}
__device__ __forceinline__ unsigned wd_lds_addr(const void* p) { return (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ wd_s16x8 wd_cat(wd_s16x4 lo, wd_s16x4 hi) { return wd_s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]}; }

template <typename T16, bool ONE_BY_ONE, int D>
__global__ void __launch_bounds__(WD_THREADS) wgrad_dma_kernel(const WDmaP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [D][WD_STAGE]
    const Gather& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cblocks = g.Cin / 64;                              // 64-channel blocks per tap
    const int kblocks = g.KH * g.KW * cblocks;                   // 64-column blocks of K
    const int tiles_co = p.Cout / 64, tiles_k = (kblocks + 1) / 2;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / (tiles_co * tiles_k), rem_t = logical - split * (tiles_co * tiles_k);
    const int co0 = (rem_t % tiles_co) * 64, kb0 = (rem_t / tiles_co) * 2;       // all tiles of one pixel range run together on one XCD
    const int mbeg = split * p.rows_per_split, mend = min(g.M, mbeg + p.rows_per_split);
    const int ns = (mend - mbeg + WD_PIX - 1) / WD_PIX;

    const __amdgpu_buffer_rsrc_t rsx = make_rsrc(g.x0, (long)g.Bn * g.H0 * g.W0 * g.C0 * 2L);
    const __amdgpu_buffer_rsrc_t rsk = make_rsrc(g.x1 ? g.x1 : g.x0, g.x1 ? (long)g.Bn * g.IH * g.IW * g.C1 * 2L : 0);      // the skip tensor of an up-sample + concat source
    const __amdgpu_buffer_rsrc_t rsd = make_rsrc(p.dy, (long)g.M * p.ldd * 2L);

    // ---- DMA side.  Wave w fills rows 16 w .. 16 w + 15 of each of the three row groups; lane = (row lane >> 3 of a piece, 16-byte slot lane & 7).
    const int drow = lane >> 3, dslot = lane & 7;
    int hk[2], hw_[2], hc[2];                    // K block h of the tile: filter tap (kh, kw) and channel offset; hk < 0: past the last block
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int kb = kb0 + h;
        const int tap = kb / cblocks;
        hc[h] = (kb - tap * cblocks) * 64;
        hk[h] = kb < kblocks ? tap / g.KW : -1;
        hw_[h] = tap - (tap / g.KW) * g.KW;
    }
    int pn[2], poh[2], pow_[2];                  // output pixel of this lane's two rows in the NEXT stage to issue
    unsigned chunk[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 16 + i * 8 + drow;
        chunk[i] = (unsigned)((dslot ^ (((row >> 1) & 1) << 2)) * 16);
        const int m = mbeg + row;
        pn[i] = m / (g.OH * g.OW);
        const int r = m - pn[i] * (g.OH * g.OW);
        poh[i] = r / g.OW; pow_[i] = r - poh[i] * g.OW;
    }
    // geometry in registers (read from the kernel arguments inside the loop it costs a scalar load + wait per stage on the critical path)
    const int gs = g.stride, gp = g.pad, IH = g.IH, IW = g.IW, H0 = g.H0, W0 = g.W0, C0 = g.C0, C1 = g.C1, OW = g.OW, OH = g.OH, ldd = p.ldd;
    const bool refl = g.reflect != 0, upcat = g.mode == SDE_SRC_UPCAT;
    // up-sample + concat (decoder upconv(i,1), depth_decoder.py:L102-105): K block h lies entirely in the up-sampled x0 (channel < C0: C0 % 64 == 0) or in the skip x1
    bool hskip[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) hskip[h] = upcat && hc[h] >= C0;
    const int adv_h = WD_PIX / OW, adv_w = WD_PIX - adv_h * OW;      // one stage = 64 pixels further along the row-major pixel order
    int issued = 0;
    auto issue = [&]() {                         // stage `issued` (uniform: every wave issues the same stages)
        unsigned char* slot = smem + (issued % D) * WD_STAGE + (wave * 16) * 128;
        const int mrow0 = mbeg + issued * WD_PIX + wave * 16 + drow;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mrow0 + i * 8;
            const bool mok = m < mend;
            const unsigned od = (unsigned)((m * ldd + co0) * 2) + chunk[i];
            wd_dma16(rsd, slot + i * 1024, mok ? od : kOOB);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                unsigned off;
                if (ONE_BY_ONE) {                // 1x1, stride 1: the output pixel is the input pixel
                    const unsigned o = (unsigned)((m * C0 + hc[h]) * 2) + chunk[i];
                    off = (mok & (hk[h] >= 0)) ? o : kOOB;
                } else {                         // branch-free: the offset is always computed, the select picks it or the out-of-range value
                    int ih = poh[i] * gs - gp + hk[h], iw = pow_[i] * gs - gp + hw_[h];
                    bool ok = mok & (hk[h] >= 0);
                    if (refl) { ih = reflect1(ih, IH); iw = reflect1(iw, IW); }          // (uniform) ReflectionPad2d(1): every tap has a source pixel
                    else ok = ok & ((unsigned)ih < (unsigned)IH) & ((unsigned)iw < (unsigned)IW);
                    unsigned o;
                    if (hskip[h]) o = (unsigned)((((pn[i] * IH + ih) * IW + iw) * C1 + (hc[h] - C0)) * 2) + chunk[i];
                    else if (upcat) o = (unsigned)((((pn[i] * H0 + (ih >> 1)) * W0 + (iw >> 1)) * C0 + hc[h]) * 2) + chunk[i];
                    else o = (unsigned)((((pn[i] * H0 + ih) * W0 + iw) * C0 + hc[h]) * 2) + chunk[i];
                    off = ok ? o : kOOB;
                }
                if (hskip[h]) wd_dma16(rsk, slot + (1 + h) * WD_PIX * 128 + i * 1024, off);
                else wd_dma16(rsx, slot + (1 + h) * WD_PIX * 128 + i * 1024, off);
            }
            if (!ONE_BY_ONE) {                   // advance this row by one stage (64 pixels)
                pow_[i] += adv_w; poh[i] += adv_h;
                if (pow_[i] >= OW) { pow_[i] -= OW; ++poh[i]; }
                while (poh[i] >= OH) { poh[i] -= OH; ++pn[i]; }
            }
        }
        ++issued;
    };

    // ---- compute side: lane (column lane & 31 of a 32-wide operand block, K half lane >> 5); 16-lane group -> 16-column half of the block
    const int q = (lane & 15) >> 2, pp = lane & 3, g0 = (lane >> 4) & 1, kh8 = lane >> 5;
    // logical 16-byte chunk of this lane's 8 bytes inside a 64-byte operand block: 2 g0 + (pp >> 1); bytes (pp & 1) * 8 inside it
    unsigned aoff[2], boff;                      // byte offsets inside a stage of this lane's first read (K step 0), per 32-co block / for its k block
    {
        const int r0 = 8 * kh8 + q;              // pixel row of K value 0 of this lane's first read; rows r0 + {0..3} share (row >> 1) & 1 parity pattern
        const unsigned sw = (unsigned)(((r0 >> 1) & 1) << 2);
#pragma unroll
        for (int b = 0; b < 2; ++b) aoff[b] = (unsigned)(r0 * 128) + (((unsigned)(b * 4 + 2 * g0 + (pp >> 1)) ^ sw) << 4) + (unsigned)((pp & 1) * 8);
        const int kc = wave * 32;                // this wave's 32 k columns: X block (wave >> 1), 64-byte half (wave & 1)
        boff = (unsigned)((1 + (wave >> 1)) * WD_PIX * 128 + r0 * 128) + (((unsigned)((wave & 1) * 4 + 2 * g0 + (pp >> 1)) ^ sw) << 4) + (unsigned)((pp & 1) * 8);
        (void)kc;
    }
    wd_f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

#pragma unroll 1
    for (int i = 0; i < D - 1 && i < ns; ++i) issue();
#pragma unroll 1
    for (int s = 0; s < ns; ++s) {
        // stage s must have landed: everything this wave issued for it is older than the newest (D-2) stages
        if (issued - s - 1 >= D - 2) wd_wait_vmcnt<(D - 2) * WD_L>(); else wd_wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();            // every wave's share has landed, and every wave is done with stage s - 1
        if (issued < ns) issue();                // refill the slot of stage s - 1
        const unsigned st = wd_lds_addr(smem) + (unsigned)((s % D) * WD_STAGE);
        // all 24 reads of the stage first (16 pixels per MFMA K step: rows advance by 16 * 128 bytes, the swizzle term repeats every 4 rows;
        // the second read of a fragment is 4 rows = 512 bytes further), one wait, then the 8 MFMAs
        wd_s16x4 fb[4][2], fa[2][4][2];
        const unsigned ab = st + boff, aa0 = st + aoff[0], aa1 = st + aoff[1];
        fb[0][0] = wd_tr4<0>(ab); fb[0][1] = wd_tr4<512>(ab); fb[1][0] = wd_tr4<2048>(ab); fb[1][1] = wd_tr4<2560>(ab);
        fb[2][0] = wd_tr4<4096>(ab); fb[2][1] = wd_tr4<4608>(ab); fb[3][0] = wd_tr4<6144>(ab); fb[3][1] = wd_tr4<6656>(ab);
        fa[0][0][0] = wd_tr4<0>(aa0); fa[0][0][1] = wd_tr4<512>(aa0); fa[0][1][0] = wd_tr4<2048>(aa0); fa[0][1][1] = wd_tr4<2560>(aa0);
        fa[0][2][0] = wd_tr4<4096>(aa0); fa[0][2][1] = wd_tr4<4608>(aa0); fa[0][3][0] = wd_tr4<6144>(aa0); fa[0][3][1] = wd_tr4<6656>(aa0);
        fa[1][0][0] = wd_tr4<0>(aa1); fa[1][0][1] = wd_tr4<512>(aa1); fa[1][1][0] = wd_tr4<2048>(aa1); fa[1][1][1] = wd_tr4<2560>(aa1);
        fa[1][2][0] = wd_tr4<4096>(aa1); fa[1][2][1] = wd_tr4<4608>(aa1); fa[1][3][0] = wd_tr4<6144>(aa1); fa[1][3][1] = wd_tr4<6656>(aa1);
        // the wait names every fragment as an in/out operand, so no MFMA can be scheduled in front of it
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[2][0]), "+v"(fb[2][1]), "+v"(fb[3][0]), "+v"(fb[3][1]),
                       "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[0][1][0]), "+v"(fa[0][1][1]), "+v"(fa[0][2][0]), "+v"(fa[0][2][1]), "+v"(fa[0][3][0]),
                       "+v"(fa[0][3][1]), "+v"(fa[1][0][0]), "+v"(fa[1][0][1]), "+v"(fa[1][1][0]), "+v"(fa[1][1][1]), "+v"(fa[1][2][0]), "+v"(fa[1][2][1]),
                       "+v"(fa[1][3][0]), "+v"(fa[1][3][1])
                     :: "memory");
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const wd_s16x8 bf = wd_cat(fb[ks][0], fb[ks][1]);
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[b] = WdMma<T16>::mma(wd_cat(fa[b][ks][0], fa[b][ks][1]), bf, acc[b]);
        }
    }
    wd_wait_vmcnt<0>();

    // ---- slab[split][co][k]: lane holds column k = lane & 31, rows 8 (v / 4) + 4 (lane >> 5) + v % 4 of each 32-co block
    const int kcol = (kb0 + (wave >> 1)) * 64 + (wave & 1) * 32 + (lane & 31);
    if (kb0 + (wave >> 1) < kblocks) {
        float* out = p.slab + ((size_t)split * p.Cout + co0) * g.Ktot + kcol;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int co = b * 32 + 8 * (v >> 2) + 4 * kh8 + (v & 3);
                out[(size_t)co * g.Ktot] = acc[b][v];
            }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
bool wgrad_dma_applicable(const Gather& g, int dtype, int Cout, int ldd) {
    if (!SDE_IS16(dtype)) return false;
    if (g.mode != SDE_SRC_PLAIN && g.mode != SDE_SRC_UPCAT) return false;
    if (g.mode == SDE_SRC_UPCAT && !g.reflect) return false;
    if (g.reflect && (g.pad != 1 || g.IH < 2 || g.IW < 2)) return false;
    if (g.Cin % 64 || g.C0 % 64 || g.C0 + g.C1 != g.Cin || Cout % 64 || ldd % 8) return false;
    if (g.KH != g.KW || g.KH > 7) return false;
    if ((long)g.Bn * g.H0 * g.W0 * g.C0 * 2L >= 0x7fffffffL || (long)g.Bn * g.IH * g.IW * g.C1 * 2L >= 0x7fffffffL || (long)g.M * ldd * 2L >= 0x7fffffffL) return false;
    return true;
}

template <typename T16, bool ONE, int D>
static void wd_launch(const WDmaP& p, int grid, hipStream_t s) {
    constexpr int lds = D * WD_STAGE;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_dma_kernel<T16, ONE, D>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    hipLaunchKernelGGL((wgrad_dma_kernel<T16, ONE, D>), dim3(grid), dim3(WD_THREADS), lds, s, p);
}

// same tiling (64 x 128) and pixel-split arithmetic as wgrad_kernel: rows_per_split is a multiple of 64
int wgrad_dma_run(const Gather& g, int dtype, const void* dy, int Cout, int ldd, float* slab, int splits, int rows_per_split, hipStream_t s) {
    WDmaP p;
    p.g = g; p.dy = dy; p.slab = slab; p.Cout = Cout; p.ldd = ldd; p.rows_per_split = rows_per_split;
    const int kblocks = g.KH * g.KW * (g.Cin / 64);
    const int grid = (Cout / 64) * ((kblocks + 1) / 2) * splits;
    const bool one = g.KH == 1 && g.stride == 1 && g.pad == 0;
    if (dtype == SDE_F16) { if (one) wd_launch<half_t, true, WD_DEPTH>(p, grid, s); else wd_launch<half_t, false, WD_DEPTH>(p, grid, s); }
    else { if (one) wd_launch<bf16_t, true, WD_DEPTH>(p, grid, s); else wd_launch<bf16_t, false, WD_DEPTH>(p, grid, s); }
    return 0;
}

}  // namespace sdeconv
