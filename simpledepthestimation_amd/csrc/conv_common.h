// Shared pieces of the convolution engine (conv.hip, pgemm.hip): gather description, XCD-aware tile order, buffer-resource helpers.
#pragma once
#include "common.h"
#include "sde_hip.h"

namespace sdeconv {

// Compute units the persistent kernels size their grids for: the device's count (256 on MI355X) minus SDE_OPT_CU_RESERVE -- a data-parallel run can leave
// some to RCCL's channel kernels, which otherwise queue behind workgroups that stay resident for a whole launch.  Multiples of 8 (one per XCD) keep the XCD-aware
// tile order; with a reserve the in-register statistics path of pgemm applies to fewer layer shapes (its divisibility condition), results are the same.
extern int g_cu_reserve;
int sde_persistent_cus();      // (pgemm.hip)

__device__ __forceinline__ int reflect1(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

// ------------------------------------------------------------------------------------------------------------------
// Shared gather description (forward input side)
// ------------------------------------------------------------------------------------------------------------------
struct Gather {
    const void* x0; const void* x1;
    int C0, C1, Cin;      // channels (elements) of source 0 / 1 and of the virtual input (C0 + C1)
    int H0, W0;           // stored spatial size of x0
    int IH, IW;           // virtual input spatial size
    int mode;             // SDE_SRC_PLAIN / SDE_SRC_UPCAT / SDE_SRC_ZEROINS
    int KH, KW, stride, pad, reflect;
    int Bn, OH, OW, M;    // output pixels M = Bn*OH*OW
    int Ktot;             // KH*KW*Cin
};


// XCD-aware work order (8 XCDs, each with a private L2; workgroups are dealt round-robin over the XCDs): remap the linear
// block id so that every XCD walks one CONTIGUOUS range of logical tiles.  Neighbouring tiles share halo rows, filter taps and
// the A rows of all N tiles, so those re-reads become hits in that XCD's L2 instead of refills from beyond it.  Bijective
// for any grid size; placement only affects speed, never results.
__device__ __forceinline__ int xcd_remap(int bid, int nb) {
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}


struct IGemmP {
    Gather g;
    const void* w;        // packed [Cout][Ktot]
    const float* bias;    // [Cout] or null
    void* y;              // [M][ldy]
    float* stats;         // [tiles_m][Cout][2] or null
    int Cout, ldy, act;
    int ksplit;           // > 1: the K loop is cut into ksplit ranges, each workgroup writes its raw fp32 tile to ws[split][M][ldy]
    float* ws;            //      and splitk_finish_kernel sums them in a fixed order and applies bias / activation / statistics
    int no_kfull;         // experiment switch (SDE_NO_KFULL): disable the scalar-offset 1x1 loader
    // BatchNorm-backward epilogue (sde_conv_dgrad_bnbwd): this GEMM is the data gradient that produces g = dL/d relu(bn(y_bn)).  The epilogue
    // masks it with the ReLU mask re-derived from y_bn and the BatchNorm parameters (fma(y, scale, shift) > 0, the expression bn_apply
    // evaluates), stores gm = mask * g, and writes the reduction of BatchNorm's backward -- (sum gm, sum gm * xhat) per channel -- into the
    // `stats` slab in place of (sum y, sum y^2): the separate bn_bwd_reduce pass over g and y_bn disappears.
    const void* bn_y;     // [M][ldy] raw convolution output the BatchNorm normalised (same layout as this GEMM's output), or null
    const float* bnp;     // [4][Cout]: mean, rstd, scale, shift
    // residual form (pgemm only): the BatchNorm is followed by "+ identity, ReLU" and its output has a second consumer whose gradient bn_gb arrives
    // separately: the tile becomes (g + bn_gb) * [bn_out > 0] (the mask read from the block output instead of re-derived).  Both null otherwise.
    const void* bn_gb;    // [M][ldy] gradient of the output's other consumer (the next block's skip path)
    const void* bn_out;   // [M][ldy] relu(bn(bn_y) + identity): the mask
};


// ---- branch-free gather: byte offsets for hardware-bounds-checked buffer loads (an out-of-range offset reads zeros), so
// padding, inserted zeros and ragged tiles cost no control flow and the compiler can keep several stages of loads in flight.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
constexpr unsigned kOOB = 0x80000000u;      // every tensor is < 2 GiB, so this offset is always out of range

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, long bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(bytes > 0x7fffffffL ? 0x7fffffffL : bytes), 0x00020000);
}
// voffset per lane + a wave-uniform scalar offset (the K position of the pipeline stage): no per-load VALU address arithmetic
__device__ __forceinline__ uint4 buf_load16s(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned off) {
    return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0));
}

// Source kinds.  bf16 kernels are specialised on the kind (straight-line loop bodies: with the kind decided at run time the
// loop breaks into ~70 basic blocks per 32 MFMAs and nothing overlaps); SRC_RUNTIME keeps one generic fp32 instantiation.
constexpr int SRC_RUNTIME = -1, SRC_PLAIN_ZERO = 0, SRC_PLAIN_REFLECT = 1, SRC_UPCAT_REFLECT = 2, SRC_ZEROINS_ZERO = 3, SRC_1X1 = 4;

}  // namespace sdeconv
