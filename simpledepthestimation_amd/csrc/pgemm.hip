// Persistent implicit-GEMM convolution for gfx950 (bf16 storage, fp32 accumulate): the forward / data-gradient GEMM of every layer whose
// input channel count is a multiple of 64 (all of ResNet-18/50 but the stem, the DepthDecoder down to 64 channels).
//
//   Y[M, Cout] = im2col(X)[M, K] * Wp[Cout, K]^T,   M = B*OH*OW pixels,  K = KH*KW*Cin
//
// Replaces the same reference calls as conv.hip (nn.Conv2d / ReflectionPad2d / F.interpolate / torch.cat of
// detectron2/layers/resnet_encoder.py:L88-99 and detectron2/layers/depth_decoder.py:L21-53,L95-110, and their autograd backward).
//
// Why a second main loop (profiles/README.md, round 1): the register-staged 64x64 kernel spends ~110 address / staging instructions
// per 8 MFMAs and drains its pipeline at every tile; the layers of this network are 25-40 us GEMMs with 1-36 K stages, so a launch is
// mostly prologue and epilogue latency.  This kernel is built the other way round:
//   * operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds): no staging registers, no ds_write, the XOR swizzle is applied
//     on the per-lane SOURCE address (the LDS image of a DMA piece is lane-linear), padding / ragged rows are out-of-range buffer
//     offsets (the DMA then writes zeros);
//   * a ring of D stages per workgroup with D-1 stages of DMA always in flight behind a counted s_waitcnt vmcnt(N) and ONE raw
//     s_barrier per stage;
//   * workgroups are PERSISTENT: each walks a strided list of (tile, K-range) work items and the ring runs on across items, so the first
//     stages of the next tile are already landing while the current tile is multiplied and written out (short-K 1x1 layers never see
//     an empty pipeline);
//   * v_mfma_f32_32x32x16_bf16 with the WEIGHTS as the row operand: a lane then holds 4 consecutive output channels of ONE pixel per
//     accumulator group, i.e. 8-byte NHWC pieces, staged through the just-consumed ring slot for 16-byte coalesced stores and the
//     per-tile BatchNorm (sum, sum^2) partials;
//   * no VGPR-destination global load anywhere (the bias comes in by DMA with a tile's first stage and initialises the accumulators),
//     so the compiler never has a reason to drain vmcnt.
#include "conv_common.h"

namespace sdeconv {

typedef __attribute__((ext_vector_type(8))) __bf16 pg_bf16x8;
typedef __attribute__((ext_vector_type(16))) float pg_f32x16;
typedef __attribute__((ext_vector_type(4))) float pg_f32x4;
typedef __attribute__((ext_vector_type(2))) float pg_f32x2;
template <int V> struct PgIC { static constexpr int value = V; };
typedef __attribute__((address_space(3))) void pg_lds_void;

constexpr int PG_THREADS = 256;
constexpr int PG_STAGE_K_BYTES = 128;      // 64 bf16 of K per stage

template <int N> __device__ __forceinline__ void pg_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ void pg_dma16(__amdgpu_buffer_rsrc_t r, unsigned char* lds, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (pg_lds_void*)lds, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void pg_dma4(__amdgpu_buffer_rsrc_t r, unsigned char* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (pg_lds_void*)lds, 4, voff, 0, 0, 0);
}

// 16-bit storage type of the operands / outputs: bf16 or IEEE half (fp16 + loss scaling, BASELINE.json configs[4]); fp32 accumulate either way
template <typename T16> struct PgType;
template <> struct PgType<bf16_t> {
    typedef pg_bf16x8 frag;
    static __device__ __forceinline__ pg_f32x16 mma(frag a, frag b, pg_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        typedef __attribute__((ext_vector_type(2))) __bf16 v2;
        v2 v = {(__bf16)a, (__bf16)b};
        return __builtin_bit_cast(unsigned, v);
    }
    static __device__ __forceinline__ pg_f32x2 unpack2(unsigned u) {      // two stored values -> fp32 (exact)
        return pg_f32x2{__builtin_bit_cast(float, u << 16), __builtin_bit_cast(float, u & 0xffff0000u)};
    }
};
template <> struct PgType<half_t> {
    typedef __attribute__((ext_vector_type(8))) _Float16 frag;
    static __device__ __forceinline__ pg_f32x16 mma(frag a, frag b, pg_f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ unsigned pack2(float a, float b) {
        typedef __attribute__((ext_vector_type(2))) _Float16 v2;
        v2 v = {(_Float16)a, (_Float16)b};
        return __builtin_bit_cast(unsigned, v);
    }
    static __device__ __forceinline__ pg_f32x2 unpack2(unsigned u) {
        typedef __attribute__((ext_vector_type(2))) _Float16 v2;
        const v2 v = __builtin_bit_cast(v2, u);
        return pg_f32x2{(float)v[0], (float)v[1]};
    }
};

// LDS stores go through inline asm: hipcc orders every LDS store it can see behind ALL LDS-DMA still in flight (s_waitcnt vmcnt(0)), which
// would drain the ring at every epilogue.  These stores only touch the slot that was just consumed; the caller waits lgkmcnt(0) itself.
typedef __attribute__((ext_vector_type(2))) unsigned pg_u32x2;
__device__ __forceinline__ unsigned pg_lds_addr(const void* p) { return (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)p; }
__device__ __forceinline__ void pg_lds_store8(unsigned addr, unsigned a, unsigned b) {
    pg_u32x2 v = {a, b};
    asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

__device__ __forceinline__ void pg_lds_store16(unsigned addr, float a, float b, float c, float d) {
    const pg_f32x4 v = {a, b, c, d};
    asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}

struct PGemmP {
    IGemmP p;
    int tiles_n, tiles_mn, total;       // work items = tiles_mn * ksplit, N tiles of one M tile adjacent
    int nk_total, nk_per;               // K stages of the whole GEMM / per split
    // SRC_ZEROINS_ZERO (data gradient of a stride-2 convolution) only: the output pixels are split into the four parity classes of
    // (row, column); class c = 2a + b owns the filter taps th = a, a+2, ... / tw = b, b+2, ... -- the only ones that meet a real (not an
    // inserted zero) gradient pixel -- so every class is a dense GEMM with K = (its taps) x C instead of all KH*KW taps at 1/4 density.
    int cls_end[4];                     // exclusive prefix sums of the classes' tile counts (tiles_m(c) * tiles_n)
    int zero_siblings;                  // 1x1 kernels: only class 0 has a tap; its tiles also write the zeros of the other three pixels of each 2x2 cell
    // BatchNorm statistics, many tiles per workgroup: when (grid / 8) % tiles_n == 0 every tile of a workgroup has the same tile_n (work w ->
    // logical base(xcd) + (w >> 3), and w advances by grid), so the workgroup keeps its (sum, sum^2) partials in registers over all its tiles and
    // writes ONE slab row at the end: grid / tiles_n rows instead of tiles_m (768 instead of 5760 for the 96x320 layers), no per-tile LDS
    // combine.  Row of workgroup bid = (bid & 7) * (grid / 8 / tiles_n) + (bid >> 3) / tiles_n (dense per tile_n; see pgemm_stats_rows).
    int stats_acc;
    // Work decode without divisions (the short-K layers spend more issue slots on per-tile bookkeeping than on MFMAs): under the same
    // condition -- one K range, (grid / 8) % tiles_n == 0 -- the next item of a workgroup is (tile_m + tm_step, same tile_n).
    int fast, tm_step;
};

// One work item: tile (tile_m, tile_n), K stages [s_begin, s_begin + nk)
struct PGWork {
    int tile_m, tile_n, split, s_begin, nk;
    int cls;                            // parity class (SRC_ZEROINS_ZERO)
};

// geometry of a parity class of the zero-insertion gather (g: virtual input = zero-inserted dz, stride 1, pad p = KH-1-pad_fwd)
struct PGClass {
    int a, b, ih0, iw0, IHc, IWc, nta, ntb, M;
};
__host__ __device__ inline PGClass pg_class(const Gather& g, int c) {
    PGClass k;
    k.a = c >> 1; k.b = c & 1;
    k.ih0 = (k.a + g.pad) & 1; k.iw0 = (k.b + g.pad) & 1;           // output rows ih with (ih - pad + th) even for th = a (mod 2)
    k.IHc = g.OH > k.ih0 ? (g.OH - k.ih0 + 1) / 2 : 0; k.IWc = g.OW > k.iw0 ? (g.OW - k.iw0 + 1) / 2 : 0;
    k.nta = g.KH > k.a ? (g.KH - k.a + 1) / 2 : 0; k.ntb = g.KW > k.b ? (g.KW - k.b + 1) / 2 : 0;
    k.M = g.Bn * k.IHc * k.IWc;
    return k;
}

template <int BM, int BN, int SRC>
__device__ __forceinline__ PGWork pg_work(const PGemmP& q, int w) {
    PGWork r;
    const int logical = xcd_remap(w, q.total);
    if (SRC == SRC_ZEROINS_ZERO) {
        int c = 0;
        while (c < 3 && logical >= q.cls_end[c]) ++c;
        const int local = logical - (c ? q.cls_end[c - 1] : 0);
        const PGClass k = pg_class(q.p.g, c);
        r.cls = c; r.split = 0; r.s_begin = 0;
        r.tile_m = local / q.tiles_n; r.tile_n = local - r.tile_m * q.tiles_n;
        r.nk = k.nta * k.ntb * (q.p.g.C0 / 64);
        return r;
    }
    r.cls = 0;
    r.split = logical / q.tiles_mn;
    const int rem = logical - r.split * q.tiles_mn;
    r.tile_m = rem / q.tiles_n;
    r.tile_n = rem - r.tile_m * q.tiles_n;
    r.s_begin = r.split * q.nk_per;
    const int e = r.s_begin + q.nk_per;
    r.nk = (e < q.nk_total ? e : q.nk_total) - r.s_begin;
    if (r.nk < 0) r.nk = 0;
    return r;
}

// the item `grid` work indices after item wk (index w + grid < total)
template <int BM, int BN, int SRC>
__device__ __forceinline__ PGWork pg_next(const PGemmP& q, const PGWork& wk, int w_next) {
    if (q.fast) { PGWork r = wk; r.tile_m += q.tm_step; return r; }
    return pg_work<BM, BN, SRC>(q, w_next);
}

__device__ __forceinline__ void pg_lds_store4(unsigned addr, unsigned a) { asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(a) : "memory"); }

// BNB: BatchNorm-backward epilogue (IGemmP::bn_y / bnp): the output tile is masked with relu'(bn(y_bn)) and the statistics slab receives
// (sum gm, sum gm * xhat) instead of (sum y, sum y^2).
// NW waves per workgroup as WGM x (NW / WGM): 4 = 2 x 2 (64x64 tiles, three workgroups per CU); 8 = 4 x 2 / 2 x 4 for the 128x64 / 128x128 tiles of the
// long-K layers, where a 64x64 tile moves 1 KB of L2 -> LDS traffic per MFMA and 128x128 half of that (round 3; the round-2 128-wide tiles kept 4 waves,
// 236 VGPRs and one workgroup per CU, and lost).
template <typename T16, int BM, int BN, int SRC, int D, bool BNB = false, int NW = 4, int WGM = 2>
__global__ void __launch_bounds__(64 * NW) pgemm_kernel(const PGemmP q) {
    typedef typename PgType<T16>::frag frag_t;
    constexpr int PG_THREADS = 64 * NW;                // (shadows the namespace constant: every loop below strides by the workgroup's own size)
    constexpr int WGN = NW / WGM;
    constexpr int WTM = BM / WGM, WTN = BN / WGN;      // WGM x WGN waves
    constexpr int FM = WTM / 32, FN = WTN / 32;        // 32x32 fragments per wave: pixels (MFMA columns) x channels (MFMA rows)
    constexpr int XP = BM / (8 * NW), WP = BN / (8 * NW);   // DMA pieces (8 rows x 128 B) per wave per stage
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && WTM % 32 == 0 && WTN % 32 == 0, "tile / wave grid");
    constexpr int L = XP + WP;                         // DMA instructions per wave per stage
    constexpr int STAGE_BYTES = (BM + BN) * PG_STAGE_K_BYTES;
    constexpr int BIAS_BYTES = BN * 4;
    constexpr int SLOT_BYTES = STAGE_BYTES;
    constexpr int BIAS_BASE = D * STAGE_BYTES;          // bias ring [D][BN] floats behind the stage ring, indexed by the workgroup's item count
    constexpr int EN = BN > 64 ? 64 : BN, NCH = BN / EN;   // the epilogue stages EN output channels at a time
    constexpr int CST = EN * 2 + 16;                   // staging-tile row stride (bytes): 16-byte aligned, conflict-free 8-byte column writes
    // staging tile + the per-tile BatchNorm reduction scratch red[PARTS][EN][2] floats behind it
    constexpr int ST = PG_THREADS > 256 ? 256 : PG_THREADS;      // threads of the statistics passes (the scratch behind the staging tile holds ST / 32 row sets)
    static_assert(BM * CST + (ST / (EN / 2)) * EN * 8 <= STAGE_BYTES, "the output staging tile and the statistics scratch must fit into one ring slot");
    static_assert(D >= 3 && (D - 2) * L <= 60, "ring depth");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];      // [D][X: BM rows | W: BN rows], [D][BN] bias

    const IGemmP& p = q.p;
    const Gather& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int grid = gridDim.x;

    const __amdgpu_buffer_rsrc_t rs0 = make_rsrc(g.x0, (long)g.Bn * g.H0 * g.W0 * g.C0 * 2L);
    const __amdgpu_buffer_rsrc_t rs1 = make_rsrc(g.x1 ? g.x1 : g.x0, g.x1 ? (long)g.Bn * g.IH * g.IW * g.C1 * 2L : 0);
    const __amdgpu_buffer_rsrc_t rsw = make_rsrc(p.w, (long)p.ldy * g.Ktot * 2L);          // packed operand has ldy (padded Cout) rows
    const __amdgpu_buffer_rsrc_t rsb = make_rsrc(p.bias ? (const void*)p.bias : p.w, p.bias ? (long)p.Cout * 4L : 0);
    const bool has_bias = p.bias != nullptr && p.ksplit == 1;

    // ---- DMA side: this lane fills 16-byte slot (lane & 7) of rows (lane >> 3) + 8 i of its wave's pieces.  The LDS image of a
    // piece is lane-linear, so the XOR swizzle of the 16-byte chunk index is applied to the SOURCE: chunk = slot ^ ((row >> 1) & 7).
    const int drow = lane >> 3, dslot = lane & 7;
    // ---- load cursor (runs D-1 stages ahead of the compute cursor, across work items)
    int lw = blockIdx.x;                 // work item being loaded
    PGWork lwk = pg_work<BM, BN, SRC>(q, lw < q.total ? lw : 0);
    if (lw >= q.total) lwk.nk = 0;
    int ls = 0;                          // next stage of that item
    int l_item = 0;                      // items started by the load cursor (bias ring slot = l_item % D)
    int l_kh = 0, l_kw = 0, l_cb = 0;    // filter tap and channel offset of that stage (wave-uniform)
    unsigned xbase[XP];                  // per row: byte offset of the pixel (tap (0,0), channel 0) + this lane's swizzled chunk, or kOOB
    int xih[XP], xiw[XP], xnb[XP];       // per row: input coordinate of tap (0,0) and image index (gather kinds with taps)
    unsigned woff[WP];                   // per weight row: byte offset of row + swizzled chunk, or kOOB
    int l_a = 0, l_b = 0, l_nta = 1, l_ntb = 1;      // SRC_ZEROINS_ZERO: parity class of the item being loaded, its tap counts

    auto setup_item = [&]() {            // per-lane addressing state of work item `lwk`
        const int m0 = lwk.tile_m * BM, n0 = lwk.tile_n * BN;
        if (SRC == SRC_ZEROINS_ZERO) {
            const PGClass k = pg_class(g, lwk.cls);
            l_a = k.a; l_b = k.b; l_nta = k.nta; l_ntb = k.ntb;
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const int row = wave * (BM / NW) + i * 8 + drow;
                const int chunk = dslot ^ ((row >> 1) & 7);
                const int m = m0 + row;
                const bool ok = m < k.M;
                const int mm = ok ? m : 0;
                const int n = mm / (k.IHc * k.IWc);
                const int rem = mm - n * (k.IHc * k.IWc);
                const int ci = rem / k.IWc, cj = rem - ci * k.IWc;
                // source pixel of tap (a + 2 ta, b + 2 tb): (ci + (ih0 - pad + a) / 2 + ta, cj + (iw0 - pad + b) / 2 + tb)  [the halves are exact]
                xih[i] = ci + ((k.ih0 - g.pad + k.a) >> 1); xiw[i] = cj + ((k.iw0 - g.pad + k.b) >> 1); xnb[i] = ok ? n : -1;
                xbase[i] = ok ? (unsigned)(((n * g.H0 + xih[i]) * g.W0 + xiw[i]) * g.C0 * 2 + chunk * 16) : kOOB;
            }
#pragma unroll
            for (int i = 0; i < WP; ++i) {
                const int row = wave * (BN / NW) + i * 8 + drow;
                const int chunk = dslot ^ ((row >> 1) & 7);
                const int n = n0 + row;
                woff[i] = n < p.ldy ? (unsigned)(n * g.Ktot * 2 + chunk * 16) : kOOB;
            }
            l_cb = 0; l_kh = 0; l_kw = 0;         // here: tap COUNTERS ta, tb inside the class
            return;
        }
        if (SRC == SRC_1X1 && g.stride == 1) {       // output pixel m IS input pixel m: no coordinates needed
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const int row = wave * (BM / NW) + i * 8 + drow;
                const int chunk = dslot ^ ((row >> 1) & 7);
                const int m = m0 + row;
                xnb[i] = 0; xih[i] = 0; xiw[i] = 0;
                xbase[i] = m < g.M ? (unsigned)(m * g.C0 * 2 + chunk * 16) : kOOB;
            }
        } else
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const int row = wave * (BM / NW) + i * 8 + drow;
            const int chunk = dslot ^ ((row >> 1) & 7);
            const int m = m0 + row;
            const bool ok = m < g.M;
            const int mm = ok ? m : 0;
            const int n = mm / (g.OH * g.OW);
            const int rem = mm - n * (g.OH * g.OW);
            const int oh = rem / g.OW, ow = rem - oh * g.OW;
            const int ih0 = oh * g.stride - g.pad, iw0 = ow * g.stride - g.pad;
            xih[i] = ih0; xiw[i] = iw0; xnb[i] = ok ? n : -1;
            if (SRC == SRC_1X1 || SRC == SRC_PLAIN_ZERO)
                xbase[i] = ok ? (unsigned)(((n * g.H0 + ih0) * g.W0 + iw0) * g.C0 * 2 + chunk * 16) : kOOB;      // may wrap for border pixels: only used when the tap is in range
            else
                xbase[i] = (unsigned)(chunk * 16);
        }
        if (!(q.fast && l_item > 0)) {           // fast decode: tile_n never changes
#pragma unroll
            for (int i = 0; i < WP; ++i) {
                const int row = wave * (BN / NW) + i * 8 + drow;
                const int chunk = dslot ^ ((row >> 1) & 7);
                const int n = n0 + row;
                woff[i] = n < p.ldy ? (unsigned)(n * g.Ktot * 2 + chunk * 16) : kOOB;
            }
        }
        if (SRC != SRC_1X1) {
            const int k = lwk.s_begin * 64;
            const int tap = k / g.Cin;
            l_cb = k - tap * g.Cin; l_kh = tap / g.KW; l_kw = tap - l_kh * g.KW;
        }
    };
    if (lwk.nk > 0) setup_item();
    if (!has_bias) {                     // the epilogue adds the ring's row unconditionally
        for (int i = tid; i < D * BN / 2; i += PG_THREADS) pg_lds_store8(pg_lds_addr(smem) + BIAS_BASE + i * 8, 0u, 0u);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    int issued = 0;                      // stages issued so far (ring slot = issued % D)
    auto issue_next = [&]() {
        if (lwk.nk == 0) return;         // no work left: nothing to issue (uniform)
        unsigned char* slot = smem + (issued % D) * SLOT_BYTES;
        unsigned char* sX = slot + (wave * (BM / NW)) * PG_STAGE_K_BYTES;
        unsigned char* sW = slot + BM * PG_STAGE_K_BYTES + (wave * (BN / NW)) * PG_STAGE_K_BYTES;
        const unsigned soff = SRC == SRC_ZEROINS_ZERO
                                  ? (unsigned)((((l_a + 2 * l_kh) * g.KW + l_b + 2 * l_kw) * g.C0 + l_cb) * 2)      // K position of tap (a + 2 ta, b + 2 tb), channel block cb
                                  : (unsigned)(lwk.s_begin + ls) * (unsigned)PG_STAGE_K_BYTES;
#pragma unroll
        for (int i = 0; i < WP; ++i) pg_dma16(rsw, sW + i * 1024, woff[i], soff);
        if (SRC == SRC_ZEROINS_ZERO) {
            const unsigned toff = (unsigned)(((l_kh * g.W0 + l_kw) * g.C0 + l_cb) * 2);
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const bool ok = (xnb[i] >= 0) & ((unsigned)(xih[i] + l_kh) < (unsigned)g.H0) & ((unsigned)(xiw[i] + l_kw) < (unsigned)g.W0);
                pg_dma16(rs0, sX + i * 1024, ok ? xbase[i] + toff : kOOB, 0);
            }
        } else if (SRC == SRC_1X1) {
#pragma unroll
            for (int i = 0; i < XP; ++i) pg_dma16(rs0, sX + i * 1024, xbase[i], soff);
        } else if (SRC == SRC_PLAIN_ZERO) {
            const unsigned toff = (unsigned)(((l_kh * g.W0 + l_kw) * g.C0 + l_cb) * 2);
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const bool ok = (xnb[i] >= 0) & ((unsigned)(xih[i] + l_kh) < (unsigned)g.IH) & ((unsigned)(xiw[i] + l_kw) < (unsigned)g.IW);
                pg_dma16(rs0, sX + i * 1024, ok ? xbase[i] + toff : kOOB, 0);
            }
        } else if (SRC == SRC_PLAIN_REFLECT) {
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const int ih = reflect1(xih[i] + l_kh, g.IH), iw = reflect1(xiw[i] + l_kw, g.IW);
                const unsigned o = (unsigned)((((xnb[i] * g.H0 + ih) * g.W0 + iw) * g.C0 + l_cb) * 2) + xbase[i];
                pg_dma16(rs0, sX + i * 1024, xnb[i] >= 0 ? o : kOOB, 0);
            }
        } else {        // SRC_UPCAT_REFLECT: channels [0, C0) from x0 at half resolution, [C0, Cin) from the skip tensor x1
            const bool first = l_cb < g.C0;       // wave-uniform (C0 % 64 == 0)
#pragma unroll
            for (int i = 0; i < XP; ++i) {
                const int ih = reflect1(xih[i] + l_kh, g.IH), iw = reflect1(xiw[i] + l_kw, g.IW);
                const unsigned o0 = (unsigned)((((xnb[i] * g.H0 + (ih >> 1)) * g.W0 + (iw >> 1)) * g.C0 + l_cb) * 2) + xbase[i];
                const unsigned o1 = (unsigned)((((xnb[i] * g.IH + ih) * g.IW + iw) * g.C1 + (l_cb - g.C0)) * 2) + xbase[i];
                if (first) pg_dma16(rs0, sX + i * 1024, xnb[i] >= 0 ? o0 : kOOB, 0);
                else pg_dma16(rs1, sX + i * 1024, xnb[i] >= 0 ? o1 : kOOB, 0);
            }
        }
        if (ls == 0 && has_bias && wave < BN / 64)        // the item's bias row rides with its first stage (4 B per lane, 64 channels per wave)
            pg_dma4(rsb, smem + BIAS_BASE + (l_item % D) * BIAS_BYTES + wave * 256, (unsigned)((lwk.tile_n * BN + wave * 64 + lane) * 4));
        ++issued;
        if (SRC == SRC_ZEROINS_ZERO) {
            l_cb += 64;
            if (l_cb == g.C0) { l_cb = 0; if (++l_kw == l_ntb) { l_kw = 0; ++l_kh; } }
        } else if (SRC != SRC_1X1) {
            l_cb += 64;
            if (l_cb == g.Cin) { l_cb = 0; if (++l_kw == g.KW) { l_kw = 0; ++l_kh; } }
        }
        if (++ls == lwk.nk) {            // next work item of this workgroup
            ls = 0;
            ++l_item;
            lw += grid;
            if (lw < q.total) { lwk = pg_next<BM, BN, SRC>(q, lwk, lw); if (lwk.nk > 0) setup_item(); }
            else lwk.nk = 0;
        }
    };

    // ---- compute side
    const int swz = (lane >> 1) & 7;
    unsigned koff[4];                    // byte offset of this lane's 16-byte chunk inside its 128-byte row, per 16-deep K step
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) koff[kk] = (unsigned)(((2 * kk + lh) ^ swz) << 4);
    const unsigned xrow0 = (unsigned)((wm * WTM + l31) * PG_STAGE_K_BYTES);
    const unsigned wrow0 = (unsigned)((BM + wn * WTN + l31) * PG_STAGE_K_BYTES);

    pg_f32x16 acc[FN][FM];
    int consumed = 0, c_item = 0;        // stages consumed / items finished by the compute side
    pg_f32x2 st1[NCH], st2[NCH];         // stats_acc: this thread's (channel pair tid % (EN/2), row set tid / (EN/2)) sums over all tiles of the workgroup
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) { st1[ch] = pg_f32x2{0.f, 0.f}; st2[ch] = pg_f32x2{0.f, 0.f}; }
    int st_n0 = 0;

#pragma unroll 1
    for (int i = 0; i < D - 1; ++i) issue_next();

    PGWork wk = pg_work<BM, BN, SRC>(q, blockIdx.x < q.total ? blockIdx.x : 0);
#pragma unroll 1
    for (int cw = blockIdx.x; cw < q.total; cw += grid, wk = pg_next<BM, BN, SRC>(q, wk, cw < q.total ? cw : 0)) {
        if (wk.nk == 0) continue;        // (cannot happen with the host's split arithmetic; the load cursor skips such items the same way)
        const int m0 = wk.tile_m * BM, n0 = wk.tile_n * BN;
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;
#pragma unroll 1
        for (int s = 0; s < wk.nk; ++s) {
            // stage `consumed` must have landed: everything this wave issued for it is older than the newest (D-2) stages
#ifdef PG_EXP_LAXWAIT       // experiment only (not safe in general): leave room for two epilogues' stores
            if (issued - consumed - 1 >= D - 2) pg_wait_vmcnt<(D - 2) * L + 4>(); else pg_wait_vmcnt<0>();
#else
            if (issued - consumed - 1 >= D - 2) pg_wait_vmcnt<(D - 2) * L>(); else pg_wait_vmcnt<0>();
#endif
            __builtin_amdgcn_s_barrier();          // every wave's share has landed, and every wave is done with stage consumed-1
            issue_next();                          // refill the slot of stage consumed-1
            const unsigned char* slot = smem + (consumed % D) * SLOT_BYTES;
            frag_t wf[4][FN], xf[4][FM];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int j = 0; j < FN; ++j) wf[kk][j] = *reinterpret_cast<const frag_t*>(slot + wrow0 + j * 32 * PG_STAGE_K_BYTES + koff[kk]);
#pragma unroll
                for (int i = 0; i < FM; ++i) xf[kk][i] = *reinterpret_cast<const frag_t*>(slot + xrow0 + i * 32 * PG_STAGE_K_BYTES + koff[kk]);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int i = 0; i < FM; ++i) acc[j][i] = PgType<T16>::mma(wf[kk][j], xf[kk][i], acc[j][i]);
            ++consumed;
        }

        // ---- epilogue.  Lane (pixel l31, half lh) holds channels 8g + 4 lh + {0..3} of each 32-channel fragment.
        if (p.ksplit > 1) {          // raw fp32 partial tile -> ws[split][m][n]; splitk_finish_kernel applies bias / activation / statistics
            float* wsp = p.ws + (size_t)wk.split * g.M * p.ldy;
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int m = m0 + wm * WTM + i * 32 + l31;
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int n = n0 + wn * WTN + j * 32 + 8 * gq + 4 * lh;
                        if (m < g.M && n < p.ldy)
                            *reinterpret_cast<float4*>(wsp + (size_t)m * p.ldy + n) =
                                make_float4(acc[j][i][4 * gq], acc[j][i][4 * gq + 1], acc[j][i][4 * gq + 2], acc[j][i][4 * gq + 3]);
                    }
            }
            ++c_item;
            continue;
        }
        unsigned char* sC = smem + ((consumed - 1) % D) * SLOT_BYTES;       // the slot just consumed: its refill is issued after the next barrier
        const unsigned sC_a = pg_lds_addr(sC);
        const float* sbias = reinterpret_cast<const float*>(smem + BIAS_BASE + (c_item % D) * BIAS_BYTES);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {                                 // 64 output channels at a time through the staging tile
            __builtin_amdgcn_s_barrier();                                  // every wave is done reading the slot / the previous chunk
            // accumulators (+ bias: the ring holds zeros when there is none) -> activation -> 16-bit -> staging tile.  The activation and the
            // ragged-Cout mask are resolved ONCE per tile (uniform), not per value.
            auto stage_values = [&](auto ELU, auto RAGGED) {
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                    if ((wn * WTN + j * 32) / EN != ch) continue;              // wave-uniform
#pragma unroll
                    for (int i = 0; i < FM; ++i) {
                        const int row = wm * WTM + i * 32 + l31;
#pragma unroll
                        for (int gq = 0; gq < 4; ++gq) {
                            const int nt = wn * WTN + j * 32 + 8 * gq + 4 * lh;    // channel inside the tile
                            const pg_f32x4 b4 = *reinterpret_cast<const pg_f32x4*>(sbias + nt);
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                float t = acc[j][i][4 * gq + e] + b4[e];
                                if (decltype(ELU)::value) t = t > 0.f ? t : expm1f(t);
                                if (decltype(RAGGED)::value) { if (n0 + nt + e >= p.Cout) t = 0.f; }       // padded output channels are exact zeros
                                v[e] = t;
                            }
                            pg_lds_store8(sC_a + row * CST + (nt - ch * EN) * 2, PgType<T16>::pack2(v[0], v[1]), PgType<T16>::pack2(v[2], v[3]));
                        }
                    }
                }
            };
            const bool ragged = n0 + BN > p.Cout;
            if (p.act == SDE_ACT_ELU) { if (ragged) stage_values(PgIC<1>{}, PgIC<1>{}); else stage_values(PgIC<1>{}, PgIC<0>{}); }
            else { if (ragged) stage_values(PgIC<0>{}, PgIC<1>{}); else stage_values(PgIC<0>{}, PgIC<0>{}); }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            // NOTE: LDS is read through ext_vector types only.  A load typed as a HIP struct vector (uint4, float4 ...) carries struct TBAA info,
            // which makes hipcc's waitcnt pass alias-check it against the LDS-DMA writes in flight and emit s_waitcnt vmcnt(0) in front of it.
            constexpr int G8 = EN / 8;                                      // 16-byte groups per staged row
            const int nc0 = n0 + ch * EN;
            if constexpr (BNB) {
                // BatchNorm-backward pass over the staged gradient tile, in the statistics pass's thread mapping (channel pair c2, row set part):
                // read g (LDS) and y_bn (global, the same [m][n] addresses as the output), mask, write gm back into the tile for the store loop
                // below, accumulate (sum gm, sum gm * xhat).  The host guarantees Cout % 64 == 0 and ldy == Cout (no ragged channels).
                constexpr int PARTS = PG_THREADS / (EN / 2);
                constexpr int RPT = BM / PARTS;                             // rows per thread
                const int c2 = tid % (EN / 2), part = tid / (EN / 2);
                const int cg = nc0 + 2 * c2;
                const int rows = (g.M - m0) < BM ? (g.M - m0) : BM;
                // Every global load of this pass goes through inline asm and is waited for by hand: a VGPR-destination load the compiler can SEE
                // makes its waitcnt pass track vector-memory events in this loop nest, and it then orders the K loop's LDS reads behind the
                // LDS-DMA in flight (s_waitcnt vmcnt(0) per stage: the ring would run empty).  In-order vmcnt: waiting for these loads also
                // waits for the (older) stages of the next tile already in flight, which the K loop would wait for next anyway.
                // A register written by an asm load is "defined" for the compiler the moment that statement ends: it may copy it (live-range splits under
                // pressure, joins of a branch) before the data has arrived.  So: no join between a load and its wait (each form below is straight-line and
                // loads its own BatchNorm parameters), and the residual form -- 24 pending registers made the allocator split -- puts loads and wait
                // into ONE statement, four rows at a time (operand limit).
                const float* bp = p.bnp + cg;
                const size_t tile_off = ((size_t)m0 * p.ldy + cg) * 2;
                const unsigned char* yb = reinterpret_cast<const unsigned char*>(p.bn_y) + tile_off;
                pg_f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
                static_assert(RPT == 8, "the waits below name eight rows");
                if (p.bn_gb == nullptr) {
                    pg_f32x2 mean, rstd, sc, sf;
                    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(mean) : "v"(bp) : "memory");
                    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(rstd) : "v"(bp + p.Cout) : "memory");
                    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(sc) : "v"(bp + 2 * p.Cout) : "memory");
                    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(sf) : "v"(bp + 3 * p.Cout) : "memory");
                    unsigned yraw[RPT];
#pragma unroll
                    for (int k = 0; k < RPT; ++k) {
                        const int r = part + k * PARTS;
                        const unsigned char* src = yb + (size_t)(r < rows ? r : 0) * p.ldy * 2;       // (rows >= 1; the value of an out-of-range row is not used)
                        asm volatile("global_load_dword %0, %1, off" : "=v"(yraw[k]) : "v"(src) : "memory");
                    }
                    asm volatile("s_waitcnt vmcnt(0)"
                                 : "+v"(mean), "+v"(rstd), "+v"(sc), "+v"(sf), "+v"(yraw[0]), "+v"(yraw[1]), "+v"(yraw[2]), "+v"(yraw[3]), "+v"(yraw[4]), "+v"(yraw[5]),
                                   "+v"(yraw[6]), "+v"(yraw[7])
                                 :: "memory");
#pragma unroll
                    for (int k = 0; k < RPT; ++k) {
                        const int r = part + k * PARTS;
                        if (r < rows) {
                            pg_f32x2 t = PgType<T16>::unpack2(*reinterpret_cast<const unsigned*>(sC + r * CST + c2 * 4));
                            const pg_f32x2 yv = PgType<T16>::unpack2(yraw[k]);
                            t[0] = fmaf(yv[0], sc[0], sf[0]) > 0.f ? t[0] : 0.f;
                            t[1] = fmaf(yv[1], sc[1], sf[1]) > 0.f ? t[1] : 0.f;
                            pg_lds_store4(sC_a + r * CST + c2 * 4, PgType<T16>::pack2(t[0], t[1]));
                            s1 += t;
                            s2 += t * ((yv - mean) * rstd);
                        }
                    }
                } else {
                    // residual form: g += the other consumer's gradient (bn_gb), mask read from the block output (bn_out)
                    const unsigned char* gbb = reinterpret_cast<const unsigned char*>(p.bn_gb) + tile_off;
                    const unsigned char* ob = reinterpret_cast<const unsigned char*>(p.bn_out) + tile_off;
                    pg_f32x2 mean, rstd;
#pragma unroll
                    for (int h = 0; h < RPT; h += 4) {
                        unsigned yr[4], gr[4], orr[4];
                        size_t ro[4];
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int r = part + (h + k) * PARTS;
                            ro[k] = (size_t)(r < rows ? r : 0) * p.ldy * 2;
                        }
                        if (h == 0)
                            asm volatile(
                                "global_load_dwordx2 %12, %26, off\n\tglobal_load_dwordx2 %13, %27, off\n\t"
                                "global_load_dword %0, %14, off\n\tglobal_load_dword %1, %15, off\n\tglobal_load_dword %2, %16, off\n\tglobal_load_dword %3, %17, off\n\t"
                                "global_load_dword %4, %18, off\n\tglobal_load_dword %5, %19, off\n\tglobal_load_dword %6, %20, off\n\tglobal_load_dword %7, %21, off\n\t"
                                "global_load_dword %8, %22, off\n\tglobal_load_dword %9, %23, off\n\tglobal_load_dword %10, %24, off\n\tglobal_load_dword %11, %25, off\n\t"
                                "s_waitcnt vmcnt(0)"
                                : "=&v"(yr[0]), "=&v"(yr[1]), "=&v"(yr[2]), "=&v"(yr[3]), "=&v"(gr[0]), "=&v"(gr[1]), "=&v"(gr[2]), "=&v"(gr[3]),
                                  "=&v"(orr[0]), "=&v"(orr[1]), "=&v"(orr[2]), "=&v"(orr[3]), "=&v"(mean), "=&v"(rstd)
                                : "v"(yb + ro[0]), "v"(yb + ro[1]), "v"(yb + ro[2]), "v"(yb + ro[3]), "v"(gbb + ro[0]), "v"(gbb + ro[1]), "v"(gbb + ro[2]), "v"(gbb + ro[3]),
                                  "v"(ob + ro[0]), "v"(ob + ro[1]), "v"(ob + ro[2]), "v"(ob + ro[3]), "v"(bp), "v"(bp + p.Cout)
                                : "memory");
                        else
                            asm volatile(
                                "global_load_dword %0, %12, off\n\tglobal_load_dword %1, %13, off\n\tglobal_load_dword %2, %14, off\n\tglobal_load_dword %3, %15, off\n\t"
                                "global_load_dword %4, %16, off\n\tglobal_load_dword %5, %17, off\n\tglobal_load_dword %6, %18, off\n\tglobal_load_dword %7, %19, off\n\t"
                                "global_load_dword %8, %20, off\n\tglobal_load_dword %9, %21, off\n\tglobal_load_dword %10, %22, off\n\tglobal_load_dword %11, %23, off\n\t"
                                "s_waitcnt vmcnt(0)"
                                : "=&v"(yr[0]), "=&v"(yr[1]), "=&v"(yr[2]), "=&v"(yr[3]), "=&v"(gr[0]), "=&v"(gr[1]), "=&v"(gr[2]), "=&v"(gr[3]),
                                  "=&v"(orr[0]), "=&v"(orr[1]), "=&v"(orr[2]), "=&v"(orr[3])
                                : "v"(yb + ro[0]), "v"(yb + ro[1]), "v"(yb + ro[2]), "v"(yb + ro[3]), "v"(gbb + ro[0]), "v"(gbb + ro[1]), "v"(gbb + ro[2]), "v"(gbb + ro[3]),
                                  "v"(ob + ro[0]), "v"(ob + ro[1]), "v"(ob + ro[2]), "v"(ob + ro[3])
                                : "memory");
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int r = part + (h + k) * PARTS;
                            if (r < rows) {
                                pg_f32x2 t = PgType<T16>::unpack2(*reinterpret_cast<const unsigned*>(sC + r * CST + c2 * 4));
                                const pg_f32x2 yv = PgType<T16>::unpack2(yr[k]), gv = PgType<T16>::unpack2(gr[k]), ov = PgType<T16>::unpack2(orr[k]);
                                // (g + g_other) rounded to the storage type BEFORE the mask and the sums, exactly as bn_bwd_reduce forms gm
                                const pg_f32x2 sum = PgType<T16>::unpack2(PgType<T16>::pack2(t[0] + gv[0], t[1] + gv[1]));
                                t[0] = ov[0] > 0.f ? sum[0] : 0.f;
                                t[1] = ov[1] > 0.f ? sum[1] : 0.f;
                                pg_lds_store4(sC_a + r * CST + c2 * 4, PgType<T16>::pack2(t[0], t[1]));
                                s1 += t;
                                s2 += t * ((yv - mean) * rstd);
                            }
                        }
                    }
                }
                if (q.stats_acc) { st1[ch] += s1; st2[ch] += s2; st_n0 = n0; }
                else pg_lds_store16(sC_a + BM * CST + (part * EN + 2 * c2) * 8, s1[0], s2[0], s1[1], s2[1]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (SRC == SRC_ZEROINS_ZERO) {       // row m of the class -> output pixel (2 ci + ih0, 2 cj + iw0)
                const PGClass k = pg_class(g, wk.cls);
#pragma unroll
                for (int id = tid; id < BM * G8; id += PG_THREADS) {
                    const int row = id / G8, c8 = id - row * G8;
                    const int m = m0 + row, n = nc0 + c8 * 8;
                    if (m >= k.M || n >= p.ldy) continue;
                    const int b_ = m / (k.IHc * k.IWc), rem = m - b_ * (k.IHc * k.IWc);
                    const int ci = rem / k.IWc, cj = rem - ci * k.IWc;
                    const int ih = 2 * ci + k.ih0, iw = 2 * cj + k.iw0;
                    unsigned char* dst = (unsigned char*)p.y + (((size_t)(b_ * g.OH + ih) * g.OW + iw) * p.ldy + n) * 2;
                    *reinterpret_cast<u32x4*>(dst) = *reinterpret_cast<const u32x4*>(sC + row * CST + c8 * 16);
                    if (q.zero_siblings) {
                        const u32x4 z = {0u, 0u, 0u, 0u};
                        const size_t dx = (size_t)p.ldy * 2, dyb = (size_t)g.OW * p.ldy * 2;
                        if (iw + 1 < g.OW) *reinterpret_cast<u32x4*>(dst + dx) = z;
                        if (ih + 1 < g.OH) {
                            *reinterpret_cast<u32x4*>(dst + dyb) = z;
                            if (iw + 1 < g.OW) *reinterpret_cast<u32x4*>(dst + dyb + dx) = z;
                        }
                    }
                }
            } else {
#pragma unroll
            for (int id = tid; id < BM * G8; id += PG_THREADS) {
                const int row = id / G8, c8 = id - row * G8;
                const int m = m0 + row, n = nc0 + c8 * 8;
#ifdef PG_EXP_NOSTORE
                if (m < 0)
#else
                if (m < g.M && n < p.ldy)                                   // ldy % 8 == 0
#endif
                    *reinterpret_cast<u32x4*>((unsigned char*)p.y + ((size_t)m * p.ldy + n) * 2) = *reinterpret_cast<const u32x4*>(sC + row * CST + c8 * 16);
            }
            }
            if constexpr (BNB) {
                constexpr int PARTS = PG_THREADS / (EN / 2);
                const float* red = reinterpret_cast<const float*>(sC + BM * CST);       // [PARTS][EN][2], written by the pass above
                if (!q.stats_acc && tid < EN && nc0 + tid < p.Cout) {
                    float a = 0.f, b = 0.f;
#pragma unroll
                    for (int z = 0; z < PARTS; ++z) {
                        const pg_f32x2 v = *reinterpret_cast<const pg_f32x2*>(red + (z * EN + tid) * 2);
                        a += v[0]; b += v[1];
                    }
                    p.stats[((size_t)wk.tile_m * p.Cout + nc0 + tid) * 2 + 0] = a;
                    p.stats[((size_t)wk.tile_m * p.Cout + nc0 + tid) * 2 + 1] = b;
                }
            } else
            if (p.stats) {
                // per-tile column sums of y and y^2 (of the rounded values) over the valid rows: PARTS interleaved row sets per column
                // thread = (channel pair tid % (EN/2), row set tid / (EN/2)): one 4-byte LDS read per row, packed fp32 math
                constexpr int PARTS = ST / (EN / 2);
                float* red = reinterpret_cast<float*>(sC + BM * CST);       // [PARTS][EN][2]
                const int c2 = tid % (EN / 2), part = tid / (EN / 2);
                const int rows = (g.M - m0) < BM ? (g.M - m0) : BM;
                pg_f32x2 s1 = {0.f, 0.f}, s2 = {0.f, 0.f};
                if (PG_THREADS == ST || tid < ST) {
#pragma unroll 4
                    for (int r = part; r < rows; r += PARTS) {
                        const pg_f32x2 t = PgType<T16>::unpack2(*reinterpret_cast<const unsigned*>(sC + r * CST + c2 * 4));
                        s1 += t; s2 += t * t;
                    }
                }
                if (q.stats_acc) { st1[ch] += s1; st2[ch] += s2; st_n0 = n0; continue; }
                // (one 16-byte store: two 8-byte asm stores fed from the halves of the packed sums were given the SAME register pair by hipcc)
                if (PG_THREADS == ST || tid < ST) pg_lds_store16(sC_a + BM * CST + (part * EN + 2 * c2) * 8, s1[0], s2[0], s1[1], s2[1]);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (tid < EN && nc0 + tid < p.Cout) {
                    float a = 0.f, b = 0.f;
#pragma unroll
                    for (int z = 0; z < PARTS; ++z) { a += red[(z * EN + tid) * 2]; b += red[(z * EN + tid) * 2 + 1]; }
                    p.stats[((size_t)wk.tile_m * p.Cout + nc0 + tid) * 2 + 0] = a;
                    p.stats[((size_t)wk.tile_m * p.Cout + nc0 + tid) * 2 + 1] = b;
                }
            }
        }
        ++c_item;
        // the next stage's barrier orders these LDS reads before the slot's refill
    }
    pg_wait_vmcnt<0>();      // nothing may be in flight into LDS when the workgroup ends
    if (p.stats && q.stats_acc) {
        constexpr int PARTS = ST / (EN / 2);
        float* red = reinterpret_cast<float*>(smem);                       // [NCH][PARTS][EN][2]
        __builtin_amdgcn_s_barrier();                                      // every wave is done with the ring
        const int c2 = tid % (EN / 2), part = tid / (EN / 2);
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) {
            if (PG_THREADS == ST || tid < ST) pg_lds_store16(pg_lds_addr(smem) + (((ch * PARTS + part) * EN) + 2 * c2) * 8, st1[ch][0], st2[ch][0], st1[ch][1], st2[ch][1]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int row = (blockIdx.x & 7) * ((grid >> 3) / q.tiles_n) + (blockIdx.x >> 3) / q.tiles_n;
        for (int id = tid; id < BN; id += PG_THREADS) {
            const int ch = id / EN, c = id - ch * EN;
            if (st_n0 + id >= p.Cout) continue;
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int z = 0; z < PARTS; ++z) { a += red[((ch * PARTS + z) * EN + c) * 2]; b += red[((ch * PARTS + z) * EN + c) * 2 + 1]; }
            p.stats[((size_t)row * p.Cout + st_n0 + id) * 2 + 0] = a;
            p.stats[((size_t)row * p.Cout + st_n0 + id) * 2 + 1] = b;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------
// Workgroups of a full launch: as many as fit per CU by LDS (stage ring + bias ring), at most 4, times 256 CUs.
static int pg_grid_max(int BM, int BN, int D) {
    const int lds = D * ((BM + BN) * PG_STAGE_K_BYTES + BN * 4);
    const int per_cu = (160 * 1024) / lds;
#ifndef PG_MAX_PER_CU
#define PG_MAX_PER_CU 4
#endif
    return sde_persistent_cus() * (per_cu < 1 ? 1 : (per_cu > PG_MAX_PER_CU ? PG_MAX_PER_CU : per_cu));
}

static bool pg_stats_acc(int tiles_n, int tiles_mn, int ksplit, int src, int BM, int BN, int D) {
    const int grid = pg_grid_max(BM, BN, D);
    return ksplit == 1 && src != SRC_ZEROINS_ZERO && tiles_mn > grid && (grid >> 3) % tiles_n == 0;
}

template <typename T16, int BM, int BN, int SRC, int D, bool BNB = false, int NW = 4, int WGM = 2>
static int pg_launch(const PGemmP& q, hipStream_t s) {
    constexpr int lds = D * ((BM + BN) * PG_STAGE_K_BYTES + BN * 4);      // stage ring + bias ring
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pgemm_kernel<T16, BM, BN, SRC, D, BNB, NW, WGM>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_done = true;
    }
    int grid = pg_grid_max(BM, BN, D);
    if (grid > q.total) grid = q.total;
    PGemmP qq = q;
    qq.fast = (q.p.ksplit == 1 && SRC != SRC_ZEROINS_ZERO && grid % 8 == 0 && (grid >> 3) % q.tiles_n == 0) ? 1 : 0;
    qq.tm_step = qq.fast ? (grid >> 3) / q.tiles_n : 0;
    hipLaunchKernelGGL((pgemm_kernel<T16, BM, BN, SRC, D, BNB, NW, WGM>), dim3(grid), dim3(64 * NW), lds, s, qq);
    return 0;
}

template <typename T16, int BM, int BN, int D, int NW = 4, int WGM = 2>
static int pg_dispatch_src(const PGemmP& q, int src, hipStream_t s) {
    switch (src) {
        case SRC_1X1: return pg_launch<T16, BM, BN, SRC_1X1, D, false, NW, WGM>(q, s);
        case SRC_PLAIN_ZERO: return pg_launch<T16, BM, BN, SRC_PLAIN_ZERO, D, false, NW, WGM>(q, s);
        case SRC_PLAIN_REFLECT: return pg_launch<T16, BM, BN, SRC_PLAIN_REFLECT, D, false, NW, WGM>(q, s);
        case SRC_ZEROINS_ZERO: return pg_launch<T16, BM, BN, SRC_ZEROINS_ZERO, D, false, NW, WGM>(q, s);
        default: return pg_launch<T16, BM, BN, SRC_UPCAT_REFLECT, D, false, NW, WGM>(q, s);
    }
}

// The layers this kernel takes: bf16, every source a multiple of 64 channels (a stage is one filter tap x 64 channels), no zero insertion.
bool pgemm_applicable(const Gather& g, int dtype, int ldy) {
    if (!SDE_IS16(dtype)) return false;
    if (g.Cin % 64 || g.C0 % 64 || (g.mode == SDE_SRC_UPCAT && !g.reflect)) return false;
    if (g.mode == SDE_SRC_ZEROINS) {          // data gradient of a stride-2 convolution: parity classes; either every class has a tap (K >= 2) or it is 1x1
        if (g.reflect || g.stride != 1 || g.KH != g.KW || g.pad < 0) return false;
        if (g.KH == 1 && g.pad != 0) return false;
    }
    if (ldy % 8) return false;
    if ((long)g.Bn * g.IH * g.IW * (g.C0 > g.C1 ? g.C0 : g.C1) * 2L >= 0x7fffffffL) return false;      // 32-bit byte offsets
    return true;
}

int pgemm_src_kind(const Gather& g) {
    if (g.mode == SDE_SRC_UPCAT) return SRC_UPCAT_REFLECT;
    if (g.mode == SDE_SRC_ZEROINS) return SRC_ZEROINS_ZERO;
    if (g.KH == 1 && g.KW == 1 && g.pad == 0 && !g.reflect) return SRC_1X1;
    return g.reflect ? SRC_PLAIN_REFLECT : SRC_PLAIN_ZERO;
}

// Tile choice (one place): BM*1000 + BN.  64x64 with a 3-stage ring (four waves, three workgroups per CU) everywhere.  Round 3 built eight-wave
// workgroups for the 128-row tiles (128x64: 4 x 2 waves, two workgroups per CU; 128x128: 2 x 4 waves, one per CU) -- the round-2 forms kept four waves,
// 236 VGPRs, and lost 5-40 %.  Stand-alone (profiles/r03l_gemm_microbench.txt, every forward / data-gradient shape of the ResNet-50 workload) 128x64 is
// 3-10 % faster wherever a launch has >= 640 of those tiles (layer1, the expanding 1x1 layers of layer2 / layer3, the decoder's full-correlation data
// gradients) and 5-10 % slower at 360 tiles (tail of the last wave of tiles), 128x128 wins on three shapes.  Inside the step a rule "128x64 from 640
// tiles" measured 6.55-6.58 against 6.58-6.60 ms (Supervised-R50), 4.34 against 4.27 (MonoDepth2-R18), 7.70 against 7.67 (MonoDepth2-R50): next to the
// weight-gradient queue the bigger workgroups lose what they win alone, so the rule is NOT applied; the instantiations stay reachable through
// SDE_OPT_PGEMM_TILE (tests/test_gpu_pgemm.py covers them).
int g_cu_reserve = 0;           // sde_conv_set_option(SDE_OPT_CU_RESERVE, n)
int sde_persistent_cus() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8) n = 256;
        cus = n;
    }
    const int left = cus - g_cu_reserve;
    return left < 8 ? 8 : left;
}
int g_pgemm_force_tile = 0;     // sde_conv_set_option(SDE_OPT_PGEMM_TILE, 64064 | 128064 | 128128 | 0 = automatic)
int pgemm_tile(const Gather& g, int ldy) {
    (void)g; (void)ldy;
    return g_pgemm_force_tile ? g_pgemm_force_tile : 64064;
}

static int pg_depth(int tile, int depth) { return tile != 64064 || depth == 3 ? 3 : 4; }      // the ring depth pgemm_run_t instantiates

// The layers whose BatchNorm-backward reduction can ride in this kernel's epilogue: 64x64 tiles, ring depth 3, one K range, 1x1 or zero-padded
// k x k stride-1 sources, full 64-channel output tiles.
bool pgemm_bnbwd_ok(const Gather& g, int dtype, int ldy, int Cout, int depth) {
    if (!pgemm_applicable(g, dtype, ldy) || depth != 3 || (g_pgemm_force_tile && g_pgemm_force_tile != 64064)) return false;      // (the fused epilogue exists for 64x64 tiles: pgemm_run takes them for such a launch whatever pgemm_tile says)
    const int src = pgemm_src_kind(g);
    return (src == SRC_1X1 || src == SRC_PLAIN_ZERO) && g.stride == 1 && Cout % 64 == 0 && ldy == Cout;
}

template <typename T16>
static int pgemm_run_t(const PGemmP& q, int tile, int src, int depth, hipStream_t s) {
    if (q.p.bn_y) return src == SRC_1X1 ? pg_launch<T16, 64, 64, SRC_1X1, 3, true>(q, s) : pg_launch<T16, 64, 64, SRC_PLAIN_ZERO, 3, true>(q, s);
    if (tile == 128128) return pg_dispatch_src<T16, 128, 128, 3, 8, 2>(q, src, s);      // eight waves, 2 x 4: one workgroup per CU
    if (tile == 128064) return pg_dispatch_src<T16, 128, 64, 3, 8, 4>(q, src, s);       // eight waves, 4 x 2: two workgroups per CU
    return depth == 3 ? pg_dispatch_src<T16, 64, 64, 3>(q, src, s) : pg_dispatch_src<T16, 64, 64, 4>(q, src, s);
}

int pgemm_run(const IGemmP& p, int dtype, int depth, hipStream_t s) {
    PGemmP q;
    q.p = p;
    const int tile = p.bn_y ? 64064 : pgemm_tile(p.g, p.ldy);
    const int BM = tile / 1000, BN = tile % 1000;
    q.tiles_n = sde_cdiv(p.ldy, BN);
    q.tiles_mn = sde_cdiv(p.g.M, BM) * q.tiles_n;
    q.total = q.tiles_mn * p.ksplit;
    q.nk_total = p.g.Ktot / 64;
    q.nk_per = sde_cdiv(q.nk_total, p.ksplit);
    const int src = pgemm_src_kind(p.g);
    q.zero_siblings = 0;
    q.stats_acc = p.stats && pg_stats_acc(q.tiles_n, q.tiles_mn, p.ksplit, src, BM, BN, pg_depth(tile, depth)) ? 1 : 0;
    for (int c = 0; c < 4; ++c) q.cls_end[c] = 0;
    if (src == SRC_ZEROINS_ZERO) {
        int end = 0;
        for (int c = 0; c < 4; ++c) {
            const PGClass k = pg_class(p.g, c);
            if (k.nta > 0 && k.ntb > 0) end += sde_cdiv(k.M, BM) * q.tiles_n;
            q.cls_end[c] = end;
        }
        q.total = end;
        q.zero_siblings = (p.g.KH == 1 && p.g.KW == 1) ? 1 : 0;
        if (q.total == 0) return 0;
    }
    return dtype == SDE_F16 ? pgemm_run_t<half_t>(q, tile, src, depth, s) : pgemm_run_t<bf16_t>(q, tile, src, depth, s);
}

// Rows of the BatchNorm-statistics slab pgemm_run writes for this layer (split-K layers: the finish kernel's, one per 64 rows).
int pgemm_stats_rows(const Gather& g, int ldy, int depth, bool bnbwd) {
    const int tile = bnbwd ? 64064 : pgemm_tile(g, ldy);
    const int BM = tile / 1000, BN = tile % 1000;
    depth = pg_depth(tile, depth);
    const int tiles_n = sde_cdiv(ldy, BN), tiles_mn = sde_cdiv(g.M, BM) * tiles_n;
    if (pg_stats_acc(tiles_n, tiles_mn, 1, pgemm_src_kind(g), BM, BN, depth)) return pg_grid_max(BM, BN, depth) / tiles_n;
    return sde_cdiv(g.M, BM);
}

}  // namespace sdeconv
