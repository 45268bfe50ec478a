// Shared helpers for the gfx950 kernels behind include/sde_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define SDE_OK 0
#define SDE_ERR_ARG (-1)     // bad argument (shape / pointer / alignment)
#define SDE_ERR_LAUNCH (-2)  // HIP launch error
#define SDE_ERR_UNSUPPORTED (-3)

void sde_set_error(const char* fmt, ...);

#define SDE_CHECK_ARG(cond, ...)            \
    do {                                    \
        if (!(cond)) {                      \
            sde_set_error(__VA_ARGS__);     \
            return SDE_ERR_ARG;             \
        }                                   \
    } while (0)

// Launch check: hipGetLastError only (never synchronises; capture-safe).
#define SDE_CHECK_LAUNCH(name)                                                   \
    do {                                                                         \
        hipError_t e__ = hipGetLastError();                                      \
        if (e__ != hipSuccess) {                                                 \
            sde_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return SDE_ERR_LAUNCH;                                               \
        }                                                                        \
    } while (0)

static inline int sde_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- device helpers -------------------------------------------------------------------------
// Sum over the 64 lanes of a wave, returned in every lane.  DPP adds instead of six __shfl_xor steps (ds_bpermute_b32: an LDS-crossbar round
// trip each): quad swaps, row half-mirror and mirror give every lane its 16-lane row sum; row_bcast15 / row_bcast31 chain the four rows into
// lane 63, which v_readlane broadcasts.  Fixed order: deterministic.
__device__ __forceinline__ float sde_wave_sum(float v) {
    auto step = [](float x, auto CTRL, auto ROWS) {
        const int moved = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(CTRL)::value, decltype(ROWS)::value, 0xf, false);
        return x + __builtin_bit_cast(float, moved);
    };
    struct C_B1 { enum { value = 0xB1 }; }; struct C_4E { enum { value = 0x4E }; }; struct C_141 { enum { value = 0x141 }; };
    struct C_140 { enum { value = 0x140 }; }; struct C_142 { enum { value = 0x142 }; }; struct C_143 { enum { value = 0x143 }; };
    struct R_F { enum { value = 0xf }; }; struct R_A { enum { value = 0xa }; }; struct R_C { enum { value = 0xc }; };
    v = step(v, C_B1{}, R_F{});      // quad_perm [1,0,3,2]
    v = step(v, C_4E{}, R_F{});      // quad_perm [2,3,0,1]
    v = step(v, C_141{}, R_F{});     // row_half_mirror
    v = step(v, C_140{}, R_F{});     // row_mirror: every lane holds its row's sum
    v = step(v, C_142{}, R_A{});     // row_bcast15 into rows 1 and 3
    v = step(v, C_143{}, R_C{});     // row_bcast31 into rows 2 and 3: lane 63 holds the wave's sum
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// Sum over a block (blockDim multiple of 64, <= 1024). Result valid in thread 0. `red` >= 16 floats.
__device__ __forceinline__ float sde_block_sum(float v, float* red) {
    v = sde_wave_sum(v);
    const int tid = threadIdx.x + blockDim.x * (threadIdx.y + blockDim.y * threadIdx.z);
    const int nw = (blockDim.x * blockDim.y * blockDim.z + 63) >> 6;
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    float r = 0.f;
    if (tid == 0)
        for (int i = 0; i < nw; ++i) r += red[i];
    return r;
}

// bf16 <-> f32 (round-to-nearest-even via the compiler's cast; keeps NaN a NaN on gfx950)
typedef __bf16 bf16_t;
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }
// fp16 storage (BASELINE.json configs[4]: fp16 + dynamic loss scaling): same kernels, instantiated on _Float16
typedef _Float16 half_t;
#define SDE_IS16(dtype) ((dtype) == SDE_BF16 || (dtype) == SDE_F16)
#define SDE_DTYPE_OK(dtype) ((dtype) == SDE_F32 || (dtype) == SDE_BF16 || (dtype) == SDE_F16)
