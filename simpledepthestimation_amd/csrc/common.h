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
__device__ __forceinline__ float sde_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
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
