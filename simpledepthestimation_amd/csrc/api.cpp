// Error string + version for libsde_hip.so (see include/sde_hip.h).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"
#include "sde_hip.h"

static thread_local char g_err[512] = "";

void sde_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* sde_last_error(void) { return g_err; }
extern "C" int sde_version(void) { return 1; }

// Timeline markers: one tiny kernel that stores the device's constant-rate wall clock.  Enqueued (and captured into the step's hipGraph) at chosen points of
// the step's streams, they give the replayed step's real timeline -- a profiler's kernel trace serialises the graph's parallel branches
// (profiles/README.md, round 3), these do not.  Diagnostic only (bench.py --marks); nothing reads them in training.
__global__ void sde_mark_time_kernel(unsigned long long* slot) { *slot = wall_clock64(); }

extern "C" int sde_mark_time(uint64_t* slot, sde_stream_t stream) {
    if (!slot) { sde_set_error("sde_mark_time: null slot"); return -1; }
    hipLaunchKernelGGL(sde_mark_time_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)slot);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { sde_set_error("sde_mark_time: %s", hipGetErrorString(e)); return -2; }
    return 0;
}

extern "C" int sde_wall_clock_khz(void) {
    int dev = 0, khz = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return -1;
    return khz;
}

// Stream-ordered "this point has been reached" flag for the host: one thread stores `value` to *dst (host-pinned, device-visible memory) with a
// system-scope release.  The device prefetcher (data/build.py) hands batches over with these instead of hipEventRecord / hipStreamWaitEvent pairs between
// the copy stream and the training stream: on this stack every event recorded on the training stream between two replays of the step graph cost
// 0.2-0.3 ms of idle GPU per step (profiles/README.md, round 3), a one-thread kernel costs its 2 us.
__global__ void sde_store_u64_kernel(unsigned long long* dst, unsigned long long value) {
    __hip_atomic_store(dst, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

extern "C" int sde_store_u64(uint64_t* dst, uint64_t value, sde_stream_t stream) {
    if (!dst) { sde_set_error("sde_store_u64: null destination"); return -1; }
    hipLaunchKernelGGL(sde_store_u64_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)dst, (unsigned long long)value);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { sde_set_error("sde_store_u64: %s", hipGetErrorString(e)); return -2; }
    return 0;
}
