// Shared host-side helpers for liblad_hip.so (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

#include "../../include/lad_hip.h"

namespace lad {

std::string &last_error_ref();
int fail(int code, const char *fmt, ...);

#define LAD_HIP_CHECK(expr)                                                                         \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess)                                                                       \
            return ::lad::fail(LAD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                               __FILE__, __LINE__);                                                 \
    } while (0)

#define LAD_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) return ::lad::fail(LAD_ERR_INVALID, __VA_ARGS__); \
    } while (0)

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(LAD_ERR_HIP, "launch of %s failed: %s", what, hipGetErrorString(e));
    return LAD_OK;
}

// "done once per DEVICE" flag for hipFuncSetAttribute(MaxDynamicSharedMemorySize): the attribute belongs to the device's copy of the
// kernel, so a process that drives a second GPU has to opt in again there.  Used as `static DeviceOnce attr_set; if (!attr_set) { ...;
// attr_set = true; }` -- thread-safe (a race only repeats the cheap, idempotent call).
struct DeviceOnce {
    std::atomic<uint64_t> mask[4] = {};
    static int dev() {
        int d = 0;
        (void)hipGetDevice(&d);
        return d & 255;
    }
    bool operator!() const { const int d = dev(); return !((mask[d >> 6].load(std::memory_order_acquire) >> (d & 63)) & 1); }
    DeviceOnce &operator=(bool) {
        const int d = dev();
        mask[d >> 6].fetch_or(1ull << (d & 63), std::memory_order_release);
        return *this;
    }
};

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Partial slabs of one weight-gradient launch (slab_reduce.hip): slab[wg][tap][ci][co], bias_slab[wg][co] (or nullptr)
struct SlabReduce {
    const float *slabs, *bias_slabs;
    float *dw, *dbias;
    int groups, cin, cout, taps;
};
// sums them into dw / dbias now -- or, between lad_wgrad_defer_begin and lad_wgrad_defer_flush, at the flush
int reduce_slabs(const SlabReduce &d, hipStream_t st);

// device address of the next launch's ticket (common.hip; nullptr: the runtime refused, the caller reports it)
unsigned int *launch_ticket();

}  // namespace lad
