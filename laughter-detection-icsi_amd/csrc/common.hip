// liblad_hip.so: version + thread-local error string.
#include "lad_common.h"

namespace lad {

std::string &last_error_ref() {
    static thread_local std::string s;
    return s;
}

int fail(int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    last_error_ref() = buf;
    return code;
}

// Tickets of the launches whose LAST workgroup finishes the job (bn.hip: the two-level BatchNorm sums; stem.hip: the stem's backward):
// a workgroup takes a ticket when its part is out, the one that draws the last number does the rest and puts the slot back to zero.
// Launches draw their slot round-robin, so launches in flight on different streams do not share one (256 slots).
constexpr int N_TICKETS = 256;
__device__ unsigned int lad_tickets[N_TICKETS];
unsigned int *launch_ticket() {
    static std::atomic<unsigned> next{0};
    static std::atomic<unsigned int *> base_of[64];   // per device: the symbol's address is looked up once
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    unsigned int *base = base_of[dev].load(std::memory_order_acquire);
    if (base == nullptr) {
        if (hipGetSymbolAddress((void **)&base, HIP_SYMBOL(lad_tickets)) != hipSuccess || base == nullptr) return nullptr;
        base_of[dev].store(base, std::memory_order_release);
    }
    return base + (next.fetch_add(1u) % N_TICKETS);
}

}  // namespace lad

extern "C" int lad_version(void) { return LAD_VERSION; }
extern "C" const char *lad_last_error(void) { return lad::last_error_ref().c_str(); }
