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

}  // namespace lad

extern "C" int lad_version(void) { return LAD_VERSION; }
extern "C" const char *lad_last_error(void) { return lad::last_error_ref().c_str(); }
