// libsdnative: ABI version + thread-local error reporting.
#include "sdn_common.h"

namespace sdn {

char *error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

}  // namespace sdn

extern "C" {

int sdn_abi_version(void) { return SDN_ABI_VERSION; }
const char *sdn_last_error(void) { return sdn::error_buffer(); }

}
