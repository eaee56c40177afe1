// libesme_hip: ABI version + thread-local error string.
#include "launch.h"

namespace esme {
char* error_buffer() {
    static thread_local char buf[kErrorBufferSize] = "";
    return buf;
}
}  // namespace esme

extern "C" int esme_hip_abi_version(void) { return ESME_HIP_ABI_VERSION; }
extern "C" const char* esme_hip_last_error(void) { return esme::error_buffer(); }
