// Thread-local error string + ABI version for libwsi_hgnn.so (see include/wsi_hgnn.h).
#include "common.h"

namespace wsi {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace wsi

extern "C" int wsi_abi_version(void) { return WSI_ABI_VERSION; }
extern "C" const char* wsi_last_error(void) { return wsi::g_err; }
