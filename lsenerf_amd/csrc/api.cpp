// Error reporting + version of the C-ABI (include/lse_hip.h).
#include "common.h"
#include <string.h>

namespace lse {
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace lse

extern "C" const char *lse_last_error(void) { return lse::g_err; }
extern "C" int lse_abi_version(void) { return LSE_ABI_VERSION; }
