// Error reporting + version of the C-ABI (include/lse_hip.h), and the tuning knobs of the DEVELOPMENT build.
#include "common.h"
#include <string.h>
#include <atomic>

namespace lse {
// the calling thread's last error MESSAGE: written by a failing call, never read by the library (lse_last_error hands it out)
static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Tuning knobs.  In the library that ships (liblse_hip.so) they are CONSTANTS: `option()` is a lookup in a read-only table and
// nothing can change a value -- the library has no state.  The development build (make dev -> liblse_hip_dev.so,
// -DLSE_DEV_KNOBS, csrc/dev_knobs.h) makes the same table writable through lse_set_option for A/B runs of whole steps.
struct Option {
    const char *name;
#ifdef LSE_DEV_KNOBS
    std::atomic<int64_t> value;
#else
    int64_t value;
#endif
};
#ifdef LSE_DEV_KNOBS
static Option g_options[] = {
#else
static const Option g_options[] = {
#endif
    {"hash_fwd_mapping", 4},
    {"hash_fwd_lds_levels", 0},
    {"compact_features_groups", 4},
    {"mlp_fwd_cfg", 28},
    {"mlp_bwd_cfg", 28},
    {"mlp_bwd_impl", 1},
    {"mlp_bwd3_cfg", 208},
    {"mlp_act_nt", 0},
    {"hash_bwd_probes", 3},
    {"hash_bwd_few_runs", 8},
    {"hash_bwd_stage_max", 48},
    {"traverse_vec", 1},
};

int64_t option(const char *name)
{
    for (auto &o : g_options)
        if (!strcmp(o.name, name)) {
#ifdef LSE_DEV_KNOBS
            return o.value.load(std::memory_order_relaxed);
#else
            return o.value;
#endif
        }
    return 0;
}
}  // namespace lse

#ifdef LSE_DEV_KNOBS
#include "dev_knobs.h"
extern "C" int lse_set_option(const char *name, int64_t value)
{
    LSE_REQUIRE(name, "lse_set_option: null name");
    for (auto &o : lse::g_options)
        if (!strcmp(o.name, name)) {
            o.value.store(value, std::memory_order_relaxed);
            return LSE_OK;
        }
    lse::set_error("lse_set_option: unknown option '%s'", name);
    return LSE_E_INVALID;
}

extern "C" int lse_get_option(const char *name, int64_t *value)
{
    LSE_REQUIRE(name && value, "lse_get_option: null pointer");
    for (auto &o : lse::g_options)
        if (!strcmp(o.name, name)) {
            *value = o.value.load(std::memory_order_relaxed);
            return LSE_OK;
        }
    lse::set_error("lse_get_option: unknown option '%s'", name);
    return LSE_E_INVALID;
}
#endif

extern "C" const char *lse_last_error(void) { return lse::g_err; }
extern "C" int lse_abi_version(void) { return LSE_ABI_VERSION; }
