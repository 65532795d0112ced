// Error reporting + version of the C-ABI (include/lse_hip.h).
#include "common.h"
#include <string.h>
#include <atomic>

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

namespace lse {
// run-time development knobs (include/lse_hip.h: lse_set_option); plain ints, read at every launch
struct Option { const char *name; std::atomic<int64_t> value; };
static Option g_options[] = {
    {"hash_fwd_mapping", {4}},
    {"hash_fwd_lds_levels", {0}},
    {"compact_features_groups", {4}},
    {"mlp_fwd_cfg", {28}},
    {"mlp_bwd_cfg", {28}},
    {"mlp_bwd_impl", {1}},
    {"mlp_fwd_impl", {2}},
    {"mlp_bwd3_cfg", {208}},
    {"mlp_act_nt", {0}},
    {"hash_bwd_probes", {3}},
    {"hash_bwd_few_runs", {6}},
    {"hash_bwd_stage_max", {16}},
    {"traverse_vec", {1}},
    {"traverse_fma", {0}},
};
static thread_local const int64_t *g_device_count = nullptr;
const int64_t *device_count() { return g_device_count; }

int64_t option(const char *name)
{
    for (auto &o : g_options)
        if (!strcmp(o.name, name)) return o.value.load(std::memory_order_relaxed);
    return 0;
}
}  // namespace lse

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

extern "C" int lse_set_device_count(const int64_t *n_dev)
{
    lse::g_device_count = n_dev;
    return LSE_OK;
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

extern "C" const char *lse_last_error(void) { return lse::g_err; }
extern "C" int lse_abi_version(void) { return LSE_ABI_VERSION; }
