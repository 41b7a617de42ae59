// gf_abi.hip — library identity + thread-local error channel of the C ABI.
#include "gf_common.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define GF_ABI_VERSION 18

static thread_local char g_err[512] = "";

void gf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" GF_API const char* gf_version(void) { return "goalforce-hip 0.1.0 gfx950"; }
extern "C" GF_API const char* gf_last_error(void) { return g_err; }
extern "C" GF_API int gf_abi_version(void) { return GF_ABI_VERSION; }

// ---- dispatch overrides (gf_common.h::GfOptions): set by explicit calls only — the library never reads the environment
static GfOptions g_options;

const GfOptions& gf_options() { return g_options; }

namespace {
struct OptionSlot {
    const char* name;
    std::atomic<int> GfOptions::*field;
    int lo, hi, dflt;
};
const OptionSlot kSlots[] = {
    {"prefer_8wave", &GfOptions::prefer_8wave, 0, 1, 0}, {"a4_stagger", &GfOptions::a4_stagger, 0, 1 << 20, 2},
    {"a4_group_m", &GfOptions::a4_group_m, 0, 64, 0},    {"conv_nb", &GfOptions::conv_nb, 0, 2, 0},
    {"conv_gather", &GfOptions::conv_gather, 0, 1, 0},   {"vae_rms3", &GfOptions::vae_rms3, 0, 1, 1},
    {"conv_direct", &GfOptions::conv_direct, 0, 1, 1},
};
}  // namespace

extern "C" GF_API int gf_set_option(const char* name, int value) {
    GF_CHECK_ARG(name, "gf_set_option: null name");
    for (const OptionSlot& s : kSlots)
        if (strcmp(name, s.name) == 0) {
            // out-of-range values are clamped, never used raw (a negative stagger would become a huge K offset)
            (g_options.*(s.field)).store(value < s.lo ? s.lo : (value > s.hi ? s.hi : value), std::memory_order_relaxed);
            return GF_OK;
        }
    gf_set_error("gf_set_option: unknown option '%s'", name);
    return GF_ERR_INVALID_ARG;
}

extern "C" GF_API int gf_get_option(const char* name, int* value) {
    GF_CHECK_ARG(name && value, "gf_get_option: null argument");
    for (const OptionSlot& s : kSlots)
        if (strcmp(name, s.name) == 0) {
            *value = (g_options.*(s.field)).load(std::memory_order_relaxed);
            return GF_OK;
        }
    gf_set_error("gf_get_option: unknown option '%s'", name);
    return GF_ERR_INVALID_ARG;
}

extern "C" GF_API void gf_reset_options(void) {
    for (const OptionSlot& s : kSlots) (g_options.*(s.field)).store(s.dflt, std::memory_order_relaxed);
}
