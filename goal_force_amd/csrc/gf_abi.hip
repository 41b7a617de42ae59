// gf_abi.hip — library identity + thread-local error channel of the C ABI.
#include "gf_common.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#define GF_ABI_VERSION 14

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

// ---- A/B and diagnostic knobs (gf_common.h::GfOptions)
static GfOptions g_options;
static std::once_flag g_options_once;

static int env_int(const char* name, int dflt, int lo, int hi) {
    const char* v = getenv(name);
    if (!v || !*v) return dflt;
    const long x = strtol(v, nullptr, 10);
    return (int)(x < lo ? lo : (x > hi ? hi : x));     // out-of-range values are clamped, never used raw (a negative stagger
}                                                      // used to become a huge K offset)

static void load_options() {
    const char* k = getenv("GF_GEMM_KERNEL");
    int gk = 0;
    if (k && k[0] == 'p') gk = 1;
    else if (k && k[0] == 's') gk = (k[1] == 'l' && k[2] == '8') ? 3 : 2;
    g_options.gemm_kernel.store(gk, std::memory_order_relaxed);
    const char* v1 = getenv("GF_GEMM_V1");
    g_options.gemm_v1.store(v1 && *v1 ? (v1[0] == '1' ? 1 : 0) : -1, std::memory_order_relaxed);
    g_options.a4_stagger.store(env_int("GF_A4_STAGGER", 2, 0, 1 << 20), std::memory_order_relaxed);
    g_options.a4_group_m.store(env_int("GF_A4_GROUP_M", 0, 0, 64), std::memory_order_relaxed);
    const char* el = getenv("GF_A4_LOOP");
    g_options.a4_loop_h.store((el && el[0] == 'h') ? 1 : 0, std::memory_order_relaxed);
    g_options.a4_whatif.store(env_int("GF_A4_WHATIF", 0, 0, 1 << 20), std::memory_order_relaxed);
    const char* ak = getenv("GF_ATTN_KERNEL");
    g_options.attn_kernel1.store((ak && ak[0] == '1') ? 1 : 0, std::memory_order_relaxed);
    const char* bw = getenv("GF_ATTN_BWD");
    g_options.bwd_v1.store((bw && bw[0] == 'v' && bw[1] == '1') ? 1 : 0, std::memory_order_relaxed);
    g_options.conv_nb.store(env_int("GF_CONV_NB", 0, 0, 2), std::memory_order_relaxed);
    g_options.conv_gather.store(env_int("GF_CONV_GATHER", 0, 0, 1), std::memory_order_relaxed);
    g_options.conv_direct.store(env_int("GF_CONV_DIRECT", 1, 0, 1), std::memory_order_relaxed);
    g_options.vae_rms3.store(env_int("GF_VAE_RMS3", 1, 0, 1), std::memory_order_relaxed);
}

const GfOptions& gf_options() {
    std::call_once(g_options_once, load_options);
    return g_options;
}

extern "C" GF_API void gf_reload_options(void) {
    std::call_once(g_options_once, [] {});
    load_options();
}
