// mmf_amd :: library-level C ABI (version, error string).
#include "common.h"
#include "mmf_amd.h"
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" void mmf_amd_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* mmf_amd_last_error(void) { return g_err; }
extern "C" int mmf_amd_abi_version(void) { return MMF_AMD_ABI_VERSION; }
extern "C" const char* mmf_amd_target(void) { return "gfx950"; }

// Tunables: integer knobs for on-hardware sweeps (tools/*_bench.py).  0 = built-in heuristic.
static int g_tun[MMF_TUN_COUNT] = {0};
extern "C" int mmf_amd_set_tunable(int which, int value) {
    if (which < 0 || which >= MMF_TUN_COUNT) { mmf_amd_set_error("mmf_amd_set_tunable: unknown tunable"); return 1; }
    g_tun[which] = value;
    return 0;
}
extern "C" int mmf_amd_get_tunable(int which) { return (which >= 0 && which < MMF_TUN_COUNT) ? g_tun[which] : 0; }
