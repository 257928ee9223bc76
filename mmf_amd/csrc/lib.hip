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
