// errors.hip — last-error string of the C ABI (thread-local; the library keeps no other global state).
#include "mi_internal.hpp"
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

int mi_fail(int code, const char* msg) {
    snprintf(g_err, sizeof(g_err), "%s", msg ? msg : "");
    return code;
}

int mi_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) return MI_OK;
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return MI_ERR_LAUNCH;
}

extern "C" const char* mi_last_error(void) { return g_err; }

extern "C" int mi_abi_version(void) { return 1; }

// device properties the host side reports in bench output (no torch needed)
extern "C" int mi_device_info(int device, int* cu_count, int* wave_size, char* arch, int arch_len) {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return mi_fail(MI_ERR_STATE, "hipGetDeviceProperties failed");
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (wave_size) *wave_size = prop.warpSize;
    if (arch && arch_len > 0) snprintf(arch, arch_len, "%s", prop.gcnArchName);
    return MI_OK;
}
