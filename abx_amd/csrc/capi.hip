// Library-level entry points: version, thread-local error text, device check.
#include <string.h>

#include "common.h"
#include "abx_hip.h"

static thread_local char g_err[512] = "";

void abx_set_error(const char* msg) {
    strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
    g_err[sizeof(g_err) - 1] = 0;
}

int abx_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
        return (int)e;
    }
    return ABX_OK;
}

extern "C" int abx_version(void) { return ABX_HIP_ABI_VERSION; }

extern "C" const char* abx_last_error_string(void) { return g_err; }

extern "C" int abx_init(int device) {
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "abx_init: %s", hipGetErrorString(e));
        return (int)e;
    }
    if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_err, sizeof(g_err), "abx_init: device %d is %s, this library is built for gfx950 only", device, p.gcnArchName);
        return ABX_ERR_ARG;
    }
    return ABX_OK;
}
