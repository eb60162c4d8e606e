// Library-level entry points: version, thread-local error text, device check.
#include <string.h>

#include <mutex>
#include <set>
#include <utility>

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

int abx_ensure_dynamic_lds(const void* kernel, int bytes, const char* what) {
    static std::mutex mu;
    static std::set<std::pair<int, const void*>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) {
        std::lock_guard<std::mutex> lock(mu);
        if (done.count({dev, kernel})) return ABX_OK;
        e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess) done.insert({dev, kernel});
    }
    if (e != hipSuccess) {
        snprintf(g_err, sizeof(g_err), "%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed: %s", what, hipGetErrorString(e));
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

// Diagnostics: overwrite the LDS of every CU with a bit pattern (one 160 KB workgroup per CU slot, a few rounds), so that a test
// can show that no kernel's result depends on what an earlier workgroup - possibly of another process sharing the GPU - left
// in LDS (tests/test_gpu_model.py::test_results_do_not_depend_on_stale_lds).
namespace {
__global__ __launch_bounds__(1024) void poison_lds_kernel(unsigned pattern, unsigned* sink) {
    extern __shared__ unsigned lds_words[];
    constexpr int WORDS = 160 * 1024 / 4;
    for (int i = threadIdx.x; i < WORDS; i += 1024) lds_words[i] = pattern;
    __syncthreads();
    // keep the stores alive and the workgroup resident for a moment
    unsigned acc = 0;
    for (int i = threadIdx.x; i < WORDS; i += 1024) acc ^= lds_words[i];
    if (acc == 0x12345678u && sink) sink[0] = acc;
}
}  // namespace

extern "C" int abx_debug_poison_lds(unsigned pattern, hipStream_t st) {
    if (int rc = abx_ensure_dynamic_lds(reinterpret_cast<const void*>(poison_lds_kernel), 160 * 1024, "abx_debug_poison_lds")) return rc;
    hipLaunchKernelGGL(poison_lds_kernel, dim3(256 * 4), dim3(1024), 160 * 1024, st, pattern, (unsigned*)nullptr);
    return abx_check_launch("abx_debug_poison_lds");
}
