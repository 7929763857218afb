// Error plumbing and version of the C ABI (host only).
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <vector>

#include <hip/hip_runtime.h>

#include "../../include/dgl_amd.h"

static thread_local char g_err[512] = "";

int dgl_set_error(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

// Raise a kernel's dynamic-LDS limit to `bytes`, once per (current device, kernel): the attribute belongs to the device's
// copy of the kernel, and asking again before every launch is a runtime call on paths whose launches are microseconds
// apart.  The limit only grows.
void dgl_max_dynamic_lds(const void *kernel, int bytes) {
    struct Seen {
        const void *kernel;
        int device, bytes;
    };
    static std::mutex mu;
    static std::vector<Seen> seen;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) device = -1;
    std::lock_guard<std::mutex> lock(mu);
    for (Seen &e : seen) {
        if (e.kernel == kernel && e.device == device) {
            if (e.bytes >= bytes) return;
            e.bytes = bytes;
            (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
            return;
        }
    }
    (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (device >= 0) seen.push_back(Seen{kernel, device, bytes});
}

// ---- square-root model (common.h: DglSqrtIeee / DglSqrtMkl) -----------------------------------------------------
// One setting per device, read by the launchers of the state kernels; stored as model + 1, 0 = never set = the default
// DGL_SQRT_MKL_AVX512.
#define DGL_MAX_DEVICES 64
static int g_sqrt_model[DGL_MAX_DEVICES];
static std::mutex g_sqrt_mutex;

bool dgl_sqrt_reference_bits() {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess || device < 0 || device >= DGL_MAX_DEVICES) return true;
    std::lock_guard<std::mutex> lock(g_sqrt_mutex);
    return g_sqrt_model[device] == 0 || g_sqrt_model[device] == DGL_SQRT_MKL_AVX512 + 1;
}

extern "C" int dgl_set_sqrt_model(int32_t model) {
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return dgl_set_error(DGL_ERR_LAUNCH, "hipGetDevice: %s", hipGetErrorString(e));
    if (device < 0 || device >= DGL_MAX_DEVICES) return dgl_set_error(DGL_ERR_UNSUPPORTED, "device ordinal %d", device);
    if (model != DGL_SQRT_IEEE && model != DGL_SQRT_MKL_AVX512)
        return dgl_set_error(DGL_ERR_INVALID_ARGUMENT, "unknown square-root model %d", model);
    std::lock_guard<std::mutex> lock(g_sqrt_mutex);
    g_sqrt_model[device] = model + 1;
    return DGL_OK;
}

extern "C" int dgl_version(void) { return DGL_AMD_VERSION; }
extern "C" const char *dgl_last_error_string(void) { return g_err; }
