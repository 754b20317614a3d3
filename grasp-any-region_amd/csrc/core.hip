// Error reporting, ABI version and device check.
#include <stdarg.h>
#include <string.h>

#include "common.h"

static thread_local char g_err[512] = "";

void gar_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char* gar_last_error(void) { return g_err; }
extern "C" int gar_abi_version(void) { return GAR_ABI_VERSION; }

extern "C" int gar_check_device(int device) {
    hipDeviceProp_t prop;
    hipError_t e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        gar_set_error("gar_check_device: %s", hipGetErrorString(e));
        return GAR_ERR_LAUNCH;
    }
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        gar_set_error("gar_check_device: device %d is %s; libgar_hip.so holds gfx950 code only", device, prop.gcnArchName);
        return GAR_ERR_ARCH;
    }
    return GAR_OK;
}

__global__ void counter_add_kernel(int32_t* c, int n, int delta) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) c[i] += delta;
}

extern "C" int gar_counter_add(int32_t* counters, int n, int delta, gar_stream_t stream) {
    GAR_CHECK_ARG(counters && n > 0, "gar_counter_add: bad args");
    hipLaunchKernelGGL(counter_add_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, counters, n, delta);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}
