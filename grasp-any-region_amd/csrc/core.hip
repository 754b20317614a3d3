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

// ---------------------------------------------------------------------------------------------------------------
// COCO run-length masks (host side). The benchmark loops of the reference decode `mask_rles` / `segmentation` entries
// with pycocotools (evaluation/GAR-Bench/inference.py:142-145, evaluation/DLC-Bench/inference.py:121-122); this is
// the published format restated: `counts` is a string of 6-bit groups (char - 48: 5 payload bits + continuation bit
// 0x20, sign-extended from the last group's 0x10 bit), every run after the second is stored as a difference to the
// run two places earlier, runs alternate background / foreground starting with background over the COLUMN-major
// pixel order. Output: row-major [h, w] bytes (0 / 1). Returns the number of foreground pixels or a negative code.
extern "C" int64_t gar_rle_decode(const char* counts, int64_t len, int h, int w, uint8_t* mask) {
    if (!counts || !mask || h <= 0 || w <= 0 || len < 0) {
        gar_set_error("gar_rle_decode: bad args");
        return GAR_ERR_ARG;
    }
    const int64_t total = (int64_t)h * w;
    int64_t pos = 0, fg = 0, prev2 = 0, prev1 = 0;       // runs m-2 and m-1
    int64_t m = 0, p = 0;
    uint8_t val = 0;
    while (p < len) {
        int64_t x = 0;
        int k = 0;
        bool more = true;
        while (more) {
            if (p >= len) { gar_set_error("gar_rle_decode: truncated counts string"); return GAR_ERR_ARG; }
            const int c = (int)counts[p] - 48;
            if (c < 0 || c > 63) { gar_set_error("gar_rle_decode: bad character at %lld", (long long)p); return GAR_ERR_ARG; }
            x |= (int64_t)(c & 0x1f) << (5 * k);
            more = (c & 0x20) != 0;
            ++p;
            ++k;
            if (!more && (c & 0x10)) x |= -((int64_t)1 << (5 * k));
        }
        if (m > 2) x += prev2;
        prev2 = prev1;
        prev1 = x;
        if (x < 0 || pos + x > total) { gar_set_error("gar_rle_decode: runs exceed %d x %d", h, w); return GAR_ERR_ARG; }
        for (int64_t i = 0; i < x; ++i, ++pos) {
            const int64_t col = pos / h, row = pos - col * h;
            mask[row * w + col] = val;
        }
        if (val) fg += x;
        val ^= 1;
        ++m;
    }
    if (pos != total) { gar_set_error("gar_rle_decode: runs cover %lld of %lld pixels", (long long)pos, (long long)total); return GAR_ERR_ARG; }
    return fg;
}
