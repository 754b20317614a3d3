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

// ---------------------------------------------------------------------------------------------------------------
// CU partitioning (round 4): a stream whose kernels run on a subset of the chip's compute units, and the number of CUs the
// persistent tile GEMM sizes its grid for. GenerationPipeline gives the prompt phase (MFMA-bound: ViT + prefill) most of the
// chip and the decode loop of the previous batch (HBM-bound: weight + KV streaming) a few CUs of every XCD, so that the two
// phases of consecutive batches really run side by side — without a partition the persistent GEMM owns every CU and a
// second stream only gets the seams (round 3: +0.7 %).
static int g_cu_budget[GAR_MAX_DEVICES] = {};

extern "C" int gar_stream_create_cu_mask(const uint32_t* mask, int words, void** stream) {
    GAR_CHECK_ARG(mask && words > 0 && stream, "gar_stream_create_cu_mask: bad args");
    hipStream_t s = nullptr;
    hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask);
    if (e != hipSuccess) {
        gar_set_error("gar_stream_create_cu_mask: %s", hipGetErrorString(e));
        return GAR_ERR_UNSUPPORTED;
    }
    *stream = (void*)s;
    return GAR_OK;
}

extern "C" int gar_stream_destroy(void* stream) {
    if (stream && hipStreamDestroy((hipStream_t)stream) != hipSuccess) {
        gar_set_error("gar_stream_destroy failed");
        return GAR_ERR_LAUNCH;
    }
    return GAR_OK;
}

extern "C" int gar_set_cu_budget(int cus) {
    const int d = gar_current_device();
    GAR_CHECK_ARG(d >= 0 && d < GAR_MAX_DEVICES && cus >= 0, "gar_set_cu_budget: bad args");
    g_cu_budget[d] = cus;
    return GAR_OK;
}

int gar_cu_budget() {          // CUs a persistent grid may count on: the budget when one is set, else every CU of the device
    const int d = gar_current_device();
    const int n = gar_num_cus();
    if (d >= 0 && d < GAR_MAX_DEVICES && g_cu_budget[d] > 0 && g_cu_budget[d] < n) return g_cu_budget[d];
    return n;
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
