// CU-mask probe (round 4): does hipExtStreamCreateWithCUMask partition the chip on this box, how do mask bits map to XCDs, and
// do two masked streams run side by side?   hipcc --offload-arch=gfx950 -O2 tools/cu_mask_probe.hip -o tools/bin/cu_mask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void where_kernel(unsigned* out, int spin) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);        // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(4);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}

__global__ void busy_kernel(unsigned* out, long long cycles) {          // one workgroup per CU it is given, spins `cycles`
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)cycles) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) out[blockIdx.x] = 1;
}

__global__ void stream_kernel(const float4* __restrict__ src, float* __restrict__ out, long long n4) {   // HBM reader
    float acc = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}

static void census(hipStream_t s, const char* name, unsigned* dbuf, int nblk) {
    std::vector<unsigned> h(2 * nblk);
    CK(hipMemsetAsync(dbuf, 0xff, 8 * nblk, s));
    hipLaunchKernelGGL(where_kernel, dim3(nblk), dim3(64), 0, s, dbuf, 200000);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(h.data(), dbuf, 8 * nblk, hipMemcpyDeviceToHost));
    std::map<unsigned, int> cus;
    int per_xcc[16] = {};
    for (int i = 0; i < nblk; ++i) {
        const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
        if (!cus.count(key)) per_xcc[xcc]++;
        cus[key]++;
    }
    printf("%-28s distinct CUs %3zu | per XCC:", name, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %2d", per_xcc[x]);
    printf("\n");
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs\n", prop.name, ncu);
    const int nblk = 8192;
    unsigned* dbuf;
    CK(hipMalloc(&dbuf, 8 * nblk));
    hipStream_t s0;
    CK(hipStreamCreate(&s0));
    census(s0, "unmasked", dbuf, nblk);
    const int words = (ncu + 31) / 32;
    struct M { const char* name; std::vector<uint32_t> m; };
    std::vector<M> masks;
    { M m{"first 16 bits off", std::vector<uint32_t>(words, 0xffffffffu)}; m.m[0] = 0xffff0000u; masks.push_back(m); }
    { M m{"last 16 bits off", std::vector<uint32_t>(words, 0xffffffffu)}; m.m[words - 1] = 0x0000ffffu; masks.push_back(m); }
    { M m{"only first 16 bits", std::vector<uint32_t>(words, 0u)}; m.m[0] = 0x0000ffffu; masks.push_back(m); }
    { M m{"only bits 0..7", std::vector<uint32_t>(words, 0u)}; m.m[0] = 0x000000ffu; masks.push_back(m); }
    { M m{"only bits 0,8,16,..,120", std::vector<uint32_t>(words, 0u)}; for (int b = 0; b < 128; b += 8) m.m[b / 32] |= 1u << (b % 32); masks.push_back(m); }
    { M m{"only bits 0..31", std::vector<uint32_t>(words, 0u)}; m.m[0] = 0xffffffffu; masks.push_back(m); }
    { M m{"only bits 224..255", std::vector<uint32_t>(words, 0u)}; m.m[words - 1] = 0xffffffffu; masks.push_back(m); }
    std::vector<hipStream_t> ss;
    for (auto& m : masks) {
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)m.m.size(), m.m.data());
        if (e != hipSuccess) { printf("%-28s hipExtStreamCreateWithCUMask: %s\n", m.name, hipGetErrorString(e)); ss.push_back(nullptr); continue; }
        census(s, m.name, dbuf, nblk);
        ss.push_back(s);
    }
    // ---- side by side: a big masked stream kept busy by a chip-filling spin kernel, a small masked stream streaming from HBM
    const long long n4 = (1ll << 30) / 16;                  // 1 GiB
    float4* src;
    float* o;
    CK(hipMalloc(&src, n4 * 16));
    CK(hipMalloc(&o, 64));
    CK(hipMemset(src, 0, n4 * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_stream = [&](hipStream_t s, int blocks, const char* what) {
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(stream_kernel, dim3(blocks), dim3(256), 0, s, src, o, n4);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  %-58s %8.3f ms per GiB  %7.1f GB/s\n", what, ms / 4, 4.0 * 1.073741824 / (ms * 1e-3));
    };
    if (ss[0] && ss[2]) {
        hipStream_t big = ss[0], small = ss[2];            // "first 16 bits off" / "only first 16 bits"
        printf("HBM reads of 1 GiB (x4):\n");
        time_stream(s0, 4096, "unmasked stream, 4096 blocks, idle chip");
        time_stream(small, 256, "16-CU stream, 256 blocks, idle chip");
        unsigned* flag;
        CK(hipMalloc(&flag, 4 * 4096));
        // the big stream spins on its CUs for ~50 ms (2048 blocks of 1024 threads: more than it can hold -> queued rounds)
        hipLaunchKernelGGL(busy_kernel, dim3(240 * 4), dim3(1024), 0, big, flag, 30000000ll);
        time_stream(small, 256, "16-CU stream while the 240-CU stream is saturated (spin)");
        CK(hipStreamSynchronize(big));
        hipLaunchKernelGGL(busy_kernel, dim3(256 * 4), dim3(1024), 0, s0, flag, 30000000ll);
        time_stream(small, 256, "16-CU stream while an UNMASKED stream saturates every CU");
        CK(hipStreamSynchronize(s0));
        // and the other way round: does the masked big stream keep its CUs when the small one is busy?
        hipLaunchKernelGGL(busy_kernel, dim3(16 * 4), dim3(1024), 0, small, flag, 30000000ll);
        time_stream(big, 4096, "240-CU stream while the 16-CU stream is saturated (spin)");
        CK(hipStreamSynchronize(small));
    }
    printf("done\n");
    return 0;
}
