// How many workgroups of a given shape does a gfx950 CU actually hold at once? (diagnostic, not part of the library)
//   hipcc --offload-arch=gfx950 -O3 tools/occupancy_probe.hip -o /tmp/occupancy_probe && /tmp/occupancy_probe
// Every workgroup spins for a fixed number of shader clocks; with a grid of N workgroups per CU the launch takes
// ceil(N / resident) spin periods. Also prints what hipOccupancyMaxActiveBlocksPerMultiprocessor reports.
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int VG>
__global__ __launch_bounds__(1024) void spin(float* out, unsigned long long ticks) {
    extern __shared__ char smem[];
    float keep[VG];                       // VG live VGPRs
#pragma unroll
    for (int i = 0; i < VG; ++i) keep[i] = (float)(threadIdx.x + i);
    if (threadIdx.x == 0) smem[0] = 1;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < VG; ++i) keep[i] = keep[i] * 1.0001f + 0.5f;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VG; ++i) s += keep[i];
    if (s == 1.2345e30f) out[threadIdx.x] = s + smem[0];
}

template <int VG>
static void run(int cus, float* d, int lds, int threads = 256) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&spin<VG>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spin<VG>, threads, lds);
    hipFuncAttributes fa;
    hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&spin<VG>));
    printf("%4d threads, %3d VGPRs, %6d B LDS: runtime says %2d workgroups / CU;  launch time in spin periods for N per CU:",
           threads, fa.numRegs, lds, nb);
    const unsigned long long ticks = 400000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float base = 0.f;
    for (int n : {1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 16, 17, 24, 32, 33}) {
        if (threads >= 256 && n > 9) break;
        hipLaunchKernelGGL(spin<VG>, dim3(cus * n), dim3(threads), lds, 0, d, ticks);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(spin<VG>, dim3(cus * n), dim3(threads), lds, 0, d, ticks);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (n == 1) base = ms;
        printf("  %d:%.2f", n, ms / base);
    }
    printf("\n");
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* d;
    hipMalloc(&d, 4096);
    for (int threads : {64, 128, 512, 1024}) run<32>(cus, d, 0, threads);
    run<100>(cus, d, 0, 64);
    run<56>(cus, d, 0, 256);
    run<80>(cus, d, 0, 256);
    for (int lds : {0, 16384, 32768, 49152, 65536}) {
        run<32>(cus, d, lds);
        run<100>(cus, d, lds);
        run<140>(cus, d, lds);
    }
    return 0;
}
