// Shared device/host helpers for libgar_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <mutex>

#include "../../include/gar_hip.h"

typedef uint16_t bf16_t;  // raw bfloat16 bits

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;   // MFMA bf16 operand (4 VGPRs)
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned int;

// The library's 16-bit element type: bfloat16, or — in the twin build -DGAR_HALF_F16=1 (libgar_hip_f16.so, the reference's
// --data_type fp16) — IEEE binary16. Storage type, layouts, kernels and entry points are the same; only these conversions, the
// matrix instruction's operand format and two range constants differ. "bf16" in names below means "the 16-bit type".
#ifndef GAR_HALF_F16
#define GAR_HALF_F16 0
#endif
typedef float f32x2_hw_t __attribute__((ext_vector_type(2)));
#if GAR_HALF_F16
typedef _Float16 h16x2_hw_t __attribute__((ext_vector_type(2)));
typedef _Float16 h16x8_hw_t __attribute__((ext_vector_type(8)));
__device__ __forceinline__ float bf2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
__device__ __forceinline__ float unpk_lo(unsigned int w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
__device__ __forceinline__ float unpk_hi(unsigned int w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
// fp32 -> fp16, round to nearest even (overflow -> inf, as the reference's .half())
__device__ __forceinline__ unsigned int pack_bf2(float lo, float hi) {
    const f32x2_hw_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, h16x2_hw_t));
}
#define MFMA_32x32x16(a, b, c) \
    __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8_hw_t, a), __builtin_bit_cast(h16x8_hw_t, b), c, 0, 0, 0)
#define MFMA_16x16x32(a, b, c) \
    __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8_hw_t, a), __builtin_bit_cast(h16x8_hw_t, b), c, 0, 0, 0)
#define H16_ONE 0x3C00
// c + a.lo * b.lo + a.hi * b.hi on packed pairs of the 16-bit type (products exact, fp32 accumulation): v_dot2_f32_f16
__device__ __forceinline__ float dot2_acc(unsigned int a, unsigned int b, float c) {
    return __builtin_amdgcn_fdot2(__builtin_bit_cast(h16x2_hw_t, a), __builtin_bit_cast(h16x2_hw_t, b), c, false);
}
#define H16_MAX_LOG2 15      /* largest power of two a stored softmax weight may reach (fp16 ends at 65504) */
#else
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((unsigned int)v) << 16); }
__device__ __forceinline__ float unpk_lo(unsigned int w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float unpk_hi(unsigned int w) { return __uint_as_float(w & 0xffff0000u); }
// fp32 -> bf16, round to nearest even, NaN quieted: gfx950's v_cvt_pk_bf16_f32 (the integer formulation costs ~10
// instructions per element, four of them exec-mask juggling for the NaN case — it dominated the GEMM epilogue)
typedef __bf16 bf16x2_hw_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pack_bf2(float lo, float hi) {
    const f32x2_hw_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, bf16x2_hw_t));
}
#define MFMA_32x32x16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define MFMA_16x16x32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#define H16_ONE 0x3F80
// c + a.lo * b.lo + a.hi * b.hi on packed pairs of the 16-bit type (products exact, fp32 accumulation): v_dot2c_f32_bf16
__device__ __forceinline__ float dot2_acc(unsigned int a, unsigned int b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_hw_t, a), __builtin_bit_cast(bf16x2_hw_t, b), c, false);
}
#define H16_MAX_LOG2 16
#endif
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }

template <typename T> struct DT;
template <> struct DT<float> {
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct DT<bf16_t> {
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// vector of 4 elements <-> 4 floats
__device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
__device__ __forceinline__ void ld4(const bf16_t* p, float (&v)[4]) {
    uint2 t = *reinterpret_cast<const uint2*>(p);
    v[0] = unpk_lo(t.x); v[1] = unpk_hi(t.x);
    v[2] = unpk_lo(t.y); v[3] = unpk_hi(t.y);
}
__device__ __forceinline__ void st4(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void st4(bf16_t* p, const float (&v)[4]) {
    *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
}
// 8 elements
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
    float4 a = reinterpret_cast<const float4*>(p)[0], b = reinterpret_cast<const float4*>(p)[1];
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void ld8(const bf16_t* p, float (&v)[8]) {
    uint4 t = *reinterpret_cast<const uint4*>(p);
    v[0] = unpk_lo(t.x); v[1] = unpk_hi(t.x);
    v[2] = unpk_lo(t.y); v[3] = unpk_hi(t.y);
    v[4] = unpk_lo(t.z); v[5] = unpk_hi(t.z);
    v[6] = unpk_lo(t.w); v[7] = unpk_hi(t.w);
}
__device__ __forceinline__ void st8(float* p, const float (&v)[8]) {
    reinterpret_cast<float4*>(p)[0] = make_float4(v[0], v[1], v[2], v[3]);
    reinterpret_cast<float4*>(p)[1] = make_float4(v[4], v[5], v[6], v[7]);
}
__device__ __forceinline__ void st8(bf16_t* p, const float (&v)[8]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]),
                                              pack_bf2(v[6], v[7]));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// HF Llama apply_rotary_pos_emb on one (x[d], x[d + hd/2]) pair: q*cos + rotate_half(q)*sin, then the scale folded into q.
// One explicit operation order, shared by gar_llm_qkv_post and the decode attention's fused prologue: the two paths are
// bit-identical.
__device__ __forceinline__ void rope_half_pair(float x1, float x2, float c, float s, float sc, float& o1, float& o2) {
    o1 = __builtin_fmaf(-x2, s, x1 * c) * sc;
    o2 = __builtin_fmaf(x1, s, x2 * c) * sc;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
// erf via Abramowitz-Stegun 7.1.26 (|err| < 1.5e-7) without the 1+erf cancellation: gelu(x) = max(x,0) - |x|/2 * poly(t) * exp(-x^2/2)
// (bf16 kernels only; the f32 parity kernels keep erff)
__device__ __forceinline__ float gelu_fast(float x) {
    const float ax = fabsf(x);
    const float z = ax * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float e = exp2f(-z * z * 1.4426950408889634f);
    return fmaxf(x, 0.f) - 0.5f * ax * poly * e;
}
__device__ __forceinline__ float silu(float x) { return x / (1.0f + expf(-x)); }
// bf16 kernels: hardware exp2 / rcp (1 ulp in f32) instead of expf + an IEEE division — the result is rounded to bf16 anyway
__device__ __forceinline__ float silu_fast(float x) {
    return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
}

// ---- host side -----------------------------------------------------------------------------------------------
void gar_set_error(const char* fmt, ...);
// One-time setup that is per DEVICE (hipFuncSetAttribute, the CU count): a process may drive several GPUs (one GARModel
// per device), so "done once" is keyed by the current HIP device. Devices beyond the table redo the setup every call.
#define GAR_MAX_DEVICES 32
static inline int gar_current_device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess) d = -1;
    return d;
}
struct gar_once_per_device {
    // run(f): f() once per device, and no caller returns before the f() of its device has COMPLETED — a second host thread
    // driving the same device must not launch a kernel that needs the attribute f sets (dynamic LDS above 64 KiB) before it is
    // set (the flag used to be raised before the caller ran hipFuncSetAttribute)
    std::atomic<bool> done[GAR_MAX_DEVICES] = {};
    std::mutex mu;
    template <class F>
    void run(F&& f) {
        const int d = gar_current_device();
        if (d < 0 || d >= GAR_MAX_DEVICES) {
            f();
            return;
        }
        if (done[d].load(std::memory_order_acquire)) return;
        std::lock_guard<std::mutex> lock(mu);
        if (done[d].load(std::memory_order_relaxed)) return;
        f();
        done[d].store(true, std::memory_order_release);
    }
};
static inline int gar_num_cus() {          // of the current device
    static int cached[GAR_MAX_DEVICES] = {};
    const int d = gar_current_device();
    if (d >= 0 && d < GAR_MAX_DEVICES && cached[d] > 0) return cached[d];
    int n = 256;
    if (d < 0 || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, d) != hipSuccess || n <= 0) n = 256;
    if (d >= 0 && d < GAR_MAX_DEVICES) cached[d] = n;
    return n;
}
#define GAR_CHECK_ARG(cond, ...)                      \
    do {                                              \
        if (!(cond)) {                                \
            gar_set_error(__VA_ARGS__);               \
            return GAR_ERR_ARG;                       \
        }                                             \
    } while (0)
#define GAR_CHECK_LAUNCH()                                                            \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            gar_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e__)); \
            return GAR_ERR_LAUNCH;                                                    \
        }                                                                             \
    } while (0)
