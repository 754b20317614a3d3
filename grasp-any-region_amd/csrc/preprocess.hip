// GPU-side image preprocessing: what PerceptionLMImageProcessorFast.resize -> _split ->
// rescale_and_normalize (image_processing_perception_lm_fast.py:268-372) do on the host for every region, on the
// device: the raw uint8 HWC image goes over PCIe once (3 MB for 1024^2 instead of 2 x 41 MB of bf16 tiles) and the
// tiles are written directly in the layout the vision tower reads ([tile, 3, ts, ts], model dtype).
//
//   bicubic + antialias (image):  separable two-pass resampling, horizontal then vertical, fp32, with the per-output
//       tap tables (first tap, tap count, normalised weights) supplied by the host. Accumulation order is the host
//       kernel's: t = src[0]*w[0]; t = fma(src[j], w[j], t) — results are bit-identical to the fp32 path of
//       torch's upsample_bicubic2d_aa, then round-half-even + clamp to [0,255] as torchvision does for uint8 images.
//   nearest (visual-prompt id matrix): index tables from the host.
//   epilogue of both: (v - 255 mean) / (255 std) in fp32 — HF's fused rescale_and_normalize (BaseImageProcessorFast:
//       mean and std are multiplied by 1 / rescale_factor, then ONE subtract and ONE divide; (v / 255 - mean) / std
//       differs from it by one ulp for some uint8 values) —, cast, scatter into tile (yo / ts) * ncw + (xo / ts).
// HBM-bound, tiny next to the vision tower: a 1024^2 region is ~40 MB of tile writes.
#include "common.h"

__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ src, float* __restrict__ tmp, int H,
                                                       int W, int Wout, const int32_t* __restrict__ xmin,
                                                       const int32_t* __restrict__ xsize,
                                                       const float* __restrict__ wx, int kmax) {
    const int xo = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (xo >= Wout) return;
    const int lo = xmin[xo], n = xsize[xo];
    const float* w = wx + (int64_t)xo * kmax;
    const uint8_t* s = src + ((int64_t)y * W + lo) * 3;
    float t0 = 0.f, t1 = 0.f, t2 = 0.f;
    if (n > 0) {
        const float w0 = w[0];
        t0 = (float)s[0] * w0;
        t1 = (float)s[1] * w0;
        t2 = (float)s[2] * w0;
        for (int j = 1; j < n; ++j) {
            const float wj = w[j];
            t0 = __fmaf_rn((float)s[3 * j + 0], wj, t0);
            t1 = __fmaf_rn((float)s[3 * j + 1], wj, t1);
            t2 = __fmaf_rn((float)s[3 * j + 2], wj, t2);
        }
    }
    const int64_t plane = (int64_t)H * Wout;
    const int64_t o = (int64_t)y * Wout + xo;
    tmp[o] = t0;
    tmp[plane + o] = t1;
    tmp[2 * plane + o] = t2;
}

template <typename T>
__device__ __forceinline__ void store_tile_pixel(T* out, float v, int c, int yo, int xo, int ts, int ncw, int tile0,
                                                 float mean, float stdv) {
    const int tile = tile0 + (yo / ts) * ncw + (xo / ts);
    const float nv = (v - mean * 255.0f) / (stdv * 255.0f);
    DT<T>::st(out + (((int64_t)tile * 3 + c) * ts + (yo % ts)) * ts + (xo % ts), nv);
}

template <typename T>
__global__ __launch_bounds__(256) void resize_v_tiles_kernel(const float* __restrict__ tmp, T* __restrict__ out, int H,
                                                             int Wout, int Hout, int ts, int ncw, int tile0,
                                                             const int32_t* __restrict__ ymin,
                                                             const int32_t* __restrict__ ysize,
                                                             const float* __restrict__ wy, int kmax, float mean,
                                                             float stdv) {
    const int xo = blockIdx.x * 256 + threadIdx.x, yo = blockIdx.y, c = blockIdx.z;
    if (xo >= Wout) return;
    const int lo = ymin[yo], n = ysize[yo];
    const float* w = wy + (int64_t)yo * kmax;
    const float* s = tmp + ((int64_t)c * H + lo) * Wout + xo;
    float t = 0.f;
    if (n > 0) {
        t = s[0] * w[0];
        for (int j = 1; j < n; ++j) t = __fmaf_rn(s[(int64_t)j * Wout], w[j], t);
    }
    t = fminf(fmaxf(rintf(t), 0.f), 255.f);             // torch .round() is half-to-even; uint8 clamp
    store_tile_pixel<T>(out, t, c, yo, xo, ts, ncw, tile0, mean, stdv);
}

template <typename T>
__global__ __launch_bounds__(256) void resize_nearest_tiles_kernel(const uint8_t* __restrict__ src, T* __restrict__ out,
                                                                   int W, int Wout, int ts, int ncw, int tile0,
                                                                   const int32_t* __restrict__ xi,
                                                                   const int32_t* __restrict__ yi, float mean,
                                                                   float stdv) {
    const int xo = blockIdx.x * 256 + threadIdx.x, yo = blockIdx.y;
    if (xo >= Wout) return;
    const uint8_t* s = src + ((int64_t)yi[yo] * W + xi[xo]) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) store_tile_pixel<T>(out, (float)s[c], c, yo, xo, ts, ncw, tile0, mean, stdv);
}

extern "C" int gar_resize_bicubic_h(const uint8_t* src, float* tmp, int H, int W, int Wout, const int32_t* xmin,
                                    const int32_t* xsize, const float* wx, int kmax, gar_stream_t stream) {
    GAR_CHECK_ARG(src && tmp && xmin && xsize && wx, "resize_bicubic_h: null pointer");
    GAR_CHECK_ARG(H > 0 && W > 0 && Wout > 0 && kmax > 0 && H <= 65535, "resize_bicubic_h: bad shape");
    hipLaunchKernelGGL(resize_h_kernel, dim3((Wout + 255) / 256, H), dim3(256), 0, (hipStream_t)stream, src, tmp, H, W,
                       Wout, xmin, xsize, wx, kmax);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

extern "C" int gar_resize_bicubic_v_tiles(int dtype, const float* tmp, void* out, int H, int Wout, int Hout, int ts,
                                          int ncw, int tile0, const int32_t* ymin, const int32_t* ysize,
                                          const float* wy, int kmax, float mean, float stdv, gar_stream_t stream) {
    GAR_CHECK_ARG(dtype == GAR_F32 || dtype == GAR_BF16, "resize_bicubic_v_tiles: bad dtype");
    GAR_CHECK_ARG(tmp && out && ymin && ysize && wy, "resize_bicubic_v_tiles: null pointer");
    GAR_CHECK_ARG(H > 0 && Wout > 0 && Hout > 0 && Hout <= 65535 && kmax > 0 && ts > 0 && Wout % ts == 0 &&
                      Hout % ts == 0 && ncw == Wout / ts && stdv != 0.f,
                  "resize_bicubic_v_tiles: bad shape");
    dim3 grid((Wout + 255) / 256, Hout, 3), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GAR_BF16)
        hipLaunchKernelGGL((resize_v_tiles_kernel<bf16_t>), grid, block, 0, s, tmp, (bf16_t*)out, H, Wout, Hout, ts, ncw,
                           tile0, ymin, ysize, wy, kmax, mean, stdv);
    else
        hipLaunchKernelGGL((resize_v_tiles_kernel<float>), grid, block, 0, s, tmp, (float*)out, H, Wout, Hout, ts, ncw,
                           tile0, ymin, ysize, wy, kmax, mean, stdv);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

extern "C" int gar_resize_nearest_tiles(int dtype, const uint8_t* src, void* out, int H, int W, int Hout, int Wout,
                                        int ts, int ncw, int tile0, const int32_t* xi, const int32_t* yi, float mean,
                                        float stdv, gar_stream_t stream) {
    GAR_CHECK_ARG(dtype == GAR_F32 || dtype == GAR_BF16, "resize_nearest_tiles: bad dtype");
    GAR_CHECK_ARG(src && out && xi && yi, "resize_nearest_tiles: null pointer");
    GAR_CHECK_ARG(H > 0 && W > 0 && Wout > 0 && Hout > 0 && Hout <= 65535 && ts > 0 && Wout % ts == 0 &&
                      Hout % ts == 0 && ncw == Wout / ts && stdv != 0.f,
                  "resize_nearest_tiles: bad shape");
    dim3 grid((Wout + 255) / 256, Hout), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GAR_BF16)
        hipLaunchKernelGGL((resize_nearest_tiles_kernel<bf16_t>), grid, block, 0, s, src, (bf16_t*)out, W, Wout, ts, ncw,
                           tile0, xi, yi, mean, stdv);
    else
        hipLaunchKernelGGL((resize_nearest_tiles_kernel<float>), grid, block, 0, s, src, (float*)out, W, Wout, ts, ncw,
                           tile0, xi, yi, mean, stdv);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}
