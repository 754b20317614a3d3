/* CPU ORACLE — TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * Plain-C restatement of torchvision.ops.roi_align's CPU kernel (torchvision/csrc/ops/cpu/roi_align_kernel.cpp:
 * pre_calc_for_bilinear_interpolate + roi_align_forward_kernel_impl), the op the reference calls at
 * projects/grasp_any_region/hf_models/modeling_gar.py:389-396 (also modeling_perception_lm.py:720,809).
 * torchvision is not vendored in /root/reference and not installed here: PARITY UNPINNED against the real
 * kernel; this file is an independent second restatement that pins oracle/gar_oracle.py::roi_align bit for bit.
 * Build with -ffp-contract=off so a*b+c is two roundings, as on the reference's CPU path. */
#include <math.h>
#include <stddef.h>

static void bilinear(const float* fm, int C, int H, int W, float y, float x, float* acc) {
    if (y < -1.0f || y > (float)H || x < -1.0f || x > (float)W) return;
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else yh = yl + 1;
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else xh = xl + 1;
    float ly = y - (float)yl, lx = x - (float)xl, hy = 1.0f - ly, hx = 1.0f - lx;
    float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
    for (int c = 0; c < C; ++c) {
        const float* f = fm + (size_t)c * H * W;
        acc[c] = acc[c] + (((w1 * f[yl * W + xl] + w2 * f[yl * W + xh]) + w3 * f[yh * W + xl]) + w4 * f[yh * W + xh]);
    }
}

/* inp [N,C,H,W], rois [K,5], out [K,C,ph,pw]; acc = scratch [C] */
void roi_align_ref(const float* inp, int C, int H, int W, const float* rois, int K, int ph_n, int pw_n,
                   float spatial_scale, int sampling_ratio, int aligned, float* out, float* acc) {
    for (int n = 0; n < K; ++n) {
        const float* r = rois + n * 5;
        const float* fm = inp + (size_t)((int)r[0]) * C * H * W;
        float off = aligned ? 0.5f : 0.0f;
        float sw = r[1] * spatial_scale - off, sh = r[2] * spatial_scale - off;
        float ew = r[3] * spatial_scale - off, eh = r[4] * spatial_scale - off;
        float rw = ew - sw, rh = eh - sh;
        if (!aligned) { rw = fmaxf(rw, 1.0f); rh = fmaxf(rh, 1.0f); }
        float bh = rh / (float)ph_n, bw = rw / (float)pw_n;
        int gh = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / ph_n);
        int gw = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / pw_n);
        float count = (float)(gh * gw > 1 ? gh * gw : 1);
        for (int ph = 0; ph < ph_n; ++ph)
            for (int pw = 0; pw < pw_n; ++pw) {
                for (int c = 0; c < C; ++c) acc[c] = 0.f;
                for (int iy = 0; iy < gh; ++iy) {
                    float y = sh + (float)ph * bh + ((float)iy + .5f) * bh / (float)gh;
                    for (int ix = 0; ix < gw; ++ix) {
                        float x = sw + (float)pw * bw + ((float)ix + .5f) * bw / (float)gw;
                        bilinear(fm, C, H, W, y, x, acc);
                    }
                }
                for (int c = 0; c < C; ++c) out[(((size_t)n * C + c) * ph_n + ph) * pw_n + pw] = acc[c] / count;
            }
    }
}
