// RoI-aligned feature replay and the sequence assembly around it (compiled with -ffp-contract=off: the RoI
// arithmetic follows torchvision's roi_align CPU kernel op for op in fp32, no fused multiply-add).
//
//   placeholder_scan  image-token ranks + crop-token spans, device-side (replaces the .item()/numel() host syncs of
//                     modeling_gar.py:356-360 and modeling_perception_lm.py:299-315)
//   embed_assemble    nn.Embedding gather + masked_scatter in one pass
//   roi_replay        _merge + .float() + roi_align + permute/flatten/cast + torch.cat splice as ONE kernel that
//                     reads <= 4 corner cells per sample straight from the tile-major pooled features and writes
//                     the P*P replay rows in place. One block per output token, lanes over channels (16-B vectors).
//                     Algorithmic bytes per crop token: P*P*C*sizeof(T) written + <= 16 cells * C * sizeof(T) read.
#include "common.h"

// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void placeholder_scan_kernel(const int64_t* __restrict__ ids, int S,
                                                                int64_t image_id, const int64_t* __restrict__ crop_ids,
                                                                int n_crop, int32_t* __restrict__ slot,
                                                                int32_t* __restrict__ counts,
                                                                int32_t* __restrict__ spans,
                                                                int32_t* __restrict__ rank_pos, int rank_stride) {
    __shared__ int wave_tot[16];
    __shared__ int smin[8], smax[8];
    __shared__ int running;
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t* row = ids + (int64_t)b * S;
    if (tid < 8) { smin[tid] = 0x7fffffff; smax[tid] = -1; }
    if (tid == 0) running = 0;
    int64_t cid[8];
    for (int c = 0; c < 8; ++c) cid[c] = c < n_crop ? crop_ids[c] : (int64_t)-0x7fffffffffffLL;
    int lmin[8], lmax[8];
    for (int c = 0; c < 8; ++c) { lmin[c] = 0x7fffffff; lmax[c] = -1; }
    __syncthreads();
    for (int base = 0; base < S; base += 1024) {
        const int s = base + tid;
        const int64_t v = s < S ? row[s] : (int64_t)-1;
        const bool flag = s < S && v == image_id;
        const unsigned long long m = __ballot(flag);
        const int prefix = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(m);
        __syncthreads();
        int off = running;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (s < S) {
            slot[(int64_t)b * S + s] = flag ? off + prefix : -1;
            // inverse map (rank of an image token -> its position): lets the replay read pooled features straight from
            // the assembled sequence wherever the placeholders sit
            if (rank_pos && flag && off + prefix < rank_stride) rank_pos[(int64_t)b * rank_stride + off + prefix] = s;
#pragma unroll
            for (int c = 0; c < 8; ++c)
                if (v == cid[c]) { lmin[c] = min(lmin[c], s); lmax[c] = max(lmax[c], s); }
        }
        __syncthreads();
        if (tid == 0) {
            int tot = 0;
            for (int w = 0; w < 16; ++w) tot += wave_tot[w];
            running += tot;
        }
        __syncthreads();
    }
    for (int c = 0; c < n_crop; ++c) {
        if (lmax[c] >= 0) { atomicMin(&smin[c], lmin[c]); atomicMax(&smax[c], lmax[c]); }
    }
    __syncthreads();
    if (tid == 0) counts[b] = running;
    if (tid < n_crop) {
        spans[((int64_t)b * n_crop + tid) * 2 + 0] = smax[tid] >= 0 ? smin[tid] : -1;
        spans[((int64_t)b * n_crop + tid) * 2 + 1] = smax[tid];
    }
}

extern "C" int gar_placeholder_scan(const int64_t* input_ids, int B, int S, int64_t image_token_id,
                                    const int64_t* crop_ids, int n_crop, int32_t* slot, int32_t* counts, int32_t* spans,
                                    int32_t* rank_pos, int rank_stride, gar_stream_t stream) {
    GAR_CHECK_ARG(input_ids && slot && counts && spans && B > 0 && S > 0, "placeholder_scan: bad args");
    GAR_CHECK_ARG(!rank_pos || rank_stride > 0, "placeholder_scan: rank_pos without rank_stride");
    GAR_CHECK_ARG(n_crop >= 0 && n_crop <= 8 && (n_crop == 0 || crop_ids), "placeholder_scan: n_crop must be <= 8");
    hipLaunchKernelGGL(placeholder_scan_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, input_ids, S,
                       image_token_id, crop_ids, n_crop, slot, counts, spans, rank_pos, rank_stride);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// The reference's input checks of generate() evaluated on the device (GARModel.generate(validate=False) skips the host
// syncs, not the checks): image-token count != feature rows (modeling_perception_lm.py:309-315), a crop-token span that
// is not P*P long (the splice of modeling_gar.py:404-411 would change the sequence length), a crop token present in
// input_ids without a bbox (modeling_gar.py:366: KeyError), ids outside [0, vocab). ORs GAR_INPUT_* bits into flags[0].
// One block per sample; has_box[b] = bit c set iff sample b carries a bbox for crop token c.
__global__ __launch_bounds__(256) void input_check_kernel(const int64_t* __restrict__ ids, int S, int64_t vocab,
                                                          const int32_t* __restrict__ counts, int n_rows,
                                                          const int32_t* __restrict__ spans, int n_crop, int span_len,
                                                          const int32_t* __restrict__ has_box, int32_t* __restrict__ flags,
                                                          const uint8_t* __restrict__ attn_mask) {
    const int b = blockIdx.x;
    int bad = 0;
    const int64_t* row = ids + (int64_t)b * S;
    for (int s = threadIdx.x; s < S; s += 256) {
        const int64_t v = row[s];
        if (v < 0 || v >= vocab) bad |= 8;
    }
    if (attn_mask) {
        // a generation mask has to be LEFT-padded, 0...01...1 (HF's convention; generate() derives left_pad from its zero
        // count): a 1 followed by a 0, or a 0 in the last column, is a row that would be continued after its padding
        const uint8_t* mr = attn_mask + (int64_t)b * S;
        for (int s = threadIdx.x; s < S; s += 256) {
            const bool cur = mr[s] != 0;
            const bool nxt = s + 1 < S ? mr[s + 1] != 0 : true;
            if (cur && !nxt) bad |= 16;
            if (s == S - 1 && !cur) bad |= 16;
        }
    }
    if (counts && threadIdx.x == 0 && counts[b] != n_rows) bad |= 1;
    if (spans && has_box && (int)threadIdx.x < n_crop) {
        const int lo = spans[((int64_t)b * n_crop + threadIdx.x) * 2], hi = spans[((int64_t)b * n_crop + threadIdx.x) * 2 + 1];
        const bool present = hi >= 0, box = (has_box[b] >> threadIdx.x) & 1;
        if (present && box && hi - lo + 1 != span_len) bad |= 2;
        if (present && !box) bad |= 4;
    }
    if (__any(bad != 0) && bad) atomicOr(flags, bad);
}

extern "C" int gar_input_check(const int64_t* input_ids, int B, int S, int64_t vocab, const int32_t* counts, int n_rows,
                               const int32_t* spans, int n_crop, int span_len, const int32_t* has_box, int32_t* flags,
                               const uint8_t* attn_mask, gar_stream_t stream) {
    // counts / (spans, has_box) NULL: that group of checks is skipped (a text-only prompt has no placeholders to count)
    GAR_CHECK_ARG(input_ids && flags && B > 0 && S > 0 && n_crop >= 0 && n_crop <= 8 && (!spans == !has_box),
                  "input_check: bad args");
    hipLaunchKernelGGL(input_check_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, input_ids, S, vocab, counts, n_rows,
                       spans, n_crop, span_len, has_box, flags, attn_mask);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void embed_assemble_kernel(const int64_t* __restrict__ ids,
                                                             const int32_t* __restrict__ slot, const T* __restrict__ E,
                                                             const T* __restrict__ feats, T* __restrict__ out, int S,
                                                             int C, int64_t n_feat_rows, int64_t vocab) {
    const int64_t r = blockIdx.x;                         // row over B*S
    const int b = (int)(r / S);
    const int32_t sl = slot ? slot[r] : -1;
    const T* src;
    if (sl >= 0) {
        src = feats + ((int64_t)b * n_feat_rows + min((int64_t)sl, n_feat_rows - 1)) * C;
    } else {
        int64_t id = ids[r];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        src = E + id * C;
    }
    T* dst = out + r * C;
    for (int i = threadIdx.x * 8; i < C; i += 256 * 8) {
        if (sizeof(T) == 2) *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(src + i);
        else {
            reinterpret_cast<float4*>(dst + i)[0] = reinterpret_cast<const float4*>(src + i)[0];
            reinterpret_cast<float4*>(dst + i)[1] = reinterpret_cast<const float4*>(src + i)[1];
        }
    }
}

extern "C" int gar_embed_assemble(int dtype, const int64_t* input_ids, const int32_t* slot, const void* E,
                                  const void* feats, void* out, int B, int S, int C, int64_t n_feat_rows, int64_t vocab,
                                  gar_stream_t stream) {
    GAR_CHECK_ARG(input_ids && E && out && B > 0 && S > 0 && C % 8 == 0, "embed_assemble: bad args");
    GAR_CHECK_ARG(!slot || (feats && n_feat_rows > 0), "embed_assemble: slot without feats");
    dim3 grid((unsigned)((int64_t)B * S)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GAR_BF16)
        hipLaunchKernelGGL((embed_assemble_kernel<bf16_t>), grid, block, 0, s, input_ids, slot, (const bf16_t*)E,
                           (const bf16_t*)feats, (bf16_t*)out, S, C, n_feat_rows, vocab);
    else
        hipLaunchKernelGGL((embed_assemble_kernel<float>), grid, block, 0, s, input_ids, slot, (const float*)E,
                           (const float*)feats, (float*)out, S, C, n_feat_rows, vocab);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// pool_assemble: PerceptionLMAdaptiveAvgPooling (2x2 mean, modeling_perception_lm.py:47-60) + nn.Embedding +
// masked_scatter (modeling_gar.py:332,341-346) in ONE pass: an image-token row of the sequence is the 2x2 mean of four
// projector-output rows (rounded once to the storage dtype, as the pool's output is), any other row is its embedding.
// The pooled features are never written on their own: the RoI replay below reads them back from the sequence rows
// (rank_pos). Algorithmic bytes per region: projector grid rows read once (T*g*g*C) + the sequence written once (S*C)
// = SURVEY.md section 8d's ~90 MB at GAR-1B / 1024^2. One block per sequence row, 8 channels per thread.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pool_assemble_kernel(const int64_t* __restrict__ ids,
                                                            const int32_t* __restrict__ slot, const T* __restrict__ E,
                                                            const T* __restrict__ proj, T* __restrict__ out, int S, int C,
                                                            int64_t n_feat_rows, int64_t vocab, int g,
                                                            int in_tile_tokens, int in_token_offset,
                                                            int64_t proj_rows_per_sample) {
    const int64_t r = blockIdx.x;                         // row over B*S
    const int b = (int)(r / S);
    const int32_t sl = slot[r];
    T* dst = out + r * C;
    if (sl >= 0) {
        const int go = g >> 1;
        const int64_t rk = min((int64_t)sl, n_feat_rows - 1);
        const int t = (int)(rk / (go * go)), k = (int)(rk - (int64_t)t * go * go);
        const int oy = k / go, ox = k - oy * go;
        const T* base = proj + ((int64_t)b * proj_rows_per_sample + (int64_t)t * in_tile_tokens + in_token_offset) * C;
        const T* p00 = base + (int64_t)((2 * oy) * g + 2 * ox) * C;
        for (int i = threadIdx.x * 8; i < C; i += 256 * 8) {
            float a[8], bb[8], c[8], d[8], o[8];
            ld8(p00 + i, a);
            ld8(p00 + C + i, bb);
            ld8(p00 + (int64_t)g * C + i, c);
            ld8(p00 + (int64_t)(g + 1) * C + i, d);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (((a[e] + bb[e]) + c[e]) + d[e]) * 0.25f;      // pool2x2_kernel's order
            st8(dst + i, o);
        }
    } else {
        int64_t id = ids[r];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const T* src = E + id * C;
        for (int i = threadIdx.x * 8; i < C; i += 256 * 8) {
            if (sizeof(T) == 2) *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(src + i);
            else {
                reinterpret_cast<float4*>(dst + i)[0] = reinterpret_cast<const float4*>(src + i)[0];
                reinterpret_cast<float4*>(dst + i)[1] = reinterpret_cast<const float4*>(src + i)[1];
            }
        }
    }
}

extern "C" int gar_pool_assemble(int dtype, const int64_t* input_ids, const int32_t* slot, const void* E,
                                 const void* proj, void* out, int B, int S, int C, int tiles_per_sample, int g,
                                 int in_tile_tokens, int in_token_offset, int64_t vocab, gar_stream_t stream) {
    GAR_CHECK_ARG(input_ids && slot && E && proj && out && B > 0 && S > 0 && C % 8 == 0, "pool_assemble: bad args");
    GAR_CHECK_ARG(tiles_per_sample > 0 && g > 0 && g % 2 == 0, "pool_assemble: bad grid");
    if (in_tile_tokens <= 0) in_tile_tokens = g * g;
    GAR_CHECK_ARG(in_token_offset >= 0 && in_token_offset + g * g <= in_tile_tokens, "pool_assemble: bad token window");
    const int64_t n_feat_rows = (int64_t)tiles_per_sample * (g / 2) * (g / 2);
    const int64_t proj_rows = (int64_t)tiles_per_sample * in_tile_tokens;
    dim3 grid((unsigned)((int64_t)B * S)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GAR_BF16)
        hipLaunchKernelGGL((pool_assemble_kernel<bf16_t>), grid, block, 0, s, input_ids, slot, (const bf16_t*)E,
                           (const bf16_t*)proj, (bf16_t*)out, S, C, n_feat_rows, vocab, g, in_tile_tokens,
                           in_token_offset, proj_rows);
    else
        hipLaunchKernelGGL((pool_assemble_kernel<float>), grid, block, 0, s, input_ids, slot, (const float*)E,
                           (const float*)proj, (float*)out, S, C, n_feat_rows, vocab, g, in_tile_tokens, in_token_offset,
                           proj_rows);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// roi_replay
// ---------------------------------------------------------------------------------------------------------------
struct Sample { int yl, yh, xl, xh; float w1, w2, w3, w4; bool valid; };

// torchvision pre_calc_for_bilinear_interpolate, fp32
__device__ __forceinline__ Sample make_sample(float y, float x, int H, int W) {
    Sample s;
    s.valid = !(y < -1.0f || y > (float)H || x < -1.0f || x > (float)W);
    if (y <= 0.f) y = 0.f;
    if (x <= 0.f) x = 0.f;
    int yl = (int)y, xl = (int)x, yh, xh;
    if (yl >= H - 1) { yh = yl = H - 1; y = (float)yl; } else { yh = yl + 1; }
    if (xl >= W - 1) { xh = xl = W - 1; x = (float)xl; } else { xh = xl + 1; }
    const float ly = y - (float)yl, lx = x - (float)xl;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    s.yl = yl; s.yh = yh; s.xl = xl; s.xh = xh;
    s.w1 = hy * hx; s.w2 = hy * lx; s.w3 = ly * hx; s.w4 = ly * lx;
    if (!s.valid) { s.yl = s.yh = s.xl = s.xh = 0; s.w1 = s.w2 = s.w3 = s.w4 = 0.f; }
    return s;
}

// One axis of torchvision's pre_calc_for_bilinear_interpolate (fp32): low / high cell, the two lerp weights and
// whether the coordinate is inside [-1, size]. The sample's 4 corner weights are products of a y pair and an x pair.
struct Axis { int lo, hi; float l, h; bool valid; };
__device__ __forceinline__ Axis make_axis(float v, int size) {
    Axis a;
    a.valid = !(v < -1.0f || v > (float)size);
    if (v <= 0.f) v = 0.f;
    int lo = (int)v, hi;
    if (lo >= size - 1) { hi = lo = size - 1; v = (float)lo; } else { hi = lo + 1; }
    a.lo = lo; a.hi = hi;
    a.l = v - (float)lo;
    a.h = 1.0f - a.l;
    if (!a.valid) { a.lo = a.hi = 0; }
    return a;
}

template <typename T> struct Raw8;                       // 8 channels of one map cell, as loaded
template <> struct Raw8<bf16_t> {
    uint4 v;
    __device__ __forceinline__ void load(const bf16_t* p) { v = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void get(float (&o)[8]) const {
        o[0] = unpk_lo(v.x); o[1] = unpk_hi(v.x);
        o[2] = unpk_lo(v.y); o[3] = unpk_hi(v.y);
        o[4] = unpk_lo(v.z); o[5] = unpk_hi(v.z);
        o[6] = unpk_lo(v.w); o[7] = unpk_hi(v.w);
    }
};
template <> struct Raw8<float> {
    float4 a, b;
    __device__ __forceinline__ void load(const float* p) {
        a = reinterpret_cast<const float4*>(p)[0];
        b = reinterpret_cast<const float4*>(p)[1];
    }
    __device__ __forceinline__ void get(float (&o)[8]) const {
        o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
    }
};

// One output row (bin = ph*P + pw) of one crop token, channels c0, c0+cstride, ... of it.
// The G*G samples of a bin use cells (Y[a], X[b]) with Y = {lo,hi of sample row 0, lo,hi of sample row 1} and X
// likewise: the block loads each DISTINCT cell once (typically 2 x 2 instead of 16 — the bin is smaller than a
// cell for every realistic mask) and then runs torchvision's arithmetic in its original order on the copies.
template <typename T, int G>
__device__ __forceinline__ void roi_replay_row(const T* feats, T* embeds, int head, int ph,
                                               int pw, int c0, int cstride, int first_tile, int ncw, int nch, int P,
                                               int C, int S, float rx1, float ry1, float rx2, float ry2, float ss,
                                               int aligned, const int32_t* __restrict__ rank_pos = nullptr) {
    static_assert(G == 2, "sampling_ratio 2");
    const int row = head + ph * P + pw;
    if (row >= S) return;
    const int H = nch * P, W = ncw * P;
    const float off = aligned ? 0.5f : 0.0f;
    const float sw = rx1 * ss - off, sh = ry1 * ss - off, ew = rx2 * ss - off, eh = ry2 * ss - off;
    float rw = ew - sw, rh = eh - sh;
    if (!aligned) { rw = fmaxf(rw, 1.0f); rh = fmaxf(rh, 1.0f); }
    const float bh = rh / (float)P, bw = rw / (float)P;
    const float count = (float)(G * G);
    Axis ay[G], ax[G];
#pragma unroll
    for (int i = 0; i < G; ++i) {
        ay[i] = make_axis(sh + (float)ph * bh + ((float)i + 0.5f) * bh / (float)G, H);
        ax[i] = make_axis(sw + (float)pw * bw + ((float)i + 0.5f) * bw / (float)G, W);
    }
    // an invalid sample (either axis outside [-1, size]) contributes exactly 0: its weights are zeroed below and its
    // cells are clamped to (0, 0) on the invalid axis, as pre_calc_for_bilinear_interpolate does
    const int Y[4] = {ay[0].lo, ay[0].hi, ay[1].lo, ay[1].hi};
    const int X[4] = {ax[0].lo, ax[0].hi, ax[1].lo, ax[1].hi};
    auto cell = [&](int y, int x) -> const T* {            // merged-map cell (y,x) in the tile-major layout
        const int tile = first_tile + (y / P) * ncw + (x / P);
        const int64_t rank = (int64_t)tile * P * P + (y % P) * P + (x % P);
        // in-place form: `feats` is the assembled sequence of this sample and the pooled token of rank r sits in row rank_pos[r]
        return feats + (rank_pos ? (int64_t)rank_pos[rank] : rank) * C;
    };
    T* dst = embeds + (int64_t)row * C;
    // Common case — a bin no larger than a map cell, so both samples of an axis share their (lo, hi) pair: 2 x 2 distinct
    // cells. All channel passes of the row (C / cstride: 4 at C = 2048, 8 at C = 4096) then issue their loads together,
    // 4 x 16 bytes per lane and pass, before the first blend — a wave has one dependent round trip instead of one per pass
    // (a CU holds 16 waves, so the memory-level parallelism of this latency-bound kernel has to come from inside the wave:
    // 128 jobs x 256 bins at C = 4096 ran at 0.38 of HBM with one pass in flight). Same cells, same arithmetic order.
    if (Y[2] == Y[0] && Y[3] == Y[1] && X[2] == X[0] && X[3] == X[1]) {
        const T* p00 = cell(Y[0], X[0]);
        const T* p01 = cell(Y[0], X[1]);
        const T* p10 = cell(Y[1], X[0]);
        const T* p11 = cell(Y[1], X[1]);
        constexpr int UN = 4;
        for (int cb = c0; cb < C; cb += UN * cstride) {
            Raw8<T> q[UN][4];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int c = cb + u * cstride;
                if (c < C) {
                    q[u][0].load(p00 + c);
                    q[u][1].load(p01 + c);
                    q[u][2].load(p10 + c);
                    q[u][3].load(p11 + c);
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int c = cb + u * cstride;
                if (c < C) {
                    float v1[8], v2[8], v3[8], v4[8], acc[8];
                    q[u][0].get(v1);
                    q[u][1].get(v2);
                    q[u][2].get(v3);
                    q[u][3].get(v4);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
                    for (int iy = 0; iy < G; ++iy) {
#pragma unroll
                        for (int ix = 0; ix < G; ++ix) {
                            const bool valid = ay[iy].valid && ax[ix].valid;
                            const float w1 = valid ? ay[iy].h * ax[ix].h : 0.f, w2 = valid ? ay[iy].h * ax[ix].l : 0.f;
                            const float w3 = valid ? ay[iy].l * ax[ix].h : 0.f, w4 = valid ? ay[iy].l * ax[ix].l : 0.f;
#pragma unroll
                            for (int e = 0; e < 8; ++e)
                                acc[e] = acc[e] + (((w1 * v1[e] + w2 * v2[e]) + w3 * v3[e]) + w4 * v4[e]);
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[e] = acc[e] / count;
                    st8(dst + c, acc);
                }
            }
        }
        return;
    }
    for (int c = c0; c < C; c += cstride) {
        Raw8<T> g[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            int ra = -1;                                    // earlier identical row (block-uniform)
#pragma unroll
            for (int a2 = 0; a2 < 4; ++a2)
                if (a2 < a && ra < 0 && Y[a2] == Y[a]) ra = a2;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                int cb = -1;
#pragma unroll
                for (int b2 = 0; b2 < 4; ++b2)
                    if (b2 < b && cb < 0 && X[b2] == X[b]) cb = b2;
                if (ra >= 0) {
#pragma unroll
                    for (int a2 = 0; a2 < 4; ++a2)
                        if (a2 == ra) g[a][b] = g[a2][b];
                } else if (cb >= 0) {
#pragma unroll
                    for (int b2 = 0; b2 < 4; ++b2)
                        if (b2 == cb) g[a][b] = g[a][b2];
                } else {
                    g[a][b].load(cell(Y[a], X[b]) + c);
                }
            }
        }
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int iy = 0; iy < G; ++iy) {
#pragma unroll
            for (int ix = 0; ix < G; ++ix) {
                const bool valid = ay[iy].valid && ax[ix].valid;
                const float w1 = valid ? ay[iy].h * ax[ix].h : 0.f, w2 = valid ? ay[iy].h * ax[ix].l : 0.f;
                const float w3 = valid ? ay[iy].l * ax[ix].h : 0.f, w4 = valid ? ay[iy].l * ax[ix].l : 0.f;
                float v1[8], v2[8], v3[8], v4[8];
                g[2 * iy][2 * ix].get(v1);
                g[2 * iy][2 * ix + 1].get(v2);
                g[2 * iy + 1][2 * ix].get(v3);
                g[2 * iy + 1][2 * ix + 1].get(v4);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    acc[e] = acc[e] + (((w1 * v1[e] + w2 * v2[e]) + w3 * v3[e]) + w4 * v4[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = acc[e] / count;
        st8(dst + c, acc);
    }
}

// Work split: ONE WAVE per output row (bin), lanes cover the channels 8 at a time (4 passes at C = 2048). Measured at
// 64 crop tokens (71 MB): 256-thread blocks per bin 120 us, a block per map row with a wave per bin 96 us, this
// 78 us — the kernel is bound by the dependent-access chain (job -> span -> cells -> store) of tiny rows, not by HBM;
// the pass it belongs to (pool2x2 -> embed_assemble -> replay) runs at ~5.3 TB/s (tools/bench_replay.py).
#define RB 64
template <typename T, int G>
__global__ __launch_bounds__(RB) void roi_replay_kernel(const T* __restrict__ feats, T* __restrict__ embeds,
                                                        const int32_t* __restrict__ spans, int crop_index,
                                                        int first_tile, int ncw, int nch, int P, int C, int S,
                                                        float rx1, float ry1, float rx2, float ry2, float ss,
                                                        int aligned) {
    const int head = spans[2 * crop_index];
    if (head < 0) return;                                   // crop token absent from input_ids
    roi_replay_row<T, G>(feats, embeds, head, blockIdx.x / P, blockIdx.x % P, threadIdx.x * 8, RB * 8, first_tile, ncw,
                         nch, P, C, S, rx1, ry1, rx2, ry2, ss, aligned);
}

// every crop token of every sample of a batch in ONE launch: blockIdx.y = job
template <typename T, int G>
__global__ __launch_bounds__(RB) void roi_replay_batched_kernel(const T* __restrict__ feats, T* __restrict__ embeds,
                                                                const int32_t* __restrict__ spans,
                                                                const gar_roi_job* __restrict__ jobs, int n_crop,
                                                                int tiles_per_sample, int P, int C, int S,
                                                                int aligned) {
    const gar_roi_job j = jobs[blockIdx.y];
    const int head = spans[((int64_t)j.sample * n_crop + j.crop_index) * 2];
    if (head < 0) return;
    roi_replay_row<T, G>(feats + (int64_t)j.sample * tiles_per_sample * P * P * C, embeds + (int64_t)j.sample * S * C,
                         head, blockIdx.x / P, blockIdx.x % P, threadIdx.x * 8, RB * 8, j.first_tile, j.ncw, j.nch, P, C,
                         S, j.x1, j.y1, j.x2, j.y2, j.spatial_scale, aligned);
}

// in-place form: the pooled features are the image-token rows of `embeds` itself (written by pool_assemble); rank_pos
// [B, rank_stride] maps the rank of a pooled token (tile * P*P + token) to its sequence position. The crop-token rows it
// writes are disjoint from the image-token rows it reads.
template <typename T, int G>
__global__ __launch_bounds__(RB) void roi_replay_inplace_kernel(T* __restrict__ embeds, const int32_t* __restrict__ spans,
                                                                const int32_t* __restrict__ rank_pos, int rank_stride,
                                                                const gar_roi_job* __restrict__ jobs, int n_crop, int P,
                                                                int C, int S, int aligned) {
    const gar_roi_job j = jobs[blockIdx.y];
    const int head = spans[((int64_t)j.sample * n_crop + j.crop_index) * 2];
    if (head < 0) return;
    T* seq = embeds + (int64_t)j.sample * S * C;
    roi_replay_row<T, G>(seq, seq, head, blockIdx.x / P, blockIdx.x % P, threadIdx.x * 8, RB * 8, j.first_tile, j.ncw,
                         j.nch, P, C, S, j.x1, j.y1, j.x2, j.y2, j.spatial_scale, aligned,
                         rank_pos + (int64_t)j.sample * rank_stride);
}

extern "C" int gar_roi_replay_inplace(int dtype, void* embeds, const int32_t* spans, const int32_t* rank_pos,
                                      int rank_stride, const gar_roi_job* jobs, int n_jobs, int n_crop, int P, int C,
                                      int S, int sampling_ratio, int aligned, gar_stream_t stream) {
    GAR_CHECK_ARG(embeds && spans && jobs && rank_pos, "roi_replay_inplace: null pointer");
    GAR_CHECK_ARG(n_jobs > 0 && n_jobs <= 65535 && n_crop > 0 && rank_stride > 0 && P > 0 && C % 8 == 0 && S > 0,
                  "roi_replay_inplace: bad shape");
    GAR_CHECK_ARG(sampling_ratio == 2, "roi_replay_inplace: sampling_ratio %d not built (the reference uses 2)",
                  sampling_ratio);
    dim3 grid(P * P, n_jobs), block(RB);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GAR_BF16)
        hipLaunchKernelGGL((roi_replay_inplace_kernel<bf16_t, 2>), grid, block, 0, s, (bf16_t*)embeds, spans, rank_pos,
                           rank_stride, jobs, n_crop, P, C, S, aligned);
    else
        hipLaunchKernelGGL((roi_replay_inplace_kernel<float, 2>), grid, block, 0, s, (float*)embeds, spans, rank_pos,
                           rank_stride, jobs, n_crop, P, C, S, aligned);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

extern "C" int gar_roi_replay(int dtype, const void* feats, void* embeds, const int32_t* spans, int crop_index,
                              int first_tile, int ncw, int nch, int P, int C, int S, float rx1, float ry1, float rx2,
                              float ry2, float spatial_scale, int sampling_ratio, int aligned, gar_stream_t stream) {
    GAR_CHECK_ARG(feats && embeds && spans, "roi_replay: null pointer");
    GAR_CHECK_ARG(ncw > 0 && nch > 0 && P > 0 && C % 8 == 0 && S > 0 && crop_index >= 0, "roi_replay: bad shape");
    GAR_CHECK_ARG(sampling_ratio == 2, "roi_replay: sampling_ratio %d not built (the reference uses 2)", sampling_ratio);
    dim3 grid(P * P), block(RB);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GAR_BF16)
        hipLaunchKernelGGL((roi_replay_kernel<bf16_t, 2>), grid, block, 0, s, (const bf16_t*)feats, (bf16_t*)embeds, spans,
                           crop_index, first_tile, ncw, nch, P, C, S, rx1, ry1, rx2, ry2, spatial_scale, aligned);
    else
        hipLaunchKernelGGL((roi_replay_kernel<float, 2>), grid, block, 0, s, (const float*)feats, (float*)embeds, spans,
                           crop_index, first_tile, ncw, nch, P, C, S, rx1, ry1, rx2, ry2, spatial_scale, aligned);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

extern "C" int gar_roi_replay_batched(int dtype, const void* feats, void* embeds, const int32_t* spans,
                                      const gar_roi_job* jobs, int n_jobs, int n_crop, int tiles_per_sample, int P, int C,
                                      int S, int sampling_ratio, int aligned, gar_stream_t stream) {
    GAR_CHECK_ARG(feats && embeds && spans && jobs, "roi_replay_batched: null pointer");
    GAR_CHECK_ARG(n_jobs > 0 && n_jobs <= 65535 && n_crop > 0 && tiles_per_sample > 0 && P > 0 && C % 8 == 0 && S > 0,
                  "roi_replay_batched: bad shape");
    GAR_CHECK_ARG(sampling_ratio == 2, "roi_replay_batched: sampling_ratio %d not built (the reference uses 2)",
                  sampling_ratio);
    dim3 grid(P * P, n_jobs), block(RB);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GAR_BF16)
        hipLaunchKernelGGL((roi_replay_batched_kernel<bf16_t, 2>), grid, block, 0, s, (const bf16_t*)feats,
                           (bf16_t*)embeds, spans, jobs, n_crop, tiles_per_sample, P, C, S, aligned);
    else
        hipLaunchKernelGGL((roi_replay_batched_kernel<float, 2>), grid, block, 0, s, (const float*)feats, (float*)embeds,
                           spans, jobs, n_crop, tiles_per_sample, P, C, S, aligned);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}
