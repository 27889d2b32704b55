// MCPixelCNN-specific kernels (gfx950): the gated activation with its BatchNorm + MultimodalController
// (forward, and the two-pass backward through the batch statistics), BN -> MC -> residual of `horiz_resid`,
// NHWC im2col / col2im for the 7x7 mask-A layer (its 4x7 and 1x4 stacks then run on the fused 1x1 convolution),
// and the per-pixel cross-entropy over the 512 code classes with its gradient.
// All 3x3-embeddable stacks, the 1x1 convolutions and every weight gradient run on conv_fused.hip / wgrad.hip.
// Reference: models/mcpixelcnn.py (line numbers cited per entry point in include/mcgen_hip.h).
#include "mcgen_common.h"

namespace {
#define STREAM(s) reinterpret_cast<hipStream_t>(s)
inline int grid_for(size_t n, int block = 256, int cap = 4096) {
    size_t b = (n + block - 1) / block; if (b < 1) b = 1; if (b > (size_t)cap) b = cap; return (int)b;
}

// ---- im2col / col2im, NHWC, taps t = i * KW + j at offset (i - oh, j - ow); col channel = t * Cp + c ---------------
template <typename T>
__global__ void im2col_kernel(const T* __restrict__ x, T* __restrict__ col, int N, int H, int W, int Cp, int KH, int KW,
                              int oh, int ow) {
    const int cv = Cp / 8, T_ = KH * KW;
    const size_t total = (size_t)N * H * W * T_ * cv;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cv); size_t r = i / cv;
        const int t = (int)(r % T_); r /= T_;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); const int n = (int)(r / H);
        const int hs = h + t / KW - oh, ws = w + t % KW - ow;
        u32x4 v = {0, 0, 0, 0}; u32x4 v2 = {0, 0, 0, 0};
        const bool in = hs >= 0 && hs < H && ws >= 0 && ws < W;
        T* dst = col + ((((size_t)n * H + h) * W + w) * T_ + t) * Cp + c8 * 8;
        if (in) {
            const T* src = x + (((size_t)n * H + hs) * W + ws) * Cp + c8 * 8;
            v = *reinterpret_cast<const u32x4*>(src);
            if (sizeof(T) == 4) v2 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(src) + 16);
        }
        *reinterpret_cast<u32x4*>(dst) = v;
        if (sizeof(T) == 4) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(dst) + 16) = v2;
    }
}
// dx[n,h,w,c] (+)= sum_t dcol[n, h - dh_t, w - dw_t, t, c]   (gather form: fixed summation order)
template <typename T>
__global__ void col2im_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int N, int H, int W, int Cp, int KH, int KW,
                              int oh, int ow, int accumulate) {
    const int cv = Cp / 8, T_ = KH * KW;
    const size_t total = (size_t)N * H * W * cv;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cv); size_t r = i / cv;
        const int w = (int)(r % W); r /= W;
        const int h = (int)(r % H); const int n = (int)(r / H);
        float acc[8];
        T* dst = dx + (((size_t)n * H + h) * W + w) * Cp + c8 * 8;
        if (accumulate) Elem<T>::load8(dst, acc);
        else {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] = 0.f;
        }
        for (int t = 0; t < T_; ++t) {
            const int ho = h - (t / KW - oh), wo = w - (t % KW - ow);      // the output pixel that read (h, w) through tap t
            if (ho < 0 || ho >= H || wo < 0 || wo >= W) continue;
            float v[8];
            Elem<T>::load8(dcol + ((((size_t)n * H + ho) * W + wo) * T_ + t) * Cp + c8 * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += v[k];
        }
        Elem<T>::store8(dst, acc);
    }
}

// ---- MCGatedActivation (mcpixelcnn.py:16-20): s = [a | b] (2C channels), out = code * relu(a*sc + sh) * sigmoid(b) ----
template <typename T>
__global__ void gated_fwd_kernel(const T* __restrict__ s, const float* __restrict__ sc, const float* __restrict__ sh,
                                 const float* __restrict__ code, T* __restrict__ out, size_t pixels, int HW, int C) {
    const int cv = C / 8;
    const size_t total = pixels * cv;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * 8; const size_t p = i / cv; const size_t n = p / HW;
        float a[8], b[8], k[8], o[8];
        Elem<T>::load8(s + p * 2 * C + c, a);
        Elem<T>::load8(s + p * 2 * C + C + c, b);
        load8f(code + n * C + c, k);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float z = fmaf(a[j], sc[c + j], sh[c + j]);
            o[j] = k[j] * fmaxf(z, 0.f) / (1.f + expf(-b[j]));
        }
        Elem<T>::store8(out + p * C + c, o);
    }
}
// backward pass 1: ds[:, :C] = dz = g * code * q * [z > 0], ds[:, C:] = g * code * relu(z) * q * (1 - q), q = sigmoid(b);
// per-block partial sums of dz and dz * xhat (xhat = (a - mean) * rstd) in the conv-epilogue layout [blocks][2][C]
template <typename T>
__global__ __launch_bounds__(256)
void gated_bwd_stats_kernel(const T* __restrict__ s, const float* __restrict__ sc, const float* __restrict__ sh,
                            const float* __restrict__ mean, const float* __restrict__ rstd, const float* __restrict__ code,
                            const T* __restrict__ g, T* __restrict__ ds, float* __restrict__ part, size_t pixels, int HW, int C,
                            size_t ppb) {
    // thread -> (pixel lane, 8-channel group); C/8 groups per pixel; blockDim = 256
    const int cv = C / 8;
    const int lanes = 256 / cv;                         // pixels handled concurrently (C = 128 -> 16)
    const int grp = threadIdx.x % cv, pl = threadIdx.x / cv;
    const int c = grp * 8;
    const size_t p0 = blockIdx.x * ppb, p1 = (p0 + ppb < pixels) ? p0 + ppb : pixels;
    float s1[8], s2[8], scv[8], shv[8], mv[8], rv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; scv[j] = sc[c + j]; shv[j] = sh[c + j]; mv[j] = mean[c + j]; rv[j] = rstd[c + j]; }
    if (pl < lanes)
        for (size_t p = p0 + pl; p < p1; p += lanes) {
            const size_t n = p / HW;
            float a[8], b[8], k[8], gv[8], dz[8], db[8];
            Elem<T>::load8(s + p * 2 * C + c, a);
            Elem<T>::load8(s + p * 2 * C + C + c, b);
            Elem<T>::load8(g + p * C + c, gv);
            load8f(code + n * C + c, k);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float z = fmaf(a[j], scv[j], shv[j]);
                const float q = 1.f / (1.f + expf(-b[j]));
                const float gk = gv[j] * k[j];
                dz[j] = z > 0.f ? gk * q : 0.f;
                db[j] = gk * fmaxf(z, 0.f) * q * (1.f - q);
                s1[j] += dz[j];
                s2[j] += dz[j] * ((a[j] - mv[j]) * rv[j]);
            }
            Elem<T>::store8(ds + p * 2 * C + c, dz);
            Elem<T>::store8(ds + p * 2 * C + C + c, db);
        }
    __shared__ float red[256][17];
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[threadIdx.x][j] = s1[j]; red[threadIdx.x][8 + j] = s2[j]; }
    __syncthreads();
    for (int cc = threadIdx.x; cc < 2 * C; cc += 256) {
        const int which = cc / C, ch = cc % C;
        float t = 0.f;
        for (int l = 0; l < lanes; ++l) t += red[l * cv + ch / 8][which * 8 + ch % 8];
        part[((size_t)blockIdx.x * 2 + which) * C + ch] = t;
    }
}
// backward pass 2 (in place on the first C channels of ds): da = sc * (dz - (S1 + xhat * S2) / count), sc = gamma * rstd
template <typename T>
__global__ void gated_bwd_apply_kernel(T* __restrict__ ds, const T* __restrict__ s, const float* __restrict__ sums,
                                       const float* __restrict__ sc, const float* __restrict__ mean, const float* __restrict__ rstd,
                                       float inv_count, size_t pixels, int C) {
    const int cv = C / 8;
    const size_t total = pixels * cv;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * 8; const size_t p = i / cv;
        float dz[8], a[8];
        Elem<T>::load8(ds + p * 2 * C + c, dz);
        Elem<T>::load8(s + p * 2 * C + c, a);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (a[j] - mean[c + j]) * rstd[c + j];
            dz[j] = sc[c + j] * (dz[j] - (sums[c + j] + xh * sums[C + c + j]) * inv_count);
        }
        Elem<T>::store8(ds + p * 2 * C + c, dz);
    }
}

// ---- horiz_resid tail (mcpixelcnn.py:37-40,57-60): y = (x*sc + sh) * code (+ res) ---------------------------------------
template <typename T>
__global__ void affine_code_res_kernel(const T* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
                                       const float* __restrict__ code, const T* __restrict__ res, T* __restrict__ y,
                                       size_t pixels, int HW, int C) {
    const int cv = C / 8;
    const size_t total = pixels * cv;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * 8; const size_t p = i / cv; const size_t n = p / HW;
        float v[8], k[8], r[8];
        Elem<T>::load8(x + p * C + c, v);
        load8f(code + n * C + c, k);
        if (res) Elem<T>::load8(res + p * C + c, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], sc[c + j], sh[c + j]) * k[j] + (res ? r[j] : 0.f);
        Elem<T>::store8(y + p * C + c, v);
    }
}
// its backward, pass 1: dz = g * code (written out) and the BN-backward partial sums over x
template <typename T>
__global__ __launch_bounds__(256)
void code_bn_stats_kernel(const T* __restrict__ g, const float* __restrict__ code, const T* __restrict__ x,
                          const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dz_out,
                          float* __restrict__ part, size_t pixels, int HW, int C, size_t ppb) {
    const int cv = C / 8;
    const int lanes = 256 / cv;
    const int grp = threadIdx.x % cv, pl = threadIdx.x / cv;
    const int c = grp * 8;
    const size_t p0 = blockIdx.x * ppb, p1 = (p0 + ppb < pixels) ? p0 + ppb : pixels;
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
    if (pl < lanes)
        for (size_t p = p0 + pl; p < p1; p += lanes) {
            const size_t n = p / HW;
            float gv[8], k[8], xv[8];
            Elem<T>::load8(g + p * C + c, gv);
            Elem<T>::load8(x + p * C + c, xv);
            load8f(code + n * C + c, k);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                gv[j] *= k[j];
                s1[j] += gv[j];
                s2[j] += gv[j] * ((xv[j] - mean[c + j]) * rstd[c + j]);
            }
            Elem<T>::store8(dz_out + p * C + c, gv);
        }
    __shared__ float red[256][17];
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[threadIdx.x][j] = s1[j]; red[threadIdx.x][8 + j] = s2[j]; }
    __syncthreads();
    for (int cc = threadIdx.x; cc < 2 * C; cc += 256) {
        const int which = cc / C, ch = cc % C;
        float t = 0.f;
        for (int l = 0; l < lanes; ++l) t += red[l * cv + ch / 8][which * 8 + ch % 8];
        part[((size_t)blockIdx.x * 2 + which) * C + ch] = t;
    }
}

// ---- cross-entropy over channels (mcpixelcnn.py:100): one wave per pixel ------------------------------------------------
// loss_rows[p] = logsumexp(logits[p, :]) - logits[p, target[p]];  dlogits[p, c] = (softmax - onehot) * gscale
template <typename T>
__global__ __launch_bounds__(256)
void ce_kernel(const T* __restrict__ logits, const int64_t* __restrict__ target, float* __restrict__ loss_rows,
               T* __restrict__ dlogits, float gscale, size_t pixels, int C, int Cp) {
    const int lane = threadIdx.x & 63;
    const size_t p = blockIdx.x * (size_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= pixels) return;
    const T* row = logits + p * Cp;
    float m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmaxf(m, Elem<T>::to_f(row[c]));
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float se = 0.f;
    for (int c = lane; c < C; c += 64) se += expf(Elem<T>::to_f(row[c]) - m);
    for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
    const int tgt = (int)target[p];
    const float lse = m + logf(se);
    if (lane == 0) loss_rows[p] = lse - Elem<T>::to_f(row[tgt]);
    if (dlogits) {
        T* drow = dlogits + p * Cp;
        for (int c = lane; c < Cp; c += 64) {
            float v = 0.f;
            if (c < C) v = (expf(Elem<T>::to_f(row[c]) - lse) - (c == tgt ? 1.f : 0.f)) * gscale;
            drow[c] = Elem<T>::from_f(v);
        }
    }
}
}  // namespace

#define DISPATCH_T(dtype, F32, BF16) \
    do { if ((dtype) == MCGEN_F32) { F32; } else if ((dtype) == MCGEN_BF16) { BF16; } else return mcgen_fail("bad dtype %d", (dtype)); } while (0)

extern "C" int mcgen_im2col(const void* x, void* col, int dtype, int N, int H, int W, int Cp, int KH, int KW, int oh, int ow, void* stream) {
    MCGEN_CHECK(x && col && N > 0 && H > 0 && W > 0 && Cp % 8 == 0 && KH > 0 && KW > 0, "im2col: bad arguments");
    const size_t total = (size_t)N * H * W * KH * KW * (Cp / 8);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(im2col_kernel<float>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const float*)x, (float*)col, N, H, W, Cp, KH, KW, oh, ow),
        hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const bf16_t*)x, (bf16_t*)col, N, H, W, Cp, KH, KW, oh, ow));
    MCGEN_LAUNCH_CHECK("im2col"); return 0;
}
extern "C" int mcgen_col2im(const void* dcol, void* dx, int dtype, int N, int H, int W, int Cp, int KH, int KW, int oh, int ow,
                            int accumulate, void* stream) {
    MCGEN_CHECK(dcol && dx && N > 0 && H > 0 && W > 0 && Cp % 8 == 0 && KH > 0 && KW > 0, "col2im: bad arguments");
    const size_t total = (size_t)N * H * W * (Cp / 8);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(col2im_kernel<float>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const float*)dcol, (float*)dx, N, H, W, Cp, KH, KW, oh, ow, accumulate),
        hipLaunchKernelGGL(col2im_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const bf16_t*)dcol, (bf16_t*)dx, N, H, W, Cp, KH, KW, oh, ow, accumulate));
    MCGEN_LAUNCH_CHECK("col2im"); return 0;
}
extern "C" int mcgen_gated_fwd(const void* s, const float* scale, const float* shift, const float* code, void* out, int dtype,
                               int N, int HW, int C, void* stream) {
    MCGEN_CHECK(s && scale && shift && code && out && C % 8 == 0 && N > 0 && HW > 0, "gated_fwd: bad arguments");
    const size_t pixels = (size_t)N * HW, total = pixels * (C / 8);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(gated_fwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const float*)s, scale, shift, code, (float*)out, pixels, HW, C),
        hipLaunchKernelGGL(gated_fwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const bf16_t*)s, scale, shift, code, (bf16_t*)out, pixels, HW, C));
    MCGEN_LAUNCH_CHECK("gated_fwd"); return 0;
}
extern "C" int mcgen_gated_bwd_stats(const void* s, const float* scale, const float* shift, const float* mean, const float* rstd,
                                     const float* code, const void* g, void* ds, float* partials, int blocks, int dtype,
                                     int N, int HW, int C, void* stream) {
    MCGEN_CHECK(s && scale && shift && mean && rstd && code && g && ds && partials && blocks > 0 && N > 0 && HW > 0, "gated_bwd_stats: bad arguments");
    MCGEN_CHECK(C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0, "gated_bwd_stats: C/8 must divide 256");
    const size_t pixels = (size_t)N * HW;
    const size_t ppb = (pixels + blocks - 1) / blocks;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(gated_bwd_stats_kernel<float>, dim3(blocks), dim3(256), 0, STREAM(stream), (const float*)s, scale, shift, mean, rstd, code, (const float*)g, (float*)ds, partials, pixels, HW, C, ppb),
        hipLaunchKernelGGL(gated_bwd_stats_kernel<bf16_t>, dim3(blocks), dim3(256), 0, STREAM(stream), (const bf16_t*)s, scale, shift, mean, rstd, code, (const bf16_t*)g, (bf16_t*)ds, partials, pixels, HW, C, ppb));
    MCGEN_LAUNCH_CHECK("gated_bwd_stats"); return 0;
}
extern "C" int mcgen_gated_bwd_apply(void* ds, const void* s, const float* sums, const float* scale, const float* mean,
                                     const float* rstd, double count, int dtype, int64_t pixels, int C, void* stream) {
    MCGEN_CHECK(ds && s && sums && scale && mean && rstd && count > 0 && C % 8 == 0 && pixels > 0, "gated_bwd_apply: bad arguments");
    const size_t total = (size_t)pixels * (C / 8);
    const float inv = (float)(1.0 / count);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(gated_bwd_apply_kernel<float>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (float*)ds, (const float*)s, sums, scale, mean, rstd, inv, (size_t)pixels, C),
        hipLaunchKernelGGL(gated_bwd_apply_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (bf16_t*)ds, (const bf16_t*)s, sums, scale, mean, rstd, inv, (size_t)pixels, C));
    MCGEN_LAUNCH_CHECK("gated_bwd_apply"); return 0;
}
extern "C" int mcgen_affine_code_res(const void* x, const float* scale, const float* shift, const float* code, const void* res,
                                     void* y, int dtype, int N, int HW, int C, void* stream) {
    MCGEN_CHECK(x && scale && shift && code && y && C % 8 == 0 && N > 0 && HW > 0, "affine_code_res: bad arguments");
    const size_t pixels = (size_t)N * HW, total = pixels * (C / 8);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(affine_code_res_kernel<float>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const float*)x, scale, shift, code, (const float*)res, (float*)y, pixels, HW, C),
        hipLaunchKernelGGL(affine_code_res_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const bf16_t*)x, scale, shift, code, (const bf16_t*)res, (bf16_t*)y, pixels, HW, C));
    MCGEN_LAUNCH_CHECK("affine_code_res"); return 0;
}
extern "C" int mcgen_code_bn_stats(const void* g, const float* code, const void* x, const float* mean, const float* rstd,
                                   void* dz, float* partials, int blocks, int dtype, int N, int HW, int C, void* stream) {
    MCGEN_CHECK(g && code && x && mean && rstd && dz && partials && blocks > 0 && N > 0 && HW > 0, "code_bn_stats: bad arguments");
    MCGEN_CHECK(C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0, "code_bn_stats: C/8 must divide 256");
    const size_t pixels = (size_t)N * HW;
    const size_t ppb = (pixels + blocks - 1) / blocks;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(code_bn_stats_kernel<float>, dim3(blocks), dim3(256), 0, STREAM(stream), (const float*)g, code, (const float*)x, mean, rstd, (float*)dz, partials, pixels, HW, C, ppb),
        hipLaunchKernelGGL(code_bn_stats_kernel<bf16_t>, dim3(blocks), dim3(256), 0, STREAM(stream), (const bf16_t*)g, code, (const bf16_t*)x, mean, rstd, (bf16_t*)dz, partials, pixels, HW, C, ppb));
    MCGEN_LAUNCH_CHECK("code_bn_stats"); return 0;
}
extern "C" int mcgen_cross_entropy(const void* logits, const int64_t* target, float* loss_rows, void* dlogits, float gscale,
                                   int dtype, int64_t pixels, int C, int Cp, void* stream) {
    MCGEN_CHECK(logits && target && loss_rows && pixels > 0 && C > 0 && Cp >= C, "cross_entropy: bad arguments");
    const int blocks = (int)((pixels + 3) / 4);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(ce_kernel<float>, dim3(blocks), dim3(256), 0, STREAM(stream), (const float*)logits, target, loss_rows, (float*)dlogits, gscale, (size_t)pixels, C, Cp),
        hipLaunchKernelGGL(ce_kernel<bf16_t>, dim3(blocks), dim3(256), 0, STREAM(stream), (const bf16_t*)logits, target, loss_rows, (bf16_t*)dlogits, gscale, (size_t)pixels, C, Cp));
    MCGEN_LAUNCH_CHECK("cross_entropy"); return 0;
}
