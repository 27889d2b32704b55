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

// ---- im2col / col2im, NHWC, taps t = i * KW + j; col channel = t * Cp + c -------------------------------------------
// col[n, ho, wo, t, c] = X[n, ho * stride + i - oh, wo * stride + j - ow, c] (0 outside), where X is x after the optional
// prologue relu?(x * scale + shift) * code  (so that zero padding applies to the ACTIVATED tensor, as in the reference).
template <typename T>
__global__ void im2col_kernel(const T* __restrict__ x, T* __restrict__ col, int N, int H, int W, int Cp, int KH, int KW,
                              int oh, int ow, int stride, int Ho, int Wo, const float* __restrict__ scale,
                              const float* __restrict__ shift, int relu, const float* __restrict__ code) {
    const int cv = Cp / 8, T_ = KH * KW;
    const size_t total = (size_t)N * Ho * Wo * T_ * cv;
    const bool plain = !scale && !relu && !code;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % cv); size_t r = i / cv;
        const int t = (int)(r % T_); r /= T_;
        const int w = (int)(r % Wo); r /= Wo;
        const int h = (int)(r % Ho); const int n = (int)(r / Ho);
        const int hs = h * stride + t / KW - oh, ws = w * stride + t % KW - ow;
        const bool in = hs >= 0 && hs < H && ws >= 0 && ws < W;
        T* dst = col + ((((size_t)n * Ho + h) * Wo + w) * T_ + t) * Cp + c8 * 8;
        const T* src = x + (((size_t)n * H + hs) * W + ws) * Cp + c8 * 8;
        if (plain) {
            u32x4 v = {0, 0, 0, 0}; u32x4 v2 = {0, 0, 0, 0};
            if (in) {
                v = *reinterpret_cast<const u32x4*>(src);
                if (sizeof(T) == 4) v2 = *reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(src) + 16);
            }
            *reinterpret_cast<u32x4*>(dst) = v;
            if (sizeof(T) == 4) *reinterpret_cast<u32x4*>(reinterpret_cast<char*>(dst) + 16) = v2;
        } else {
            float v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = 0.f;
            if (in) {
                Elem<T>::load8(src, v);
                const int c = c8 * 8;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    float z = scale ? fmaf(v[k], scale[c + k], shift[c + k]) : v[k];
                    if (relu) z = fmaxf(z, 0.f);
                    if (code) z *= code[(size_t)n * Cp + c + k];
                    v[k] = z;
                }
            }
            Elem<T>::store8(dst, v);
        }
    }
}
// adjoint (gather form, fixed summation order): dx[n,h,w,c] (+)= bias[c] + sum_t dcol[n, (h+oh-i)/s, (w+ow-j)/s, t, c]
// over the taps whose source position is integral and inside the [Ho, Wo] column grid.  With stride 2, 4x4 taps and
// oh = ow = 1 this IS nn.ConvTranspose2d(.., 4, 2, 1) applied to dcol = x @ W (mcvae.py:89,95).
template <typename T>
__global__ void col2im_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int N, int H, int W, int Cp, int KH, int KW,
                              int oh, int ow, int stride, int Ho, int Wo, const float* __restrict__ bias, int C, int accumulate) {
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
            for (int k = 0; k < 8; ++k) acc[k] = (bias && c8 * 8 + k < C) ? bias[c8 * 8 + k] : 0.f;
        }
        for (int t = 0; t < T_; ++t) {
            const int hn = h + oh - t / KW, wn = w + ow - t % KW;
            if (hn < 0 || wn < 0 || hn % stride || wn % stride) continue;
            const int ho = hn / stride, wo = wn / stride;
            if (ho >= Ho || wo >= Wo) continue;
            float v[8];
            Elem<T>::load8(dcol + ((((size_t)n * Ho + ho) * Wo + wo) * T_ + t) * Cp + c8 * 8, v);
#pragma unroll
            for (int k = 0; k < 8; ++k) acc[k] += v[k];
        }
        Elem<T>::store8(dst, acc);
    }
}

// ---- MCGatedActivation (mcpixelcnn.py:16-20): s = [a | b] (2C channels), out = code * relu(a*sc + sh) * sigmoid(b) ----
template <typename T>
__device__ __forceinline__ void gated_fwd_body(const T* __restrict__ s, const float* __restrict__ sc, const float* __restrict__ sh,
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
template <typename T>
__global__ void gated_fwd_kernel(const T* __restrict__ s, const float* __restrict__ sc, const float* __restrict__ sh,
                                 const float* __restrict__ code, T* __restrict__ out, size_t pixels, int HW, int C) {
    gated_fwd_body<T>(s, sc, sh, code, out, pixels, HW, C);
}
// the vertical and the horizontal gate of a layer (independent of each other) in one launch: blockIdx.y = gate
struct GatedJobs { mcgen_gated_t j[MCGEN_GATED_MAX]; };
template <typename T>
__global__ void gated_fwd_batch_kernel(const GatedJobs jobs) {
    const mcgen_gated_t& j = jobs.j[blockIdx.y];
    gated_fwd_body<T>(reinterpret_cast<const T*>(j.s), j.scale, j.shift, j.code, reinterpret_cast<T*>(j.out), (size_t)j.N * j.HW, j.HW, j.C);
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
                                       size_t pixels, int HW, int C, int pre_relu, int post_relu) {
    const int cv = C / 8;
    const size_t total = pixels * cv;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * 8; const size_t p = i / cv; const size_t n = p / HW;
        float v[8], k[8], r[8];
        Elem<T>::load8(x + p * C + c, v);
        if (code) load8f(code + n * C + c, k);
        if (res) Elem<T>::load8(res + p * C + c, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float z = fmaf(v[j], sc[c + j], sh[c + j]);
            if (pre_relu) z = fmaxf(z, 0.f);
            if (code) z *= k[j];
            if (res) z += r[j];
            v[j] = post_relu ? fmaxf(z, 0.f) : z;
        }
        Elem<T>::store8(y + p * C + c, v);
    }
}
// ---- classifier block tail (models/classifier.py:17-29): y = MaxPool2d(2)(relu(x * sc + sh)), eval-mode BatchNorm as an affine --------
template <typename T>
__global__ void affine_relu_maxpool2_kernel(const T* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
                                            T* __restrict__ y, int N, int Ho, int Wo, int C) {
    const int cv = C / 8;
    const size_t total = (size_t)N * Ho * Wo * cv;
    const int W = 2 * Wo;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cv) * 8; size_t t = i / cv;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho); const size_t n = t / Ho;
        float a[8], b[8], o[8];
        load8f(sc + c, a); load8f(sh + c, b);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = 0.f;                           // max over relu(.) >= 0
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[8];
            Elem<T>::load8(x + (((n * 2 * Ho + 2 * ho + (q >> 1)) * W) + 2 * wo + (q & 1)) * C + c, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = fmaxf(o[j], fmaf(v[j], a[j], b[j]));
        }
        Elem<T>::store8(y + i * 8, o);
    }
}
// its backward, pass 1: dz = g * code (written out) and the BN-backward partial sums over x
// gates: y_post != NULL -> g *= [y_post > 0] (the tail ended in a ReLU; the gated g is also the residual's gradient,
// written to g_gated when given); pre_relu -> dz *= [x * sc + sh > 0]
template <typename T>
__global__ __launch_bounds__(256)
void code_bn_stats_kernel(const T* __restrict__ g, const float* __restrict__ code, const T* __restrict__ x,
                          const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dz_out,
                          float* __restrict__ part, size_t pixels, int HW, int C, size_t ppb,
                          const float* __restrict__ sc, const float* __restrict__ sh, int pre_relu,
                          const T* __restrict__ y_post, T* __restrict__ g_gated) {
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
            float gv[8], k[8], xv[8], yv[8];
            Elem<T>::load8(g + p * C + c, gv);
            Elem<T>::load8(x + p * C + c, xv);
            if (code) load8f(code + n * C + c, k);
            if (y_post) {
                Elem<T>::load8(y_post + p * C + c, yv);
#pragma unroll
                for (int j = 0; j < 8; ++j) gv[j] = yv[j] > 0.f ? gv[j] : 0.f;
                if (g_gated) Elem<T>::store8(g_gated + p * C + c, gv);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (code) gv[j] *= k[j];
                if (pre_relu && !(fmaf(xv[j], sc[c + j], sh[c + j]) > 0.f)) gv[j] = 0.f;
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

// ---- VAE reconstruction loss (mcvae.py:10-14): recon = sigmoid(a); BCE(recon, t) summed; d a = (recon - t) * gscale ---------
// log terms clamped at -100 as F.binary_cross_entropy does; per-block partial sums in fixed order.
template <typename T>
__global__ __launch_bounds__(256)
void bce_kernel(const T* __restrict__ a, const float* __restrict__ t, T* __restrict__ recon, T* __restrict__ da,
                float* __restrict__ part, float gscale, size_t pixels, int C, int Cp) {
    __shared__ float red[4];
    const size_t total = pixels * Cp;
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cp);
        float r = 0.f, d = 0.f;
        if (c < C) {
            const float av = Elem<T>::to_f(a[i]), tv = t[i];
            r = 1.f / (1.f + expf(-av));
            const float sp_pos = fmaxf(av, 0.f) + log1pf(expf(-fabsf(av)));     // softplus(a)  = -log(1 - sigmoid(a))
            const float sp_neg = sp_pos - av;                                    // softplus(-a) = -log(sigmoid(a))
            s += tv * fminf(sp_neg, 100.f) + (1.f - tv) * fminf(sp_pos, 100.f);
            d = (r - tv) * gscale;
        }
        recon[i] = Elem<T>::from_f(r);
        if (da) da[i] = Elem<T>::from_f(d);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
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
// ---- arg-min over channels (VectorQuantization's nearest code, modules.py:25): one wave per pixel, first minimum wins ---
template <typename T>
__global__ __launch_bounds__(256)
void argmin_kernel(const T* __restrict__ x, int64_t* __restrict__ idx, size_t pixels, int C, int Cp) {
    const int lane = threadIdx.x & 63;
    const size_t p = blockIdx.x * (size_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
    if (p >= pixels) return;
    const T* row = x + p * Cp;
    float best = INFINITY; int bi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) { const float v = Elem<T>::to_f(row[c]); if (v < best) { best = v; bi = c; } }
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o); const int oi = __shfl_xor(bi, o);
        if (ov < best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) idx[p] = bi;
}
}  // namespace

#define DISPATCH_T(dtype, F32, BF16) \
    do { if ((dtype) == MCGEN_F32) { F32; } else if ((dtype) == MCGEN_BF16) { BF16; } else return mcgen_fail("bad dtype %d", (dtype)); } while (0)

extern "C" int mcgen_im2col(const void* x, void* col, int dtype, int N, int H, int W, int Cp, int KH, int KW, int oh, int ow,
                            int stride, const float* scale, const float* shift, int relu, const float* code, void* stream) {
    MCGEN_CHECK(x && col && N > 0 && H > 0 && W > 0 && Cp % 8 == 0 && KH > 0 && KW > 0 && stride >= 1 && H % stride == 0 && W % stride == 0,
                "im2col: bad arguments");
    MCGEN_CHECK(!scale == !shift, "im2col: scale and shift come together");
    const int Ho = H / stride, Wo = W / stride;
    const size_t total = (size_t)N * Ho * Wo * KH * KW * (Cp / 8);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(im2col_kernel<float>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const float*)x, (float*)col, N, H, W, Cp, KH, KW, oh, ow, stride, Ho, Wo, scale, shift, relu, code),
        hipLaunchKernelGGL(im2col_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const bf16_t*)x, (bf16_t*)col, N, H, W, Cp, KH, KW, oh, ow, stride, Ho, Wo, scale, shift, relu, code));
    MCGEN_LAUNCH_CHECK("im2col"); return 0;
}
extern "C" int mcgen_col2im(const void* dcol, void* dx, int dtype, int N, int H, int W, int Cp, int KH, int KW, int oh, int ow,
                            int stride, const float* bias, int C, int accumulate, void* stream) {
    MCGEN_CHECK(dcol && dx && N > 0 && H > 0 && W > 0 && Cp % 8 == 0 && KH > 0 && KW > 0 && stride >= 1 && H % stride == 0 && W % stride == 0,
                "col2im: bad arguments");
    const int Ho = H / stride, Wo = W / stride;
    const size_t total = (size_t)N * H * W * (Cp / 8);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(col2im_kernel<float>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const float*)dcol, (float*)dx, N, H, W, Cp, KH, KW, oh, ow, stride, Ho, Wo, bias, C, accumulate),
        hipLaunchKernelGGL(col2im_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const bf16_t*)dcol, (bf16_t*)dx, N, H, W, Cp, KH, KW, oh, ow, stride, Ho, Wo, bias, C, accumulate));
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
extern "C" int mcgen_gated_fwd_batch(const mcgen_gated_t* jobs, int n, int dtype, void* stream) {
    MCGEN_CHECK(jobs && n >= 1 && n <= MCGEN_GATED_MAX, "gated_fwd_batch: 1 .. %d gates", MCGEN_GATED_MAX);
    GatedJobs t; size_t most = 1;
    for (int i = 0; i < n; ++i) {
        const mcgen_gated_t& j = jobs[i];
        MCGEN_CHECK(j.s && j.scale && j.shift && j.code && j.out && j.C % 8 == 0 && j.N > 0 && j.HW > 0, "gated_fwd_batch: bad job %d", i);
        t.j[i] = j;
        const size_t total = (size_t)j.N * j.HW * (j.C / 8);
        if (total > most) most = total;
    }
    for (int i = n; i < MCGEN_GATED_MAX; ++i) t.j[i] = jobs[0];
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(gated_fwd_batch_kernel<float>, dim3(grid_for(most), n), dim3(256), 0, STREAM(stream), t),
        hipLaunchKernelGGL(gated_fwd_batch_kernel<bf16_t>, dim3(grid_for(most), n), dim3(256), 0, STREAM(stream), t));
    MCGEN_LAUNCH_CHECK("gated_fwd_batch"); return 0;
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
                                     void* y, int dtype, int N, int HW, int C, int pre_relu, int post_relu, void* stream) {
    MCGEN_CHECK(x && scale && shift && y && C % 8 == 0 && N > 0 && HW > 0, "affine_code_res: bad arguments");
    const size_t pixels = (size_t)N * HW, total = pixels * (C / 8);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(affine_code_res_kernel<float>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const float*)x, scale, shift, code, (const float*)res, (float*)y, pixels, HW, C, pre_relu, post_relu),
        hipLaunchKernelGGL(affine_code_res_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const bf16_t*)x, scale, shift, code, (const bf16_t*)res, (bf16_t*)y, pixels, HW, C, pre_relu, post_relu));
    MCGEN_LAUNCH_CHECK("affine_code_res"); return 0;
}
extern "C" int mcgen_affine_relu_maxpool2(const void* x, const float* scale, const float* shift, void* y, int dtype,
                                          int N, int Ho, int Wo, int C, void* stream) {
    MCGEN_CHECK(x && scale && shift && y && N > 0 && Ho > 0 && Wo > 0 && C > 0 && C % 8 == 0, "affine_relu_maxpool2: bad arguments (C a multiple of 8)");
    const size_t total = (size_t)N * Ho * Wo * (C / 8);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(affine_relu_maxpool2_kernel<float>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const float*)x, scale, shift, (float*)y, N, Ho, Wo, C),
        hipLaunchKernelGGL(affine_relu_maxpool2_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const bf16_t*)x, scale, shift, (bf16_t*)y, N, Ho, Wo, C));
    MCGEN_LAUNCH_CHECK("affine_relu_maxpool2"); return 0;
}
extern "C" int mcgen_code_bn_stats(const void* g, const float* code, const void* x, const float* mean, const float* rstd,
                                   void* dz, float* partials, int blocks, int dtype, int N, int HW, int C,
                                   const float* scale, const float* shift, int pre_relu, const void* y_post, void* g_gated,
                                   void* stream) {
    MCGEN_CHECK(g && x && mean && rstd && dz && partials && blocks > 0 && N > 0 && HW > 0, "code_bn_stats: bad arguments");
    MCGEN_CHECK(!pre_relu || (scale && shift), "code_bn_stats: the pre-ReLU gate needs the BatchNorm affine");
    MCGEN_CHECK(C % 8 == 0 && C / 8 <= 256 && 256 % (C / 8) == 0, "code_bn_stats: C/8 must divide 256");
    const size_t pixels = (size_t)N * HW;
    const size_t ppb = (pixels + blocks - 1) / blocks;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(code_bn_stats_kernel<float>, dim3(blocks), dim3(256), 0, STREAM(stream), (const float*)g, code, (const float*)x, mean, rstd, (float*)dz, partials, pixels, HW, C, ppb, scale, shift, pre_relu, (const float*)y_post, (float*)g_gated),
        hipLaunchKernelGGL(code_bn_stats_kernel<bf16_t>, dim3(blocks), dim3(256), 0, STREAM(stream), (const bf16_t*)g, code, (const bf16_t*)x, mean, rstd, (bf16_t*)dz, partials, pixels, HW, C, ppb, scale, shift, pre_relu, (const bf16_t*)y_post, (bf16_t*)g_gated));
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
extern "C" int mcgen_bce_logits(const void* logits, const float* target, void* recon, void* dlogits, float* partials, int blocks,
                                float gscale, int dtype, int64_t pixels, int C, int Cp, void* stream) {
    MCGEN_CHECK(logits && target && recon && partials && blocks > 0 && blocks <= 4096 && pixels > 0 && C > 0 && Cp >= C, "bce_logits: bad arguments");
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(bce_kernel<float>, dim3(blocks), dim3(256), 0, STREAM(stream), (const float*)logits, target, (float*)recon, (float*)dlogits, partials, gscale, (size_t)pixels, C, Cp),
        hipLaunchKernelGGL(bce_kernel<bf16_t>, dim3(blocks), dim3(256), 0, STREAM(stream), (const bf16_t*)logits, target, (bf16_t*)recon, (bf16_t*)dlogits, partials, gscale, (size_t)pixels, C, Cp));
    MCGEN_LAUNCH_CHECK("bce_logits"); return 0;
}
extern "C" int mcgen_argmin_channels(const void* x, int64_t* idx, int dtype, int64_t pixels, int C, int Cp, void* stream) {
    MCGEN_CHECK(x && idx && pixels > 0 && C > 0 && Cp >= C, "argmin_channels: bad arguments");
    const int blocks = (int)((pixels + 3) / 4);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(argmin_kernel<float>, dim3(blocks), dim3(256), 0, STREAM(stream), (const float*)x, idx, (size_t)pixels, C, Cp),
        hipLaunchKernelGGL(argmin_kernel<bf16_t>, dim3(blocks), dim3(256), 0, STREAM(stream), (const bf16_t*)x, idx, (size_t)pixels, C, Cp));
    MCGEN_LAUNCH_CHECK("argmin_channels"); return 0;
}
