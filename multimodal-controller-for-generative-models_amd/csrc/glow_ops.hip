// MCGlow-specific kernels (gfx950): squeeze / unsqueeze, channel statistics + ActNorm data-dependent
// initialisation, LU-parameterised invertible 1x1 convolution weight (and its inverse), affine coupling
// forward / reverse with per-sample log-determinants, Gaussian prior log-density / sampling.
// The 3x3 / 1x1 convolutions of the coupling networks run on the fused convolution (conv_fused.hip) with the
// ActNorm affine + ReLU + MultimodalController code as its prologue.
// Reference: models/mcglow.py (line numbers cited per kernel in include/mcgen_hip.h).
#include "mcgen_common.h"

namespace {
#define STREAM(s) reinterpret_cast<hipStream_t>(s)
inline int grid_for(size_t n, int block = 256, int cap = 4096) {
    size_t b = (n + block - 1) / block; if (b < 1) b = 1; if (b > (size_t)cap) b = cap; return (int)b;
}

__device__ float block_sum_g(float v, float* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = blockDim.x >> 6;
    __syncthreads();
    if (l == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// ---- squeeze: [N, H, W, C] -> [N, H/2, W/2, 4C], channel c*4 + 2*dh + dw  (mcglow.py:221-223) -------------
template <typename T>
__global__ void squeeze_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W, int C, int Cpi, int Cpo, int inverse) {
    const int Ho = H / 2, Wo = W / 2;
    const size_t total = (size_t)N * Ho * Wo * Cpo;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = (int)(i % Cpo); size_t t = i / Cpo;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho); const int n = (int)(t / Ho);
        const int c = co >> 2, dh = (co >> 1) & 1, dw = co & 1;
        const size_t big = (((size_t)n * H + 2 * ho + dh) * W + 2 * wo + dw) * Cpi + c;     // index in the unsqueezed tensor
        if (!inverse) y[i] = (co < 4 * C) ? x[big] : Elem<T>::from_f(0.f);
        else if (co < 4 * C) y[big] = x[i];                                                  // x squeezed -> y unsqueezed
    }
}
template <typename T>
__global__ void zero_pad_channels_kernel(T* __restrict__ y, size_t pixels, int C, int Cp) {
    const size_t total = pixels * (Cp - C);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        y[(i / (Cp - C)) * Cp + C + i % (Cp - C)] = Elem<T>::from_f(0.f);
}

// ---- per-channel sum / sum of squares, partials in the conv-epilogue format [blocks][2][Cp] ------------------
template <typename T>
__global__ void channel_stats_kernel(const T* __restrict__ x, size_t pixels, int Cp, float* __restrict__ part, size_t ppb) {
    const size_t p0 = blockIdx.x * ppb, p1 = (p0 + ppb < pixels) ? p0 + ppb : pixels;
    for (int c = threadIdx.x; c < Cp; c += blockDim.x) {
        float s1 = 0.f, s2 = 0.f;
        for (size_t p = p0; p < p1; ++p) { const float v = Elem<T>::to_f(x[p * Cp + c]); s1 += v; s2 += v * v; }
        part[((size_t)blockIdx.x * 2 + 0) * Cp + c] = s1;
        part[((size_t)blockIdx.x * 2 + 1) * Cp + c] = s2;
    }
}
// ActNorm.initialize (mcglow.py:32-39): loc = -mean, scale = 1 / (unbiased std + 1e-6)
__global__ void actnorm_init_kernel(const float* __restrict__ part, int tiles, int pitch, int C, double count,
                                    float* loc, float* scale) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s1 = 0.0, s2 = 0.0;
    for (int t = 0; t < tiles; ++t) { s1 += (double)part[((size_t)t * 2) * pitch + c]; s2 += (double)part[((size_t)t * 2 + 1) * pitch + c]; }
    const double mean = s1 / count;
    double var = (s2 - s1 * mean) / (count > 1.0 ? count - 1.0 : 1.0); if (var < 0.0) var = 0.0;
    loc[c] = (float)(-mean);
    scale[c] = (float)(1.0 / (sqrt(var) + 1e-6));
}
// prologue vectors of the op that consumes an ActNorm: y = s * (x + loc) = x * s + s * loc
__device__ __forceinline__ void actnorm_affine_body(const float* loc, const float* scale, int C, int Cp, float* a, float* b, float* negloc) {
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < Cp; c += gridDim.x * blockDim.x) {
        a[c] = c < C ? scale[c] : 0.f;
        b[c] = c < C ? scale[c] * loc[c] : 0.f;
        if (negloc) negloc[c] = c < C ? -loc[c] : 0.f;       // "mean" of the backward gate: x - mean = x + loc
    }
}
__global__ void actnorm_affine_kernel(const float* loc, const float* scale, int C, int Cp, float* a, float* b, float* negloc) {
    actnorm_affine_body(loc, scale, C, Cp, a, b, negloc);
}
// Batched forms (blockIdx.y = job, the job table travels by value in the kernel arguments: graph-capture safe).  MCGlow
// at K = 16, L = 3 runs 144 ActNorms, 48 LU weights and 51 ZeroConv2d scales per step: as one launch each per KIND instead
// of one per module -- these launches were ~700 of the step's 1400, 3-6 us apiece.
struct AnAffineJobs { mcgen_an_affine_t j[MCGEN_GLOW_BATCH_MAX]; };
__global__ void actnorm_affine_batch_kernel(const AnAffineJobs jobs) {
    const mcgen_an_affine_t& j = jobs.j[blockIdx.y];
    actnorm_affine_body(j.loc, j.scale, j.C, j.Cp, j.a, j.b, j.negloc);
}
struct PldJobs { mcgen_pld_t j[MCGEN_GLOW_PLD_MAX]; };
// all flows' parameter-only log-determinants in one launch: logdet[n] += sum_f hw_f * (sum log|scale_f| + sum w_s_f)
__global__ void glow_param_logdet_batch_kernel(const PldJobs jobs, int njobs, float* __restrict__ logdet, int N) {
    __shared__ float red[32];
    float s = 0.f;
    // (unrolled: the loads of eight flows go out together -- 48 flows one after the other were 48 dependent round trips, 34 us)
#pragma unroll 8
    for (int f = 0; f < njobs; ++f) {
        const mcgen_pld_t& j = jobs.j[f];
        float t = 0.f;
        for (int i = threadIdx.x; i < j.C; i += blockDim.x) t += logf(fabsf(j.scale[i]));
        for (int i = threadIdx.x; i < j.Cw; i += blockDim.x) t += j.w_s[i];
        s = fmaf(j.hw, t, s);
    }
    s = block_sum_g(s, red);
    for (int n = threadIdx.x; n < N; n += blockDim.x) logdet[n] += s;
}
// parameter-only log-determinant of a flow (mcglow.py:46-47,101): logdet[n] += HW * (sum log|scale| + sum w_s), one launch
__global__ void glow_param_logdet_kernel(const float* __restrict__ scale, int C, const float* __restrict__ ws, int Cw, float hw,
                                         float* __restrict__ logdet, int N) {
    __shared__ float red[32];
    float s = 0.f;
    for (int i = threadIdx.x; i < C; i += blockDim.x) s += logf(fabsf(scale[i]));
    for (int i = threadIdx.x; i < Cw; i += blockDim.x) s += ws[i];
    s = block_sum_g(s, red);
    for (int n = threadIdx.x; n < N; n += blockDim.x) logdet[n] += hw * s;
}

// ---- InvConv2dLU.calc_weight (mcglow.py:105-111) and its inverse, one workgroup, C <= 64 -----------------------
// W = P (L o l_mask + I) (U o u_mask + diag(s_sign * exp(w_s)))
__device__ void invconv_weight_body(const float* __restrict__ wp, const float* __restrict__ wl, const float* __restrict__ wu,
                                    const float* __restrict__ ws, const float* __restrict__ ssign, int C,
                                    float* __restrict__ W, float* __restrict__ Winv) {
    extern __shared__ float sh[];
    float* Lm = sh; float* Um = sh + C * C; float* A = sh + 2 * C * C; float* B = sh + 3 * C * C;
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) {
        const int r = i / C, c = i % C;
        Lm[i] = (r > c ? wl[i] : 0.f) + (r == c ? 1.f : 0.f);
        Um[i] = (c > r ? wu[i] : 0.f) + (r == c ? ssign[r] * expf(ws[r]) : 0.f);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) {          // A = L U
        const int r = i / C, c = i % C;
        float s = 0.f;
        for (int k = 0; k < C; ++k) s = fmaf(Lm[r * C + k], Um[k * C + c], s);
        A[i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) {          // W = P A
        const int r = i / C, c = i % C;
        float s = 0.f;
        for (int k = 0; k < C; ++k) s = fmaf(wp[r * C + k], A[k * C + c], s);
        B[i] = s; W[i] = s;
    }
    if (!Winv) return;
    __syncthreads();
    // Gauss-Jordan with partial pivoting on [B | I] -> [I | B^-1]; A is reused as the right-hand side
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) A[i] = (i / C == i % C) ? 1.f : 0.f;
    __shared__ int piv;
    for (int col = 0; col < C; ++col) {
        __syncthreads();
        if (threadIdx.x == 0) {
            int best = col; float bv = fabsf(B[col * C + col]);
            for (int r = col + 1; r < C; ++r) { const float v = fabsf(B[r * C + col]); if (v > bv) { bv = v; best = r; } }
            piv = best;
        }
        __syncthreads();
        if (piv != col)
            for (int c = threadIdx.x; c < C; c += blockDim.x) {
                float t = B[col * C + c]; B[col * C + c] = B[piv * C + c]; B[piv * C + c] = t;
                t = A[col * C + c]; A[col * C + c] = A[piv * C + c]; A[piv * C + c] = t;
            }
        __syncthreads();
        const float d = 1.f / B[col * C + col];
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += blockDim.x) { B[col * C + c] *= d; A[col * C + c] *= d; }
        __syncthreads();
        for (int i = threadIdx.x; i < C * C; i += blockDim.x) {
            const int r = i / C, c = i % C;
            if (r != col) {
                const float f = B[r * C + col];
                if (c != col) B[i] -= f * B[col * C + c];
                A[i] -= f * A[col * C + c];
            }
        }
        __syncthreads();
        for (int r = threadIdx.x; r < C; r += blockDim.x) if (r != col) B[r * C + col] = 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) Winv[i] = A[i];
}
__global__ void invconv_weight_kernel(const float* __restrict__ wp, const float* __restrict__ wl, const float* __restrict__ wu,
                                      const float* __restrict__ ws, const float* __restrict__ ssign, int C,
                                      float* __restrict__ W, float* __restrict__ Winv) {
    invconv_weight_body(wp, wl, wu, ws, ssign, C, W, Winv);
}
struct IcwJobs { mcgen_icw_t j[MCGEN_GLOW_BATCH_MAX]; };
__global__ void invconv_weight_batch_kernel(const IcwJobs jobs) {
    const mcgen_icw_t& j = jobs.j[blockIdx.x];
    invconv_weight_body(j.w_p, j.w_l, j.w_u, j.w_s, j.s_sign, j.C, j.weight, j.weight_inv);
}

// ---- affine coupling (mcglow.py:153-175), one workgroup per sample --------------------------------------------
// forward : y[:, :C/2] = x[:, :C/2];  s = sigmoid(h[:, :C/2] + 2);  y[:, C/2:] = (x[:, C/2:] + h[:, C/2:]) * s;
//           logdet[n] (+)= sum log s
// reverse : x[:, C/2:] = y[:, C/2:] / s - h[:, C/2:]
template <typename T>
__global__ void coupling_kernel(const T* __restrict__ x, const T* __restrict__ h, T* __restrict__ y, float* logdet,
                                int HW, int C, int Cp, int reverse, int accumulate) {
    __shared__ float red[32];
    const int n = blockIdx.x, half = C / 2;
    const size_t base = (size_t)n * HW * Cp;
    float ld = 0.f;
    for (int i = threadIdx.x; i < HW * half; i += blockDim.x) {
        const int p = i / half, j = i % half;
        const size_t o = base + (size_t)p * Cp;
        const float log_s = Elem<T>::to_f(h[o + j]) + 2.f;
        const float s = 1.f / (1.f + expf(-log_s));
        const float t = Elem<T>::to_f(h[o + half + j]);
        const float xb = Elem<T>::to_f(x[o + half + j]);
        y[o + j] = x[o + j];
        y[o + half + j] = Elem<T>::from_f(reverse ? xb / s - t : (xb + t) * s);
        ld += logf(s);
    }
    for (int i = threadIdx.x; i < HW * (Cp - C); i += blockDim.x)       // keep the padded channels zero
        y[base + (size_t)(i / (Cp - C)) * Cp + C + i % (Cp - C)] = Elem<T>::from_f(0.f);
    if (logdet && !reverse) {
        ld = block_sum_g(ld, red);
        if (threadIdx.x == 0) logdet[n] = accumulate ? logdet[n] + ld : ld;
    }
}

// ---- Gaussian prior (mcglow.py:16-21,229-238,253-262) -----------------------------------------------------------
// logp[n] (+)= sum over channels [c0, c0+Cz) of z of log N(z; mean, exp(log_sd)), with (mean, log_sd) = the two
// channel halves of `prior` (2*Cz channels).  sample: z = mean + exp(log_sd) * eps.
template <typename T>
__global__ void gaussian_logp_kernel(const T* __restrict__ z, int Cpz, int c0, const T* __restrict__ prior, int Cpp,
                                     int HW, int Cz, float* logp, int accumulate) {
    __shared__ float red[32];
    const int n = blockIdx.x;
    float acc = 0.f;
    for (int i = threadIdx.x; i < HW * Cz; i += blockDim.x) {
        const int p = i / Cz, j = i % Cz;
        const float zv = Elem<T>::to_f(z[((size_t)n * HW + p) * Cpz + c0 + j]);
        const float mean = Elem<T>::to_f(prior[((size_t)n * HW + p) * Cpp + j]);
        const float lsd = Elem<T>::to_f(prior[((size_t)n * HW + p) * Cpp + Cz + j]);
        const float d = zv - mean;
        acc += -0.9189385332046727f - lsd - 0.5f * d * d * expf(-2.f * lsd);
    }
    acc = block_sum_g(acc, red);
    if (threadIdx.x == 0) logp[n] = accumulate ? logp[n] + acc : acc;
}
template <typename T>
__global__ void gaussian_sample_kernel(const T* __restrict__ eps, int Cpe, const T* __restrict__ prior, int Cpp,
                                       T* __restrict__ out, int Cpo, int c0, size_t pixels, int Cz) {
    const size_t total = pixels * Cz;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / Cz; const int j = (int)(i % Cz);
        const float mean = Elem<T>::to_f(prior[p * Cpp + j]), lsd = Elem<T>::to_f(prior[p * Cpp + Cz + j]);
        out[p * Cpo + c0 + j] = Elem<T>::from_f(mean + expf(lsd) * Elem<T>::to_f(eps[p * Cpe + j]));
    }
}
// copy channels [0, Cn) of src into channels [c0, c0+Cn) of dst (split / concat of the multi-scale architecture)
template <typename T>
__global__ void copy_channels_kernel(const T* __restrict__ src, int Cps, int s0, T* __restrict__ dst, int Cpd, int c0,
                                     size_t pixels, int Cn) {
    const size_t total = pixels * Cn;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / Cn; const int j = (int)(i % Cn);
        dst[p * Cpd + c0 + j] = src[p * Cps + s0 + j];
    }
}


// ---- backward pieces -------------------------------------------------------------------------------------------
// Affine coupling backward.  Forward: y_a = v_a, y_b = (v_b + t) * s, s = sigmoid(log_s + 2), logdet_n += sum log s.
// Given dy and g = dL/dlogdet_n (one scalar, the same for every sample):
//   dv_a = dy_a (the coupling network's input gradient is added by its own dgrad launch), dv_b = dy_b * s,
//   dt = dy_b * s,  dlog_s = dy_b * (v_b + t) * s * (1 - s) + g * (1 - s)
template <typename T>
__global__ void coupling_bwd_kernel(const T* __restrict__ v, const T* __restrict__ h, const T* __restrict__ dy,
                                    T* __restrict__ dv, T* __restrict__ dh, float g, size_t pixels, int C, int Cp) {
    const int half = C / 2;
    const size_t total = pixels * Cp;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cp);
        if (c >= C) { dv[i] = Elem<T>::from_f(0.f); dh[i] = Elem<T>::from_f(0.f); continue; }
        if (c < half) {
            const float log_s = Elem<T>::to_f(h[i]) + 2.f;
            const float s = 1.f / (1.f + expf(-log_s));
            const float t = Elem<T>::to_f(h[i + half]), vb = Elem<T>::to_f(v[i + half]), dyb = Elem<T>::to_f(dy[i + half]);
            dv[i] = dy[i];
            dh[i] = Elem<T>::from_f(dyb * (vb + t) * s * (1.f - s) + g * (1.f - s));
        } else {
            const float log_s = Elem<T>::to_f(h[i - half]) + 2.f;
            const float s = 1.f / (1.f + expf(-log_s));
            const float dyb = Elem<T>::to_f(dy[i]);
            dv[i] = Elem<T>::from_f(dyb * s);
            dh[i] = Elem<T>::from_f(dyb * s);
        }
    }
}
// Gaussian prior backward: logp_n = sum log N(z; mean, exp(lsd)); g = dL/dlogp_n.
//   dz (+)= -g * (z - mean) * exp(-2 lsd);  dmean = +g * (z - mean) * exp(-2 lsd);  dlsd = g * (-1 + (z - mean)^2 exp(-2 lsd))
template <typename T>
__global__ void gaussian_logp_bwd_kernel(const T* __restrict__ z, int Cpz, int c0, const T* __restrict__ prior, int Cpp,
                                         T* __restrict__ dz, int Cpd, int d0, T* __restrict__ dprior, float g,
                                         size_t pixels, int Cz, int accumulate_dz) {
    const size_t total = pixels * Cz;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t p = i / Cz; const int j = (int)(i % Cz);
        const float zv = Elem<T>::to_f(z[p * Cpz + c0 + j]);
        const float mean = Elem<T>::to_f(prior[p * Cpp + j]), lsd = Elem<T>::to_f(prior[p * Cpp + Cz + j]);
        const float e = expf(-2.f * lsd), d = zv - mean;
        const float gz = -g * d * e;
        T* o = dz + p * Cpd + d0 + j;
        *o = Elem<T>::from_f(accumulate_dz ? Elem<T>::to_f(*o) + gz : gz);
        dprior[p * Cpp + j] = Elem<T>::from_f(-gz);
        dprior[p * Cpp + Cz + j] = Elem<T>::from_f(g * (-1.f + d * d * e));
    }
}
// out[c] (+)= alpha * sum_p a[p, c] * b[p, c]   (two stages, fixed order)
template <typename T>
__device__ __forceinline__ void prod_colsum_stage1_body(const T* __restrict__ a, int pa, const T* __restrict__ b, int pb, size_t pixels, int C,
                                                        float* __restrict__ ws, size_t ppb) {
    __shared__ float sh[4][64];
    const size_t p0 = blockIdx.x * ppb, p1 = (p0 + ppb < pixels) ? p0 + ppb : pixels;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        float s = 0.f;
        if (c < C)
            for (size_t p = p0 + w; p < p1; p += 4) s = fmaf(Elem<T>::to_f(a[p * pa + c]), Elem<T>::to_f(b[p * pb + c]), s);
        sh[w][lane] = s;
        __syncthreads();
        if (w == 0 && c < C) ws[(size_t)blockIdx.x * C + c] = (sh[0][lane] + sh[1][lane]) + (sh[2][lane] + sh[3][lane]);
        __syncthreads();
    }
}
template <typename T>
__global__ __launch_bounds__(256)
void prod_colsum_stage1(const T* __restrict__ a, int pa, const T* __restrict__ b, int pb, size_t pixels, int C,
                        float* __restrict__ ws, size_t ppb) {
    prod_colsum_stage1_body<T>(a, pa, b, pb, pixels, C, ws, ppb);
}
static __host__ __device__ inline int pcs_blocks(long pixels) { return pixels < 1024 ? (int)((pixels + 3) / 4) : 256; }
struct PcsJobs { mcgen_pcs_t j[MCGEN_GLOW_BATCH_MAX]; };
template <typename T>
__global__ __launch_bounds__(256)
void prod_colsum_stage1_batch(const PcsJobs jobs, float* __restrict__ ws, int ws_stride) {
    const mcgen_pcs_t& j = jobs.j[blockIdx.y];
    const int blocks = pcs_blocks(j.pixels);
    if ((int)blockIdx.x >= blocks) return;
    prod_colsum_stage1_body<T>(reinterpret_cast<const T*>(j.a), j.pitch_a, reinterpret_cast<const T*>(j.b), j.pitch_b, (size_t)j.pixels, j.C,
                               ws + (size_t)blockIdx.y * ws_stride, ((size_t)j.pixels + blocks - 1) / blocks);
}
__device__ __forceinline__ void prod_colsum_stage2_body(const float* __restrict__ ws, int blocks, int C, float* out, float alpha, int accumulate) {
    const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);          // one wave per channel
    if (c >= C) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int b = lane; b < blocks; b += 64) s += ws[(size_t)b * C + c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
    if (lane == 0) { const float v = alpha * s; out[c] = accumulate ? out[c] + v : v; }
}
__global__ void prod_colsum_stage2(const float* __restrict__ ws, int blocks, int C, float* out, float alpha, int accumulate) {
    prod_colsum_stage2_body(ws, blocks, C, out, alpha, accumulate);
}
__global__ void prod_colsum_stage2_batch(const PcsJobs jobs, const float* __restrict__ ws, int ws_stride) {
    const mcgen_pcs_t& j = jobs.j[blockIdx.y];
    prod_colsum_stage2_body(ws + (size_t)blockIdx.y * ws_stride, pcs_blocks(j.pixels), j.C, j.out, j.alpha, j.accumulate);
}
// ActNorm parameter gradients from the (sum dx, sum dx * (x + loc)) partials of a dgrad epilogue, where
// dx is the gradient w.r.t. the ActNorm INPUT (dx = scale * du):
//   dloc = sum dx,   dscale = (1/scale) * sum dx * (x + loc) + ld_coef / scale      (ld_coef = g * N * H * W or 0)
__device__ __forceinline__ void actnorm_bwd_body(const float* __restrict__ part, int tiles, int pitch, int C, const float* __restrict__ scale,
                                                 float ld_coef, int input_side, float* dloc, float* dscale, int accumulate) {
    __shared__ double sh[16][2][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    double s1 = 0.0, s2 = 0.0;
    if (c < C)
        for (int t = w; t < tiles; t += 16) { s1 += (double)part[((size_t)t * 2) * pitch + c]; s2 += (double)part[((size_t)t * 2 + 1) * pitch + c]; }
    sh[w][0][lane] = s1; sh[w][1][lane] = s2;
    __syncthreads();
    if (w != 0 || c >= C) return;
    s1 = 0.0; s2 = 0.0;
    for (int i = 0; i < 16; ++i) { s1 += sh[i][0][lane]; s2 += sh[i][1][lane]; }
    const float sc = scale[c];
    // input_side = 1: partials were taken on dx = scale * du;  0: on du itself
    const float gl = input_side ? (float)s1 : (float)s1 * sc;
    const float gs = (input_side ? (float)s2 / sc : (float)s2) + ld_coef / sc;
    dloc[c] = accumulate ? dloc[c] + gl : gl;
    dscale[c] = accumulate ? dscale[c] + gs : gs;
}
__global__ __launch_bounds__(1024)
void actnorm_bwd_kernel(const float* __restrict__ part, int tiles, int pitch, int C, const float* __restrict__ scale,
                        float ld_coef, int input_side, float* dloc, float* dscale, int accumulate) {
    actnorm_bwd_body(part, tiles, pitch, C, scale, ld_coef, input_side, dloc, dscale, accumulate);
}
struct AnBwdJobs { mcgen_an_bwd_t j[MCGEN_GLOW_BATCH_MAX]; };
__global__ __launch_bounds__(1024)
void actnorm_bwd_batch_kernel(const AnBwdJobs jobs) {
    const mcgen_an_bwd_t& j = jobs.j[blockIdx.y];
    if ((int)blockIdx.x * 64 >= j.C) return;
    actnorm_bwd_body(j.partials, j.tiles, j.pitch, j.C, j.scale, j.ld_coef, j.input_side, j.dloc, j.dscale, j.accumulate);
}
// InvConv2dLU parameter gradients from dW (gradient w.r.t. the C x C weight):  W = P L U,
//   dL = P^T dW U^T (strictly lower part -> dw_l),  dU = (P L)^T dW (strictly upper -> dw_u,
//   diagonal * sign * exp(w_s) + ld_coef -> dw_s)
__device__ void invconv_bwd_body(const float* __restrict__ wp, const float* __restrict__ wl, const float* __restrict__ wu,
                                 const float* __restrict__ ws, const float* __restrict__ ssign, const float* __restrict__ dW,
                                 int C, int ldw, float ld_coef, float* dwl, float* dwu, float* dws, int accumulate) {
    extern __shared__ float sh[];
    float* Lm = sh; float* Um = sh + C * C; float* A = sh + 2 * C * C; float* B = sh + 3 * C * C;
    float* Pm = sh + 4 * C * C; float* G = sh + 5 * C * C;
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) {
        const int r = i / C, c = i % C;
        Lm[i] = (r > c ? wl[i] : 0.f) + (r == c ? 1.f : 0.f);
        Um[i] = (c > r ? wu[i] : 0.f) + (r == c ? ssign[r] * expf(ws[r]) : 0.f);
        Pm[i] = wp[i]; G[i] = dW[r * ldw + c];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) {          // A = P^T dW
        const int r = i / C, c = i % C;
        float s = 0.f;
        for (int k = 0; k < C; ++k) s = fmaf(Pm[k * C + r], G[k * C + c], s);
        A[i] = s;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < C * C; i += blockDim.x) {          // dL = A U^T ; B = L^T A = dU
        const int r = i / C, c = i % C;
        float s = 0.f, u = 0.f;
        for (int k = 0; k < C; ++k) { s = fmaf(A[r * C + k], Um[c * C + k], s); u = fmaf(Lm[k * C + r], A[k * C + c], u); }
        if (r > c) dwl[i] = accumulate ? dwl[i] + s : s; else if (!accumulate) dwl[i] = 0.f;
        if (c > r) dwu[i] = accumulate ? dwu[i] + u : u; else if (!accumulate) dwu[i] = 0.f;
        B[i] = u;
    }
    __syncthreads();
    for (int r = threadIdx.x; r < C; r += blockDim.x) {
        const float v = B[r * C + r] * ssign[r] * expf(ws[r]) + ld_coef;
        dws[r] = accumulate ? dws[r] + v : v;
    }
}
__global__ void invconv_bwd_kernel(const float* __restrict__ wp, const float* __restrict__ wl, const float* __restrict__ wu,
                                   const float* __restrict__ ws, const float* __restrict__ ssign, const float* __restrict__ dW,
                                   int C, int ldw, float ld_coef, float* dwl, float* dwu, float* dws, int accumulate) {
    invconv_bwd_body(wp, wl, wu, ws, ssign, dW, C, ldw, ld_coef, dwl, dwu, dws, accumulate);
}
struct IcbJobs { mcgen_icb_t j[MCGEN_GLOW_BATCH_MAX]; };
__global__ void invconv_bwd_batch_kernel(const IcbJobs jobs) {
    const mcgen_icb_t& j = jobs.j[blockIdx.x];
    invconv_bwd_body(j.w_p, j.w_l, j.w_u, j.w_s, j.s_sign, j.dW, j.C, j.ldw, j.ld_coef, j.dw_l, j.dw_u, j.dw_s, j.accumulate);
}
// global L2 norm of a flat gradient buffer (clip_grad_norm_, train_vae.py:110) and the scaled copy
__global__ void sqsum_kernel(const float* __restrict__ g, size_t n, float* __restrict__ part) {
    __shared__ float red[32];
    float s = 0.f;
    // (eight loads in flight per thread, the same serial order of additions: one element per trip was a dependent round trip each)
    const size_t st = (size_t)gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + 7 * st < n; i += 8 * st) {
        float q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e] = g[i + e * st];
#pragma unroll
        for (int e = 0; e < 8; ++e) s = fmaf(q[e], q[e], s);
    }
    for (; i < n; i += st) s = fmaf(g[i], g[i], s);
    s = block_sum_g(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}
__global__ void clip_scale_kernel(float* __restrict__ g, size_t n, const float* __restrict__ part, int blocks, float max_norm,
                                  float* norm_out) {
    __shared__ float coef;
    if (threadIdx.x == 0) {
        double s = 0.0;
        int b = 0;
        for (; b + 16 <= blocks; b += 16) {                     // (sixteen loads in flight, the same order of additions)
            float q[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) q[e] = part[b + e];
#pragma unroll
            for (int e = 0; e < 16; ++e) s += (double)q[e];
        }
        for (; b < blocks; ++b) s += (double)part[b];
        const float nrm = (float)sqrt(s);
        if (norm_out && blockIdx.x == 0) norm_out[0] = nrm;
        const float c = max_norm / (nrm + 1e-6f);
        coef = c < 1.f ? c : 1.f;
    }
    __syncthreads();
    const float c = coef;
    if (c >= 1.f) return;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) g[i] *= c;
}

}  // namespace

#define DISPATCH_T(dtype, CALL_F32, CALL_BF16) \
    if (dtype == MCGEN_F32) { CALL_F32; } else if (dtype == MCGEN_BF16) { CALL_BF16; } \
    else return mcgen_fail("unknown dtype %d", dtype)

extern "C" int mcgen_glow_squeeze(const void* x, void* y, int dtype, int N, int H, int W, int C, int Cp_big, int Cp_small,
                                  int inverse, void* stream) {
    MCGEN_CHECK(x && y && H % 2 == 0 && W % 2 == 0 && Cp_big >= C && Cp_small >= 4 * C, "glow_squeeze: bad arguments");
    const size_t total = (size_t)N * (H / 2) * (W / 2) * Cp_small;
    if (inverse && Cp_big > C) {
        const size_t px = (size_t)N * H * W;
        DISPATCH_T(dtype,
            hipLaunchKernelGGL(zero_pad_channels_kernel<float>, dim3(grid_for(px * (Cp_big - C))), dim3(256), 0, STREAM(stream), (float*)y, px, C, Cp_big),
            hipLaunchKernelGGL(zero_pad_channels_kernel<bf16_t>, dim3(grid_for(px * (Cp_big - C))), dim3(256), 0, STREAM(stream), (bf16_t*)y, px, C, Cp_big));
    }
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(squeeze_kernel<float>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const float*)x, (float*)y, N, H, W, C, Cp_big, Cp_small, inverse),
        hipLaunchKernelGGL(squeeze_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const bf16_t*)x, (bf16_t*)y, N, H, W, C, Cp_big, Cp_small, inverse));
    MCGEN_LAUNCH_CHECK("glow_squeeze"); return 0;
}

extern "C" int mcgen_channel_stats(const void* x, int dtype, int64_t pixels, int Cp, float* partials, int blocks, void* stream) {
    MCGEN_CHECK(x && partials && pixels > 0 && Cp > 0 && blocks > 0, "channel_stats: bad arguments");
    const size_t ppb = ((size_t)pixels + blocks - 1) / blocks;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(channel_stats_kernel<float>, dim3(blocks), dim3(64), 0, STREAM(stream), (const float*)x, (size_t)pixels, Cp, partials, ppb),
        hipLaunchKernelGGL(channel_stats_kernel<bf16_t>, dim3(blocks), dim3(64), 0, STREAM(stream), (const bf16_t*)x, (size_t)pixels, Cp, partials, ppb));
    MCGEN_LAUNCH_CHECK("channel_stats"); return 0;
}

extern "C" int mcgen_actnorm_init(const float* partials, int tiles, int pitch, int C, double count, float* loc, float* scale, void* stream) {
    MCGEN_CHECK(partials && loc && scale && tiles > 0 && pitch >= C && count > 0, "actnorm_init: bad arguments");
    hipLaunchKernelGGL(actnorm_init_kernel, dim3((C + 63) / 64), dim3(64), 0, STREAM(stream), partials, tiles, pitch, C, count, loc, scale);
    MCGEN_LAUNCH_CHECK("actnorm_init"); return 0;
}
extern "C" int mcgen_actnorm_affine(const float* loc, const float* scale, int C, int Cp, float* a, float* b, float* negloc, void* stream) {
    MCGEN_CHECK(loc && scale && a && b && Cp >= C, "actnorm_affine: bad arguments");
    hipLaunchKernelGGL(actnorm_affine_kernel, dim3((Cp + 63) / 64), dim3(64), 0, STREAM(stream), loc, scale, C, Cp, a, b, negloc);
    MCGEN_LAUNCH_CHECK("actnorm_affine"); return 0;
}
extern "C" int mcgen_glow_param_logdet(const float* scale, int C, const float* w_s, int Cw, float hw, float* logdet, int N, void* stream) {
    MCGEN_CHECK(scale && w_s && logdet && C > 0 && Cw > 0 && N > 0, "glow_param_logdet: bad arguments");
    hipLaunchKernelGGL(glow_param_logdet_kernel, dim3(1), dim3(256), 0, STREAM(stream), scale, C, w_s, Cw, hw, logdet, N);
    MCGEN_LAUNCH_CHECK("glow_param_logdet"); return 0;
}

extern "C" int mcgen_invconv_weight(const float* w_p, const float* w_l, const float* w_u, const float* w_s, const float* s_sign,
                                    int C, float* weight, float* weight_inv, void* stream) {
    MCGEN_CHECK(w_p && w_l && w_u && w_s && s_sign && weight && C > 0 && C <= 64, "invconv_weight: C must be in 1..64");
    hipLaunchKernelGGL(invconv_weight_kernel, dim3(1), dim3(256), 4 * C * C * sizeof(float), STREAM(stream),
                       w_p, w_l, w_u, w_s, s_sign, C, weight, weight_inv);
    MCGEN_LAUNCH_CHECK("invconv_weight"); return 0;
}

extern "C" int mcgen_glow_coupling(const void* x, const void* h, void* y, int dtype, float* logdet, int N, int HW, int C, int Cp,
                                   int reverse, int accumulate, void* stream) {
    MCGEN_CHECK(x && h && y && C % 2 == 0 && Cp >= C, "glow_coupling: bad arguments");
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(coupling_kernel<float>, dim3(N), dim3(256), 0, STREAM(stream), (const float*)x, (const float*)h, (float*)y, logdet, HW, C, Cp, reverse, accumulate),
        hipLaunchKernelGGL(coupling_kernel<bf16_t>, dim3(N), dim3(256), 0, STREAM(stream), (const bf16_t*)x, (const bf16_t*)h, (bf16_t*)y, logdet, HW, C, Cp, reverse, accumulate));
    MCGEN_LAUNCH_CHECK("glow_coupling"); return 0;
}

extern "C" int mcgen_gaussian_logp(const void* z, int Cpz, int c0, const void* prior, int Cpp, int dtype, int N, int HW, int Cz,
                                   float* logp, int accumulate, void* stream) {
    MCGEN_CHECK(z && prior && logp && Cpp >= 2 * Cz && Cpz >= c0 + Cz, "gaussian_logp: bad arguments");
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(gaussian_logp_kernel<float>, dim3(N), dim3(256), 0, STREAM(stream), (const float*)z, Cpz, c0, (const float*)prior, Cpp, HW, Cz, logp, accumulate),
        hipLaunchKernelGGL(gaussian_logp_kernel<bf16_t>, dim3(N), dim3(256), 0, STREAM(stream), (const bf16_t*)z, Cpz, c0, (const bf16_t*)prior, Cpp, HW, Cz, logp, accumulate));
    MCGEN_LAUNCH_CHECK("gaussian_logp"); return 0;
}
extern "C" int mcgen_gaussian_sample(const void* eps, int Cpe, const void* prior, int Cpp, void* out, int Cpo, int c0, int dtype,
                                     int64_t pixels, int Cz, void* stream) {
    MCGEN_CHECK(eps && prior && out && Cpp >= 2 * Cz && Cpo >= c0 + Cz && Cpe >= Cz, "gaussian_sample: bad arguments");
    const size_t total = (size_t)pixels * Cz;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(gaussian_sample_kernel<float>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const float*)eps, Cpe, (const float*)prior, Cpp, (float*)out, Cpo, c0, (size_t)pixels, Cz),
        hipLaunchKernelGGL(gaussian_sample_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const bf16_t*)eps, Cpe, (const bf16_t*)prior, Cpp, (bf16_t*)out, Cpo, c0, (size_t)pixels, Cz));
    MCGEN_LAUNCH_CHECK("gaussian_sample"); return 0;
}
extern "C" int mcgen_copy_channels(const void* src, int Cps, int s0, void* dst, int Cpd, int c0, int dtype, int64_t pixels, int Cn, void* stream) {
    MCGEN_CHECK(src && dst && Cps >= s0 + Cn && Cpd >= c0 + Cn, "copy_channels: bad arguments");
    const size_t total = (size_t)pixels * Cn;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(copy_channels_kernel<float>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const float*)src, Cps, s0, (float*)dst, Cpd, c0, (size_t)pixels, Cn),
        hipLaunchKernelGGL(copy_channels_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const bf16_t*)src, Cps, s0, (bf16_t*)dst, Cpd, c0, (size_t)pixels, Cn));
    MCGEN_LAUNCH_CHECK("copy_channels"); return 0;
}

extern "C" int mcgen_glow_coupling_bwd(const void* v, const void* h, const void* dy, void* dv, void* dh, int dtype, float g,
                                       int64_t pixels, int C, int Cp, void* stream) {
    MCGEN_CHECK(v && h && dy && dv && dh && C % 2 == 0 && Cp >= C, "glow_coupling_bwd: bad arguments");
    const size_t total = (size_t)pixels * Cp;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(coupling_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const float*)v, (const float*)h, (const float*)dy, (float*)dv, (float*)dh, g, (size_t)pixels, C, Cp),
        hipLaunchKernelGGL(coupling_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const bf16_t*)v, (const bf16_t*)h, (const bf16_t*)dy, (bf16_t*)dv, (bf16_t*)dh, g, (size_t)pixels, C, Cp));
    MCGEN_LAUNCH_CHECK("glow_coupling_bwd"); return 0;
}
extern "C" int mcgen_gaussian_logp_bwd(const void* z, int Cpz, int c0, const void* prior, int Cpp, void* dz, int Cpd, int d0,
                                       void* dprior, int dtype, float g, int64_t pixels, int Cz, int accumulate_dz, void* stream) {
    MCGEN_CHECK(z && prior && dz && dprior && Cpp >= 2 * Cz && Cpz >= c0 + Cz && Cpd >= d0 + Cz, "gaussian_logp_bwd: bad arguments");
    const size_t total = (size_t)pixels * Cz;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(gaussian_logp_bwd_kernel<float>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const float*)z, Cpz, c0, (const float*)prior, Cpp, (float*)dz, Cpd, d0, (float*)dprior, g, (size_t)pixels, Cz, accumulate_dz),
        hipLaunchKernelGGL(gaussian_logp_bwd_kernel<bf16_t>, dim3(grid_for(total)), dim3(256), 0, STREAM(stream), (const bf16_t*)z, Cpz, c0, (const bf16_t*)prior, Cpp, (bf16_t*)dz, Cpd, d0, (bf16_t*)dprior, g, (size_t)pixels, Cz, accumulate_dz));
    MCGEN_LAUNCH_CHECK("gaussian_logp_bwd"); return 0;
}
extern "C" int mcgen_prod_colsum(const void* a, int pitch_a, const void* b, int pitch_b, int dtype, int64_t pixels, int C,
                                 float* out, float alpha, int accumulate, float* workspace, void* stream) {
    MCGEN_CHECK(a && b && out && workspace && pixels > 0 && C > 0, "prod_colsum: bad arguments (workspace: 256*C floats)");
    const int blocks = pixels < 1024 ? (int)((pixels + 3) / 4) : 256;
    const size_t ppb = ((size_t)pixels + blocks - 1) / blocks;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(prod_colsum_stage1<float>, dim3(blocks), dim3(256), 0, STREAM(stream), (const float*)a, pitch_a, (const float*)b, pitch_b, (size_t)pixels, C, workspace, ppb),
        hipLaunchKernelGGL(prod_colsum_stage1<bf16_t>, dim3(blocks), dim3(256), 0, STREAM(stream), (const bf16_t*)a, pitch_a, (const bf16_t*)b, pitch_b, (size_t)pixels, C, workspace, ppb));
    hipLaunchKernelGGL(prod_colsum_stage2, dim3((C + 3) / 4), dim3(256), 0, STREAM(stream), workspace, blocks, C, out, alpha, accumulate);
    MCGEN_LAUNCH_CHECK("prod_colsum"); return 0;
}
extern "C" int mcgen_actnorm_bwd(const float* partials, int tiles, int pitch, int C, const float* scale, float ld_coef,
                                 int input_side, float* dloc, float* dscale, int accumulate, void* stream) {
    MCGEN_CHECK(partials && scale && dloc && dscale && tiles > 0 && pitch >= C, "actnorm_bwd: bad arguments");
    hipLaunchKernelGGL(actnorm_bwd_kernel, dim3((C + 63) / 64), dim3(1024), 0, STREAM(stream), partials, tiles, pitch, C, scale, ld_coef, input_side, dloc, dscale, accumulate);
    MCGEN_LAUNCH_CHECK("actnorm_bwd"); return 0;
}
extern "C" int mcgen_invconv_bwd(const float* w_p, const float* w_l, const float* w_u, const float* w_s, const float* s_sign,
                                 const float* dW, int C, int ldw, float ld_coef, float* dw_l, float* dw_u, float* dw_s,
                                 int accumulate, void* stream) {
    MCGEN_CHECK(w_p && w_l && w_u && w_s && s_sign && dW && dw_l && dw_u && dw_s && C > 0 && C <= 64 && ldw >= C, "invconv_bwd: bad arguments");
    if (6 * C * C * sizeof(float) > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(invconv_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 64 * 64 * sizeof(float));
        if (e != hipSuccess) return mcgen_fail("invconv_bwd: cannot raise LDS limit: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(invconv_bwd_kernel, dim3(1), dim3(1024), 6 * C * C * sizeof(float), STREAM(stream),
                       w_p, w_l, w_u, w_s, s_sign, dW, C, ldw, ld_coef, dw_l, dw_u, dw_s, accumulate);
    MCGEN_LAUNCH_CHECK("invconv_bwd"); return 0;
}
extern "C" int mcgen_clip_grad_norm(float* g, int64_t n, float max_norm, float* norm_out, float* workspace, void* stream) {
    MCGEN_CHECK(g && workspace && n > 0 && max_norm > 0, "clip_grad_norm: bad arguments (workspace: 256 floats)");
    const int blocks = 256;
    hipLaunchKernelGGL(sqsum_kernel, dim3(blocks), dim3(256), 0, STREAM(stream), g, (size_t)n, workspace);
    hipLaunchKernelGGL(clip_scale_kernel, dim3(grid_for((size_t)n, 256, 1024)), dim3(256), 0, STREAM(stream), g, (size_t)n, workspace, blocks, max_norm, norm_out);
    MCGEN_LAUNCH_CHECK("clip_grad_norm"); return 0;
}

// ---- batched forms: one launch per kind for all modules of a pass (job tables by value, MCGEN_GLOW_BATCH_MAX per launch) ----
template <typename J, typename F>
static int glow_batches(const J* jobs, int n, int cap, F launch) {
    for (int base = 0; base < n; base += cap) {
        const int m = n - base < cap ? n - base : cap;
        if (int rc = launch(jobs + base, m)) return rc;
    }
    return 0;
}
extern "C" int mcgen_actnorm_affine_batch(const mcgen_an_affine_t* jobs, int n, void* stream) {
    MCGEN_CHECK(jobs && n > 0, "actnorm_affine_batch: bad arguments");
    return glow_batches(jobs, n, MCGEN_GLOW_BATCH_MAX, [&](const mcgen_an_affine_t* j, int m) {
        AnAffineJobs t; int cpmax = 1;
        for (int i = 0; i < m; ++i) {
            MCGEN_CHECK(j[i].loc && j[i].scale && j[i].a && j[i].b && j[i].C > 0 && j[i].Cp >= j[i].C, "actnorm_affine_batch: bad job %d", i);
            t.j[i] = j[i]; if (j[i].Cp > cpmax) cpmax = j[i].Cp;
        }
        for (int i = m; i < MCGEN_GLOW_BATCH_MAX; ++i) t.j[i] = t.j[0];
        hipLaunchKernelGGL(actnorm_affine_batch_kernel, dim3((cpmax + 63) / 64, m), dim3(64), 0, STREAM(stream), t);
        MCGEN_LAUNCH_CHECK("actnorm_affine_batch"); return 0;
    });
}
extern "C" int mcgen_glow_param_logdet_batch(const mcgen_pld_t* jobs, int n, float* logdet, int N, void* stream) {
    MCGEN_CHECK(jobs && n > 0 && logdet && N > 0, "glow_param_logdet_batch: bad arguments");
    return glow_batches(jobs, n, MCGEN_GLOW_PLD_MAX, [&](const mcgen_pld_t* j, int m) {
        PldJobs t;
        for (int i = 0; i < m; ++i) { MCGEN_CHECK(j[i].scale && j[i].w_s && j[i].C > 0 && j[i].Cw > 0, "glow_param_logdet_batch: bad job %d", i); t.j[i] = j[i]; }
        for (int i = m; i < MCGEN_GLOW_PLD_MAX; ++i) t.j[i] = t.j[0];
        hipLaunchKernelGGL(glow_param_logdet_batch_kernel, dim3(1), dim3(256), 0, STREAM(stream), t, m, logdet, N);
        MCGEN_LAUNCH_CHECK("glow_param_logdet_batch"); return 0;
    });
}
extern "C" int mcgen_invconv_weight_batch(const mcgen_icw_t* jobs, int n, void* stream) {
    MCGEN_CHECK(jobs && n > 0, "invconv_weight_batch: bad arguments");
    return glow_batches(jobs, n, MCGEN_GLOW_BATCH_MAX, [&](const mcgen_icw_t* j, int m) {
        IcwJobs t; int cmax = 1;
        for (int i = 0; i < m; ++i) {
            MCGEN_CHECK(j[i].w_p && j[i].w_l && j[i].w_u && j[i].w_s && j[i].s_sign && j[i].weight && j[i].C > 0 && j[i].C <= 64, "invconv_weight_batch: bad job %d (C in 1..64)", i);
            t.j[i] = j[i]; if (j[i].C > cmax) cmax = j[i].C;
        }
        for (int i = m; i < MCGEN_GLOW_BATCH_MAX; ++i) t.j[i] = t.j[0];
        hipLaunchKernelGGL(invconv_weight_batch_kernel, dim3(m), dim3(256), 4 * cmax * cmax * sizeof(float), STREAM(stream), t);
        MCGEN_LAUNCH_CHECK("invconv_weight_batch"); return 0;
    });
}
extern "C" int mcgen_invconv_bwd_batch(const mcgen_icb_t* jobs, int n, void* stream) {
    MCGEN_CHECK(jobs && n > 0, "invconv_bwd_batch: bad arguments");
    return glow_batches(jobs, n, MCGEN_GLOW_BATCH_MAX, [&](const mcgen_icb_t* j, int m) {
        IcbJobs t; int cmax = 1;
        for (int i = 0; i < m; ++i) {
            MCGEN_CHECK(j[i].w_p && j[i].w_l && j[i].w_u && j[i].w_s && j[i].s_sign && j[i].dW && j[i].dw_l && j[i].dw_u && j[i].dw_s &&
                        j[i].C > 0 && j[i].C <= 64 && j[i].ldw >= j[i].C, "invconv_bwd_batch: bad job %d", i);
            t.j[i] = j[i]; if (j[i].C > cmax) cmax = j[i].C;
        }
        for (int i = m; i < MCGEN_GLOW_BATCH_MAX; ++i) t.j[i] = t.j[0];
        const size_t lds = 6 * (size_t)cmax * cmax * sizeof(float);
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(invconv_bwd_batch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 6 * 64 * 64 * sizeof(float));
            if (e != hipSuccess) return mcgen_fail("invconv_bwd_batch: cannot raise LDS limit: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(invconv_bwd_batch_kernel, dim3(m), dim3(1024), lds, STREAM(stream), t);
        MCGEN_LAUNCH_CHECK("invconv_bwd_batch"); return 0;
    });
}
extern "C" int mcgen_actnorm_bwd_batch(const mcgen_an_bwd_t* jobs, int n, void* stream) {
    MCGEN_CHECK(jobs && n > 0, "actnorm_bwd_batch: bad arguments");
    return glow_batches(jobs, n, MCGEN_GLOW_BATCH_MAX, [&](const mcgen_an_bwd_t* j, int m) {
        AnBwdJobs t; int cmax = 1;
        for (int i = 0; i < m; ++i) {
            MCGEN_CHECK(j[i].partials && j[i].scale && j[i].dloc && j[i].dscale && j[i].tiles > 0 && j[i].pitch >= j[i].C && j[i].C > 0, "actnorm_bwd_batch: bad job %d", i);
            t.j[i] = j[i]; if (j[i].C > cmax) cmax = j[i].C;
        }
        for (int i = m; i < MCGEN_GLOW_BATCH_MAX; ++i) t.j[i] = t.j[0];
        hipLaunchKernelGGL(actnorm_bwd_batch_kernel, dim3((cmax + 63) / 64, m), dim3(1024), 0, STREAM(stream), t);
        MCGEN_LAUNCH_CHECK("actnorm_bwd_batch"); return 0;
    });
}
extern "C" int mcgen_prod_colsum_batch(const mcgen_pcs_t* jobs, int n, int dtype, float* workspace, void* stream) {
    MCGEN_CHECK(jobs && n > 0 && workspace, "prod_colsum_batch: bad arguments (workspace: n * 256 * max C floats)");
    int cmax = 1;
    for (int i = 0; i < n; ++i) {
        MCGEN_CHECK(jobs[i].a && jobs[i].b && jobs[i].out && jobs[i].pixels > 0 && jobs[i].C > 0, "prod_colsum_batch: bad job %d", i);
        if (jobs[i].C > cmax) cmax = jobs[i].C;
    }
    const int ws_stride = 256 * cmax;
    int base_done = 0;
    return glow_batches(jobs, n, MCGEN_GLOW_BATCH_MAX, [&](const mcgen_pcs_t* j, int m) {
        PcsJobs t;
        for (int i = 0; i < m; ++i) t.j[i] = j[i];
        for (int i = m; i < MCGEN_GLOW_BATCH_MAX; ++i) t.j[i] = t.j[0];
        float* ws = workspace + (size_t)base_done * ws_stride;
        base_done += m;
        DISPATCH_T(dtype,
            hipLaunchKernelGGL(prod_colsum_stage1_batch<float>, dim3(256, m), dim3(256), 0, STREAM(stream), t, ws, ws_stride),
            hipLaunchKernelGGL(prod_colsum_stage1_batch<bf16_t>, dim3(256, m), dim3(256), 0, STREAM(stream), t, ws, ws_stride));
        hipLaunchKernelGGL(prod_colsum_stage2_batch, dim3((cmax + 3) / 4, m), dim3(256), 0, STREAM(stream), t, ws, ws_stride);
        MCGEN_LAUNCH_CHECK("prod_colsum_batch"); return 0;
    });
}
