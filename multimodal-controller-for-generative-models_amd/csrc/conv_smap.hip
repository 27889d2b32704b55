// Convolutions on 8x8 maps, one or two whole images per workgroup (gfx950, bf16).
//
//   y[n, h, w, co] = epilogue( alpha * sum_seg sum_{tap, ci} prologue_seg(x_seg)[n, h + dh, w + dw, ci] * W_seg[co, ci, tap] )
//
// The discriminator's 8x8 residual blocks (DisResBlock, mcgan.py:95-138: 3x3, 128 -> 128, forward and input gradient: 48
// of the ~350 launches of a training iteration) and every layer of MCGatedPixelCNN (mcpixelcnn.py:16-61 on 8x8 code
// maps: 3x3 / 1x1, 128 or 256 channels, optionally a K-concatenated second segment, BatchNorm statistics in the epilogue)
// and the generator's first block (GenResBlock at 8x8, mcgan.py:9-44: 256 channels, inputs through the nearest x2 upsample,
// BatchNorm groups of the grouped 5 N pass).
// On the general 64x64 tile such a launch is 256-1024 workgroups that each run their K chunks behind barriers with the
// weights re-staged through LDS per workgroup: 13-22 us for 2-10 GFLOP (rocprofv3, profiles/r03_*_kernel_stats.csv).
//
// Here a workgroup (512 threads) owns IMGS whole images and 64 output channels:
//   * every segment's window of every image (10 x 10 pixels for 3x3, 8 x 8 for 1x1, all channels, prologue applied:
//     affine, ReLU, MultimodalController code -- modules.py:71-76) is staged into LDS ONCE, XOR-swizzled so that the 16
//     pixels of an MFMA fragment hit 16 different 16-byte bank groups under ds_read_b128's lane grouping
//     (MI355X_MICROARCH.md, LDS: lane l15 of fragment f is pixel (2 f + (l15 & 1), l15 >> 1), swizzle (8 r + 2 c) & 15);
//   * K (all segments, chunks, taps: KT steps of 32 channels) is split into 4 parts; wave = (32 of the 64 channels) x
//     (K part), for every image of the workgroup.  Its weight fragments (16 co x 32 ci, 1 KB contiguous in the
//     [chunk][tap][co][32] image) come straight from L2 into registers, up to nine K steps ahead -- no wave loads a
//     fragment another wave loads, nothing is re-staged; per K step and image 4 window fragments from LDS, 8 MFMAs;
//   * the K parts meet in LDS; the epilogue (alpha, bias, bias2, output code, ReLU gate, residual, BatchNorm partial sums)
//     runs on 16-byte units.
// What bounds it: an L2-resident operand streams into a CU's registers at ~70 GB/s (MI355X_MICROARCH.md, "Indexed rows";
// measured here: the form whose two image halves each loaded the 64-channel slab took 4.2 us longer than the form that
// loads it once -- 147 KB at 70 GB/s = 2.1 us per copy), so the slab (K x 128 B) is loaded ONCE per workgroup and two
// images share it when the launch still fills the chip.
#include "conv_tile.h"

namespace {

constexpr int IM_NT = 512, IM_COT = 64;
constexpr int IM_EP = IM_COT + 4;                      // floats per pixel row of a K-part exchange buffer
constexpr int IM_UP1 = 5;                              // segment kind: 1x1 over the 4x4 source of a x2 upsample

template <int C0, int KS0, int C1, int KS1, int IMGS>
struct ImCfg {
    static constexpr int KH = 4;                        // K parts: wave = (32 of the 64 channels) x (K part), every image
    // (KS1 == IM_UP1: a 1x1 segment whose input arrives through the nearest x2 upsample -- its window holds the 4 x 4 SOURCE
    // pixels, 8 KB instead of 32: two images' windows of the generator's conv_b ++ shortcut launch fit the LDS)
    static constexpr int T0 = KS0 * KS0, T1 = KS1 == IM_UP1 ? 1 : KS1 * KS1;
    static constexpr int KT0 = (C0 / 32) * T0, KT1 = (C1 / 32) * T1, KT = KT0 + KT1;
    static_assert(KT % KH == 0, "K steps split evenly over the K parts");
    static constexpr int KPW = KT / KH;                 // K steps per wave
    static constexpr int PF = KPW < 9 ? KPW : 9;        // weight fragments in flight (K steps)
    static constexpr int PP0 = KS0 == 3 ? 100 : 64, PP1 = KS1 == 3 ? 100 : (KS1 == IM_UP1 ? 16 : 64);
    static constexpr int WIN0 = PP0 * C0 * 2, WIN1 = PP1 * C1 * 2;     // bytes per image
    static constexpr int WIN = IMGS * (WIN0 + WIN1);
    static constexpr int EBUF = IMGS * 64 * IM_EP * 4;  // one K part's accumulators
    static constexpr int LDS = WIN > KH * EBUF ? WIN : KH * EBUF;
    static_assert(LDS <= 160 * 1024, "LDS");
};

// byte offset of 16-byte unit `unit` of window pixel (wr, wc); PC = pixels per window row (10 with the halo, 8 without)
template <int PC, int ROWB>
static __device__ __forceinline__ int im_off(int wr, int wc, int unit) {
    return (wr * PC + wc) * ROWB + ((unit ^ ((8 * wr + 2 * wc) & 15)) << 4);
}

// one segment of one image: global -> prologue -> swizzled window
template <int C, int KS>
struct ImSeg {
    static constexpr int NPX = KS == IM_UP1 ? 16 : 64;   // pixels staged per image
    static constexpr int UPP = C / 8, NI = (NPX * UPP + IM_NT - 1) / IM_NT, PSTEP = IM_NT / UPP;
    static constexpr int PC = KS == 3 ? 10 : (KS == IM_UP1 ? 4 : 8), HALO = KS == 3 ? 1 : 0, ROWB = C * 2;
    static constexpr int LGP = KS == IM_UP1 ? 2 : 3;     // log2 of the staged map's side
    static __device__ __forceinline__ void load(const mcgen_seg_t& sg, int n, int tid, u32x4 (&raw)[NI]) {
        const int u = tid & (UPP - 1), px0 = tid / UPP;
        // ups: x is the 4x4 map under a nearest x2 upsample (mcgan.py:17,27) -- pixel (r, c) reads (r >> 1, c >> 1)
        const bf16_t* xs = reinterpret_cast<const bf16_t*>(sg.x) + ((size_t)n * ((sg.ups || KS == IM_UP1) ? 16 : 64)) * C + u * 8;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int px = px0 + PSTEP * k;
            const int src = (sg.ups && KS != IM_UP1) ? ((px >> 4) << 2) + ((px & 7) >> 1) : px;
            raw[k] = *reinterpret_cast<const u32x4*>(xs + (size_t)(src < NPX || KS != IM_UP1 ? src : 0) * C);
        }
    }
    static __device__ __forceinline__ void write(const mcgen_seg_t& sg, int tid, const u32x4 (&raw)[NI], const float (&sc)[8],
                                                 const float (&sh)[8], const float (&cd)[8], char* win) {
        const int u = tid & (UPP - 1), px0 = tid / UPP;
        const float relu_lo = sg.relu ? 0.f : -__builtin_inff();
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int px = px0 + PSTEP * k;
            union { bf16x8 h; u32x4 w; } o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v0 = fmaxf(fmaf(__uint_as_float(raw[k][e] << 16), sc[2 * e], sh[2 * e]), relu_lo) * cd[2 * e];
                const float v1 = fmaxf(fmaf(__uint_as_float(raw[k][e] & 0xffff0000u), sc[2 * e + 1], sh[2 * e + 1]), relu_lo) * cd[2 * e + 1];
                o.h[2 * e] = (bf16_t)v0; o.h[2 * e + 1] = (bf16_t)v1;
            }
            if (KS != IM_UP1 || px < NPX)
                *reinterpret_cast<u32x4*>(win + im_off<PC, ROWB>((px >> LGP) + HALO, (px & ((1 << LGP) - 1)) + HALO, u)) = o.w;
        }
        if (KS == 3) {                                  // the halo is the convolution's zero padding
            for (int i = tid; i < 36 * UPP; i += IM_NT) {
                const int h = i / UPP, hu = i & (UPP - 1);
                int wr, wc;
                if (h < 10) { wr = 0; wc = h; } else if (h < 20) { wr = 9; wc = h - 10; } else if (h < 28) { wr = h - 19; wc = 0; } else { wr = h - 27; wc = 9; }
                *reinterpret_cast<u32x4*>(win + im_off<PC, ROWB>(wr, wc, hu)) = u32x4{0u, 0u, 0u, 0u};
            }
        }
    }
};

// prologue vectors of a segment for this thread's channel unit (unconditional loads from a valid address, selected
// afterwards: a branch here makes the compiler wait for them before it requests the weights)
template <int C>
static __device__ __forceinline__ void im_affine(const mcgen_seg_t& sg, int n0, int tid, const float* anyf, float (&sc)[8], float (&sh)[8]) {
    const int u = tid & (C / 8 - 1);
    // group_n > 0: BatchNorm statistics groups of group_n images each; the workgroup's images lie in one group (host check)
    const size_t row = sg.group_n > 0 ? (size_t)(n0 / sg.group_n) * C : 0;
    load8f(sg.scale ? sg.scale + row + u * 8 : anyf, sc);
    load8f(sg.scale ? sg.shift + row + u * 8 : anyf, sh);
}
template <int C>
static __device__ __forceinline__ void im_code(const mcgen_seg_t& sg, int n, int tid, const float* anyf, float (&cd)[8]) {
    const int u = tid & (C / 8 - 1);
    load8f(sg.code ? sg.code + (size_t)n * C + u * 8 : anyf, cd);
}

template <int C0, int KS0, int C1, int KS1, int IMGS>
__global__ __launch_bounds__(IM_NT, 1)
void conv_img_kernel(const mcgen_conv_t p) {
    using G = ImCfg<C0, KS0, C1, KS1, IMGS>;
    using S0 = ImSeg<C0, KS0>;
    using S1 = ImSeg<(C1 ? C1 : 128), (C1 ? KS1 : 1)>;
    constexpr int KH = G::KH, KPW = G::KPW, PF = G::PF;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wv & 1, kh = wv >> 1;
    const int l15 = lane & 15, lg = lane >> 4;
    const int n0 = blockIdx.x * IMGS, co_t = blockIdx.y * IM_COT;
    const float* anyf = reinterpret_cast<const float*>(p.w);

    // ---- inputs first (they return first): raw pixels, then the prologue vectors
    u32x4 raw0[IMGS][S0::NI], raw1[IMGS][S1::NI];
#pragma unroll
    for (int k = 0; k < IMGS; ++k) {
        S0::load(p.seg[0], n0 + k, tid, raw0[k]);
        if (C1) S1::load(p.seg[1], n0 + k, tid, raw1[k]);
    }
    float sc0[8], sh0[8], sc1[8], sh1[8], cd0[IMGS][8], cd1[IMGS][8];
    im_affine<C0>(p.seg[0], n0, tid, anyf, sc0, sh0);
    if (C1) im_affine<(C1 ? C1 : 128)>(p.seg[1], n0, tid, anyf, sc1, sh1);
#pragma unroll
    for (int k = 0; k < IMGS; ++k) {
        im_code<C0>(p.seg[0], n0 + k, tid, anyf, cd0[k]);
        if (C1) im_code<(C1 ? C1 : 128)>(p.seg[1], n0 + k, tid, anyf, cd1[k]);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- this wave's weight fragments: K step ks of the image [K step][co][32] -> rows co_t + 32 cg + 16 a + l15, elements 8 lg ..
    const int ks0 = kh * KPW;
    const bf16_t* wimg = reinterpret_cast<const bf16_t*>(p.w) + ((size_t)ks0 * p.Cout_w + co_t + 32 * cg + l15) * MCGEN_CK + lg * 8;
    const size_t wstep = (size_t)p.Cout_w * MCGEN_CK;
    bf16x8 wf[KPW][2];
#pragma unroll
    for (int i = 0; i < PF; ++i) {
        wf[i][0] = *reinterpret_cast<const bf16x8*>(wimg + i * wstep);
        wf[i][1] = *reinterpret_cast<const bf16x8*>(wimg + i * wstep + 16 * MCGEN_CK);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- prologue, window stores
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        sc0[e] = p.seg[0].scale ? sc0[e] : 1.f; sh0[e] = p.seg[0].scale ? sh0[e] : 0.f;
        if (C1) { sc1[e] = p.seg[1].scale ? sc1[e] : 1.f; sh1[e] = p.seg[1].scale ? sh1[e] : 0.f; }
    }
#pragma unroll
    for (int k = 0; k < IMGS; ++k) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            cd0[k][e] = p.seg[0].code ? cd0[k][e] : 1.f;
            if (C1) cd1[k][e] = p.seg[1].code ? cd1[k][e] : 1.f;
        }
        char* w0 = smem + k * (G::WIN0 + G::WIN1);
        S0::write(p.seg[0], tid, raw0[k], sc0, sh0, cd0[k], w0);
        if (C1) S1::write(p.seg[1], tid, raw1[k], sc1, sh1, cd1[k], w0 + G::WIN0);
    }
    __syncthreads();

    f32x4 acc[IMGS][2][4];
#pragma unroll
    for (int k = 0; k < IMGS; ++k)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int f = 0; f < 4; ++f) acc[k][a][f] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the epilogue's operands: one 8-channel unit of one pixel per image and thread
    const int eu = tid & 7, epx = tid >> 3;
    const int co = co_t + eu * 8;
    u32x4 graw[IMGS], rraw[IMGS];
    float oc[IMGS][8], bs[8], b2[8];
    const int r0 = l15 & 1, c0 = l15 >> 1;              // this lane's pixel inside fragment f: (2 f + r0, c0)
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        if (i + PF < KPW) {
            wf[i + PF][0] = *reinterpret_cast<const bf16x8*>(wimg + (i + PF) * wstep);
            wf[i + PF][1] = *reinterpret_cast<const bf16x8*>(wimg + (i + PF) * wstep + 16 * MCGEN_CK);
        }
        if (i == (KPW > PF ? KPW - PF : 0)) {           // behind the last weight request: gate, residual, codes, biases
            const bf16_t* any16 = reinterpret_cast<const bf16_t*>(p.w);
#pragma unroll
            for (int k = 0; k < IMGS; ++k) {
                const size_t off = ((size_t)(n0 + k) * 64 + epx) * p.Cy + co;
                graw[k] = *reinterpret_cast<const u32x4*>(p.gate_x ? reinterpret_cast<const bf16_t*>(p.gate_x) + off : any16);
                rraw[k] = *reinterpret_cast<const u32x4*>(p.res ? reinterpret_cast<const bf16_t*>(p.res) + off : any16);
                load8f(p.ocode ? p.ocode + (size_t)(n0 + k) * p.Cout + co : anyf, oc[k]);
            }
            load8f(p.bias ? p.bias + co : anyf, bs);
            load8f(p.bias2 ? p.bias2 + co : anyf, b2);
        }
        const int ks = ks0 + i;                          // wave-uniform: segment, chunk, tap of this K step
        int xoff[4];                                       // byte offsets of this lane's four window fragments inside an image
        if (C1 == 0 || ks < G::KT0) {
            const int q = ks / G::T0, tap = ks - q * G::T0;
            const int dh = KS0 == 3 ? tap / 3 : 0, dw = KS0 == 3 ? tap - 3 * dh : 0;
#pragma unroll
            for (int f = 0; f < 4; ++f) xoff[f] = im_off<S0::PC, S0::ROWB>(2 * f + r0 + dh, c0 + dw, q * 4 + lg);
        } else {
            const int k1 = ks - G::KT0;
            const int q = k1 / G::T1, tap = k1 - q * G::T1;
            const int dh = KS1 == 3 ? tap / 3 : 0, dw = KS1 == 3 ? tap - 3 * dh : 0;
#pragma unroll
            for (int f = 0; f < 4; ++f)
                xoff[f] = G::WIN0 + (KS1 == IM_UP1 ? im_off<S1::PC, S1::ROWB>((2 * f + r0) >> 1, c0 >> 1, q * 4 + lg)      // (pixel (r, c) reads source (r >> 1, c >> 1))
                                                   : im_off<S1::PC, S1::ROWB>(2 * f + r0 + dh, c0 + dw, q * 4 + lg));
        }
#pragma unroll
        for (int k = 0; k < IMGS; ++k) {
            bf16x8 xf[4];
#pragma unroll
            for (int f = 0; f < 4; ++f) xf[f] = *reinterpret_cast<const bf16x8*>(smem + k * (G::WIN0 + G::WIN1) + xoff[f]);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                acc[k][0][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][0], xf[f], acc[k][0][f], 0, 0, 0);
                acc[k][1][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][1], xf[f], acc[k][1][f], 0, 0, 0);
            }
        }
    }
    // ---- the K parts meet: D[co = 32 cg + 16 a + 4 lg + r][pixel (2 f + r0, c0)] -> ebuf[kh][image][pixel][co]
    __syncthreads();                                      // (the exchange buffers lie over the windows)
    float* ebuf = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int k = 0; k < IMGS; ++k) {
        float* eb = ebuf + kh * (G::EBUF / 4) + k * (64 * IM_EP);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int f = 0; f < 4; ++f)
                *reinterpret_cast<f32x4*>(eb + ((2 * f + r0) * 8 + c0) * IM_EP + 32 * cg + 16 * a + 4 * lg) = acc[k][a][f];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) bs[e] = (p.bias ? bs[e] : 0.f) + (p.bias2 ? b2[e] : 0.f);
    float s1[IMGS][8], s2[IMGS][8];
#pragma unroll
    for (int k = 0; k < IMGS; ++k) {
        const float* e0 = ebuf + k * (64 * IM_EP) + epx * IM_EP + eu * 8;
        float v[8];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 s = *reinterpret_cast<const f32x4*>(e0 + 4 * h);
#pragma unroll
            for (int part = 1; part < KH; ++part) {
                const f32x4 t = *reinterpret_cast<const f32x4*>(e0 + part * (G::EBUF / 4) + 4 * h);
                s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[4 * h + e] = fmaf(s[e], p.alpha, bs[4 * h + e]) * (p.ocode ? oc[k][4 * h + e] : 1.f);
        }
        if (p.gate_x) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (!(__uint_as_float(graw[k][e] << 16) > 0.f)) v[2 * e] = 0.f;
                if (!(__uint_as_float(graw[k][e] & 0xffff0000u) > 0.f)) v[2 * e + 1] = 0.f;
            }
        }
        if (p.res) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[2 * e] += __uint_as_float(rraw[k][e] << 16);
                v[2 * e + 1] += __uint_as_float(rraw[k][e] & 0xffff0000u);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[k][e] = v[e]; s2[k][e] = v[e] * v[e]; }
        Elem<bf16_t>::store8(reinterpret_cast<bf16_t*>(p.y) + ((size_t)(n0 + k) * 64 + epx) * p.Cy + co, v);
    }
    if (p.stats_mode == 1) {
        // BatchNorm partial sums of one image = one row of `stats` ([image][2][Cy]): 8 pixels per wave by lane exchange
        // (lane bits 3..5 are the pixel), the 8 waves through LDS, in a fixed order
#pragma unroll
        for (int k = 0; k < IMGS; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#pragma unroll
                for (int m = 8; m < 64; m <<= 1) { s1[k][e] += __shfl_xor(s1[k][e], m); s2[k][e] += __shfl_xor(s2[k][e], m); }
            }
        __syncthreads();                                  // everyone is done reading the exchange buffers
        float* red = reinterpret_cast<float*>(smem);      // [wave][image][2][64]
        if (lane < 8) {
#pragma unroll
            for (int k = 0; k < IMGS; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    red[((wv * IMGS + k) * 2 + 0) * IM_COT + lane * 8 + e] = s1[k][e];
                    red[((wv * IMGS + k) * 2 + 1) * IM_COT + lane * 8 + e] = s2[k][e];
                }
        }
        __syncthreads();
        if (tid < IMGS * 2 * IM_COT) {
            const int c = tid & (IM_COT - 1), s = (tid / IM_COT) & 1, k = tid / (2 * IM_COT);
            float a = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) a += red[((w * IMGS + k) * 2 + s) * IM_COT + c];
            p.stats[((size_t)(n0 + k) * 2 + s) * p.Cy + co_t + c] = a;
        }
    }
}

struct ImPick { int c0, k0, c1, k1, imgs; };

static bool im_pick(const mcgen_conv_t* p, int dtype, ImPick* out) {
    if (dtype != MCGEN_BF16 || p->w_layout != 0 || p->nseg < 1 || p->nseg > 2) return false;
    if (p->H != 8 || p->W != 8 || p->Cout % IM_COT || p->Cout_w != p->Cout || p->Cy != p->Cout) return false;
    if (p->pool || p->gscale || p->tanh_out || p->ycmap || p->stats_mode > 1) return false;       // (p->stats: validated by mcgen_conv_fused)
    for (int s = 0; s < p->nseg; ++s) {
        const mcgen_seg_t& g = p->seg[s];
        if ((g.ksize != 1 && g.ksize != 3) || g.cmap || (g.C != 128 && g.C != 256)) return false;
        if (g.group_n < 0 || (g.group_n > 0 && p->N % g.group_n)) return false;
    }
    ImPick k{p->seg[0].C, p->seg[0].ksize, p->nseg == 2 ? p->seg[1].C : 0, p->nseg == 2 ? p->seg[1].ksize : 0, 1};
    const bool known = (k.c1 == 0 && k.k0 == 3) || (k.c1 == 0 && k.k0 == 1) || (k.c0 == 256 && k.k0 == 1 && k.c1 == 128 && k.k1 == 3) ||
                       (k.c0 == 256 && k.k0 == 3 && k.c1 == 256 && k.k1 == 1 && !p->seg[0].ups && p->seg[1].ups);      // GenResBlock conv_b ++ shortcut
    if (!known) return false;
    // two images per workgroup while the launch still covers the chip (halves the weight bytes per pixel), the windows of
    // both fit in LDS, and the pair shares its BatchNorm group
    const bool up1 = k.c1 && k.k1 == 1 && p->seg[1].ups;             // (only the form above: its shortcut window holds the 4x4 source)
    const int win = (k.k0 == 3 ? 100 : 64) * k.c0 * 2 + (k.c1 ? (k.k1 == 3 ? 100 : (up1 ? 16 : 64)) * k.c1 * 2 : 0);
    bool pair_ok = p->N % 2 == 0 && (long)p->N * (p->Cout / IM_COT) >= 512 && 2 * win <= 160 * 1024;
    for (int s = 0; s < p->nseg; ++s) if (p->seg[s].group_n > 0 && p->seg[s].group_n % 2) pair_ok = false;
    if (pair_ok) k.imgs = 2;
    *out = k;
    return true;
}

template <int C0, int KS0, int C1, int KS1, int IMGS>
static int launch_img(const mcgen_conv_t* p, hipStream_t st) {
    using G = ImCfg<C0, KS0, C1, KS1, IMGS>;
    auto k = conv_img_kernel<C0, KS0, C1, KS1, IMGS>;
    static bool raised = false;
    if (!raised) {
        raised = true;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        if (e != hipSuccess) return mcgen_fail("conv_smap: cannot raise LDS limit: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k, dim3(p->N / IMGS, p->Cout / IM_COT), dim3(IM_NT), G::LDS, st, *p);
    MCGEN_LAUNCH_CHECK("conv_smap");
    return 0;
}

}  // namespace

// 1 when mcgen_conv_fused hands `p` to the whole-image kernel (declared in conv_tile.h for conv_fused.hip)
int mcgen_conv_smap_ok(const mcgen_conv_t* p, int dtype) {
    ImPick k;
    return im_pick(p, dtype, &k) ? 1 : 0;
}

int mcgen_conv_smap(const mcgen_conv_t* p, hipStream_t st) {
    ImPick k;
    if (!im_pick(p, MCGEN_BF16, &k)) return mcgen_fail("conv_smap: not a whole-image launch");
#define IM_CASE(A, B, C, D)                                                                                  \
    if (k.c0 == A && k.k0 == B && k.c1 == C && k.k1 == D)                                                    \
        return k.imgs == 2 ? launch_img<A, B, C, D, 2>(p, st) : launch_img<A, B, C, D, 1>(p, st);
    IM_CASE(128, 3, 0, 0) IM_CASE(256, 3, 0, 0) IM_CASE(128, 1, 0, 0) IM_CASE(256, 1, 0, 0) IM_CASE(256, 1, 128, 3)
#undef IM_CASE
    if (k.c0 == 256 && k.k0 == 3 && k.c1 == 256 && k.k1 == 1)
        return k.imgs == 2 ? launch_img<256, 3, 256, IM_UP1, 2>(p, st) : launch_img<256, 3, 256, IM_UP1, 1>(p, st);
    return mcgen_fail("conv_smap: no instantiation");
}
