// 1x1 convolution 512 -> 512 with the pixel tile resident in LDS (gfx950, bf16).
//
//   y[m, co] = epilogue( alpha * sum_ci prologue(x)[m, ci] * W[co, ci] ),   m = (n, h, w) flattened
//
// MCGlow's coupling networks (mcglow.py:133-160: Conv3x3 -> ActNorm -> ReLU -> MC -> Conv1x1(512, 512) -> ActNorm -> ReLU
// -> MC -> ZeroConv) run this GEMM 48 times forward and 48 times as an input gradient per training step, on 16x16, 8x8
// and 4x4 maps (M = 32768 / 8192 / 2048 pixels at batch 128).  A 1x1 convolution has ONE tap per 32-channel chunk, so the
// general tiles spend a barrier round (stage a chunk's window, wait, DMA its weights, wait) per MFMA step: 61-64 us for
// the 17 GFLOP of the 16x16 level (270 TFLOP/s), 31-38 us at 8x8, 16-20 us at 4x4 (gpurun_out/r3l_glow_shapes.json).
//
// Here a workgroup owns BM = 16 PXF pixels (128 / 32 / 16: a tile never leaves its image) and ALL 512 output channels:
//   * the tile's 512 input channels are staged into LDS once (1 KB per pixel, prologue applied: ActNorm affine, ReLU,
//     MultimodalController code -- modules.py:71-76), XOR-swizzled (unit ^ (pixel & 15)) for conflict-free ds_read_b128;
//     every activation is read from HBM exactly once per launch;
//   * wave w owns output channels 64 w .. 64 w + 63: per K step (32 channels) 4 weight fragments straight from L2 into
//     registers (four K steps ahead), PXF pixel fragments from LDS, 4 PXF MFMAs -- no barrier inside the K loop;
//   * the epilogue is wave-private: the wave's 64 channels x 32 pixels pass through its own LDS region and leave as
//     16-byte units (bias, output code, ReLU gate through the ActNorm affine, residual, ActNorm-gradient partial sums).
#include "conv_tile.h"

namespace {

constexpr int P1_NT = 512, P1_K = 512, P1_KS = P1_K / 32, P1_ROWB = P1_K * 2;
constexpr int P1_EP = 64 + 4;                          // floats per pixel row of a wave's epilogue region

template <int PXF>
struct P1Cfg {
    static constexpr int BM = 16 * PXF;
    static constexpr int NI = BM / 8;                   // staging items per thread (one 16-byte unit each)
    static constexpr int RPX = BM < 32 ? BM : 32;       // pixels per epilogue round (bounds the live registers beside the accumulators)
    static constexpr int ROUNDS = BM / RPX, FPR = RPX / 16;
    static constexpr int EPW = RPX * P1_EP * 4;         // one wave's epilogue region
    static constexpr int LDS = BM * P1_ROWB > 8 * EPW ? BM * P1_ROWB : 8 * EPW;
    static_assert(LDS <= 160 * 1024, "LDS");
};

template <int PXF>
__global__ __launch_bounds__(P1_NT, 1)
void conv_px1_kernel(const mcgen_conv_t p) {
    using G = P1Cfg<PXF>;
    constexpr int BM = G::BM, NI = G::NI;
    constexpr int P1_PF = PXF == 8 ? 3 : 4;             // weight K steps in flight
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const mcgen_seg_t sg = p.seg[0];
    const int HW = p.H * p.W;
    const size_t m0 = (size_t)blockIdx.x * BM;
    const int n = (int)(m0 / HW);                        // the tile's image (BM divides H * W)
    const float* anyf = reinterpret_cast<const float*>(p.w);

    // ---- the pixel tile: thread = 16-byte unit u of pixels (tid >> 6) + 8 k
    const int u = tid & 63, px0 = tid >> 6;
    u32x4 raw[NI];
    {
        const bf16_t* xs = reinterpret_cast<const bf16_t*>(sg.x) + (m0 + px0) * P1_K + u * 8;
#pragma unroll
        for (int k = 0; k < NI; ++k) raw[k] = *reinterpret_cast<const u32x4*>(xs + (size_t)8 * k * P1_K);
    }
    float sc[8], sh[8], cd[8];
    load8f(sg.scale ? sg.scale + u * 8 : anyf, sc);
    load8f(sg.scale ? sg.shift + u * 8 : anyf, sh);
    load8f(sg.code ? sg.code + (size_t)n * P1_K + u * 8 : anyf, cd);
    __builtin_amdgcn_sched_barrier(0);
    // ---- this wave's weight fragments: image [K step][co][32], rows 64 wv + 16 a + l15, elements 8 lg ..
    const bf16_t* wimg = reinterpret_cast<const bf16_t*>(p.w) + ((size_t)64 * wv + l15) * MCGEN_CK + lg * 8;
    const size_t wstep = (size_t)p.Cout_w * MCGEN_CK;
    bf16x8 wf[P1_KS][4];
#pragma unroll
    for (int i = 0; i < P1_PF; ++i)
#pragma unroll
        for (int a = 0; a < 4; ++a) wf[i][a] = *reinterpret_cast<const bf16x8*>(wimg + i * wstep + a * 16 * MCGEN_CK);
    __builtin_amdgcn_sched_barrier(0);
    {
        const float relu_lo = sg.relu ? 0.f : -__builtin_inff();
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = sg.scale ? sc[e] : 1.f; sh[e] = sg.scale ? sh[e] : 0.f; cd[e] = sg.code ? cd[e] : 1.f; }
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int px = px0 + 8 * k;
            union { bf16x8 h; u32x4 w; } o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v0 = fmaxf(fmaf(__uint_as_float(raw[k][e] << 16), sc[2 * e], sh[2 * e]), relu_lo) * cd[2 * e];
                const float v1 = fmaxf(fmaf(__uint_as_float(raw[k][e] & 0xffff0000u), sc[2 * e + 1], sh[2 * e + 1]), relu_lo) * cd[2 * e + 1];
                o.h[2 * e] = (bf16_t)v0; o.h[2 * e + 1] = (bf16_t)v1;
            }
            *reinterpret_cast<u32x4*>(smem + px * P1_ROWB + ((u ^ (px & 15)) << 4)) = o.w;
        }
    }
    __syncthreads();

    f32x4 acc[4][PXF];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int f = 0; f < PXF; ++f) acc[a][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < P1_KS; ++i) {
        if (i + P1_PF < P1_KS) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
                wf[i + P1_PF][a] = *reinterpret_cast<const bf16x8*>(wimg + (i + P1_PF) * wstep + a * 16 * MCGEN_CK);
        }
        constexpr int XB = PXF < 4 ? PXF : 4;               // pixel fragments per LDS batch (bounds the live registers)
#pragma unroll
        for (int fb = 0; fb < PXF; fb += XB) {
            bf16x8 xf[XB];
#pragma unroll
            for (int f = 0; f < XB; ++f)
                xf[f] = *reinterpret_cast<const bf16x8*>(smem + (16 * (fb + f) + l15) * P1_ROWB + (((i * 4 + lg) ^ l15) << 4));
#pragma unroll
            for (int f = 0; f < XB; ++f)
#pragma unroll
                for (int a = 0; a < 4; ++a)
                    acc[a][fb + f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i][a], xf[f], acc[a][fb + f], 0, 0, 0);
        }
    }

    // ---- epilogue, private to the wave: lane = 8-channel unit eu of pixels (lane >> 3) + 8 j of the round
    const int eu = lane & 7, ep0 = lane >> 3;
    const int co = 64 * wv + eu * 8;
    float bs[8], b2[8], oc[8], gsc[8], gsh[8], gme[8], grs[8];
    load8f(p.bias ? p.bias + co : anyf, bs);
    load8f(p.bias2 ? p.bias2 + co : anyf, b2);
    load8f(p.ocode ? p.ocode + (size_t)n * p.Cout + co : anyf, oc);
    load8f(p.gscale ? p.gscale + co : anyf, gsc);
    load8f(p.gscale ? p.gshift + co : anyf, gsh);
    load8f(p.gmean ? p.gmean + co : anyf, gme);
    load8f(p.grstd ? p.grstd + co : anyf, grs);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        bs[e] = (p.bias ? bs[e] : 0.f) + (p.bias2 ? b2[e] : 0.f);
        oc[e] = p.ocode ? oc[e] : 1.f;
        gsc[e] = p.gscale ? gsc[e] : 1.f; gsh[e] = p.gscale ? gsh[e] : 0.f;
        gme[e] = p.gmean ? gme[e] : 0.f; grs[e] = p.grstd ? grs[e] : 0.f;
    }
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    __syncthreads();                                      // every wave is done reading the pixel tile
    float* eb = reinterpret_cast<float*>(smem + wv * G::EPW);
    const bf16_t* any16 = reinterpret_cast<const bf16_t*>(p.w);
#pragma unroll
    for (int rd = 0; rd < G::ROUNDS; ++rd) {
        constexpr int NJ = G::RPX / 8;
        // gate / residual operands of the round, requested before the LDS pass
        u32x4 graw[NJ], rraw[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const size_t off = (m0 + rd * G::RPX + ep0 + 8 * j) * p.Cy + co;
            graw[j] = *reinterpret_cast<const u32x4*>(p.gate_x ? reinterpret_cast<const bf16_t*>(p.gate_x) + off : any16);
            rraw[j] = *reinterpret_cast<const u32x4*>(p.res ? reinterpret_cast<const bf16_t*>(p.res) + off : any16);
        }
        if (rd > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // the previous round's reads of the region
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int f = 0; f < G::FPR; ++f)
                *reinterpret_cast<f32x4*>(eb + (16 * f + l15) * P1_EP + 16 * a + 4 * lg) = acc[a][rd * G::FPR + f];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // wave-private region: no barrier
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const float* e0 = eb + (ep0 + 8 * j) * P1_EP + eu * 8;
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(e0), a1 = *reinterpret_cast<const f32x4*>(e0 + 4);
            float v[8], gx[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] = fmaf(a0[e], p.alpha, bs[e]) * oc[e];
                v[4 + e] = fmaf(a1[e], p.alpha, bs[4 + e]) * oc[4 + e];
                gx[2 * e] = __uint_as_float(graw[j][e] << 16); gx[2 * e + 1] = __uint_as_float(graw[j][e] & 0xffff0000u);
            }
            if (p.gate_x) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (fmaf(gx[e], gsc[e], gsh[e]) > 0.f) ? v[e] : 0.f;
                if (p.stats_mode == 2) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] = fmaf(v[e], (gx[e] - gme[e]) * grs[e], s2[e]); }
                }
            }
            if (p.res) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] += __uint_as_float(rraw[j][e] << 16);
                    v[2 * e + 1] += __uint_as_float(rraw[j][e] & 0xffff0000u);
                }
            }
            if (p.stats_mode == 1) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] = fmaf(v[e], v[e], s2[e]); }
            }
            Elem<bf16_t>::store8(reinterpret_cast<bf16_t*>(p.y) + (m0 + rd * G::RPX + ep0 + 8 * j) * p.Cy + co, v);
        }
    }
    if (p.stats_mode != 0) {
        // the tile's partial sums (one row of `stats`): lanes with equal unit across lane bits 3..5, fixed order
#pragma unroll
        for (int e = 0; e < 8; ++e) {
#pragma unroll
            for (int m = 8; m < 64; m <<= 1) { s1[e] += __shfl_xor(s1[e], m); s2[e] += __shfl_xor(s2[e], m); }
        }
        if (lane < 8) {
            float* st = p.stats + (size_t)blockIdx.x * 2 * p.Cy + co;
            Elem<float>::store8(st, s1);
            Elem<float>::store8(st + p.Cy, s2);
        }
    }
}

static int p1_pxf(const mcgen_conv_t* p, int dtype) {
    if (dtype != MCGEN_BF16 || p->w_layout != 0 || p->nseg != 1) return 0;
    const mcgen_seg_t& g = p->seg[0];
    if (g.ksize != 1 || g.ups || g.group_n || g.cmap || g.C != P1_K) return 0;
    if (p->Cout != 512 || p->Cout_w != 512 || p->Cy != 512 || p->pool || p->tanh_out || p->ycmap) return 0;
    // (no test of p->stats here: mcgen_conv_m_tiles asks before the caller has allocated it; mcgen_conv_fused validates it)
    if (p->stats_mode == 2 && !(p->gate_x && p->gmean && p->grstd)) return 0;
    const int hw = p->H * p->W;
    return hw == 256 ? 8 : hw == 64 ? 2 : hw == 16 ? 1 : 0;
}

template <int PXF>
static int launch_px1(const mcgen_conv_t* p, hipStream_t st) {
    using G = P1Cfg<PXF>;
    auto k = conv_px1_kernel<PXF>;
    static bool raised = false;
    if (!raised) {
        raised = true;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        if (e != hipSuccess) return mcgen_fail("conv_px1: cannot raise LDS limit: %s", hipGetErrorString(e));
    }
    const long M = (long)p->N * p->H * p->W;
    hipLaunchKernelGGL(k, dim3((unsigned)(M / G::BM)), dim3(P1_NT), G::LDS, st, *p);
    MCGEN_LAUNCH_CHECK("conv_px1");
    return 0;
}

}  // namespace

// pixels per tile (= rows of `stats` per M pixels) when mcgen_conv_fused hands `p` to this kernel, else 0
int mcgen_conv_px1_bm(const mcgen_conv_t* p, int dtype) { return 16 * p1_pxf(p, dtype); }

int mcgen_conv_px1(const mcgen_conv_t* p, hipStream_t st) {
    switch (p1_pxf(p, MCGEN_BF16)) {
        case 8: return launch_px1<8>(p, st);
        case 2: return launch_px1<2>(p, st);
        case 1: return launch_px1<1>(p, st);
    }
    return mcgen_fail("conv_px1: not a resident-tile 1x1 launch");
}
