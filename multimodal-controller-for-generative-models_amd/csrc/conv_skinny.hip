// Skinny 3x3 convolution (Cout <= 16) over a deep K on SMALL maps, K split over the waves of the workgroup (gfx950, bf16).
//
//   y[n, h, w, co] = alpha * sum_{tap, ci} prologue(x)[n, h + dh, w + dw, ci] * W[co, ci, tap] + bias[co] (+ res)
//
// MCGlow's ZeroConv2d (512 -> C, mcglow.py:119-130) and the input gradient of a coupling net's first convolution
// (512 -> C/2, autograd of mcglow.py:148) on 16x16 / 8x8 / 4x4 maps: a GEMM with N <= 16 and K = 9 * 512.  On the
// general tiles this is ONE serial chain per workgroup -- 16 chunks x 9 taps behind barriers, 37-41 us per launch at
// 58-930 GB/s whatever the map (gpurun_out/r3l_glow_shapes.json) -- with 32 ... 512 workgroups to hide it.  Here every
// wave of a 16-wave workgroup owns ONE 32-channel chunk: it loads its own window slice (tile + halo, 64 bytes per pixel),
// applies the prologue (ActNorm affine, ReLU, MultimodalController code: modules.py:71-76), parks it in its private LDS
// region, runs its 9 taps x (pixels / 16) MFMAs with the tap's 16 x 32 weight fragment straight from the weight image
// (L2-resident, 1 KB per wave-instruction), and the 16 partial accumulator sets meet in LDS once.  The chain is one
// global round trip + 36 MFMAs + one LDS exchange.
#include "conv_tile.h"

namespace {

constexpr int SK_WAVES = 16, SK_NT = 64 * SK_WAVES;
constexpr int SK_PITCH = 80;                       // bytes per window pixel: 32 channels + 16 (16-byte aligned rows, odd multiple of 16)

template <int LGW>
struct SkGeo {
    static constexpr int W = 1 << LGW, H = W, HW = W * H;
    static constexpr int BM = (HW >= 64) ? 64 : 32;                    // pixels per tile (4x4 maps: two images)
    static constexpr int TI = BM > HW ? BM / HW : 1, TH = BM > HW ? H : BM / W;
    static constexpr int LGTHW = (BM > HW) ? 2 * LGW : (BM == 64 ? 6 : 5);
    static constexpr int PR = TH + 2, PC = W + 2, PP = TI * PR * PC;
    static constexpr int NIT = (PP * 4 + 63) / 64;                     // 16-byte window units per lane
    static constexpr int FM = BM / 16;                                 // pixel fragments
    static constexpr int WIN = ((PP * SK_PITCH + 127) / 128) * 128;     // one wave's window region
    static constexpr int LDS = SK_WAVES * (WIN > FM * 1024 ? WIN : FM * 1024);
    static_assert(LDS <= 160 * 1024, "LDS");
};

template <int LGW>
__global__ __launch_bounds__(SK_NT, 1)
void conv_skinny_kernel(const mcgen_conv_t p) {
    using G = SkGeo<LGW>;
    constexpr int W = G::W, H = G::H, HW = G::HW, BM = G::BM, TI = G::TI, PR = G::PR, PC = G::PC, PP = G::PP, NIT = G::NIT, FM = G::FM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);             // wave = input-channel chunk
    const int l15 = lane & 15, lg = lane >> 4;
    const mcgen_seg_t sg = p.seg[0];
    const int nchunk = (sg.C + MCGEN_CK - 1) / MCGEN_CK;
    const int tile = blockIdx.x;
    const int pix0 = tile * BM;
    const int n0 = pix0 >> (2 * LGW), h0 = (TI == 1) ? ((pix0 & (HW - 1)) >> LGW) : 0;
    char* win = smem + q * (G::LDS / SK_WAVES);
    f32x4 acc[FM];
#pragma unroll
    for (int f = 0; f < FM; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (q < nchunk) {
        // ---- this wave's window slice: unit (window pixel pp, 8-channel group u = lane & 3), prologue, LDS
        const int u = lane & 3, cx = q * MCGEN_CK + u * 8;
        const bool cok = cx < sg.C;
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = 1.f; sh[e] = 0.f; }
        if (sg.scale && cok) { load8f(sg.scale + cx, sc); load8f(sg.shift + cx, sh); }
        const float relu_lo = sg.relu ? 0.f : -__builtin_inff();
        const bf16_t* xs = reinterpret_cast<const bf16_t*>(sg.x);
        u32x4 raw[NIT];
        int lds_off[NIT], img_n[NIT];
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            const int pp = (lane + 64 * k) >> 2;
            const int ti = pp / (PR * PC), rem = pp - ti * (PR * PC);
            const int pr = rem / PC, pc = rem - pr * PC;
            const int n = n0 + ti, h = h0 + pr - 1, w = pc - 1;
            const bool ok = pp < PP && cok && n < p.N && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
            lds_off[k] = pp < PP ? pp * SK_PITCH + u * 16 : -1;
            img_n[k] = ok ? n : -1;
            const bf16_t* src = ok ? xs + ((size_t)(n * H + h) * W + w) * sg.C + cx : xs;
            raw[k] = *reinterpret_cast<const u32x4*>(src);
        }
#pragma unroll
        for (int k = 0; k < NIT; ++k) {
            if (lds_off[k] < 0) continue;
            float cd[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) cd[e] = 1.f;
            if (sg.code && img_n[k] >= 0) load8f(sg.code + (size_t)img_n[k] * sg.C + cx, cd);
            union { bf16x8 h; u32x4 w; } o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v0 = fmaxf(fmaf(__uint_as_float(raw[k][e] << 16), sc[2 * e], sh[2 * e]), relu_lo) * cd[2 * e];
                const float v1 = fmaxf(fmaf(__uint_as_float(raw[k][e] & 0xffff0000u), sc[2 * e + 1], sh[2 * e + 1]), relu_lo) * cd[2 * e + 1];
                o.h[2 * e] = (bf16_t)v0; o.h[2 * e + 1] = (bf16_t)v1;
            }
            if (img_n[k] < 0) o.w = u32x4{0u, 0u, 0u, 0u};                 // zero padding / channels beyond C
            *reinterpret_cast<u32x4*>(win + lds_off[k]) = o.w;
        }
        // ---- 9 taps: A = the tap's 16 co x 32 ci weight fragment (image [chunk][tap][co_w = 16][32]), B = window fragments
        const bf16_t* wimg = reinterpret_cast<const bf16_t*>(p.w) + ((size_t)q * 9 * 16 + l15) * MCGEN_CK + lg * 8;
        bf16x8 wf[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(wimg + (size_t)j * 16 * MCGEN_CK);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // this wave's window stores (the region is private to the wave)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < FM; ++f) {
            const int m = f * 16 + l15;                                  // the lane's pixel (column of B)
            const int ti = m >> G::LGTHW, rem = m & ((1 << G::LGTHW) - 1);
            const int r = rem >> LGW, c = rem & (W - 1);
            const char* b0 = win + ((ti * PR + r) * PC + c) * SK_PITCH + lg * 16;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(b0 + ((j / 3) * PC + (j % 3)) * SK_PITCH);
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf, acc[f], 0, 0, 0);
            }
        }
    }
    // ---- the chunks' partial sums meet in LDS: [wave][fragment][lane] f32x4, then thread (pixel, 8-channel unit) finishes
    __syncthreads();                                                   // every wave is done with its window region
    f32x4* part = reinterpret_cast<f32x4*>(smem);
#pragma unroll
    for (int f = 0; f < FM; ++f) part[(q * FM + f) * 64 + lane] = acc[f];
    __syncthreads();
    const int units = p.Cy >> 3;                                       // 16-byte output units per pixel (1 or 2)
    if (tid < BM * units) {
        const int m = tid / units, uo = tid - m * units;
        const int pix = pix0 + m;
        if (pix < p.N * HW) {
            const int f = m >> 4, px = m & 15;
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int co = uo * 8 + e;                              // D[co = 4 lg + r][px = l15]
                const int ln = (co >> 2) * 16 + px, r = co & 3;
                float s = 0.f;
                for (int w = 0; w < nchunk; ++w) s += part[(w * FM + f) * 64 + ln][r];
                v[e] = (co < p.Cout) ? fmaf(s, p.alpha, p.bias ? p.bias[co] : 0.f) : 0.f;
            }
            bf16_t* yp = reinterpret_cast<bf16_t*>(p.y) + (size_t)pix * p.Cy + uo * 8;
            if (p.res) {
                float rv[8];
                Elem<bf16_t>::load8(reinterpret_cast<const bf16_t*>(p.res) + (size_t)pix * p.Cy + uo * 8, rv);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += rv[e];
            }
            Elem<bf16_t>::store8(yp, v);
        }
    }
}

template <int LGW>
static int launch_skinny(const mcgen_conv_t* p, hipStream_t st) {
    using G = SkGeo<LGW>;
    const long M = (long)p->N * G::HW;
    const int tiles = (int)((M + G::BM - 1) / G::BM);
    auto k = conv_skinny_kernel<LGW>;
    static bool raised = false;
    if (!raised) {
        raised = true;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        if (e != hipSuccess) return mcgen_fail("conv_skinny: cannot raise LDS limit: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(k, dim3(tiles), dim3(SK_NT), G::LDS, st, *p);
    MCGEN_LAUNCH_CHECK("conv_skinny");
    return 0;
}

}  // namespace

// 1 when mcgen_conv_fused hands `p` to the split-K skinny kernel (declared in conv_tile.h for conv_fused.hip)
int mcgen_conv_skinny_ok(const mcgen_conv_t* p, int dtype) {
    if (dtype != MCGEN_BF16 || p->nseg != 1 || p->w_layout != 0) return 0;
    const mcgen_seg_t& g = p->seg[0];
    if (g.ksize != 3 || g.ups || g.group_n || g.cmap || g.C % 8 || g.C < 128 || g.C > SK_WAVES * MCGEN_CK) return 0;
    if (p->Cout_w != 16 || p->Cy > 16 || p->H != p->W || (p->W != 4 && p->W != 8 && p->W != 16)) return 0;
    if (p->pool || p->ocode || p->gate_x || p->stats_mode || p->tanh_out || p->ycmap || p->bias2) return 0;
    if (((long)p->N * p->H * p->W) % (p->W == 4 ? 32 : 64)) return 0;
    return 1;
}

int mcgen_conv_skinny(const mcgen_conv_t* p, hipStream_t st) {
    switch (p->W) {
        case 16: return launch_skinny<4>(p, st);
        case 8: return launch_skinny<3>(p, st);
        default: return launch_skinny<2>(p, st);
    }
}
