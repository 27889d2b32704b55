// 3x3 convolution of an 8-channel (image) tensor to 128 channels on 32x32 maps (gfx950, bf16).
//
//   y[n, h, w, co] = alpha * sum_{tap, ci < 8} prologue(x)[n, h + dh, w + dw, ci] * W[co, ci, tap] + bias[co]
//
// The first convolution of the discriminator (FirstDisResBlock conv, mcgan.py:72-93: 3 -> 128 on the 32x32 image, channel
// pitch 8): 6 launches per training iteration whose cost is the 67 MB of output (256 images), not the 4.8 GFLOP.  The
// general 128x128 tile runs the 9 taps as 9 K steps of 32 channels (24 of them zero padding) and sends every accumulator
// through LDS as fp32 on the way out: 28 us per launch (rocprofv3) = 2.4 TB/s.  Here:
//   * K is (tap, channel): one MFMA K step = 4 taps x 8 channels, 3 steps for the 9 taps -- the window fragment of a step
//     is ONE 16-byte LDS read per lane (the tap-shifted pixel's 8 channels), the weight fragment one 16-byte read of the
//     standard [tap][co][32] image (its first 8 channels);
//   * a workgroup owns a whole image (or half of one): the window (34 x 34 pixels x 16 B) is staged once, every wave keeps
//     ALL weight fragments (8 x 3) in registers and walks its rows;
//   * the weight rows of a fragment pair are permuted so that a lane's accumulators are 8 CONSECUTIVE output channels
//     (fragment 2 a: row 4 g + r is channel 32 a + 8 g + r, fragment 2 a + 1: + 4): the epilogue is bias, pack, one
//     16-byte global store per fragment pair and pixel -- no LDS round trip.
#include "conv_tile.h"

namespace {

constexpr int C8_NT = 512, C8_W = 32, C8_CO = 128;
constexpr int C8_PC = C8_W + 2;                          // window pixels per row

template <int RPW>                                       // rows per wave: 4 (whole image per workgroup) or 2 (half)
__global__ __launch_bounds__(C8_NT, 1)
void conv_c8_kernel(const mcgen_conv_t p) {
    constexpr int RB = 8 * RPW;                          // rows per workgroup
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int n = blockIdx.x, h0 = blockIdx.y * RB;
    const mcgen_seg_t sg = p.seg[0];
    const float* anyf = reinterpret_cast<const float*>(p.w);

    // ---- weights first (L2-resident, independent of everything): fragment a, K step s = taps 4 s .. 4 s + 3
    // A row i = l15 -> channel 32 (a >> 1) + 8 (i >> 2) + 4 (a & 1) + (i & 3); K slice lg = tap 4 s + lg, channels 0 .. 7
    bf16x8 wf[3][8];
    {
        const bf16_t* wimg = reinterpret_cast<const bf16_t*>(p.w);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int tap = 4 * s + lg;
#pragma unroll
            for (int a = 0; a < 8; ++a) {
                const int co = 32 * (a >> 1) + 8 * (l15 >> 2) + 4 * (a & 1) + (l15 & 3);
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(wimg + ((size_t)(tap < 9 ? tap : 0) * p.Cout_w + co) * MCGEN_CK);
                wf[s][a] = tap < 9 ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
    }
    // ---- the window: rows h0 - 1 .. h0 + RB, pixels -1 .. 32, 16 bytes each; prologue applied, zeros outside the image
    {
        float sc[8], sh[8], cd[8];
        load8f(sg.scale ? sg.scale : anyf, sc);
        load8f(sg.scale ? sg.shift : anyf, sh);
        load8f(sg.code ? sg.code + (size_t)n * 8 : anyf, cd);
        const float relu_lo = sg.relu ? 0.f : -__builtin_inff();
#pragma unroll
        for (int e = 0; e < 8; ++e) { sc[e] = sg.scale ? sc[e] : 1.f; sh[e] = sg.scale ? sh[e] : 0.f; cd[e] = sg.code ? cd[e] : 1.f; }
        const bf16_t* xs = reinterpret_cast<const bf16_t*>(sg.x) + (size_t)n * C8_W * C8_W * 8;
        for (int i = tid; i < (RB + 2) * C8_PC; i += C8_NT) {
            const int wr = i / C8_PC, wc = i - wr * C8_PC;
            const int h = h0 + wr - 1, w = wc - 1;
            u32x4 o = u32x4{0u, 0u, 0u, 0u};
            if ((unsigned)h < (unsigned)C8_W && (unsigned)w < (unsigned)C8_W) {
                const u32x4 raw = *reinterpret_cast<const u32x4*>(xs + ((size_t)h * C8_W + w) * 8);
                union { bf16x8 hh; u32x4 ww; } q;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v0 = fmaxf(fmaf(__uint_as_float(raw[e] << 16), sc[2 * e], sh[2 * e]), relu_lo) * cd[2 * e];
                    const float v1 = fmaxf(fmaf(__uint_as_float(raw[e] & 0xffff0000u), sc[2 * e + 1], sh[2 * e + 1]), relu_lo) * cd[2 * e + 1];
                    q.hh[2 * e] = (bf16_t)v0; q.hh[2 * e + 1] = (bf16_t)v1;
                }
                o = q.ww;
            }
            *reinterpret_cast<u32x4*>(smem + i * 16) = o;
        }
    }
    // bias of this lane's channels: fragment pair ap -> channels 32 ap + 8 lg .. + 7
    float bs[4][8];
#pragma unroll
    for (int ap = 0; ap < 4; ++ap) {
        float b1[8], b2[8];
        load8f(p.bias ? p.bias + 32 * ap + 8 * lg : anyf, b1);
        load8f(p.bias2 ? p.bias2 + 32 * ap + 8 * lg : anyf, b2);
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[ap][e] = (p.bias ? b1[e] : 0.f) + (p.bias2 ? b2[e] : 0.f);
    }
    __syncthreads();

    bf16_t* y = reinterpret_cast<bf16_t*>(p.y) + (size_t)n * C8_W * C8_W * p.Cy;
#pragma unroll 1
    for (int rr = 0; rr < RPW; ++rr) {
        const int row = wv + 8 * rr;                      // row inside the workgroup's block
        // window fragments: pixel (row, 16 f + l15), K slice lg of step s = tap 4 s + lg
        bf16x8 xf[3][2];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int tap = 4 * s + lg, tq = tap < 9 ? tap : 0;
            const int dh = tq / 3, dw = tq - 3 * dh;
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(smem + ((row + dh) * C8_PC + 16 * f + l15 + dw) * 16);
                xf[s][f] = tap < 9 ? v : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        f32x4 acc[8][2];
#pragma unroll
        for (int a = 0; a < 8; ++a)
#pragma unroll
            for (int f = 0; f < 2; ++f) acc[a][f] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int a = 0; a < 8; ++a)
#pragma unroll
                for (int f = 0; f < 2; ++f) acc[a][f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[s][a], xf[s][f], acc[a][f], 0, 0, 0);
        // D[row i = 4 lg + r][pixel l15]: this lane holds channels 32 ap + 8 lg + r (fragment 2 ap) and + 4 + r (2 ap + 1)
        bf16_t* yrow = y + ((size_t)(h0 + row) * C8_W) * p.Cy;
#pragma unroll
        for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int ap = 0; ap < 4; ++ap) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = fmaf(acc[2 * ap][f][r], p.alpha, bs[ap][r]);
                    v[4 + r] = fmaf(acc[2 * ap + 1][f][r], p.alpha, bs[ap][4 + r]);
                }
                Elem<bf16_t>::store8(yrow + (size_t)(16 * f + l15) * p.Cy + 32 * ap + 8 * lg, v);
            }
    }
}

}  // namespace

// 1 when mcgen_conv_fused hands `p` to this kernel (declared in conv_tile.h for conv_fused.hip)
int mcgen_conv_c8_ok(const mcgen_conv_t* p, int dtype) {
    if (dtype != MCGEN_BF16 || p->w_layout != 0 || p->nseg != 1) return 0;
    const mcgen_seg_t& g = p->seg[0];
    if (g.ksize != 3 || g.ups || g.group_n || g.cmap || g.C != 8) return 0;
    if (p->H != C8_W || p->W != C8_W || p->Cout != C8_CO || p->Cout_w != C8_CO || p->Cy != C8_CO) return 0;
    if (p->pool || p->res || p->ocode || p->gate_x || p->stats_mode || p->tanh_out || p->ycmap) return 0;
    return 1;
}

int mcgen_conv_c8(const mcgen_conv_t* p, hipStream_t st) {
    // a whole image per workgroup while the launch still covers the chip, half an image otherwise
    if (p->N >= 192) {
        hipLaunchKernelGGL(conv_c8_kernel<4>, dim3(p->N, 1), dim3(C8_NT), 34 * C8_PC * 16, st, *p);
    } else {
        hipLaunchKernelGGL(conv_c8_kernel<2>, dim3(p->N, 2), dim3(C8_NT), 18 * C8_PC * 16, st, *p);
    }
    MCGEN_LAUNCH_CHECK("conv_c8");
    return 0;
}
