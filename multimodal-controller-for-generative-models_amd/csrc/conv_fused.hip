// Fused direct convolution on MFMA tiles for gfx950 (MI355X).
//
// One workgroup computes a BM-pixel x BN-channel output tile of
//     y = epilogue( sum_seg conv_{3x3|1x1}( prologue_seg(x_seg), W_seg ) )
// (see include/mcgen_hip.h, mcgen_conv_t).  For every chunk of 32 input channels the
// workgroup stages the tile's input WINDOW (tile rows + halo) into LDS once, applying the
// prologue (nearest-x2 upsample by index, BatchNorm scale/shift, ReLU, MultimodalController
// code multiply) on the way, and then runs all nine filter taps as shifted LDS reads of that
// window: each activation goes through the prologue once per chunk instead of once per tap.
// Weights arrive as a pre-built "image" [chunk][tap][cout][32] (mcgen_prep_weight), so a tap's
// B tile is one contiguous block.
//
// MFMA orientation: A operand = weights (rows = output channels), B operand = activations
// (columns = pixels), D[cout][pixel]: lane l holds 4 consecutive output channels of pixel l&15,
// which makes the LDS-staged epilogue a 16-byte write per fragment.
// bf16 uses v_mfma_f32_16x16x32_bf16; f32 uses 8 x v_mfma_f32_16x16x4_f32 over the same
// fragment (exact fp32 FMA chains) -- same LDS images, same epilogue.
#include "conv_tile.h"
#include <stdlib.h>
#include <type_traits>

namespace {

template <typename T, int BM, int BN, int WM, int WN>
struct ConvCfg {
    static constexpr int NT = 64 * WM * WN;
    static constexpr int FM = BM / WM / 16;        // pixel fragments per wave
    static constexpr int FN = BN / WN / 16;        // cout fragments per wave
    static constexpr int ESZ = Elem<T>::BYTES;
    static constexpr int APITCH = MCGEN_CK * ESZ + 16 * ESZ;   // bf16: 96 B (conflict-free b128 reads)
    static constexpr int BROW = MCGEN_CK * ESZ;                // bytes per weight row in LDS
    static constexpr int BBYTES = BN * BROW;                   // one tap's weight tile
    static constexpr int EP = BN + 4;                          // epilogue pitch in floats
    static constexpr int NI = (BM * 9 + NT - 1) / NT;          // staging items per thread: PP*4 <= BM*2.25*4
    static constexpr int UPR = BROW / 16;                      // 16-byte units per weight row
    static constexpr int UNITS = BN * UPR;
    static constexpr int NU = (UNITS + NT - 1) / NT;           // weight units per thread per tap
    static constexpr int EPX = (BN >= 128 && BM > 64) ? BM / 64 : 1;             // epilogue passes of >= 64 pixels
    static constexpr int PPX = BM / EPX;                       // pixels per epilogue pass
    static constexpr int CH = BN / 8;                          // 8-channel chunks per output pixel
    static constexpr int PROWS = NT / CH;                      // threads sharing one chunk
    static_assert(NT % CH == 0, "epilogue thread mapping");
    static_assert(BM % EPX == 0 && PPX % 16 == 0, "epilogue passes");
};

// compacted-output table behind the epilogue tile: int16 column per output slot, then fp32 bias per slot
constexpr int YTAB_COLS = 320, YTAB_BYTES = YTAB_COLS * 2 + YTAB_COLS * 4;

// Fast form of the shared epilogue below for a tile that is whole in every way: all BN channels inside Cout (Cout % 8 == 0),
// every image of the tile inside the batch, no compacted output, no tanh, the output code (if any) one row for the tile.
// The tile's output pixels are then CONTIGUOUS in y (tiles are whole image rows, or whole images): pixel mt of the tile is
// opix0 + mt.  What the general loop spends per output pixel -- 64-bit index arithmetic, per-channel range selects, unpacked
// fp32 math, a residual / gate load whose latency is exposed in every iteration -- goes: the pass's residual and gate rows
// are requested before the accumulators take their turn through LDS, the math runs on float2 (v_pk_fma_f32 / v_pk_add_f32),
// and nothing is predicated per channel.  Same arithmetic per element and same accumulation order of the statistics as the
// general loop: results are bit-identical with NP = 0.  Measured on the 256 x 256 pp tile (tools/pp_fc.py, 74 us per tile at 256 input channels):
// epilogue 12.1 -> 9.6 us, of which acc -> LDS 2.0, the read loop's LDS reads + math 4.0, its global stores 2.7; over the
// whole step (every conv kernel shares this epilogue) 10.10 -> 9.62 ms.
// MODE: what the launch fuses behind alpha / bias, decided once per workgroup -- with everything optional in one loop the
// compiler if-converts the cheap options into selects and keeps every option's per-channel vectors live (measured on the
// 256 x 256 tile: 3.5 us of the epilogue's 9.6; 9.62 -> 9.49 ms/step).  1 plain (the forward convolutions in front of a
// BatchNorm: statistics and the store only), 0 any combination.  (2 residual only and 3 gate only compile, and measured no
// faster than 0 on the step: not dispatched.)
// YC (with MODE 1): compacted output -- the loop accumulates the statistics only, a gather pass per round stores, per pixel,
// the channels of the image's map (the table sits behind the accumulator rows).
// YC == 2 (with MODE 1; mcgen_conv_t.yperm): the weight rows of the image's set were permuted at prep time, the tile's columns
// ARE the compacted order -- the plain loop with the store limited to the first Cy columns, bias and the statistics rows
// indexed through the permutation.
template <typename T, typename C, int BM, int BN, int WM, int WN, bool POOL, int NP = 0, int MODE = 0, int YC = 0>
__device__ __forceinline__ void conv_epilogue_fast(const mcgen_conv_t& p, const Geo& g, f32x4 (&acc)[C::FN][C::FM],
                                                   float* epi, int tid, int wm, int wn, int l15, int lg,
                                                   int tile_m, int cout0, float alpha) {
    using E = Elem<T>;
    constexpr int NT = C::NT, FM = C::FM, FN = C::FN, EP = C::EP, CH = C::CH, PROWS = C::PROWS;
    // NP > 0 (kernels whose LDS holds BM / NP pixel rows of fp32: the pp form): NP passes instead of EPX, and EVERY wave takes
    // part in every pass -- pass k holds fragments [k FMP, (k + 1) FMP) of each wave row (GRP = 16 FMP pixels, whole image
    // rows, pooling pairs inside).  Fewer write | barrier | read | barrier rounds: the 256 x 128 tile has one instead of four.
    constexpr bool REMAP = NP > 0;
    constexpr int NPASS = REMAP ? NP : C::EPX, PPX = BM / NPASS;
    constexpr int FMP = REMAP ? FM / NPASS : FM, GRP = FMP * 16, WROW = BM / WM;
    static_assert(!REMAP || (FM % NPASS == 0 && GRP % 64 == 0), "merged passes hold whole row pairs of every wave row");
    constexpr int OUT_PP = POOL ? PPX / 4 : PPX;                       // output pixels per pass
    constexpr int ITERS = (OUT_PP + PROWS - 1) / PROWS;
    constexpr bool RAGGED = OUT_PP % PROWS != 0;                       // (fewer output pixels per pass than thread rows)
    constexpr int PB = ITERS < 4 ? ITERS : 4;                          // residual / gate rows requested ahead, per batch
    const int ch = tid % CH, prow = tid / CH;
    const int co = cout0 + ch * 8;
    const int W = p.W, Cy = p.Cy;
    const int Ho = POOL ? (p.H >> 1) : p.H, Wo = POOL ? (W >> 1) : W;
    const size_t opix0 = ((size_t)g.n0 * Ho + (POOL ? (g.h0 >> 1) : g.h0)) * Wo;
    T* yb = reinterpret_cast<T*>(p.y) + opix0 * Cy + co;
    constexpr bool HAS_OC = MODE == 0, HAS_RES = MODE == 0 || MODE == 2, HAS_GATE = MODE == 0 || MODE == 3, PLAIN = MODE == 1;
    const T* rb = (HAS_RES && p.res) ? reinterpret_cast<const T*>(p.res) + opix0 * Cy + co : nullptr;
    const T* gb = (HAS_GATE && p.gate_x) ? reinterpret_cast<const T*>(p.gate_x) + opix0 * Cy + co : nullptr;
    const bool use_res = MODE == 2 || rb != nullptr, use_gate = MODE == 3 || gb != nullptr;
    const int stats_mode = p.stats_mode;

    auto ld4 = [&](const float* q, float dflt, f32x2 (&o)[4]) {
        if (q) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(q + co), b = *reinterpret_cast<const f32x4*>(q + co + 4);
            o[0] = f32x2{a[0], a[1]}; o[1] = f32x2{a[2], a[3]}; o[2] = f32x2{b[0], b[1]}; o[3] = f32x2{b[2], b[3]};
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = f32x2{dflt, dflt};
        }
    };
    f32x2 bias[4], oc[4], gsc[4], gsh[4], gme[4], grs[4];
    const int16_t* yperm = nullptr;
    if constexpr (YC == 2) {
        yperm = p.yperm + (size_t)p.wsel[g.n0] * p.yperm_stride;
        const u32x4 c4 = *reinterpret_cast<const u32x4*>(yperm + co);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int c0 = (int)(c4[i] & 0xffffu), c1 = (int)(c4[i] >> 16);
            bias[i] = p.bias ? f32x2{p.bias[c0], p.bias[c1]} : f32x2{0.f, 0.f};
            if (p.bias2) bias[i] += f32x2{p.bias2[c0], p.bias2[c1]};
        }
    } else {
    ld4(p.bias, 0.f, bias);
    }
    if (YC != 2 && p.bias2) {
        f32x2 b2[4];
        ld4(p.bias2, 0.f, b2);
#pragma unroll
        for (int i = 0; i < 4; ++i) bias[i] += b2[i];
    }
    ld4((HAS_OC && p.ocode) ? p.ocode + (size_t)g.n0 * p.Cout : nullptr, 1.f, oc);
    ld4(gb ? p.gscale : nullptr, 1.f, gsc);
    ld4(gb && p.gscale ? p.gshift : nullptr, 0.f, gsh);
    ld4(gb && stats_mode == 2 ? p.gmean : nullptr, 0.f, gme);
    ld4(gb && stats_mode == 2 ? p.grstd : nullptr, 0.f, grs);
    const f32x2 al = {alpha, alpha};
    f32x2 s1[4], s2[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { s1[i] = f32x2{0.f, 0.f}; s2[i] = f32x2{0.f, 0.f}; }
    const int lgWo = POOL ? g.lgW - 1 : g.lgW, lgTHWo = POOL ? g.lgTHW - 2 : g.lgTHW;

    static_assert(YC == 0 || (MODE == 1 && !POOL), "a compacted output takes bias and statistics only");
    int16_t* ycol = reinterpret_cast<int16_t*>(epi + PPX * EP);       // tile column (or -1) and bias of every output slot
    float* ybias = reinterpret_cast<float*>(ycol + YTAB_COLS);
    const int ycgrp = Cy >> 3;
    if constexpr (YC == 1) {
        const int16_t* cidx = p.ycmap + (size_t)g.n0 * p.ycmap_stride + ((p.Cout + 7) & ~7);   // record: [cpos: C][cidx: C + 32]...
        for (int j = tid; j < Cy; j += NT) {
            const int ct = (int)(uint16_t)cidx[j];                     // true channel (the zero row's index beyond the image's count)
            const int c = ct - cout0;
            const bool ok = c >= 0 && c < BN && ct < p.Cout;
            ycol[j] = (int16_t)(ok ? c : -1);
            ybias[j] = (ok && p.bias) ? p.bias[ct] : 0.f;
        }
    }
    // output pixel `mo` of pass `pass` -> pixel of the (pooled) tile
    auto tile_pix = [&](int pass, int mo) -> int {
        if constexpr (REMAP) {
            constexpr int G = POOL ? GRP / 4 : GRP, WR = POOL ? WROW / 4 : WROW;
            return (mo / G) * WR + pass * G + (mo % G);
        } else {
            return pass * OUT_PP + mo;
        }
    };
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        // residual / gate rows of the first PB iterations: in flight while the accumulators go through LDS
        typename E::vec8 rraw[PB], graw[PB];
        auto request = [&](int it0) {
#pragma unroll
            for (int j = 0; j < PB; ++j) {
                const int mo = prow + (it0 + j) * PROWS;
                if (it0 + j < ITERS && (!RAGGED || mo < OUT_PP)) {
                    const int off = tile_pix(pass, mo) * Cy;
                    if (use_res) rraw[j] = E::load8v(rb + off);
                    if (use_gate) graw[j] = E::load8v(gb + off);
                }
            }
        };
        if constexpr (!PLAIN) request(0);
        if (pass > 0) __syncthreads();                     // previous pass's reads of epi are done
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                const int cc = wn * (BN / WN) + fn * 16 + lg * 4;
                if constexpr (REMAP) {
                    if (fm / FMP == pass)
                        *reinterpret_cast<f32x4*>(epi + (wm * GRP + (fm - pass * FMP) * 16 + l15) * EP + cc) = acc[fn][fm];
                } else {
                    const int m0 = wm * (BM / WM) + fm * 16;
                    if (m0 / PPX == pass) *reinterpret_cast<f32x4*>(epi + (m0 - pass * PPX + l15) * EP + cc) = acc[fn][fm];
                }
            }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int mo = prow + it * PROWS;
            if (RAGGED && mo >= OUT_PP) continue;
            const int mt = tile_pix(pass, mo);
            f32x2 v[4];
            if constexpr (POOL) {
                int m00;
                if constexpr (REMAP) {
                    constexpr int G = GRP / 4;
                    const int gq = mo / G, within = mo % G;
                    const int ro = within >> lgWo, wo = within & ((1 << lgWo) - 1);
                    m00 = gq * GRP + ((2 * ro) << g.lgW) + 2 * wo;
                } else {
                    const int ti = mt >> lgTHWo, rem = mt & ((1 << lgTHWo) - 1);
                    const int ro = rem >> lgWo, wo = rem & ((1 << lgWo) - 1);
                    m00 = (ti << g.lgTHW) + ((2 * ro) << g.lgW) + 2 * wo - pass * PPX;
                }
                const float* e0 = epi + m00 * EP + ch * 8;
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(e0 + 4 * hh), b = *reinterpret_cast<const f32x4*>(e0 + EP + 4 * hh);
                    const f32x4 c = *reinterpret_cast<const f32x4*>(e0 + W * EP + 4 * hh), d = *reinterpret_cast<const f32x4*>(e0 + (W + 1) * EP + 4 * hh);
                    const f32x4 q = (a + b) + (c + d);
                    v[2 * hh] = f32x2{q[0], q[1]}; v[2 * hh + 1] = f32x2{q[2], q[3]};
                }
            } else {
                const float* e0 = epi + mo * EP + ch * 8;
                const f32x4 a = *reinterpret_cast<const f32x4*>(e0), b = *reinterpret_cast<const f32x4*>(e0 + 4);
                v[0] = f32x2{a[0], a[1]}; v[1] = f32x2{a[2], a[3]}; v[2] = f32x2{b[0], b[1]}; v[3] = f32x2{b[2], b[3]};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = __builtin_elementwise_fma(v[i], al, bias[i]);
            if (HAS_OC && p.ocode) {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] *= oc[i];
            }
            if (use_gate) {
                float xv[8];
                E::unpack8(graw[it % PB], xv);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const f32x2 x2 = {xv[2 * i], xv[2 * i + 1]};
                    const f32x2 z = __builtin_elementwise_fma(x2, gsc[i], gsh[i]);
                    v[i][0] = (z[0] > 0.f) ? v[i][0] : 0.f;
                    v[i][1] = (z[1] > 0.f) ? v[i][1] : 0.f;
                    if (stats_mode == 2) { s1[i] += v[i]; s2[i] = __builtin_elementwise_fma(v[i], (x2 - gme[i]) * grs[i], s2[i]); }
                }
            }
            if (use_res) {
                float rv[8];
                E::unpack8(rraw[it % PB], rv);
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] += f32x2{rv[2 * i], rv[2 * i + 1]};
            }
            if (stats_mode == 1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) { s1[i] += v[i]; s2[i] = __builtin_elementwise_fma(v[i], v[i], s2[i]); }
            }
            const float vo[8] = {v[0][0], v[0][1], v[1][0], v[1][1], v[2][0], v[2][1], v[3][0], v[3][1]};
            if constexpr (YC == 0) E::store8(yb + mt * Cy, vo);
            if constexpr (YC == 2) { if (ch * 8 < Cy) E::store8(yb + mt * Cy, vo); }
            if constexpr (!PLAIN) if ((it + 1) % PB == 0 && it + 1 < ITERS) request(it + 1);        // the next batch's rows
        }
        if constexpr (YC == 1) {
            // gather: output slot group jg of local row mo <- tile columns ycol[8 jg ..] (zeros beyond the image's count); reads
            // the accumulator rows only, like the loop above: no barrier between them
            T* y0 = reinterpret_cast<T*>(p.y) + opix0 * Cy;
            const float yrcp = 1.0f / (float)ycgrp;
            for (int u = tid; u < OUT_PP * ycgrp; u += NT) {
                const int mo = (int)(((float)u + 0.5f) * yrcp), jg = u - mo * ycgrp;       // u / ycgrp (exact: u < 2^16, ycgrp <= 40)
                const u32x4 c4 = *reinterpret_cast<const u32x4*>(ycol + jg * 8);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(ybias + jg * 8), b1 = *reinterpret_cast<const f32x4*>(ybias + jg * 8 + 4);
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = (int)(int16_t)((i & 1) ? (c4[i >> 1] >> 16) : (c4[i >> 1] & 0xffffu));
                    v[i] = c >= 0 ? fmaf(epi[mo * EP + c], alpha, i < 4 ? b0[i & 3] : b1[i & 3]) : 0.f;
                }
                E::store8(y0 + tile_pix(pass, mo) * Cy + jg * 8, v);
            }
        }
    }

    if (stats_mode != 0 && p.stats) {
        __syncthreads();                               // everyone is done reading epi
        f32x2* red = reinterpret_cast<f32x2*>(epi);    // [PROWS][BN] of (sum, sum of squares)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            red[prow * BN + ch * 8 + 2 * i] = f32x2{s1[i][0], s2[i][0]};
            red[prow * BN + ch * 8 + 2 * i + 1] = f32x2{s1[i][1], s2[i][1]};
        }
        __syncthreads();
        for (int c = tid; c < BN; c += NT) {
            float a = 0.f, b = 0.f;
            for (int r = 0; r < PROWS; ++r) { const f32x2 t = red[r * BN + c]; a += t[0]; b += t[1]; }
            const int spitch = YC ? p.Cout_w : Cy;                    // (compacted output: statistics over the true channels)
            const int ct = YC == 2 ? (int)yperm[c] : cout0 + c;
            p.stats[((size_t)tile_m * 2 + 0) * spitch + ct] = a;
            p.stats[((size_t)tile_m * 2 + 1) * spitch + ct] = b;
        }
    }
}

// Shared epilogue: accumulators -> LDS (fp32 [pixel][cout]) -> fused output pass, PPX pixels at a time.
template <typename T, typename C, int BM, int BN, int WM, int WN, int NP = 0>
__device__ __forceinline__ void conv_epilogue(const mcgen_conv_t& p, const Geo& g, f32x4 (&acc)[C::FN][C::FM],
                                              float* epi, int tid, int wm, int wn, int l15, int lg,
                                              int tile_m, int cout0, float alpha) {
    using E = Elem<T>;
    constexpr int NT = C::NT, FM = C::FM, FN = C::FN, EP = C::EP;
    const int H = p.H, W = p.W, N = p.N;
    constexpr int CH = C::CH, PROWS = C::PROWS, PPX = C::PPX;
    const int ch = tid % CH, prow = tid / CH;
    const int co = cout0 + ch * 8;             // first channel of this thread's chunk
    const int Ho = p.pool ? (H >> 1) : H, Wo = p.pool ? (W >> 1) : W;
    // compacted output (p.ycmap): y holds, per image, only the channels its consumer's MultimodalController keeps, in
    // compacted order (pitch Cy); the statistics still cover every true channel (pitch Cout_w)
    const int spitch = p.ycmap ? p.Cout_w : p.Cy;
    const bool chunk_live = co < spitch;
    // (workgroup-uniform) the whole tile is inside the output: the fast form above
    if constexpr (sizeof(T) == 2) {
        // compacted output of a whole tile (validated: bf16, one image, every channel in this tile, bias and statistics only)
        if (p.ycmap && cout0 == 0 && BN == p.Cout && g.n0 < N && p.Cy <= YTAB_COLS) {
            conv_epilogue_fast<T, C, BM, BN, WM, WN, false, NP, 1, 1>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0, alpha);
            return;
        }
        if constexpr (NP > 0) {
            // weight rows permuted per set (validated: bf16 pp form, one image per tile, every channel in this tile, bias and statistics only)
            if (p.yperm) {
                conv_epilogue_fast<T, C, BM, BN, WM, WN, false, NP, 1, 2>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0, alpha);
                return;
            }
        }
    }
    if (!p.ycmap && !p.tanh_out && (p.Cout & 7) == 0 && cout0 + BN <= p.Cout && g.n0 + g.TI <= N && (!p.ocode || g.TI == 1)) {
#define MCGEN_EPI_CASE(POOLV, MODEV) conv_epilogue_fast<T, C, BM, BN, WM, WN, POOLV, NP, MODEV>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0, alpha)
        if (p.pool) MCGEN_EPI_CASE(true, 0);
        else if (!p.ocode && !p.gate_x && !p.res) MCGEN_EPI_CASE(false, 1);
        else MCGEN_EPI_CASE(false, 0);
#undef MCGEN_EPI_CASE
        return;
    }
    T* y = reinterpret_cast<T*>(p.y);
    const T* res = reinterpret_cast<const T*>(p.res);
    const T* gx = reinterpret_cast<const T*>(p.gate_x);

    // per-thread channel vectors (this thread always handles the same 8 channels)
    const bool vec_ok = (p.Cout % 8 == 0) && (co + 8 <= p.Cout);
    float bias[8], gsc[8], gsh[8], gme[8], grs[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool ok = (co + i) < p.Cout;
        bias[i] = (p.bias && ok) ? p.bias[co + i] : 0.f;
        if (p.bias2 && ok) bias[i] += p.bias2[co + i];
        gsc[i] = (p.gscale && ok) ? p.gscale[co + i] : 1.f;
        gsh[i] = (p.gscale && ok) ? p.gshift[co + i] : 0.f;
        gme[i] = (p.gmean && ok) ? p.gmean[co + i] : 0.f;
        grs[i] = (p.grstd && ok) ? p.grstd[co + i] : 0.f;
    }
    float s1[8], s2[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
    // output code of a tile that lies inside one image: one row for every pixel of the tile, loaded once
    const bool oc_once = p.ocode && vec_ok && g.TI == 1 && g.n0 < N && chunk_live;
    float oc1[8];
    if (oc_once) load8f(p.ocode + (size_t)g.n0 * p.Cout + co, oc1);
    const int lgWo = p.pool ? g.lgW - 1 : g.lgW;
    const int lgTHWo = p.pool ? g.lgTHW - 2 : g.lgTHW;
    const int out_pp = p.pool ? (PPX >> 2) : PPX;          // output pixels per pass
    // compacted output: tile column (or -1) and bias of every output slot of this tile's image, kept in LDS behind the
    // accumulator tile (the launch reserves YTAB_BYTES for it); visible after the first barrier of pass 0
    int16_t* ycol = reinterpret_cast<int16_t*>(epi + C::PPX * EP);
    float* ybias = reinterpret_cast<float*>(ycol + YTAB_COLS);
    const bool ylive = p.ycmap && g.n0 < N;
    const int ycgrp = p.Cy >> 3;
    const unsigned yinv = 65536u / (unsigned)ycgrp + 1u;          // u / ycgrp == (u * yinv) >> 16 for u < 4096
    if (ylive) {
        const int16_t* cidx = p.ycmap + (size_t)g.n0 * p.ycmap_stride + ((p.Cout + 7) & ~7);   // record: [cpos: C][cidx: C + 32]...
        for (int j = tid; j < p.Cy; j += NT) {
            const int ct = (int)(uint16_t)cidx[j];                 // true channel (the zero row's index beyond the image's count)
            const int c = ct - cout0;
            const bool ok = c >= 0 && c < BN && ct < p.Cout;
            ycol[j] = (int16_t)(ok ? c : -1);
            ybias[j] = (ok && p.bias) ? p.bias[ct] : 0.f;
        }
    }
#pragma unroll
    for (int pass = 0; pass < C::EPX; ++pass) {
        if (pass > 0) __syncthreads();                     // previous pass's reads of epi are done
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                const int m0 = wm * (BM / WM) + fm * 16;
                if (m0 / PPX == pass) {
                    const int m = m0 - pass * PPX + l15;
                    const int cc = wn * (BN / WN) + fn * 16 + lg * 4;
                    *reinterpret_cast<f32x4*>(epi + m * EP + cc) = acc[fn][fm];
                }
            }
        __syncthreads();
        for (int mo = prow; mo < out_pp; mo += PROWS) {
            // output pixel mo of this pass -> (ti, ro, wo) inside the tile
            const int mt = pass * out_pp + mo;
            const int ti = mt >> lgTHWo, rem = mt & ((1 << lgTHWo) - 1);
            const int ro = rem >> lgWo, wo = rem & ((1 << lgWo) - 1);
            const int n = g.n0 + ti;
            if (n >= N || !chunk_live) continue;
            float v[8];
            if (p.pool) {
                const int m00 = (ti << g.lgTHW) + ((2 * ro) << g.lgW) + 2 * wo - pass * PPX;
                const float* e0 = epi + m00 * EP + ch * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (e0[i] + e0[EP + i]) + (e0[W * EP + i] + e0[(W + 1) * EP + i]);
            } else {
                const float* e0 = epi + mo * EP + ch * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = e0[i];
            }
            const int ho = (p.pool ? (g.h0 >> 1) : g.h0) + ro;
            const size_t opix = ((size_t)n * Ho + ho) * Wo + wo;
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], alpha, bias[i]);
            if (p.ocode) {
                if (oc_once) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= oc1[i];
                } else if (vec_ok) {
                    float oc[8];
                    load8f(p.ocode + (size_t)n * p.Cout + co, oc);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= oc[i];
                } else {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= ((co + i) < p.Cout) ? p.ocode[(size_t)n * p.Cout + co + i] : 0.f;
                }
            }
            if (gx) {
                float xv[8];
                E::load8(gx + opix * p.Cy + co, xv);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float z = fmaf(xv[i], gsc[i], gsh[i]);
                    v[i] = (z > 0.f) ? v[i] : 0.f;
                }
                if (p.stats_mode == 2) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) { s1[i] += v[i]; s2[i] = fmaf(v[i], (xv[i] - gme[i]) * grs[i], s2[i]); }
                }
            }
            if (res) {
                float rv[8];
                E::load8(res + opix * p.Cy + co, rv);
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] += rv[i];
            }
            if (p.tanh_out) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = tanhf(v[i]);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) if ((co + i) >= p.Cout) v[i] = 0.f;
            if (p.stats_mode == 1) {
#pragma unroll
                for (int i = 0; i < 8; ++i) { s1[i] += v[i]; s2[i] = fmaf(v[i], v[i], s2[i]); }
            }
            if (!p.ycmap) E::store8(y + opix * p.Cy + co, v);          // (compacted output: the gather pass below stores)
        }
        if (ylive) {
            // gather pass: output slot j of a pixel <- tile column ycol[j] (zeros beyond the image's count).  A compacted
            // output carries bias only (validated), so the value is rebuilt from the raw accumulator tile: both loops of
            // this pass only READ the tile -- no hand-back, no barrier between them.
            for (int u = tid; u < out_pp * ycgrp; u += NT) {
                const int mo = (int)(((unsigned)u * yinv) >> 16), jg = u - mo * ycgrp;
                const int mt = pass * out_pp + mo;
                const int ro = mt >> lgWo, wo = mt & ((1 << lgWo) - 1);
                const size_t opix = ((size_t)g.n0 * Ho + (g.h0 + ro)) * Wo + wo;
                const u32x4 c4 = *reinterpret_cast<const u32x4*>(ycol + jg * 8);
                const f32x4 b0 = *reinterpret_cast<const f32x4*>(ybias + jg * 8), b1 = *reinterpret_cast<const f32x4*>(ybias + jg * 8 + 4);
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = (int)(int16_t)((i & 1) ? (c4[i >> 1] >> 16) : (c4[i >> 1] & 0xffffu));
                    v[i] = c >= 0 ? fmaf(epi[mo * EP + c], alpha, i < 4 ? b0[i & 3] : b1[i & 3]) : 0.f;
                }
                E::store8(y + opix * p.Cy + jg * 8, v);
            }
        }
    }

    if (p.stats_mode != 0 && p.stats) {
        __syncthreads();                               // everyone is done reading epi
        float* red = epi;                              // [PROWS][BN][2]
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            red[(prow * BN + ch * 8 + i) * 2 + 0] = s1[i];
            red[(prow * BN + ch * 8 + i) * 2 + 1] = s2[i];
        }
        __syncthreads();
        for (int c = tid; c < BN; c += NT) {
            float a = 0.f, b = 0.f;
            for (int r = 0; r < PROWS; ++r) { a += red[(r * BN + c) * 2]; b += red[(r * BN + c) * 2 + 1]; }
            if (cout0 + c < spitch) {
                p.stats[((size_t)tile_m * 2 + 0) * spitch + cout0 + c] = a;
                p.stats[((size_t)tile_m * 2 + 1) * spitch + cout0 + c] = b;
            }
        }
    }
}

// Simple form (the fp32 parity build): one window buffer, one weight buffer staged through registers, two barriers per
// tap; latency is hidden by running several workgroups per CU (small LDS / register footprint).
template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN)
void conv_fused_kernel(const mcgen_conv_t p, const int a_bytes) {
    using C = ConvCfg<T, BM, BN, WM, WN>;
    using M = Mma<T>;
    constexpr int NT = C::NT, FM = C::FM, FN = C::FN, ESZ = C::ESZ, APITCH = C::APITCH, BROW = C::BROW, EP = C::EP;
    constexpr int UPR = C::UPR, NU = C::NU;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsA0 = smem;
    char* const ldsB0 = smem + a_bytes;
    float* epi = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lg = lane >> 4;

    const int H = p.H, W = p.W, N = p.N;
    const int tile_m = blockIdx.x;
    const int cout0 = blockIdx.y * BN;
    const Geo g = make_geo(BM, blockIdx.x, H, W);

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- weight-tile staging ---------------------------------------------------------------------
    const char* wimg = reinterpret_cast<const char*>(p.w);
    const size_t wblock_bytes = (size_t)p.Cout_w * BROW;
    int b_goff[NU], b_loff[NU];                    // per-thread global / LDS byte offsets inside a tile
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        const int u = tid + k * NT;
        b_goff[k] = -1; b_loff[k] = -1;
        if (u < C::UNITS) {
            const int row = u / UPR, gu = u % UPR;
            const int grp = gu / (ESZ / 2), within = gu % (ESZ / 2);          // 8-channel group
            const int sw = grp ^ (3 * ((row >> 3) & 1));
            b_loff[k] = row * BROW + (sw * (ESZ / 2) + within) * 16;
            if (cout0 + row < p.Cout_w) b_goff[k] = (cout0 + row) * BROW + gu * 16;
        }
    }
    u32x4 breg[NU];
    auto B_load = [&](int blk) {
        const char* wb = wimg + (size_t)blk * wblock_bytes;
#pragma unroll
        for (int k = 0; k < NU; ++k) {
            breg[k] = u32x4{0u, 0u, 0u, 0u};
            if (b_goff[k] >= 0) breg[k] = *reinterpret_cast<const u32x4*>(wb + b_goff[k]);
        }
    };
    auto B_write = [&](char* dst) {
#pragma unroll
        for (int k = 0; k < NU; ++k)
            if (b_loff[k] >= 0) *reinterpret_cast<u32x4*>(dst + b_loff[k]) = breg[k];
    };
    int w_row_off[FN];                             // per-lane LDS offset of each weight fragment
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
        const int row = wn * (BN / WN) + fn * 16 + l15;
        w_row_off[fn] = row * BROW + (lg ^ (3 * ((row >> 3) & 1))) * 8 * ESZ;
    }

    // simple form: one window buffer, one weight buffer, two barriers per tap; latency is hidden by
    // running several workgroups per CU (small LDS / register footprint)
    int blk = 0;
    for (int s = 0; s < p.nseg; ++s) {
        const mcgen_seg_t sg = seg_for_tile(p.seg[s], g);
        const int halo = sg.ksize >> 1;
        const int PR = g.TH + 2 * halo, PC = W + 2 * halo;
        PatchStager<T, NT, C::NI, APITCH> stager;
        stager.setup(sg, g, N, H, W, tid);
        int a_base[FM];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int m = wm * (BM / WM) + fm * 16 + l15;
            const int ti = m >> g.lgTHW, rem = m & ((1 << g.lgTHW) - 1);
            const int r = rem >> g.lgW, c = rem & (W - 1);
            a_base[fm] = ((ti * PR + r) * PC + c) * APITCH + lg * 8 * ESZ;
        }
        const int nchunk = (sg.C + MCGEN_CK - 1) / MCGEN_CK;
        const int ntap = sg.ksize * sg.ksize;
#pragma unroll 1
        for (int q = 0; q < nchunk; ++q) {
            __syncthreads();                                  // previous chunk's MFMA reads are done
            stager.stage(sg, q * MCGEN_CK, ldsA0);
#pragma unroll 1
            for (int tap = 0; tap < ntap; ++tap) {
                if (tap > 0) __syncthreads();                 // previous tap's weight reads are done
                B_load(blk);
                B_write(ldsB0);
                __syncthreads();
                const int kh = (sg.ksize == 3) ? tap / 3 : 0, kw = (sg.ksize == 3) ? tap % 3 : 0;
                const int tapoff = (kh * PC + kw) * APITCH;
                typename M::frag af[FM], wf[FN];
#pragma unroll
                for (int fm = 0; fm < FM; ++fm)
                    af[fm] = *reinterpret_cast<const typename M::frag*>(ldsA0 + a_base[fm] + tapoff);
#pragma unroll
                for (int fn = 0; fn < FN; ++fn)
                    wf[fn] = *reinterpret_cast<const typename M::frag*>(ldsB0 + w_row_off[fn]);
#pragma unroll
                for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                    for (int fm = 0; fm < FM; ++fm) M::run(wf[fn], af[fm], acc[fn][fm]);
                ++blk;
            }
        }
    }
    __syncthreads();

    // ---- epilogue ------------------------------------------------------------------------------
    conv_epilogue<T, C, BM, BN, WM, WN>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0, p.alpha);
}

// ---- "dma1" form: one tap per barrier, three single-tap slots ------------------------------------------------------
// The weight ring holds three single taps (48 KB at 256 output channels instead of dma3's 96 KB), two taps ahead of
// their use, one raw s_barrier per tap behind a counted vmcnt.  With a 128-pixel x 256-channel tile on FOUR waves
// (one per SIMD, each 128 pixels x 64 channels: the same per-wave MFMA / fragment-read mix as the 256 x 256 tile) a
// workgroup needs 68 KB of LDS, so TWO workgroups share a CU: one's window staging, barrier waits and epilogue run
// under the other's MFMAs -- the overlap the one-workgroup-per-CU 256 x 256 tile cannot have.
template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4) ? 2 : 0)      // four-wave tiles: two workgroups (= two waves per SIMD) per CU
void conv_dma1_kernel(const mcgen_conv_t p, const int a_bytes) {
    using C = ConvCfg<T, BM, BN, WM, WN>;
    using M = Mma<T>;
    constexpr int NT = C::NT, FM = C::FM, FN = C::FN, ESZ = C::ESZ, APITCH = C::APITCH, BROW = C::BROW;
    constexpr int RB = 3, DIST = 2;                        // ring slots, prefetch distance (taps)
    constexpr int NW = WM * WN;
    constexpr int KB = C::BBYTES / 1024;                   // 1 KB DMA pieces per weight tile
    constexpr int PPW = (KB + NW - 1) / NW;                // pieces per wave per tap
    static_assert(DIST == 2 && PPW <= 31, "counted vmcnt immediate");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsA = smem;
    char* const ldsB0 = smem + a_bytes;
    float* epi = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lg = lane >> 4;
    const int H = p.H, W = p.W, N = p.N;
    const int tile_m = blockIdx.x;
    const int cout0 = blockIdx.y * BN;
    const Geo g = make_geo(BM, blockIdx.x, H, W);

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const char* wimg = reinterpret_cast<const char*>(p.w);
    const size_t wblock_bytes = (size_t)p.Cout_w * BROW;
    int total_steps = 0;
    for (int s = 0; s < p.nseg; ++s)
        total_steps += ((p.seg[s].C + MCGEN_CK - 1) / MCGEN_CK) * p.seg[s].ksize * p.seg[s].ksize;

    // DMA piece k of this wave: 1 KB = rows [16*ESZ/2 rows...]; lane -> (row, physical 16-byte unit)
    constexpr int UPR = C::UPR;                            // 16-byte units per row (4 bf16 / 8 fp32)
    constexpr int RPP = 64 / UPR;                          // rows per 1 KB piece
    int d_src[PPW];                                        // per-lane source byte offset inside a weight block, -1 = zero rows
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
        const int piece = wave * PPW + k;
        const int row = piece * RPP + lane / UPR, pu = lane % UPR;
        const int grp = pu / (ESZ / 2), within = pu % (ESZ / 2);
        const int lgrp = grp ^ (3 * ((row >> 3) & 1));     // logical 8-channel group stored at this physical slot
        d_src[k] = (piece < KB && cout0 + row < p.Cout_w) ? (cout0 + row) * BROW + (lgrp * (ESZ / 2) + within) * 16 : -1;
    }
    auto B_dma = [&](int blk) {
        if (blk >= total_steps) return;
        const char* wb = wimg + (size_t)blk * wblock_bytes;
        char* slot = ldsB0 + (blk % RB) * C::BBYTES;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int piece = wave * PPW + k;
            if (piece < KB) {
                // rows beyond Cout_w read row 0 of the block (in bounds); their outputs are never stored
                const char* src = wb + (d_src[k] >= 0 ? d_src[k] : (lane % UPR) * 16);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(slot + piece * 1024), 16, 0, 0);
            }
        }
    };
    int w_row_off[FN];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
        const int row = wn * (BN / WN) + fn * 16 + l15;
        w_row_off[fn] = row * BROW + (lg ^ (3 * ((row >> 3) & 1))) * 8 * ESZ;
    }

    int blk = 0;
    B_dma(0);
    B_dma(1);
    for (int s = 0; s < p.nseg; ++s) {
        const mcgen_seg_t sg = seg_for_tile(p.seg[s], g);
        const int halo = sg.ksize >> 1;
        const int PR = g.TH + 2 * halo, PC = W + 2 * halo;
        PatchStager<T, NT, C::NI, APITCH> stager;
        stager.setup(sg, g, N, H, W, tid);
        int a_base[FM];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int m = wm * (BM / WM) + fm * 16 + l15;
            const int ti = m >> g.lgTHW, rem = m & ((1 << g.lgTHW) - 1);
            const int r = rem >> g.lgW, c = rem & (W - 1);
            a_base[fm] = ((ti * PR + r) * PC + c) * APITCH + lg * 8 * ESZ;
        }
        const int nchunk = (sg.C + MCGEN_CK - 1) / MCGEN_CK;
        const int ntap = sg.ksize * sg.ksize;
#pragma unroll 1
        for (int q = 0; q < nchunk; ++q) {
            // window of this chunk: everyone is past the previous chunk's reads (barrier), then publish
            __builtin_amdgcn_s_barrier();
            stager.stage(sg, q * MCGEN_CK, ldsA);
#pragma unroll 1
            for (int tap = 0; tap < ntap; ++tap) {
                // this tap's weight tile has landed (this wave's pieces), then all waves' pieces + window writes
                if (blk + 1 < total_steps) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                B_dma(blk + DIST);                       // slot (blk+2)%3 == (blk-1)%3: its readers passed the barrier
                const int kh = (sg.ksize == 3) ? tap / 3 : 0, kw = (sg.ksize == 3) ? tap % 3 : 0;
                const int tapoff = (kh * PC + kw) * APITCH;
                const char* ldsB = ldsB0 + (blk % RB) * C::BBYTES;
                typename M::frag af[FM], wf[FN];
#pragma unroll
                for (int fm = 0; fm < FM; ++fm)
                    af[fm] = *reinterpret_cast<const typename M::frag*>(ldsA + a_base[fm] + tapoff);
#pragma unroll
                for (int fn = 0; fn < FN; ++fn)
                    wf[fn] = *reinterpret_cast<const typename M::frag*>(ldsB + w_row_off[fn]);
#pragma unroll
                for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                    for (int fm = 0; fm < FM; ++fm) M::run(wf[fn], af[fm], acc[fn][fm]);
                ++blk;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    conv_epilogue<T, C, BM, BN, WM, WN>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0, p.alpha);
}


// ---- "dma3" form ------------------------------------------------------------------------------------------
// The weight tiles move by LDS-DMA (global_load_lds, no VGPR round trip); the DMA writes LDS linearly (wave base +
// lane*16), so the bank swizzle is applied to the SOURCE address.  THREE taps per barrier: a ring slot holds the weight tiles of a group of up to 3 taps (one kernel row of a 3x3 filter); the
// group after the current one is in flight (2 slots).  Every group issues the same number of DMA
// instructions (short groups re-load their last tap), so the counted vmcnt is a compile-time constant.
// (measured: asking for a 256-register budget on the 256x256 tile -- __launch_bounds__(512, 2) -- removes its 24 bytes of
// scratch but runs 4-9 % slower; the default budget stays)
#ifndef MCGEN_STAGE2
#define MCGEN_STAGE2 1
#endif
#ifndef MCGEN_STAGE2_BIG
#define MCGEN_STAGE2_BIG 0
#endif
#ifndef MCGEN_STAGE2_SMALL
#define MCGEN_STAGE2_SMALL 0
#endif
#ifndef MCGEN_BIG_WAVES
#define MCGEN_BIG_WAVES 0
#endif
template <typename T, int BM, int BN, int WM, int WN>
// (the 64 x 64 tile of 8 waves: two workgroups per CU = 4 waves per SIMD, i.e. at most 128 registers -- the allocator sits at
//  110 .. 131 depending on unrelated code; at 131 the 8x8 input-gradient launches of the generator ran 25 -> 44 us)
__global__ __launch_bounds__(64 * WM * WN, (WM * WN == 4 && BM * BN >= 256 * 128) ? 2 : ((BM * BN <= 64 * 64 && WM * WN == 8) ? 4 : ((MCGEN_BIG_WAVES && BM * BN >= 256 * 256) ? MCGEN_BIG_WAVES : 0)))
void conv_dma3_kernel(const mcgen_conv_t p, const int a_bytes) {
    using C = ConvCfg<T, BM, BN, WM, WN>;
    using M = Mma<T>;
    constexpr int NT = C::NT, FM = C::FM, FN = C::FN, ESZ = C::ESZ, APITCH = C::APITCH, BROW = C::BROW;
    // taps per group: a kernel row; on the skinny Cout <= 16 tile the whole 3x3 filter of a chunk (9 KB).  That tile does 18
    // MFMAs per wave and chunk, so with a row per group every group waited out the round trip of a DMA issued one (tiny)
    // group earlier -- three exposed L2 latencies per chunk; with the chunk as the group the next chunk's taps are in flight
    // behind a whole chunk's staging and MFMAs
    constexpr int TPS = (BN <= 16) ? 9 : 3;
    // 128-pixel tiles and the skinny Cout <= 16 tile (8 accumulators; item by item it exposed three round trips per chunk:
    // 231 -> 184 us on the generator's 640-image head with this and the whole-chunk tap group below); the 256x256 tile has no
    // registers to spare (128 accumulators), 64-pixel tiles have 2 items per thread
    constexpr bool STAGE2 = MCGEN_STAGE2 && (BM == 128 || BN <= 16 || (MCGEN_STAGE2_SMALL && BM < 128) || (MCGEN_STAGE2_BIG && BM * BN >= 256 * 256));
    constexpr int NW = WM * WN;
    constexpr int KB = C::BBYTES / 1024;                   // 1 KB DMA pieces per weight tile
    constexpr int PPW = (KB + NW - 1) / NW;                // pieces per wave per tap
    constexpr int SLOT = TPS * C::BBYTES;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsA = smem;
    char* const ldsB0 = smem + a_bytes;
    float* epi = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lg = lane >> 4;
    const int H = p.H, W = p.W, N = p.N;
    const int tile_m = blockIdx.x;
    const int cout0 = blockIdx.y * BN;
    const Geo g = make_geo(BM, blockIdx.x, H, W);

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const char* wimg = reinterpret_cast<const char*>(p.w);
    const size_t wblock_bytes = (size_t)p.Cout_w * BROW;
    // group bookkeeping over the linear (segment, chunk, tap) sequence
    const int nt0 = p.seg[0].ksize * p.seg[0].ksize, nc0 = (p.seg[0].C + MCGEN_CK - 1) / MCGEN_CK;
    const int gpc0 = (nt0 == 9) ? 9 / TPS : 1;                   // groups per chunk
    const int G0 = nc0 * gpc0, S0 = nc0 * nt0;
    int nt1 = 1, nc1 = 0, gpc1 = 1;
    if (p.nseg > 1) { nt1 = p.seg[1].ksize * p.seg[1].ksize; nc1 = (p.seg[1].C + MCGEN_CK - 1) / MCGEN_CK; gpc1 = (nt1 == 9) ? 9 / TPS : 1; }
    const int GT = G0 + nc1 * gpc1;                        // total groups

    constexpr int UPR = C::UPR, RPP = 64 / UPR;
    int d_src[PPW];
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
        const int piece = wave * PPW + k;
        const int row = piece * RPP + lane / UPR, pu = lane % UPR;
        const int grp = pu / (ESZ / 2), within = pu % (ESZ / 2);
        const int lgrp = grp ^ (3 * ((row >> 3) & 1));
        d_src[k] = (piece < KB && cout0 + row < p.Cout_w) ? (cout0 + row) * BROW + (lgrp * (ESZ / 2) + within) * 16 : -1;
    }
    auto G_dma = [&](int gi) {                             // all taps of group gi -> slot gi & 1
        if (gi >= GT) return;
        int blk0, ntg;
        if (gi < G0) { ntg = (nt0 == 9) ? TPS : 1; blk0 = gi * ntg; }
        else { const int gj = gi - G0; ntg = (nt1 == 9) ? TPS : 1; blk0 = S0 + gj * ntg; }
        char* slot = ldsB0 + (gi & 1) * SLOT;
#pragma unroll
        for (int t = 0; t < TPS; ++t) {
            if (TPS > 3 && t >= ntg) break;                        // (the waits below are vmcnt(0): no constant count needed)
            const int blk = blk0 + (t < ntg ? t : ntg - 1);        // short group: re-load the last tap (constant DMA count)
            const char* wb = wimg + (size_t)blk * wblock_bytes;
#pragma unroll
            for (int k = 0; k < PPW; ++k) {
                const int piece = wave * PPW + k;
                if (piece < KB) {
                    const char* src = wb + (d_src[k] >= 0 ? d_src[k] : (lane % UPR) * 16);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(slot + t * C::BBYTES + piece * 1024), 16, 0, 0);
                }
            }
        }
    };
    int w_row_off[FN];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
        const int row = wn * (BN / WN) + fn * 16 + l15;
        w_row_off[fn] = row * BROW + (lg ^ (3 * ((row >> 3) & 1))) * 8 * ESZ;
    }

    int gi = 0;
    G_dma(0);
    for (int s = 0; s < p.nseg; ++s) {
        const mcgen_seg_t sg = seg_for_tile(p.seg[s], g);
        const int halo = sg.ksize >> 1;
        const int PR = g.TH + 2 * halo, PC = W + 2 * halo;
        PatchStager<T, NT, C::NI, APITCH> stager;
        stager.setup(sg, g, N, H, W, tid);
        int a_base[FM];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int m = wm * (BM / WM) + fm * 16 + l15;
            const int ti = m >> g.lgTHW, rem = m & ((1 << g.lgTHW) - 1);
            const int r = rem >> g.lgW, c = rem & (W - 1);
            a_base[fm] = ((ti * PR + r) * PC + c) * APITCH + lg * 8 * ESZ;
        }
        const int nchunk = (sg.C + MCGEN_CK - 1) / MCGEN_CK;
        const int ntap = sg.ksize * sg.ksize;
        const int gpc = (ntap == 9) ? 9 / TPS : 1, ntg = (ntap == 9) ? TPS : 1;
#pragma unroll 1
        for (int q = 0; q < nchunk; ++q) {
            __builtin_amdgcn_s_barrier();                  // everyone is past the previous chunk's window reads
            if constexpr (STAGE2) {
                // all global loads of the chunk's window first, then prologue + LDS stores: one exposed round trip per
                // chunk instead of one per item (the item-sequential form keeps fewer registers live)
                typename PatchStager<T, NT, C::NI, APITCH>::raw_t raw;
                stager.load(sg, q * MCGEN_CK, raw);
                stager.write(sg, q * MCGEN_CK, raw, ldsA, (g.TI == 1 && g.n0 < N) ? g.n0 : -1);
            } else {
                stager.stage(sg, q * MCGEN_CK, ldsA);
            }
#pragma unroll 1
            for (int gq = 0; gq < gpc; ++gq) {
                // this group's tiles have landed (this wave's pieces); then all waves' pieces + window writes
                if (gi + 1 < GT) { /* the next group is NOT yet issued here: nothing newer in flight */ }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                G_dma(gi + 1);                            // slot (gi+1)&1: its readers (group gi-1) passed the barrier
                const char* slot = ldsB0 + (gi & 1) * SLOT;
#pragma unroll
                for (int t = 0; t < TPS; ++t) {
                    if (t < ntg) {
                        const int tap = gq * ntg + t;
                        const int kh = (ntap == 9) ? tap / 3 : 0, kw = (ntap == 9) ? tap % 3 : 0;
                        const int tapoff = (kh * PC + kw) * APITCH;
                        const char* ldsB = slot + t * C::BBYTES;
                        typename M::frag af[FM], wf[FN];
#pragma unroll
                        for (int fm = 0; fm < FM; ++fm)
                            af[fm] = *reinterpret_cast<const typename M::frag*>(ldsA + a_base[fm] + tapoff);
#pragma unroll
                        for (int fn = 0; fn < FN; ++fn)
                            wf[fn] = *reinterpret_cast<const typename M::frag*>(ldsB + w_row_off[fn]);
#pragma unroll
                        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                            for (int fm = 0; fm < FM; ++fm) M::run(wf[fn], af[fm], acc[fn][fm]);
                    }
                }
                ++gi;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    conv_epilogue<T, C, BM, BN, WM, WN>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0, p.alpha);
}

// ---- "dma3g" form: dma3 for PURE 1x1 launches, three chunks per barrier round ------------------------------------
// A 1x1 convolution has one tap per 32-channel chunk, so dma3 spends a barrier round (window staging + wait) per MFMA
// step; here the windows of three consecutive chunks are staged side by side and their three weight tiles (which are
// consecutive in the image) form one DMA group.  A separate kernel: sharing dma3's code cost the 3x3 launches 1-3 %.
template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN, (BM * BN <= 64 * 64 && WM * WN == 8) ? 4 : 0)
void conv_dma3g_kernel(const mcgen_conv_t p, const int a_bytes) {
    constexpr bool G3 = true;
    constexpr int g1 = 3;
    using C = ConvCfg<T, BM, BN, WM, WN>;
    using M = Mma<T>;
    constexpr int NT = C::NT, FM = C::FM, FN = C::FN, ESZ = C::ESZ, APITCH = C::APITCH, BROW = C::BROW;
    constexpr int TPS = 3;
    constexpr int NW = WM * WN;
    constexpr int KB = C::BBYTES / 1024;                   // 1 KB DMA pieces per weight tile
    constexpr int PPW = (KB + NW - 1) / NW;                // pieces per wave per tap
    constexpr int SLOT = TPS * C::BBYTES;
    constexpr int SUBW = BM * APITCH;                      // one chunk's window of a 1x1 segment (no halo)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsA = smem;
    char* const ldsB0 = smem + a_bytes;
    float* epi = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lg = lane >> 4;
    const int H = p.H, W = p.W, N = p.N;
    const int tile_m = blockIdx.x;
    const int cout0 = blockIdx.y * BN;
    const Geo g = make_geo(BM, blockIdx.x, H, W);

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const char* wimg = reinterpret_cast<const char*>(p.w);
    const size_t wblock_bytes = (size_t)p.Cout_w * BROW;
    // group bookkeeping over the linear (segment, chunk, tap) sequence
    const int nt0 = p.seg[0].ksize * p.seg[0].ksize, nc0 = (p.seg[0].C + MCGEN_CK - 1) / MCGEN_CK;
    // a group = 3 taps of one chunk (3x3) or g1 consecutive chunks of a 1x1 segment (their weight tiles are consecutive)
    const int G0 = (nt0 == 9) ? nc0 * 3 : (nc0 + g1 - 1) / g1, S0 = nc0 * nt0;
    int nt1 = 1, nc1 = 0;
    if (p.nseg > 1) { nt1 = p.seg[1].ksize * p.seg[1].ksize; nc1 = (p.seg[1].C + MCGEN_CK - 1) / MCGEN_CK; }
    const int GT = G0 + ((p.nseg > 1) ? ((nt1 == 9) ? nc1 * 3 : (nc1 + g1 - 1) / g1) : 0);   // total groups

    constexpr int UPR = C::UPR, RPP = 64 / UPR;
    int d_src[PPW];
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
        const int piece = wave * PPW + k;
        const int row = piece * RPP + lane / UPR, pu = lane % UPR;
        const int grp = pu / (ESZ / 2), within = pu % (ESZ / 2);
        const int lgrp = grp ^ (3 * ((row >> 3) & 1));
        d_src[k] = (piece < KB && cout0 + row < p.Cout_w) ? (cout0 + row) * BROW + (lgrp * (ESZ / 2) + within) * 16 : -1;
    }
    auto G_dma = [&](int gi) {                             // all taps of group gi -> slot gi & 1
        if (gi >= GT) return;
        int blk0, ntg;
        if (gi < G0) {
            if (nt0 == 9) { ntg = 3; blk0 = gi * 3; }
            else { blk0 = gi * g1; ntg = (nc0 - blk0) < g1 ? (nc0 - blk0) : g1; }
        } else {
            const int gj = gi - G0;
            if (nt1 == 9) { ntg = 3; blk0 = S0 + gj * 3; }
            else { const int c0 = gj * g1; ntg = (nc1 - c0) < g1 ? (nc1 - c0) : g1; blk0 = S0 + c0; }
        }
        char* slot = ldsB0 + (gi & 1) * SLOT;
#pragma unroll
        for (int t = 0; t < TPS; ++t) {
            const int blk = blk0 + (t < ntg ? t : ntg - 1);        // short group: re-load the last tap (constant DMA count)
            const char* wb = wimg + (size_t)blk * wblock_bytes;
#pragma unroll
            for (int k = 0; k < PPW; ++k) {
                const int piece = wave * PPW + k;
                if (piece < KB) {
                    const char* src = wb + (d_src[k] >= 0 ? d_src[k] : (lane % UPR) * 16);
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(slot + t * C::BBYTES + piece * 1024), 16, 0, 0);
                }
            }
        }
    };
    int w_row_off[FN];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
        const int row = wn * (BN / WN) + fn * 16 + l15;
        w_row_off[fn] = row * BROW + (lg ^ (3 * ((row >> 3) & 1))) * 8 * ESZ;
    }

    int gi = 0;
    G_dma(0);
    for (int s = 0; s < p.nseg; ++s) {
        const mcgen_seg_t sg = seg_for_tile(p.seg[s], g);
        const int halo = sg.ksize >> 1;
        const int PR = g.TH + 2 * halo, PC = W + 2 * halo;
        PatchStager<T, NT, C::NI, APITCH> stager;
        stager.setup(sg, g, N, H, W, tid);
        int a_base[FM];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int m = wm * (BM / WM) + fm * 16 + l15;
            const int ti = m >> g.lgTHW, rem = m & ((1 << g.lgTHW) - 1);
            const int r = rem >> g.lgW, c = rem & (W - 1);
            a_base[fm] = ((ti * PR + r) * PC + c) * APITCH + lg * 8 * ESZ;
        }
        const int nchunk = (sg.C + MCGEN_CK - 1) / MCGEN_CK;
        const int ntap = sg.ksize * sg.ksize;
        // 3x3: one chunk per round, 3 groups of 3 taps; 1x1: g1 chunks per round (windows side by side), one group
        const int gpc = (ntap == 9) ? 3 : 1, qs = (ntap == 9) ? 1 : g1;
#pragma unroll 1
        for (int q = 0; q < nchunk; q += qs) {
            const int ntg = (ntap == 9) ? 3 : ((nchunk - q) < g1 ? (nchunk - q) : g1);
            __builtin_amdgcn_s_barrier();                  // everyone is past the previous round's window reads
            stager.stage(sg, q * MCGEN_CK, ldsA);
            if constexpr (G3) {
                if (qs > 1 && q + 1 < nchunk) stager.stage(sg, (q + 1) * MCGEN_CK, ldsA + SUBW);
                if (qs > 1 && q + 2 < nchunk) stager.stage(sg, (q + 2) * MCGEN_CK, ldsA + 2 * SUBW);
            }
#pragma unroll 1
            for (int gq = 0; gq < gpc; ++gq) {
                // this group's tiles have landed (this wave's pieces); then all waves' pieces + window writes
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                G_dma(gi + 1);                            // slot (gi+1)&1: its readers (group gi-1) passed the barrier
                const char* slot = ldsB0 + (gi & 1) * SLOT;
#pragma unroll
                for (int t = 0; t < TPS; ++t) {
                    if (t < ntg) {
                        const int tap = gq * 3 + t;
                        const int tapoff = (ntap == 9) ? ((tap / 3) * PC + (tap % 3)) * APITCH : t * SUBW;
                        const char* ldsB = slot + t * C::BBYTES;
                        typename M::frag af[FM], wf[FN];
#pragma unroll
                        for (int fm = 0; fm < FM; ++fm)
                            af[fm] = *reinterpret_cast<const typename M::frag*>(ldsA + a_base[fm] + tapoff);
#pragma unroll
                        for (int fn = 0; fn < FN; ++fn)
                            wf[fn] = *reinterpret_cast<const typename M::frag*>(ldsB + w_row_off[fn]);
#pragma unroll
                        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                            for (int fm = 0; fm < FM; ++fm) M::run(wf[fn], af[fm], acc[fn][fm]);
                    }
                }
                ++gi;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    conv_epilogue<T, C, BM, BN, WM, WN>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0, p.alpha);
}

// ---- "cp" form (chunk-pipelined), for the tiles of small maps ---------------------------------------------------
// On 8x8 / 16x16 maps (and for skinny-N convolutions) a workgroup's MFMA work per 32-channel chunk is a fraction of a
// microsecond, so the forms above are a serial chain of exposed round trips: stage the window, wait, DMA a tap group,
// wait, ...  Here one STEP = all tap units of a chunk (the 9 taps of a 3x3 chunk, or up to 8 chunks of a 1x1 segment):
// while step k runs its MFMAs, step k+1's weight tiles are already streaming into the other half of an LDS ring by
// LDS-DMA and its input window is in flight to registers.  Per step: one s_waitcnt vmcnt(0), the prologue + LDS
// store of the window from registers, two barriers, no wait inside the tap loop.
template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN)
void conv_cp_kernel(const mcgen_conv_t p, const int a_bytes, const int subw) {
    using C = ConvCfg<T, BM, BN, WM, WN>;
    using M = Mma<T>;
    constexpr int NT = C::NT, FM = C::FM, FN = C::FN, ESZ = C::ESZ, APITCH = C::APITCH, BROW = C::BROW;
    constexpr int NW = WM * WN;
    constexpr int KB = (C::BBYTES + 1023) / 1024;            // 1 KB DMA pieces per weight tile
    constexpr int UMAX = 9, NQ1 = 8;                          // tap units per step; chunks per step of a 1x1 segment
    constexpr int HALF = UMAX * C::BBYTES;
    constexpr int NI3 = C::NI, NI1 = (BM * 4 + NT - 1) / NT;
    constexpr int DPW = (UMAX * KB + NW - 1) / NW;            // DMA instructions per wave per step (upper bound)

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsA = smem;
    char* const ldsB0 = smem + a_bytes;
    float* epi = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lg = lane >> 4;
    const int H = p.H, W = p.W, N = p.N;
    const int tile_m = blockIdx.x;
    const int cout0 = blockIdx.y * BN;
    const Geo g = make_geo(BM, blockIdx.x, H, W);

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const char* wimg = reinterpret_cast<const char*>(p.w);
    const size_t wblock_bytes = (size_t)p.Cout_w * BROW;
    constexpr int UPR = C::UPR, RPP = 64 / UPR;
    // weight tiles blk0 .. blk0+U-1 -> ring half `half`; (unit, piece) pairs are dealt round-robin to the waves
    auto dma_issue = [&](int blk0, int U, int half) {
        char* base = ldsB0 + half * HALF;
#pragma unroll
        for (int i = 0; i < DPW; ++i) {
            const int up = wave + i * NW;
            if (up < U * KB) {
                const int u = up / KB, kb = up % KB;
                const int row = kb * RPP + lane / UPR, pu = lane % UPR;
                const int grp = pu / (ESZ / 2), within = pu % (ESZ / 2);
                const int lgrp = grp ^ (3 * ((row >> 3) & 1));
                const bool ok = (row < BN) && (cout0 + row < p.Cout_w);
                const char* src = wimg + (size_t)(blk0 + u) * wblock_bytes +
                                  (ok ? (cout0 + row) * BROW + (lgrp * (ESZ / 2) + within) * 16 : pu * 16);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(base + u * C::BBYTES + kb * 1024), 16, 0, 0);
            }
        }
    };
    int w_row_off[FN];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
        const int row = wn * (BN / WN) + fn * 16 + l15;
        w_row_off[fn] = row * BROW + (lg ^ (3 * ((row >> 3) & 1))) * 8 * ESZ;
    }

    int half = 0, blk_seg = 0;                                 // ring half of the step being consumed; first tile of the segment
    for (int s = 0; s < p.nseg; ++s) {
        const mcgen_seg_t sg = seg_for_tile(p.seg[s], g);
        const int halo = sg.ksize >> 1;
        const int PR = g.TH + 2 * halo, PC = W + 2 * halo;
        const int nchunk = (sg.C + MCGEN_CK - 1) / MCGEN_CK;
        const int ntap = sg.ksize * sg.ksize;
        int a_base[FM];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int m = wm * (BM / WM) + fm * 16 + l15;
            const int ti = m >> g.lgTHW, rem = m & ((1 << g.lgTHW) - 1);
            const int r = rem >> g.lgW, c = rem & (W - 1);
            a_base[fm] = ((ti * PR + r) * PC + c) * APITCH + lg * 8 * ESZ;
        }
        auto mma_unit = [&](const char* ldsAu, const char* ldsBu) {
            typename M::frag af[FM], wf[FN];
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) af[fm] = *reinterpret_cast<const typename M::frag*>(ldsAu + a_base[fm]);
#pragma unroll
            for (int fn = 0; fn < FN; ++fn) wf[fn] = *reinterpret_cast<const typename M::frag*>(ldsBu + w_row_off[fn]);
#pragma unroll
            for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) M::run(wf[fn], af[fm], acc[fn][fm]);
        };
        if (ntap == 9) {
            PatchStager<T, NT, NI3, APITCH> st;
            st.setup(sg, g, N, H, W, tid);
            typename PatchStager<T, NT, NI3, APITCH>::raw_t raw;
            dma_issue(blk_seg, 9, half);
            st.load(sg, 0, raw);
#pragma unroll 1
            for (int q = 0; q < nchunk; ++q) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                  // the previous step's readers are done with the window
                st.write(sg, q * MCGEN_CK, raw, ldsA);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();                  // window + every wave's weight pieces are visible
                if (q + 1 < nchunk) {
                    dma_issue(blk_seg + (q + 1) * 9, 9, half ^ 1);
                    st.load(sg, (q + 1) * MCGEN_CK, raw);
                }
                const char* ldsB = ldsB0 + half * HALF;
#pragma unroll
                for (int u = 0; u < 9; ++u)
                    mma_unit(ldsA + ((u / 3) * PC + (u % 3)) * APITCH, ldsB + u * C::BBYTES);
                half ^= 1;
            }
        } else {
            PatchStager<T, NT, NI1, APITCH> st;
            st.setup(sg, g, N, H, W, tid);
            typename PatchStager<T, NT, NI1, APITCH>::raw_t raw[NQ1];
            auto load_step = [&](int q0, int nq) {
#pragma unroll
                for (int j = 0; j < NQ1; ++j)
                    if (j < nq) st.load(sg, (q0 + j) * MCGEN_CK, raw[j]);
            };
            int nq = nchunk < NQ1 ? nchunk : NQ1;
            dma_issue(blk_seg, nq, half);
            load_step(0, nq);
#pragma unroll 1
            for (int q0 = 0; q0 < nchunk; q0 += NQ1) {
                nq = (nchunk - q0) < NQ1 ? (nchunk - q0) : NQ1;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int j = 0; j < NQ1; ++j)
                    if (j < nq) st.write(sg, (q0 + j) * MCGEN_CK, raw[j], ldsA + j * subw);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                const int qn = q0 + NQ1;
                if (qn < nchunk) {
                    const int nn = (nchunk - qn) < NQ1 ? (nchunk - qn) : NQ1;
                    dma_issue(blk_seg + qn, nn, half ^ 1);
                    load_step(qn, nn);
                }
                const char* ldsB = ldsB0 + half * HALF;
#pragma unroll
                for (int u = 0; u < NQ1; ++u)
                    if (u < nq) mma_unit(ldsA + u * subw, ldsB + u * C::BBYTES);
                half ^= 1;
            }
        }
        blk_seg += nchunk * ntap;
        __builtin_amdgcn_s_barrier();                          // segment change: the window is re-staged with another geometry
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    conv_epilogue<T, C, BM, BN, WM, WN>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0, p.alpha);
}

// ---- "mc" form: mode-compacted K loop ---------------------------------------------------------------------------
// MultimodalController (modules.py:71-76) multiplies every conv input by a per-sample 0/1 code: with controller_rate
// 0.5 half of the input channels of a sample are exact zeros.  A tile that lies inside one image therefore needs only
// the K slices of that sample's ACTIVE channels.  This form visits them alone:
//   * activations: the window of a dense 32-channel chunk is loaded and run through the prologue as in the dma3 form, but
//     each value is written to the LDS window at its COMPACTED position (mcgen_mc_cmap: cpos), into a ring of two
//     32-slot halves; an MFMA K step (9 taps) fires whenever a half is full -- on average after two dense chunks;
//   * weights: the image is K-major ([tap][k][cout], mcgen_prep_weight_k), so a compacted K step is a gather of 32
//     ROWS: each 1 KB LDS-DMA piece takes its rows' indices from the map (cidx, scalar loads).  In LDS a tap tile is
//     [32 k][BN cout] with the 32-byte cout slots XOR-swizzled by (k & 7); the MFMA A fragments come out of it through
//     ds_read_b64_tr_b16 (conflict-free: the 8 rows of a half-wave read fall on 8 distinct bank groups);
//   * the k order inside a 32-slot step is permuted (slot s = 8*lg + 4*a + b holds logical k = 16*a + 4*lg + b) so that
//     one transposing read covers 8 consecutive rows; the activation scatter applies the same permutation.
// Padded slots of the last step point at the image's zero row; the window is zeroed once so they multiply finite data.
// FLOP accounting stays dense (roofline fractions are quoted on 2*N*H*W*Cout*K with K = all channels).
// wave-uniform dword through the scalar cache (constant address space): no vmcnt entry, so it neither waits for nor
// drains the LDS-DMAs in flight
static __device__ __forceinline__ uint32_t mc_sload(const void* p) {
    return *reinterpret_cast<const __attribute__((address_space(4))) uint32_t*>(reinterpret_cast<uintptr_t>(p));
}
static __device__ __forceinline__ s16x4 mc_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(reinterpret_cast<uintptr_t>(p)));
}

template <int BM, int BN, int WM, int WN, bool MC_PREFETCH>
__global__ __launch_bounds__(64 * WM * WN)
void conv_mc_kernel(const mcgen_conv_t p, const int a_bytes) {
    using T = bf16_t;
    using C = ConvCfg<T, BM, BN, WM, WN>;
    constexpr int NT = C::NT, FM = C::FM, FN = C::FN, APITCH = C::APITCH;
    constexpr int NW = WM * WN;
    constexpr int ROWB = BN * 2;                           // bytes per k row of a tap tile
    constexpr int RPP = 1024 / ROWB;                       // k rows per 1 KB DMA piece (2 or 4)
    constexpr int LPR = 64 / RPP;                          // lanes (16-byte units) per row
    constexpr int NPIECE = 32 / RPP;                       // pieces per tap tile
    constexpr int PPW = (NPIECE + NW - 1) / NW;            // pieces per wave per tap
    constexpr int TAPB = 32 * ROWB;                        // bytes per tap tile
    constexpr int TPS = 3, SLOT = TPS * TAPB;
    constexpr int NI = C::NI;
    static_assert(BN == 128 || BN == 256, "k-major tap tiles: 2 or 4 rows per DMA piece");
    static_assert(NPIECE % NW == 0 || NPIECE < NW, "pieces are dealt evenly to the waves");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsA = smem;                               // two window halves of a_bytes each
    char* const ldsB0 = smem + 2 * a_bytes;                // two weight slots
    float* epi = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lg = lane >> 4;
    const int H = p.H, W = p.W, N = p.N;
    const int tile_m = blockIdx.x;
    const int cout0 = blockIdx.y * BN;
    const Geo g = make_geo(BM, blockIdx.x, H, W);          // host guarantees TI == 1: the tile lies inside image g.n0
    const int n_img = g.n0 < N ? g.n0 : N - 1;

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // zero both window halves once (padded slots / first use): 16 bytes per thread per step
    for (int o = tid * 16; o < 2 * a_bytes; o += NT * 16) *reinterpret_cast<u32x4*>(ldsA + o) = u32x4{0u, 0u, 0u, 0u};

    // ---- per-segment bookkeeping (wave-uniform) -------------------------------------------------------------
    const char* wimg = reinterpret_cast<const char*>(p.w);
    const int16_t* rec0 = p.seg[0].cmap + (size_t)n_img * p.seg[0].cmap_stride;
    const int C0 = p.seg[0].C, nd0 = (C0 + 31) >> 5;
    const int cnt0 = (int)mc_sload(rec0 + 2 * C0 + 32 + 2 * nd0);                 // cpre[nd]: the sample's active channels
    const int nt0 = p.seg[0].ksize * p.seg[0].ksize, gpc0 = (nt0 == 9) ? 3 : 1;
    const int nks0 = (cnt0 + 31) >> 5;
    const size_t tapstride0 = (size_t)(C0 + 1) * p.Cout_w * 2;                     // bytes per tap of segment 0's image
    const size_t seg1_off = (size_t)nt0 * tapstride0;
    int C1 = 8, nd1 = 0, cnt1 = 0, nt1 = 1, gpc1 = 1, nks1 = 0;
    const int16_t* rec1 = rec0;
    size_t tapstride1 = 0;
    if (p.nseg > 1) {
        rec1 = p.seg[1].cmap + (size_t)n_img * p.seg[1].cmap_stride;
        C1 = p.seg[1].C; nd1 = (C1 + 31) >> 5;
        cnt1 = (int)mc_sload(rec1 + 2 * C1 + 32 + 2 * nd1);
        nt1 = p.seg[1].ksize * p.seg[1].ksize; gpc1 = (nt1 == 9) ? 3 : 1;
        nks1 = (cnt1 + 31) >> 5;
        tapstride1 = (size_t)(C1 + 1) * p.Cout_w * 2;
    }
    const int G0 = nks0 * gpc0, GT = G0 + nks1 * gpc1;     // DMA groups of segment 0 / of the launch

    // ---- weight DMA: group gi -> slot gi & 1 ----------------------------------------------------------------
    // lane -> (row inside the piece, 16-byte unit inside the row); the unit's 32-byte slot is XOR-swizzled by the row
    const int rsub = lane / LPR, ci = lane % LPR;
    auto G_dma = [&](int gi) {
        if (gi >= GT) return;
        const bool s1 = gi >= G0;
        const int gl = s1 ? gi - G0 : gi;
        const int gpc = s1 ? gpc1 : gpc0;
        const int t = gl / gpc, gq = gl - t * gpc;                                  // K step, tap group of the step
        const int ntg = (gpc == 3) ? 3 : 1;
        const int16_t* cidx = (s1 ? rec1 + C1 : rec0 + C0) + 32 * t;
        const size_t tapstride = s1 ? tapstride1 : tapstride0;
        const char* wseg = wimg + (s1 ? seg1_off : 0);
        char* slot = ldsB0 + (gi & 1) * SLOT;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int piece = wave * PPW + k;
            if (piece < NPIECE) {
                // the RPP row indices of this piece: consecutive int16 entries, read as dwords (wave-uniform address)
                const uint32_t d01 = mc_sload(cidx + RPP * piece);
                uint32_t sel = d01;
                if constexpr (RPP == 4) {
                    const uint32_t d23 = mc_sload(cidx + RPP * piece + 2);
                    sel = (rsub & 2) ? d23 : d01;
                }
                const int dense = (int)((rsub & 1) ? (sel >> 16) : (sel & 0xffffu));   // dense channel = image row
                const int kk = RPP * piece + rsub;                                      // logical k inside the step
                int co = cout0 + 16 * ((ci >> 1) ^ (kk & 7)) + 8 * (ci & 1);
                if (co > p.Cout_w - 8) co = p.Cout_w - 8;                               // partial N tile: stay in bounds
                const size_t roff = ((size_t)dense * p.Cout_w + co) * 2;
#pragma unroll
                for (int tt = 0; tt < TPS; ++tt) {
                    const int tap = gq * ntg + (tt < ntg ? tt : ntg - 1);               // short group: re-load the last tap
                    const char* src = wseg + (size_t)tap * tapstride + roff;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(slot + tt * TAPB + piece * 1024), 16, 0, 0);
                }
            }
        }
    };
    // A-fragment (weights) read offsets: rows 4*lg + q (+16 for the second half), 32-byte slot (c0/16) ^ (row & 7)
    int w_off[FN];
    {
        const int q = l15 >> 2, pq = l15 & 3, r0 = 4 * lg + q;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            const int c0 = wn * (BN / WN) + fn * 16;
            w_off[fn] = r0 * ROWB + 32 * ((c0 >> 4) ^ (r0 & 7)) + 8 * pq;
        }
    }

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // the zero fill is complete before any scatter write
    int gi = 0;
    G_dma(0);
    for (int s = 0; s < p.nseg; ++s) {
        const mcgen_seg_t sg = seg_for_tile(p.seg[s], g);
        const int16_t* rec = s ? rec1 : rec0;
        const int Cs = sg.C, nd = (Cs + 31) >> 5, cnt = s ? cnt1 : cnt0;
        const int16_t* cpre = rec + 2 * Cs + 32;          // int32 entries (two int16 slots each)
        const int halo = sg.ksize >> 1;
        const int PC = W + 2 * halo;
        PatchStager<T, NT, NI, APITCH> stager;
        stager.setup(sg, g, N, H, W, tid);
        int a_base[FM];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int m = wm * (BM / WM) + fm * 16 + l15;
            const int rem = m & ((1 << g.lgTHW) - 1);
            const int r = rem >> g.lgW, c = rem & (W - 1);
            a_base[fm] = (r * PC + c) * APITCH + lg * 16;
        }
        const int ntap = sg.ksize * sg.ksize;
        const int gpc = (ntap == 9) ? 3 : 1, ntg = (ntap == 9) ? 3 : 1;
        int consumed = 0;                                  // compacted slots the MFMA steps have used (multiple of 32)
        int filled = 0;
        int filled_next = (int)mc_sload(cpre + 2);       // cpre[1], fetched one chunk ahead of its use
        // The window loads of the NEXT dense chunk are issued inside the current K step (behind its first weight DMA), so
        // their round trip runs under the step's MFMAs; `have_raw` says the registers already hold chunk dq's window.
        typename PatchStager<T, NT, NI, APITCH>::raw_t raw;
        u32x4 cp4 = u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
        bool have_raw = false;
        const int csub = stager.it_sub[0];
        auto fetch = [&](int cq) {                         // window + compaction positions of dense chunk cq
            stager.load(sg, cq * MCGEN_CK, raw);
            cp4 = u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
            if (cq * MCGEN_CK + csub < Cs) cp4 = *reinterpret_cast<const u32x4*>(rec + cq * MCGEN_CK + csub);
        };
#pragma unroll 1
        for (int dq = 0; dq < nd && cnt > 0; ++dq) {
            const int c0 = dq * MCGEN_CK;
            const int filled0 = filled;
            filled = filled_next;
            if (dq + 2 <= nd) filled_next = (int)mc_sload(cpre + 2 * (dq + 2));
            const bool last = (dq == nd - 1);
            const bool next_live = !last && filled_next > filled;          // chunk dq + 1 has active channels
            if (filled > filled0) {
                __builtin_amdgcn_s_barrier();              // everyone is past the window reads of the previous K step
                // ---- stage dense chunk dq: loads, prologue, scatter to the compacted positions ---------------------
                if (!have_raw) fetch(dq);
                have_raw = false;
                const int c = c0 + csub;
                const bool cok = c < Cs;
                float sc[8], sh[8], cd[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { sc[i] = 1.f; sh[i] = 0.f; cd[i] = 1.f; }
                if (sg.scale && cok) { load8f(sg.scale + c, sc); load8f(sg.shift + c, sh); }
                if (sg.code && cok) load8f(sg.code + (size_t)n_img * Cs + c, cd);
                int soff[8];                               // byte offset of each channel's slot inside a window pixel, -1: inactive
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int kabs = (int)(int16_t)((i & 1) ? (cp4[i >> 1] >> 16) : (cp4[i >> 1] & 0xffffu));
                    const int kk = kabs & 31;
                    const int slot = 8 * ((kk >> 2) & 3) + 4 * (kk >> 4) + (kk & 3);
                    soff[i] = (kabs < 0) ? -1 : (((kabs >> 5) & 1) * a_bytes + slot * 2);
                }
#pragma unroll
                for (int k = 0; k < NI; ++k) {
                    if (stager.it_lds[k] < 0) continue;
                    float v[8];
                    PatchStager<T, NT, NI, APITCH>::unpack(raw[k], v);
                    const bool inside = stager.it_src[k] >= 0 && cok;
                    char* px = ldsA + (stager.it_lds[k] - csub * 2);                  // the window pixel's first slot
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        float a = 0.f;
                        if (inside) {
                            a = sg.scale ? fmaf(v[i], sc[i], sh[i]) : v[i];
                            if (sg.relu) a = fmaxf(a, 0.f);
                            if (sg.code) a *= cd[i];
                        }
                        if (soff[i] >= 0) *reinterpret_cast<T*>(px + soff[i]) = (T)a;
                    }
                }
            }
            // ---- MFMA K steps on every half that is now complete (the last chunk flushes the partial one) ---------------
#pragma unroll 1
            while (filled - consumed >= 32 || (last && consumed < cnt)) {
                const char* win = ldsA + ((consumed >> 5) & 1) * a_bytes;
#pragma unroll 1
                for (int gq = 0; gq < gpc; ++gq) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this group's weight pieces (and nothing newer) landed
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // this wave's window writes are in LDS
                    __builtin_amdgcn_s_barrier();
                    G_dma(gi + 1);                        // slot (gi+1)&1: its readers (group gi-1) passed the barrier
                    if (MC_PREFETCH && gq == 0 && next_live && !have_raw) { fetch(dq + 1); have_raw = true; }
                    const char* slot = ldsB0 + (gi & 1) * SLOT;
#pragma unroll
                    for (int t = 0; t < TPS; ++t) {
                        if (t < ntg) {
                            const int tap = gq * ntg + t;
                            const int kh = (ntap == 9) ? tap / 3 : 0, kw = (ntap == 9) ? tap % 3 : 0;
                            const int tapoff = (kh * PC + kw) * APITCH;
                            const char* ldsB = slot + t * TAPB;
                            bf16x8 af[FM], wf[FN];
#pragma unroll
                            for (int fm = 0; fm < FM; ++fm)
                                af[fm] = *reinterpret_cast<const bf16x8*>(win + a_base[fm] + tapoff);
#pragma unroll
                            for (int fn = 0; fn < FN; ++fn) {
                                union { bf16x8 v; s16x4 h[2]; } u;
                                u.h[0] = mc_tr16(ldsB + w_off[fn]);
                                u.h[1] = mc_tr16(ldsB + w_off[fn] + 16 * ROWB);
                                wf[fn] = u.v;
                            }
#pragma unroll
                            for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                                for (int fm = 0; fm < FM; ++fm)
                                    acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[fn], af[fm], acc[fn][fm], 0, 0, 0);
                        }
                    }
                    ++gi;
                }
                consumed += 32;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    conv_epilogue<T, C, BM, BN, WM, WN>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0, p.alpha);
}

// ---- "gk" form: gathered-K convolution over COMPACTED activations ---------------------------------------------------
// Forward-only passes (the grouped generator pass that feeds the discriminator updates, train_gan.py:145-146) never
// read a masked activation again, so the producing convolution stores, per image, only the channels its consumer's
// MultimodalController keeps (mcgen_conv_t.ycmap: compacted order, pitch Cy ~ 5/8 of the channels).  This kernel is
// the consumer: the window is staged exactly as in the dma3 form -- contiguous channels, 16-byte LDS stores, the
// BatchNorm affine and the code folded into per-image scale / shift rows (mcgen_mc_affine, group_n = 1) -- and only
// the WEIGHTS are gathered: the K-major image's rows cidx[32 t + k] of the tile's image, as in the "mc" form.  A dense
// segment (no map: the block input of the first compacted block) takes rows 32 t + k.  K steps per segment:
// ceil(active channels / 32), about 5 of 8.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN)
void conv_gk_kernel(const mcgen_conv_t p, const int a_bytes) {
    using T = bf16_t;
    using C = ConvCfg<T, BM, BN, WM, WN>;
    constexpr int NT = C::NT, FM = C::FM, FN = C::FN, APITCH = C::APITCH;
    constexpr int NW = WM * WN;
    constexpr int ROWB = BN * 2, RPP = 1024 / ROWB, LPR = 64 / RPP, NPIECE = 32 / RPP;
    constexpr int PPW = (NPIECE + NW - 1) / NW;
    constexpr int TAPB = 32 * ROWB, TPS = 3, SLOT = TPS * TAPB;
    constexpr int NI = C::NI;
    static_assert(BN == 128 || BN == 256, "k-major tap tiles: 2 or 4 rows per DMA piece");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsA = smem;
    char* const ldsB0 = smem + a_bytes;
    float* epi = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lg = lane >> 4;
    const int H = p.H, W = p.W, N = p.N;
    const int tile_m = blockIdx.x;
    const int cout0 = blockIdx.y * BN;
    const Geo g = make_geo(BM, blockIdx.x, H, W);          // host guarantees TI == 1
    const int n_img = g.n0 < N ? g.n0 : N - 1;

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- per-segment bookkeeping (wave-uniform): active count, K steps, weight rows -----------------------------
    const char* wimg = reinterpret_cast<const char*>(p.w);
    auto seg_info = [&](int s, const int16_t*& cidx, int& cw, int& cnt) {
        const mcgen_seg_t& sg = p.seg[s];
        cw = sg.Cw > 0 ? sg.Cw : sg.C;
        cidx = nullptr; cnt = sg.C;
        if (sg.cmap) {
            const int16_t* rec = sg.cmap + (size_t)n_img * sg.cmap_stride;
            cidx = rec + cw;
            cnt = (int)mc_sload(rec + 2 * cw + 32 + 2 * ((cw + 31) >> 5));
            if (cnt > sg.C) cnt = sg.C;                    // (the host sized the compacted pitch to hold every sample)
        }
    };
    const int16_t* cidx0; int cw0, cnt0;
    seg_info(0, cidx0, cw0, cnt0);
    const int nt0 = p.seg[0].ksize * p.seg[0].ksize, gpc0 = (nt0 == 9) ? 3 : 1, nks0 = (cnt0 + 31) >> 5;
    const size_t tapstride0 = (size_t)(cw0 + 1) * p.Cout_w * 2, seg1_off = (size_t)nt0 * tapstride0;
    const int16_t* cidx1 = nullptr; int cw1 = 8, cnt1 = 0, nt1 = 1, gpc1 = 1, nks1 = 0;
    size_t tapstride1 = 0;
    if (p.nseg > 1) {
        seg_info(1, cidx1, cw1, cnt1);
        nt1 = p.seg[1].ksize * p.seg[1].ksize; gpc1 = (nt1 == 9) ? 3 : 1; nks1 = (cnt1 + 31) >> 5;
        tapstride1 = (size_t)(cw1 + 1) * p.Cout_w * 2;
    }
    const int G0 = nks0 * gpc0, GT = G0 + nks1 * gpc1;

    // ---- weight DMA: LDS row rho of a tap tile holds k = krow(rho), so that the two transposing reads of lane group lg
    // (rows 4 lg + q and 16 + 4 lg + q) deliver k = 8 lg + 0..3 and 8 lg + 4..7: the activations' natural channel order
    const int rsub = lane / LPR, ci = lane % LPR;
    auto G_dma = [&](int gi) {
        if (gi >= GT) return;
        const bool s1 = gi >= G0;
        const int gl = s1 ? gi - G0 : gi;
        const int gpc = s1 ? gpc1 : gpc0;
        const int t = gl / gpc, gq = gl - t * gpc;
        const int ntg = (gpc == 3) ? 3 : 1;
        const int16_t* cidx = s1 ? cidx1 : cidx0;
        const int cw = s1 ? cw1 : cw0;
        const size_t tapstride = s1 ? tapstride1 : tapstride0;
        const char* wseg = wimg + (s1 ? seg1_off : 0);
        char* slot = ldsB0 + (gi & 1) * SLOT;
#pragma unroll
        for (int k = 0; k < PPW; ++k) {
            const int piece = wave * PPW + k;
            if (piece < NPIECE) {
                const int rho0 = RPP * piece;                                           // first LDS row of the piece
                const int k0 = (rho0 < 16) ? 8 * (rho0 >> 2) + (rho0 & 3) : 8 * ((rho0 - 16) >> 2) + 4 + ((rho0 - 16) & 3);
                int dense;
                if (cidx) {
                    const uint32_t d01 = mc_sload(cidx + 32 * t + k0);
                    uint32_t sel = d01;
                    if constexpr (RPP == 4) {
                        const uint32_t d23 = mc_sload(cidx + 32 * t + k0 + 2);
                        sel = (rsub & 2) ? d23 : d01;
                    }
                    dense = (int)((rsub & 1) ? (sel >> 16) : (sel & 0xffffu));
                } else {
                    dense = 32 * t + k0 + rsub;
                    if (dense > cw) dense = cw;                                         // beyond the channels: the zero row
                }
                const int rho = rho0 + rsub;
                int co = cout0 + 16 * ((ci >> 1) ^ (rho & 7)) + 8 * (ci & 1);
                if (co > p.Cout_w - 8) co = p.Cout_w - 8;
                const size_t roff = ((size_t)dense * p.Cout_w + co) * 2;
#pragma unroll
                for (int tt = 0; tt < TPS; ++tt) {
                    const int tap = gq * ntg + (tt < ntg ? tt : ntg - 1);
                    const char* src = wseg + (size_t)tap * tapstride + roff;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(slot + tt * TAPB + piece * 1024), 16, 0, 0);
                }
            }
        }
    };
    int w_off[FN];
    {
        const int q = l15 >> 2, pq = l15 & 3, r0 = 4 * lg + q;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            const int c0 = wn * (BN / WN) + fn * 16;
            w_off[fn] = r0 * ROWB + 32 * ((c0 >> 4) ^ (r0 & 7)) + 8 * pq;
        }
    }

    int gi = 0;
    G_dma(0);
    for (int s = 0; s < p.nseg; ++s) {
        const mcgen_seg_t sg = seg_for_tile(p.seg[s], g);
        const int nks = s ? nks1 : nks0;
        const int halo = sg.ksize >> 1;
        const int PC = W + 2 * halo;
        PatchStager<T, NT, NI, APITCH> stager;
        stager.setup(sg, g, N, H, W, tid);
        int a_base[FM];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int m = wm * (BM / WM) + fm * 16 + l15;
            const int rem = m & ((1 << g.lgTHW) - 1);
            const int r = rem >> g.lgW, c = rem & (W - 1);
            a_base[fm] = (r * PC + c) * APITCH + lg * 16;
        }
        const int ntap = sg.ksize * sg.ksize;
        const int gpc = (ntap == 9) ? 3 : 1, ntg = (ntap == 9) ? 3 : 1;
#pragma unroll 1
        for (int q = 0; q < nks; ++q) {
            __builtin_amdgcn_s_barrier();                  // everyone is past the previous chunk's window reads
            stager.stage(sg, q * MCGEN_CK, ldsA);
#pragma unroll 1
            for (int gq = 0; gq < gpc; ++gq) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                G_dma(gi + 1);
                const char* slot = ldsB0 + (gi & 1) * SLOT;
#pragma unroll
                for (int t = 0; t < TPS; ++t) {
                    if (t < ntg) {
                        const int tap = gq * ntg + t;
                        const int kh = (ntap == 9) ? tap / 3 : 0, kw = (ntap == 9) ? tap % 3 : 0;
                        const int tapoff = (kh * PC + kw) * APITCH;
                        const char* ldsB = slot + t * TAPB;
                        bf16x8 af[FM], wf[FN];
#pragma unroll
                        for (int fm = 0; fm < FM; ++fm)
                            af[fm] = *reinterpret_cast<const bf16x8*>(ldsA + a_base[fm] + tapoff);
#pragma unroll
                        for (int fn = 0; fn < FN; ++fn) {
                            union { bf16x8 v; s16x4 h[2]; } u;
                            u.h[0] = mc_tr16(ldsB + w_off[fn]);
                            u.h[1] = mc_tr16(ldsB + w_off[fn] + 16 * ROWB);
                            wf[fn] = u.v;
                        }
#pragma unroll
                        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                            for (int fm = 0; fm < FM; ++fm)
                                acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[fn], af[fm], acc[fn][fm], 0, 0, 0);
                    }
                }
                ++gi;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    conv_epilogue<T, C, BM, BN, WM, WN>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0, p.alpha);
}

// ---- "pp" form: the 3x3 main loop as a software pipeline ------------------------------------------------------------
// The dma3 form serialises, per 32-channel chunk, one exposed global round trip (the window staging) and, per tap, an
// LDS-read phase that all eight waves enter together after the group barrier: ablations of its loop (no MFMA / no
// fragment reads / no window work / no epilogue) show the pieces ADD -- nothing runs under the matrix pipe.  Here, for a
// 3x3 first segment whose tile lies inside one image:
//   * two window buffers: the global loads of chunk q + 1 are issued in the first phases of chunk q, prologue + LDS
//     stores follow ten phases later (BatchNorm affine x |code| per channel from an LDS table built once per tile);
//   * a ring of R tap tiles fed by LDS-DMA R taps ahead, retired by COUNTED s_waitcnt vmcnt (never 0 in the loop);
//   * a tap is two phases of 16 MFMAs (half of the wave's pixel fragments each), and every phase first issues the
//     fragment reads of the NEXT phase into the other register set (activations: per phase; weights: per tap), so the
//     wave's own LDS reads, window work and DMA issue sit in its MFMAs' shadow; ONE barrier per tap, after its first
//     phase, publishes the next tap's weight tile (and, once per chunk, the next window).
// Hazards: the counted wait for tap t + 1 and the barrier end phase (t, 0), its weight reads are issued in phase (t, 1);
// slot t % R was last read in phase (t - 1, 1) and is re-filled from phase (t, 1) on (tap t + R); window buffer
// (q + 1) & 1 is written in phases <= 16 of chunk q, behind the barrier of phase 16, and first read in phase 17.
// Every wave issues the same VMEM sequence per phase (window loads are unconditional, clamped to a valid address; the
// DMAs past the last tap re-load it into the free slot), so the vmcnt counts are compile-time constants (pp_vmcnt_*).
// Further (1x1) segments run after the pipeline has drained, as in the dma3 form, on window buffer 0 and ring slots 0 / 1.
template <int I, int E, typename F>
static __device__ __forceinline__ void pp_static_for(F&& f) {
    if constexpr (I < E) { f(std::integral_constant<int, I>{}); pp_static_for<I + 1, E>(f); }
}
constexpr int pp_mod(int a, int m) { return ((a % m) + m) % m; }
// VMEM instructions a wave issues in chunk-local phase ph (negative: the previous chunk), in order: window load, DMA piece
constexpr int pp_wl(int ph, int NIW) { return pp_mod(ph, 18) < NIW ? 1 : 0; }
constexpr int pp_dma(int ph, int PPW) { return PPW == 2 ? 1 : (pp_mod(ph, 2) == 1 ? 1 : 0); }
// wait at the END of even phase pc = (t, 0): this wave's pieces of tap t + 1 have landed.  Tap u's pieces are issued in
// phases (u - R, 1) [piece 0] and (u - R + 1, 0) [piece 1, PPW == 2].
constexpr int pp_vmcnt_b(int pc, int R, int NIW, int PPW) {
    const int t = pc / 2;
    const int plast = PPW == 2 ? 2 * (t + 2 - R) : 2 * (t + 1 - R) + 1;
    int n = 0;
    for (int ph = plast + 1; ph <= pc; ++ph) n += pp_wl(ph, NIW) + pp_dma(ph, PPW);
    return n;
}
// wait at the START of phase pw, before that phase's own issues: window load j (issued in phase j, ahead of that phase's DMA)
constexpr int pp_vmcnt_w(int j, int pw, int NIW, int PPW) {
    int n = pp_dma(j, PPW);
    for (int ph = j + 1; ph < pw; ++ph) n += pp_wl(ph, NIW) + pp_dma(ph, PPW);
    return n;
}
#ifndef MCGEN_PP128_WM
#define MCGEN_PP128_WM 4                   // wave rows of the 256 x 128 pp tile (4 x 2 waves of 64 x 64; 2 x 4 of 128 x 32 measured slower)
#endif
// epilogue passes of a pp tile (conv_epilogue_fast: NP): as few as keep BM / NP pixel rows of fp32 within 136 KB
#define PP_EPI_PASSES(BM, BN) (((BM) * ((BN) + 4) * 4 <= 136 * 1024) ? 1 : 2)
template <int NCNT> static __device__ __forceinline__ void pp_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NCNT) : "memory"); }
static __device__ __forceinline__ void pp_barrier() {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
}
constexpr int PP_NIW_MAX = 4;        // at most four window items per thread and chunk (registers)

// (LGW = log2 of the map width is a template parameter: every LDS address of the main loop is one per-lane base plus a
// compile-time offset, i.e. the immediate field of the ds instruction -- no address VALU, no per-fragment registers.)
// GK: the K-major / gathered-K weight form of conv_gk_kernel (compacted activations): the window side is the same (its
// channels are the image's compacted ones, the affine rows are per image), a tap tile is 32 k-rows of BN channels whose
// rows are the image's cidx entries (read through the scalar cache one chunk ahead), fragments come from transposing reads,
// and the number of chunks is the IMAGE's: ceil(active channels / 32).
template <int BM, int BN, int WM, int WN, int R, int LGW, bool GK = false, int ABL = 0>
__global__ __launch_bounds__(64 * WM * WN)
void conv_pp_kernel(const mcgen_conv_t p, const int a_bytes) {
    using T = bf16_t;
    using C = ConvCfg<T, BM, BN, WM, WN>;
    using M = Mma<T>;
    constexpr int NT = C::NT, FM = C::FM, FN = C::FN, APITCH = C::APITCH, BROW = C::BROW, BB = C::BBYTES;
    constexpr int NW = WM * WN, KB = BB / 1024, PPW = KB / NW;
    static_assert(KB % NW == 0 && (PPW == 1 || PPW == 2), "one or two 1 KB weight pieces per wave and tap");
    static_assert(FM % 2 == 0, "two pixel halves per wave");
    constexpr int HF = FM / 2;
    constexpr int PW0 = 10, PWS = 2;                       // window item j is stored in phase PW0 + PWS * j (<= 16)
    constexpr int W = 1 << LGW, TH = BM / W, PC = W + 2, PR = TH + 2, PP = PR * PC;
    constexpr int NIW = (PP * 4 + NT - 1) / NT;            // window items per thread and chunk
    static_assert(NIW >= 2 && NIW <= PP_NIW_MAX && W >= 16 && TH >= 1, "window items per thread");
    static_assert(PW0 + PWS * (NIW - 1) <= 16 && 2 * R < 18, "window stores end before the chunk's last barrier");
    constexpr int ITEM_STEP = (NT / 4) * APITCH;           // LDS distance between a thread's consecutive window items
    // K-major tap tile: 32 rows of ROWB bytes, RPPK rows per 1 KB DMA piece, LPR lanes per row
    constexpr int ROWB = BN * 2, RPPK = 1024 / ROWB, LPR = 64 / RPPK;
    static_assert(!GK || ((BN == 128 || BN == 256) && 32 * ROWB == BB), "k-major tap tiles: 2 or 4 rows per DMA piece");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsA0 = smem;
    char* const ldsB0 = smem + 2 * a_bytes;
    float* const aff = reinterpret_cast<float*>(ldsB0 + R * BB);   // [C0] scale * code, [C0] shift * code; then one 16-byte dump slot
    float* epi = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lg = lane >> 4;
    const int H = p.H, N = p.N;
    int tile_m = blockIdx.x;
    int wset = 0;
    if (p.order || p.wsel) {
        // per-mode weight sets: position i of the walk = image order[i]; its weights = set wsel[i] (both loads go out together)
        const int tpi = (H * W) / BM, pos = tile_m / tpi;
        if (p.wsel) wset = p.wsel[pos];
        if (p.order) tile_m = p.order[pos] * tpi + (tile_m - pos * tpi);
    }
    const int cout0 = blockIdx.y * BN;
    const Geo g = make_geo(BM, tile_m, H, W);              // host guarantees TI == 1 and p.W == W
    const int n_img = g.n0 < N ? g.n0 : N - 1;

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const mcgen_seg_t sg0 = seg_for_tile(p.seg[0], g);
    const int C0 = sg0.C;
    const char* wimg = reinterpret_cast<const char*>(p.w) + (size_t)wset * (size_t)p.wsel_stride * 2;
    const size_t wblock_bytes = (size_t)p.Cout_w * BROW;
    // gathered-K bookkeeping of a segment (wave-uniform): weight rows of the image, its active count
    auto seg_info = [&](int si, const int16_t*& cidx, int& cw, int& cnt) {
        const mcgen_seg_t& sg = p.seg[si];
        cw = sg.Cw > 0 ? sg.Cw : sg.C;
        cidx = nullptr; cnt = sg.C;
        if (sg.cmap) {
            const int16_t* rec = sg.cmap + (size_t)n_img * sg.cmap_stride;
            cidx = rec + cw;
            cnt = (int)mc_sload(rec + 2 * cw + 32 + 2 * ((cw + 31) >> 5));
            if (cnt > sg.C) cnt = sg.C;                    // (the host sized the compacted pitch to hold every sample)
        }
    };
    const int16_t* cidx0 = nullptr; int cw0 = C0, cnt0 = C0;
    if constexpr (GK) seg_info(0, cidx0, cw0, cnt0);
    const int nchunk = GK ? ((cnt0 + 31) >> 5) : (C0 >> 5), T0 = nchunk * 9;
    const size_t tapstride0 = (size_t)(cw0 + 1) * p.Cout_w * 2;          // (GK) bytes per tap of segment 0's image

    // ---- weight DMA: piece k of this wave -> ring slot -----------------------------------------------------------------
    // dense form: tap block blk of the [chunk][tap][cout][32] image.  GK: tap `tap` of the k-major image, row = the lane's
    // dense channel of the chunk (ksel), LDS row rho holds k(rho) so that the transposing reads deliver natural k order.
    constexpr int UPR = C::UPR, RPP = 64 / UPR;
    int d_src[PPW];                                        // dense: byte offset inside a tap block; GK: byte offset of the lane's channels inside a row
    int k0_[PPW];                                          // GK: first k of the piece's rows
    const int rsub = lane / LPR, ci = lane % LPR;          // GK: row inside the piece, 16-byte unit inside the row
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
        const int piece = wave * PPW + k;
        if constexpr (!GK) {
            const int row = piece * RPP + lane / UPR, pu = lane % UPR;
            const int lgrp = pu ^ (3 * ((row >> 3) & 1));
            d_src[k] = (cout0 + row < p.Cout_w) ? (cout0 + row) * BROW + lgrp * 16 : (lane % UPR) * 16;
            k0_[k] = 0;
        } else {
            const int rho0 = RPPK * piece, rho = rho0 + rsub;
            k0_[k] = (rho0 < 16) ? 8 * (rho0 >> 2) + (rho0 & 3) : 8 * ((rho0 - 16) >> 2) + 4 + ((rho0 - 16) & 3);
            int co = cout0 + 16 * ((ci >> 1) ^ (rho & 7)) + 8 * (ci & 1);
            if (co > p.Cout_w - 8) co = p.Cout_w - 8;
            d_src[k] = co * 2;
        }
    }
    // (GK) the lane's weight row for chunk t, piece k: cidx[32 t + k0 + rsub] (scalar-cache dwords, selected per lane) or,
    // for a dense segment, 32 t + k0 + rsub clamped to the zero row
    auto ksel_map = [&](const int16_t* cidx, int t, int k) -> int {       // (branch-free: the main loop calls it)
        const uint32_t d01 = mc_sload(cidx + 32 * t + k0_[k]);
        uint32_t sel = d01;
        if constexpr (RPPK == 4) {
            const uint32_t d23 = mc_sload(cidx + 32 * t + k0_[k] + 2);
            sel = (rsub & 2) ? d23 : d01;
        }
        return (int)((rsub & 1) ? (sel >> 16) : (sel & 0xffffu));
    };
    auto ksel = [&](const int16_t* cidx, int cw, int t, int k) -> int {
        if (cidx) return ksel_map(cidx, t, k);
        const int d = 32 * t + k0_[k] + rsub;
        return d > cw ? cw : d;
    };
    int krow_c[PPW], krow_n[PPW];                          // (GK) byte offset of the lane's row + channels inside a tap: the chunk being DMA'd / the next one
    auto dma_piece = [&](int blk, int slot, int k) {       // dense form
        const char* src = wimg + (size_t)blk * wblock_bytes + d_src[k];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(ldsB0 + slot * BB + (wave * PPW + k) * 1024), 16, 0, 0);
    };
    auto dma_piece_gk = [&](const char* wseg, size_t tapstride, int krow, int tap, int slot, int k) {
        const char* src = wseg + (size_t)tap * tapstride + krow;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(ldsB0 + slot * BB + (wave * PPW + k) * 1024), 16, 0, 0);
    };
    // weight fragment fn of this lane.  dense: row wn * (BN / WN) + 16 fn + l15 (the swizzle term depends on l15 only: one
    // base + constants).  GK: two transposing reads at rows 4 lg + q and 16 + 4 lg + q, 32-byte slot XOR-swizzled by the row
    const int w_lane = (wn * (BN / WN) + l15) * BROW + (lg ^ (3 * ((l15 >> 3) & 1))) * 16;
    // GK: row r0 = 4 lg + q4, 32-byte slot (c0 >> 4) ^ (r0 & 7) with c0 >> 4 = 4 wn + fn: the fn part is (fn ^ q4) -- one base
    // register plus an XOR per read instead of a register per fragment
    const int q4 = l15 >> 2;
    const int w_gk = (4 * lg + q4) * ROWB + 32 * ((4 * wn) ^ (4 * (lg & 1))) + 8 * (l15 & 3);
    auto wfrag = [&](const char* slotp, int fn) -> typename M::frag {
        if constexpr (GK) {
            union { bf16x8 v; s16x4 h[2]; } u;
            const int o = w_gk + 32 * (fn ^ q4);
            u.h[0] = mc_tr16(slotp + o);
            u.h[1] = mc_tr16(slotp + o + 16 * ROWB);
            return u.v;
        } else {
            return *reinterpret_cast<const typename M::frag*>(slotp + w_lane + fn * 16 * BROW);
        }
    };

    bool neg_tile = false;  // (workgroup-uniform) the tile's code row has a negative entry in front of a ReLU: outputs become NaN
    if (nchunk > 0) {       // (GK: an image without active channels contributes nothing from this segment)
    // (prologue order: weight DMAs, then the table's global loads, then the window loads of chunk 0 -- ONE exposed round trip
    // for all three; the table is written to LDS and the window transformed once everything has landed)
    // taps 0 .. R - 1 -> slots 0 .. R - 1, except the last piece of tap R - 1 (PPW == 2), which phase (0, 0) issues
    if constexpr (GK) {
#pragma unroll
        for (int k = 0; k < PPW; ++k) krow_c[k] = ksel_map(cidx0, 0, k) * p.Cout_w * 2 + d_src[k];      // (host: segment 0 of a pp launch has a map)
    }
#pragma unroll
    for (int t = 0; t < R; ++t)
#pragma unroll
        for (int k = 0; k < PPW; ++k)
            if (!(PPW == 2 && t == R - 1 && k == 1)) {
                if constexpr (GK) dma_piece_gk(wimg, tapstride0, krow_c[k], t, t, k);      // (R <= 9: all in chunk 0)
                else dma_piece(t < T0 ? t : T0 - 1, t, k);
            }

    // ---- per-channel prologue table: v -> max(v * sc' + sh', relu ? 0 : -inf) with the code folded into the affine in front of
    // the ReLU: sc' = scale * code, sh' = shift * code.  That is code * relu(.) only for code >= 0 -- always the case for
    // MultimodalController codes (a 0/1 codebook times a non-negative indicator, modules.py:58-76).  The C ABI takes any
    // float, so a tile whose code row holds a negative entry in front of a ReLU is made to FAIL LOUDLY: its outputs are NaN
    // (neg_tile, applied through the epilogue's alpha) -- a branch for the general form inside the main loop costs every
    // launch 30 spilled registers, a sign multiply per element 1.5 % of the step.  One channel per thread
    float t_sc = 1.f, t_sh = 0.f, t_cd = 1.f;
    if (tid < C0) {
        if (sg0.scale) { t_sc = sg0.scale[tid]; t_sh = sg0.shift[tid]; }
        if (sg0.code) t_cd = sg0.code[(size_t)n_img * C0 + tid];
    }

    // ---- window items of this thread: item j = window unit tid + j * NT (unit = 8 channels of one window pixel) ----------
    const T* xs0 = reinterpret_cast<const T*>(sg0.x);
    const int sub = (tid & 3) * 8;
    const int lds_item0 = (tid >> 2) * APITCH + (tid & 3) * 16;
    int it_src[NIW];                                       // element offset of the source, -1: zero padding, -2: no such item
    {
        const int Hs = sg0.ups ? (H >> 1) : H, Ws = sg0.ups ? (W >> 1) : W;
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            const int pp = (tid >> 2) + j * (NT / 4);
            const int pr = pp / PC, pc = pp - pr * PC;
            const int h = g.h0 + pr - 1, w = pc - 1;
            it_src[j] = -1;
            if (g.n0 < N && h >= 0 && h < H && w >= 0 && w < W) {
                const int hs = sg0.ups ? (h >> 1) : h, ws = sg0.ups ? (w >> 1) : w;
                it_src[j] = ((g.n0 * Hs + hs) * Ws + ws) * C0 + sub;
            }
            if (pp >= PP) it_src[j] = -2;
        }
    }
    // (window loads and window stores are inline asm: hipcc drains vmcnt to 0 in front of a DS store it can see while LDS-DMA
    // is in flight; the waits are the counted ones below)
    auto wload = [&](int j, int c0, u32x4& r) {
        const int off = it_src[j] >= 0 ? it_src[j] + c0 : 0;           // anything valid: zeroed / dropped below
        const T* ptr = xs0 + off;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(ptr) : "memory");
    };
    // (straight-line on purpose: with branches around the stores hipcc spills ~50 registers into the main loop)
    const float relu_lo = sg0.relu ? 0.f : -__builtin_inff();
    const uint32_t lds_dump = (uint32_t)reinterpret_cast<uintptr_t>(reinterpret_cast<char*>(aff) + C0 * 8);      // one 16-byte dump slot for the whole workgroup
    auto wwrite = [&](int j, const u32x4& r, char* abuf, int c0) {
        union { bf16x8 h; u32x4 w; } o;
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {                               // four channels at a time: fewer live registers
            const f32x4 a = *reinterpret_cast<const f32x4*>(aff + c0 + sub + 4 * hh);
            const f32x4 b = *reinterpret_cast<const f32x4*>(aff + C0 + c0 + sub + 4 * hh);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const float v0 = fmaxf(fmaf(__uint_as_float(r[2 * hh + i] << 16), a[2 * i], b[2 * i]), relu_lo);
                const float v1 = fmaxf(fmaf(__uint_as_float(r[2 * hh + i] & 0xffff0000u), a[2 * i + 1], b[2 * i + 1]), relu_lo);
                o.h[4 * hh + 2 * i] = (bf16_t)v0; o.h[4 * hh + 2 * i + 1] = (bf16_t)v1;
            }
        }
        if (it_src[j] < 0) o.w = u32x4{0u, 0u, 0u, 0u};               // the convolution's zero padding
        uint32_t la = (uint32_t)reinterpret_cast<uintptr_t>(abuf + lds_item0 + j * ITEM_STEP);
        if (j == NIW - 1 && it_src[j] == -2) la = lds_dump;            // no such item: the store goes to the dump slot
        asm volatile("ds_write_b128 %0, %1" :: "v"(la), "v"(o.w) : "memory");
    };
    // activation fragment fm of this lane: tile pixel wm * (BM / WM) + 16 fm + l15 (16 | W: the fm part is a constant)
    const int a_lane = (((wm * (BM / WM)) >> LGW) * PC + l15) * APITCH + lg * 16;
    auto a_off = [](int fm) constexpr { return (((fm * 16) >> LGW) * PC + ((fm * 16) & (W - 1))) * APITCH; };

    // window of chunk 0: loads first, then the table (its values and the window arrive together), then the prologue
    {
        u32x4 raw0[NIW];
#pragma unroll
        for (int j = 0; j < NIW; ++j) wload(j, 0, raw0[j]);
        if (tid < C0) { aff[tid] = t_sc * t_cd; aff[C0 + tid] = t_sh * t_cd; }
        neg_tile = __syncthreads_or((tid < C0 && sg0.relu && t_cd < 0.f) ? 1 : 0) != 0;
        if constexpr (NIW == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw0[0]), "+v"(raw0[1]), "+v"(raw0[2]), "+v"(raw0[3]) :: "memory");
        else if constexpr (NIW == 3) asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw0[0]), "+v"(raw0[1]), "+v"(raw0[2]) :: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(raw0[0]), "+v"(raw0[1]) :: "memory");
#pragma unroll
        for (int j = 0; j < NIW; ++j) wwrite(j, raw0[j], ldsA0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    pp_barrier();

    // ---- main loop ------------------------------------------------------------------------------------------------
    // registers: activations af[phase parity] (the next phase's half is read while this one multiplies); weights wf, one
    // set: in a tap's second phase the MFMAs run weight fragment by weight fragment and each fragment is re-read for
    // the next tap as soon as its four MFMAs are issued.  Phase 0's fragments are read here.
    typename M::frag af[2][HF], wf[FN];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) wf[fn] = wfrag(ldsB0, fn);
#pragma unroll
    for (int i = 0; i < HF; ++i) af[0][i] = *reinterpret_cast<const typename M::frag*>(ldsA0 + a_lane + a_off(i));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    int slot_c = 0, tcur = 0;                              // ring slot / index of the tap being computed
    u32x4 raw[NIW];
#pragma unroll 1
    for (int q = 0; q < nchunk; ++q) {
        const char* acur = ldsA0 + ((q & 1) ? a_bytes : 0) + a_lane;
        const char* anxr = ldsA0 + ((q & 1) ? 0 : a_bytes) + a_lane;
        char* anext = ldsA0 + ((q & 1) ? 0 : a_bytes);
        const bool more = q + 1 < nchunk;
        const int cn = (more ? q + 1 : q) * MCGEN_CK;                 // last chunk: re-stages itself into the idle buffer
        pp_static_for<0, 18>([&](auto PH) {
            constexpr int ph = decltype(PH)::value, k = ph >> 1, h = ph & 1;
            constexpr int np = (ph + 1) % 18, nk = np >> 1, nh = np & 1;
            constexpr int ntapoff = ((nk / 3) * PC + (nk % 3)) * APITCH;
            constexpr int cs = ph & 1, ns = cs ^ 1;                      // activation sets: this phase / the next
            // ---- activation fragments of the next phase
            {
                const char* asrc = (ph == 17) ? anxr : acur;
#pragma unroll
                for (int i = 0; i < HF; ++i) af[ns][i] = *reinterpret_cast<const typename M::frag*>(asrc + a_off(nh * HF + i) + ntapoff);
            }
            // ---- (GK) weight rows of the next chunk, ahead of the first DMA that needs them (phase 2 R - 1 ... of this chunk)
            if constexpr (GK && ph == 1) {
#pragma unroll
                for (int kk = 0; kk < PPW; ++kk) krow_n[kk] = ksel_map(cidx0, more ? q + 1 : q, kk) * p.Cout_w * 2 + d_src[kk];
            }
            // ---- window of the next chunk: stores of the loads issued in phases 0 .. 2
            if constexpr (ph >= PW0 && (ph - PW0) % PWS == 0 && (ph - PW0) / PWS < NIW) {
                constexpr int j = (ph - PW0) / PWS;
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(raw[j]) : "n"(pp_vmcnt_w(j, ph, NIW, PPW)) : "memory");
                wwrite(j, raw[j], anext, cn);
            }
            if constexpr (ph < NIW) wload(ph, cn, raw[ph]);
            // ---- weight DMA R taps ahead: piece 0 in the tap's second phase (its slot's reads are behind the barrier), piece 1 next
            if constexpr (h == 1 || PPW == 2) {
                constexpr int lt = h == 1 ? k + R : k + R - 1;             // chunk-local index of the tap to load
                constexpr int pk = h == 1 ? 0 : 1;                         // which of the wave's pieces
                const int slot = h == 1 ? slot_c : (slot_c == 0 ? R - 1 : slot_c - 1);
                if constexpr (GK) {
                    if constexpr (lt < 9) dma_piece_gk(wimg, tapstride0, krow_c[pk], lt, slot, pk);
                    else dma_piece_gk(wimg, tapstride0, krow_n[pk], more ? lt - 9 : 8, slot, pk);   // past the end: the last tap again
                } else {
                    const int blk = tcur - k + lt < T0 ? tcur - k + lt : T0 - 1;
                    dma_piece(blk, slot, pk);
                }
            }
            // ---- this phase's MFMAs
            if constexpr (h == 0) {
#pragma unroll
                for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                    for (int i = 0; i < HF; ++i) M::run(wf[fn], af[cs][i], acc[fn][i]);
            } else {
                const int sn = (slot_c + 1 == R) ? 0 : slot_c + 1;
                const char* ldsB = ldsB0 + sn * BB;
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) {
#pragma unroll
                    for (int i = 0; i < HF; ++i) M::run(wf[fn], af[cs][i], acc[fn][HF + i]);
                    wf[fn] = wfrag(ldsB, fn);                                   // the next tap's
                }
            }
            if constexpr (h == 0) {
                // every LDS read of this wave so far is complete (the slot re-filled next, the window buffer stored next) ...
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                pp_wait_vm<pp_vmcnt_b(ph, R, NIW, PPW)>();           // ... tap + 1 has landed (this wave's pieces) ...
                pp_barrier();                                          // ... everyone's; window stores of this chunk so far too
            } else {
                slot_c = (slot_c + 1 == R) ? 0 : slot_c + 1;
                ++tcur;
            }
        });
        if constexpr (GK) {
#pragma unroll
            for (int kk = 0; kk < PPW; ++kk) krow_c[kk] = krow_n[kk];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }       // nchunk > 0

    // ---- further (1x1) segments: synchronous, window buffer 0, ring slots 0 / 1 ---------------------------------------
    if (p.nseg > 1) {
        int tg = 0;
        int nrest = 0;
        // (GK: one further segment; its chunks are the image's active ones)
        const int16_t* cidx1 = nullptr; int cw1 = 8, cnt1 = 0;
        if constexpr (GK) { seg_info(1, cidx1, cw1, cnt1); nrest = (cnt1 + 31) >> 5; }
        else for (int s = 1; s < p.nseg; ++s) nrest += (p.seg[s].C + MCGEN_CK - 1) / MCGEN_CK;
        const char* wseg1 = wimg + 9 * tapstride0;
        const size_t tapstride1 = (size_t)(cw1 + 1) * p.Cout_w * 2;
        auto tail_dma = [&](int t, int slot) {
#pragma unroll
            for (int k = 0; k < PPW; ++k) {
                if constexpr (GK) dma_piece_gk(wseg1, tapstride1, ksel(cidx1, cw1, t, k) * p.Cout_w * 2 + d_src[k], 0, slot, k);
                else dma_piece(T0 + t, slot, k);
            }
        };
        pp_barrier();                                          // every wave is past its last ring / window read
        // One exposed round trip per GROUP of chunks instead of one per chunk: the windows of up to TG consecutive chunks are
        // requested together and staged side by side (over the two window buffers and ring slots 0 / 1 -- the main loop is
        // done with them), the weight tiles cycle through ring slots 2 .. 4 two chunks ahead with counted waits, one barrier
        // per chunk.  (The chunk-by-chunk form -- stage, wait, DMA, wait, 32 MFMAs -- spent 12 us per tile on the 160-channel
        // shortcut of the grouped pass: 121 of that launch's 575 us.)
        constexpr int SUBW = BM * APITCH;                      // one chunk's window of a 1x1 segment (no halo)
        constexpr int A2 = 2 * (((PP * APITCH + 1023) / 1024) * 1024);
        constexpr int TGMAX = (A2 + 2 * BB) / SUBW;
#ifndef MCGEN_PP_TG
#define MCGEN_PP_TG 4
#endif
        constexpr int TG = TGMAX < MCGEN_PP_TG ? TGMAX : MCGEN_PP_TG;
        constexpr int NI1 = (BM * 4 + NT - 1) / NT;           // window items per thread of a 1x1 chunk (no halo)
        static_assert(TG >= 1 && R >= 5, "tail groups: windows over the window buffers + two ring slots, weights in slots 2 .. 4");
        auto tslot = [](int t) { return 2 + t % 3; };
        auto tail_dma_c = [&](int t) { tail_dma(t < nrest ? t : nrest - 1, tslot(t)); };      // (past the end: the last tile again -- constant counts)
        if (nrest > 0) { tail_dma_c(0); tail_dma_c(1); }
        for (int s = 1; s < p.nseg; ++s) {
            const mcgen_seg_t sg = seg_for_tile(p.seg[s], g);
            PatchStager<T, NT, NI1, APITCH> stager;
            stager.setup(sg, g, N, H, W, tid);
            const int b_lane = (wm * (BM / WM) + l15) * APITCH + lg * 16;
            const int nch = GK ? nrest : (sg.C + MCGEN_CK - 1) / MCGEN_CK;
#pragma unroll 1
            for (int q0 = 0; q0 < nch; q0 += TG) {
                const int ng = (nch - q0) < TG ? (nch - q0) : TG;
                if (q0 > 0) __builtin_amdgcn_s_barrier();      // everyone is past the previous group's window reads
                {
                    typename PatchStager<T, NT, NI1, APITCH>::raw_t raw[TG];
#pragma unroll
                    for (int i = 0; i < TG; ++i) if (i < ng) stager.load(sg, (q0 + i) * MCGEN_CK, raw[i]);
#pragma unroll
                    for (int i = 0; i < TG; ++i) if (i < ng) stager.write(sg, (q0 + i) * MCGEN_CK, raw[i], ldsA0 + i * SUBW, g.n0 < N ? g.n0 : -1);
                }
#pragma unroll 1
                for (int i = 0; i < ng; ++i) {
                    // tile tg has landed (this wave's pieces; tile tg + 1 may still be in flight), the windows are written ...
                    if constexpr (PPW == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    pp_barrier();                              // ... everyone's; slot(tg + 2) = slot(tg - 1): its readers are past this barrier
                    tail_dma_c(tg + 2);
                    const char* ldsB = ldsB0 + tslot(tg) * BB;
                    const char* ldsW = ldsA0 + i * SUBW + b_lane;
                    typename M::frag xf[FM], yf[FN];
#pragma unroll
                    for (int fm = 0; fm < FM; ++fm) xf[fm] = *reinterpret_cast<const typename M::frag*>(ldsW + fm * 16 * APITCH);
#pragma unroll
                    for (int fn = 0; fn < FN; ++fn) yf[fn] = wfrag(ldsB, fn);
#pragma unroll
                    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                        for (int fm = 0; fm < FM; ++fm) M::run(yf[fn], xf[fm], acc[fn][fm]);
                    ++tg;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if constexpr (ABL == 2) {
        if (p.N < 0) conv_epilogue<T, C, BM, BN, WM, WN>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0, p.alpha);
#pragma unroll
        for (int i = 0; i < FN; ++i)
#pragma unroll
            for (int j = 0; j < FM; ++j) asm volatile("" :: "v"(acc[i][j]));
    } else {
        conv_epilogue<T, C, BM, BN, WM, WN, PP_EPI_PASSES(BM, BN)>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0, neg_tile ? __builtin_nanf("") : p.alpha);
    }
}

#ifndef MCGEN_KERNELS_ONLY      // (tools/micro builds include this file for one kernel's ISA)
// ---- host side ----------------------------------------------------------------------------------
struct TilePick { int BM, BN, pipe; };

// pipe codes: 0 = simple register-staged form (fp32 parity build), 5 = dma3 (4 or 8 waves as the table says),
// 11 = dma3 on the 64x64 tile with 8 waves (4 x 2), 12 = chunk-pipelined "cp" form, 20 = software-pipelined "pp" form.
// Tuning builds (-DMCGEN_TUNING) read the overrides below ONCE per process; the shipped library has no
// environment-dependent dispatch.
#ifdef MCGEN_TUNING
static long env_long(const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; }
#else
static long env_long(const char*, long dflt) { return dflt; }
#endif

// Output tile: widest channel tile the layer fills, then the largest pixel tile that still gives
// every CU a workgroup (256 CUs); fp32 (parity build) is limited by LDS to the two small tiles.
template <int BM, int BN, int WM, int WN, int R> static bool pp_fits(const mcgen_conv_t* p);

static TilePick pick_tile(const mcgen_conv_t* p, int dtype) {
    const long M = (long)p->N * p->H * p->W;
    if (p->Cout_w <= 16) {
        // skinny-N convolutions (Glow's ZeroConv2d and the input gradients of the coupling nets, image heads) are a
        // serial chain over K: with few pixels, 64-pixel tiles double the workgroups that overlap each other's
        // staging latency (measured: MCGlow step -5 %); large maps keep 128 (MCGAN's 32x32 head)
        static const int bm16_env = (int)env_long("MCGEN_CONV_BM16", 0);
        const int bm16 = bm16_env ? bm16_env : ((M <= 32768) ? 64 : 128);
        // very large maps (MCGAN's 32x32 image head at batch 128: 1024 tiles of 128 pixels, three resident per CU = two
        // rounds): 256-pixel tiles run as one round (dma3 form, 8 waves)
        static const long big16 = env_long("MCGEN_CONV_BIG16", 131072);
        if (dtype == MCGEN_BF16 && !bm16_env && big16 > 0 && M >= big16 && 256 >= 2 * p->W) return {256, 16, 5};
        const int HW16 = p->H * p->W;
        // the chunk-pipelined form (12) wins on these K-deep, latency-bound launches (-15..20 %); elsewhere the extra LDS of
        // its two-step weight ring costs more occupancy than the prefetch gains (measured), so dma3 stays
        if (dtype == MCGEN_BF16 && bm16 != 128 && ((bm16 >= 2 * p->W) || HW16 <= bm16)) return {bm16, 16, 12};
        return {128, 16, dtype == MCGEN_BF16 ? 12 : 0};
    }
    if (dtype == MCGEN_F32) return (M <= 16384 || p->Cout_w <= 64) ? TilePick{64, 64, 0} : TilePick{128, 128, 0};
    const int HW = p->H * p->W;
#ifdef MCGEN_TUNING
    int env_bm = 0, env_bn = 0, env_pipe = 0;                 // tuning builds only: re-read per launch (tools/bench_conv.py)
    if (const char* e = getenv("MCGEN_CONV_CFG"))
        if (sscanf(e, "%d,%d,%d", &env_bm, &env_bn, &env_pipe) != 3) env_bm = 0;
    if (env_bm > 0 && ((env_bm >= 2 * p->W) || HW <= env_bm)) return {env_bm, env_bn, env_pipe};
#endif
    // measured on MI355X (tools/bench_conv.py, profiles/): the LDS-DMA weight ring with three taps per
    // barrier ("dma3" form) wins on every shape; big tiles only where there are enough pixels to fill 256 CUs
    const bool rows256 = (256 >= 2 * p->W) || (HW <= 256), rows128 = (128 >= 2 * p->W) || (HW <= 128);
    // pipe 20: the software-pipelined form of the 3x3 main loop ("pp"), where its window / ring / item plan fits (pp_fits)
    static const long pp_mode = env_long("MCGEN_PP", 15);      // tuning builds: bit 0 = the 256x256 tile, bit 1 = 256x128, bit 3 = 128x256 (bit 2: gathered K)
    if (M >= 65536 && rows256 && p->Cout_w > 128) return {256, 256, ((pp_mode & 1) && pp_fits<256, 256, 2, 4, 5>(p)) ? 20 : 5};
    // (128-channel layers on a 128 x 128 FOUR-wave tile -- 76 KB of LDS, two workgroups per CU -- measured slower: the 32x32
    // layer 106 -> 152 us, the step +0.6 ms; its window is 60 % halo and a wave keeps 16 MFMAs per tap either way)
    if (M >= 65536 && rows256 && p->Cout_w > 64 && (pp_mode & 2) && pp_fits<256, 128, MCGEN_PP128_WM, 8 / MCGEN_PP128_WM, 5>(p)) return {256, 128, 20};
    if (M >= 65536 && rows128 && p->Cout_w > 64) return {128, 128, 5};
    if (M >= 32768 && rows128 && p->Cout_w > 128) return {128, 256, ((pp_mode & 8) && pp_fits<128, 256, 2, 4, 5>(p)) ? 20 : 5};
    // (from 16384 pixels: COIL100's 256-channel 8x8 layers at 2N = 256 -- 60.5 -> 41.7 us against the 64 x 64 tile, tools/bench_c64.py)
    if (M >= 16384 && p->Cout_w > 64) return {64, 128, 5};
    // 64-channel layers on large maps (COIL100's generator tail and discriminator head, 32x32): 128 pixels x 64 channels on FOUR
    // waves -- small enough for several workgroups per CU, and a 128-pixel window is 4 rows + 2 of halo where the 64 x 64 tile's is
    // 2 + 2.  tools/bench_c64.py: N = 640 128 -> 64 343 -> 148 us, 64 (+) 128 -> 64 307 -> 139; the discriminator's 84 -> 61, 70 -> 42
    // (8 waves on the same tile: 273 / 266 / 77 / 66; 256 x 64: 205 / 194 / 59 / 49).
    if (M >= 65536 && rows128 && p->Cout_w > 16) return {128, 64, 5};
    // 64x64 tile on 8 waves (4 x 2): ~15 % faster than 4 waves on 8x8 maps, bit-identical outputs.  (Its BatchNorm partial
    // sums round differently in the last bit, which once looked like a defect: the bf16 full-size digest run is bimodal
    // in its second-iteration G loss -- 1.81 or 1.70 -- under ANY 1e-7 nudge of the batch sums, see tools/digest_probe.py
    // with MCGEN_BN_PERTURB.)
    return {64, 64, 11};
}

static int patch_pixels(const mcgen_conv_t* p, int BM) {
    int best = 0;
    for (int s = 0; s < p->nseg; ++s) {
        const int pp = mcgen_patch_pixels(BM, p->H, p->W, p->seg[s].ksize);
        if (pp > best) best = pp;
    }
    return best;
}

// raises the dynamic-LDS limit of `kern` once per process (per instantiation: `raised` is the caller's static)
static int raise_lds(const void* kern, int lds, int* raised) {
    if (lds > 64 * 1024 && lds > *raised) {
        hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return mcgen_fail("conv_fused: cannot raise LDS limit to %d: %s", lds, hipGetErrorString(e));
        *raised = lds;
    }
    return 0;
}

template <typename T, int BM, int BN, int WM, int WN>
static int launch_cfg(const mcgen_conv_t* p, hipStream_t st) {
    using C = ConvCfg<T, BM, BN, WM, WN>;
    const long Mtot = (long)p->N * p->H * p->W;
    const int mt = (int)((Mtot + BM - 1) / BM);
    const int nt = (p->Cout_w + BN - 1) / BN;
    const int PP = patch_pixels(p, BM);
    MCGEN_CHECK(PP * 4 <= C::NI * C::NT, "conv_fused: patch of %d pixels exceeds the staging plan", PP);
    int a_bytes = round_up(PP * C::APITCH, 32);
    int main_bytes = a_bytes + C::BBYTES;
    int epi_bytes = C::PPX * C::EP * 4 + (p->ycmap ? YTAB_BYTES : 0);
    int red_bytes = C::PROWS * BN * 2 * 4;
    int lds = main_bytes > epi_bytes ? main_bytes : epi_bytes;
    if (red_bytes > lds) lds = red_bytes;
    MCGEN_CHECK(lds <= 160 * 1024, "conv_fused: tile %dx%d needs %d bytes of LDS", BM, BN, lds);
    auto kern = conv_fused_kernel<T, BM, BN, WM, WN>;
    static int raised = 0;
    if (int rc = raise_lds(reinterpret_cast<const void*>(kern), lds, &raised)) return rc;
    hipLaunchKernelGGL(kern, dim3(mt, nt), dim3(C::NT), lds, st, *p, a_bytes);
    MCGEN_LAUNCH_CHECK("conv_fused");
    return 0;
}

template <typename T, int BM, int BN, int WM, int WN>
static int launch_dma(const mcgen_conv_t* p, hipStream_t st) {
    using C = ConvCfg<T, BM, BN, WM, WN>;
    const long Mtot = (long)p->N * p->H * p->W;
    const int mt = (int)((Mtot + BM - 1) / BM);
    const int nt = (p->Cout_w + BN - 1) / BN;
    const int PP = patch_pixels(p, BM);
    MCGEN_CHECK(PP * 4 <= C::NI * C::NT, "conv_fused: patch of %d pixels exceeds the staging plan", PP);
    int a_bytes = round_up(PP * C::APITCH, 1024);
    // pure 1x1 launches with several chunks: the grouped form, when its three side-by-side windows fit the LDS budget
    // (K-deep ones only: at 8 chunks and fewer the plain form measured as fast or faster)
    const int a3 = round_up(3 * BM * C::APITCH, 1024);
    const bool grouped = p->nseg == 1 && p->seg[0].ksize == 1 && p->seg[0].C >= 12 * MCGEN_CK &&
                         (a3 > a_bytes ? a3 : a_bytes) + 6 * C::BBYTES <= 96 * 1024;
    if (grouped && a3 > a_bytes) a_bytes = a3;
    int lds = a_bytes + 2 * (BN <= 16 ? 9 : 3) * C::BBYTES;      // ring: two slots of a group's taps (conv_dma3_kernel: TPS)
    const int epi_bytes = C::PPX * C::EP * 4 + (p->ycmap ? YTAB_BYTES : 0), red_bytes = C::PROWS * BN * 2 * 4;
    if (epi_bytes > lds) lds = epi_bytes;
    if (red_bytes > lds) lds = red_bytes;
    MCGEN_CHECK(lds <= 160 * 1024, "conv_fused: tile %dx%d needs %d bytes of LDS", BM, BN, lds);
    if (grouped) {
        auto kg = conv_dma3g_kernel<T, BM, BN, WM, WN>;
        static int raisedg = 0;
        if (int rc = raise_lds(reinterpret_cast<const void*>(kg), lds, &raisedg)) return rc;
        hipLaunchKernelGGL(kg, dim3(mt, nt), dim3(C::NT), lds, st, *p, a_bytes);
        MCGEN_LAUNCH_CHECK("conv_fused(dma3g)");
        return 0;
    }
    auto kern = conv_dma3_kernel<T, BM, BN, WM, WN>;
    static int raised = 0;
    if (int rc = raise_lds(reinterpret_cast<const void*>(kern), lds, &raised)) return rc;
    hipLaunchKernelGGL(kern, dim3(mt, nt), dim3(C::NT), lds, st, *p, a_bytes);
    MCGEN_LAUNCH_CHECK("conv_fused(dma3)");
    return 0;
}

template <typename T, int BM, int BN, int WM, int WN>
static int launch_dma1(const mcgen_conv_t* p, hipStream_t st) {
    using C = ConvCfg<T, BM, BN, WM, WN>;
    const long Mtot = (long)p->N * p->H * p->W;
    const int mt = (int)((Mtot + BM - 1) / BM);
    const int nt = (p->Cout_w + BN - 1) / BN;
    const int PP = patch_pixels(p, BM);
    MCGEN_CHECK(PP * 4 <= C::NI * C::NT, "conv_fused: patch of %d pixels exceeds the staging plan", PP);
    const int a_bytes = round_up(PP * C::APITCH, 1024);
    int lds = a_bytes + 3 * C::BBYTES;
    const int epi_bytes = C::PPX * C::EP * 4 + (p->ycmap ? YTAB_BYTES : 0), red_bytes = C::PROWS * BN * 2 * 4;
    if (epi_bytes > lds) lds = epi_bytes;
    if (red_bytes > lds) lds = red_bytes;
    MCGEN_CHECK(lds <= 160 * 1024, "conv_fused: tile %dx%d needs %d bytes of LDS", BM, BN, lds);
    auto kern = conv_dma1_kernel<T, BM, BN, WM, WN>;
    static int raised = 0;
    if (int rc = raise_lds(reinterpret_cast<const void*>(kern), lds, &raised)) return rc;
    hipLaunchKernelGGL(kern, dim3(mt, nt), dim3(C::NT), lds, st, *p, a_bytes);
    MCGEN_LAUNCH_CHECK("conv_fused(dma1)");
    return 0;
}

template <typename T, int BM, int BN, int WM, int WN>
static int launch_cp(const mcgen_conv_t* p, hipStream_t st) {
    using C = ConvCfg<T, BM, BN, WM, WN>;
    const long Mtot = (long)p->N * p->H * p->W;
    const int mt = (int)((Mtot + BM - 1) / BM);
    const int nt = (p->Cout_w + BN - 1) / BN;
    const int PP1 = mcgen_patch_pixels(BM, p->H, p->W, 1);
    const int subw = round_up(PP1 * C::APITCH, 32);                 // one chunk's window of a 1x1 segment
    int a_bytes = 0;
    for (int s = 0; s < p->nseg; ++s) {
        const int ks = p->seg[s].ksize;
        const int PP = mcgen_patch_pixels(BM, p->H, p->W, ks);
        MCGEN_CHECK(PP * 4 <= (ks == 3 ? C::NI : (BM * 4 + C::NT - 1) / C::NT) * C::NT, "conv_fused(cp): patch of %d pixels exceeds the staging plan", PP);
        const int need = (ks == 3) ? PP * C::APITCH : 8 * subw;
        if (need > a_bytes) a_bytes = need;
    }
    a_bytes = round_up(a_bytes, 1024);
    int lds = a_bytes + 2 * 9 * C::BBYTES;
    const int epi_bytes = C::PPX * C::EP * 4 + (p->ycmap ? YTAB_BYTES : 0), red_bytes = C::PROWS * BN * 2 * 4;
    if (epi_bytes > lds) lds = epi_bytes;
    if (red_bytes > lds) lds = red_bytes;
    MCGEN_CHECK(lds <= 160 * 1024, "conv_fused(cp): tile %dx%d needs %d bytes of LDS", BM, BN, lds);
    auto kern = conv_cp_kernel<T, BM, BN, WM, WN>;
    static int raised = 0;
    if (int rc = raise_lds(reinterpret_cast<const void*>(kern), lds, &raised)) return rc;
    hipLaunchKernelGGL(kern, dim3(mt, nt), dim3(C::NT), lds, st, *p, a_bytes, subw);
    MCGEN_LAUNCH_CHECK("conv_fused(cp)");
    return 0;
}

// K-major, mode-compacted launches (conv_mc_kernel).  Tile: 256 x 256 where the map has the pixels for it, else
// 128 x 256 / 128 x 128; always inside one image.
static bool mc_tile(const mcgen_conv_t* p, int* bm, int* bn) {
    const long M = (long)p->N * p->H * p->W;
    const int HW = p->H * p->W;
    *bn = p->Cout_w > 128 ? 256 : 128;
    int b = (M >= 65536 && HW >= 256 && 256 >= 2 * p->W) ? 256 : 128;
    if (*bn == 128 && b == 256) b = 128;                   // instantiated: 256x256, 128x256, 128x128
    if (HW < b || b < 2 * p->W) return false;
    *bm = b;
    return true;
}

template <int BM, int BN, int WM, int WN>
static int launch_mc(const mcgen_conv_t* p, hipStream_t st) {
    using C = ConvCfg<bf16_t, BM, BN, WM, WN>;
    const long Mtot = (long)p->N * p->H * p->W;
    MCGEN_CHECK(Mtot % BM == 0 && p->H * p->W >= BM, "conv_fused(mc): tiles of %d pixels must lie inside one %dx%d image", BM, p->H, p->W);
    MCGEN_CHECK(!p->ycmap || p->Cout_w <= BN, "conv_fused(mc): compacted output needs all channels in one tile");
    const int mt = (int)(Mtot / BM);
    const int nt = (p->Cout_w + BN - 1) / BN;
    const int PP = patch_pixels(p, BM);
    MCGEN_CHECK(PP * 4 <= C::NI * C::NT, "conv_fused(mc): patch of %d pixels exceeds the staging plan", PP);
    const int a_bytes = round_up(PP * C::APITCH, 16);
    int lds = 2 * a_bytes + 2 * 3 * 32 * BN * 2;
    const int epi_bytes = C::PPX * C::EP * 4 + (p->ycmap ? YTAB_BYTES : 0), red_bytes = C::PROWS * BN * 2 * 4;
    if (epi_bytes > lds) lds = epi_bytes;
    if (red_bytes > lds) lds = red_bytes;
    MCGEN_CHECK(lds <= 160 * 1024, "conv_fused(mc): tile %dx%d needs %d bytes of LDS", BM, BN, lds);
    // window loads of the next dense chunk prefetched across the K step (tuning builds can switch it off)
    static const bool prefetch = env_long("MCGEN_MC_PREFETCH", 0) != 0;
    static int raised = 0, raised0 = 0;
    if (prefetch) {
        auto kern = conv_mc_kernel<BM, BN, WM, WN, true>;
        if (int rc = raise_lds(reinterpret_cast<const void*>(kern), lds, &raised)) return rc;
        hipLaunchKernelGGL(kern, dim3(mt, nt), dim3(C::NT), lds, st, *p, a_bytes);
    } else {
        auto kern = conv_mc_kernel<BM, BN, WM, WN, false>;
        if (int rc = raise_lds(reinterpret_cast<const void*>(kern), lds, &raised0)) return rc;
        hipLaunchKernelGGL(kern, dim3(mt, nt), dim3(C::NT), lds, st, *p, a_bytes);
    }
    MCGEN_LAUNCH_CHECK("conv_fused(mc)");
    return 0;
}

static int dispatch_mc(const mcgen_conv_t* p, int dtype, hipStream_t st) {
    MCGEN_CHECK(dtype == MCGEN_BF16, "conv_fused: K-major (mode-compacted) launches are bf16");
    for (int s = 0; s < p->nseg; ++s) {
        MCGEN_CHECK(p->seg[s].cmap && p->seg[s].cmap_stride >= 2 * p->seg[s].C + 32, "conv_fused: K-major launch: segment %d has no compaction map", s);
        MCGEN_CHECK(p->seg[s].C <= 2048, "conv_fused(mc): at most 2048 channels per segment");
    }
    MCGEN_CHECK(p->Cout_w % 8 == 0 && p->Cout_w >= 64, "conv_fused(mc): at least 64 output channels");
    int bm = 0, bn = 0;
    MCGEN_CHECK(mc_tile(p, &bm, &bn), "conv_fused(mc): no tile of a %dx%d map lies inside one image", p->H, p->W);
    if (bm == 256 && bn == 256) return launch_mc<256, 256, 2, 4>(p, st);
    if (bm == 128 && bn == 256) return launch_mc<128, 256, 2, 4>(p, st);
    return launch_mc<128, 128, 2, 2>(p, st);
}

template <int BM, int BN, int WM, int WN, int R, bool GK> static bool pp_fits_(const mcgen_conv_t* p);
template <typename T, int BM, int BN, int WM, int WN, int R, bool GK> static int launch_pp(const mcgen_conv_t* p, hipStream_t st);

template <int BM, int BN, int WM, int WN>
static int launch_gk(const mcgen_conv_t* p, hipStream_t st) {
    using C = ConvCfg<bf16_t, BM, BN, WM, WN>;
    const long Mtot = (long)p->N * p->H * p->W;
    MCGEN_CHECK(Mtot % BM == 0 && p->H * p->W >= BM, "conv_fused(gk): tiles of %d pixels must lie inside one %dx%d image", BM, p->H, p->W);
    MCGEN_CHECK(!p->ycmap || p->Cout_w <= BN, "conv_fused(gk): compacted output needs all channels in one tile");
    const int mt = (int)(Mtot / BM);
    const int nt = (p->Cout_w + BN - 1) / BN;
    const int PP = patch_pixels(p, BM);
    MCGEN_CHECK(PP * 4 <= C::NI * C::NT, "conv_fused(gk): patch of %d pixels exceeds the staging plan", PP);
    const int a_bytes = round_up(PP * C::APITCH, 1024);
    int lds = a_bytes + 2 * 3 * 32 * BN * 2;
    const int epi_bytes = C::PPX * C::EP * 4 + (p->ycmap ? YTAB_BYTES : 0), red_bytes = C::PROWS * BN * 2 * 4;
    if (epi_bytes > lds) lds = epi_bytes;
    if (red_bytes > lds) lds = red_bytes;
    MCGEN_CHECK(lds <= 160 * 1024, "conv_fused(gk): tile %dx%d needs %d bytes of LDS", BM, BN, lds);
    auto kern = conv_gk_kernel<BM, BN, WM, WN>;
    static int raised = 0;
    if (int rc = raise_lds(reinterpret_cast<const void*>(kern), lds, &raised)) return rc;
    hipLaunchKernelGGL(kern, dim3(mt, nt), dim3(C::NT), lds, st, *p, a_bytes);
    MCGEN_LAUNCH_CHECK("conv_fused(gk)");
    return 0;
}

static int dispatch_gk(const mcgen_conv_t* p, int dtype, hipStream_t st) {
    MCGEN_CHECK(dtype == MCGEN_BF16, "conv_fused: K-major launches are bf16");
    for (int s = 0; s < p->nseg; ++s) {
        const mcgen_seg_t& g = p->seg[s];
        const int cw = g.Cw > 0 ? g.Cw : g.C;
        MCGEN_CHECK(cw % 8 == 0 && cw <= 2048 && g.C <= cw + 32, "conv_fused(gk): segment %d: bad channel counts C=%d Cw=%d", s, g.C, cw);
        MCGEN_CHECK(g.cmap || g.C == cw, "conv_fused(gk): segment %d: compacted channels need the map that orders them", s);
        MCGEN_CHECK(!g.cmap || g.cmap_stride >= 2 * cw + 32, "conv_fused(gk): segment %d: map stride too small", s);
        MCGEN_CHECK(!g.cmap || g.code == nullptr, "conv_fused(gk): segment %d: the code of a compacted segment rides in its scale / shift rows", s);
    }
    MCGEN_CHECK(p->Cout_w % 8 == 0 && p->Cout_w >= 64, "conv_fused(gk): at least 64 output channels");
    int bm = 0, bn = 0;
    MCGEN_CHECK(mc_tile(p, &bm, &bn), "conv_fused(gk): no tile of a %dx%d map lies inside one image", p->H, p->W);
    // the big tile's 3x3 launches with a mapped first segment: the software-pipelined form (tuning builds: MCGEN_PP bit 2 off)
    static const long pp_mode = env_long("MCGEN_PP", 7);
    if (bm == 256 && bn == 256 && (pp_mode & 4) && pp_fits_<256, 256, 2, 4, 5, true>(p)) return launch_pp<bf16_t, 256, 256, 2, 4, 5, true>(p, st);
    if (bm == 256 && bn == 256) return launch_gk<256, 256, 2, 4>(p, st);
    if (bm == 128 && bn == 256) return launch_gk<128, 256, 2, 4>(p, st);
    return launch_gk<128, 128, 2, 2>(p, st);
}

// "pp" form (conv_pp_kernel): 3x3 first segment with whole 32-channel chunks, further segments 1x1, tile inside one image,
// window of two or three items per thread, two windows + ring + table within the CU's LDS.
template <int BM, int BN, int WM, int WN, int R, bool GK>
static bool pp_fits_(const mcgen_conv_t* p) {
    using C = ConvCfg<bf16_t, BM, BN, WM, WN>;
    if (p->w_layout != (GK ? 2 : 0) || p->H * p->W < BM || (p->W != 16 && p->W != 32)) return false;
    if (p->seg[0].ksize != 3 || p->seg[0].C % MCGEN_CK != 0 || p->seg[0].C < 2 * MCGEN_CK || p->seg[0].C > C::NT) return false;
    if ((p->seg[0].cmap != nullptr) != GK) return false;               // gathered K: the first segment's rows come from its map
    for (int s = 1; s < p->nseg; ++s) if (p->seg[s].ksize != 1 || (!GK && p->seg[s].cmap)) return false;
    const int PP = mcgen_patch_pixels(BM, p->H, p->W, 3);
    if (PP * 4 > PP_NIW_MAX * C::NT || PP * 4 <= C::NT) return false;
    const int a_bytes = round_up(PP * C::APITCH, 1024);
    const int lds = 2 * a_bytes + R * C::BBYTES + p->seg[0].C * 8 + 16;
    return lds <= 160 * 1024;
}
template <int BM, int BN, int WM, int WN, int R>
static bool pp_fits(const mcgen_conv_t* p) { return pp_fits_<BM, BN, WM, WN, R, false>(p); }
template <typename T, int BM, int BN, int WM, int WN, int R, bool GK>
static int launch_pp(const mcgen_conv_t* p, hipStream_t st) {
    static_assert(sizeof(T) == 2, "the pp form is bf16");
    using C = ConvCfg<bf16_t, BM, BN, WM, WN>;
    const long Mtot = (long)p->N * p->H * p->W;
    const int mt = (int)(Mtot / BM);
    const int nt = (p->Cout_w + BN - 1) / BN;
    const int PP = mcgen_patch_pixels(BM, p->H, p->W, 3);
    const int a_bytes = round_up(PP * C::APITCH, 1024);
    int lds = 2 * a_bytes + R * C::BBYTES + p->seg[0].C * 8 + 16;
    const int epi_bytes = C::PPX * C::EP * 4 + (p->ycmap ? YTAB_BYTES : 0), red_bytes = C::PROWS * BN * 2 * 4;
    const int epi_fast = (BM / PP_EPI_PASSES(BM, BN)) * C::EP * 4 + (p->ycmap ? YTAB_BYTES : 0);      // the merged passes of the fast epilogue
    if (epi_bytes > lds) lds = epi_bytes;
    if (epi_fast > lds) lds = epi_fast;
    if (red_bytes > lds) lds = red_bytes;
    MCGEN_CHECK(lds <= 160 * 1024, "conv_fused(pp): tile %dx%d needs %d bytes of LDS", BM, BN, lds);
    void (*kern)(const mcgen_conv_t, const int) = p->W == 32 ? conv_pp_kernel<BM, BN, WM, WN, R, 5, GK> : conv_pp_kernel<BM, BN, WM, WN, R, 4, GK>;
#ifdef MCGEN_TUNING
    static const long abl = env_long("MCGEN_PP_ABL", 0);
    if (abl == 2 && p->W == 32) kern = conv_pp_kernel<BM, BN, WM, WN, R, 5, GK, 2>;
#endif
    // (LDS limit per kernel symbol; tuning builds switch symbols, so they set it on every launch)
#ifdef MCGEN_TUNING
    int raised_now = 0;
    if (int rc = raise_lds(reinterpret_cast<const void*>(kern), lds, &raised_now)) return rc;
#else
    static int raised[2] = {0, 0};
    if (int rc = raise_lds(reinterpret_cast<const void*>(kern), lds, &raised[p->W == 32])) return rc;
#endif
    hipLaunchKernelGGL(kern, dim3(mt, nt), dim3(C::NT), lds, st, *p, a_bytes);
    MCGEN_LAUNCH_CHECK("conv_fused(pp)");
    return 0;
}

typedef int (*launch_fn)(const mcgen_conv_t*, hipStream_t);
struct CfgEntry { int BM, BN, pipe; launch_fn fn; };

static const CfgEntry* f32_table(int* n) {
    using T = float;
    static const CfgEntry t[] = {
        {128, 16, 0, launch_cfg<T, 128, 16, 4, 1>},
        {64, 64, 0, launch_cfg<T, 64, 64, 2, 2>},
        {128, 128, 0, launch_cfg<T, 128, 128, 2, 2>},
    };
    *n = (int)(sizeof(t) / sizeof(t[0]));
    return t;
}
// every entry is reachable from pick_tile's default policy (the parity tests in tests/test_kernels_gpu.py name each
// one); tuning builds add the alternatives that were measured and lost
static const CfgEntry* bf16_table(int* n) {
    using T = bf16_t;
    static const CfgEntry t[] = {
        {256, 256, 5, launch_dma<T, 256, 256, 2, 4>}, {128, 256, 5, launch_dma<T, 128, 256, 2, 4>},
        {128, 128, 5, launch_dma<T, 128, 128, 2, 2>}, {64, 128, 5, launch_dma<T, 64, 128, 2, 2>},
        {64, 64, 11, launch_dma<T, 64, 64, 4, 2>},    {256, 16, 5, launch_dma<T, 256, 16, 8, 1>},
        {64, 16, 12, launch_cp<T, 64, 16, 4, 1>},     {128, 16, 12, launch_cp<T, 128, 16, 4, 1>},
        {256, 256, 20, launch_pp<T, 256, 256, 2, 4, 5, false>}, {256, 128, 20, launch_pp<T, 256, 128, MCGEN_PP128_WM, 8 / MCGEN_PP128_WM, 5, false>},
        {128, 256, 20, launch_pp<T, 128, 256, 2, 4, 5, false>},
        {128, 64, 5, launch_dma<T, 128, 64, 2, 2>},
#ifdef MCGEN_TUNING
        {128, 256, 4, launch_dma1<T, 128, 256, 1, 4>}, {128, 256, 14, launch_dma1<T, 128, 256, 2, 4>},
        {256, 128, 15, launch_dma<T, 256, 128, 4, 1>}, {256, 128, 16, launch_dma<T, 256, 128, 2, 2>},
        {64, 128, 4, launch_dma1<T, 64, 128, 1, 4>},   {128, 128, 4, launch_dma1<T, 128, 128, 1, 4>},
        {256, 128, 5, launch_dma<T, 256, 128, 4, 2>}, {64, 64, 5, launch_dma<T, 64, 64, 2, 2>},
        {128, 16, 5, launch_dma<T, 128, 16, 4, 1>},   {64, 16, 5, launch_dma<T, 64, 16, 4, 1>},
        {64, 64, 12, launch_cp<T, 64, 64, 2, 2>},     {32, 64, 12, launch_cp<T, 32, 64, 1, 2>},
        {64, 128, 9, launch_dma<T, 64, 128, 4, 4>},   {64, 64, 9, launch_dma<T, 64, 64, 4, 4>},
        {256, 64, 5, launch_dma<T, 256, 64, 4, 2>},   {128, 64, 15, launch_dma<T, 128, 64, 4, 2>},
        {256, 64, 15, launch_dma<T, 256, 64, 2, 2>},  {128, 64, 16, launch_dma<T, 128, 64, 2, 1>},
#endif
    };
    *n = (int)(sizeof(t) / sizeof(t[0]));
    return t;
}

static int dispatch(const mcgen_conv_t* p, int dtype, const TilePick& t, hipStream_t st) {
    int n = 0;
    const CfgEntry* tab = dtype == MCGEN_BF16 ? bf16_table(&n) : f32_table(&n);
    for (int i = 0; i < n; ++i)
        if (tab[i].BM == t.BM && tab[i].BN == t.BN && tab[i].pipe == t.pipe) return tab[i].fn(p, st);
    return mcgen_fail("conv_fused: no instantiation for tile %dx%d pipe=%d dtype=%d", t.BM, t.BN, t.pipe, dtype);
}

static int validate(const mcgen_conv_t* p) {
    MCGEN_CHECK(p && p->nseg >= 1 && p->nseg <= 2, "conv_fused: nseg must be 1 or 2");
    MCGEN_CHECK(p->N > 0 && ilog2_exact(p->H) >= 0 && ilog2_exact(p->W) >= 0, "conv_fused: H and W must be powers of two (got %dx%d)", p->H, p->W);
    MCGEN_CHECK(p->W <= 64 && p->H * p->W >= 1, "conv_fused: W up to 64 supported");
    MCGEN_CHECK(p->Cout > 0 && p->Cout_w == round_up(p->Cout, 16), "conv_fused: Cout_w must be Cout rounded up to 16");
    MCGEN_CHECK(p->Cy % 8 == 0 && p->Cy > 0, "conv_fused: Cy must be a positive multiple of 8");
    MCGEN_CHECK(p->w && p->y, "conv_fused: null weight image or output");
    for (int s = 0; s < p->nseg; ++s) {
        const mcgen_seg_t& g = p->seg[s];
        MCGEN_CHECK(g.x && g.C > 0 && g.C % 8 == 0, "conv_fused: segment %d: C must be a positive multiple of 8", s);
        MCGEN_CHECK(g.ksize == 1 || g.ksize == 3, "conv_fused: segment %d: ksize must be 1 or 3", s);
        MCGEN_CHECK(!g.ups || (p->H >= 2 && p->W >= 2), "conv_fused: upsampled segment needs H, W >= 2");
        MCGEN_CHECK(g.group_n >= 0 && (g.group_n == 0 || p->N % g.group_n == 0), "conv_fused: segment %d: group_n must divide N", s);
    }
    if (p->pool) MCGEN_CHECK(p->H >= 2 && p->W >= 2, "conv_fused: pooling needs H, W >= 2");
    MCGEN_CHECK(p->stats_mode >= 0 && p->stats_mode <= 2, "conv_fused: bad stats_mode");
    MCGEN_CHECK(p->stats_mode != 2 || (p->gate_x && p->gmean && p->grstd), "conv_fused: stats_mode 2 needs gate_x, gmean, grstd");
    MCGEN_CHECK(p->stats_mode == 0 || p->stats, "conv_fused: stats_mode set without a stats buffer");
    MCGEN_CHECK(p->w_layout >= 0 && p->w_layout <= 2, "conv_fused: unknown weight layout %d", p->w_layout);
    if (p->wsel || p->order) {
        MCGEN_CHECK(p->w_layout == 0 && (p->y_group == 0 || !p->order), "conv_fused: per-mode weight sets (wsel / order) go with the chunked weight image (w_layout 0)");
        MCGEN_CHECK(!p->wsel || p->wsel_stride > 0, "conv_fused: wsel needs wsel_stride");
    }
    if (p->yperm) {
        MCGEN_CHECK(p->wsel && !p->order && !p->ycmap, "conv_fused: permuted weight rows (yperm) come with wsel, without order / ycmap");
        MCGEN_CHECK(!p->pool && !p->res && !p->gate_x && !p->ocode && !p->tanh_out, "conv_fused: a compacted output takes bias and statistics only");
        MCGEN_CHECK(p->Cy % 8 == 0 && p->Cy <= p->Cout && p->Cout % 8 == 0 && p->Cout_w == p->Cout && p->Cout_w <= 256 && p->yperm_stride >= p->Cout && p->yperm_stride % 8 == 0,
                    "conv_fused: yperm: bad pitch %d / stride %d for %d channels", p->Cy, p->yperm_stride, p->Cout);
    }
    if (p->ycmap) {
        MCGEN_CHECK(!p->pool && !p->res && !p->gate_x && !p->ocode && !p->tanh_out && !p->bias2, "conv_fused: a compacted output takes bias and statistics only");
        MCGEN_CHECK(p->Cy % 32 == 0 && p->Cy <= round_up(p->Cout, 8) + 32 && p->ycmap_stride >= 2 * round_up(p->Cout, 8) + 32,
                    "conv_fused: compacted output: bad pitch %d / map stride %d", p->Cy, p->ycmap_stride);
        MCGEN_CHECK(p->Cout_w <= 256, "conv_fused: a compacted output needs all channels in one tile");
    } else if (!p->yperm) {
        MCGEN_CHECK(p->Cy >= p->Cout, "conv_fused: Cy must be >= Cout");
    }
    return 0;
}

}  // namespace

extern "C" int mcgen_conv_m_tiles(const mcgen_conv_t* p, int dtype) {
    if (!p) return 0;
    if (p->w_layout != 0) {
        int bm = 0, bn = 0;
        if (!mc_tile(p, &bm, &bn)) return 0;
        return (int)(((long)p->N * p->H * p->W + bm - 1) / bm);
    }
    if (mcgen_conv_smap_ok(p, dtype)) return p->N;              // whole-image kernel: one statistics row per image
    if (const int bm = mcgen_conv_px1_bm(p, dtype)) return (int)((long)p->N * p->H * p->W / bm);
    const TilePick t = pick_tile(p, dtype);
    const long Mtot = (long)p->N * p->H * p->W;
    return (int)((Mtot + t.BM - 1) / t.BM);
}

extern "C" int mcgen_conv_tile(const mcgen_conv_t* p, int dtype, int* bm, int* bn) {
    if (!p || !bm || !bn) return mcgen_fail("conv_tile: null pointer");
    if (p->w_layout != 0) {
        MCGEN_CHECK(mc_tile(p, bm, bn), "conv_tile: no K-major tile for a %dx%d map", p->H, p->W);
        return 0;
    }
    const TilePick t = pick_tile(p, dtype);
    *bm = t.BM; *bn = t.BN;
    return 0;
}

extern "C" int mcgen_conv_form(const mcgen_conv_t* p, int dtype) {
    if (!p || p->w_layout != 0 || p->order) return 0;
    if (p->wsel) return mcgen_conv_head_ok(p, dtype) ? 5 : 0;
    if (mcgen_conv_skinny_ok(p, dtype)) return 1;
    if (mcgen_conv_smap_ok(p, dtype)) return 2;
    if (mcgen_conv_px1_bm(p, dtype)) return 3;
    if (mcgen_conv_c8_ok(p, dtype)) return 4;
    if (mcgen_conv_head_ok(p, dtype)) return 5;
    return 0;
}

extern "C" int mcgen_conv_fused(const mcgen_conv_t* p, int dtype, void* stream) {
    if (int rc = validate(p)) return rc;
    if (p->y_group != 0) {
        MCGEN_CHECK(mcgen_conv_head_ok(p, dtype), "conv_fused: the paired output layout (y_group) is built for the image head only "
                    "(bf16, one 3x3 segment to <= 8 channels of pitch 8 on 32x32 maps, N %% y_group == 0)");
        return mcgen_conv_head(p, reinterpret_cast<hipStream_t>(stream));
    }
    if (p->w_layout == 1) return dispatch_mc(p, dtype, reinterpret_cast<hipStream_t>(stream));
    if (p->w_layout == 2) return dispatch_gk(p, dtype, reinterpret_cast<hipStream_t>(stream));
    for (int s = 0; s < p->nseg; ++s) MCGEN_CHECK(p->seg[s].cmap == nullptr, "conv_fused: a compaction map needs a K-major launch (w_layout 1 or 2)");
    if (p->wsel && mcgen_conv_head_ok(p, dtype)) return mcgen_conv_head(p, reinterpret_cast<hipStream_t>(stream));
    if (p->wsel || p->order) {
        const TilePick tw = pick_tile(p, dtype);
        MCGEN_CHECK(dtype == MCGEN_BF16 && tw.pipe == 20, "conv_fused: per-mode weight sets need the software-pipelined bf16 form "
                    "(3x3 first segment of 64 .. 512 channels in whole chunks, 16x16 / 32x32 maps, >= 65536 pixels)");
        if (p->ycmap) MCGEN_CHECK(tw.BM <= p->H * p->W && p->Cout_w <= tw.BN, "conv_fused: compacted output: the %dx%d tile must hold all %d channels", tw.BM, tw.BN, p->Cout_w);
        if (p->yperm) MCGEN_CHECK(tw.BM <= p->H * p->W && p->Cout_w == tw.BN, "conv_fused: yperm: the %dx%d tile must hold exactly the %d channels", tw.BM, tw.BN, p->Cout_w);
        return dispatch(p, dtype, tw, reinterpret_cast<hipStream_t>(stream));
    }
    if (mcgen_conv_skinny_ok(p, dtype)) return mcgen_conv_skinny(p, reinterpret_cast<hipStream_t>(stream));
    if (mcgen_conv_smap_ok(p, dtype)) return mcgen_conv_smap(p, reinterpret_cast<hipStream_t>(stream));
    if (mcgen_conv_px1_bm(p, dtype)) return mcgen_conv_px1(p, reinterpret_cast<hipStream_t>(stream));
    if (mcgen_conv_c8_ok(p, dtype)) return mcgen_conv_c8(p, reinterpret_cast<hipStream_t>(stream));
    if (mcgen_conv_head_ok(p, dtype)) return mcgen_conv_head(p, reinterpret_cast<hipStream_t>(stream));
    const TilePick t = pick_tile(p, dtype);
    if (p->ycmap) MCGEN_CHECK(dtype == MCGEN_BF16 && t.BM <= p->H * p->W && p->Cout_w <= t.BN,
                              "conv_fused: compacted output: the %dx%d tile must lie inside one image and hold all %d channels", t.BM, t.BN, p->Cout_w);
    // pooling / whole-row tiles need at least two rows per tile
    MCGEN_CHECK(t.BM >= 2 * p->W || p->H * p->W <= t.BM, "conv_fused: tile of %d pixels too small for W=%d", t.BM, p->W);
    for (int s = 0; s < p->nseg; ++s) {
        const int ti = t.BM > p->H * p->W ? t.BM / (p->H * p->W) : 1;          // images per tile
        MCGEN_CHECK(p->seg[s].group_n == 0 || p->seg[s].group_n % ti == 0,
                    "conv_fused: a tile of %d images would straddle BatchNorm groups of %d images", ti, p->seg[s].group_n);
    }
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype != MCGEN_F32 && dtype != MCGEN_BF16) return mcgen_fail("conv_fused: unknown dtype %d", dtype);
    return dispatch(p, dtype, t, st);
}
#endif  // MCGEN_KERNELS_ONLY
