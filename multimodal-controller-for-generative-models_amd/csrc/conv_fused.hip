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

// Shared epilogue: accumulators -> LDS (fp32 [pixel][cout]) -> fused output pass, PPX pixels at a time.
template <typename T, typename C, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void conv_epilogue(const mcgen_conv_t& p, const Geo& g, f32x4 (&acc)[C::FN][C::FM],
                                              float* epi, int tid, int wm, int wn, int l15, int lg,
                                              int tile_m, int cout0) {
    using E = Elem<T>;
    constexpr int NT = C::NT, FM = C::FM, FN = C::FN, EP = C::EP;
    const int H = p.H, W = p.W, N = p.N;
    constexpr int CH = C::CH, PROWS = C::PROWS, PPX = C::PPX;
    const int ch = tid % CH, prow = tid / CH;
    const int co = cout0 + ch * 8;             // first channel of this thread's chunk
    const int Ho = p.pool ? (H >> 1) : H, Wo = p.pool ? (W >> 1) : W;
    const bool chunk_live = co < p.Cy;
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
            for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], p.alpha, bias[i]);
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
                    for (int i = 0; i < 8; ++i) { s1[i] += v[i]; s2[i] += v[i] * ((xv[i] - gme[i]) * grs[i]); }
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
                for (int i = 0; i < 8; ++i) { s1[i] += v[i]; s2[i] += v[i] * v[i]; }
            }
            E::store8(y + opix * p.Cy + co, v);
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
            if (cout0 + c < p.Cy) {
                p.stats[((size_t)tile_m * 2 + 0) * p.Cy + cout0 + c] = a;
                p.stats[((size_t)tile_m * 2 + 1) * p.Cy + cout0 + c] = b;
            }
        }
    }
}

// Software pipeline (one barrier per tap):
//   step t reads weights from Bbuf[t&1] and the input window from Abuf[cur];
//   the weight tile of step t+1 sits in registers (its global loads were issued during step t-1) and is
//   written to Bbuf[(t+1)&1] after step t's MFMAs; the loads of step t+2 are issued right after;
//   the NEXT chunk's input window is fetched at the first tap of the current chunk and goes through
//   the prologue into Abuf[cur^1] at its last tap.
template <typename T, int BM, int BN, int WM, int WN, bool PIPE>
__global__ __launch_bounds__(64 * WM * WN)
void conv_fused_kernel(const mcgen_conv_t p, const int a_bytes) {
    using C = ConvCfg<T, BM, BN, WM, WN>;
    using M = Mma<T>;
    constexpr int NT = C::NT, FM = C::FM, FN = C::FN, ESZ = C::ESZ, APITCH = C::APITCH, BROW = C::BROW, EP = C::EP;
    constexpr int UPR = C::UPR, NU = C::NU;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsA0 = smem;                                   // PIPE: two input-window buffers
    char* const ldsB0 = smem + (PIPE ? 2 : 1) * a_bytes;        // PIPE: two weight-tile buffers
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
    int total_steps = 0;
    for (int s = 0; s < p.nseg; ++s)
        total_steps += ((p.seg[s].C + MCGEN_CK - 1) / MCGEN_CK) * p.seg[s].ksize * p.seg[s].ksize;
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

    if constexpr (!PIPE) {
        // simple form: one window buffer, one weight buffer, two barriers per tap; latency is hidden by
        // running several workgroups per CU (small LDS / register footprint)
        int blk = 0;
        for (int s = 0; s < p.nseg; ++s) {
            const mcgen_seg_t sg = p.seg[s];
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
    } else {
    int blk = 0, par = 0, acur = 0;
        B_load(0);
        B_write(ldsB0);
        if (total_steps > 1) B_load(1);

        for (int s = 0; s < p.nseg; ++s) {
            const mcgen_seg_t sg = p.seg[s];
            const int halo = sg.ksize >> 1;
            const int PR = g.TH + 2 * halo, PC = W + 2 * halo;
            PatchStager<T, NT, C::NI, APITCH> stager;
            stager.setup(sg, g, N, H, W, tid);
            int a_base[FM];                            // per-lane LDS offset of each pixel fragment at tap (0,0)
#pragma unroll
            for (int fm = 0; fm < FM; ++fm) {
                const int m = wm * (BM / WM) + fm * 16 + l15;
                const int ti = m >> g.lgTHW, rem = m & ((1 << g.lgTHW) - 1);
                const int r = rem >> g.lgW, c = rem & (W - 1);
                a_base[fm] = ((ti * PR + r) * PC + c) * APITCH + lg * 8 * ESZ;
            }
            const int nchunk = (sg.C + MCGEN_CK - 1) / MCGEN_CK;
            const int ntap = sg.ksize * sg.ksize;
            // first chunk of the segment: staged synchronously into the idle window buffer
            acur ^= 1;
            stager.stage(sg, 0, ldsA0 + acur * a_bytes);
            __syncthreads();
            typename PatchStager<T, NT, C::NI, APITCH>::raw_t araw;
#pragma unroll 1
            for (int q = 0; q < nchunk; ++q) {
                const bool more = (q + 1 < nchunk);
                const char* ldsA = ldsA0 + acur * a_bytes;
#pragma unroll 1
                for (int tap = 0; tap < ntap; ++tap) {
                    if (tap == 0 && more) stager.load(sg, (q + 1) * MCGEN_CK, araw);
                    const int kh = (sg.ksize == 3) ? tap / 3 : 0, kw = (sg.ksize == 3) ? tap % 3 : 0;
                    const int tapoff = (kh * PC + kw) * APITCH;
                    const char* ldsB = ldsB0 + par * C::BBYTES;
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
                    if (blk + 1 < total_steps) {
                        B_write(ldsB0 + (par ^ 1) * C::BBYTES);
                        if (blk + 2 < total_steps) B_load(blk + 2);
                    }
                    if (tap == ntap - 1 && more) stager.write(sg, (q + 1) * MCGEN_CK, araw, ldsA0 + (acur ^ 1) * a_bytes);
                    __syncthreads();
                    par ^= 1; ++blk;
                }
                acur ^= 1;
            }
            acur ^= 1;                                  // undo the last flip: acur is the buffer read last
        }
    }

    // ---- epilogue ------------------------------------------------------------------------------
    conv_epilogue<T, C, BM, BN, WM, WN>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0);
}

// ---- "direct" form -----------------------------------------------------------------------------
// Weights never touch LDS: the weight image stores each (chunk, tap) block as [cout][32], so the MFMA
// A-operand fragment of 16 output channels is one contiguous, L2-resident 1 KB block that a wave loads
// straight into registers, one tap ahead of its use.  The four waves split the OUTPUT CHANNELS (each wave
// owns all BM pixels x BN/4 channels), so no weight byte is loaded twice and the only shared data is the
// input window, double-buffered in LDS: ONE barrier per 32-channel chunk (9 taps of MFMAs) instead of two
// per tap.  Between barriers a wave issues back-to-back MFMAs with its next weights and the next window
// in flight.
template <typename T, int BM, int BN>
__global__ __launch_bounds__(256)
void conv_direct_kernel(const mcgen_conv_t p, const int a_bytes) {
    constexpr int WM = 1, WN = 4;
    using C = ConvCfg<T, BM, BN, WM, WN>;
    using M = Mma<T>;
    constexpr int NT = C::NT, FM = C::FM, FN = C::FN, ESZ = C::ESZ, APITCH = C::APITCH, BROW = C::BROW;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsA0 = smem;                      // two input-window buffers
    float* epi = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wn = tid >> 6, wm = 0;
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
    int w_off[FN];                                 // byte offset of this lane's 8-channel group, -1 = beyond Cout_w
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
        const int row = cout0 + wn * (BN / WN) + fn * 16 + l15;
        w_off[fn] = (row < p.Cout_w) ? row * BROW + lg * 8 * ESZ : -1;
    }
    typename M::frag wfc[FN], wfn[FN];
    auto W_load = [&](int blk, typename M::frag (&dst)[FN]) {
        const char* wb = wimg + (size_t)blk * wblock_bytes;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            typename M::frag z = {};
            dst[fn] = (w_off[fn] >= 0) ? *reinterpret_cast<const typename M::frag*>(wb + w_off[fn]) : z;
        }
    };

    int blk = 0, acur = 0;
    W_load(0, wfc);
    for (int s = 0; s < p.nseg; ++s) {
        const mcgen_seg_t sg = p.seg[s];
        const int halo = sg.ksize >> 1;
        const int PR = g.TH + 2 * halo, PC = W + 2 * halo;
        PatchStager<T, NT, C::NI, APITCH> stager;
        stager.setup(sg, g, N, H, W, tid);
        int a_base[FM];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int m = fm * 16 + l15;
            const int ti = m >> g.lgTHW, rem = m & ((1 << g.lgTHW) - 1);
            const int r = rem >> g.lgW, c = rem & (W - 1);
            a_base[fm] = ((ti * PR + r) * PC + c) * APITCH + lg * 8 * ESZ;
        }
        const int nchunk = (sg.C + MCGEN_CK - 1) / MCGEN_CK;
        const int ntap = sg.ksize * sg.ksize;
        // first chunk of the segment goes into the idle buffer (last read one chunk ago, a barrier back)
        acur ^= 1;
        stager.stage(sg, 0, ldsA0 + acur * a_bytes);
        __syncthreads();
        typename PatchStager<T, NT, C::NI, APITCH>::raw_t araw;
#pragma unroll 1
        for (int q = 0; q < nchunk; ++q) {
            const bool more = (q + 1 < nchunk);
            const char* ldsA = ldsA0 + acur * a_bytes;
#pragma unroll 1
            for (int tap = 0; tap < ntap; ++tap) {
                if (tap == 0 && more) stager.load(sg, (q + 1) * MCGEN_CK, araw);
                if (blk + 1 < total_steps) W_load(blk + 1, wfn);
                const int kh = (sg.ksize == 3) ? tap / 3 : 0, kw = (sg.ksize == 3) ? tap % 3 : 0;
                const int tapoff = (kh * PC + kw) * APITCH;
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) {
                    const typename M::frag af = *reinterpret_cast<const typename M::frag*>(ldsA + a_base[fm] + tapoff);
#pragma unroll
                    for (int fn = 0; fn < FN; ++fn) M::run(wfc[fn], af, acc[fn][fm]);
                }
                if (tap == ntap - 1 && more) stager.write(sg, (q + 1) * MCGEN_CK, araw, ldsA0 + (acur ^ 1) * a_bytes);
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) wfc[fn] = wfn[fn];
                ++blk;
            }
            if (more) { __syncthreads(); acur ^= 1; }
        }
    }
    __syncthreads();                               // all window reads done before LDS becomes the epilogue tile
    conv_epilogue<T, C, BM, BN, WM, WN>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0);
}

// ---- "direct ring" form ------------------------------------------------------------------------
// As the direct form, but the weight fragments live in a rolling ring of RING register slots over the
// linear (segment, chunk, tap) step sequence: step t consumes slot t % RING and immediately re-issues
// the load of step t + RING into it, so every weight load has RING taps of MFMAs to land.
template <typename T, int BM, int BN>
__global__ __launch_bounds__(256)
void conv_ring_kernel(const mcgen_conv_t p, const int a_bytes) {
    constexpr int WM = 1, WN = 4, RING = 9;
    using C = ConvCfg<T, BM, BN, WM, WN>;
    using M = Mma<T>;
    constexpr int NT = C::NT, FM = C::FM, FN = C::FN, ESZ = C::ESZ, APITCH = C::APITCH, BROW = C::BROW;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsA0 = smem;
    float* epi = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wn = tid >> 6, wm = 0;
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
    int w_off[FN];
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
        const int row = cout0 + wn * (BN / WN) + fn * 16 + l15;
        w_off[fn] = (row < p.Cout_w) ? row * BROW + lg * 8 * ESZ : -1;
    }
    typename M::frag wf[RING][FN];
    auto W_load = [&](int blk, typename M::frag (&dst)[FN]) {
        const char* wb = wimg + (size_t)blk * wblock_bytes;
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) {
            typename M::frag z = {};
            dst[fn] = (w_off[fn] >= 0 && blk < total_steps) ? *reinterpret_cast<const typename M::frag*>(wb + w_off[fn]) : z;
        }
    };
#pragma unroll
    for (int j = 0; j < RING; ++j) W_load(j, wf[j]);

    // (segment, chunk, tap) state of the current step
    int seg = 0, q = 0, tap = 0, acur = 0;
    mcgen_seg_t sg = p.seg[0];
    int nchunk = (sg.C + MCGEN_CK - 1) / MCGEN_CK, ntap = sg.ksize * sg.ksize;
    int PC = W + 2 * (sg.ksize >> 1);
    PatchStager<T, NT, C::NI, APITCH> stager;
    typename PatchStager<T, NT, C::NI, APITCH>::raw_t araw;
    int a_base[FM];
    auto enter_segment = [&]() {
        const int halo = sg.ksize >> 1;
        const int PR = g.TH + 2 * halo;
        PC = W + 2 * halo;
        stager.setup(sg, g, N, H, W, tid);
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int m = fm * 16 + l15;
            const int ti = m >> g.lgTHW, rem = m & ((1 << g.lgTHW) - 1);
            const int r = rem >> g.lgW, c = rem & (W - 1);
            a_base[fm] = ((ti * PR + r) * PC + c) * APITCH + lg * 8 * ESZ;
        }
        acur ^= 1;                                   // idle buffer: last read one chunk (a barrier) ago
        stager.stage(sg, 0, ldsA0 + acur * a_bytes);
        __syncthreads();
    };
    enter_segment();

#pragma unroll 1
    for (int base = 0; base < total_steps; base += RING) {
#pragma unroll
        for (int j = 0; j < RING; ++j) {
            const int blk = base + j;
            if (blk < total_steps) {
                const bool more = (q + 1 < nchunk);
                if (tap == 0 && more) stager.load(sg, (q + 1) * MCGEN_CK, araw);
                const int kh = (ntap == 9) ? tap / 3 : 0, kw = (ntap == 9) ? tap % 3 : 0;
                const char* ldsA = ldsA0 + acur * a_bytes + (kh * PC + kw) * APITCH;
#pragma unroll
                for (int fm = 0; fm < FM; ++fm) {
                    const typename M::frag af = *reinterpret_cast<const typename M::frag*>(ldsA + a_base[fm]);
#pragma unroll
                    for (int fn = 0; fn < FN; ++fn) M::run(wf[j][fn], af, acc[fn][fm]);
                }
                W_load(blk + RING, wf[j]);           // same slot, RING steps ahead
                if (tap == ntap - 1) {
                    if (more) {
                        stager.write(sg, (q + 1) * MCGEN_CK, araw, ldsA0 + (acur ^ 1) * a_bytes);
                        __syncthreads();
                        acur ^= 1; ++q; tap = 0;
                    } else if (seg + 1 < p.nseg) {
                        ++seg; sg = p.seg[seg];
                        nchunk = (sg.C + MCGEN_CK - 1) / MCGEN_CK; ntap = sg.ksize * sg.ksize;
                        q = 0; tap = 0;
                        enter_segment();
                    }
                } else {
                    ++tap;
                }
            }
        }
    }
    __syncthreads();
    conv_epilogue<T, C, BM, BN, WM, WN>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0);
}

// ---- "dma" form ---------------------------------------------------------------------------------
// The two-barrier form with the weight tiles moved by LDS-DMA (global_load_lds, no VGPR round trip) into
// a ring of RB tap slots, DIST taps ahead of their use: per tap ONE raw s_barrier behind a counted vmcnt.
// The DMA writes LDS linearly (wave base + lane*16), so the bank swizzle is applied to the SOURCE address.
template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN)
void conv_dma_kernel(const mcgen_conv_t p, const int a_bytes) {
    using C = ConvCfg<T, BM, BN, WM, WN>;
    using M = Mma<T>;
    constexpr int NT = C::NT, FM = C::FM, FN = C::FN, ESZ = C::ESZ, APITCH = C::APITCH, BROW = C::BROW;
    constexpr int RB = 3, DIST = 2;                        // ring slots, prefetch distance (taps)
    constexpr int NW = WM * WN;
    constexpr int KB = C::BBYTES / 1024;                   // 1 KB DMA pieces per weight tile
    constexpr int PPW = (KB + NW - 1) / NW;                // pieces per wave per tap
    static_assert(DIST == 2 && PPW <= 2, "vmcnt immediates below assume DIST 2 and <= 2 pieces per wave");

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
        const mcgen_seg_t sg = p.seg[s];
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
    conv_epilogue<T, C, BM, BN, WM, WN>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0);
}

// ---- "dma3" form: as "dma", but THREE taps per barrier ------------------------------------------------
// A ring slot holds the weight tiles of a group of up to 3 taps (one kernel row of a 3x3 filter); the
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
__global__ __launch_bounds__(64 * WM * WN, (MCGEN_BIG_WAVES && BM * BN >= 256 * 256) ? MCGEN_BIG_WAVES : 0)
void conv_dma3_kernel(const mcgen_conv_t p, const int a_bytes) {
    using C = ConvCfg<T, BM, BN, WM, WN>;
    using M = Mma<T>;
    constexpr int NT = C::NT, FM = C::FM, FN = C::FN, ESZ = C::ESZ, APITCH = C::APITCH, BROW = C::BROW;
    constexpr int TPS = 3;
    // 128-pixel tiles only: the 256x256 tile has no registers to spare (128 accumulators), 64-pixel tiles have 2 items per thread
    constexpr bool STAGE2 = MCGEN_STAGE2 && (BM == 128 || (MCGEN_STAGE2_SMALL && BM < 128) || (MCGEN_STAGE2_BIG && BM * BN >= 256 * 256));
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
    const int gpc0 = (nt0 == 9) ? 3 : 1;                   // groups per chunk
    const int G0 = nc0 * gpc0, S0 = nc0 * nt0;
    int nt1 = 1, nc1 = 0, gpc1 = 1;
    if (p.nseg > 1) { nt1 = p.seg[1].ksize * p.seg[1].ksize; nc1 = (p.seg[1].C + MCGEN_CK - 1) / MCGEN_CK; gpc1 = (nt1 == 9) ? 3 : 1; }
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
        if (gi < G0) { ntg = (nt0 == 9) ? 3 : 1; blk0 = gi * ntg; }
        else { const int gj = gi - G0; ntg = (nt1 == 9) ? 3 : 1; blk0 = S0 + gj * ntg; }
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
        const mcgen_seg_t sg = p.seg[s];
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
        const int gpc = (ntap == 9) ? 3 : 1, ntg = (ntap == 9) ? 3 : 1;
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
    conv_epilogue<T, C, BM, BN, WM, WN>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0);
}

// ---- "dma3g" form: dma3 for PURE 1x1 launches, three chunks per barrier round ------------------------------------
// A 1x1 convolution has one tap per 32-channel chunk, so dma3 spends a barrier round (window staging + wait) per MFMA
// step; here the windows of three consecutive chunks are staged side by side and their three weight tiles (which are
// consecutive in the image) form one DMA group.  A separate kernel: sharing dma3's code cost the 3x3 launches 1-3 %.
template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN)
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
        const mcgen_seg_t sg = p.seg[s];
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
    conv_epilogue<T, C, BM, BN, WM, WN>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0);
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
        const mcgen_seg_t sg = p.seg[s];
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
    conv_epilogue<T, C, BM, BN, WM, WN>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0);
}

// ---- "res" form: whole input window resident --------------------------------------------------------------
// For layers with few pixels (8x8, 16x16 maps) a workgroup's MFMA time is a few microseconds, so every
// exposed global round trip shows.  Here ALL channels of the tile's input window are staged once (all
// loads in flight together), the weight tiles stream through a 3-slot LDS-DMA ring of 3-tap groups two
// groups ahead, and the main loop is barrier + MFMAs only.  LDS: window [pixel][C] at pitch 2C+32 bytes.
template <typename T, int BM, int BN, int WM, int WN, int NQ, int RB>
__global__ __launch_bounds__(64 * WM * WN)
void conv_res_kernel(const mcgen_conv_t p, const int a_bytes) {
    using C = ConvCfg<T, BM, BN, WM, WN>;
    using M = Mma<T>;
    constexpr int NT = C::NT, FM = C::FM, FN = C::FN, ESZ = C::ESZ, BROW = C::BROW;
    constexpr int TPS = 3, DIST = RB - 1;                  // ring slots; groups in flight ahead of the consumer
    constexpr int NW = WM * WN;
    constexpr int KB = C::BBYTES / 1024;
    constexpr int PPW = (KB + NW - 1) / NW;
    constexpr int SLOT = TPS * C::BBYTES;
    constexpr int NIR = (BM * 9 + NT - 1) / NT;            // staging items per thread per 32-channel chunk
    static_assert(PPW * TPS * (DIST - 1) <= 63, "vmcnt immediate");

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
    const int nt0 = p.seg[0].ksize * p.seg[0].ksize, nc0 = (p.seg[0].C + MCGEN_CK - 1) / MCGEN_CK;
    const int G0 = nc0 * ((nt0 == 9) ? 3 : 1), S0 = nc0 * nt0;
    int nt1 = 1, nc1 = 0;
    if (p.nseg > 1) { nt1 = p.seg[1].ksize * p.seg[1].ksize; nc1 = (p.seg[1].C + MCGEN_CK - 1) / MCGEN_CK; }
    const int GT = G0 + nc1 * ((nt1 == 9) ? 3 : 1);

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
    auto G_dma = [&](int gi) {
        if (gi >= GT) return;
        int blk0, ntg;
        if (gi < G0) { ntg = (nt0 == 9) ? 3 : 1; blk0 = gi * ntg; }
        else { const int gj = gi - G0; ntg = (nt1 == 9) ? 3 : 1; blk0 = S0 + gj * ntg; }
        char* slot = ldsB0 + (gi % RB) * SLOT;
#pragma unroll
        for (int t = 0; t < TPS; ++t) {
            const int blk = blk0 + (t < ntg ? t : ntg - 1);
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
#pragma unroll
    for (int j = 0; j < DIST; ++j) G_dma(j);
    for (int s = 0; s < p.nseg; ++s) {
        const mcgen_seg_t sg = p.seg[s];
        const int halo = sg.ksize >> 1;
        const int PR = g.TH + 2 * halo, PC = W + 2 * halo;
        const int nchunk = (sg.C + MCGEN_CK - 1) / MCGEN_CK;
        const int apitch = nchunk * MCGEN_CK * ESZ + 16 * ESZ;       // bytes per window pixel: all channels + pad
        // ---- stage the whole window: all chunks' loads first, then prologue + LDS stores -----------------
        {
            // the stager works on 32-channel chunks with its own (compile-time) pitch; here the pitch is a run
            // time value, so the LDS offsets are recomputed: pixel index = it_lds / CHUNK_PITCH
            constexpr int CP = MCGEN_CK * ESZ + 16 * ESZ;
            PatchStager<T, NT, NIR, CP> stager;
            stager.setup(sg, g, N, H, W, tid);
            typename PatchStager<T, NT, NIR, CP>::raw_t raw[NQ];
            if (s > 0) __builtin_amdgcn_s_barrier();       // previous segment's window reads are done
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (q < nchunk) stager.load(sg, q * MCGEN_CK, raw[q]);
            // re-base the LDS offsets to the resident layout
#pragma unroll
            for (int k = 0; k < NIR; ++k)
                if (stager.it_lds[k] >= 0) {
                    const int pp = stager.it_lds[k] / CP, sub = stager.it_lds[k] % CP;
                    stager.it_lds[k] = pp * apitch + sub;
                }
#pragma unroll
            for (int q = 0; q < NQ; ++q)
                if (q < nchunk) stager.write(sg, q * MCGEN_CK, raw[q], ldsA + q * MCGEN_CK * ESZ);
        }
        int a_base[FM];
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int m = wm * (BM / WM) + fm * 16 + l15;
            const int ti = m >> g.lgTHW, rem = m & ((1 << g.lgTHW) - 1);
            const int r = rem >> g.lgW, c = rem & (W - 1);
            a_base[fm] = ((ti * PR + r) * PC + c) * apitch + lg * 8 * ESZ;
        }
        const int ntap = sg.ksize * sg.ksize;
        const int gpc = (ntap == 9) ? 3 : 1, ntg = (ntap == 9) ? 3 : 1;
#pragma unroll 1
        for (int q = 0; q < nchunk; ++q) {
#pragma unroll 1
            for (int gq = 0; gq < gpc; ++gq) {
                // group gi landed (this wave's pieces; gi+1 may still be in flight), then everyone's + the window
                // DIST-1 newer groups are in flight behind group gi (fewer near the end: then drain)
                if (gi + DIST - 1 < GT) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PPW * TPS * (DIST - 1)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                G_dma(gi + DIST);                         // slot (gi+DIST)%RB == (gi-1)%RB: its readers passed the barrier
                const char* slot = ldsB0 + (gi % RB) * SLOT;
                const char* ldsAq = ldsA + q * MCGEN_CK * ESZ;
#pragma unroll
                for (int t = 0; t < TPS; ++t) {
                    if (t < ntg) {
                        const int tap = gq * ntg + t;
                        const int kh = (ntap == 9) ? tap / 3 : 0, kw = (ntap == 9) ? tap % 3 : 0;
                        const int tapoff = (kh * PC + kw) * apitch;
                        const char* ldsB = slot + t * C::BBYTES;
                        typename M::frag af[FM], wf[FN];
#pragma unroll
                        for (int fm = 0; fm < FM; ++fm)
                            af[fm] = *reinterpret_cast<const typename M::frag*>(ldsAq + a_base[fm] + tapoff);
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
    conv_epilogue<T, C, BM, BN, WM, WN>(p, g, acc, epi, tid, wm, wn, l15, lg, tile_m, cout0);
}

// ---- host side ----------------------------------------------------------------------------------
struct TilePick { int BM, BN, pipe; };

// Output tile: widest channel tile the layer fills, then the largest pixel tile that still gives
// every CU a workgroup (256 CUs); fp32 (parity build) is limited by LDS to the two small tiles.
// MCGEN_CONV_CFG="BM,BN,PIPE" overrides the choice for bf16 launches with Cout_w > 16 (tuning runs).
static TilePick pick_tile(const mcgen_conv_t* p, int dtype) {
    const long M = (long)p->N * p->H * p->W;
    if (p->Cout_w <= 16) {
        // skinny-N convolutions (Glow's ZeroConv2d and the input gradients of the coupling nets, image heads) are a
        // serial chain over K: with few pixels, 64-pixel tiles double the workgroups that overlap each other's
        // staging latency (measured: MCGlow step -5 %); large maps keep 128 (MCGAN's 32x32 head)
        int bm16 = (M <= 32768) ? 64 : 128;
        if (const char* e = getenv("MCGEN_CONV_BM16")) bm16 = atoi(e);
        // very large maps (MCGAN's 32x32 image head at batch 128: 1024 tiles of 128 pixels, three resident per CU = two
        // rounds): 256-pixel tiles run as one round (dma3 form, 8 waves)
        static const long big16 = getenv("MCGEN_CONV_BIG16") ? atol(getenv("MCGEN_CONV_BIG16")) : 131072;
        if (dtype == MCGEN_BF16 && !getenv("MCGEN_CONV_BM16") && big16 > 0 && M >= big16 && 256 >= 2 * p->W) return {256, 16, 5};
        const int HW16 = p->H * p->W;
        // the chunk-pipelined form (12) wins on these K-deep, latency-bound launches (-15..20 %); elsewhere the extra LDS of
        // its two-step weight ring costs more occupancy than the prefetch gains (measured), so dma3 stays
        static const int mode16 = getenv("MCGEN_CONV_MODE16") ? atoi(getenv("MCGEN_CONV_MODE16")) : 12;
        if (dtype == MCGEN_BF16 && bm16 != 128 && ((bm16 >= 2 * p->W) || HW16 <= bm16)) return {bm16, 16, mode16};
        return {128, 16, dtype == MCGEN_BF16 ? mode16 : 0};
    }
    if (dtype == MCGEN_F32) return (M <= 16384 || p->Cout_w <= 64) ? TilePick{64, 64, 0} : TilePick{128, 128, 0};
    int env_bm = 0, env_bn = 0, env_pipe = 0;
    if (const char* e = getenv("MCGEN_CONV_CFG")) {
        if (sscanf(e, "%d,%d,%d", &env_bm, &env_bn, &env_pipe) != 3) env_bm = 0;
    }
    const int HW = p->H * p->W;
    if (env_bm > 0 && ((env_bm >= 2 * p->W) || HW <= env_bm)) return {env_bm, env_bn, env_pipe};
    // measured on MI355X (tools/bench_conv.py, profiles/): the LDS-DMA weight ring with three taps per
    // barrier ("dma3" form, mode 5) wins on every shape; big tiles only where there are enough pixels to
    // fill 256 CUs
    const bool rows256 = (256 >= 2 * p->W) || (HW <= 256), rows128 = (128 >= 2 * p->W) || (HW <= 128);
    static const int m128 = getenv("MCGEN_CONV_M128") ? atoi(getenv("MCGEN_CONV_M128")) : 5;
    static const int m128w = getenv("MCGEN_CONV_M128W") ? atoi(getenv("MCGEN_CONV_M128W")) : 5;
    if (M >= 65536 && rows256 && p->Cout_w > 128) return {256, 256, 5};
    static const long big128 = getenv("MCGEN_CONV_BIG128") ? atol(getenv("MCGEN_CONV_BIG128")) : 0;
    if (big128 > 0 && M >= big128 && rows256 && p->Cout_w > 64) return {256, 128, 5};
    if (M >= 65536 && rows128 && p->Cout_w > 64) return {128, 128, m128};
    if (M >= 32768 && rows128 && p->Cout_w > 128) return {128, 256, m128w};
    static const long t64x128 = getenv("MCGEN_CONV_T64X128") ? atol(getenv("MCGEN_CONV_T64X128")) : 32768;
    if (M >= t64x128 && p->Cout_w > 64) return {64, 128, 5};
    // 64x64 tile on 8 waves (4 x 2): ~15 % faster than 4 waves on 8x8 maps, bit-identical outputs.  (Its BatchNorm partial
    // sums round differently in the last bit, which once looked like a defect: the bf16 full-size digest run is bimodal
    // in its second-iteration G loss -- 1.81 or 1.70 -- under ANY 1e-7 nudge of the batch sums, see tools/digest_probe.py
    // with MCGEN_BN_PERTURB.)
    int small_mode = 11;
    if (const char* e = getenv("MCGEN_CONV_SMALL")) small_mode = atoi(e);       // tuning override for the 64x64 fallback
    return {64, 64, small_mode};
}

static int patch_pixels(const mcgen_conv_t* p, int BM) {
    int best = 0;
    for (int s = 0; s < p->nseg; ++s) {
        const int pp = mcgen_patch_pixels(BM, p->H, p->W, p->seg[s].ksize);
        if (pp > best) best = pp;
    }
    return best;
}

template <typename T, int BM, int BN, int WM, int WN, bool PIPE>
static int launch_cfg(const mcgen_conv_t* p, hipStream_t st) {
    using C = ConvCfg<T, BM, BN, WM, WN>;
    const long Mtot = (long)p->N * p->H * p->W;
    const int mt = (int)((Mtot + BM - 1) / BM);
    const int nt = (p->Cout_w + BN - 1) / BN;
    const int PP = patch_pixels(p, BM);
    MCGEN_CHECK(PP * 4 <= C::NI * C::NT, "conv_fused: patch of %d pixels exceeds the staging plan", PP);
    int a_bytes = round_up(PP * C::APITCH, 32);
    int main_bytes = (PIPE ? 2 : 1) * (a_bytes + C::BBYTES);
    int epi_bytes = C::PPX * C::EP * 4;
    int red_bytes = C::PROWS * BN * 2 * 4;
    int lds = main_bytes > epi_bytes ? main_bytes : epi_bytes;
    if (red_bytes > lds) lds = red_bytes;
    MCGEN_CHECK(lds <= 160 * 1024, "conv_fused: tile %dx%d needs %d bytes of LDS", BM, BN, lds);
    auto kern = conv_fused_kernel<T, BM, BN, WM, WN, PIPE>;
    static int raised = 0;
    if (lds > 64 * 1024 && lds > raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return mcgen_fail("conv_fused: cannot raise LDS limit to %d: %s", lds, hipGetErrorString(e));
        raised = lds;
    }
    hipLaunchKernelGGL(kern, dim3(mt, nt), dim3(C::NT), lds, st, *p, a_bytes);
    MCGEN_LAUNCH_CHECK("conv_fused");
    return 0;
}

template <typename T, int BM, int BN, bool RINGED>
static int launch_direct(const mcgen_conv_t* p, hipStream_t st) {
    using C = ConvCfg<T, BM, BN, 1, 4>;
    const long Mtot = (long)p->N * p->H * p->W;
    const int mt = (int)((Mtot + BM - 1) / BM);
    const int nt = (p->Cout_w + BN - 1) / BN;
    const int PP = patch_pixels(p, BM);
    MCGEN_CHECK(PP * 4 <= C::NI * C::NT, "conv_fused: patch of %d pixels exceeds the staging plan", PP);
    const int a_bytes = round_up(PP * C::APITCH, 32);
    int lds = 2 * a_bytes;
    const int epi_bytes = C::PPX * C::EP * 4, red_bytes = C::PROWS * BN * 2 * 4;
    if (epi_bytes > lds) lds = epi_bytes;
    if (red_bytes > lds) lds = red_bytes;
    MCGEN_CHECK(lds <= 160 * 1024, "conv_fused: tile %dx%d needs %d bytes of LDS", BM, BN, lds);
    auto kern = RINGED ? conv_ring_kernel<T, BM, BN> : conv_direct_kernel<T, BM, BN>;
    static int raised = 0;
    if (lds > 64 * 1024 && lds > raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return mcgen_fail("conv_fused: cannot raise LDS limit to %d: %s", lds, hipGetErrorString(e));
        raised = lds;
    }
    hipLaunchKernelGGL(kern, dim3(mt, nt), dim3(C::NT), lds, st, *p, a_bytes);
    MCGEN_LAUNCH_CHECK("conv_fused(direct)");
    return 0;
}

template <typename T, int BM, int BN, int WM, int WN, int TPS = 1>
static int launch_dma(const mcgen_conv_t* p, hipStream_t st) {
    using C = ConvCfg<T, BM, BN, WM, WN>;
    const long Mtot = (long)p->N * p->H * p->W;
    const int mt = (int)((Mtot + BM - 1) / BM);
    const int nt = (p->Cout_w + BN - 1) / BN;
    const int PP = patch_pixels(p, BM);
    MCGEN_CHECK(PP * 4 <= C::NI * C::NT, "conv_fused: patch of %d pixels exceeds the staging plan", PP);
    int a_bytes = round_up(PP * C::APITCH, 1024);
    // pure 1x1 launches with several chunks: the grouped form, when its three side-by-side windows fit the LDS budget
    static const int g1_env = getenv("MCGEN_CONV_G1") ? atoi(getenv("MCGEN_CONV_G1")) : 3;
    const int a3 = round_up(3 * BM * C::APITCH, 1024);
    // (K-deep ones only: at 8 chunks and fewer the plain form measured as fast or faster)
    const bool grouped = TPS == 3 && g1_env == 3 && p->nseg == 1 && p->seg[0].ksize == 1 && p->seg[0].C >= 12 * MCGEN_CK &&
                         (a3 > a_bytes ? a3 : a_bytes) + 6 * C::BBYTES <= 96 * 1024;
    if (grouped && a3 > a_bytes) a_bytes = a3;
    int lds = a_bytes + (TPS == 3 ? 6 : 3) * C::BBYTES;
    const int epi_bytes = C::PPX * C::EP * 4, red_bytes = C::PROWS * BN * 2 * 4;
    if (epi_bytes > lds) lds = epi_bytes;
    if (red_bytes > lds) lds = red_bytes;
    MCGEN_CHECK(lds <= 160 * 1024, "conv_fused: tile %dx%d needs %d bytes of LDS", BM, BN, lds);
    if (grouped) {
        auto kg = conv_dma3g_kernel<T, BM, BN, WM, WN>;
        static int raisedg = 0;
        if (lds > 64 * 1024 && lds > raisedg) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kg), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return mcgen_fail("conv_fused: cannot raise LDS limit to %d: %s", lds, hipGetErrorString(e));
            raisedg = lds;
        }
        hipLaunchKernelGGL(kg, dim3(mt, nt), dim3(C::NT), lds, st, *p, a_bytes);
        MCGEN_LAUNCH_CHECK("conv_fused(dma3g)");
        return 0;
    }
    auto kern = (TPS == 3) ? conv_dma3_kernel<T, BM, BN, WM, WN> : conv_dma_kernel<T, BM, BN, WM, WN>;
    static int raised = 0;
    if (lds > 64 * 1024 && lds > raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return mcgen_fail("conv_fused: cannot raise LDS limit to %d: %s", lds, hipGetErrorString(e));
        raised = lds;
    }
    hipLaunchKernelGGL(kern, dim3(mt, nt), dim3(C::NT), lds, st, *p, a_bytes);
    MCGEN_LAUNCH_CHECK("conv_fused(dma)");
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
    const int epi_bytes = C::PPX * C::EP * 4, red_bytes = C::PROWS * BN * 2 * 4;
    if (epi_bytes > lds) lds = epi_bytes;
    if (red_bytes > lds) lds = red_bytes;
    MCGEN_CHECK(lds <= 160 * 1024, "conv_fused(cp): tile %dx%d needs %d bytes of LDS", BM, BN, lds);
    auto kern = conv_cp_kernel<T, BM, BN, WM, WN>;
    static int raised = 0;
    if (lds > 64 * 1024 && lds > raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return mcgen_fail("conv_fused: cannot raise LDS limit to %d: %s", lds, hipGetErrorString(e));
        raised = lds;
    }
    hipLaunchKernelGGL(kern, dim3(mt, nt), dim3(C::NT), lds, st, *p, a_bytes, subw);
    MCGEN_LAUNCH_CHECK("conv_fused(cp)");
    return 0;
}

template <typename T, int BM, int BN, int WM, int WN, int NQ, int RB>
static int launch_res(const mcgen_conv_t* p, hipStream_t st) {
    using C = ConvCfg<T, BM, BN, WM, WN>;
    const long Mtot = (long)p->N * p->H * p->W;
    const int mt = (int)((Mtot + BM - 1) / BM);
    const int nt = (p->Cout_w + BN - 1) / BN;
    const int PP = patch_pixels(p, BM);
    MCGEN_CHECK(PP * 4 <= ((BM * 9 + C::NT - 1) / C::NT) * C::NT, "conv_fused: patch of %d pixels exceeds the staging plan", PP);
    int cmax = 0;
    for (int s = 0; s < p->nseg; ++s) if (p->seg[s].C > cmax) cmax = p->seg[s].C;
    const int nchunk = (cmax + MCGEN_CK - 1) / MCGEN_CK;
    MCGEN_CHECK(nchunk <= NQ, "conv_fused(res): %d channels exceed the resident window plan", cmax);
    const int apitch = nchunk * MCGEN_CK * C::ESZ + 16 * C::ESZ;
    const int a_bytes = round_up(PP * apitch, 1024);
    int lds = a_bytes + RB * 3 * C::BBYTES;
    const int epi_bytes = C::PPX * C::EP * 4, red_bytes = C::PROWS * BN * 2 * 4;
    if (epi_bytes > lds) lds = epi_bytes;
    if (red_bytes > lds) lds = red_bytes;
    MCGEN_CHECK(lds <= 160 * 1024, "conv_fused(res): tile %dx%d with %d channels needs %d bytes of LDS", BM, BN, cmax, lds);
    auto kern = conv_res_kernel<T, BM, BN, WM, WN, NQ, RB>;
    static int raised = 0;
    if (lds > 64 * 1024 && lds > raised) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return mcgen_fail("conv_fused: cannot raise LDS limit to %d: %s", lds, hipGetErrorString(e));
        raised = lds;
    }
    hipLaunchKernelGGL(kern, dim3(mt, nt), dim3(C::NT), lds, st, *p, a_bytes);
    MCGEN_LAUNCH_CHECK("conv_fused(res)");
    return 0;
}

typedef int (*launch_fn)(const mcgen_conv_t*, hipStream_t);
struct CfgEntry { int BM, BN, pipe; launch_fn fn; };

template <typename T>
static const CfgEntry* small_table(int* n) {
    static const CfgEntry t[] = {
        {128, 16, 0, launch_cfg<T, 128, 16, 4, 1, false>},
        {64, 64, 0, launch_cfg<T, 64, 64, 2, 2, false>},
        {128, 128, 0, launch_cfg<T, 128, 128, 2, 2, false>},
    };
    *n = 3;
    return t;
}
static const CfgEntry* bf16_table(int* n) {
    using T = bf16_t;
    static const CfgEntry t[] = {
        {256, 256, 1, launch_cfg<T, 256, 256, 2, 4, true>},  {256, 256, 0, launch_cfg<T, 256, 256, 2, 4, false>},
        {128, 256, 1, launch_cfg<T, 128, 256, 2, 4, true>},  {128, 256, 0, launch_cfg<T, 128, 256, 2, 4, false>},
        {64, 256, 1, launch_cfg<T, 64, 256, 1, 4, true>},    {64, 256, 0, launch_cfg<T, 64, 256, 1, 4, false>},
        {256, 128, 1, launch_cfg<T, 256, 128, 4, 2, true>},  {256, 128, 0, launch_cfg<T, 256, 128, 4, 2, false>},
        {128, 128, 1, launch_cfg<T, 128, 128, 2, 2, true>},  {128, 128, 0, launch_cfg<T, 128, 128, 2, 2, false>},
        {64, 128, 1, launch_cfg<T, 64, 128, 2, 2, true>},    {64, 128, 0, launch_cfg<T, 64, 128, 2, 2, false>},
        {128, 64, 1, launch_cfg<T, 128, 64, 2, 2, true>},    {128, 64, 0, launch_cfg<T, 128, 64, 2, 2, false>},
        {64, 64, 1, launch_cfg<T, 64, 64, 2, 2, true>},      {64, 64, 0, launch_cfg<T, 64, 64, 2, 2, false>},
        {128, 16, 0, launch_cfg<T, 128, 16, 4, 1, false>},
        {128, 256, 2, launch_direct<T, 128, 256, false>}, {128, 128, 2, launch_direct<T, 128, 128, false>},
        {64, 256, 2, launch_direct<T, 64, 256, false>},   {64, 128, 2, launch_direct<T, 64, 128, false>},
        {128, 64, 2, launch_direct<T, 128, 64, false>},   {64, 64, 2, launch_direct<T, 64, 64, false>},
        {256, 128, 2, launch_direct<T, 256, 128, false>}, {256, 64, 2, launch_direct<T, 256, 64, false>},
        {256, 256, 4, launch_dma<T, 256, 256, 2, 4>}, {128, 256, 4, launch_dma<T, 128, 256, 2, 4>},
        {256, 128, 4, launch_dma<T, 256, 128, 4, 2>}, {128, 128, 4, launch_dma<T, 128, 128, 2, 2>},
        {64, 128, 4, launch_dma<T, 64, 128, 2, 2>},   {64, 64, 4, launch_dma<T, 64, 64, 2, 2>},
        {256, 256, 5, launch_dma<T, 256, 256, 2, 4, 3>}, {128, 256, 5, launch_dma<T, 128, 256, 2, 4, 3>},
        {256, 128, 5, launch_dma<T, 256, 128, 4, 2, 3>}, {128, 128, 5, launch_dma<T, 128, 128, 2, 2, 3>},
        {64, 128, 5, launch_dma<T, 64, 128, 2, 2, 3>},   {64, 64, 5, launch_dma<T, 64, 64, 2, 2, 3>},
        {128, 16, 5, launch_dma<T, 128, 16, 4, 1, 3>},
        {32, 64, 5, launch_dma<T, 32, 64, 1, 2, 3>},   {32, 128, 5, launch_dma<T, 32, 128, 1, 4, 3>},
        {32, 64, 8, launch_dma<T, 32, 64, 2, 2, 3>},   {64, 32, 5, launch_dma<T, 64, 32, 2, 1, 3>},
        {64, 64, 12, launch_cp<T, 64, 64, 2, 2>},
        {64, 16, 12, launch_cp<T, 64, 16, 4, 1>},      {128, 16, 12, launch_cp<T, 128, 16, 4, 1>},
        {32, 64, 12, launch_cp<T, 32, 64, 1, 2>},
        {64, 16, 5, launch_dma<T, 64, 16, 4, 1, 3>},   {32, 16, 5, launch_dma<T, 32, 16, 2, 1, 3>},
        {64, 16, 9, launch_dma<T, 64, 16, 2, 1, 3>},   {256, 16, 5, launch_dma<T, 256, 16, 8, 1, 3>},
        {64, 64, 9, launch_dma<T, 64, 64, 4, 4, 3>},   {64, 64, 10, launch_dma<T, 64, 64, 2, 4, 3>},
        {64, 64, 11, launch_dma<T, 64, 64, 4, 2, 3>},  {64, 128, 9, launch_dma<T, 64, 128, 4, 4, 3>},
        {64, 128, 10, launch_dma<T, 64, 128, 2, 4, 3>},
        {64, 64, 6, launch_res<T, 64, 64, 2, 2, 8, 3>},   {64, 128, 6, launch_res<T, 64, 128, 2, 2, 8, 3>},
        {128, 128, 6, launch_res<T, 128, 128, 2, 2, 8, 3>}, {128, 16, 6, launch_res<T, 128, 16, 4, 1, 8, 3>},
        {64, 64, 7, launch_res<T, 64, 64, 2, 2, 8, 8>},   {64, 128, 7, launch_res<T, 64, 128, 2, 2, 4, 4>},
        {128, 16, 7, launch_res<T, 128, 16, 4, 1, 8, 8>},
        {128, 256, 3, launch_direct<T, 128, 256, true>}, {128, 128, 3, launch_direct<T, 128, 128, true>},
        {64, 256, 3, launch_direct<T, 64, 256, true>},   {64, 128, 3, launch_direct<T, 64, 128, true>},
        {128, 64, 3, launch_direct<T, 128, 64, true>},   {64, 64, 3, launch_direct<T, 64, 64, true>},
    };
    *n = (int)(sizeof(t) / sizeof(t[0]));
    return t;
}

static int dispatch(const mcgen_conv_t* p, int dtype, const TilePick& t, hipStream_t st) {
    int n = 0;
    const CfgEntry* tab = dtype == MCGEN_BF16 ? bf16_table(&n) : small_table<float>(&n);
    for (int i = 0; i < n; ++i)
        if (tab[i].BM == t.BM && tab[i].BN == t.BN && tab[i].pipe == t.pipe) return tab[i].fn(p, st);
    return mcgen_fail("conv_fused: no instantiation for tile %dx%d pipe=%d dtype=%d", t.BM, t.BN, t.pipe, dtype);
}

static int validate(const mcgen_conv_t* p) {
    MCGEN_CHECK(p && p->nseg >= 1 && p->nseg <= 2, "conv_fused: nseg must be 1 or 2");
    MCGEN_CHECK(p->N > 0 && ilog2_exact(p->H) >= 0 && ilog2_exact(p->W) >= 0, "conv_fused: H and W must be powers of two (got %dx%d)", p->H, p->W);
    MCGEN_CHECK(p->W <= 64 && p->H * p->W >= 1, "conv_fused: W up to 64 supported");
    MCGEN_CHECK(p->Cout > 0 && p->Cout_w == round_up(p->Cout, 16), "conv_fused: Cout_w must be Cout rounded up to 16");
    MCGEN_CHECK(p->Cy % 8 == 0 && p->Cy >= p->Cout, "conv_fused: Cy must be a multiple of 8 and >= Cout");
    MCGEN_CHECK(p->w && p->y, "conv_fused: null weight image or output");
    for (int s = 0; s < p->nseg; ++s) {
        const mcgen_seg_t& g = p->seg[s];
        MCGEN_CHECK(g.x && g.C > 0 && g.C % 8 == 0, "conv_fused: segment %d: C must be a positive multiple of 8", s);
        MCGEN_CHECK(g.ksize == 1 || g.ksize == 3, "conv_fused: segment %d: ksize must be 1 or 3", s);
        MCGEN_CHECK(!g.ups || (p->H >= 2 && p->W >= 2), "conv_fused: upsampled segment needs H, W >= 2");
    }
    if (p->pool) MCGEN_CHECK(p->H >= 2 && p->W >= 2, "conv_fused: pooling needs H, W >= 2");
    MCGEN_CHECK(p->stats_mode >= 0 && p->stats_mode <= 2, "conv_fused: bad stats_mode");
    MCGEN_CHECK(p->stats_mode != 2 || (p->gate_x && p->gmean && p->grstd), "conv_fused: stats_mode 2 needs gate_x, gmean, grstd");
    MCGEN_CHECK(p->stats_mode == 0 || p->stats, "conv_fused: stats_mode set without a stats buffer");
    return 0;
}

}  // namespace

extern "C" int mcgen_conv_m_tiles(const mcgen_conv_t* p, int dtype) {
    if (!p) return 0;
    const TilePick t = pick_tile(p, dtype);
    const long Mtot = (long)p->N * p->H * p->W;
    return (int)((Mtot + t.BM - 1) / t.BM);
}

extern "C" int mcgen_conv_tile(const mcgen_conv_t* p, int dtype, int* bm, int* bn) {
    if (!p || !bm || !bn) return mcgen_fail("conv_tile: null pointer");
    const TilePick t = pick_tile(p, dtype);
    *bm = t.BM; *bn = t.BN;
    return 0;
}

extern "C" int mcgen_conv_fused(const mcgen_conv_t* p, int dtype, void* stream) {
    if (int rc = validate(p)) return rc;
    const TilePick t = pick_tile(p, dtype);
    // pooling / whole-row tiles need at least two rows per tile
    MCGEN_CHECK(t.BM >= 2 * p->W || p->H * p->W <= t.BM, "conv_fused: tile of %d pixels too small for W=%d", t.BM, p->W);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype != MCGEN_F32 && dtype != MCGEN_BF16) return mcgen_fail("conv_fused: unknown dtype %d", dtype);
    return dispatch(p, dtype, t, st);
}
