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

namespace {

template <typename T, int BM, int BN, int WM, int WN>
struct ConvCfg {
    static constexpr int NT = 64 * WM * WN;
    static constexpr int FM = BM / WM / 16;        // pixel fragments per wave
    static constexpr int FN = BN / WN / 16;        // cout fragments per wave
    static constexpr int ESZ = Elem<T>::BYTES;
    static constexpr int APITCH = MCGEN_CK * ESZ + 16 * ESZ;   // bf16: 96 B (conflict-free b128 reads)
    static constexpr int BROW = MCGEN_CK * ESZ;                // bytes per weight row in LDS
    static constexpr int EP = BN + 4;                          // epilogue pitch in floats
    static constexpr int NI = (BM * 9 + NT - 1) / NT;          // staging items per thread: PP*4 <= BM*2.25*4
};

template <typename T, int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(64 * WM * WN)
void conv_fused_kernel(const mcgen_conv_t p, const int a_bytes) {
    using C = ConvCfg<T, BM, BN, WM, WN>;
    using E = Elem<T>;
    using M = Mma<T>;
    constexpr int NT = C::NT, FM = C::FM, FN = C::FN, ESZ = C::ESZ, APITCH = C::APITCH, BROW = C::BROW, EP = C::EP;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ldsA = smem;
    char* ldsB = smem + a_bytes;
    float* epi = reinterpret_cast<float*>(smem);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, lg = lane >> 4;

    const int H = p.H, W = p.W, N = p.N;
    const int HW = H * W;
    const int tile_m = blockIdx.x;
    const int cout0 = blockIdx.y * BN;

    const Geo g = make_geo(BM, blockIdx.x, H, W);

    f32x4 acc[FN][FM];
#pragma unroll
    for (int i = 0; i < FN; ++i)
#pragma unroll
        for (int j = 0; j < FM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const T* wimg = reinterpret_cast<const T*>(p.w);
    size_t wblock = 0;                                   // running [chunk][tap] block index
    const size_t wblock_elems = (size_t)p.Cout_w * MCGEN_CK;

    for (int s = 0; s < p.nseg; ++s) {
        const mcgen_seg_t sg = p.seg[s];
        const int halo = sg.ksize >> 1;
        const int PR = g.TH + 2 * halo, PC = W + 2 * halo;

        PatchStager<T, NT, C::NI, APITCH> stager;
        stager.setup(sg, g, N, H, W, tid);
        // per-lane LDS base of each pixel fragment at tap (0,0)
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
        for (int q = 0; q < nchunk; ++q) {
            const int c0 = q * MCGEN_CK;
            __syncthreads();                                  // previous chunk's MFMA reads are done
            // ---- stage the input window of this chunk, prologue applied -----------------------
            stager.stage(sg, c0, ldsA);
            // ---- taps --------------------------------------------------------------------------
            for (int tap = 0; tap < ntap; ++tap) {
                if (tap > 0) __syncthreads();                 // previous tap's weight reads are done
                {   // stage this tap's weight block rows [cout0, cout0+BN) with the 16B-unit swizzle
                    const T* wb = wimg + wblock * wblock_elems;
                    constexpr int UPR = BROW / 16;            // 16-byte units per row
                    for (int u = tid; u < BN * UPR; u += NT) {
                        const int row = u / UPR, gu = u % UPR;
                        u32x4 val = {0u, 0u, 0u, 0u};
                        if (cout0 + row < p.Cout_w)
                            val = *reinterpret_cast<const u32x4*>(
                                reinterpret_cast<const char*>(wb + (size_t)(cout0 + row) * MCGEN_CK) + gu * 16);
                        const int grp = gu / (ESZ / 2), within = gu % (ESZ / 2);   // 8-channel group
                        const int sw = grp ^ (3 * ((row >> 3) & 1));
                        *reinterpret_cast<u32x4*>(ldsB + row * BROW + (sw * (ESZ / 2) + within) * 16) = val;
                    }
                }
                __syncthreads();
                const int kh = (sg.ksize == 3) ? tap / 3 : 0, kw = (sg.ksize == 3) ? tap % 3 : 0;
                const int tapoff = (kh * PC + kw) * APITCH;
                typename M::frag af[FM], wf[FN];
#pragma unroll
                for (int fm = 0; fm < FM; ++fm)
                    af[fm] = *reinterpret_cast<const typename M::frag*>(ldsA + a_base[fm] + tapoff);
#pragma unroll
                for (int fn = 0; fn < FN; ++fn) {
                    const int row = wn * (BN / WN) + fn * 16 + l15;
                    const int sw = lg ^ (3 * ((row >> 3) & 1));
                    wf[fn] = *reinterpret_cast<const typename M::frag*>(ldsB + row * BROW + sw * 8 * ESZ);
                }
#pragma unroll
                for (int fn = 0; fn < FN; ++fn)
#pragma unroll
                    for (int fm = 0; fm < FM; ++fm) M::run(wf[fn], af[fm], acc[fn][fm]);
                ++wblock;
            }
        }
    }

    // ---- epilogue: accumulators -> LDS (fp32 [pixel][cout]) -> fused output pass ----------------
    __syncthreads();
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm) {
            const int m = wm * (BM / WM) + fm * 16 + l15;
            const int co = wn * (BN / WN) + fn * 16 + lg * 4;
            *reinterpret_cast<f32x4*>(epi + m * EP + co) = acc[fn][fm];
        }
    __syncthreads();

    constexpr int CH = BN / 8;                 // 8-channel chunks per output pixel
    constexpr int PROWS = NT / CH;             // threads sharing one chunk
    const int ch = tid % CH, prow = tid / CH;
    const int co = cout0 + ch * 8;             // first channel of this thread's chunk
    const int out_pix = p.pool ? (BM >> 2) : BM;
    const int Ho = p.pool ? (H >> 1) : H, Wo = p.pool ? (W >> 1) : W;
    const bool chunk_live = co < p.Cy;
    T* y = reinterpret_cast<T*>(p.y);
    const T* res = reinterpret_cast<const T*>(p.res);
    const T* gx = reinterpret_cast<const T*>(p.gate_x);

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

    const int lgWo = p.pool ? g.lgW - 1 : g.lgW;
    const int lgTHWo = p.pool ? g.lgTHW - 2 : g.lgTHW;
    for (int mo = prow; mo < out_pix; mo += PROWS) {
        // output pixel mo of the tile -> (ti, ro, wo)
        const int ti = mo >> lgTHWo, rem = mo & ((1 << lgTHWo) - 1);
        const int ro = rem >> lgWo, wo = rem & ((1 << lgWo) - 1);
        const int n = g.n0 + ti;
        if (n >= N || !chunk_live) continue;
        float v[8];
        if (p.pool) {
            const int m00 = (ti << g.lgTHW) + ((2 * ro) << g.lgW) + 2 * wo;
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
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] *= ((co + i) < p.Cout) ? p.ocode[(size_t)n * p.Cout + co + i] : 0.f;
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

// ---- host side ----------------------------------------------------------------------------------
struct TilePick { int BM, BN; };

static TilePick pick_tile(const mcgen_conv_t* p) {
    const long M = (long)p->N * p->H * p->W;
    if (p->Cout_w <= 16) return {128, 16};
    if (M <= 16384 || p->Cout_w <= 64) return {64, 64};
    return {128, 128};
}

static int patch_pixels(const mcgen_conv_t* p, int BM) {
    int best = 0;
    for (int s = 0; s < p->nseg; ++s) {
        const int pp = mcgen_patch_pixels(BM, p->H, p->W, p->seg[s].ksize);
        if (pp > best) best = pp;
    }
    return best;
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
    int main_bytes = a_bytes + BN * C::BROW;
    int epi_bytes = BM * C::EP * 4;
    int red_bytes = (C::NT / (BN / 8)) * BN * 2 * 4;
    int lds = main_bytes > epi_bytes ? main_bytes : epi_bytes;
    if (red_bytes > lds) lds = red_bytes;
    auto kern = conv_fused_kernel<T, BM, BN, WM, WN>;
    static bool raised = false;
    if (lds > 64 * 1024 && !raised) {
        raised = true;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return mcgen_fail("conv_fused: cannot raise LDS limit to %d: %s", lds, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(mt, nt), dim3(C::NT), lds, st, *p, a_bytes);
    MCGEN_LAUNCH_CHECK("conv_fused");
    return 0;
}

template <typename T>
static int launch_dtype(const mcgen_conv_t* p, hipStream_t st) {
    const TilePick t = pick_tile(p);
    if (t.BN == 16) return launch_cfg<T, 128, 16, 4, 1>(p, st);
    if (t.BM == 64) return launch_cfg<T, 64, 64, 2, 2>(p, st);
    return launch_cfg<T, 128, 128, 2, 2>(p, st);
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
    (void)dtype;
    if (!p) return 0;
    const TilePick t = pick_tile(p);
    const long Mtot = (long)p->N * p->H * p->W;
    return (int)((Mtot + t.BM - 1) / t.BM);
}

extern "C" int mcgen_conv_tile(const mcgen_conv_t* p, int* bm, int* bn) {
    if (!p || !bm || !bn) return mcgen_fail("conv_tile: null pointer");
    const TilePick t = pick_tile(p);
    *bm = t.BM; *bn = t.BN;
    return 0;
}

extern "C" int mcgen_conv_fused(const mcgen_conv_t* p, int dtype, void* stream) {
    if (int rc = validate(p)) return rc;
    // pooling / whole-row tiles need at least two rows per tile
    const TilePick t = pick_tile(p);
    MCGEN_CHECK(t.BM >= 2 * p->W || p->H * p->W <= t.BM, "conv_fused: tile of %d pixels too small for W=%d", t.BM, p->W);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MCGEN_F32) return launch_dtype<float>(p, st);
    if (dtype == MCGEN_BF16) return launch_dtype<bf16_t>(p, st);
    return mcgen_fail("conv_fused: unknown dtype %d", dtype);
}
