// Weight gradients of the 3x3 convolutions of one backward pass in ONE launch (gfx950 MFMA, bf16).
//
//   dW[co][tap][ci] = sum over pixels  dy[pixel][co] * prologue(x)[pixel + tap][ci]          (mcgen_wgrad, wgrad.hip)
//
// What bounded the per-layer kernels of wgrad.hip on the big layers: a 64 co x 32 ci workgroup tile does
// 72 MFMAs per SIMD per 128-pixel step -- 0.48 us of matrix pipe against ~0.9 us of fixed per-step cost (barrier, DMA wait,
// the window's prologue pass, re-done by every output-channel block) -- and the small layers (8x8 maps, 20 launches per
// iteration) are all pipeline fill.  Here:
//   * the workgroup tile is 128 co x 64 ci x 9 taps (8 waves, 144 accumulator registers each: 64 co x 16 ci x 9 taps):
//     288 MFMAs per SIMD per step, the window prologue runs once per 128 output channels, operand bytes per FLOP halve;
//   * up to MCGEN_WGRAD_MULTI_MAX layers share the launch: the host sizes each layer's pixel splits by its share of the
//     FLOPs, so every workgroup walks about the same number of steps, the chip is filled once per PASS, and the split-K
//     slabs (one accumulator set per workgroup, whatever the tile) are paid once per pass instead of once per layer;
//   * dy tiles travel by LDS-DMA (unpadded 256-byte rows, 16-byte units XOR-swizzled by (row & 7) << 1 on the SOURCE
//     address: the transposing fragment reads are conflict-free); the x window goes through registers -- loaded two steps
//     ahead, prologue (BatchNorm affine, ReLU, MultimodalController code: modules.py:71-76) + ds_write one step ahead,
//     right after the barrier, where the other waves' MFMAs cover it (cdna_hip_programming.md T14);
//   * every LDS address of the step is one per-lane base + an instruction immediate (LGW, LGH are template parameters).
// Slabs keep wgrad.hip's layout [split][chunk of 32 ci][tap][Cout_w][32], so mcgen_wgrad_reduce(_batch) serves both.
#include "conv_tile.h"
#include <type_traits>

namespace {

constexpr int WB_BM = 128;                 // pixels per step
constexpr int WB_CO = 128, WB_CI = 64;     // workgroup tile
constexpr int WB_NT = 512;
// The two MFMA forms want different LDS images (the lane groups of a transposing read differ): M32 = v_mfma_f32_32x32x16_bf16.
//   16x16x32: window pitch 160 B (8 consecutive pixels fall on 8 distinct 32-byte bank groups), dy units swizzled by (row & 7) << 1;
//   32x32x16: window pitch 192 B (4 consecutive pixels x 64 bytes fall on 4 distinct 64-byte bank groups), dy units by (row & 3) << 2.
template <bool M32> struct WbLds {
    static constexpr int XPITCH = M32 ? 192 : 160;
    static constexpr int ABUF = M32 ? 39936 : 32768;             // one window buffer (<= 204 pixels x pitch, + 128 spare bytes)
    static constexpr int AFF = 2 * ABUF + 2 * WB_BM * WB_CO * 2; // BatchNorm scale | shift of the tile's 64 channels: 2 x 256 B
    static constexpr int CODE = AFF + 512;                       // code rows: 2 buffers x 8 images x 256 B (wave w brings image w % TI)
    static constexpr int TOTAL = CODE + 2 * 8 * 256;             // A0 A1 D0 D1 affine codes
    static __device__ __forceinline__ int dswz(int row) { return M32 ? ((row & 3) << 2) : ((row & 7) << 1); }
};
constexpr int WB_DROW = WB_CO * 2;         // 256 B per dy row
constexpr int WB_DBUF = WB_BM * WB_DROW;   // one dy tile
constexpr int WB_LDS = WbLds<true>::TOTAL; // (the larger of the two)
typedef float f32x16 __attribute__((ext_vector_type(16)));

static __device__ __forceinline__ s16x4 wb_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(reinterpret_cast<uintptr_t>(p)));
}
static __device__ __forceinline__ bf16x8 wb_frag(const char* p0, const char* p1) {
    union { bf16x8 v; s16x4 h[2]; } u;
    u.h[0] = wb_tr16(p0); u.h[1] = wb_tr16(p1);
    return u.v;
}

struct WgMulti {
    mcgen_wgrad_t l[MCGEN_WGRAD_MULTI_MAX];
    int first[MCGEN_WGRAD_MULTI_MAX + 1];      // first workgroup of each layer (first[n] = grid size)
    int n;
};

// Geometry of a 128-pixel step on a (1 << LGH) x (1 << LGW) map: TH whole rows of one image, or TI whole images; KS x KS taps.
template <int LGW, int LGH, int KS>
struct WbGeo {
    static constexpr int W = 1 << LGW, H = 1 << LGH, HW = W * H, HALO = KS >> 1, NTAP = KS * KS;
    static constexpr int TI = HW >= WB_BM ? 1 : WB_BM / HW;
    static constexpr int TH = HW >= WB_BM ? WB_BM / W : H;
    static constexpr int LGTHW = (HW >= WB_BM) ? 7 : LGW + LGH;        // log2(TH * W)
    static constexpr int PR = TH + 2 * HALO, PC = W + 2 * HALO, PP = TI * PR * PC;
    static constexpr int NIX = (PP * 8 + WB_NT - 1) / WB_NT;            // 16-byte window units per thread
    static_assert(PP * WbLds<false>::XPITCH <= WbLds<false>::ABUF - 128 && PP * WbLds<true>::XPITCH <= WbLds<true>::ABUF - 128 && TI <= 8 && PR + 1 < 32, "window buffer");
    static_assert(TH >= 2 || TI > 1, "tiles of a single row are not built (upsampled operands need even first rows)");
    // window position (pixel index) of tile pixel m, halo included
    static constexpr int winpos(int m) {
        return ((m >> LGTHW) * PR + ((m & ((1 << LGTHW) - 1)) >> LGW) + HALO) * PC + (m & (W - 1)) + HALO;
    }
};

template <int LGW, int LGH, int KS, bool M32 = false>
static __device__ __forceinline__ void wgrad_big_body(const mcgen_wgrad_t& p, const int bx, const int by, const int bz,
                                                      const int splits, char* smem) {
    using G = WbGeo<LGW, LGH, KS>;
    using L = WbLds<M32>;
    constexpr int W = G::W, H = G::H, HW = G::HW, PR = G::PR, PC = G::PC, PP = G::PP, NIX = G::NIX, TI = G::TI;
    constexpr int HALO = G::HALO, NTAP = G::NTAP;
    constexpr int WB_XPITCH = L::XPITCH, WB_ABUF = L::ABUF, WB_AFF = L::AFF, WB_CODE = L::CODE;
    static_assert(!M32 || KS == 3, "the 32x32x16 form splits the nine taps between the two waves of a SIMD");
    char* const ldsA = smem;
    char* const ldsD = smem + 2 * WB_ABUF;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wa = wave >> 2, wb = wave & 3;             // output-channel half (64), input-channel quarter (16)
    const int l15 = lane & 15, lg = lane >> 4, q4 = l15 >> 2, p4 = l15 & 3;
    const mcgen_seg_t sg = p.seg;
    const int co0 = bx * WB_CO, c0 = by * WB_CI;
    const long Mtot = (long)p.N * HW;
    const int m_tiles = (int)(Mtot / WB_BM);
    // step walk of this split: all steps with stride `splits`, or (halves) one half of them with stride splits / 2
    const int zs = p.halves ? (splits >> 1) : splits;
    const int mt = p.halves ? (m_tiles >> 1) : m_tiles;
    const int t_first = (p.halves ? (bz / zs) * mt : 0) + bz % zs;
    const int cnt = (mt - bz % zs + zs - 1) / zs;
    const bool do_bias = (p.bias_slabs != nullptr) && (by == 0);

    // ---- x window items of this thread: unit (window pixel pp, 8-channel group u); everything step-independent is a constant
    const int u8 = tid & 7;                                        // (WB_NT % 8 == 0: every item of a thread has the same u)
    const int cx = c0 + u8 * 8;
    const bool cok = cx < sg.C;
    // x_pk = LDS byte offset | (window row + 1) << 16 (0: the item never passes the row test) ; x_off = source element offset
    int x_pk[NIX], x_off[NIX];
#pragma unroll
    for (int k = 0; k < NIX; ++k) {
        const int pp = (tid + k * WB_NT) >> 3;
        const int ti = pp / (PR * PC), rem = pp - ti * (PR * PC);
        const int pr = rem / PC, pc = rem - pr * PC;
        const int dh = pr - HALO, w = pc - HALO;
        const bool item = pp < PP;
        const bool colok = item && cok && w >= 0 && w < W;
        // (threads past the window's last unit store zeros into the buffer's spare 128 bytes: no branch around an item)
        x_pk[k] = (item ? pp * WB_XPITCH + u8 * 16 : WB_ABUF - 128 + u8 * 16) | ((colok ? pr + 1 : 0) << 16);
        // element offset from the step's first source pixel; through the x2 upsample the source pixel of (h, w) is (h >> 1, w >> 1)
        // (steps start on even rows, so the halving splits into a step part and this constant part; dh = -1 -> row h0/2 - 1)
        x_off[k] = (sg.ups ? (ti * (HW >> 2) + (dh >> 1) * (W >> 1) + (w >> 1)) : (ti * HW + dh * W + w)) * sg.C + cx;
    }
    // the tile's BatchNorm affine: once per workgroup, kept in LDS (the registers are the accumulators')
    float* const ldsAff = reinterpret_cast<float*>(smem + WB_AFF);
    if (tid < 2 * WB_CI) {
        const int c = c0 + (tid & (WB_CI - 1));
        float v = (tid < WB_CI) ? 1.f : 0.f;
        if (sg.scale && c < sg.C) v = (tid < WB_CI) ? sg.scale[c] : sg.shift[c];
        ldsAff[tid] = v;
    }
    const float relu_lo = sg.relu ? 0.f : -__builtin_inff();
    const char* xs = reinterpret_cast<const char*>(sg.x);
    const char* dyb = reinterpret_cast<const char*>(p.dy);
    const size_t dpix = (size_t)p.Cdy * 2;

    // ---- dy units of this lane: DMA instruction d = wave * 4 + k covers rows 4 d .. 4 d + 3, 16 units of 16 bytes each
    int d_src[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int m = (wave * 4 + k) * 4 + (lane >> 4);
        int u = (lane & 15) ^ L::dswz(m);
        if (co0 + u * 8 + 8 > p.Cdy) u = 0;                       // beyond the dy pitch: any in-bounds unit (those rows are dropped)
        const int ti = m >> G::LGTHW, rem = m & ((1 << G::LGTHW) - 1);
        const int r = rem >> LGW, c = rem & (W - 1);
        const int mp = p.dy_ups ? (ti * (HW >> 2) + (r >> 1) * (W >> 1) + (c >> 1)) : m;
        d_src[k] = mp * (int)dpix + co0 * 2 + u * 16;
    }
    // first source pixel of step `tile` for an operand stored at the map's resolution or at half of it
    auto first_pixel = [&](int tile, bool up) -> size_t {
        const int pix0 = tile * WB_BM;
        if (!up) return (size_t)pix0;
        const int n0 = pix0 >> (LGW + LGH), h0 = (pix0 & (HW - 1)) >> LGW;
        return ((size_t)n0 * (H >> 1) + (h0 >> 1)) * (W >> 1);
    };
    auto tile_of = [&](int i) { return t_first + (i < cnt ? i : cnt - 1) * zs; };

    // ---- register stage of the x window: raw units; the step's code row(s) travel by LDS-DMA (64 floats per image)
    u32x4 raw[NIX];
    char* const ldsCode = smem + WB_CODE;
    auto load_x = [&](int i) {
        const int tile = tile_of(i);
        const int pix0 = tile * WB_BM;
        const int n0 = pix0 >> (LGW + LGH), h0 = (TI == 1) ? ((pix0 & (HW - 1)) >> LGW) : 0;
        const char* xb = xs + first_pixel(tile, sg.ups != 0) * sg.C * 2;
#pragma unroll
        for (int k = 0; k < NIX; ++k) {
            const int row = (x_pk[k] >> 16) & 31;                                      // window row + 1, 0 = never
            const bool ok = row != 0 && (unsigned)(h0 + row - 1 - HALO) < (unsigned)H;
            const char* src = ok ? xb + (ptrdiff_t)x_off[k] * 2 : xs;                   // outside: any in-bounds address
            raw[k] = *reinterpret_cast<const u32x4*>(src);
        }
        if (sg.code && wave < TI) {                                                    // wave t: the code row of image n0 + t
            const int c = c0 + lane;
            const float* src = sg.code + (size_t)(n0 + wave) * sg.C + (c < sg.C ? c : 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(ldsCode + ((i & 1) * 8 + wave) * 256), 4, 0, 0);
        }
    };
    // prologue + LDS store of the staged window: v -> max(v * sc + sh, relu ? 0 : -inf) * code, zeros outside the image
    auto write_x = [&](int i) {
        const int pix0 = tile_of(i) * WB_BM;
        const int h0 = (TI == 1) ? ((pix0 & (HW - 1)) >> LGW) : 0;
        char* dst = ldsA + (i & 1) * WB_ABUF;
        float sc[8], sh[8];
        load8f(ldsAff + u8 * 8, sc); load8f(ldsAff + WB_CI + u8 * 8, sh);
        auto pack_store = [&](int k, const float (&v)[8]) {
            const int row = (x_pk[k] >> 16) & 31, lo = x_pk[k] & 0xffff;
            const bool ok = row != 0 && (unsigned)(h0 + row - 1 - HALO) < (unsigned)H;
            union { bf16x8 h; u32x4 w; } o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o.h[e] = (bf16_t)v[e];
#pragma unroll
            for (int e = 0; e < 4; ++e) o.w[e] = ok ? o.w[e] : 0u;
            *reinterpret_cast<u32x4*>(dst + lo) = o.w;
        };
        if constexpr (TI == 1) {
            // One image per step: the thread's eight code entries are the same for all of its items.  Loaded ONCE and folded
            // into the affine in front of the ReLU: code * max(x sc + sh, lo) = sign(code) * max(x (sc |code|) + sh |code|, lo)
            // (lo = 0 or -inf) -- exact for any sign (the C ABI takes any float; MultimodalController codes are >= 0,
            // modules.py:58-76); the sign goes onto the packed bf16 words as an xor.  32 + 4 VALU per item where the literal form
            // with its per-item code reads took ~60.
            float cd[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) cd[e] = 1.f;
            if (sg.code) load8f(reinterpret_cast<const float*>(ldsCode + ((i & 1) * 8) * 256) + u8 * 8, cd);
            uint32_t smask[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                smask[e] = ((__float_as_uint(cd[2 * e]) >> 16) & 0x8000u) | (__float_as_uint(cd[2 * e + 1]) & 0x80000000u);
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] *= fabsf(cd[e]); sh[e] *= fabsf(cd[e]); }
#pragma unroll
            for (int k = 0; k < NIX; ++k) {
                const int row = (x_pk[k] >> 16) & 31, lo = x_pk[k] & 0xffff;
                const bool ok = row != 0 && (unsigned)(h0 + row - 1 - HALO) < (unsigned)H;
                union { bf16x8 h; u32x4 w; } o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    o.h[2 * e] = (bf16_t)fmaxf(fmaf(__uint_as_float(raw[k][e] << 16), sc[2 * e], sh[2 * e]), relu_lo);
                    o.h[2 * e + 1] = (bf16_t)fmaxf(fmaf(__uint_as_float(raw[k][e] & 0xffff0000u), sc[2 * e + 1], sh[2 * e + 1]), relu_lo);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) o.w[e] = ok ? (o.w[e] ^ smask[e]) : 0u;
                *reinterpret_cast<u32x4*>(dst + lo) = o.w;
                __builtin_amdgcn_sched_barrier(0);             // one item at a time: the accumulators leave ~100 registers for all of this
            }
        } else {
#pragma unroll
            for (int k = 0; k < NIX; ++k) {
                const int lo = x_pk[k] & 0xffff;
                const int img = lo / (PR * PC * WB_XPITCH);                            // image of the window the unit belongs to
                float cd[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) cd[e] = 1.f;
                if (sg.code) load8f(reinterpret_cast<const float*>(ldsCode + ((i & 1) * 8 + (img < TI ? img : 0)) * 256) + u8 * 8, cd);
                float v[8];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[2 * e] = fmaxf(fmaf(__uint_as_float(raw[k][e] << 16), sc[2 * e], sh[2 * e]), relu_lo) * cd[2 * e];
                    v[2 * e + 1] = fmaxf(fmaf(__uint_as_float(raw[k][e] & 0xffff0000u), sc[2 * e + 1], sh[2 * e + 1]), relu_lo) * cd[2 * e + 1];
                }
                pack_store(k, v);
                __builtin_amdgcn_sched_barrier(0);             // one item at a time: the accumulators leave ~100 registers for all of this
            }
        }
    };
    auto dma_dy = [&](int i) {
        const char* db = dyb + first_pixel(tile_of(i), p.dy_ups != 0) * dpix;
        char* ds = ldsD + (i & 1) * WB_DBUF + wave * 4096;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(db + d_src[k]),
                                             (__attribute__((address_space(3))) void*)(ds + k * 1024), 16, 0, 0);
    };

    if constexpr (!M32) {
        // ---- fragment addresses: one per-lane base per operand (four for dy: the swizzle moves the co fragment), steps and taps by immediates
        const int m0 = 4 * lg + q4;                                    // the lane's pixel row inside a 16-pixel group
        int aoff;                                                      // window: pixel m0 of the tile, tap (0, 0) -> halo origin
        {
            const int ti = 0, r = m0 >> LGW, c = m0 & (W - 1);         // (m0 < 16 <= TH * W: never leaves the first image)
            aoff = ((ti * PR + r) * PC + c) * WB_XPITCH + wb * 32 + p4 * 8;
        }
        int doff[4];
#pragma unroll
        for (int c = 0; c < 4; ++c)
            doff[c] = m0 * WB_DROW + 32 * ((wa * 4 + c) ^ (m0 & 7)) + 8 * p4;

        f32x4 acc[NTAP][4];
#pragma unroll
        for (int j = 0; j < NTAP; ++j)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[j][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        // bias gradient = column sums of dy: thread (16-byte unit tid & 15, row group tid >> 4) adds rows rg, rg + 32, rg + 64, rg + 96 of
        // every tile (four 16-byte LDS reads per step; the row groups meet in LDS once, after the last step)
        float bsum8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum8[e] = 0.f;

        // ---- pipeline: x of step i + 2 in registers, x of step i + 1 written and dy of step i + 1 in flight while step i multiplies
        load_x(0);
        dma_dy(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                                   // the affine rows and step 0's code row(s) are in LDS
        write_x(0);
        if (cnt > 1) load_x(1);
#ifdef WB_LATE
        // Waves 4-7 (the second wave of every SIMD) multiply FIRST and stage step i + 1 behind their MFMAs, while their SIMD
        // partner stages first: one wave's prologue VALU under the other's matrix work.  Their window loads of step i + 2 then
        // leave at the END of step i and have the whole multiply phase of step i + 1 to land (they are waited for in front of
        // the late write_x, not at the barrier); only the dy DMA -- which everyone reads behind the next barrier -- goes out early.
        // (Not for steps of more than four images: there waves 4-7 carry code-row DMAs that the early waves read.)
        const bool late = (wa != 0) && TI <= 4;
#endif
        for (int i = 0; i < cnt; ++i) {
#ifdef WB_LATE
            if (late) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NIX) : "memory");   // (this wave's x loads of step i + 1 stay in flight)
            else
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // dy of step i has landed (and this thread's x of step i + 1)
            __syncthreads();                                               // step i published; the buffers of step i - 1 are free
            // Staging of step i + 1 (window prologue + LDS stores, dy DMA, register loads of step i + 2).  Waves 4-7 (the second
            // wave of every SIMD) run it in the MIDDLE of the step's multiply loop instead of in front of it: all eight waves
            // leave the barrier together, and with the same program they would all spend the next ~1 us in the prologue's VALU
            // work with the matrix pipe idle -- now one wave of each SIMD multiplies while its partner stages (any point of the
            // step is legal: the buffers of step i + 1 were last read before the barrier, and are next read behind the next one).
            auto stage_next = [&]() {
                if (i + 1 < cnt) {
#ifndef WB_ABL_NO_XWRITE
                    write_x(i + 1);
#endif
#ifndef WB_ABL_NO_DMA
                    dma_dy(i + 1);
#endif
#ifndef WB_ABL_NO_XLOAD
                    if (i + 2 < cnt) load_x(i + 2);
#endif
                }
            };
#if defined(WB_STAGGER)
            const bool late = wa != 0;
#elif !defined(WB_LATE)
            const bool late = false;
#endif
            if (!late) stage_next();
#ifdef WB_LATE
            else if (i + 1 < cnt) dma_dy(i + 1);
#endif
            __builtin_amdgcn_sched_barrier(0);                             // (the staging block's temporaries die before the fragments come alive)
            const char* A = ldsA + (i & 1) * WB_ABUF + aoff;
            const char* D = ldsD + (i & 1) * WB_DBUF;
            if (do_bias) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const u32x4 r = *reinterpret_cast<const u32x4*>(D + ((tid >> 4) + 32 * k) * WB_DROW + (tid & 15) * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        bsum8[2 * e] += __uint_as_float(r[e] << 16);
                        bsum8[2 * e + 1] += __uint_as_float(r[e] & 0xffff0000u);
                    }
                }
            }
            // fragment reads one tap ahead of their MFMAs (two window fragments in registers: the register budget allows no more);
            // the dy fragments of a 32-pixel group are read with its first tap
            constexpr int HALO0 = HALO * PC + HALO;
            auto rd_a = [&](int ks, int j) {
                const int g0 = (G::winpos(32 * ks) - HALO0) * WB_XPITCH, g1 = (G::winpos(32 * ks + 16) - HALO0) * WB_XPITCH;
                const int tap = ((j / KS) * PC + (j % KS)) * WB_XPITCH;
#ifdef WB_ABL_NO_FRAG
                bf16x8 z; for (int e = 0; e < 8; ++e) z[e] = (bf16_t)(float)(lane + ks + j);
                asm volatile("" : "+v"(z));
                return z;
#else
                return wb_frag(A + g0 + tap, A + g1 + tap);
#endif
            };
            // (two named buffers picked by the compile-time parity of the tap counter: a rotating `next -> current` copy costs
            // four v_mov per tap -- measured 3.1 VALU per MFMA with it)
#ifndef WB_DEPTH
#define WB_DEPTH 1
#endif
            constexpr int NTT = (WB_BM / 32) * NTAP;                       // taps per step, in loop order
            auto rd_t = [&](int t) { return rd_a(t / NTAP, t % NTAP); };
            bf16x8 afb[WB_DEPTH + 1];
#pragma unroll
            for (int t = 0; t < WB_DEPTH && t < NTT; ++t) afb[t] = rd_t(t);
#pragma unroll
            for (int ks = 0; ks < WB_BM / 32; ++ks) {
#ifdef WB_STAGGER
                if (ks == WB_STAGGER) {
                    if (late) {
                        __builtin_amdgcn_sched_barrier(0);
                        stage_next();
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#endif
                bf16x8 df[4];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    df[c] = wb_frag(D + doff[c] + (32 * ks) * WB_DROW, D + doff[c] + (32 * ks + 16) * WB_DROW);
#pragma unroll
                for (int j = 0; j < NTAP; ++j) {
                    const int t = ks * NTAP + j;                           // compile-time: the loops are fully unrolled
                    if (t + WB_DEPTH < NTT) afb[(t + WB_DEPTH) % (WB_DEPTH + 1)] = rd_t(t + WB_DEPTH);
#ifdef WB_ABL_NO_MFMA
#pragma unroll
                    for (int c = 0; c < 4; ++c) asm volatile("" :: "v"(df[c]), "v"(afb[t % (WB_DEPTH + 1)]));
#else
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[j][c] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df[c], afb[t % (WB_DEPTH + 1)], acc[j][c], 0, 0, 0);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#ifdef WB_LATE
            if (late && i + 1 < cnt) {
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // x of step i + 1 (loaded a step ago) and this step's dy DMA
                write_x(i + 1);
                if (i + 2 < cnt) load_x(i + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        }
        // ---- slab[z][chunk][tap][co][32]: lane holds D[co = 4 lg + r][ci = l15] of (tap j, co fragment c)
        const int nchunk = (sg.C + MCGEN_CK - 1) / MCGEN_CK;
        const size_t slab_elems = (size_t)nchunk * NTAP * p.Cout_w * MCGEN_CK;
        float* out = p.slabs + (size_t)bz * slab_elems;
        const int qc = by * 2 + (wb >> 1), col = (wb & 1) * 16 + l15;
        if (qc < nchunk) {
#pragma unroll
            for (int j = 0; j < NTAP; ++j)
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int co = co0 + (wa * 4 + c) * 16 + lg * 4 + r;
                        if (co < p.Cout_w) out[(((size_t)qc * NTAP + j) * p.Cout_w + co) * MCGEN_CK + col] = acc[j][c][r];
                    }
        }
        if (do_bias) {
            // (workgroup-uniform branch) the 32 row groups of a column meet in LDS; rows [split * 4 + 1 .. + 3] of bias_slabs stay
            // zero -- mcgen_wgrad_reduce adds all splits * 4 rows
            __syncthreads();                                               // the last step's fragment reads are done: LDS is free
            float* red = reinterpret_cast<float*>(smem);
            const int rg = tid >> 4, lu = (tid & 15) ^ L::dswz(rg);        // the unit's logical position (rows rg + 32 k share rg & 7)
#pragma unroll
            for (int e = 0; e < 8; ++e) red[rg * WB_CO + lu * 8 + e] = bsum8[e];
            __syncthreads();
            const int colb = tid & 127, part = tid >> 7;
            float s = 0.f;
            if (part == 0)
                for (int r = 0; r < 32; ++r) s += red[r * WB_CO + colb];
            if (co0 + colb < p.Cout_w) p.bias_slabs[((size_t)bz * 4 + part) * p.Cout_w + co0 + colb] = s;
        }
    } else {
        // ================= v_mfma_f32_32x32x16_bf16 form (3x3 layers) ==================================================
        // The step loop is bound by the SIMD's instruction ISSUE, not by the matrix pipe: per 128-pixel step a wave of the
        // 16x16x32 form issues 144 MFMAs (8 issue cycles each), ~100 transposing LDS reads and ~300 VALU -- two such waves
        // need ~7000 issue cycles per SIMD against 4608 cycles of matrix pipe.  A 32x32x16 MFMA does twice the work for the
        // same 8 issue cycles.  Its 32 x 32 tile wants 32 input channels per wave, so the nine taps are SPLIT between the two
        // waves of a SIMD instead of the input channels: wave = (64 co x 32 ci block blk = wave & 3) x (taps 0-4 | taps 5-8),
        // 160 / 128 accumulator registers, a K step = 16 pixels.
        // Lane l of a 32x32x16 operand holds k = 8 (l >> 5) + j of row / column l & 31; a transposing read serves a 16-lane
        // group with 4 k rows x 16 columns: group g takes columns 16 (g & 1) .., k rows 8 (g >> 1) + 4 rd + (0..3), rd = 0, 1.
        const int blk = wave & 3, tg = wave >> 2;                      // (waves w and w + 4 share a SIMD)
        const int cb = blk >> 1, ib = blk & 1;                         // output-channel half (64), input-channel half (32)
        const int kh = lg >> 1, gsel = lg & 1;
        constexpr int HALO0 = HALO * PC + HALO;
        const int aoff = (G::winpos(8 * kh + q4) - HALO0) * WB_XPITCH + ib * 64 + gsel * 32 + p4 * 8;
        int doff[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
            doff[c] = (8 * kh + q4) * WB_DROW + 32 * ((cb * 4 + c * 2 + gsel) ^ (q4 << 1)) + 8 * p4;
        float bsum8[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bsum8[e] = 0.f;
        const int nchunk = (sg.C + MCGEN_CK - 1) / MCGEN_CK;
        const size_t slab_elems = (size_t)nchunk * NTAP * p.Cout_w * MCGEN_CK;
        float* const out = p.slabs + (size_t)bz * slab_elems;

        auto run = [&](auto nt_c, auto t0_c) {
            constexpr int NT = decltype(nt_c)::value, T0 = decltype(t0_c)::value;
            f32x16 acc[NT][2];
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[j][c][r] = 0.f;
            load_x(0);
            dma_dy(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                               // the affine rows and step 0's code row(s) are in LDS
            write_x(0);
            if (cnt > 1) load_x(1);
            for (int i = 0; i < cnt; ++i) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // dy of step i has landed (and this thread's x of step i + 1)
                __syncthreads();                                           // step i published; the buffers of step i - 1 are free
                if (i + 1 < cnt) {
                    write_x(i + 1);
                    dma_dy(i + 1);
                    if (i + 2 < cnt) load_x(i + 2);
                }
                __builtin_amdgcn_sched_barrier(0);
                const char* A = ldsA + (i & 1) * WB_ABUF + aoff;
                const char* D = ldsD + (i & 1) * WB_DBUF;
                if (do_bias) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const u32x4 r = *reinterpret_cast<const u32x4*>(D + ((tid >> 4) + 32 * k) * WB_DROW + (tid & 15) * 16);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            bsum8[2 * e] += __uint_as_float(r[e] << 16);
                            bsum8[2 * e + 1] += __uint_as_float(r[e] & 0xffff0000u);
                        }
                    }
                }
                constexpr int NKS = WB_BM / 16, NTT = NKS * NT;
                auto rd_x = [&](int t) {                                   // (t: compile-time -- the loops are fully unrolled)
                    const int ks = t / NT, j = T0 + t % NT;
                    const int off = (G::winpos(16 * ks) - G::winpos(0) + (j / KS) * PC + (j % KS)) * WB_XPITCH;
                    return wb_frag(A + off, A + off + 4 * WB_XPITCH);
                };
                bf16x8 xb[2];
                xb[0] = rd_x(0);
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    bf16x8 df[2];
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        df[c] = wb_frag(D + doff[c] + (16 * ks) * WB_DROW, D + doff[c] + (16 * ks + 4) * WB_DROW);
#pragma unroll
                    for (int jj = 0; jj < NT; ++jj) {
                        const int t = ks * NT + jj;
                        if (t + 1 < NTT) xb[(t + 1) & 1] = rd_x(t + 1);
#pragma unroll
                        for (int c = 0; c < 2; ++c)
                            acc[jj][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df[c], xb[t & 1], acc[jj][c], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            // ---- slab[z][chunk][tap][co][32]: lane holds D[co = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)][ci = lane & 31] of (tap, co block c)
            const int qc = by * 2 + ib, col = lane & 31;
            if (qc < nchunk) {
#pragma unroll
                for (int jj = 0; jj < NT; ++jj)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int co = co0 + cb * 64 + c * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                            if (co < p.Cout_w) out[(((size_t)qc * NTAP + T0 + jj) * p.Cout_w + co) * MCGEN_CK + col] = acc[jj][c][r];
                        }
            }
        };
        if (tg == 0) run(std::integral_constant<int, 5>{}, std::integral_constant<int, 0>{});
        else run(std::integral_constant<int, 4>{}, std::integral_constant<int, 5>{});
        if (do_bias) {
            __syncthreads();                                               // the last step's fragment reads are done: LDS is free
            float* red = reinterpret_cast<float*>(smem);
            const int rg = tid >> 4, lu = (tid & 15) ^ L::dswz(rg);
#pragma unroll
            for (int e = 0; e < 8; ++e) red[rg * WB_CO + lu * 8 + e] = bsum8[e];
            __syncthreads();
            const int colb = tid & 127, part = tid >> 7;
            float s = 0.f;
            if (part == 0)
                for (int r = 0; r < 32; ++r) s += red[r * WB_CO + colb];
            if (co0 + colb < p.Cout_w) p.bias_slabs[((size_t)bz * 4 + part) * p.Cout_w + co0 + colb] = s;
        }
    }
}

__global__ __launch_bounds__(WB_NT, 1)
void wgrad_multi_kernel(const WgMulti a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef WB_CLOCK
    const unsigned long long wbc0 = __builtin_amdgcn_s_memtime(), wbr0 = __builtin_amdgcn_s_memrealtime();
#endif
    int l = 0;
#pragma unroll
    for (int k = 1; k < MCGEN_WGRAD_MULTI_MAX; ++k) l += (k < a.n && (int)blockIdx.x >= a.first[k]) ? 1 : 0;
    const mcgen_wgrad_t& p = a.l[l];
    const int local = (int)blockIdx.x - a.first[l];
    const int gx = p.Cout_w / WB_CO, gy = p.seg.C / WB_CI;
    const int bx = local % gx, by = (local / gx) % gy, bz = local / (gx * gy);
    const int lgw = 31 - __builtin_clz(p.W);
#ifndef WB_M32
#define WB_M32 0
#endif
    if (p.seg.ksize == 3) {
        switch (lgw) {
            case 5: wgrad_big_body<5, 5, 3, WB_M32 != 0>(p, bx, by, bz, p.splits, smem); break;
            case 4: wgrad_big_body<4, 4, 3, WB_M32 != 0>(p, bx, by, bz, p.splits, smem); break;
            default: wgrad_big_body<3, 3, 3, WB_M32 != 0>(p, bx, by, bz, p.splits, smem); break;
        }
    } else {
        switch (lgw) {
            case 5: wgrad_big_body<5, 5, 1>(p, bx, by, bz, p.splits, smem); break;
            case 4: wgrad_big_body<4, 4, 1>(p, bx, by, bz, p.splits, smem); break;
            case 3: wgrad_big_body<3, 3, 1>(p, bx, by, bz, p.splits, smem); break;
            default: wgrad_big_body<2, 2, 1>(p, bx, by, bz, p.splits, smem); break;
        }
    }
#ifdef WB_CLOCK
    // diagnostic build only: shader clock of this workgroup = d(s_memtime) / d(s_memrealtime) x 100 MHz
    if ((blockIdx.x & 63) == 5 && threadIdx.x == 0) {
        const unsigned long long c = __builtin_amdgcn_s_memtime() - wbc0, r = __builtin_amdgcn_s_memrealtime() - wbr0;
        printf("wgclk block %d cycles %llu realtime %llu -> %.3f GHz\n", (int)blockIdx.x, c, r, (double)c / (double)r * 0.1);
    }
#endif
}

}  // namespace

extern "C" int mcgen_wgrad_multi_ok(const mcgen_wgrad_t* p, int dtype) {
    if (!p || dtype != MCGEN_BF16) return 0;
    if (p->seg.ksize != 3 && p->seg.ksize != 1) return 0;
    if (p->H != p->W || (p->W != 8 && p->W != 16 && p->W != 32 && !(p->W == 4 && p->seg.ksize == 1))) return 0;
    if (p->Cout_w % WB_CO || p->seg.C % WB_CI || p->Cout != p->Cout_w || p->Cdy < p->Cout_w) return 0;
    if (((long)p->N * p->H * p->W) % WB_BM) return 0;
    if (p->seg.group_n || p->seg.cmap) return 0;
    return 1;
}

extern "C" int mcgen_wgrad_multi(const mcgen_wgrad_t* layers, int n, int dtype, void* stream) {
    MCGEN_CHECK(layers && n >= 1 && n <= MCGEN_WGRAD_MULTI_MAX, "wgrad_multi: 1 .. %d layers per launch", MCGEN_WGRAD_MULTI_MAX);
    WgMulti a;
    int total = 0;
    for (int i = 0; i < n; ++i) {
        const mcgen_wgrad_t& p = layers[i];
        MCGEN_CHECK(p.seg.x && p.dy && p.slabs, "wgrad_multi: null pointer (layer %d)", i);
        MCGEN_CHECK(mcgen_wgrad_multi_ok(&p, dtype), "wgrad_multi: layer %d is not eligible (bf16, 3x3 or 1x1, square 8/16/32 maps -- 1x1 also 4, Cout %% 128 == 0, "
                    "C %% 64 == 0, whole 128-pixel steps)", i);
        const long m_tiles = (long)p.N * p.H * p.W / WB_BM;
        MCGEN_CHECK(p.splits >= 1 && p.splits <= m_tiles, "wgrad_multi: layer %d: bad splits", i);
        MCGEN_CHECK(!p.halves || (p.splits % 2 == 0 && m_tiles % 2 == 0 && p.splits / 2 <= m_tiles / 2), "wgrad_multi: layer %d: halves needs even splits and steps", i);
        MCGEN_CHECK(!(p.seg.ups || p.dy_ups) || p.H >= 4, "wgrad_multi: layer %d: upsampled operands need maps of at least 4 rows", i);
        a.l[i] = p;
        a.first[i] = total;
        total += (p.Cout_w / WB_CO) * (p.seg.C / WB_CI) * p.splits;
    }
    for (int i = n; i < MCGEN_WGRAD_MULTI_MAX; ++i) { a.l[i] = layers[0]; a.first[i] = total; }
    a.first[MCGEN_WGRAD_MULTI_MAX] = total;
    a.n = n;
    static bool raised = false;
    if (!raised) {
        raised = true;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_multi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, WB_LDS);
        if (e != hipSuccess) return mcgen_fail("wgrad_multi: cannot raise LDS limit: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(wgrad_multi_kernel, dim3(total), dim3(WB_NT), WB_LDS, reinterpret_cast<hipStream_t>(stream), a);
    MCGEN_LAUNCH_CHECK("wgrad_multi");
    return 0;
}
