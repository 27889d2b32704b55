// Tile geometry and the prologue-applying LDS window stager shared by the fused convolution
// (conv_fused.hip) and its weight-gradient kernel (wgrad.hip).
#pragma once
#include "mcgen_common.h"

// conv_skinny.hip: Cout <= 16 over a deep K on 4x4 / 8x8 / 16x16 maps, K split over the waves of a workgroup
int mcgen_conv_skinny_ok(const mcgen_conv_t* p, int dtype);
int mcgen_conv_skinny(const mcgen_conv_t* p, hipStream_t st);
// conv_smap.hip: 3x3 128 -> 128 on 8x8 maps, one image per workgroup, weight fragments straight from L2
int mcgen_conv_smap_ok(const mcgen_conv_t* p, int dtype);
int mcgen_conv_smap(const mcgen_conv_t* p, hipStream_t st);
// conv_px1.hip: 1x1 512 -> 512 with the pixel tile resident in LDS (pixels per tile, 0 = not taken)
int mcgen_conv_px1_bm(const mcgen_conv_t* p, int dtype);
int mcgen_conv_px1(const mcgen_conv_t* p, hipStream_t st);
// conv_c8.hip: 3x3 on an 8-channel (image) tensor -> 128 channels, 32x32 maps: K = (tap, channel), stores straight from the accumulators
int mcgen_conv_c8_ok(const mcgen_conv_t* p, int dtype);
int mcgen_conv_c8(const mcgen_conv_t* p, hipStream_t st);
// conv_head.hip: 3x3 to <= 8 output channels (pitch 8) on 32x32 maps, double-buffered input windows (the generator's image head)
int mcgen_conv_head_ok(const mcgen_conv_t* p, int dtype);
int mcgen_conv_head(const mcgen_conv_t* p, hipStream_t st);

// wgrad_c8.hip: weight gradients of 3x3 / 1x1 convolutions whose input is the 8-channel image tensor (bf16, 32x32, 128 outputs)
extern "C" int mcgen_wgrad_c8_ok(const mcgen_wgrad_t* p, int dtype);
int mcgen_wgrad_c8(const mcgen_wgrad_t* p, hipStream_t st);
extern "C" int64_t mcgen_wgrad_c8_slab_elems(const mcgen_wgrad_t* p);      // compact slabs: (tap, channel) pairs as columns

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    typedef bf16x8 frag;
    static __device__ __forceinline__ void run(const frag& w, const frag& a, f32x4& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, acc, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    typedef f32x8 frag;
    static __device__ __forceinline__ void run(const frag& w, const frag& a, f32x4& acc) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(w[i], a[i], acc, 0, 0, 0);
    }
};

// A tile is BM consecutive pixels of the flattened (n, h, w) grid: either TH whole rows of one
// image, or TI whole images.  H, W and BM are powers of two.
struct Geo {
    int n0, h0;             // first image / first row of the tile
    int TI, TH;             // images and rows per tile
    int lgW, lgTHW;         // log2(W), log2(TH*W)
};

static __device__ __forceinline__ Geo make_geo(int BM, int tile_m, int H, int W) {
    Geo g;
    const int HW = H * W;
    g.lgW = 31 - __builtin_clz(W);
    if (BM <= HW) {
        g.TI = 1; g.TH = BM / W;
        const int pix0 = tile_m * BM;
        g.n0 = pix0 / HW; g.h0 = (pix0 % HW) / W;
    } else {
        g.TI = BM / HW; g.TH = H;
        g.n0 = tile_m * g.TI; g.h0 = 0;
    }
    g.lgTHW = 31 - __builtin_clz(g.TH * W);
    return g;
}

// The segment as THIS tile sees it: with BatchNorm statistics groups (mcgen_seg_t.group_n) the affine of the tile's group.
static __device__ __forceinline__ mcgen_seg_t seg_for_tile(const mcgen_seg_t& s, const Geo& g) {
    mcgen_seg_t r = s;
    if (r.group_n > 0 && r.scale) {
        const int grp = g.n0 / r.group_n;
        r.scale += (size_t)grp * r.C; r.shift += (size_t)grp * r.C;
    }
    return r;
}

// Stages the tile's input window (tile + halo) for one chunk of 32 channels into LDS as
// [window pixel][32 channels] with pitch APITCH, applying the segment's prologue:
// nearest-x2 upsample by index, BatchNorm scale/shift, ReLU, MultimodalController code.
// Out-of-image pixels and channels beyond C are written as zeros (the conv's zero padding).
template <typename T, int NT, int NI, int APITCH, int SUBS = 4>
struct PatchStager {
    using E = Elem<T>;
    int it_src[NI];      // element offset of the source pixel (+ sub-chunk), -1 = zero fill
    int it_n[NI];        // image index (row of the code table)
    int it_lds[NI];      // byte offset in the LDS window, -1 = no item
    int it_sub[1];       // channel offset inside the chunk (0, 8, 16, 24): the same for all items of a thread

    int it_pos[NI];      // packed window position of the item: pr | pc << 8 | ti << 16

    // Tile-independent part (integer divisions by the window width): once per kernel / segment.
    __device__ __forceinline__ void setup_static(int ksize, const Geo& g, int W, int tid) {
        const int halo = ksize >> 1;
        const int PR = g.TH + 2 * halo, PC = W + 2 * halo;
        const int PP = g.TI * PR * PC;
        static_assert(NT % SUBS == 0, "items of one thread share the sub-chunk");
        it_sub[0] = (tid % SUBS) * 8;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            const int it = tid + k * NT;
            it_lds[k] = -1; it_pos[k] = 0;
            if (it < PP * SUBS) {
                const int sub = it % SUBS, pp = it / SUBS;
                const int pc = pp % PC, t2 = pp / PC;
                const int pr = t2 % PR, ti = t2 / PR;
                it_lds[k] = pp * APITCH + sub * 8 * E::BYTES;
                it_pos[k] = pr | (pc << 8) | (ti << 16);
            }
        }
    }
    // Tile-dependent part: source offsets and validity for the tile described by g (a few adds per item).
    __device__ __forceinline__ void bind(const mcgen_seg_t& sg, const Geo& g, int N, int H, int W) {
        const int halo = sg.ksize >> 1;
        const int Hs = sg.ups ? (H >> 1) : H, Ws = sg.ups ? (W >> 1) : W;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            it_src[k] = -1; it_n[k] = 0;
            if (it_lds[k] >= 0) {
                const int pr = it_pos[k] & 255, pc = (it_pos[k] >> 8) & 255, ti = it_pos[k] >> 16;
                const int n = g.n0 + ti, h = g.h0 + pr - halo, w = pc - halo;
                it_n[k] = n;
                if (n < N && h >= 0 && h < H && w >= 0 && w < W) {
                    const int hs = sg.ups ? (h >> 1) : h, ws = sg.ups ? (w >> 1) : w;
                    it_src[k] = ((n * Hs + hs) * Ws + ws) * sg.C + it_sub[0];
                }
            }
        }
    }
    __device__ __forceinline__ void setup(const mcgen_seg_t& sg, const Geo& g, int N, int H, int W, int tid) {
        setup_static(sg.ksize, g, W, tid);
        bind(sg, g, N, H, W);
    }

    // raw 8-channel groups fetched by load(), consumed by write(): lets the caller issue the global
    // loads of the NEXT chunk early and do the prologue + LDS store after the current chunk's MFMAs
    typedef u32x4 raw_t[NI][E::BYTES / 2];

    __device__ __forceinline__ void load(const mcgen_seg_t& sg, int c0, raw_t& raw) const {
        const char* xs = reinterpret_cast<const char*>(sg.x);
#pragma unroll
        for (int k = 0; k < NI; ++k) {
#pragma unroll
            for (int j = 0; j < E::BYTES / 2; ++j) raw[k][j] = u32x4{0u, 0u, 0u, 0u};
            if (it_lds[k] >= 0 && it_src[k] >= 0 && c0 + it_sub[0] < sg.C) {
                const char* p = xs + ((size_t)it_src[k] + c0) * E::BYTES;
#pragma unroll
                for (int j = 0; j < E::BYTES / 2; ++j) raw[k][j] = *reinterpret_cast<const u32x4*>(p + 16 * j);
            }
        }
    }

    static __device__ __forceinline__ void unpack(const u32x4 (&r)[E::BYTES / 2], float (&v)[8]) {
        if constexpr (E::BYTES == 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[2 * i] = __uint_as_float(r[0][i] << 16);
                v[2 * i + 1] = __uint_as_float(r[0][i] & 0xffff0000u);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) { v[i] = __uint_as_float(r[0][i]); v[4 + i] = __uint_as_float(r[E::BYTES / 2 - 1][i]); }
        }
    }

    // one_n >= 0: the tile lies inside image one_n, so every item shares that code row (loaded once); -1: per item
    __device__ __forceinline__ void write(const mcgen_seg_t& sg, int c0, const raw_t& raw, char* ldsA, int one_n = -1) const {
        // every item of a thread has the same sub-chunk (NT % 4 == 0), so the BN affine is loaded once
        const int c = c0 + it_sub[0];
        const bool cok = c < sg.C;
        float sc[8], sh[8], cd1[8];
        if (sg.scale && cok) { load8f(sg.scale + c, sc); load8f(sg.shift + c, sh); }
        const bool code_once = sg.code && one_n >= 0 && cok;      // tile inside one image: one code row for all items
        if (code_once) load8f(sg.code + (size_t)one_n * sg.C + c, cd1);
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            if (it_lds[k] < 0) continue;
            float v[8];
            unpack(raw[k], v);
            if (it_src[k] >= 0 && cok) {
                if (sg.scale) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], sc[i], sh[i]);
                }
                if (sg.relu) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
                }
                if (code_once) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= cd1[i];
                } else if (sg.code) {
                    float cd[8];
                    load8f(sg.code + (size_t)it_n[k] * sg.C + c, cd);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= cd[i];
                }
            }
            E::store8(reinterpret_cast<T*>(ldsA + it_lds[k]), v);
            __builtin_amdgcn_sched_barrier(0);          // keep items sequential: bounds the live registers
        }
    }

    // ---- forms with caller-held per-tile metadata (two tiles in flight: wgrad producers) ----------------
    __device__ __forceinline__ void bind_into(const mcgen_seg_t& sg, const Geo& g, int N, int H, int W,
                                              int (&src)[NI], int (&nn)[NI]) const {
        const int halo = sg.ksize >> 1;
        const int Hs = sg.ups ? (H >> 1) : H, Ws = sg.ups ? (W >> 1) : W;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            src[k] = -1; nn[k] = 0;
            if (it_lds[k] >= 0) {
                const int pr = it_pos[k] & 255, pc = (it_pos[k] >> 8) & 255, ti = it_pos[k] >> 16;
                const int n = g.n0 + ti, h = g.h0 + pr - halo, w = pc - halo;
                nn[k] = n;
                if (n < N && h >= 0 && h < H && w >= 0 && w < W) {
                    const int hs = sg.ups ? (h >> 1) : h, ws = sg.ups ? (w >> 1) : w;
                    src[k] = ((n * Hs + hs) * Ws + ws) * sg.C + it_sub[0];
                }
            }
        }
    }
    __device__ __forceinline__ void load_ext(const mcgen_seg_t& sg, int c0, const int (&src)[NI], raw_t& raw) const {
        const char* xs = reinterpret_cast<const char*>(sg.x);
#pragma unroll
        for (int k = 0; k < NI; ++k) {
#pragma unroll
            for (int j = 0; j < E::BYTES / 2; ++j) raw[k][j] = u32x4{0u, 0u, 0u, 0u};
            if (it_lds[k] >= 0 && src[k] >= 0 && c0 + it_sub[0] < sg.C) {
                const char* p = xs + ((size_t)src[k] + c0) * E::BYTES;
#pragma unroll
                for (int j = 0; j < E::BYTES / 2; ++j) raw[k][j] = *reinterpret_cast<const u32x4*>(p + 16 * j);
            }
        }
    }
    __device__ __forceinline__ void write_ext(const mcgen_seg_t& sg, int c0, const int (&src)[NI], const int (&nn)[NI],
                                              const raw_t& raw, char* ldsA, int one_n = -1) const {
        const int c = c0 + it_sub[0];
        const bool cok = c < sg.C;
        float sc[8], sh[8], cd1[8];
        if (sg.scale && cok) { load8f(sg.scale + c, sc); load8f(sg.shift + c, sh); }
        const bool code_once = sg.code && one_n >= 0 && cok;      // tile inside one image: one code row for all items
        if (code_once) load8f(sg.code + (size_t)one_n * sg.C + c, cd1);
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            if (it_lds[k] < 0) continue;
            float v[8];
            unpack(raw[k], v);
            if (src[k] >= 0 && cok) {
                if (sg.scale) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], sc[i], sh[i]);
                }
                if (sg.relu) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
                }
                if (code_once) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= cd1[i];
                } else if (sg.code) {
                    float cd[8];
                    load8f(sg.code + (size_t)nn[k] * sg.C + c, cd);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= cd[i];
                }
            }
            E::store8(reinterpret_cast<T*>(ldsA + it_lds[k]), v);
        }
    }

    // ---- forms with caller-held prologue vectors -----------------------------------------------------------
    // Memory loads return in order, so a prologue vector requested inside write_ext() queues behind every
    // load issued before it -- including the NEXT tile's prefetch, whose latency it then exposes.  A caller
    // that keeps tiles in flight loads the BN affine once (it depends on the chunk only) and the code row
    // together with the tile's own loads, and hands both to write_pre().
    __device__ __forceinline__ void load_affine(const mcgen_seg_t& sg, int c0, float (&sc)[8], float (&sh)[8]) const {
        const int c = c0 + it_sub[0];
#pragma unroll
        for (int i = 0; i < 8; ++i) { sc[i] = 1.f; sh[i] = 0.f; }
        if (sg.scale && c < sg.C) { load8f(sg.scale + c, sc); load8f(sg.shift + c, sh); }
    }
    __device__ __forceinline__ void load_code(const mcgen_seg_t& sg, int c0, int one_n, float (&cd)[8]) const {
        const int c = c0 + it_sub[0];
#pragma unroll
        for (int i = 0; i < 8; ++i) cd[i] = 1.f;
        if (sg.code && one_n >= 0 && c < sg.C) load8f(sg.code + (size_t)one_n * sg.C + c, cd);
    }
    __device__ __forceinline__ void write_pre(const mcgen_seg_t& sg, int c0, const int (&src)[NI], const int (&nn)[NI],
                                              const raw_t& raw, char* ldsA, int one_n,
                                              const float (&sc)[8], const float (&sh)[8], const float (&cd1)[8]) const {
        const int c = c0 + it_sub[0];
        const bool cok = c < sg.C;
        const bool code_once = sg.code && one_n >= 0;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            if (it_lds[k] < 0) continue;
            float v[8];
            unpack(raw[k], v);
            if (src[k] >= 0 && cok) {
                if (sg.scale) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], sc[i], sh[i]);
                }
                if (sg.relu) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
                }
                if (code_once) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= cd1[i];
                } else if (sg.code) {
                    float cd[8];
                    load8f(sg.code + (size_t)nn[k] * sg.C + c, cd);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= cd[i];
                }
            }
            E::store8(reinterpret_cast<T*>(ldsA + it_lds[k]), v);
        }
    }

    // non-pipelined form: item by item (load, prologue, LDS store) -- few live registers, so several
    // workgroups fit on a CU and hide each other's latency
    __device__ __forceinline__ void stage(const mcgen_seg_t& sg, int c0, char* ldsA) const {
        const T* xs = reinterpret_cast<const T*>(sg.x);
        const int c = c0 + it_sub[0];
        const bool cok = c < sg.C;
#pragma unroll
        for (int k = 0; k < NI; ++k) {
            if (it_lds[k] < 0) continue;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = 0.f;
            if (it_src[k] >= 0 && cok) {
                E::load8(xs + (size_t)it_src[k] + c0, v);
                if (sg.scale) {
                    float sc[8], sh[8];
                    load8f(sg.scale + c, sc); load8f(sg.shift + c, sh);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaf(v[i], sc[i], sh[i]);
                }
                if (sg.relu) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = fmaxf(v[i], 0.f);
                }
                if (sg.code) {
                    float cd[8];
                    load8f(sg.code + (size_t)it_n[k] * sg.C + c, cd);
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] *= cd[i];
                }
            }
            E::store8(reinterpret_cast<T*>(ldsA + it_lds[k]), v);
        }
    }
};

static inline int mcgen_patch_pixels(int BM, int H, int W, int ksize) {
    const int HW = H * W;
    int TI = 1, TH;
    if (BM <= HW) TH = BM / W; else { TI = BM / HW; TH = H; }
    const int halo = ksize >> 1;
    return TI * (TH + 2 * halo) * (W + 2 * halo);
}
