// Weight gradient of the fused convolution (gfx950 MFMA).
//
//   dW[co][tap][ci] = sum over pixels  dy[pixel][co] * prologue(x)[pixel + tap][ci]
//
// A workgroup owns 64 output channels x one 32-channel chunk of the input x all taps, and
// walks pixel tiles of 128 (the K dimension), staging per tile the same prologue-applied
// input window the forward kernel uses plus the dy tile.  Both MFMA operands are indexed
// [k = pixel][row/col = channel] in LDS, i.e. k is the slow index: bf16 fragments are read
// with ds_read_b64_tr_b16 (transposing read, 4 pixels x 16 channels per 16 lanes), fp32
// fragments with scalar reads.  Pixel groups (blockIdx.z) write separate fp32 slabs in the
// weight-image layout; mcgen_wgrad_reduce adds the slabs in a fixed order (deterministic).
#include "conv_tile.h"
#include <stdlib.h>
#include <algorithm>

namespace {

constexpr int WG_BM = 128;     // pixels per K tile
constexpr int WG_BCO = 64;     // output channels per workgroup
constexpr int WG_CI = 32;      // input channels per workgroup (one 32-channel chunk of the weight image)
constexpr int WG_NT = 256;

template <typename T> struct WgTraits;
template <> struct WgTraits<float> {
    static constexpr int APITCH = WG_CI * 4 + 16;        // rows stay 16-byte aligned for the staging stores
    static constexpr int DPITCH = WG_BCO * 4 + 16;
};
template <> struct WgTraits<bf16_t> {
    // pitches of 32 B x odd: the 8 pixel rows one half-wave touches in a transposing read fall on 8
    // distinct 32-byte bank groups (conflict-free with the k mapping below)
    static constexpr int APITCH = WG_CI * 2 + 32;        // 96 B
    static constexpr int DPITCH = WG_BCO * 2 + 32;       // 160 B
};

static __device__ __forceinline__ s16x4 lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(reinterpret_cast<uintptr_t>(p)));
}

// k mapping of one 32-pixel MFMA step: element j of lane group lg is tile pixel 16*(j>>2) + 4*lg + (j&3).
// (Any bijection works as long as both operands use it; this one makes each transposing read cover 16
// consecutive pixel rows, 8 per half-wave.)
// One operand fragment: 8 pixels (k) x 16 channels (lane & 15) from an LDS image laid out [pixel][channel].
template <typename T> struct KFrag;
template <> struct KFrag<bf16_t> {
    typedef bf16x8 frag;
    static constexpr int NOFF = 2;        // byte offset of the lane's row for each half (includes the 8-byte column part)
    static __device__ __forceinline__ frag read(const char* base, const int (&off)[NOFF], int imm) {
        union { bf16x8 v; s16x4 h[2]; } u;
        u.h[0] = lds_tr16(base + off[0] + imm);
        u.h[1] = lds_tr16(base + off[1] + imm);
        return u.v;
    }
};
template <> struct KFrag<float> {
    typedef f32x8 frag;
    static constexpr int NOFF = 8;
    static __device__ __forceinline__ frag read(const char* base, const int (&off)[NOFF], int imm) {
        f32x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = *reinterpret_cast<const float*>(base + off[j] + imm);
        return r;
    }
};


// XCD-aware block order.  Hardware deals workgroup ids round-robin over the 8 XCDs (each with its own L2); the blocks
// of one pixel split (all output-channel blocks x input-channel chunks) read the SAME x / dy tiles, so they should share
// an L2: the linear id is re-read as (xcd, slot) -> virtual id xcd * (total / 8) + slot, and (bx, by, bz) are decoded from
// the virtual id with bx, by fastest.  Falls back to the identity when the grid is not a multiple of 8.
struct WgIdx { int bx, by, bz; };
// Several layers of IDENTICAL shape in one launch (mcgen_wgrad_batch): blockIdx.z = layer * splits + split.  The kernels below
// take either one mcgen_wgrad_t or this table; everything that depends on the layer is read through wg_layer().
struct WgBatch { mcgen_wgrad_t l[MCGEN_WGRAD_MULTI_MAX]; int splits; };
static __device__ __forceinline__ const mcgen_wgrad_t& wg_layer(const mcgen_wgrad_t& a) { return a; }
static __device__ __forceinline__ const mcgen_wgrad_t& wg_layer(const WgBatch& a) { return a.l[blockIdx.z / a.splits]; }
static __device__ __forceinline__ int wg_splits(const mcgen_wgrad_t&) { return (int)gridDim.z; }
static __device__ __forceinline__ int wg_splits(const WgBatch& a) { return a.splits; }
static __device__ __forceinline__ WgIdx wg_remap(const WgBatch& a, int) {
    return WgIdx{(int)blockIdx.x, (int)blockIdx.y, (int)(blockIdx.z % a.splits)};
}
static __device__ __forceinline__ WgIdx wg_remap(const mcgen_wgrad_t&, int enabled) {
    WgIdx r{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
    const int gx = gridDim.x, gy = gridDim.y, total = gx * gy * (int)gridDim.z;
    if (!enabled || (total & 7)) return r;
    const int L = r.bx + gx * (r.by + gy * r.bz);
    const int v = (L & 7) * (total >> 3) + (L >> 3);
    r.bx = v % gx; r.by = (v / gx) % gy; r.bz = v / (gx * gy);
    return r;
}

// Workgroup = 64 output channels x 32 input channels x all taps.  Waves are a 2 x 2 grid:
// wave (a, b) owns output-channel fragments {2a, 2a+1} and input-channel block b for every tap, so per
// 32-pixel step it reads 2 dy fragments + NTAP window fragments for 2*NTAP MFMAs (small footprint:
// 72 accumulator registers, ~35 KB LDS -> several workgroups per CU hide each other's staging latency).
// LGW = log2(W) is a template parameter so that every tap offset is an instruction immediate.
template <typename T, int KS, int LGW, typename PA = mcgen_wgrad_t>
__global__ __launch_bounds__(WG_NT, 3)
void wgrad_kernel(const PA pa, const int a_bytes, const int m_tiles, const int xcd_map) {
    const mcgen_wgrad_t& p = wg_layer(pa);
    using E = Elem<T>;
    using M = Mma<T>;
    using TR = WgTraits<T>;
    using KF = KFrag<T>;
    constexpr int ESZ = E::BYTES, APITCH = TR::APITCH, DPITCH = TR::DPITCH;
    constexpr int NTAP = KS * KS;
    constexpr int NCF = WG_BCO / 32;                  // output-channel fragments per wave
    constexpr int NI = (WG_BM * 9 + WG_NT - 1) / WG_NT;
    constexpr int W = 1 << LGW, halo = KS >> 1, PC = W + 2 * halo;
    constexpr int KSTEPS = WG_BM / 32;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ldsA = smem;
    char* ldsD = smem + a_bytes;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int H = p.H, N = p.N;
    const WgIdx wi = wg_remap(pa, xcd_map);
    const int co0 = wi.bx * WG_BCO;
    const int q = wi.by;                            // input-channel chunk
    const int c0 = q * MCGEN_CK;
    const mcgen_seg_t sg = p.seg;
    const char* dy = reinterpret_cast<const char*>(p.dy);
    const int Hd = p.dy_ups ? (H >> 1) : H, Wd = p.dy_ups ? (W >> 1) : W;
    const int wa = wave >> 1, wb = wave & 1;          // output-channel half, input-channel block

    f32x4 acc[NTAP][NCF];
#pragma unroll
    for (int j = 0; j < NTAP; ++j)
#pragma unroll
        for (int cf = 0; cf < NCF; ++cf) acc[j][cf] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = (p.bias_slabs != nullptr) && (q == 0);
    float bsum = 0.f;                                  // thread (column tid&63, row quarter tid>>6)

    // tile-relative geometry is the same for every tile: fragment row offsets are computed once
    const Geo g0 = make_geo(WG_BM, 0, H, W);
    const int PR = g0.TH + 2 * halo;
    auto make_off = [&](int ks, int (&oa)[KF::NOFF], int (&od)[KF::NOFF]) {
#pragma unroll
        for (int j = 0; j < KF::NOFF; ++j) {
            // bf16: j = half, this lane supplies row q4 = l15>>2 of the 4-row block; fp32: j = element
            const int kk = (KF::NOFF == 2) ? (16 * j + 4 * lg + (l15 >> 2)) : (16 * (j >> 2) + 4 * lg + (j & 3));
            const int m = ks * 32 + kk;
            const int ti = m >> g0.lgTHW, rem = m & ((1 << g0.lgTHW) - 1);
            const int r = rem >> LGW, c = rem & (W - 1);
            const int colb = (KF::NOFF == 2) ? (l15 & 3) * 8 : l15 * 4;
            oa[j] = ((ti * PR + r) * PC + c) * APITCH + colb + wb * 16 * ESZ;
            od[j] = m * DPITCH + colb + wa * NCF * 16 * ESZ;
        }
    };
    constexpr bool PRE = (KF::NOFF == 2);              // bf16: 16 registers hold all steps' offsets; fp32 recomputes
    int offA[PRE ? KSTEPS : 1][KF::NOFF], offD[PRE ? KSTEPS : 1][KF::NOFF];
    if constexpr (PRE) {
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) make_off(ks, offA[ks], offD[ks]);
    }
    // dy staging items of this thread: (pixel m, 16-byte unit) -> source pixel offset is tile dependent,
    // the LDS offset is not
    constexpr int DUNITS = WG_BCO * ESZ / 16;           // 16-byte units per dy row (8 bf16 / 16 fp32)
    constexpr int DITEMS = WG_BM * DUNITS / WG_NT;
    static_assert(WG_BM * DUNITS % WG_NT == 0, "dy staging");

    PatchStager<T, WG_NT, NI, APITCH> stager;
    stager.setup_static(KS, g0, W, tid);
    // tile walk of this split: all tiles with stride gridDim.z, or (p.halves) one half of the tiles with stride gridDim.z / 2
    const int zs = p.halves ? (wg_splits(pa) >> 1) : wg_splits(pa);
    const int mt = p.halves ? (m_tiles >> 1) : m_tiles;
    const int t_lo = p.halves ? (wi.bz / zs) * mt : 0;
    for (int tile = t_lo + wi.bz % zs; tile < t_lo + mt; tile += zs) {
        const Geo g = make_geo(WG_BM, tile, H, W);
        stager.bind(sg, g, N, H, W);
        __syncthreads();                                        // previous tile's reads are done
        if constexpr (KS == 1) {
            // all global loads of the window first, then prologue + LDS stores: one exposed round trip instead of one per
            // item (measured: -13 % on 1x1, but +30 % on 3x3 at 8x8, where the extra live registers cost a workgroup per CU)
            typename PatchStager<T, WG_NT, NI, APITCH>::raw_t raw;
            stager.load(sg, c0, raw);
            stager.write(sg, c0, raw, ldsA, (g.TI == 1 && g.n0 < N) ? g.n0 : -1);
        } else {
            stager.stage(sg, c0, ldsA);
        }
        // dy tile: [pixel m][64 co], raw 16-byte copies
#pragma unroll
        for (int k = 0; k < DITEMS; ++k) {
            const int it = tid + k * WG_NT;
            const int u = it % DUNITS, m = it / DUNITS;
            const int ti = m >> g.lgTHW, rem = m & ((1 << g.lgTHW) - 1);
            const int r = rem >> LGW, c = rem & (W - 1);
            const int n = g.n0 + ti, h = g.h0 + r;
            u32x4 v = {0u, 0u, 0u, 0u};
            const int cob = (co0 * ESZ + u * 16);                // byte offset of this unit inside the dy pixel row
            if (n < N && cob < p.Cdy * ESZ) {
                const int hd = p.dy_ups ? (h >> 1) : h, wd = p.dy_ups ? (c >> 1) : c;
                v = *reinterpret_cast<const u32x4*>(dy + ((size_t)(n * Hd + hd) * Wd + wd) * p.Cdy * ESZ + cob);
            }
            *reinterpret_cast<u32x4*>(ldsD + m * DPITCH + u * 16) = v;
        }
        __syncthreads();
        if (do_bias) {
            const int col = tid & 63, part = tid >> 6;
#pragma unroll 8
            for (int r = 0; r < WG_BM / 4; ++r)
                bsum += E::to_f(*reinterpret_cast<const T*>(ldsD + (part * (WG_BM / 4) + r) * DPITCH + col * ESZ));
        }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            if constexpr (!PRE) make_off(ks, offA[0], offD[0]);
            const int (&oA)[KF::NOFF] = offA[PRE ? ks : 0];
            const int (&oD)[KF::NOFF] = offD[PRE ? ks : 0];
            typename M::frag dfrag[NCF];
#pragma unroll
            for (int cf = 0; cf < NCF; ++cf) dfrag[cf] = KF::read(ldsD, oD, cf * 16 * ESZ);
#pragma unroll
            for (int j = 0; j < NTAP; ++j) {
                constexpr int dummy = 0; (void)dummy;
                const int tapoff = ((j / KS) * PC + (j % KS)) * APITCH;       // compile-time
                const typename M::frag afrag = KF::read(ldsA, oA, tapoff);
#pragma unroll
                for (int cf = 0; cf < NCF; ++cf) M::run(dfrag[cf], afrag, acc[j][cf]);
            }
        }
    }

    if (do_bias) {
        // four row-quarter partial sums per column; mcgen_wgrad_reduce adds all splits*4 rows in order
        const int col = tid & 63, part = tid >> 6;
        if (co0 + col < p.Cout_w) p.bias_slabs[((size_t)wi.bz * 4 + part) * p.Cout_w + co0 + col] = bsum;
    }
    // slab[z][q][tap][co][32]: lane holds D[co = 4*lg + r][ci = l15] of block (tap j, co fragment cf)
    const int nchunk = (sg.C + MCGEN_CK - 1) / MCGEN_CK;
    const size_t slab_elems = (size_t)nchunk * NTAP * p.Cout_w * MCGEN_CK;
    float* out = p.slabs + (size_t)wi.bz * slab_elems;
#pragma unroll
    for (int j = 0; j < NTAP; ++j) {
        const int col = wb * 16 + l15;
#pragma unroll
        for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + (wa * NCF + cf) * 16 + lg * 4 + r;
                if (co < p.Cout_w)
                    out[(((size_t)q * NTAP + j) * p.Cout_w + co) * MCGEN_CK + col] = acc[j][cf][r];
            }
    }
}

// ---- producer / consumer form -------------------------------------------------------------------------
// The staging of a tile (global loads, prologue VALU, LDS stores) costs several times the MFMA time of
// the tile, so the workgroup is split by role: waves 4-7 (producers) stage tile i+1 into the idle LDS
// buffers while waves 0-3 (consumers) run the transposing reads + MFMAs of tile i; one barrier per tile.
// VALU and MFMA issue from different waves of a SIMD overlap, which the single-role kernel cannot do.
// NCH > 1 (1x1 convolutions only): the workgroup owns NCH consecutive 32-channel chunks of the input, staged side by
// side and walked like taps, so one staged dy tile feeds NCH times as many MFMAs (a 1x1 weight gradient is a plain
// GEMM whose 64 x 32 output tile would otherwise re-read dy Cin/32 times: measured 132 TFLOP/s at 512 x 512).
// DDMA (bf16, whole tiles only): the dy tile goes global -> LDS by LDS-DMA instead of through producer registers: rows
// of 128 bytes without padding, 16-byte units XOR-swizzled by ((row >> 1) & 3) << 1 (applied on the SOURCE address) so
// that the transposing reads stay conflict-free; the producers' registers then hold only the x window.
template <typename T, int KS, int LGW, int NCH, bool DDMA, typename PA = mcgen_wgrad_t>
__global__ __launch_bounds__(2 * WG_NT, 2)
void wgrad_pc_kernel(const PA pa, const int a_bytes, const int m_tiles, const int xcd_map) {
    const mcgen_wgrad_t& p = wg_layer(pa);
    using E = Elem<T>;
    using M = Mma<T>;
    using TR = WgTraits<T>;
    using KF = KFrag<T>;
    constexpr int ESZ = E::BYTES, APITCH = TR::APITCH, DPITCH = TR::DPITCH;
    static_assert(NCH == 1 || KS == 1, "chunk groups are for 1x1 convolutions");
    constexpr int NTAP = KS * KS;
    constexpr int NV = NTAP * NCH;                    // accumulator planes: taps, or chunks of a 1x1 group
    constexpr int NCF = WG_BCO / 32;
    constexpr int NI = (KS == 1) ? (WG_BM * 4 + WG_NT - 1) / WG_NT : (WG_BM * 9 + WG_NT - 1) / WG_NT;
    constexpr int W = 1 << LGW, halo = KS >> 1, PC = W + 2 * halo;
    constexpr int KSTEPS = WG_BM / 32;
    constexpr int DROW = DDMA ? WG_BCO * ESZ : DPITCH;   // bytes per dy row in LDS
    constexpr int D_BYTES = WG_BM * DROW;
    static_assert(!DDMA || (ESZ == 2 && WG_BCO * ESZ == 128), "dy LDS-DMA layout is for bf16");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int a_tile = a_bytes * NCH;               // one tile's windows: NCH chunk regions of a_bytes
    char* const ldsA0 = smem;                       // [2][a_tile]
    char* const ldsD0 = smem + 2 * a_tile;          // [2][D_BYTES]

    const int tid = threadIdx.x;
    const bool producer = __builtin_amdgcn_readfirstlane(tid >> 8) != 0;
    const int rtid = tid & 255;                     // thread index inside its role group
    const int lane = tid & 63, wave = (tid >> 6) & 3;
    const int l15 = lane & 15, lg = lane >> 4;
    const int H = p.H, N = p.N;
    const WgIdx wi = wg_remap(pa, xcd_map);
    const int co0 = wi.bx * WG_BCO;
    const int q = wi.by;
    const int c0 = q * NCH * MCGEN_CK;
    const mcgen_seg_t sg = p.seg;
    const char* dy = reinterpret_cast<const char*>(p.dy);
    const int Hd = p.dy_ups ? (H >> 1) : H, Wd = p.dy_ups ? (W >> 1) : W;
    const int wa = wave >> 1, wb = wave & 1;
    const Geo g0 = make_geo(WG_BM, 0, H, W);
    const int PR = g0.TH + 2 * halo;
    // tile walk of this split: all tiles with stride gridDim.z, or (p.halves) one half of the tiles with stride gridDim.z / 2
    const int zs = p.halves ? (wg_splits(pa) >> 1) : wg_splits(pa);
    const int mt = p.halves ? (m_tiles >> 1) : m_tiles;
    const int t_first = (p.halves ? (wi.bz / zs) * mt : 0) + wi.bz % zs;
    const int cnt = (mt - wi.bz % zs + zs - 1) / zs;                          // tiles of this workgroup
    const bool do_bias = (p.bias_slabs != nullptr) && (q == 0);
    constexpr int DUNITS = WG_BCO * ESZ / 16;
    constexpr int DITEMS = WG_BM * DUNITS / WG_NT;

    if (producer) {
        PatchStager<T, WG_NT, NI, APITCH> stager;
        stager.setup_static(KS, g0, W, rtid);
        // Two register sets: while the tile that is due next goes through the prologue into LDS, the loads
        // of the tile after it are already in flight (prefetch distance of two tiles).
        struct TileRegs {
            int src[NI], nn[NI];
            typename PatchStager<T, WG_NT, NI, APITCH>::raw_t raw[NCH];
            u32x4 d[DITEMS];
            float cd[NCH][8];                        // the tile's code row(s) when it lies inside one image
            int one_n;                               // that image, else -1 (code rows per item)
        };
        // the BN affine depends on the chunk only: once per workgroup (see PatchStager::write_pre)
        float sc[NCH][8], sh[NCH][8];
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch) stager.load_affine(sg, c0 + ch * MCGEN_CK, sc[ch], sh[ch]);
        auto fetch = [&](int i, TileRegs& r) {
            const int tile = t_first + i * zs;
            const Geo g = make_geo(WG_BM, tile, H, W);
            stager.bind_into(sg, g, N, H, W, r.src, r.nn);
            r.one_n = (g.TI == 1 && g.n0 < N) ? g.n0 : -1;
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) stager.load_code(sg, c0 + ch * MCGEN_CK, r.one_n, r.cd[ch]);
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) stager.load_ext(sg, c0 + ch * MCGEN_CK, r.src, r.raw[ch]);
            if constexpr (!DDMA)
#pragma unroll
            for (int k = 0; k < DITEMS; ++k) {
                const int it = rtid + k * WG_NT;
                const int u = it % DUNITS, m = it / DUNITS;
                const int ti = m >> g.lgTHW, rem = m & ((1 << g.lgTHW) - 1);
                const int r_ = rem >> LGW, c = rem & (W - 1);
                const int n = g.n0 + ti, h = g.h0 + r_;
                r.d[k] = u32x4{0u, 0u, 0u, 0u};
                const int cob = (co0 * ESZ + u * 16);
                if (n < N && cob < p.Cdy * ESZ) {
                    const int hd = p.dy_ups ? (h >> 1) : h, wd = p.dy_ups ? (c >> 1) : c;
                    r.d[k] = *reinterpret_cast<const u32x4*>(dy + ((size_t)(n * Hd + hd) * Wd + wd) * p.Cdy * ESZ + cob);
                }
            }
        };
        auto commit = [&](int i, const TileRegs& r) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch)
                stager.write_pre(sg, c0 + ch * MCGEN_CK, r.src, r.nn, r.raw[ch], ldsA0 + (i & 1) * a_tile + ch * a_bytes, r.one_n,
                                 sc[ch], sh[ch], r.cd[ch]);
            if constexpr (!DDMA) {
            char* ldsD = ldsD0 + (i & 1) * D_BYTES;
#pragma unroll
            for (int k = 0; k < DITEMS; ++k) {
                const int it = rtid + k * WG_NT;
                *reinterpret_cast<u32x4*>(ldsD + (it / DUNITS) * DPITCH + (it % DUNITS) * 16) = r.d[k];
            }
            }
        };
        // dy tile i by LDS-DMA: 16 instructions of 64 lanes x 16 bytes (8 rows each), 4 per producer wave
        auto dma_dy = [&](int i) {
            if constexpr (DDMA) {
                const int tile = t_first + i * zs;
                const Geo g = make_geo(WG_BM, tile, H, W);
                char* base = ldsD0 + (i & 1) * D_BYTES;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int d = wave * 4 + k;
                    const int m = d * 8 + (lane >> 3), slot = lane & 7;
                    int u = slot ^ (((m >> 1) & 3) << 1);
                    if ((co0 + u * 8 + 8) > p.Cdy) u = 0;              // beyond the dy pitch: any in-bounds unit (those rows are discarded)
                    const int ti = m >> g.lgTHW, rem = m & ((1 << g.lgTHW) - 1);
                    const int r_ = rem >> LGW, c = rem & (W - 1);
                    const int n = g.n0 + ti, h = g.h0 + r_;
                    const int hd = p.dy_ups ? (h >> 1) : h, wd = p.dy_ups ? (c >> 1) : c;
                    const char* src = dy + ((size_t)(n * Hd + hd) * Wd + wd) * p.Cdy * ESZ + co0 * ESZ + u * 16;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                     (__attribute__((address_space(3))) void*)(base + d * 1024), 16, 0, 0);
                }
            }
        };
        TileRegs ra, rb;
        if (cnt > 0) { dma_dy(0); fetch(0, ra); commit(0, ra); }
        if (cnt > 1) fetch(1, ra);
        for (int i = 0; i < cnt; i += 2) {
            if constexpr (DDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the dy tile of step i has landed
            __syncthreads();                         // tile i published; the buffers of tile i-1 are free
            if (i + 1 < cnt) dma_dy(i + 1);
            if (i + 2 < cnt) fetch(i + 2, rb);
            if (i + 1 < cnt) commit(i + 1, ra);
            if (i + 1 >= cnt) break;
            if constexpr (DDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                         // tile i+1 published
            if (i + 2 < cnt) dma_dy(i + 2);
            if (i + 3 < cnt) fetch(i + 3, ra);
            if (i + 2 < cnt) commit(i + 2, rb);
        }
        if constexpr (DDMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                             // matches the consumers' final barrier
        return;
    }

    // ---- consumers ---------------------------------------------------------------------------------------
    f32x4 acc[NV][NCF];
#pragma unroll
    for (int j = 0; j < NV; ++j)
#pragma unroll
        for (int cf = 0; cf < NCF; ++cf) acc[j][cf] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    auto make_off = [&](int ks, int (&oa)[KF::NOFF], int (&od)[KF::NOFF], int (&od1)[KF::NOFF]) {
#pragma unroll
        for (int j = 0; j < KF::NOFF; ++j) {
            const int kk = (KF::NOFF == 2) ? (16 * j + 4 * lg + (l15 >> 2)) : (16 * (j >> 2) + 4 * lg + (j & 3));
            const int m = ks * 32 + kk;
            const int ti = m >> g0.lgTHW, rem = m & ((1 << g0.lgTHW) - 1);
            const int r = rem >> LGW, c = rem & (W - 1);
            const int colb = (KF::NOFF == 2) ? (l15 & 3) * 8 : l15 * 4;
            oa[j] = ((ti * PR + r) * PC + c) * APITCH + colb + wb * 16 * ESZ;
            if constexpr (DDMA) {
                // swizzled rows: byte column -> (16-byte unit ^ f(row)) * 16 + byte inside the unit, one offset per co fragment
                const int f = ((m >> 1) & 3) << 1;
                const int col0 = colb + wa * NCF * 16 * ESZ, col1 = col0 + 16 * ESZ;
                od[j] = m * DROW + (((col0 >> 4) ^ f) << 4) + (col0 & 15);
                od1[j] = m * DROW + (((col1 >> 4) ^ f) << 4) + (col1 & 15);
            } else {
                od[j] = m * DPITCH + colb + wa * NCF * 16 * ESZ;
                od1[j] = od[j] + 16 * ESZ;
            }
        }
    };
    static_assert(NCF == 2, "two output-channel fragments per wave");
    constexpr bool PRE = (KF::NOFF == 2);
    int offA[PRE ? KSTEPS : 1][KF::NOFF], offD[PRE ? KSTEPS : 1][KF::NOFF], offD1[(PRE && DDMA) ? KSTEPS : 1][KF::NOFF];
    if constexpr (PRE) {
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) make_off(ks, offA[ks], offD[ks], offD1[DDMA ? ks : 0]);
    }
    for (int i = 0; i < cnt; ++i) {
        __syncthreads();
        const char* ldsA = ldsA0 + (i & 1) * a_tile;
        const char* ldsD = ldsD0 + (i & 1) * D_BYTES;
        if (do_bias) {
            const int col = rtid & 63, part = rtid >> 6;
#pragma unroll 8
            for (int r = 0; r < WG_BM / 4; ++r)
            {
                const int row = part * (WG_BM / 4) + r;
                const int cb = col * ESZ;
                const int off = DDMA ? row * DROW + ((((cb >> 4) ^ (((row >> 1) & 3) << 1))) << 4) + (cb & 15) : row * DPITCH + cb;
                bsum += E::to_f(*reinterpret_cast<const T*>(ldsD + off));
            }
        }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            if constexpr (!PRE) make_off(ks, offA[0], offD[0], offD1[0]);
            const int (&oA)[KF::NOFF] = offA[PRE ? ks : 0];
            const int (&oD)[KF::NOFF] = offD[PRE ? ks : 0];
            typename M::frag dfrag[NCF];
            if constexpr (DDMA) {
                dfrag[0] = KF::read(ldsD, oD, 0);
                dfrag[1] = KF::read(ldsD, offD1[PRE ? ks : 0], 0);
            } else {
#pragma unroll
                for (int cf = 0; cf < NCF; ++cf) dfrag[cf] = KF::read(ldsD, oD, cf * 16 * ESZ);
            }
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                // 3x3: tap offset inside the halo window (compile-time); 1x1 chunk group: the chunk's region
                const int tapoff = (NCH == 1) ? ((j / KS) * PC + (j % KS)) * APITCH : j * a_bytes;
                const typename M::frag afrag = KF::read(ldsA, oA, tapoff);
#pragma unroll
                for (int cf = 0; cf < NCF; ++cf) M::run(dfrag[cf], afrag, acc[j][cf]);
            }
        }
    }
    __syncthreads();                                 // last barrier shared with the producers (they exit after it)
    // slab[z][q][tap][co][32]
    const int nchunk = (sg.C + MCGEN_CK - 1) / MCGEN_CK;
    const size_t slab_elems = (size_t)nchunk * NTAP * p.Cout_w * MCGEN_CK;
    float* out = p.slabs + (size_t)wi.bz * slab_elems;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int col = wb * 16 + l15;
        // plane j is tap j of chunk q (3x3) or the single tap of chunk q*NCH + j (1x1 chunk group)
        const int qc = (NCH == 1) ? q : q * NCH + j, tp = (NCH == 1) ? j : 0;
        if (qc >= nchunk) continue;
#pragma unroll
        for (int cf = 0; cf < NCF; ++cf)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = co0 + (wa * NCF + cf) * 16 + lg * 4 + r;
                if (co < p.Cout_w)
                    out[(((size_t)qc * NTAP + tp) * p.Cout_w + co) * MCGEN_CK + col] = acc[j][cf][r];
            }
    }
    if (do_bias) {
        const int col = rtid & 63, part = rtid >> 6;
        if (co0 + col < p.Cout_w) p.bias_slabs[((size_t)wi.bz * 4 + part) * p.Cout_w + co0 + col] = bsum;
    }
}

// ---- "ring" form (bf16, tiles inside one image) --------------------------------------------------------------------
// What bounds wgrad_pc_kernel on the 256 -> 256, 32x32 layer (2.08 us per 128-pixel tile against 0.48 us of MFMA; sides
// removed one at a time: consumers alone 1.7 us, producers alone 1.65 us):
//  * producers: a tile's loads were issued one tile ahead, so a tile cost a memory round trip (a deeper REGISTER pipeline
//    does not survive the compiler: it puts vmcnt(0) at the loop head).  Here tiles travel by LDS-DMA into rings -- the
//    raw window units of a tile into one of three slots, its dy tile (unpadded rows of 128 B, 16-byte units XOR-swizzled
//    on the SOURCE side so that the transposing reads stay conflict-free) into one of four, its code row into a third --
//    THREE tiles ahead, with a counted vmcnt (every wave issues the same number of DMA instructions per tile: the index
//    of a tile past the end is clamped).  The prologue is then an LDS -> LDS pass over the thread's own units, and
//    everything tile-independent is a per-thread constant (LDS offset, element offset from the tile's first pixel, row
//    offset): per tile an item costs a row-range test and an address add.
//  * consumers: EIGHT consumer waves, two per SIMD (measured: 628 -> 641 TFLOP/s against four consumer waves; the
//    SIMD's vector issue port, shared with its producer wave, is the kernel's limit), each 32 output channels x 16 input
//    channels over HALF of the tile's pixels; the halves are added once, through LDS, after the last tile.  Tried on
//    the way and neutral: all 64 output channels per wave (fewer LDS reads per MFMA), reading fragments 2-6 taps ahead
//    of their MFMAs (sched_group_barrier) -- LDS is ~20 % busy, it was never the limit.
// Measured 598 -> 641 TFLOP/s on that layer (525 -> 576, 506 -> 539 on the next two).
constexpr int WG_DSLOT = WG_BM * WG_BCO * 2;      // one dy tile, unpadded bf16 rows
constexpr int WG_ND = 4, WG_NR = 3;               // ring depths: dy tiles, raw windows
template <int KS, int LGW, typename PA = mcgen_wgrad_t>
__global__ __launch_bounds__(3 * WG_NT, 1)
void wgrad_ring_kernel(const PA pa, const int a_bytes, const int m_tiles, const int xcd_map) {
    const mcgen_wgrad_t& p = wg_layer(pa);
    typedef bf16_t T;
    using E = Elem<T>;
    using M = Mma<T>;
    using TR = WgTraits<T>;
    using KF = KFrag<T>;
    constexpr int ESZ = 2, APITCH = TR::APITCH;
    constexpr int NTAP = KS * KS, NV = NTAP, NCF = WG_BCO / 32;      // output-channel fragments per consumer wave
    constexpr int W = 1 << LGW, halo = KS >> 1, PC = W + 2 * halo;
    constexpr int TH = WG_BM / W, PR = TH + 2 * halo, PP = PR * PC;     // tile rows, window rows, window pixels
    static_assert(WG_BM % W == 0 && TH >= 1, "tiles inside one image");
    constexpr int NIX = (PP * 4 + WG_NT - 1) / WG_NT;                  // raw units per producer thread
    constexpr int RSLOT = NIX * WG_NT * 16;
    constexpr int LD = NIX + 4;                                        // DMA instructions per producer wave and tile
    constexpr int KSTEPS = WG_BM / 32, KSW = KSTEPS / 2;
    constexpr int DROW = WG_BCO * ESZ;                                 // 128-byte dy rows

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsA0 = smem;                                          // [2][a_bytes]   staged windows
    char* const ldsR0 = smem + 2 * a_bytes;                            // [WG_NR][RSLOT] raw window units
    char* const ldsD0 = ldsR0 + WG_NR * RSLOT;                         // [WG_ND][WG_DSLOT]
    char* const ldsC0 = ldsD0 + WG_ND * WG_DSLOT;                      // [WG_NR][4 waves][32 floats] code rows (per producer wave)

    const int tid = threadIdx.x;
    // waves 0-7 consume (two per SIMD), waves 8-11 produce
    const bool producer = __builtin_amdgcn_readfirstlane(tid >> 9) != 0;
    const int rtid = tid & 255;
    const int lane = tid & 63, cw = __builtin_amdgcn_readfirstlane(tid >> 6), wave = cw & 3;
    const int l15 = lane & 15, lg = lane >> 4;
    const int H = p.H;
    const WgIdx wi = wg_remap(pa, xcd_map);
    const int co0 = wi.bx * WG_BCO;
    const int q = wi.by;
    const int c0 = q * MCGEN_CK;
    const mcgen_seg_t sg = p.seg;
    const int zs = p.halves ? (wg_splits(pa) >> 1) : wg_splits(pa);
    const int mt = p.halves ? (m_tiles >> 1) : m_tiles;
    const int t_first = (p.halves ? (wi.bz / zs) * mt : 0) + wi.bz % zs;
    const int cnt = (mt - wi.bz % zs + zs - 1) / zs;                   // tiles of this workgroup (>= 1)
    const bool do_bias = (p.bias_slabs != nullptr) && (q == 0);
    const int lgHW = 31 - __builtin_clz(H << LGW);
    const int kh = (cw >> 2) & 1, wa = (cw >> 1) & 1, wb = cw & 1;     // consumers: pixel half, output-channel half, input-channel half

    f32x4 acc[NV][NCF];
    float bsum = 0.f;

    if (producer) {
        const int sub = (rtid & 3) * 8;                                // channel offset inside the chunk
        const bool cok = c0 + sub < sg.C;
        int x_lds[NIX], x_off[NIX], x_dh[NIX];
#pragma unroll
        for (int k = 0; k < NIX; ++k) {
            const int pp = (rtid + k * WG_NT) >> 2;
            const int pr = pp / PC, pc = pp - pr * PC;
            const int w = pc - halo;
            const bool item = pp < PP;
            x_lds[k] = item ? pp * APITCH + sub * ESZ : -1;
            x_dh[k] = (item && cok && w >= 0 && w < W) ? pr - halo : (1 << 20);   // never passes the row test
            // nearest-x2 upsampled input: the source pixel of (h, w) is (h >> 1, w >> 1); tiles start on even rows, so the
            // halving splits into a tile part and this constant part
            x_off[k] = (sg.ups ? (((pr - halo) >> 1) * (W >> 1) + (w >> 1)) : ((pr - halo) * W + w)) * sg.C + sub;
        }
        float sc[8], sh[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { sc[i] = 1.f; sh[i] = 0.f; }
        if (sg.scale && cok) { load8f(sg.scale + c0 + sub, sc); load8f(sg.shift + c0 + sub, sh); }
        const char* xs = reinterpret_cast<const char*>(sg.x);
        const char* dyb = reinterpret_cast<const char*>(p.dy);
        const size_t dpix = (size_t)p.Cdy * ESZ;
        // dy units of this lane: row (piece * 8 + lane / 8), 16-byte unit (lane % 8) ^ swizzle(row) on the source side
        size_t d_src[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int m = (wave * 4 + k) * 8 + (lane >> 3);
            int u = (lane & 7) ^ (((m >> 1) & 3) << 1);
            if ((co0 + u * 8 + 8) > p.Cdy) u = 0;                      // beyond the dy pitch: any in-bounds unit (rows dropped at the end)
            const int mp = p.dy_ups ? ((m >> LGW) >> 1) * (W >> 1) + ((m & (W - 1)) >> 1) : m;    // dy through an upsample: as x
            d_src[k] = (size_t)mp * dpix + (size_t)co0 * ESZ + u * 16;
        }
        // first source pixel of a tile (x / dy), directly or through the x2 upsample
        auto first_pixel = [&](int pix0, bool up) -> size_t {
            if (!up) return (size_t)pix0;
            const int n0 = pix0 >> lgHW, h0 = (pix0 & ((1 << lgHW) - 1)) >> LGW;
            return ((size_t)n0 * (H >> 1) + (h0 >> 1)) * (W >> 1);
        };
        auto tile_of = [&](int i) { return t_first + (i < cnt ? i : cnt - 1) * zs; };
        auto dma = [&](int i) {                                        // tile i -> R[i % 3], D[i % 4]; LD instructions per wave
            const int pix0 = tile_of(i) * WG_BM;
            const int h0 = (pix0 & ((1 << lgHW) - 1)) >> LGW;
            const char* xb = xs + (first_pixel(pix0, sg.ups) * sg.C + c0) * ESZ;
            char* rs = ldsR0 + (i % WG_NR) * RSLOT + wave * 1024;
#pragma unroll
            for (int k = 0; k < NIX; ++k) {
                const bool ok = (unsigned)(h0 + x_dh[k]) < (unsigned)H;
                const char* src = ok ? xb + (ptrdiff_t)x_off[k] * ESZ : xs;        // outside: any in-bounds address
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(rs + k * (WG_NT * 16)), 16, 0, 0);
            }
            const char* db = dyb + first_pixel(pix0, p.dy_ups) * dpix;
            char* ds = ldsD0 + (i % WG_ND) * WG_DSLOT + wave * 4096;
#pragma unroll
            for (int k = 0; k < 4; ++k)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(db + d_src[k]),
                                                 (__attribute__((address_space(3))) void*)(ds + k * 1024), 16, 0, 0);
            // the tile's code row (32 floats of this chunk) also by DMA, one copy per wave: an ordinary load here would
            // queue behind the DMAs just issued and its wait would drain them
            if (sg.code && lane < 8) {
                const int n0 = pix0 >> lgHW;
                const int cc = c0 + lane * 4;
                const float* src = sg.code + (size_t)n0 * sg.C + (cc + 4 <= sg.C ? cc : 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(ldsC0 + ((i % WG_NR) * 4 + wave) * 128), 16, 0, 0);
            }
        };
        // R[i % 3] -> A[i & 1]: v -> max(v * sc' + sh', relu ? 0 : -inf) with the tile's code row folded into the affine
        // (sc' = sc * code, sh' = sh * code; the tile lies inside one image) -- the SIMD's vector issue port is this kernel's
        // limit, and the fold takes the per-element code multiply out of the item loop.  In front of a ReLU the fold needs
        // code >= 0 (always true for MultimodalController codes, modules.py:73); a wave that sees a negative entry takes the
        // literal form for that tile.  Out-of-image rows are zeroed on the packed words.
        const float relu_lo = sg.relu ? 0.f : -__builtin_inff();
        auto prologue = [&](int i) {
            const int pix0 = tile_of(i) * WG_BM;
            const int h0 = (pix0 & ((1 << lgHW) - 1)) >> LGW;
            float scc[8], shc[8], cd[8];
            bool neg = false;
#pragma unroll
            for (int e = 0; e < 8; ++e) { scc[e] = sc[e]; shc[e] = sh[e]; cd[e] = 1.f; }
            if (sg.code) {
                load8f(reinterpret_cast<const float*>(ldsC0 + ((i % WG_NR) * 4 + wave) * 128) + sub, cd);
#pragma unroll
                for (int e = 0; e < 8; ++e) neg = neg || (cd[e] < 0.f);
            }
            const char* rs = ldsR0 + (i % WG_NR) * RSLOT + rtid * 16;
            char* ldsA = ldsA0 + (i & 1) * a_bytes;
            if (__builtin_expect(sg.relu && __any(neg), 0)) {          // wave-uniform; never taken for MultimodalController codes
#pragma unroll 1
                for (int k = 0; k < NIX; ++k) {
                    if (x_lds[k] < 0) continue;
                    float v[8];
                    E::load8(reinterpret_cast<const T*>(rs + k * (WG_NT * 16)), v);
                    const bool ok = (unsigned)(h0 + x_dh[k]) < (unsigned)H;
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = ok ? fmaxf(fmaf(v[e], sc[e], sh[e]), relu_lo) * cd[e] : 0.f;
                    E::store8(reinterpret_cast<T*>(ldsA + x_lds[k]), v);
                }
                return;
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) { scc[e] *= cd[e]; shc[e] *= cd[e]; }
#pragma unroll
            for (int k = 0; k < NIX; ++k) {
                if (x_lds[k] < 0) continue;
                const u32x4 r = *reinterpret_cast<const u32x4*>(rs + k * (WG_NT * 16));
                union { bf16x8 h; u32x4 w; } o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v0 = fmaxf(fmaf(__uint_as_float(r[e] << 16), scc[2 * e], shc[2 * e]), relu_lo);
                    const float v1 = fmaxf(fmaf(__uint_as_float(r[e] & 0xffff0000u), scc[2 * e + 1], shc[2 * e + 1]), relu_lo);
                    o.h[2 * e] = (bf16_t)v0; o.h[2 * e + 1] = (bf16_t)v1;
                }
                const bool ok = (unsigned)(h0 + x_dh[k]) < (unsigned)H;
                if (!ok) o.w = u32x4{0u, 0u, 0u, 0u};
                *reinterpret_cast<u32x4*>(ldsA + x_lds[k]) = o.w;
            }
        };
        auto landed = [&]() {                                          // all but the two youngest tiles' DMAs are done
            if (sg.code) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * LD + 2) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * LD) : "memory");
        };
        dma(0); dma(1); dma(2);
        landed();
        prologue(0);
        for (int i = 0; i < cnt; ++i) {
            __syncthreads();                                           // tile i published; slots of tile i-1 are free
            dma(i + 3);
            landed();                                                  // tile i+1 has landed (this wave's units)
            if (i + 1 < cnt) prologue(i + 1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");               // the clamped extra tiles: nothing lands after this
    } else {
        // ---- consumers -----------------------------------------------------------------------------------
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int cf = 0; cf < NCF; ++cf) acc[j][cf] = f32x4{0.f, 0.f, 0.f, 0.f};
        // per k step: window offsets (two halves of a fragment) and swizzled dy offsets (per output-channel fragment)
        int offA[KSW][2], offD[KSW][NCF][2];
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = (kh * KSW + ks) * 32 + 16 * j + 4 * lg + (l15 >> 2);
                const int r = m >> LGW, c = m & (W - 1);
                const int colb = (l15 & 3) * 8;
                offA[ks][j] = (r * PC + c) * APITCH + colb + wb * 16 * ESZ;
                const int fh = (m >> 1) & 3;
#pragma unroll
                for (int cf = 0; cf < NCF; ++cf)
                    offD[ks][cf][j] = m * DROW + (((colb >> 4) + 2 * ((wa * NCF + cf) ^ fh)) << 4) + (colb & 15);
            }
        for (int i = 0; i < cnt; ++i) {
            __syncthreads();
            const char* ldsA = ldsA0 + (i & 1) * a_bytes;
            const char* ldsD = ldsD0 + (i % WG_ND) * WG_DSLOT;
            if (do_bias && tid < WG_NT) {
                const int col = rtid & 63, part = rtid >> 6;
#pragma unroll 8
                for (int r = 0; r < WG_BM / 4; ++r) {
                    const int row = part * (WG_BM / 4) + r;
                    const int cb = col * ESZ;
                    bsum += E::to_f(*reinterpret_cast<const T*>(ldsD + row * DROW + ((((cb >> 4) ^ (((row >> 1) & 3) << 1))) << 4) + (cb & 15)));
                }
            }
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks) {
                typename M::frag dfrag[NCF];
#pragma unroll
                for (int cf = 0; cf < NCF; ++cf) dfrag[cf] = KF::read(ldsD, offD[ks][cf], 0);
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int tapoff = ((j / KS) * PC + (j % KS)) * APITCH;
                    const typename M::frag afrag = KF::read(ldsA, offA[ks], tapoff);
#pragma unroll
                    for (int cf = 0; cf < NCF; ++cf) M::run(dfrag[cf], afrag, acc[j][cf]);
                }
            }
        }
    }
    __syncthreads();                                 // every tile is consumed and every DMA has landed: LDS is free
    // the two pixel halves meet in LDS: waves 2/3 park their accumulators, waves 0/1 add them and write the slab
    f32x4* park = reinterpret_cast<f32x4*>(smem);     // [wa][wb][plane][cf][lane]: NV * 8 KB
    const int pw = wa * 2 + wb;
    if (!producer && kh == 1) {
#pragma unroll
        for (int j = 0; j < NV; ++j)
#pragma unroll
            for (int cf = 0; cf < NCF; ++cf) park[((pw * NV + j) * NCF + cf) * 64 + lane] = acc[j][cf];
    }
    __syncthreads();
    if (producer) return;
    if (kh == 0) {
        const int nchunk = (sg.C + MCGEN_CK - 1) / MCGEN_CK;
        const size_t slab_elems = (size_t)nchunk * NTAP * p.Cout_w * MCGEN_CK;
        float* out = p.slabs + (size_t)wi.bz * slab_elems;      // slab[z][q][tap][co][32]
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int col = wb * 16 + l15;
#pragma unroll
            for (int cf = 0; cf < NCF; ++cf) {
                const f32x4 o = park[((pw * NV + j) * NCF + cf) * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int co = co0 + (wa * NCF + cf) * 16 + lg * 4 + r;
                    if (co < p.Cout_w)
                        out[(((size_t)q * NTAP + j) * p.Cout_w + co) * MCGEN_CK + col] = acc[j][cf][r] + o[r];
                }
            }
        }
    }
    if (do_bias && tid < WG_NT) {
        const int col = rtid & 63, part = rtid >> 6;
        if (co0 + col < p.Cout_w) p.bias_slabs[((size_t)wi.bz * 4 + part) * p.Cout_w + co0 + col] = bsum;
    }
}

// Sums the split slabs in a FIXED order and scatters into the master layout.  A weight-gradient launch with many splits (the
// image-layer kernel: 256 per half) used to be the reduce launch's long pole -- one thread per element vector walked every
// split, four loads in flight.  Now SL = 1, 2, 4, 8 or 16 adjacent lanes share an element vector: lane sl sums the splits
// z = sl, sl + SL, ... in ascending order (groups of four independent loads), the lanes meet by a fixed xor butterfly and
// lane 0 writes.  The order depends on `splits` alone: run-to-run bit-identical.
__device__ __forceinline__ int reduce_lanes(int splits) {
    return splits <= 8 ? 1 : splits <= 16 ? 2 : splits <= 32 ? 4 : splits <= 64 ? 8 : 16;
}
__device__ __forceinline__ void reduce_job(const float* __restrict__ slabs, int splits, size_t slab_elems,
                                           float* __restrict__ grad, int Cout, int Cin, int KS, int Cout_w,
                                           int row_perm, float alpha, int accumulate,
                                           const float* __restrict__ bias_slabs, float* bias_grad, float* bias_grad2,
                                           const float* __restrict__ row_scale, int tapcols = 0, int tap0 = 0, int ntap_out = 0) {
    // tapcols (the image-layer kernel's compact slabs, wgrad_c8.hip): a slab is [chunk][Cout_w][32] and column tap * 8 + ci holds
    // (tap, ci) -- the taps are columns, not a slab dimension
    const int ntap = KS * KS;
    const int stap = tapcols ? 1 : ntap;
    const int nout = ntap_out > 0 ? ntap_out : ntap;              // taps the master-layout gradient holds: [tap0, tap0 + nout)
    const int SL = reduce_lanes(splits);
    const size_t gtid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x / SL;
    const int sl = (int)(gtid % SL);
    const int Cc = row_perm > 1 ? Cout / row_perm : Cout;
    const size_t nvec = slab_elems / 4;
    // (every lane of a vector's group runs the same trip count: the butterfly below needs all of them)
    for (size_t e4 = gtid / SL; e4 < nvec; e4 += stride) {
        const size_t e = e4 * 4;
        const int cl = (int)(e % MCGEN_CK); size_t t = e / MCGEN_CK;
        const int co = (int)(t % Cout_w); t /= Cout_w;
        const int col = (int)(t / stap) * MCGEN_CK + cl;
        const int tap = tapcols ? (col >> 3) : (int)(t % stap);
        const int ci = tapcols ? (col & 7) : col;
        const bool live = co < Cout && ci < Cin && tap >= tap0 && tap < tap0 + nout && tap < ntap;     // (uniform over the group: it depends on e4 alone)
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            int z = sl;
            for (; z + 3 * SL < splits; z += 4 * SL) {
                const f32x4 a0 = *reinterpret_cast<const f32x4*>(slabs + (size_t)(z + 0 * SL) * slab_elems + e);
                const f32x4 a1 = *reinterpret_cast<const f32x4*>(slabs + (size_t)(z + 1 * SL) * slab_elems + e);
                const f32x4 a2 = *reinterpret_cast<const f32x4*>(slabs + (size_t)(z + 2 * SL) * slab_elems + e);
                const f32x4 a3 = *reinterpret_cast<const f32x4*>(slabs + (size_t)(z + 3 * SL) * slab_elems + e);
                s += (a0 + a1) + (a2 + a3);
            }
            for (; z < splits; z += SL) s += *reinterpret_cast<const f32x4*>(slabs + (size_t)z * slab_elems + e);
        }
        for (int o = 1; o < SL; o <<= 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) s[j] += __shfl_xor(s[j], o);
        }
        if (!live || sl != 0) continue;
        const int com = row_perm > 1 ? (co % Cc) * row_perm + co / Cc : co;      // image row -> master row
        const float ra = row_scale ? alpha * row_scale[com] : alpha;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ci + j >= Cin) break;
            const size_t i = ((size_t)com * Cin + ci + j) * nout + (tap - tap0);
            const float v = s[j] * ra;
            grad[i] = accumulate ? grad[i] + v : v;
        }
    }
    if (bias_slabs && bias_grad) {
        // one wave per output channel: lanes split the splits*4 partial rows, fixed-order butterfly
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
        for (int co = blockIdx.x * nwv + wv; co < Cout; co += gridDim.x * nwv) {
            float s = 0.f;
            for (int z = lane; z < splits * 4; z += 64) s += bias_slabs[(size_t)z * Cout_w + co];
            for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o);
            if (lane == 0) {
                const int com = row_perm > 1 ? (co % Cc) * row_perm + co / Cc : co;
                s *= row_scale ? alpha * row_scale[com] : alpha;
                bias_grad[com] = accumulate ? bias_grad[com] + s : s;
                if (bias_grad2) bias_grad2[com] = accumulate ? bias_grad2[com] + s : s;
            }
        }
    }
}
__global__ void wgrad_reduce_kernel(const float* __restrict__ slabs, int splits, size_t slab_elems,
                                    float* __restrict__ grad, int Cout, int Cin, int KS, int Cout_w,
                                    int row_perm, float alpha, int accumulate,
                                    const float* __restrict__ bias_slabs, float* bias_grad, float* bias_grad2,
                                    const float* __restrict__ row_scale, int tapcols, int tap0, int ntap_out) {
    reduce_job(slabs, splits, slab_elems, grad, Cout, Cin, KS, Cout_w, row_perm, alpha, accumulate, bias_slabs, bias_grad, bias_grad2, row_scale, tapcols, tap0, ntap_out);
}
// floats per split of a reduce job's slabs
static __host__ __device__ inline size_t reduce_slab_elems(int Cin, int cin_slab, int ksize, int Cout_w, int tapcols) {
    if (tapcols) return (size_t)((ksize * ksize * 8 + MCGEN_CK - 1) / MCGEN_CK) * Cout_w * MCGEN_CK;
    const int cs = cin_slab > 0 ? cin_slab : Cin;
    const int nchunk = ((cs + 7) / 8 * 8 + MCGEN_CK - 1) / MCGEN_CK;
    return (size_t)nchunk * ksize * ksize * Cout_w * MCGEN_CK;
}
// All reductions of one backward pass in ONE launch: blockIdx.y picks the job, the job table travels by value
// in the kernel arguments (graph-capture safe: no host table to keep alive).
struct ReduceJobs { mcgen_wreduce_t j[MCGEN_WREDUCE_MAX]; };
__global__ void wgrad_reduce_batch_kernel(const ReduceJobs jobs) {
    const mcgen_wreduce_t& j = jobs.j[blockIdx.y];
    const size_t slab_elems = reduce_slab_elems(j.Cin, j.cin_slab, j.ksize, j.Cout_w, j.tapcols);
    reduce_job(j.slabs, j.splits, slab_elems, j.grad, j.Cout, j.Cin, j.ksize, j.Cout_w, j.row_perm, j.alpha, j.accumulate,
               j.bias_slabs, j.bias_grad, j.bias_grad2, j.row_scale, j.tapcols, j.tap0, j.ntap_out);
}

static int wgrad_chunks(const mcgen_wgrad_t* p) { return (p->seg.C + MCGEN_CK - 1) / MCGEN_CK; }

#ifdef MCGEN_TUNING
static long wg_env(const char* name, long dflt) { const char* e = getenv(name); return e ? atol(e) : dflt; }
#else
static long wg_env(const char*, long dflt) { return dflt; }
#endif

template <typename T, int KS, int LGW, typename PA>
static int launch(const mcgen_wgrad_t* p, const PA& pa, int nlayers, hipStream_t st) {
    using TR = WgTraits<T>;
    const long Mtot = (long)p->N * p->H * p->W;
    const int m_tiles = (int)((Mtot + WG_BM - 1) / WG_BM);
    const int PP = mcgen_patch_pixels(WG_BM, p->H, p->W, KS);
    const int a_bytes = round_up(PP * TR::APITCH, 32);
    const int lds = a_bytes + WG_BM * TR::DPITCH;
    dim3 grid((p->Cout_w + WG_BCO - 1) / WG_BCO, wgrad_chunks(p), p->splits * nlayers);
    // tuning builds (-DMCGEN_TUNING) read these once per process; the shipped library has no environment-dependent dispatch
    static const int xcd_map = (int)wg_env("MCGEN_WGRAD_XCD", 1);
    static const int mode = (int)wg_env("MCGEN_WGRAD_MODE", -1);      // 0 single role, 1 producer/consumer, -1 policy
    // the role split only pays when a workgroup walks several tiles (staging of tile i+1 overlaps tile i)
    const bool pc = mode >= 0 ? (mode == 1) : (m_tiles >= 4 * p->splits);
    if constexpr (sizeof(T) == 2 && (1 << LGW) <= WG_BM) {
        // ring form: bf16, every tile inside one image, and the rings fit in LDS.  First choice, also for 1x1 gradients with
        // many chunks (measured against the chunk groups below: 128->128 at 16x16 47 -> 24 us, 256->256 at 32x32 134 -> 118)
        static const int ring = (int)wg_env("MCGEN_WGRAD_RING", 1);
        const int nix = (PP * 4 + WG_NT - 1) / WG_NT;
        const int ldsr = std::max(2 * a_bytes + WG_NR * nix * WG_NT * 16 + WG_ND * WG_DSLOT + WG_NR * 4 * 128, KS * KS * 8192);
        // (upsampled reads need tiles that start on even rows: at least two rows per tile)
        const bool ups_ok = (!p->seg.ups && !p->dy_ups) || (WG_BM / (1 << LGW)) % 2 == 0;
        if (pc && ring && (long)p->H * p->W >= WG_BM && ups_ok && Mtot % WG_BM == 0 && ldsr <= 160 * 1024) {
            auto kr = wgrad_ring_kernel<KS, LGW, PA>;
            static bool raisedr = false;
            if (!raisedr) {
                raisedr = true;
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kr), hipFuncAttributeMaxDynamicSharedMemorySize, ldsr);
                if (e != hipSuccess) return mcgen_fail("wgrad: cannot raise LDS limit: %s", hipGetErrorString(e));
            }
            hipLaunchKernelGGL(kr, grid, dim3(3 * WG_NT), ldsr, st, pa, a_bytes, m_tiles, xcd_map);
            MCGEN_LAUNCH_CHECK("wgrad(ring)");
            return 0;
        }
    }
    if constexpr (KS == 1) {
        // 1x1 on small maps (no ring form): chunk groups of 4 when there are enough chunks (see wgrad_pc_kernel)
        constexpr int NCH = 4;
        static const int grp = (int)wg_env("MCGEN_WGRAD_GROUP", 1);
        const int lds4 = 2 * NCH * a_bytes + 2 * WG_BM * TR::DPITCH;     // bf16: 136 KB; fp32 does not fit -> plain path
        if (pc && wgrad_chunks(p) >= NCH && lds4 <= 160 * 1024 && grp) {
            auto kern4 = wgrad_pc_kernel<T, 1, LGW, NCH, false, PA>;
            static bool raised4 = false;
            if (lds4 > 64 * 1024 && !raised4) {
                raised4 = true;
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern4), hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
                if (e != hipSuccess) return mcgen_fail("wgrad: cannot raise LDS limit: %s", hipGetErrorString(e));
            }
            dim3 grid4((p->Cout_w + WG_BCO - 1) / WG_BCO, (wgrad_chunks(p) + NCH - 1) / NCH, p->splits * nlayers);
            hipLaunchKernelGGL(kern4, grid4, dim3(2 * WG_NT), lds4, st, pa, a_bytes, m_tiles, xcd_map);
            MCGEN_LAUNCH_CHECK("wgrad(pc, chunk groups)");
            return 0;
        }
    }
    if constexpr (sizeof(T) == 2) {
        // bf16, whole pixel tiles: the dy tile by LDS-DMA (opt-in: measured neutral -- 86.7 vs 87.5 us on the 128->128
        // 32x32 layer -- so the register path stays the default)
        static const int dy_dma = (int)wg_env("MCGEN_WGRAD_DMA", 0);
        if (pc && dy_dma && Mtot % WG_BM == 0) {
            const int ldsd = 2 * a_bytes + 2 * WG_BM * WG_BCO * 2;
            auto kd = wgrad_pc_kernel<T, KS, LGW, 1, true, PA>;
            static bool raisedd = false;
            if (ldsd > 64 * 1024 && !raisedd) {
                raisedd = true;
                hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, ldsd);
                if (e != hipSuccess) return mcgen_fail("wgrad: cannot raise LDS limit: %s", hipGetErrorString(e));
            }
            hipLaunchKernelGGL(kd, grid, dim3(2 * WG_NT), ldsd, st, pa, a_bytes, m_tiles, xcd_map);
            MCGEN_LAUNCH_CHECK("wgrad(pc, dy dma)");
            return 0;
        }
    }
    if (pc) {
        const int lds2 = 2 * a_bytes + 2 * WG_BM * TR::DPITCH;
        auto kern2 = wgrad_pc_kernel<T, KS, LGW, 1, false, PA>;
        static bool raised2 = false;
        if (lds2 > 64 * 1024 && !raised2) {
            raised2 = true;
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern2), hipFuncAttributeMaxDynamicSharedMemorySize, lds2);
            if (e != hipSuccess) return mcgen_fail("wgrad: cannot raise LDS limit: %s", hipGetErrorString(e));
        }
        hipLaunchKernelGGL(kern2, grid, dim3(2 * WG_NT), lds2, st, pa, a_bytes, m_tiles, xcd_map);
        MCGEN_LAUNCH_CHECK("wgrad(pc)");
        return 0;
    }
    auto kern = wgrad_kernel<T, KS, LGW, PA>;
    static bool raised = false;
    if (lds > 64 * 1024 && !raised) {
        raised = true;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return mcgen_fail("wgrad: cannot raise LDS limit: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, grid, dim3(WG_NT), lds, st, pa, a_bytes, m_tiles, xcd_map);
    MCGEN_LAUNCH_CHECK("wgrad");
    return 0;
}

template <typename T, typename PA>
static int launch_t(const mcgen_wgrad_t* p, const PA& pa, int nlayers, hipStream_t st) {
    const int lgw = ilog2_exact(p->W);
    if (p->seg.ksize == 3) {
        switch (lgw) {
            case 2: return launch<T, 3, 2, PA>(p, pa, nlayers, st);
            case 3: return launch<T, 3, 3, PA>(p, pa, nlayers, st);
            case 4: return launch<T, 3, 4, PA>(p, pa, nlayers, st);
            case 5: return launch<T, 3, 5, PA>(p, pa, nlayers, st);
        }
    } else {
        switch (lgw) {
            case 0: return launch<T, 1, 0, PA>(p, pa, nlayers, st);
            case 1: return launch<T, 1, 1, PA>(p, pa, nlayers, st);
            case 2: return launch<T, 1, 2, PA>(p, pa, nlayers, st);
            case 3: return launch<T, 1, 3, PA>(p, pa, nlayers, st);
            case 4: return launch<T, 1, 4, PA>(p, pa, nlayers, st);
            case 5: return launch<T, 1, 5, PA>(p, pa, nlayers, st);
        }
    }
    return mcgen_fail("wgrad: no instantiation for ksize %d at W = %d", p->seg.ksize, p->W);
}

static int wgrad_check(const mcgen_wgrad_t* p) {
    MCGEN_CHECK(p && p->seg.x && p->dy && p->slabs, "wgrad: null pointer");
    MCGEN_CHECK(p->N > 0 && ilog2_exact(p->H) >= 0 && ilog2_exact(p->W) >= 0 && p->W <= 64, "wgrad: H, W must be powers of two, W <= 64");
    MCGEN_CHECK(WG_BM >= 2 * p->W || p->H * p->W <= WG_BM, "wgrad: W too large for the pixel tile");
    MCGEN_CHECK(p->seg.C > 0 && p->seg.C % 8 == 0 && p->Cdy % 8 == 0, "wgrad: channel pitches must be multiples of 8");
    MCGEN_CHECK(p->seg.ksize == 1 || p->seg.ksize == 3, "wgrad: ksize must be 1 or 3");
    MCGEN_CHECK(p->seg.group_n == 0, "wgrad: BatchNorm statistics groups are a forward-only feature");
    MCGEN_CHECK(p->Cout > 0 && p->Cout_w == round_up(p->Cout, 16) && p->Cdy >= p->Cout, "wgrad: bad Cout/Cout_w/Cdy");
    MCGEN_CHECK(p->splits >= 1 && p->splits <= 65535, "wgrad: bad splits");
    MCGEN_CHECK(!p->halves || (p->splits % 2 == 0 && (((long)p->N * p->H * p->W + WG_BM - 1) / WG_BM) % 2 == 0 && ((long)p->N * p->H * p->W) % (2 * WG_BM) == 0),
                "wgrad: halves needs even splits and a whole number of pixel tiles per half");
    MCGEN_CHECK(!p->dy_ups || (p->H >= 2 && p->W >= 2), "wgrad: dy_ups needs H, W >= 2");
    return 0;
}

}  // namespace

extern "C" int64_t mcgen_wgrad_slab_elems(const mcgen_wgrad_t* p) {
    if (!p) return 0;
    return (int64_t)wgrad_chunks(p) * p->seg.ksize * p->seg.ksize * p->Cout_w * MCGEN_CK;
}

extern "C" int mcgen_wgrad(const mcgen_wgrad_t* p, int dtype, void* stream) {
    if (int rc = wgrad_check(p)) return rc;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (mcgen_wgrad_c8_ok(p, dtype)) return mcgen_wgrad_c8(p, st);      // the image-side layers: a stream over dy (wgrad_c8.hip)
    if (dtype == MCGEN_F32) return launch_t<float, mcgen_wgrad_t>(p, *p, 1, st);
    if (dtype == MCGEN_BF16) return launch_t<bf16_t, mcgen_wgrad_t>(p, *p, 1, st);
    return mcgen_fail("wgrad: unknown dtype %d", dtype);
}

// n layers of identical shape (geometry, channel pitches, kernel size, splits, the same operands present) as ONE launch of the
// kernel mcgen_wgrad would pick for each: MCGlow's 96 skinny 3x3 gradients per step (the coupling nets' first and last
// convolutions, 16 flows per level) are launch-latency bound one by one.
extern "C" int mcgen_wgrad_batch(const mcgen_wgrad_t* layers, int n, int dtype, void* stream) {
    MCGEN_CHECK(layers && n >= 1 && n <= MCGEN_WGRAD_MULTI_MAX, "wgrad_batch: 1 .. %d layers per launch", MCGEN_WGRAD_MULTI_MAX);
    MCGEN_CHECK(dtype == MCGEN_BF16, "wgrad_batch: bf16 only");
    WgBatch a;
    const mcgen_wgrad_t& f = layers[0];
    for (int i = 0; i < n; ++i) {
        const mcgen_wgrad_t& p = layers[i];
        if (int rc = wgrad_check(&p)) return rc;
        MCGEN_CHECK(!mcgen_wgrad_c8_ok(&p, dtype) && !p.halves, "wgrad_batch: layer %d belongs to another kernel (image layer / two-half launch)", i);
        MCGEN_CHECK(p.N == f.N && p.H == f.H && p.W == f.W && p.Cout == f.Cout && p.Cout_w == f.Cout_w && p.Cdy == f.Cdy && p.dy_ups == f.dy_ups &&
                    p.splits == f.splits && p.seg.C == f.seg.C && p.seg.ksize == f.seg.ksize && p.seg.ups == f.seg.ups &&
                    (p.bias_slabs != nullptr) == (f.bias_slabs != nullptr), "wgrad_batch: layer %d differs in shape from layer 0", i);
        a.l[i] = p;
    }
    for (int i = n; i < MCGEN_WGRAD_MULTI_MAX; ++i) a.l[i] = f;
    a.splits = f.splits;
    MCGEN_CHECK((long)f.splits * n <= 65535, "wgrad_batch: too many splits x layers for one grid");
    return launch_t<bf16_t, WgBatch>(&f, a, n, reinterpret_cast<hipStream_t>(stream));
}

extern "C" int mcgen_wgrad_reduce(const float* slabs, int splits, float* grad, int Cout, int Cin, int ksize,
                                  int Cout_w, int row_perm, float alpha, int accumulate,
                                  const float* bias_slabs, float* bias_grad, float* bias_grad2,
                                  const float* row_scale, int cin_slab, int tapcols, int tap0, int ntap_out, void* stream) {
    MCGEN_CHECK(ntap_out == 0 || (tap0 >= 0 && ntap_out > 0 && tap0 + ntap_out <= ksize * ksize), "wgrad_reduce: bad tap window");
    MCGEN_CHECK(slabs && grad && splits >= 1, "wgrad_reduce: bad arguments");
    MCGEN_CHECK(row_perm <= 1 || Cout % row_perm == 0, "wgrad_reduce: row_perm must divide Cout");
    MCGEN_CHECK(cin_slab == 0 || cin_slab >= Cin, "wgrad_reduce: cin_slab is the (padded) channel count the slabs were built for");
    MCGEN_CHECK(!tapcols || (Cin <= 8 && (cin_slab == 0 || cin_slab == 8)), "wgrad_reduce: tapcols slabs hold 8-channel layers");
    const size_t slab_elems = reduce_slab_elems(Cin, cin_slab, ksize, Cout_w, tapcols);
    const int sl = splits <= 8 ? 1 : splits <= 16 ? 2 : splits <= 32 ? 4 : splits <= 64 ? 8 : 16;                     // reduce_lanes
    int blocks = (int)((slab_elems / 4 * sl + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       slabs, splits, slab_elems, grad, Cout, Cin, ksize, Cout_w, row_perm, alpha, accumulate,
                       bias_slabs, bias_grad, bias_grad2, row_scale, tapcols, tap0, ntap_out);
    MCGEN_LAUNCH_CHECK("wgrad_reduce");
    return 0;
}

extern "C" int mcgen_wgrad_reduce_batch(const mcgen_wreduce_t* jobs, int n, void* stream) {
    MCGEN_CHECK(jobs && n > 0, "wgrad_reduce_batch: bad arguments");
    for (int base = 0; base < n; base += MCGEN_WREDUCE_MAX) {
        const int m = (n - base < MCGEN_WREDUCE_MAX) ? n - base : MCGEN_WREDUCE_MAX;
        ReduceJobs t;
        size_t most = 1;
        for (int i = 0; i < m; ++i) {
            const mcgen_wreduce_t& j = jobs[base + i];
            MCGEN_CHECK(j.slabs && j.grad && j.splits > 0 && j.Cout > 0 && j.Cin > 0 && (j.ksize == 1 || j.ksize == 3) && j.Cout_w >= j.Cout,
                        "wgrad_reduce_batch: bad job %d", base + i);
            t.j[i] = j;
            MCGEN_CHECK(j.ntap_out == 0 || (j.tap0 >= 0 && j.ntap_out > 0 && j.tap0 + j.ntap_out <= j.ksize * j.ksize), "wgrad_reduce_batch: job %d: bad tap window", base + i);
            MCGEN_CHECK(!j.tapcols || (j.Cin <= 8 && (j.cin_slab == 0 || j.cin_slab == 8)), "wgrad_reduce_batch: job %d: tapcols slabs hold 8-channel layers", base + i);
            const int sl = j.splits <= 8 ? 1 : j.splits <= 16 ? 2 : j.splits <= 32 ? 4 : j.splits <= 64 ? 8 : 16;   // reduce_lanes
            const size_t v4 = reduce_slab_elems(j.Cin, j.cin_slab, j.ksize, j.Cout_w, j.tapcols) / 4 * sl;
            if (v4 > most) most = v4;
        }
        for (int i = m; i < MCGEN_WREDUCE_MAX; ++i) t.j[i] = t.j[0];
        int blocks = (int)((most + 255) / 256); if (blocks > 1024) blocks = 1024; if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(wgrad_reduce_batch_kernel, dim3(blocks, m), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), t);
        MCGEN_LAUNCH_CHECK("wgrad_reduce_batch");
    }
    return 0;
}
