// Weight gradients of the discriminator's image-side layers (gfx950 MFMA, bf16): the 3x3 convolution 3 -> 128 of
// FirstDisResBlock and its 1x1 shortcut (mcgan.py:76-86; autograd of nn.Conv2d weights), whose conv INPUT is the image
// (channel pitch 8, no prologue):
//
//   dW[co][tap][ci] = sum over pixels  dy[pixel][co] * img[pixel + tap][ci]                     (mcgen_wgrad, wgrad.hip)
//
// K = pixels is huge (N * 1024), the output is tiny (128 x 27 or 128 x 3): the launch is a stream over dy -- 67 MB at
// 2 N = 256 -- and bound by HBM, not by the matrix pipe.  The general kernels spend it badly: a 64 co x 32 ci workgroup tile
// whose 32-channel chunk is 3/4 padding, one 16 KB dy tile per ~1.8 us latency chain per CU (9 GB/s per CU: 29 us for the
// 3x3 layer, 17 us for the shortcut, eleven launches per iteration).  Here:
//   * a workgroup (512 threads, one per CU) owns ALL 128 output channels and walks 256-pixel steps (eight rows of one 32x32
//     image); a step's dy tile (64 KB) travels by LDS-DMA into one of two buffers, one step ahead, unpadded 256-byte rows
//     with the 16-byte units XOR-swizzled on the source address (conflict-free transposing reads) -- 64 KB in flight per CU,
//     ~30 GB/s per CU; the image window (10 x 34 pixels x 16 bytes) goes through registers;
//   * the GEMM's N side is (tap, channel): an x fragment is 16 columns = two taps x 8 channels, read by ds_read_tr16_b64 at
//     the tap-shifted pixel of each lane's half (columns of the nonexistent tenth tap read a zero region); wave w owns
//     output channels 16 w .. 16 w + 15: 5 MFMAs per 32 pixels -- nothing next to the DMA time;
//   * slabs are COMPACT: the columns of a 3x3 layer's slab are the (tap, channel) pairs, column tap * 8 + ci of a 1x1-shaped
//     slab [chunk of 32 columns][Cout_w][32] (72 of 96 columns live; the standard [tap][Cout_w][32] layout would leave 3/4 of
//     every 128-byte row unused and quadruple what the reduce reads) -- mcgen_wreduce_t.tapcols tells the reduce;
//     bias gradients: the [split * 4][Cout_w] rows; `splits` / `halves` mean what they mean in mcgen_wgrad.
#include "conv_tile.h"

namespace {

constexpr int C8_NT = 512, C8_BM = 256, C8_ROWS = 8, C8_W = 32, C8_HW = 1024, C8_CO = 128;
constexpr int C8_DROW = C8_CO * 2, C8_DBUF = C8_BM * C8_DROW;     // 256-byte dy rows, 64 KB per tile
constexpr int C8_XBUF = 5504;                                     // one window buffer (340 pixels x 16 B, rounded up)
constexpr int C8_ZERO = 2 * C8_DBUF + 2 * C8_XBUF;                // 4.5 KB of zeros: columns of taps that do not exist
constexpr int C8_ZBYTES = 4608;
constexpr int C8_LDS = C8_ZERO + C8_ZBYTES;

static __device__ __forceinline__ s16x4 c8_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(reinterpret_cast<uintptr_t>(p)));
}
static __device__ __forceinline__ bf16x8 c8_frag(const char* p0, const char* p1) {
    union { bf16x8 v; s16x4 h[2]; } u;
    u.h[0] = c8_tr16(p0); u.h[1] = c8_tr16(p1);
    return u.v;
}

template <int KS>
__global__ __launch_bounds__(C8_NT, 2)
void wgrad_c8_kernel(const mcgen_wgrad_t p) {
    constexpr int HALO = KS >> 1, NTAP = KS * KS, PC = C8_W + 2 * HALO, PR = C8_ROWS + 2 * HALO, PP = PR * PC;
    constexpr int NF = (NTAP + 1) / 2;                             // x fragments: two taps (16 columns) each
    static_assert(PP * 16 <= C8_XBUF && PP <= C8_NT, "window buffer: one 16-byte unit per thread");
    static_assert((C8_ROWS - 1) * PC * 16 + 16 * 16 + 8 <= C8_ZBYTES, "the zero region covers every row offset of a fragment read");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsD = smem;
    char* const ldsX = smem + 2 * C8_DBUF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4, q4 = l15 >> 2, p4 = l15 & 3;
    const int bz = blockIdx.x, splits = p.splits;
    const int m_tiles = (int)((long)p.N * C8_HW / C8_BM);          // 256-pixel steps
    // step walk of this split (as wgrad_multi.hip): all steps with stride `splits`, or one half of them with stride splits / 2
    const int zs = p.halves ? (splits >> 1) : splits;
    const int mt = p.halves ? (m_tiles >> 1) : m_tiles;
    const int t_first = (p.halves ? (bz / zs) * mt : 0) + bz % zs;
    const int cnt = (mt - bz % zs + zs - 1) / zs;                  // (may be 0: more splits than steps -- the slab is still written)
    const bool do_bias = p.bias_slabs != nullptr;
    if (tid < C8_ZBYTES / 16) reinterpret_cast<u32x4*>(smem + C8_ZERO)[tid] = u32x4{0u, 0u, 0u, 0u};

    // ---- the thread's window unit (window pixel tid: row pr, column pc) and its dy DMA rows
    const int pr = tid / PC, pc = tid - pr * PC;
    const bool xitem = tid < PP;
    const bool xcol = xitem && (unsigned)(pc - HALO) < (unsigned)C8_W;
    const int x_rel = ((pr - HALO) * C8_W + (pc - HALO)) * 8;       // element offset from the step's first pixel
    const char* xs = reinterpret_cast<const char*>(p.seg.x);
    const char* dyb = reinterpret_cast<const char*>(p.dy);
    const size_t dpix = (size_t)p.Cdy * 2;
    int d_src[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int m = (wave * 8 + k) * 4 + (lane >> 4);
        const int u = (lane & 15) ^ ((m & 7) << 1);
        const int r = m >> 5, c = m & 31;
        const int mp = p.dy_ups ? ((r >> 1) * (C8_W >> 1) + (c >> 1)) : m;
        d_src[k] = mp * (int)dpix + u * 16;
    }
    auto tile_of = [&](int i) { return t_first + (i < cnt ? i : cnt - 1) * zs; };
    auto dy_first = [&](int tile) -> size_t {                       // first source pixel of the step's dy rows
        const int pix0 = tile * C8_BM;
        if (!p.dy_ups) return (size_t)pix0;
        const int n0 = pix0 >> 10, h0 = (pix0 & (C8_HW - 1)) >> 5;
        return ((size_t)n0 * (C8_W >> 1) + (h0 >> 1)) * (C8_W >> 1);
    };
    u32x4 raw = {0u, 0u, 0u, 0u};
    auto load_x = [&](int i) {
        const int pix0 = tile_of(i) * C8_BM;
        const int h0 = (pix0 & (C8_HW - 1)) >> 5;
        const bool ok = xcol && (unsigned)(h0 + pr - HALO) < (unsigned)C8_W;
        const char* src = ok ? xs + ((ptrdiff_t)pix0 * 8 + x_rel) * 2 : xs;          // outside: any in-bounds address
        raw = *reinterpret_cast<const u32x4*>(src);
        if (!ok) raw = u32x4{0u, 0u, 0u, 0u};
    };
    auto write_x = [&](int i) {
        if (xitem) *reinterpret_cast<u32x4*>(ldsX + (i & 1) * C8_XBUF + tid * 16) = raw;
    };
    auto dma_dy = [&](int i) {
        const char* db = dyb + dy_first(tile_of(i)) * dpix;
        char* ds = ldsD + (i & 1) * C8_DBUF + wave * 8192;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(db + d_src[k]),
                                             (__attribute__((address_space(3))) void*)(ds + k * 1024), 16, 0, 0);
    };

    // ---- fragment addresses
    const int m0 = 4 * lg + q4;                                    // the lane's pixel inside a 16-pixel group (first row of the step)
    int aoff[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int tap = 2 * f + (p4 >> 1);
        const int wp = (tap / KS) * PC + (tap % KS) + m0;          // window pixel of (pixel m0, tap): halo origin + tap shift
        aoff[f] = tap < NTAP ? wp * 16 + (p4 & 1) * 8 : -1;        // -1: the lane's half of the fragment is a tap that does not exist
    }
    const int doff = m0 * C8_DROW + 32 * (wave ^ (m0 & 7)) + 8 * p4;    // output channels 16 wave .. + 15

    f32x4 acc[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) acc[f] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum8[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum8[e] = 0.f;

    if (cnt > 0) {
        load_x(0);
        dma_dy(0);
        write_x(0);
        if (cnt > 1) load_x(1);
    }
    for (int i = 0; i < cnt; ++i) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // dy of step i has landed (and this thread's x of step i + 1)
        __syncthreads();                                           // step i published; the buffers of step i - 1 are free
        if (i + 1 < cnt) {
            write_x(i + 1);
            dma_dy(i + 1);
            if (i + 2 < cnt) load_x(i + 2);
        }
        const char* D = ldsD + (i & 1) * C8_DBUF;
        const char* X = ldsX + (i & 1) * C8_XBUF;
        const char* Z = smem + C8_ZERO;
        if (do_bias) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const u32x4 r = *reinterpret_cast<const u32x4*>(D + ((tid >> 4) + 32 * k) * C8_DROW + (tid & 15) * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum8[2 * e] += __uint_as_float(r[e] << 16);
                    bsum8[2 * e + 1] += __uint_as_float(r[e] & 0xffff0000u);
                }
            }
        }
#pragma unroll
        for (int ks = 0; ks < C8_BM / 32; ++ks) {                  // 32 pixels = one image row
            const bf16x8 df = c8_frag(D + doff + (32 * ks) * C8_DROW, D + doff + (32 * ks + 16) * C8_DROW);
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const char* a = (aoff[f] >= 0 ? X + aoff[f] : Z) + ks * PC * 16;      // (the zero region covers every row offset)
                const bf16x8 xf = c8_frag(a, a + 16 * 16);
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(df, xf, acc[f], 0, 0, 0);
            }
        }
    }
    // ---- compact slab [split][chunk q][Cout_w][32]: lane holds D[co = 16 wave + 4 lg + r][column 16 f + l15], column = tap * 8 + ci
    constexpr int NCHUNK = (NTAP * 8 + MCGEN_CK - 1) / MCGEN_CK;
    float* out = p.slabs + (size_t)bz * NCHUNK * p.Cout_w * MCGEN_CK;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int col = 16 * f + l15;
        if (col < NTAP * 8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = 16 * wave + 4 * lg + r;
                out[((size_t)(col >> 5) * p.Cout_w + co) * MCGEN_CK + (col & 31)] = acc[f][r];
            }
        }
    }
    if (do_bias) {
        __syncthreads();                                           // the last step's reads are done: the dy buffers are free
        float* red = reinterpret_cast<float*>(smem);
        const int rg = tid >> 4, lu = (tid & 15) ^ ((rg & 7) << 1);   // the unit's logical position (rows rg + 32 k share rg & 7)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[rg * C8_CO + lu * 8 + e] = bsum8[e];
        __syncthreads();
        if (tid < C8_CO) {
            float s = 0.f;
            for (int r = 0; r < 32; ++r) s += red[r * C8_CO + tid];
            float* bs = p.bias_slabs + (size_t)bz * 4 * p.Cout_w + tid;
            bs[0] = s; bs[p.Cout_w] = 0.f; bs[2 * p.Cout_w] = 0.f; bs[3 * p.Cout_w] = 0.f;
        }
    }
}

}  // namespace

// 1 when mcgen_wgrad hands `p` to this kernel
extern "C" int mcgen_wgrad_c8_ok(const mcgen_wgrad_t* p, int dtype) {
    if (!p || dtype != MCGEN_BF16) return 0;
    const mcgen_seg_t& g = p->seg;
    if (g.C != 8 || g.ups || g.relu || g.scale || g.shift || g.code || g.cmap || g.group_n) return 0;
    if (g.ksize != 3 && g.ksize != 1) return 0;
    if (p->H != C8_W || p->W != C8_W || p->Cout != C8_CO || p->Cout_w != C8_CO || p->Cdy < C8_CO) return 0;
        if ((long)p->N * C8_HW * p->Cdy >= (1L << 30)) return 0;           // (32-bit byte offsets of the dy rows)
    return 1;
}

// floats per split of the compact slab
extern "C" int64_t mcgen_wgrad_c8_slab_elems(const mcgen_wgrad_t* p) {
    if (!p) return 0;
    const int ntap = p->seg.ksize * p->seg.ksize;
    return (int64_t)((ntap * 8 + MCGEN_CK - 1) / MCGEN_CK) * p->Cout_w * MCGEN_CK;
}

int mcgen_wgrad_c8(const mcgen_wgrad_t* p, hipStream_t st) {
    const long m_tiles = (long)p->N * C8_HW / C8_BM;               // (a workgroup with no step of its own writes a zero slab)
    MCGEN_CHECK(p->splits >= 1, "wgrad(c8): bad splits");
    MCGEN_CHECK(!p->halves || (p->splits % 2 == 0 && m_tiles % 2 == 0), "wgrad(c8): halves needs even splits and an even number of images");
    static bool raised = false;
    if (!raised) {
        raised = true;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_c8_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, C8_LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_c8_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, C8_LDS);
        if (e != hipSuccess) return mcgen_fail("wgrad(c8): cannot raise LDS limit: %s", hipGetErrorString(e));
    }
    if (p->seg.ksize == 3) hipLaunchKernelGGL(wgrad_c8_kernel<3>, dim3(p->splits), dim3(C8_NT), C8_LDS, st, *p);
    else hipLaunchKernelGGL(wgrad_c8_kernel<1>, dim3(p->splits), dim3(C8_NT), C8_LDS, st, *p);
    MCGEN_LAUNCH_CHECK("wgrad(c8)");
    return 0;
}
