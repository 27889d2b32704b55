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
    static constexpr int APITCH = WG_CI * 2 + 16;        // 144 B: rows 8-byte aligned for the tr read
    static constexpr int DPITCH = WG_BCO * 2 + 16;       // 144 B
};

static __device__ __forceinline__ s16x4 lds_tr16(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (s16x4 __attribute__((address_space(3)))*)(reinterpret_cast<uintptr_t>(p)));
}

// One MFMA operand fragment with k = 8 consecutive tile pixels (lane group lg) and 16 channels
// (lane & 15) read from an LDS image laid out [pixel][channel]: bf16 by two transposing reads
// (lane (q4, p4) supplies the address of pixel 4*half + q4, channels 4*p4..4*p4+3), fp32 by 8 scalar reads.
template <typename T> struct KFrag;
template <> struct KFrag<bf16_t> {
    typedef bf16x8 frag;
    // off2[h]: byte offset of this lane's row for half h (already includes the 8-byte column part)
    static __device__ __forceinline__ frag read(const char* base, const int (&off)[8], int col_bytes16) {
        union { bf16x8 v; s16x4 h[2]; } u;
        u.h[0] = lds_tr16(base + off[0] + col_bytes16);
        u.h[1] = lds_tr16(base + off[1] + col_bytes16);
        return u.v;
    }
};
template <> struct KFrag<float> {
    typedef f32x8 frag;
    static __device__ __forceinline__ frag read(const char* base, const int (&off)[8], int col_bytes16) {
        f32x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = *reinterpret_cast<const float*>(base + off[j] + col_bytes16);
        return r;
    }
};

// Workgroup = 64 output channels x 32 input channels x all taps.  Waves are a 2 x 2 grid:
// wave (a, b) owns output-channel fragments {2a, 2a+1} and input-channel block b for every tap, so per
// 32-pixel step it reads 2 dy fragments + NTAP window fragments for 2*NTAP MFMAs (small footprint:
// 72 accumulator registers, ~35 KB LDS -> four workgroups per CU hide each other's staging latency).
template <typename T, int KS>
__global__ __launch_bounds__(WG_NT, 3)
void wgrad_kernel(const mcgen_wgrad_t p, const int a_bytes, const int m_tiles) {
    using E = Elem<T>;
    using M = Mma<T>;
    using TR = WgTraits<T>;
    constexpr int ESZ = E::BYTES, APITCH = TR::APITCH, DPITCH = TR::DPITCH;
    constexpr int NTAP = KS * KS;
    constexpr int NPAIR = NTAP;                       // taps (this wave's ci block is fixed)
    constexpr int NCF = WG_BCO / 32;                  // output-channel fragments per wave
    constexpr int SUBS = WG_CI / 8;
    constexpr int NI = (WG_BM * 9 * SUBS / 4 + WG_NT - 1) / WG_NT;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ldsA = smem;
    char* ldsD = smem + a_bytes;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int H = p.H, W = p.W, N = p.N;
    const int co0 = blockIdx.x * WG_BCO;
    const int c0 = blockIdx.y * WG_CI;
    const mcgen_seg_t sg = p.seg;
    const int halo = KS >> 1;
    const T* dy = reinterpret_cast<const T*>(p.dy);
    const int Hd = p.dy_ups ? (H >> 1) : H, Wd = p.dy_ups ? (W >> 1) : W;

    f32x4 acc[NPAIR][NCF];
#pragma unroll
    for (int j = 0; j < NPAIR; ++j)
#pragma unroll
        for (int cf = 0; cf < NCF; ++cf) acc[j][cf] = f32x4{0.f, 0.f, 0.f, 0.f};
    // bias gradient = column sums of dy: done by the ci-tile-0 workgroups on the dy tile they stage anyway
    const bool do_bias = (p.bias_slabs != nullptr) && (blockIdx.y == 0);
    float bsum = 0.f;                                  // thread (column tid&63, row quarter tid>>6)

    const int wa = wave >> 1, wb = wave & 1;          // output-channel half, input-channel block

    for (int tile = blockIdx.z; tile < m_tiles; tile += gridDim.z) {
        const Geo g = make_geo(WG_BM, tile, H, W);
        const int PR = g.TH + 2 * halo, PC = W + 2 * halo;
        PatchStager<T, WG_NT, NI, APITCH, SUBS> stager;
        stager.setup(sg, g, N, H, W, tid);
        __syncthreads();                                        // previous tile's reads are done
        stager.stage(sg, c0, ldsA);
        // dy tile: [pixel m][64 co]
        for (int it = tid; it < WG_BM * (WG_BCO / 8); it += WG_NT) {
            const int sub = it & 7, m = it >> 3;
            const int ti = m >> g.lgTHW, rem = m & ((1 << g.lgTHW) - 1);
            const int r = rem >> g.lgW, c = rem & (W - 1);
            const int n = g.n0 + ti, h = g.h0 + r;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = 0.f;
            const int co = co0 + sub * 8;
            if (n < N && co < p.Cdy) {
                const int hd = p.dy_ups ? (h >> 1) : h, wd = p.dy_ups ? (c >> 1) : c;
                E::load8(dy + ((size_t)(n * Hd + hd) * Wd + wd) * p.Cdy + co, v);
            }
            E::store8(reinterpret_cast<T*>(ldsD + m * DPITCH + sub * 8 * ESZ), v);
        }
        __syncthreads();
        if (do_bias) {
            const int col = tid & 63, part = tid >> 6;
#pragma unroll 8
            for (int r = 0; r < WG_BM / 4; ++r)
                bsum += E::to_f(*reinterpret_cast<const T*>(ldsD + (part * (WG_BM / 4) + r) * DPITCH + col * ESZ));
        }

#pragma unroll 1
        for (int ks = 0; ks < WG_BM / 32; ++ks) {
            // byte offsets of this lane's k rows (pixels ks*32 + 8*lg ...) in the window and in the dy tile
            int offA[8], offD[8];
            if constexpr (sizeof(T) == 4) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int m = ks * 32 + lg * 8 + j;
                    const int ti = m >> g.lgTHW, rem = m & ((1 << g.lgTHW) - 1);
                    const int r = rem >> g.lgW, c = rem & (W - 1);
                    offA[j] = ((ti * PR + r) * PC + c) * APITCH + l15 * 4;
                    offD[j] = m * DPITCH + l15 * 4;
                }
            } else {
                const int q4 = l15 >> 2, p4 = l15 & 3;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const int m = ks * 32 + lg * 8 + hf * 4 + q4;
                    const int ti = m >> g.lgTHW, rem = m & ((1 << g.lgTHW) - 1);
                    const int r = rem >> g.lgW, c = rem & (W - 1);
                    offA[hf] = ((ti * PR + r) * PC + c) * APITCH + p4 * 8;
                    offD[hf] = m * DPITCH + p4 * 8;
                }
            }
            typename M::frag dfrag[NCF];
#pragma unroll
            for (int cf = 0; cf < NCF; ++cf) dfrag[cf] = KFrag<T>::read(ldsD, offD, (wa * NCF + cf) * 16 * ESZ);
#pragma unroll
            for (int j = 0; j < NPAIR; ++j) {
                const int tapoff = ((j / KS) * PC + (j % KS)) * APITCH;
                const typename M::frag afrag = KFrag<T>::read(ldsA + tapoff, offA, wb * 16 * ESZ);
#pragma unroll
                for (int cf = 0; cf < NCF; ++cf) M::run(dfrag[cf], afrag, acc[j][cf]);
            }
        }
    }

    if (do_bias) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
        red[tid] = bsum;
        __syncthreads();
        if (tid < 64 && co0 + tid < p.Cout_w)
            p.bias_slabs[(size_t)blockIdx.z * p.Cout_w + co0 + tid] = (red[tid] + red[64 + tid]) + (red[128 + tid] + red[192 + tid]);
    }
    // slab[z][q][tap][co][32]: lane holds D[co = 4*lg + r][ci = l15] of block (tap j, co fragment cf)
    const int nchunk = (sg.C + MCGEN_CK - 1) / MCGEN_CK;
    const size_t slab_elems = (size_t)nchunk * NTAP * p.Cout_w * MCGEN_CK;
    float* out = p.slabs + (size_t)blockIdx.z * slab_elems;
    const int q = blockIdx.y;
#pragma unroll
    for (int j = 0; j < NPAIR; ++j) {
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

// Sums the split slabs in slab order (coalesced 16-byte reads, 4 splits in flight) and scatters into
// the master layout.
__global__ void wgrad_reduce_kernel(const float* __restrict__ slabs, int splits, size_t slab_elems,
                                    float* __restrict__ grad, int Cout, int Cin, int KS, int Cout_w,
                                    int row_perm, float alpha, int accumulate,
                                    const float* __restrict__ bias_slabs, float* bias_grad, float* bias_grad2) {
    const int ntap = KS * KS;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const int Cc = row_perm > 1 ? Cout / row_perm : Cout;
    const size_t nvec = slab_elems / 4;
    for (size_t e4 = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e4 < nvec; e4 += stride) {
        const size_t e = e4 * 4;
        const int cl = (int)(e % MCGEN_CK); size_t t = e / MCGEN_CK;
        const int co = (int)(t % Cout_w); t /= Cout_w;
        const int tap = (int)(t % ntap); const int q = (int)(t / ntap);
        const int ci = q * MCGEN_CK + cl;
        if (co >= Cout || ci >= Cin) continue;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        int z = 0;
        for (; z + 4 <= splits; z += 4) {
            const f32x4 a0 = *reinterpret_cast<const f32x4*>(slabs + (size_t)(z + 0) * slab_elems + e);
            const f32x4 a1 = *reinterpret_cast<const f32x4*>(slabs + (size_t)(z + 1) * slab_elems + e);
            const f32x4 a2 = *reinterpret_cast<const f32x4*>(slabs + (size_t)(z + 2) * slab_elems + e);
            const f32x4 a3 = *reinterpret_cast<const f32x4*>(slabs + (size_t)(z + 3) * slab_elems + e);
            s += (a0 + a1) + (a2 + a3);
        }
        for (; z < splits; ++z) s += *reinterpret_cast<const f32x4*>(slabs + (size_t)z * slab_elems + e);
        const int com = row_perm > 1 ? (co % Cc) * row_perm + co / Cc : co;      // image row -> master row
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ci + j >= Cin) break;
            const size_t i = ((size_t)com * Cin + ci + j) * ntap + tap;
            const float v = s[j] * alpha;
            grad[i] = accumulate ? grad[i] + v : v;
        }
    }
    if (bias_slabs && bias_grad) {
        for (size_t co = blockIdx.x * (size_t)blockDim.x + threadIdx.x; co < (size_t)Cout; co += stride) {
            float s = 0.f;
            for (int z = 0; z < splits; ++z) s += bias_slabs[(size_t)z * Cout_w + co];
            s *= alpha;
            const int com = row_perm > 1 ? ((int)co % Cc) * row_perm + (int)co / Cc : (int)co;
            bias_grad[com] = accumulate ? bias_grad[com] + s : s;
            if (bias_grad2) bias_grad2[com] = accumulate ? bias_grad2[com] + s : s;
        }
    }
}

static int wgrad_chunks(const mcgen_wgrad_t* p) { return (p->seg.C + MCGEN_CK - 1) / MCGEN_CK; }

template <typename T, int KS>
static int launch(const mcgen_wgrad_t* p, hipStream_t st) {
    using TR = WgTraits<T>;
    const long Mtot = (long)p->N * p->H * p->W;
    const int m_tiles = (int)((Mtot + WG_BM - 1) / WG_BM);
    const int PP = mcgen_patch_pixels(WG_BM, p->H, p->W, KS);
    const int a_bytes = round_up(PP * TR::APITCH, 32);
    const int lds = a_bytes + WG_BM * TR::DPITCH;
    dim3 grid((p->Cout_w + WG_BCO - 1) / WG_BCO, (p->seg.C + WG_CI - 1) / WG_CI, p->splits);
    auto kern = wgrad_kernel<T, KS>;
    static bool raised = false;
    if (lds > 64 * 1024 && !raised) {
        raised = true;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return mcgen_fail("wgrad: cannot raise LDS limit: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, grid, dim3(WG_NT), lds, st, *p, a_bytes, m_tiles);
    MCGEN_LAUNCH_CHECK("wgrad");
    return 0;
}

}  // namespace

extern "C" int64_t mcgen_wgrad_slab_elems(const mcgen_wgrad_t* p) {
    if (!p) return 0;
    return (int64_t)wgrad_chunks(p) * p->seg.ksize * p->seg.ksize * p->Cout_w * MCGEN_CK;
}

extern "C" int mcgen_wgrad(const mcgen_wgrad_t* p, int dtype, void* stream) {
    MCGEN_CHECK(p && p->seg.x && p->dy && p->slabs, "wgrad: null pointer");
    MCGEN_CHECK(p->N > 0 && ilog2_exact(p->H) >= 0 && ilog2_exact(p->W) >= 0 && p->W <= 64, "wgrad: H, W must be powers of two, W <= 64");
    MCGEN_CHECK(WG_BM >= 2 * p->W || p->H * p->W <= WG_BM, "wgrad: W too large for the pixel tile");
    MCGEN_CHECK(p->seg.C > 0 && p->seg.C % 8 == 0 && p->Cdy % 8 == 0, "wgrad: channel pitches must be multiples of 8");
    MCGEN_CHECK(p->seg.ksize == 1 || p->seg.ksize == 3, "wgrad: ksize must be 1 or 3");
    MCGEN_CHECK(p->Cout > 0 && p->Cout_w == round_up(p->Cout, 16) && p->Cdy >= p->Cout, "wgrad: bad Cout/Cout_w/Cdy");
    MCGEN_CHECK(p->splits >= 1 && p->splits <= 65535, "wgrad: bad splits");
    MCGEN_CHECK(!p->dy_ups || (p->H >= 2 && p->W >= 2), "wgrad: dy_ups needs H, W >= 2");
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    if (dtype == MCGEN_F32) return p->seg.ksize == 3 ? launch<float, 3>(p, st) : launch<float, 1>(p, st);
    if (dtype == MCGEN_BF16) return p->seg.ksize == 3 ? launch<bf16_t, 3>(p, st) : launch<bf16_t, 1>(p, st);
    return mcgen_fail("wgrad: unknown dtype %d", dtype);
}

extern "C" int mcgen_wgrad_reduce(const float* slabs, int splits, float* grad, int Cout, int Cin, int ksize,
                                  int Cout_w, int row_perm, float alpha, int accumulate,
                                  const float* bias_slabs, float* bias_grad, float* bias_grad2, void* stream) {
    MCGEN_CHECK(slabs && grad && splits >= 1, "wgrad_reduce: bad arguments");
    MCGEN_CHECK(row_perm <= 1 || Cout % row_perm == 0, "wgrad_reduce: row_perm must divide Cout");
    const int nchunk = (round_up(Cin, 8) + MCGEN_CK - 1) / MCGEN_CK;
    const size_t slab_elems = (size_t)nchunk * ksize * ksize * Cout_w * MCGEN_CK;
    int blocks = (int)((slab_elems / 4 + 255) / 256); if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                       slabs, splits, slab_elems, grad, Cout, Cin, ksize, Cout_w, row_perm, alpha, accumulate,
                       bias_slabs, bias_grad, bias_grad2);
    MCGEN_LAUNCH_CHECK("wgrad_reduce");
    return 0;
}
