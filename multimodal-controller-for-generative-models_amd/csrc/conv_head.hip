// 3x3 convolution to at most 16 output channels on 32x32 maps, bound by reading its input (gfx950, bf16).
//
//   y[n, h, w, co] = tanh?( alpha * sum_{tap, ci} prologue(x)[n, h + dh, w + dw, ci] * W[co, ci, tap] + bias[co] )
//
// The generator's image head (mcgan.py:55-60: BatchNorm -> ReLU -> MultimodalController -> Conv3x3(256, 3) -> Tanh): 3
// launches per training iteration, one of them over the 640 images of the grouped pass -- 335 MB of activations for 1.2
// GFLOP.  The general 256x16 tile stages a chunk, waits, computes, per barrier round (183 us at N = 640).  Here
//   * a workgroup owns 8 rows of an image; per 32-channel chunk the window (10 x 34 pixels x 64 B) goes global ->
//     registers (prologue: BatchNorm affine of the image's statistics group, ReLU, code) -> one of TWO LDS buffers while
//     the previous chunk's 18 MFMAs per wave run: one barrier per chunk, the next chunk's loads in flight;
//   * 54 KB of LDS and 116 registers: two workgroups per CU;
//   * the chunk's nine 16 x 32 weight fragments come straight from the [chunk][tap][16][32] image into registers;
//   * the epilogue stores 8 bytes per lane straight from the accumulators (channels 0 .. 3 | 4 .. 7 of the 8-channel pitch).
// In the grouped forward-only pass the input arrives COMPACTED (160 of 256 channels, per-image affine rows from mcgen_mc_affine,
// the image's mode's own weight image: mcgen_conv_t.wsel): 5 chunks instead of 8.
// Measured 151-178 us at N = 640 (2.0-2.3 TB/s of input; same-box 3 % under the general tile's step).  What did NOT move it
// (same-box, tools/so_shapes.sh): pixels of two chunks in flight, weight fragments a chunk ahead (192 registers: one
// workgroup per CU, 201 us), one workgroup per CU with the shallow pipeline, 64-channel steps (whole 128-byte lines per
// visit; 256 registers with spills: 304 us) -- the bound is not any single latency this kernel exposes.
#include "conv_tile.h"

namespace {

constexpr int HD_NT = 512, HD_W = 32, HD_ROWS = 8;
constexpr int HD_PC = HD_W + 2, HD_PP = (HD_ROWS + 2) * HD_PC;       // 340 window pixels
constexpr int HD_PITCH = 80;                                         // bytes per window pixel (64 + 16: odd multiple of 16)
constexpr int HD_BUF = ((HD_PP * HD_PITCH + 127) / 128) * 128;
constexpr int HD_NI = (HD_PP * 4 + HD_NT - 1) / HD_NT;               // 16-byte units per thread and chunk (3)

__global__ __launch_bounds__(HD_NT, 4)                               // (waves per SIMD: two workgroups per CU)
void conv_head_kernel(const mcgen_conv_t p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int n = blockIdx.x >> 2, h0 = (blockIdx.x & 3) * HD_ROWS;
    const mcgen_seg_t sg = p.seg[0];
    const int C = sg.C, nchunk = C >> 5;
    const float* anyf = reinterpret_cast<const float*>(p.w);
    const size_t arow = sg.group_n > 0 ? (size_t)(n / sg.group_n) * C : 0;      // the image's BatchNorm group
    const float* scp = sg.scale ? sg.scale + arow : anyf;
    const float* shp = sg.scale ? sg.shift + arow : anyf;
    const float* cdp = sg.code ? sg.code + (size_t)n * C : anyf;
    const float relu_lo = sg.relu ? 0.f : -__builtin_inff();

    // ---- this thread's window units: unit u = tid & 3 (8 channels) of window pixels (tid >> 2) + 128 k
    const int u = tid & 3;
    int src[HD_NI], dst[HD_NI];
#pragma unroll
    for (int k = 0; k < HD_NI; ++k) {
        const int wp = (tid >> 2) + 128 * k;
        const int wr = wp / HD_PC, wc = wp - wr * HD_PC;
        const int h = h0 + wr - 1, w = wc - 1;
        dst[k] = wp < HD_PP ? wp * HD_PITCH + u * 16 : -1;
        src[k] = (wp < HD_PP && (unsigned)h < (unsigned)HD_W && (unsigned)w < (unsigned)HD_W) ? ((n * HD_W + h) * HD_W + w) * C + u * 8 : -1;
    }
    const bf16_t* xs = reinterpret_cast<const bf16_t*>(sg.x);
    u32x4 raw[HD_NI];
    float sc[8], sh[8], cd[8];
    auto load_chunk = [&](int q) {
#pragma unroll
        for (int k = 0; k < HD_NI; ++k) raw[k] = *reinterpret_cast<const u32x4*>(xs + (src[k] >= 0 ? (size_t)src[k] + q * 32 : 0));
        load8f(sg.scale ? scp + q * 32 + u * 8 : anyf, sc);
        load8f(sg.scale ? shp + q * 32 + u * 8 : anyf, sh);
        load8f(sg.code ? cdp + q * 32 + u * 8 : anyf, cd);
    };
    auto write_chunk = [&](char* buf) {
#pragma unroll
        for (int k = 0; k < HD_NI; ++k) {
            if (dst[k] < 0) continue;
            union { bf16x8 h; u32x4 w; } o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s0 = sg.scale ? sc[2 * e] : 1.f, t0 = sg.scale ? sh[2 * e] : 0.f, c0 = sg.code ? cd[2 * e] : 1.f;
                const float s1 = sg.scale ? sc[2 * e + 1] : 1.f, t1 = sg.scale ? sh[2 * e + 1] : 0.f, c1 = sg.code ? cd[2 * e + 1] : 1.f;
                o.h[2 * e] = (bf16_t)(fmaxf(fmaf(__uint_as_float(raw[k][e] << 16), s0, t0), relu_lo) * c0);
                o.h[2 * e + 1] = (bf16_t)(fmaxf(fmaf(__uint_as_float(raw[k][e] & 0xffff0000u), s1, t1), relu_lo) * c1);
            }
            if (src[k] < 0) o.w = u32x4{0u, 0u, 0u, 0u};                // the convolution's zero padding
            *reinterpret_cast<u32x4*>(buf + dst[k]) = o.w;
        }
    };

    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    // weight fragment (chunk q, tap j): rows l15 (16 channels), elements 8 lg .. of [chunk][tap][16][32]
    // (per-mode weight sets, mcgen_conv_t.wsel: the image's own dense image over its compacted channels)
    const bf16_t* wimg = reinterpret_cast<const bf16_t*>(p.w) + (p.wsel ? (size_t)p.wsel[n] * (size_t)p.wsel_stride : 0) + (size_t)l15 * MCGEN_CK + lg * 8;
    load_chunk(0);
    write_chunk(smem);
#pragma unroll 1
    for (int q = 0; q < nchunk; ++q) {
        bf16x8 wf[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(wimg + ((size_t)(q * 9 + j) * 16) * MCGEN_CK);
        if (q + 1 < nchunk) load_chunk(q + 1);
        __syncthreads();                                  // chunk q's window is complete; chunk q - 1's reads are done
        const char* buf = smem + (q & 1) * HD_BUF;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
#pragma unroll
            for (int f = 0; f < 2; ++f) {
                const bf16x8 xf = *reinterpret_cast<const bf16x8*>(buf + ((wv + j / 3) * HD_PC + 16 * f + l15 + j % 3) * HD_PITCH + lg * 16);
                acc[f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], xf, acc[f], 0, 0, 0);
            }
        }
        if (q + 1 < nchunk) write_chunk(smem + ((q + 1) & 1) * HD_BUF);
    }
    // ---- D[co = 4 lg + r][pixel 16 f + l15]: lanes lg < 2 store channels 4 lg .. 4 lg + 3 of the pixel's 8-channel row
    if (lg < 2) {
        float bs[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = 4 * lg + r;
            bs[r] = (p.bias && co < p.Cout) ? p.bias[co] : 0.f;
            if (p.bias2 && co < p.Cout) bs[r] += p.bias2[co];
        }
        // paired output layout (mcgen_conv_t.y_group): the second half of the (n / y_group)-th batch of 2 * y_group images
        const int ny = p.y_group > 0 ? (n / p.y_group) * 2 * p.y_group + p.y_group + n % p.y_group : n;
        bf16_t* y = reinterpret_cast<bf16_t*>(p.y) + ((size_t)(ny * HD_W + h0 + wv) * HD_W) * p.Cy + 4 * lg;
#pragma unroll
        for (int f = 0; f < 2; ++f) {
            union { bf16_t h[4]; uint2 w; } o;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = fmaf(acc[f][r], p.alpha, bs[r]);
                if (p.tanh_out) v = tanhf(v);
                o.h[r] = (bf16_t)((4 * lg + r) < p.Cout ? v : 0.f);
            }
            *reinterpret_cast<uint2*>(y + (size_t)(16 * f + l15) * p.Cy) = o.w;
        }
    }
}

}  // namespace

// 1 when mcgen_conv_fused hands `p` to this kernel (declared in conv_tile.h for conv_fused.hip)
int mcgen_conv_head_ok(const mcgen_conv_t* p, int dtype) {
    if (dtype != MCGEN_BF16 || p->w_layout != 0 || p->nseg != 1) return 0;
    const mcgen_seg_t& g = p->seg[0];
    if (g.ksize != 3 || g.ups || g.cmap || g.C % 32 || g.C < 64 || g.group_n < 0 || (g.group_n > 0 && p->N % g.group_n)) return 0;
    if (p->H != HD_W || p->W != HD_W || p->Cout > 8 || p->Cout_w != 16 || p->Cy != 8) return 0;
    if (p->pool || p->res || p->ocode || p->gate_x || p->stats_mode || p->ycmap || p->order) return 0;
    if ((long)p->N * HD_W * HD_W * g.C >= (1L << 31)) return 0;          // (32-bit element offsets)
    if (p->y_group < 0 || (p->y_group > 0 && p->N % p->y_group)) return 0;
    return 1;
}

int mcgen_conv_head(const mcgen_conv_t* p, hipStream_t st) {
    static bool raised = false;
    if (!raised) {
        raised = true;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_head_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * HD_BUF);
        if (e != hipSuccess) return mcgen_fail("conv_head: cannot raise LDS limit: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(conv_head_kernel, dim3(p->N * 4), dim3(HD_NT), 2 * HD_BUF, st, *p);
    MCGEN_LAUNCH_CHECK("conv_head");
    return 0;
}
