/*
 * mcgen_hip.h -- C ABI of libmcgen_hip.so: the MI355X (gfx950) kernels behind the
 * MultimodalController training hot path.
 *
 * The reference (diaoenmao/Multimodal-Controller-for-Generative-Models) is pure
 * Python on stock PyTorch: its "operator API" for this path is the nn.Module
 * surface of src/modules/modules.py and src/models/mcgan.py, and every entry
 * point below replaces a chain of ATen calls those modules make.  Each
 * declaration cites the reference lines whose arithmetic it implements.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - activations are NHWC ("channels last"), element type selected by `dtype`
 *     (MCGEN_F32 = float, MCGEN_BF16 = bfloat16), channel pitch a multiple of 8;
 *   - parameters, statistics, codes and gradients of parameters are float32;
 *   - `stream` is a hipStream_t passed as void*; nothing synchronises, nothing
 *     allocates: the caller owns every buffer (workspace sizes are queryable);
 *   - return value 0 = launched, non-zero = rejected before launch
 *     (mcgen_last_error() holds the reason).  Device-side faults surface through
 *     the HIP runtime on the caller's next synchronisation, as for any kernel.
 */
#ifndef MCGEN_HIP_H
#define MCGEN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { MCGEN_F32 = 0, MCGEN_BF16 = 1 };

const char* mcgen_last_error(void);
int mcgen_abi_version(void);      /* 9: mcgen_conv_t.y_group (paired output layout of the image head), mcgen_onehot_rep, MCGEN_WREDUCE_MAX 32, mcgen_dtail_hinge_fused, mcgen_wgrad_c8_ok + tapcols slabs (mcgen_wgrad_reduce gained an argument), mcgen_mc_gather_batch(n_label, scale, n_half), mcgen_conv_t.wsel / wsel_stride / order / yperm + mcgen_prep_t.kmap / rmap (per-mode dense weight sets), mcgen_prep_weight_batch_codes, mcgen_wgrad_batch, mcgen_bn_finalize_batch, mcgen_gated_fwd_batch, mcgen_wreduce_t.tap0 / ntap_out; 8: mcgen_adam / mcgen_sn_fix_pair_adam take lr_dev (learning rate read on the device at execution time); 7: + mcgen_conv_form, mcgen_sn_power_iter_rounds, the MCGlow *_batch entry points, mcgen_sn_fix_pair_adam(advance_step); 6: + mcgen_wgrad_multi (5: + mcgen_conv_t.bias2, mcgen_sn_power_iter_snap; 4: compacted activations between forward-only launches) */

/* One K-segment of a fused convolution: the input tensor and the prologue that
 * is applied while the tile is staged into LDS:
 *     a = x[n, h>>ups, w>>ups, c]                 nn.Upsample(2,'nearest')   mcgan.py:17,27
 *     a = a * scale[c] + shift[c]    (if scale)   nn.BatchNorm2d apply       mcgan.py:15,20,55
 *     a = max(a, 0)                  (if relu)    nn.ReLU                    mcgan.py:16,21,56,79,102,105
 *     a = a * code[n, c]             (if code)    MultimodalController.forward  modules.py:71-76
 */
typedef struct {
    const void*  x;       /* [N, H>>ups, W>>ups, C]                                   */
    const float* scale;   /* [C] ([N/group_n][C] when group_n > 0) or NULL            */
    const float* shift;   /* same shape as scale (read only when scale != NULL)       */
    const float* code;    /* [N, C] = indicator @ codebook, or NULL.  MultimodalController codes are >= 0 (a 0/1 codebook
                           * times a non-negative indicator, modules.py:58-76).  With relu = 1 the software-pipelined bf16
                           * convolution folds the code into the affine in front of the ReLU: a tile whose code row has a
                           * NEGATIVE entry returns NaN (fails loudly) instead of code * relu(.); every other form and
                           * the weight-gradient kernels compute code * relu(.) for any sign                           */
    const int16_t* cmap;  /* per-sample compaction map built by mcgen_mc_cmap from `code` (see there), or NULL:
                           * the K loop then visits only the channels whose code is non-zero -- with
                           * controller_rate 0.5 half of them (modules.py:58-76); bf16 launches on K-major
                           * weight images (mcgen_conv_t.w_layout = 1) whose tiles lie inside one image     */
    int32_t C;            /* channels of x (multiple of 8)                            */
    int32_t ups;          /* 1: x is at half the convolution's resolution             */
    int32_t relu;
    int32_t ksize;        /* 3 (padding 1) or 1 (padding 0); stride is always 1       */
    int32_t group_n;      /* > 0: the batch is N/group_n independent BatchNorm batches of group_n images each
                           * (several training-mode generator forwards run as one pass, train_gan.py:145-146):
                           * image n is normalised with row n / group_n of scale / shift.  0: one batch.
                           * Convolution launches only (mcgen_wgrad requires 0); a tile never straddles groups. */
    int32_t cmap_stride;  /* int16 elements per sample record of cmap                 */
    int32_t Cw;           /* w_layout 2 only: channels of the segment's WEIGHTS (rows of its K-major image, without the
                           * zero row) when x holds compacted channels (C < Cw); 0: Cw = C                         */
    int32_t reserved_;
} mcgen_seg_t;

/* Fused convolution  y = epilogue( sum_seg conv(prologue_seg(x_seg), W_seg) ).
 * Replaces, per call, one of the op chains of GenResBlock / FirstDisResBlock /
 * DisResBlock (mcgan.py:9-44, 72-138) in the forward direction, and the matching
 * autograd input-gradient chain in the backward direction (the same kernel run on
 * flipped/transposed weights prepared by mcgen_prep_weight).
 *
 * Epilogue, in this order, on v = alpha * acc (acc summed over 2x2 windows when pool):
 *     v += bias[co]                                   conv bias
 *     v *= ocode[n, co]              (if ocode)       d(MC mask)             modules.py:75
 *     v *= (z > 0), z = gate_x*gscale+gshift (if gate_x)  d(ReLU) [after BN when gscale]
 *     stats_mode 2: partial sums of v and v*xhat, xhat = (gate_x-gmean)*grstd   (BN backward)
 *     v += res[n, ho, wo, co]        (if res)         residual / shortcut add   mcgan.py:42,91,136
 *     v  = tanh(v)                   (if tanh_out)    nn.Tanh                   mcgan.py:60
 *     stats_mode 1: partial sums of v and v*v          (next BatchNorm's batch statistics)
 * pool=1 with alpha=0.25 is nn.AvgPool2d(2) (mcgan.py:82,85,110,114); with alpha=1
 * it is the adjoint of the nearest x2 upsample.
 */
typedef struct {
    mcgen_seg_t  seg[2];
    int32_t      nseg;
    const void*  w;        /* weight image from mcgen_prep_weight (all segments, in order) */
    const float* bias;     /* [Cout] or NULL                                               */
    void*        y;        /* [N, Ho, Wo, Cy]  (Ho = H>>pool)                              */
    int32_t N, H, W;       /* convolution (pre-pool) resolution; H, W powers of two        */
    int32_t Cout;          /* logical output channels                                      */
    int32_t Cout_w;        /* rows per block of the weight image (Cout rounded up to 16)   */
    int32_t Cy;            /* channel pitch of y / res / gate_x (multiple of 8, >= Cout)   */
    int32_t pool;
    float   alpha;
    const void*  res;
    const float* ocode;    /* [N, Cout]                                                    */
    const void*  gate_x;
    const float* gscale; const float* gshift; const float* gmean; const float* grstd;
    int32_t tanh_out;
    float*  stats;         /* [m_tiles][2][Cy] partial sums, or NULL                       */
    int32_t stats_mode;
    int32_t w_layout;      /* 0: `w` is the [chunk][tap][cout][32] image of mcgen_prep_weight;
                            * 1: the K-major image of mcgen_prep_weight_k, activations compacted while they are
                            *    staged (every segment carries a cmap);
                            * 2: K-major image, activations ALREADY compacted by the producing launch's ycmap
                            *    (segment: C = compacted pitch, Cw = weight channels, cmap = row gather; a segment
                            *    without a map is dense)                                           */
    const int16_t* ycmap;  /* compacted OUTPUT: channel slot j of a pixel of image n receives true channel cidx_n[j] of the
                            * map of the CONSUMER's code (zeros beyond its active count); y has pitch Cy (a multiple of 32
                            * that holds every sample's active channels); stats keep covering all true channels, pitch
                            * Cout_w.  Forward-only passes: needs pool = 0, no res / gate_x / ocode, tiles inside one image,
                            * one channel tile.  NULL: dense output.                              */
    int32_t ycmap_stride;
    int32_t y_group;       /* > 0: y holds 2 N images in PAIRED layout -- output image n is stored in image slot
                            * (n / y_group) * 2 * y_group + y_group + n % y_group, i.e. the second half of the
                            * (n / y_group)-th [real (+) generated] batch of 2 * y_group images that a paired discriminator
                            * update reads (train_gan.py:143-147: D(real), D(G(z)) on the same labels); the first halves are
                            * the caller's.  The image head (form 5) only; every other form requires 0.             */
    const float* bias2;    /* [Cout] or NULL: added to `bias` (a fused shortcut segment's own bias: the sum is formed in
                            * fp32 before it meets the accumulator, as bias + bias2 on the host would be)   */
    /* Per-mode dense weight sets (the software-pipelined bf16 form only; tiles inside one image):                       */
    const int32_t* wsel;   /* [N] or NULL: the image processed at POSITION i (see `order`) reads the weight image
                            * w + wsel[i] * wsel_stride -- one dense image per MultimodalController mode, its input channels the
                            * mode's active ones in order (mcgen_prep_t.kmap), for activations the producing launch stored compacted
                            * (ycmap): the K loop is DENSE over the compacted pitch, no per-sample row gather (w_layout 2's) */
    int64_t wsel_stride;   /* elements of the weight dtype between two weight sets                                       */
    const int32_t* order;  /* [N] or NULL: a permutation of the images -- workgroups walk images order[0], order[1], ...; images of
                            * one mode made adjacent keep that mode's weight set in L2 while they run.  Outputs, statistics rows
                            * and every per-image input keep their true image index.                                     */
    const int16_t* yperm;  /* [sets][yperm_stride] or NULL (with wsel, no order): the weight ROWS of set s were permuted at prep
                            * time (mcgen_prep_t.rmap = yperm + s * yperm_stride: image row r holds true output channel
                            * yperm[s][r] -- the consumer's active channels first, in order, then the others), so the accumulators
                            * come out in COMPACTED order: y (pitch Cy < Cout_w, a multiple of 8) receives columns 0 .. Cy - 1 with
                            * plain 16-byte stores -- the compacted output of `ycmap` without its gather pass.  bias and the
                            * statistics (pitch Cout_w) are indexed through the permutation, i.e. stay in true channel order.
                            * Columns between an image's active count and Cy hold channels its consumer masks: the consumer's
                            * weight image has zero columns there (mcgen_prep_t.kmap).  Same restrictions as ycmap.       */
    int32_t yperm_stride, reserved_;
} mcgen_conv_t;

/* number of M tiles (rows of `stats`) the launch of `p` will use.  Depends on the shape / mode fields only -- callers ask
 * BEFORE they allocate `stats`, so no kernel's eligibility may depend on p->stats (or any other buffer pointer being set) */
int mcgen_conv_m_tiles(const mcgen_conv_t* p, int dtype);
/* the (pixels x channels) output tile the launcher will pick for `p` (names the kernel instantiation) */
int mcgen_conv_tile(const mcgen_conv_t* p, int dtype, int* bm, int* bn);
/* which kernel family mcgen_conv_fused hands `p` to: 0 the tiled forms named by mcgen_conv_tile, 1 the split-K skinny
 * kernel (Cout <= 16, deep K, maps up to 16x16), 2 the whole-image kernel (8x8 maps, 128 / 256 channels), 3 the
 * resident-pixel-tile 1x1 kernel (512 -> 512 on 16x16 / 8x8 / 4x4 maps), 4 the image convolution (8 -> 128 channels, 3x3, 32x32),
 * 5 the image head (3x3 to <= 8 channels of pitch 8, 32x32) */
int mcgen_conv_form(const mcgen_conv_t* p, int dtype);
int mcgen_conv_fused(const mcgen_conv_t* p, int dtype, void* stream);

/* Weight gradient of the same fused convolution for ONE segment:
 *     dW[co, ci, kh, kw] = alpha * sum_{n,h,w} dy[n, (h,w)>>dy_ups, co] * prologue(x)[n, h+kh-1, w+kw-1, ci]
 * (autograd of nn.Conv2d / nn.Linear weights, mcgan.py:19,23,29,51,59,77,80,84,...).
 * Partial sums over pixel groups go to `slabs` [splits][weight-image of the segment];
 * mcgen_wgrad_reduce sums them in a fixed order into the fp32 master-layout gradient.
 */
typedef struct {
    mcgen_seg_t  seg;
    const void*  dy;       /* [N, H>>dy_ups, W>>dy_ups, Cdy]                                */
    int32_t N, H, W;
    int32_t Cout, Cout_w, Cdy;
    int32_t dy_ups;        /* 1: dy is the gradient of a 2x2-pooled output                  */
    float*  slabs;
    int32_t splits;
    int32_t halves;        /* 1: splits [0, splits/2) walk the first half of the pixel tiles, the rest the second half
                            * (paired discriminator pass: one slab set per half of the batch); needs even splits, m_tiles */
    float*  bias_slabs;    /* [splits*4][Cout_w] partial column sums of dy (the bias gradient), or NULL    */
} mcgen_wgrad_t;

int64_t mcgen_wgrad_slab_elems(const mcgen_wgrad_t* p);          /* floats per split */
int mcgen_wgrad(const mcgen_wgrad_t* p, int dtype, void* stream);
/* 1 when mcgen_wgrad runs `p` on the image-layer kernel (wgrad_c8.hip: bf16, conv input = the 8-channel image tensor without
 * prologue, 32x32 maps, 128 output channels, 3x3 or 1x1 -- FirstDisResBlock's convolution and shortcut, mcgan.py:76-86): a
 * stream over dy with all 128 output channels per workgroup; it wants about two workgroups per CU (`splits` ~ 512, at most
 * the number of 128-pixel steps) where the general kernels want ~128.  Its slabs are COMPACT ((tap, ci) pairs as the columns of
 * a 1x1-shaped slab, mcgen_wgrad_c8_slab_elems floats per split): reduce them with tapcols = 1. */
int mcgen_wgrad_c8_ok(const mcgen_wgrad_t* p, int dtype);
int64_t mcgen_wgrad_c8_slab_elems(const mcgen_wgrad_t* p);      /* floats per split of those compact slabs: the stride between splits */
/* The weight gradients of up to MCGEN_WGRAD_MULTI_MAX 3x3 layers of ONE backward pass in one launch (wgrad_multi.hip:
 * 128 co x 64 ci x 9 tap workgroup tiles, one accumulator set per workgroup): same operands, slab layout and `splits` /
 * `halves` meaning as mcgen_wgrad per layer, so mcgen_wgrad_reduce(_batch) finishes either.  The caller gives each layer
 * `splits` in proportion to its share of the pass's FLOPs (every workgroup then walks about the same number of 128-pixel
 * steps).  Replaces, per backward pass, the autograd weight gradients of the block convolutions of mcgan.py:19,23,77,80,
 * 103-113.  Eligible (mcgen_wgrad_multi_ok != 0): bf16, ksize 3, square maps of side 8 / 16 / 32, Cout a multiple of 128,
 * seg.C a multiple of 64, N*H*W a multiple of 128, no statistics groups / compaction map. */
#define MCGEN_WGRAD_MULTI_MAX 16
int mcgen_wgrad_multi_ok(const mcgen_wgrad_t* p, int dtype);
int mcgen_wgrad_multi(const mcgen_wgrad_t* layers, int n, int dtype, void* stream);
/* n <= MCGEN_WGRAD_MULTI_MAX layers of IDENTICAL shape (N, H, W, channel pitches, ksize, ups / dy_ups, splits, bias slabs present
 * or not; no two-half launches, no image layers) as one launch of the kernel mcgen_wgrad picks for that shape: grid z = layer x
 * split.  Same slabs per layer as n calls of mcgen_wgrad.  MCGlow's coupling networks (mcglow.py:118-160) have 16 flows per
 * level whose first (C/2 -> 512) and zero-initialised last (512 -> C) 3x3 convolutions have skinny weight gradients: 96 launches
 * of 12-25 us per step one by one (bf16 only). */
int mcgen_wgrad_batch(const mcgen_wgrad_t* layers, int n, int dtype, void* stream);
/* grad[master layout] (+)= alpha * sum_s slabs[s]; master layout = [Cout][Cin][k][k] with
 * row co stored at (co % rows_inner) * row_perm + co / rows_inner when row_perm > 1
 * (the generator's Linear(128 -> C*4*4) viewed as NHWC, mcgan.py:51,67).
 * transpose=0 only (weight gradients are always produced in forward orientation). */
int mcgen_wgrad_reduce(const float* slabs, int splits, float* grad, int Cout, int Cin, int ksize,
                       int Cout_w, int row_perm, float alpha, int accumulate,
                       const float* bias_slabs, float* bias_grad, float* bias_grad2,
                       const float* row_scale /* optional [Cout] */, int cin_slab /* 0 = Cin */,
                       int tapcols /* 1: the compact slabs of mcgen_wgrad_c8_ok layers: [chunk][Cout_w][32], column tap * 8 + ci */,
                       int tap0, int ntap_out /* as mcgen_wreduce_t; 0, 0: every tap */, void* stream);
/* bias_slabs/bias_grad (optional): bias_grad[Cout] (+)= alpha * sum_s bias_slabs[s] (same row_perm);
 * bias_grad2 receives the same values (a second conv that shares dy, e.g. the 1x1 shortcut). */

/* Build the kernel-side weight image from fp32 master weights [Cout][Cin][k][k]:
 *   image[q][tap][co_w][32] (q = input-channel chunk of 32), element type `dtype`,
 *   multiplied by wscale / (*sigma if sigma != NULL)   -- spectral norm's W / sigma,
 *   torch.nn.utils.spectral_norm as applied by models/utils.py:17-21;
 *   transpose=1 builds the image of the transposed, spatially flipped filter used for
 *   input gradients (rows = Cin, K = Cout).
 * Returns the number of elements written through *elems_out (may be NULL). */
int64_t mcgen_weight_image_elems(int Cout, int Cin, int ksize, int transpose);
int mcgen_prep_weight(const float* w, void* image, int dtype, int Cout, int Cin, int ksize,
                      int transpose, int row_perm, const float* sigma, float wscale, void* stream);

/* forward-orientation image with every output row co multiplied by row_scale[co]
 * (ZeroConv2d's exp(3*scale), mcglow.py:127-130; ActNorm.reverse folded into the inverse 1x1 conv, mcglow.py:53-55) */
int mcgen_prep_weight_rows(const float* w, void* image, int dtype, int Cout, int Cin, int ksize,
                           const float* row_scale, void* stream);
/* generalised builder: strided source [Cout][Cin][KH][KW] (element strides s_*), embedded at tap (kh0, kw0) of the
 * ksize x ksize image, source scaled by row_scale[co] * col_scale[ci] * wscale, forward or transposed+flipped, image
 * extents rows_img x k_img (zero outside the source).  Covers PixelCNN's (k/2+1) x k / 1 x (k/2+1) stacks inside 3x3
 * images (mcpixelcnn.py:29-35), Glow's exp(3*scale) / ActNorm-scaled and zero-padded coupling weights and every
 * input-gradient image of those models without intermediate tensor ops. */
int mcgen_prep_weight_ex(const float* w, int64_t s_co, int64_t s_ci, int64_t s_kh, int64_t s_kw, int Cout, int Cin,
                         int KH, int KW, int kh0, int kw0, int ksize, int transpose, int rows_img, int k_img,
                         const float* row_scale, const float* col_scale, float wscale, void* image, int dtype, void* stream);
/* many generalised images in a few launches (jobs travel by value in the kernel arguments, 16 per launch) */
#define MCGEN_PREPEX_MAX 32
typedef struct {
    const float* w; int64_t s_co, s_ci, s_kh, s_kw;
    int32_t Cout, Cin, KH, KW, kh0, kw0, ksize, transpose, rows_img, k_img;
    const float* row_scale; const float* col_scale;
    void*   image;
    float   wscale; int32_t _pad;
} mcgen_prepex_t;
int mcgen_prep_weight_ex_batch(const mcgen_prepex_t* jobs, int n, int dtype, void* stream);
/* the same for all layers of a network pass in ONE launch; sigma = sigma_base[sigma_idx] (idx < 0: none) */
typedef struct {
    const float* w; void* image;
    int32_t Cout, Cin, ksize, transpose, row_perm, sigma_idx;
    float wscale; int32_t layout;   /* 0: chunked image (mcgen_prep_weight); 1: K-major (mcgen_prep_weight_k, transpose = 0) */
    const int16_t* kmap;            /* layout 0, transpose 0: NULL, or the image's input channel k is source channel kmap[k]    */
    int32_t kcount, _pad;           /* (k < kcount; kmap[k] >= Cin: a zero column) -- a mode's compacted weight image: kmap =
                                     * the cidx part of that mode's mcgen_mc_cmap record, kcount = the compacted pitch       */
    const int16_t* rmap;            /* layout 0, transpose 0: NULL, or image row r is weight row rmap[r] (a permutation of the
                                     * output channels: mcgen_conv_t.yperm)                                                    */
} mcgen_prep_t;
int mcgen_prep_weight_batch(const mcgen_prep_t* descs_dev, int n, const float* sigma_base, int dtype, void* stream);

/* K-major weight image for mode-compacted launches: image[tap][k][co_w], k = 0 .. round_up(Cin, 8) (the last row,
 * index round_up(Cin, 8), is all zeros: padded K slots point at it), co_w = Cout rounded up to 16; bf16 or fp32,
 * multiplied by wscale / (*sigma if sigma != NULL).  A compacted K step gathers 32 rows of this image by index. */
int64_t mcgen_weight_image_k_elems(int Cout, int Cin, int ksize);
int mcgen_prep_weight_k(const float* w, void* image, int dtype, int Cout, int Cin, int ksize,
                        const float* sigma, float wscale, void* stream);

/* Per-sample compaction map of one MultimodalController code tensor code[N, C] (C a multiple of 8): record n
 * (cmap + n * stride, int16 elements, stride = mcgen_cmap_stride(C)) holds
 *     cpos[C]       position of channel c among the sample's active channels (code != 0), -1 if inactive
 *     cidx[C + 32]  the active channels in order, padded with C (the K-major image's zero row)
 *     cpre[C/32+1]  (int32 entries) number of active channels below each 32-channel boundary; the last = their count
 * modules.py:71-76: out = x * code -- a zero code entry removes the channel from every product that follows. */
/* Per-image prologue rows for a launch that reads COMPACTED activations (w_layout 2): for image n and slot j with
 * c = cidx_n[j]:  scale_out[n][j] = (scale ? scale[n / group_n][c] : 1) * code[n][c],  shift_out likewise (0 without
 * scale); slots beyond the image's active count get 0.  relu(bn(x)) * code == relu(x * scale_out + shift_out) because
 * MultimodalController codes are non-negative (modules.py:58-69).  Pass the rows as the segment's scale / shift with
 * group_n = 1 and no code. */
int mcgen_mc_affine(const float* scale, const float* shift, int group_n, const float* code, const int16_t* cmap,
                    int N, int C, int Ccap, float* scale_out, float* shift_out, void* stream);
int32_t mcgen_cmap_stride(int C);
int mcgen_mc_cmap(const float* code, int N, int C, int16_t* cmap, void* stream);

/* layout / dtype conversion at the module boundary (the reference works on NCHW fp32) */
int mcgen_nchw_to_nhwc(const float* src, void* dst, int dtype, int N, int C, int H, int W, int Cp, void* stream);
/* out[r][i][m] = (label[i] == m) for r < reps: F.one_hot(label, classes).float() (mcgan.py:196,201) written `reps` times back to
 * back -- the indicator of a paired discriminator batch (2 N rows) and of the grouped generator pass (d_iters * N rows) are
 * prefixes of one such buffer.  Labels outside [0, classes) fail the launch's precondition (the row stays all zero). */
int mcgen_onehot_rep(const int64_t* label, float* out, int* lab32, int N, int classes, int reps, void* stream);
int mcgen_nhwc_to_nchw(const void* src, float* dst, int dtype, int N, int C, int H, int W, int Cp, void* stream);
/* y[N, Ho, Wo, C] = 2x2 sums of x[N, 2 Ho, 2 Wo, C] (NHWC, C a multiple of 8): the adjoint of the nearest x2 upsample.
 * The generator's shortcut conv1x1(Up(x)) (mcgan.py:26-30,42) commutes with the upsample, so its weight gradient and
 * input gradient are taken at x's resolution from the pooled output gradient -- a quarter of the FLOPs. */
int mcgen_pool2_sum(const void* x, void* y, int dtype, int N, int Ho, int Wo, int C, void* stream);

/* code[N, C] = indicator[N, M] @ codebook[M, C]      MultimodalController.forward, modules.py:73 */
int mcgen_mc_code(const float* indicator, const float* codebook, float* code, int N, int M, int C, void* stream);
/* codes of all MultimodalController layers of a network in ONE launch: code_base + out_off <- indicator @ codebook */
typedef struct { const float* codebook; int64_t out_off; int32_t M, C; int32_t scale_idx, _pad; } mcgen_code_t;
/* optional per-sample scaling (paired discriminator pass): rows n >= n_half of job j are multiplied by
 * scale[descs[j].scale_idx] (scale_idx < 0 or scale == NULL: no scaling) */
int mcgen_mc_code_batch(const float* indicator, const mcgen_code_t* descs_dev, int n, float* code_base, int N,
                        const float* scale, int n_half, void* stream);
/* the same for one-hot indicators given as int64 labels: code_j[n, :] = codebook_j[label[n % n_label], :] (every C a multiple
 * of 4; labels outside 0 .. M-1 are clamped).  N a multiple of n_label: the label vector repeats (the indicator of a paired
 * discriminator batch is the batch's twice, of the grouped generator pass d_iters times); scale / n_half as in
 * mcgen_mc_code_batch.  With one-hot rows this IS indicator @ codebook (modules.py:73) -- a row gather instead of M
 * multiply-adds per output, which is what COIL100's 100 and Omniglot's 1623 modes need. */
int mcgen_mc_gather_batch(const int64_t* label, int n_label, const mcgen_code_t* descs_dev, int n, float* code_base, int N,
                          const float* scale, int n_half, void* stream);
/* mcgen_prep_weight_batch + mcgen_mc_gather_batch as ONE launch: the weight images W / sigma (models/utils.py:17-21 behind
 * mcgan.py's SpectralNorm wrappers) and the codes of a discriminator pass both wait for the power iteration and for
 * nothing else.  Same results as the two calls. */
int mcgen_prep_weight_batch_codes(const mcgen_prep_t* descs_dev, int n, const float* sigma_base, int dtype,
                                  const int64_t* label, int n_label, const mcgen_code_t* code_descs_dev, int n_code,
                                  float* code_base, int N, const float* scale, int n_half, void* stream);
/* y = x * code (broadcast over HW), standalone form of modules.py:75 for unfused callers;
 * x is [N, HW, C] when channels_last, else [N, C, HW] (the reference's NCHW / [N, C] inputs) */
int mcgen_mc_apply(const void* x, const float* code, void* y, int dtype, int N, int HW, int C, int channels_last, void* stream);

/* mcgen_bn_finalize_groups with the statistics groups in parallel (one workgroup row per group).  The running statistics take
 * the groups' updates in order, so this form leaves them alone: it also returns every group's unbiased variance (`unb`,
 * [groups, C]) and the caller applies the updates of ALL layers of the forward pass with one mcgen_bn_running_batch launch
 * (running = (1 - m) running + m stat, group after group: the arithmetic of the serial form). */
int mcgen_bn_finalize_par(const float* partials, int tiles, int pitch, int fold, int C, double count, int groups,
                          const float* gamma, const float* beta, float eps,
                          float* scale, float* shift, float* mean, float* rstd, float* unb, void* stream);
#define MCGEN_BN_RUN_MAX 24
typedef struct { float* running_mean; float* running_var; const float* mean; const float* unb; int32_t groups, C; float momentum; int32_t _pad; } mcgen_bn_run_t;
int mcgen_bn_running_batch(const mcgen_bn_run_t* jobs, int n, void* stream);
/* BatchNorm2d training statistics from per-tile partial sums (mcgan.py:15,20,55):
 * mean/var over `count` elements per channel, scale = gamma*rstd, shift = beta-mean*scale,
 * running stats updated with momentum (unbiased variance), all in one launch.
 * `fold` > 1: partial column j belongs to channel j % C (Linear output viewed as [C*4*4]). */
int mcgen_bn_finalize(const float* partials, int tiles, int pitch, int fold, int C, double count,
                      const float* gamma, const float* beta, float* running_mean, float* running_var,
                      float momentum, float eps, float* scale, float* shift, float* mean, float* rstd,
                      void* stream);
/* `groups` consecutive, equally sized runs of tiles are independent BatchNorm batches (count = elements per channel
 * of ONE group): outputs are [groups][C]; the running statistics take the groups' momentum updates one after the
 * other, in order -- what `groups` successive training-mode forwards of the module would leave behind
 * (train_gan.py:145-146 runs the generator once per discriminator update; its weights do not change in between). */
int mcgen_bn_finalize_groups(const float* partials, int tiles, int pitch, int fold, int C, double count, int groups,
                             const float* gamma, const float* beta, float* running_mean, float* running_var,
                             float momentum, float eps, float* scale, float* shift, float* mean, float* rstd,
                             void* stream);
/* several INDEPENDENT BatchNorm layers (one statistics group each) in one launch: per layer what mcgen_bn_finalize does
 * (MCGatedPixelCNN's vertical and horizontal gate BatchNorms of a layer, mcpixelcnn.py:16-20,44-56, wait for the same two
 * convolutions and for nothing else) */
#define MCGEN_BN_FIN_MAX 4
typedef struct {
    const float* partials; int32_t tiles, pitch, fold, C; double count;
    const float* gamma; const float* beta; float* running_mean; float* running_var; float momentum, eps;
    float* scale; float* shift; float* mean; float* rstd;
} mcgen_bn_fin_t;
int mcgen_bn_finalize_batch(const mcgen_bn_fin_t* jobs, int n, void* stream);
/* eval mode: scale/shift from the running statistics */
int mcgen_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean,
                         const float* running_var, float eps, int C, float* scale, float* shift, void* stream);
/* BatchNorm backward: reduce partial (sum dz, sum dz*xhat) -> dgamma, dbeta (+= when accumulate),
 * then dx = scale * (dz - dbeta/count - xhat * dgamma/count) (+ add), xhat = (x-mean)*rstd */
int mcgen_bn_bwd_finalize(const float* partials, int tiles, int pitch, int C,
                          float* dgamma, float* dbeta, float* sums, int accumulate, void* stream);
int mcgen_bn_bwd_apply(const void* dz, const void* x, const void* add, void* dx, int dtype,
                       int64_t pixels, int C, const float* sums, double count,
                       const float* scale, const float* mean, const float* rstd, void* stream);

/* column sums: out[c] (+)= alpha * sum_p x[p, c]   (bias gradients) */
int mcgen_colsum(const void* x, int dtype, int64_t rows, int C, int pitch, float* out, int row_perm,
                 float alpha, int accumulate, float* workspace, void* stream);

/* Spectral norm (torch.nn.utils.spectral_norm via models/utils.py:17-21), all layers of a
 * network in one launch.  Per layer l: W = w + off[l] viewed [rows[l]][cols[l]];
 * if do_iter: v = normalize(W^T u), u = normalize(W v) (eps 1e-12, in place); sigma[l] = u.(W v). */
typedef struct { int64_t w_off, u_off, v_off; int32_t rows, cols; } mcgen_sn_layer_t;
int mcgen_sn_power_iter(const float* w_base, float* uv_base, const mcgen_sn_layer_t* layers_dev, int nlayers,
                        int do_iter, float* sigma, float* workspace /* nlayers * (32*max_cols + max_rows) floats */,
                        int max_rows, int max_cols, void* stream);
/* One training-mode iteration (do_iter = 1) that also leaves the new u, v in `uv_snap` (same offsets as uv_base): the
 * forward's own copy for the backward pass -- torch's spectral_norm hook clones u and v (torch/nn/utils/spectral_norm.py),
 * this writes the clone from the kernels that produce the values.  Every float of the u/v buffer must belong to a layer. */
int mcgen_sn_power_iter_snap(const float* w_base, float* uv_base, const mcgen_sn_layer_t* layers_dev, int nlayers,
                             float* sigma, float* workspace, int max_rows, int max_cols, float* uv_snap, void* stream);
/* `rounds` successive training-mode power iterations of all layers in 2 * rounds + 1 launches (column-slice W^T u without
 * row partials, the next round's u formed from the previous round's t on the fly; a paired discriminator update -- two
 * rounds, train_gan.py:144-150 -- costs five launches instead of eight).  sigma: [rounds, nlayers]; uv_snap (or NULL):
 * [rounds, uv_total] receives (u, v) after every round (the forward's copy, as mcgen_sn_power_iter_snap); workspace as
 * mcgen_sn_power_iter.  ratio (or NULL; rounds >= 2): [nlayers] receives sigma[rounds - 2] / sigma[rounds - 1], the factor a
 * paired pass scales its second half by.  Layers up to 1024 rows. */
int mcgen_sn_power_iter_rounds(const float* w_base, float* uv_base, const mcgen_sn_layer_t* layers_dev, int nlayers,
                               int rounds, float* sigma, float* workspace, int max_rows, int max_cols,
                               float* uv_snap, int64_t uv_total, float* ratio, void* stream);
/* `rounds` successive power iterations of every layer in ONE launch (one workgroup per layer; u, v, W v stay in LDS):
 * sigma[r][l] and -- when uv_snap != NULL -- the whole u/v buffer as it stands after round r (uv_snap[r][uv_total]:
 * torch's hook clones u, v for the backward, torch/nn/utils/spectral_norm.py) are written per round; uv_base holds the
 * final state.  do_iter = 0 (evaluation mode): one round, sigma = u . (W v), nothing is updated.  The two
 * training-mode forwards of a discriminator update (train_gan.py:144-150) are rounds = 2. */
int mcgen_sn_power_iter_fused(const float* w_base, float* uv_base, const mcgen_sn_layer_t* layers_dev, int nlayers,
                              int rounds, int do_iter, float* sigma, float* uv_snap, int64_t uv_total,
                              int max_rows, int max_cols, void* stream);
/* gradient through W/sigma:  dst (+)= (g - <g, W/sigma> u v^T) / sigma per layer, where g (at g_src + w_off)
 * is the gradient w.r.t. the normalised weight; u, v, sigma are the values the FORWARD used (the caller
 * keeps a snapshot per forward, as torch's hook does by cloning u and v). g_src may equal g_dst.
 * A layer entry with rows == 0 is a plain parameter (bias) of `cols` elements: dst (+)= src. */
int mcgen_sn_grad_fix(const float* g_src, float* g_dst, const float* w_base, const float* uv_base,
                      const mcgen_sn_layer_t* layers_dev, int nlayers, const float* sigma, int accumulate,
                      float* workspace /* 32 * nlayers floats */, void* stream);
/* The same for the two halves of a paired discriminator pass (each half with the u, v, sigma of its own forward,
 * train_gan.py:144-150) in ONE dot launch and ONE apply launch: dst (+)= fix(g_src0; uv0, sigma0) + fix(g_src1; uv1, sigma1).
 * workspace: 2 * 32 * nlayers floats; dst must not alias a source. */
int mcgen_sn_grad_fix_pair(const float* g_src0, const float* g_src1, float* g_dst, const float* w_base,
                           const float* uv0, const float* uv1, const mcgen_sn_layer_t* layers_dev, int nlayers,
                           const float* sigma0, const float* sigma1, int accumulate, float* workspace, void* stream);

/* Discriminator tail (mcgan.py:158-165): logit[n] = b + sum_c (w[c]/sigma) * sum_hw relu(x)*code */
int mcgen_dtail_fwd(const void* x, int dtype, const float* code, const float* w, const float* b,
                    const float* sigma, float* pooled, float* logit, int N, int HW, int C, void* stream);
int mcgen_dtail_bwd(const float* dlogit, const void* x, int dtype, const float* code, const float* w,
                    const float* sigma, const float* pooled, void* dx, float* dw, float* db,
                    int N, int HW, int C, int accumulate, void* stream);
/* Paired pass (D(real) and D(fake) as one 2N batch, gan_engine.DiscriminatorEngine.forward_pair): the tail's weight and bias
 * gradients per half.  dlogit[2N], pooled[2N][C]; dw1/db1 from rows [0, N), dw2/db2 from rows [N, 2N) with dw2 divided
 * by ratio[0] (= sigma_1 / sigma_2 of the tail layer, which the second half's pooled features carry). */
int mcgen_dtail_pair_wgrad(const float* dlogit, const float* pooled, const float* ratio, int N, int C,
                           float* dw1, float* db1, float* dw2, float* db2, void* stream);
/* The tail's forward, the hinge loss's derivative and the tail's input gradient in ONE launch (a workgroup per sample):
 * mcgen_dtail_fwd, then dlogit[n] from the sample's own logit -- mode 0: hinge_d over a paired batch (train_gan.py:154; samples
 * [0, N/2) real: -1/(N/2) where 1 - logit > 0; [N/2, N) generated: +1/(N/2) where 1 + logit > 0), mode 1: hinge_g
 * (train_gan.py:172; -1/N) -- then dx as mcgen_dtail_bwd writes it.  The loss value is mcgen_dtail_pair_wgrad_loss's /
 * mcgen_hinge_g's to add.  C a multiple of 8, at most 2048. */
int mcgen_dtail_hinge_fused(const void* x, int dtype, const float* code, const float* w, const float* b, const float* sigma,
                            float* pooled, float* logit, float* dlogit, void* dx, int N, int HW, int C, int mode, void* stream);
/* mcgen_dtail_pair_wgrad + the discriminator's hinge loss value from the 2N logits (mean relu(1 - real) + mean relu(1 + generated)) */
int mcgen_dtail_pair_wgrad_loss(const float* dlogit, const float* pooled, const float* ratio, const float* logit, int N, int C,
                                float* dw1, float* db1, float* dw2, float* db2, float* loss, void* stream);

/* Hinge losses (train_gan.py:154,172).  d: loss = mean relu(1-real) + mean relu(1+fake);
 * g: loss = -mean(fake).  Writes the loss and d(loss)/d(logit). */
int mcgen_hinge_d(const float* real, const float* fake, int N, float* loss, float* dreal, float* dfake, void* stream);
int mcgen_hinge_g(const float* fake, int N, float* loss, float* dfake, void* stream);

/* dx = dy * (1 - y*y)      (nn.Tanh backward, mcgan.py:60) */
int mcgen_tanh_bwd(const void* dy, const void* y, void* dx, int dtype, int64_t n, void* stream);

/* Adam over one flat fp32 buffer (torch.optim.Adam as configured at train_gan.py:43-47,231):
 * step = int64[2] on the device: step[0] is the counter the call increments (by the last workgroup to finish: no
 * follow-up launch), step[1] a ticket word that is 0 between calls.
 * lr_dev: NULL, or one float in device memory that holds the learning rate -- read when the kernel RUNS, so a launch
 * captured into a HIP graph follows a learning-rate scheduler (train_gan.py:105-106, train_vae.py:78-81) without a
 * re-capture; `lr` is ignored then. */
int mcgen_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, const float* lr_dev, float beta1, float beta2,
               float eps, float weight_decay, int64_t* step, void* stream);
/* mcgen_sn_grad_fix_pair with Adam's update in place of the store (a single-rank discriminator update, train_gan.py:154-158:
 * d/d(weight_orig) is never materialised): for the layers of the table, g = fix(g_src0; uv0, sigma0) + fix(g_src1; uv1, sigma1)
 * goes straight into m, v, p (p doubles as w_base: the dot <g, W> is taken before any element moves).  `advance_step` = 1:
 * the call's dot launch increments step[0] and the update launch behind it reads the new count -- a step that covers its
 * parameters with several tables passes 1 for the first table and 0 for the rest.  workspace: 2 * 32 * nlayers floats;
 * step and lr_dev as in mcgen_adam (the ticket word is not used). */
int mcgen_sn_fix_pair_adam(const float* g_src0, const float* g_src1, float* p, float* m, float* v,
                           const float* uv0, const float* uv1, const mcgen_sn_layer_t* layers_dev, int nlayers,
                           const float* sigma0, const float* sigma1, float* workspace,
                           float lr, const float* lr_dev, float beta1, float beta2, float eps, float weight_decay, int64_t* step,
                           int advance_step, void* stream);

/* ---- MCGlow-specific kernels (reference: models/mcglow.py) ------------------------------------------------- */
/* Block squeeze / unsqueeze (mcglow.py:221-223, 262-265): [N,H,W,C] <-> [N,H/2,W/2,4C], channel c*4 + 2*dh + dw.
 * inverse = 0: x is the big tensor (pitch Cp_big), y the squeezed one (pitch Cp_small); inverse = 1: the other way. */
int mcgen_glow_squeeze(const void* x, void* y, int dtype, int N, int H, int W, int C, int Cp_big, int Cp_small,
                       int inverse, void* stream);
/* per-channel (sum, sum of squares) over all pixels, written as `blocks` partial rows [blocks][2][Cp]
 * (the layout of the conv epilogue's stats) */
int mcgen_channel_stats(const void* x, int dtype, int64_t pixels, int Cp, float* partials, int blocks, void* stream);
/* ActNorm.initialize (mcglow.py:32-39): loc = -mean, scale = 1 / (unbiased std + 1e-6) from such partials */
int mcgen_actnorm_init(const float* partials, int tiles, int pitch, int C, double count, float* loc, float* scale, void* stream);
/* ActNorm.forward as a conv prologue (mcglow.py:41-51): a = scale, b = scale * loc (zero beyond C) */
int mcgen_actnorm_affine(const float* loc, const float* scale, int C, int Cp, float* a, float* b,
                         float* negloc /* optional: -loc, the gate mean of the backward */, void* stream);
/* logdet[n] += HW * (sum_c log|scale_c| + sum_c w_s_c): the parameter-only log-determinants of one flow
 * (ActNorm mcglow.py:46-47, InvConv2dLU mcglow.py:101) */
int mcgen_glow_param_logdet(const float* scale, int C, const float* w_s, int Cw, float hw, float* logdet, int N, void* stream);
/* InvConv2dLU.calc_weight (mcglow.py:105-111): W = P (L o mask + I) (U o mask + diag(sign * exp(w_s))), and
 * W^-1 (reverse, mcglow.py:113-116) when weight_inv != NULL; one workgroup, matrices in LDS, C <= 64 */
int mcgen_invconv_weight(const float* w_p, const float* w_l, const float* w_u, const float* w_s, const float* s_sign,
                         int C, float* weight, float* weight_inv, void* stream);
/* AffineCoupling.forward / reverse (mcglow.py:153-175) given the coupling network output h = [log_s | t]:
 * forward: y_b = (x_b + t) * sigmoid(log_s + 2), logdet[n] (+)= sum log sigmoid(log_s + 2); reverse: x_b = y_b / s - t */
int mcgen_glow_coupling(const void* x, const void* h, void* y, int dtype, float* logdet, int N, int HW, int C, int Cp,
                        int reverse, int accumulate, void* stream);
/* gaussian_log_p summed per sample / gaussian_sample (mcglow.py:16-21, 229-238, 253-262); prior = [mean | log_sd] */
int mcgen_gaussian_logp(const void* z, int Cpz, int c0, const void* prior, int Cpp, int dtype, int N, int HW, int Cz,
                        float* logp, int accumulate, void* stream);
int mcgen_gaussian_sample(const void* eps, int Cpe, const void* prior, int Cpp, void* out, int Cpo, int c0, int dtype,
                          int64_t pixels, int Cz, void* stream);
/* dst[..., c0:c0+Cn] = src[..., s0:s0+Cn]   (split / concat of the multi-scale architecture) */
int mcgen_copy_channels(const void* src, int Cps, int s0, void* dst, int Cpd, int c0, int dtype, int64_t pixels, int Cn, void* stream);

/* every split-K reduction of one backward pass in one launch (the per-layer mcgen_wgrad_reduce calls, batched);
 * slab size is derived from (Cin, ksize, Cout_w) as in mcgen_wgrad_slab_elems */
#define MCGEN_WREDUCE_MAX 32
typedef struct {
    const float* slabs;       /* [splits][slab_elems]                       */
    float*       grad;        /* [Cout][Cin][k][k] master layout             */
    const float* bias_slabs;  /* [splits*4][Cout_w] or NULL                  */
    float*       bias_grad;   /* [Cout] or NULL                              */
    float*       bias_grad2;  /* optional second destination                 */
    int32_t splits, Cout, Cin, ksize, Cout_w, row_perm, accumulate;
    float   alpha;
    const float* row_scale;   /* optional [Cout]: grad row co is scaled by row_scale[co] (output-side factors such as
                               * ActNorm's scale or ZeroConv2d's exp(3*scale) that follow the convolution)     */
    int32_t cin_slab;         /* channel count the slabs were built for (the padded activation); 0 = Cin        */
    int32_t tapcols;          /* 1: compact slabs of an image-layer launch (mcgen_wgrad_c8_ok): (tap, ci) pairs are columns */
    int32_t tap0, ntap_out;   /* ntap_out > 0: grad is [Cout][Cin][ntap_out] and receives taps tap0 .. tap0 + ntap_out - 1 of the
                               * k x k slab only -- a (rows x cols) sub-kernel embedded in the 3x3 image, e.g. MCGatedMaskedConv2d's
                               * (2 x 3) vertical / (1 x 2) horizontal stacks (mcpixelcnn.py:29-35): taps 0-5 / 3-4.  0: all k * k */
} mcgen_wreduce_t;
int mcgen_wgrad_reduce_batch(const mcgen_wreduce_t* jobs, int n, void* stream);

/* ---- MCGlow backward (autograd of the same lines) ------------------------------------------------------------ */
/* coupling backward: dv = [dy_a | dy_b * s], dh = [dy_b (v_b + t) s (1 - s) + g (1 - s) | dy_b * s], g = dL/dlogdet_n */
int mcgen_glow_coupling_bwd(const void* v, const void* h, const void* dy, void* dv, void* dh, int dtype, float g,
                            int64_t pixels, int C, int Cp, void* stream);
/* prior backward: dz[..., d0:d0+Cz] (+)= -g (z - mean) e^{-2 lsd}; dprior = [+g (z - mean) e^{-2 lsd} | g (-1 + (z - mean)^2 e^{-2 lsd})] */
int mcgen_gaussian_logp_bwd(const void* z, int Cpz, int c0, const void* prior, int Cpp, void* dz, int Cpd, int d0,
                            void* dprior, int dtype, float g, int64_t pixels, int Cz, int accumulate_dz, void* stream);
/* out[c] (+)= alpha * sum_p a[p, c] * b[p, c]   (ZeroConv2d scale gradient: 3 * sum out * dout) */
int mcgen_prod_colsum(const void* a, int pitch_a, const void* b, int pitch_b, int dtype, int64_t pixels, int C,
                      float* out, float alpha, int accumulate, float* workspace /* 256*C floats */, void* stream);
/* ActNorm loc/scale gradients from dgrad-epilogue partials (sum d, sum d * (x + loc)); ld_coef = dL/dlogdet * N*H*W */
int mcgen_actnorm_bwd(const float* partials, int tiles, int pitch, int C, const float* scale, float ld_coef,
                      int input_side, float* dloc, float* dscale, int accumulate, void* stream);
/* InvConv2dLU: (dw_l, dw_u, dw_s) from the gradient of the C x C weight (row pitch ldw) */
int mcgen_invconv_bwd(const float* w_p, const float* w_l, const float* w_u, const float* w_s, const float* s_sign,
                      const float* dW, int C, int ldw, float ld_coef, float* dw_l, float* dw_u, float* dw_s,
                      int accumulate, void* stream);
/* ---- batched forms of the per-module MCGlow kernels above: all modules of a pass in one launch per kind (the job
 * table travels by value, MCGEN_GLOW_BATCH_MAX jobs per launch).  Same arithmetic per job as the single forms; the
 * parameter-only log-determinants of all flows are added to logdet[n] as ONE sum (mcglow.py:46-47,101). */
#define MCGEN_GLOW_BATCH_MAX 24
#define MCGEN_GLOW_PLD_MAX 64
typedef struct { const float* loc; const float* scale; float* a; float* b; float* negloc; int32_t C, Cp; } mcgen_an_affine_t;
typedef struct { const float* scale; const float* w_s; int32_t C, Cw; float hw; int32_t _pad; } mcgen_pld_t;
typedef struct { const float *w_p, *w_l, *w_u, *w_s, *s_sign; float* weight; float* weight_inv; int32_t C, _pad; } mcgen_icw_t;
typedef struct { const float *w_p, *w_l, *w_u, *w_s, *s_sign; const float* dW; float *dw_l, *dw_u, *dw_s;
                 int32_t C, ldw, accumulate; float ld_coef; } mcgen_icb_t;
typedef struct { const float* partials; const float* scale; float* dloc; float* dscale;
                 int32_t tiles, pitch, C, input_side, accumulate; float ld_coef; } mcgen_an_bwd_t;
typedef struct { const void* a; const void* b; float* out; int64_t pixels; int32_t pitch_a, pitch_b, C, accumulate;
                 float alpha; int32_t _pad; } mcgen_pcs_t;
int mcgen_actnorm_affine_batch(const mcgen_an_affine_t* jobs, int n, void* stream);
int mcgen_glow_param_logdet_batch(const mcgen_pld_t* jobs, int n, float* logdet, int N, void* stream);
int mcgen_invconv_weight_batch(const mcgen_icw_t* jobs, int n, void* stream);
int mcgen_invconv_bwd_batch(const mcgen_icb_t* jobs, int n, void* stream);
int mcgen_actnorm_bwd_batch(const mcgen_an_bwd_t* jobs, int n, void* stream);
int mcgen_prod_colsum_batch(const mcgen_pcs_t* jobs, int n, int dtype, float* workspace /* n * 256 * max C floats */, void* stream);
/* torch.nn.utils.clip_grad_norm_(params, max_norm) over one flat gradient buffer (train_vae.py:110) */
int mcgen_clip_grad_norm(float* g, int64_t n, float max_norm, float* norm_out, float* workspace /* 256 floats */, void* stream);

/* ---- MCPixelCNN (models/mcpixelcnn.py) -------------------------------------------------------------------------
 * The (k/2+1) x k vertical and 1 x (k/2+1) horizontal stacks of the 3x3 layers are 3x3 convolutions with zero taps
 * (mcpixelcnn.py:29-35,50-54: asymmetric kernel + crop == shifted taps) and run on mcgen_conv_fused; the 7x7 mask-A
 * layer goes through im2col + the fused 1x1 convolution. */
/* col[n,ho,wo, t*Cp + c] = X[n, ho*stride + t/KW - oh, wo*stride + t%KW - ow, c] (0 outside), [Ho,Wo] = [H,W]/stride,
 * X = relu?(x*scale + shift) * code (all optional).  col2im is the adjoint in gather form (+ bias when not
 * accumulating).  4x4 taps, stride 2, oh = ow = 1: im2col + 1x1 conv == nn.Conv2d(.., 4, 2, 1) (mcvae.py:40,45),
 * 1x1 conv + col2im == nn.ConvTranspose2d(.., 4, 2, 1) (mcvae.py:89,95). */
int mcgen_im2col(const void* x, void* col, int dtype, int N, int H, int W, int Cp, int KH, int KW, int oh, int ow,
                 int stride, const float* scale, const float* shift, int relu, const float* code, void* stream);
int mcgen_col2im(const void* dcol, void* dx, int dtype, int N, int H, int W, int Cp, int KH, int KW, int oh, int ow,
                 int stride, const float* bias, int C, int accumulate, void* stream);
/* MCGatedActivation.forward (mcpixelcnn.py:16-20): s = [a | b] with 2C channels (pitch 2C);
 * out[.., C] = code * relu(a * scale + shift) * sigmoid(b), (scale, shift) = the BatchNorm affine of this batch */
int mcgen_gated_fwd(const void* s, const float* scale, const float* shift, const float* code, void* out, int dtype,
                    int N, int HW, int C, void* stream);
/* the same for up to MCGEN_GATED_MAX independent gates in one launch (a layer's vertical and horizontal gate) */
#define MCGEN_GATED_MAX 4
typedef struct { const void* s; const float* scale; const float* shift; const float* code; void* out; int32_t N, HW, C, _pad; } mcgen_gated_t;
int mcgen_gated_fwd_batch(const mcgen_gated_t* jobs, int n, int dtype, void* stream);
/* its backward, pass 1: ds = [dz | db] and per-block partial sums (sum dz, sum dz * xhat) as [blocks][2][C] */
int mcgen_gated_bwd_stats(const void* s, const float* scale, const float* shift, const float* mean, const float* rstd,
                          const float* code, const void* g, void* ds, float* partials, int blocks, int dtype,
                          int N, int HW, int C, void* stream);
/* pass 2, in place on ds[.., :C]: da = scale * (dz - (S1 + xhat * S2) / count); sums = [S1 | S2] from mcgen_bn_bwd_finalize */
int mcgen_gated_bwd_apply(void* ds, const void* s, const float* sums, const float* scale, const float* mean,
                          const float* rstd, double count, int dtype, int64_t pixels, int C, void* stream);
/* BN -> (ReLU) -> MC -> (+res) -> (ReLU) tails: y = post_relu?( pre_relu?(x*scale + shift) * code + res )
 * (mcpixelcnn.py:37-40,57-60 horiz_resid; mcvae.py:17-35 ResBlock; mcvae.py:40-48,86-92 stage activations) */
int mcgen_affine_code_res(const void* x, const float* scale, const float* shift, const float* code, const void* res,
                          void* y, int dtype, int N, int HW, int C, int pre_relu, int post_relu, void* stream);
/* y[n, ho, wo, c] = max over the 2x2 window of relu(x * scale[c] + shift[c]): Conv -> BatchNorm2d (eval) -> ReLU -> MaxPool2d(2)
 * of the COIL100 / Omniglot feature network behind IS / FID (models/classifier.py:17-29, metrics/metrics.py:49-62,89-113);
 * x is [N, 2 Ho, 2 Wo, C], C a multiple of 8 (padding channels: scale = shift = 0). */
int mcgen_affine_relu_maxpool2(const void* x, const float* scale, const float* shift, void* y, int dtype,
                               int N, int Ho, int Wo, int C, void* stream);
/* backward of such a tail, pass 1: g' = g * [y_post > 0] (if y_post; also written to g_gated = the residual's gradient),
 * dz = g' * code * [x*scale + shift > 0 if pre_relu], and the BatchNorm-backward partial sums of dz over x */
int mcgen_code_bn_stats(const void* g, const float* code, const void* x, const float* mean, const float* rstd,
                        void* dz, float* partials, int blocks, int dtype, int N, int HW, int C,
                        const float* scale, const float* shift, int pre_relu, const void* y_post, void* g_gated, void* stream);
/* VAE reconstruction term (mcvae.py:10-11): recon = sigmoid(logits), partials[b] = block sums of BCE(recon, target),
 * dlogits (optional) = (recon - target) * gscale; target fp32 in the logits' NHWC pitch */
int mcgen_bce_logits(const void* logits, const float* target, void* recon, void* dlogits, float* partials, int blocks,
                     float gscale, int dtype, int64_t pixels, int C, int Cp, void* stream);
/* F.cross_entropy(logits, codes) per pixel (mcpixelcnn.py:100): loss_rows[p] = logsumexp - logit[target];
 * dlogits (optional) = (softmax - onehot) * gscale */
int mcgen_cross_entropy(const void* logits, const int64_t* target, float* loss_rows, void* dlogits, float gscale,
                        int dtype, int64_t pixels, int C, int Cp, void* stream);

/* nearest-code search of VectorQuantization.forward (modules.py:20-25): idx[p] = argmin_c x[p, c] (first minimum);
 * x = |e_c|^2 - 2 <f_p, e_c> comes from one fused 1x1 convolution over the codebook */
int mcgen_argmin_channels(const void* x, int64_t* idx, int dtype, int64_t pixels, int C, int Cp, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MCGEN_HIP_H */
