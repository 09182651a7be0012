/*
 * ppms.h -- C ABI of libppms (MI355X / gfx950 kernels for the PPMStereo hot path).
 *
 * The reference (cocowy1/PPMStereo) has no native code: every op below replaces a PyTorch call site on the path
 * PPMStereo.forward_update_block (/root/reference/models/core/ppmstereo.py:426-594).  Each entry point cites the
 * reference interface it replaces.  Conventions:
 *   - plain pointers and sizes only; all pointers are DEVICE pointers owned by the caller (PyTorch caching
 *     allocator); the library never allocates, frees or retains device memory;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*); no call synchronises the device;
 *   - return 0 on success, a negative PPMS_E* code otherwise; ppms_last_error() gives the thread-local message;
 *   - "SP" tensors are the library's internal activation format: channel-last [pixel][ld] bf16 hi plane and
 *     bf16 lo plane with x ~= hi + lo (fp32-accurate bf16x3 MFMA operands).  pixel = (frame*H + y)*W + x.
 */
#ifndef PPMS_H
#define PPMS_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PPMS_OK 0
#define PPMS_EINVAL (-1)   /* shape / argument contract violated (e.g. W too small for the pyramid) */
#define PPMS_ELAUNCH (-2)  /* HIP launch error */
#define PPMS_ENODEV (-3)   /* no gfx950 device */

#define PPMS_ABI_VERSION 4

int ppms_version(void);
const char* ppms_last_error(void);
/* 0 when a gfx950 device is current; fills name (<= 63 chars) */
int ppms_device_info(char* name, int name_cap, int* cu_count, int* clock_mhz);

/* ---------------------------------------------------------------- correlation (models/core/corr.py) */
/* CorrBlock1D.__init__ + CorrBlock1D.corr, corr.py:56-72,96-104.
 * fmap1,fmap2: fp32 NCHW (B,C,H,W).  pyr[l] (l=0..4): fp32 (B*H*W, W>>l) with W>>l floored; level 4 is stored
 * but never read, as in the reference.  vol = sum_c f1 f2 / sqrt(C); level l+1 = avg of adjacent pairs. */
int ppms_corr_build(const float* fmap1, const float* fmap2, float* const pyr[5], int B, int C, int H, int W, void* stream);

/* CorrBlock1D.__call__ + bilinear_sampler + coords_grid, corr.py:74-94,10-27,47-52.
 * flow: fp32, (B,2,H,W) NCHW when flow_nhwc == 0, [pixel][2] otherwise (only channel 0 is read).
 * out_nchw (optional): fp32 (B,36,H,W).  out_hi/out_lo (optional): SP [pixel][out_ld], channels 36..63 zeroed.
 * flow_sp_hi/lo (optional): SP copy of both flow channels written at [pixel][flow_sp_ld] + {0,1}. */
int ppms_corr_lookup(const float* const pyr[4], const float* flow, int flow_nhwc, float* out_nchw, void* out_hi, void* out_lo,
                     int out_ld, void* flow_sp_hi, void* flow_sp_lo, int flow_sp_ld, int B, int H, int W, void* stream);

/* ---------------------------------------------------------------- implicit-GEMM convolution */
/* Replaces every cuDNN conv2d/conv3d call of the update block (ppmtereo_update.py:292-310, 473-480, 673-674,
 * 889-893, 910-914, 646, 129): out[cout][pixel] = sum_{tap,ci} W[cout][tap][ci] * X[ci][pixel + tap], zero padded,
 * computed as hi*hi + hi*lo + lo*hi on bf16 MFMA with fp32 accumulation. */
typedef struct ppms_sp {          /* a channel-last split-bf16 view */
    void* hi;
    void* lo;
    int32_t ld;                   /* elements between pixels */
    int32_t c;                    /* channels in this view (inputs: multiple of 32) */
} ppms_sp;

enum {                            /* epilogue kinds */
    PPMS_EPI_STORE = 0,           /* y = act(acc + bias) * scale                                 */
    PPMS_EPI_RESID = 1,           /* y = act(aux_sp + acc + bias)                                  */
    PPMS_EPI_RH = 2,              /* y = sigmoid(acc + bias) * aux_sp               (GRU r * h)    */
    PPMS_EPI_GRU = 3,             /* y = (1 - z) * aux_sp + z * tanh(acc + bias), z = aux_f32      */
    PPMS_EPI_ADDF32 = 4           /* out_f32 += acc + bias (in place)              (flow += dflow) */
};
enum { PPMS_ACT_NONE = 0, PPMS_ACT_RELU = 1, PPMS_ACT_GELU = 2, PPMS_ACT_SIGMOID = 3, PPMS_ACT_TANH = 4, PPMS_ACT_ELU1 = 5 /* elu(x)+1 */ };

typedef struct ppms_epilogue {
    int32_t kind, act;
    float scale;
    int32_t n_valid;              /* couts [0, n_valid) of this half are stored                    */
    ppms_sp out_sp;               /* optional (hi == NULL: skip)                                   */
    float* out_f32;               /* optional [pixel][out_f32_ld]                                  */
    int32_t out_f32_ld;
    int32_t vt_f16;               /* format of out_vt: 0 = bf16(y); 1 = the fp16 image of bf16(y), saturated at +-65504 -- the same numbers
                                   * (every bf16 value of magnitude 2^-14 .. 65504 is an fp16 value; below, fp16 subnormals keep 2^-25 absolute)
                                   * in the operand format of ppms_mem_attn's PPMS_ATTN_P_FP16 mode.  (Sits in what was padding: ABI size unchanged.) */
    void* out_vt;                 /* optional 16-bit [frame][n_valid][H*W] (transposed, attention V), format: vt_f16 */
    ppms_sp aux_sp;               /* residual / h                                                  */
    const float* aux_f32;         /* z                                                             */
    int32_t aux_f32_ld;
    int32_t pre_f32_ld;
    const float* pre_f32;         /* optional [pixel][pre_f32_ld]: added to acc + bias before the activation -- the
                                   * contribution of input channels that do not change between iterations (computed
                                   * once per scale by another launch of the same conv on those channels)         */
} ppms_epilogue;

typedef struct ppms_conv {
    ppms_sp seg[2];               /* input channel segments, concatenated along K                 */
    int32_t nseg;
    int32_t groups;               /* 0 / 1: every cout reads all input channels (the segments concatenated along K).  2: a GROUPED convolution of two
                                   * groups -- nseg == 2, input segment s feeds the couts of epilogue half s only (couts [0, m_split) read seg[0], couts
                                   * [m_split, M) read seg[1]; weights: the two convolutions' packed images interleaved per k-step, pack_conv6_grouped).
                                   * Served by ppms_conv_gemm6 only (ppms_conv_gemm6_applicable tells); every other entry point refuses it.
                                   * (Sits in what was padding: ABI size unchanged.) */
    const void* w;                /* packed weights, see ppmstereo_amd/packing.py                  */
    const float* bias;            /* [M] fp32 (never NULL; zeros when the conv has no bias)        */
    int32_t T, H, W;              /* volume; pixels = T*H*W                                        */
    int32_t kt, kh, kw;           /* odd kernel extents, "same" zero padding                       */
    int32_t M;                    /* padded couts, multiple of 64                                  */
    int32_t m_split;              /* couts >= m_split use epi[1] with cout - m_split (multiple of 64; >= M: unused) */
    int32_t t_halo;               /* frames that exist (and may be read by temporal taps) before frame 0 and after frame T-1 of
                                   * every input segment: 0 = zero padding at both ends (one GPU holds the whole window); > 0 when
                                   * the window's frames are sharded over GPUs and the neighbours' boundary frames sit in halo
                                   * slabs around this rank's T frames (ppmstereo_amd/dist.py).  Outputs cover [0, T) only. */
    int32_t lo_zero_from;         /* > 0: the input channels from this index on (counted over the concatenated segments; a multiple of
                                   * 64) are known to hold bf16-exact values, i.e. their lo plane is all zero (the memory read-out `hid` of
                                   * the attention, ppmstereo.py:550-552, is a bf16 tensor): the kernels skip the products with that plane
                                   * -- the same bits, a third of those channels' MFMAs saved.  A PROMISE by the caller: with a non-zero
                                   * lo plane there the result silently loses that plane's contribution.  0: no such channels. */
    ppms_epilogue epi[2];
} ppms_conv;

/* desc: host copy (validated, sizes the grid, and -- since ABI v2 -- copied into the kernel arguments at launch, so it may be
 * changed or freed as soon as the call returns); dev_desc: the same bytes in device memory (caller-owned; must be non-NULL, kept in
 * the signature for ABI stability: the kernels no longer read it -- a descriptor in the kernel arguments saves a dependent memory
 * round trip at the head of every launch and is known not to alias the kernel's stores). */
/* Data-reuse tiling: all couts of a pixel tile per workgroup, LDS activation window swept by the kw taps.
 * desc->w must be in the pack_conv2 layout (ppmstereo_amd/packing.py).
 * wm_hint: 64-cout blocks per workgroup (1..4), 0 = let the library choose from the grid size. */
int ppms_conv_gemm2(const ppms_conv* desc, const ppms_conv* dev_desc, int wm_hint, void* stream);
/* K-sliced form of the same kernel for small maps (1/16 and 1/8 scales: fewer workgroups than CUs, long K loops):
 * nslice workgroups share each output tile and leave fp32 partial sums in `workspace` (caller-owned,
 * ppms_conv_gemm2_slice_workspace_bytes() bytes), a second launch sums them in slice order (deterministic) and runs the
 * fused epilogue.  ppms_conv_gemm2_slices() returns the slice count that pays off for a descriptor (1: use
 * ppms_conv_gemm2).  Not for epilogues with out_vt. */
int ppms_conv_gemm2_slices(const ppms_conv* desc);
int64_t ppms_conv_gemm2_slice_workspace_bytes(const ppms_conv* desc, int nslice);
int ppms_conv_gemm2_sliced(const ppms_conv* desc, const ppms_conv* dev_desc, int nslice, void* workspace, void* stream);
/* Convs with kh > 1 by the same kernel with ONE halo'd window per (dt, chunk) for all their taps (ppms_conv_gemm2 loads a
 * window per kernel row): (kt, kh, 1) kernels swept along y (weights: pack_conv2 order with the kh / kw axes swapped),
 * kernels with kw > 1 as well through a 2-D window (weights: (ky, kx) flattened into x) -- the sweep-ordered packs
 * ppms_conv_gemm5 / ppms_conv_gemm6 use too.  nslice / workspace as for ppms_conv_gemm2_sliced (nslice == 1: none);
 * ppms_conv_gemm2_ysweep_slices() = the slice count that pays off (0: not a candidate / window does not fit). */
int ppms_conv_gemm2_ysweep_slices(const ppms_conv* desc);
int ppms_conv_gemm2_ysweep(const ppms_conv* desc, const ppms_conv* dev_desc, int nslice, void* workspace, void* stream);
/* Barrier-free k-loop (conv_gemm5.hip): the weights are packed in MFMA-fragment order (ppmstereo_amd/packing.py pack_conv4,
 * sweep-ordered: y-swept kernels with kh / kw swapped, 2-D swept ones with (ky, kx) flattened into x) and go from L2 straight to registers; the activation window holds 16 channels and is
 * double buffered, so the loop synchronises once per window instead of once per k-step.  One 8-wave workgroup per CU owns ALL
 * couts (M == 256, or M == 128 with the K loop split between two wave groups) of a tile of nbt = 7 or 8 blocks of 32 pixels, split
 * 4 + 3 (4 + 4) so that every SIMD carries the same load: 51 200 pixels = 240 tiles of 224 on 256 CUs.  nbt = 0: the library picks. */
int ppms_conv_gemm5_applicable(const ppms_conv* desc);
int ppms_conv_gemm5(const ppms_conv* desc, const ppms_conv* dev_desc, int nbt, void* stream);
/* K-sliced form of the same kernel for maps with fewer tiles than CUs (the 1/8 and 1/16 scales): nslice workgroups share each
 * tile, each sweeps its share of the activation windows and writes fp32 partial sums to the caller's workspace
 * (ppms_conv_gemm2_slice_workspace_bytes(desc, nslice)); the slice-reduce kernel then sums them in slice order and runs the fused
 * epilogue (bit-reproducible).  ppms_conv_gemm5_slices: the slice count that fills the chip in one round, 0 = not applicable. */
int ppms_conv_gemm5_slices(const ppms_conv* desc);
int ppms_conv_gemm5_sliced(const ppms_conv* desc, const ppms_conv* dev_desc, int nbt, int nslice, void* workspace, void* stream);
/* Large-map kernel on v_mfma_f32_16x16x32_bf16 (conv_gemm6.hip, round 5): ONE wave per SIMD -- a 4-wave workgroup per CU owns all couts
 * (M == 256 / 192: 64 / 48 couts x 13 pixel blocks per wave; M == 128: 64 couts x 7 or 6 blocks) of a tile of 16 rows x 13 columns = 13
 * blocks of 16 pixels: 51 200 pixels = 250 tiles on 256 CUs.  Weights in MFMA-fragment order for 16-cout x 32-channel blocks
 * (ppmstereo_amd/packing.py pack_conv6, sweep-ordered like pack_conv4), straight from L2 to registers one k32-step ahead; 32-channel
 * activation windows, column-major, double buffered by LDS-DMA through a buffer resource (the lo plane of a segment must follow its hi
 * plane inside one 4 GiB range), swept by the taps along x, along y (kh <= 5) or over all kh x kw taps; without a spatial sweep (kh = kw = 1:
 * (kt,1,1) and 1x1 convolutions) every k32-step streams its own window through a ring of three buffers.  Same descriptor and epilogues as ppms_conv_gemm5 except
 * out_vt; input segments in multiples of 32 channels.  applicable: 0 = not served, 1 = served and the 208-pixel tiles fill >= 85 % of the CU-slots of the launch's rounds,
 * 2 = served with a poor fill (a caller keeps ppms_conv_gemm5 there). */
int ppms_conv_gemm6_applicable(const ppms_conv* desc);
int ppms_conv_gemm6(const ppms_conv* desc, const ppms_conv* dev_desc, void* stream);
/* Thin GEMM for 1x1 convolutions / Linear layers (gemm1.hip): kt = kh = kw = 1, K = 128 / 192 / 256 / 384 / 512 input channels in one or two
 * 16-channel-aligned segments, M % 32 == 0, weights in the pack_gemm1 layout ([M/32][K/16][hi, lo][lane][8]: the MFMA A-operand image).
 * One workgroup = four waves that split K between them, both operands straight to registers, partial tiles summed through LDS in wave
 * order, the shared row epilogue (every kind but ADDF32; out_vt with a STORE epilogue): one memory round trip deep, no slices. */
/* 0: not served; 1: served and the faster choice (maps of <= 16 384 pixels, or <= 64 couts); 2: served, the implicit GEMM is as fast */
int ppms_gemm1_applicable(const ppms_conv* desc);
/* cb_hint: 32-cout blocks per workgroup (1, 2, 4), 0 = let the library choose from the grid size */
int ppms_gemm1(const ppms_conv* desc, const ppms_conv* dev_desc, int cb_hint, void* stream);
/* Register-streamed convolution for small maps (conv_stream.hip; the 1/8 and 1/16 scales, where ppms_conv_gemm2 needs K slices and a reduce
 * launch to fill the chip): any odd (kt, kh, kw), input channels a multiple of 64 in one or two 16-channel-aligned segments, M % 64 == 0,
 * weights in the pack_stream layout ([M/32][tap][K/16][hi, lo][lane][8], taps in natural order).  One workgroup = four waves that share the
 * 16-channel chunks of every tap of ONE 32- or 64-pixel x 64-cout tile; both operands go straight to registers through a ring of requests
 * (no LDS, no barrier in the K loop); partial tiles summed through LDS in wave order; the shared row epilogue (every kind but ADDF32, no
 * out_vt).  No workspace, no second launch, bit-reproducible.  Replaces, for the same ppms_conv descriptor, what the reference runs as one
 * cuDNN convolution (ppmtereo_update.py:254-312, 445-482, 670-678, 889-893, 910-914).
 * applicable: 0 = not served; 1 = served AND measured faster than the LDS-staged kernels with their K slices + reduce launch; 2 = served,
 * but the LDS-staged kernels (ppms_conv_gemm2 / _sliced) win -- a binding sends a convolution here only on 1.  The rating
 * (conv_stream.hip, profiles/r04_conv_stream_probe.txt): maps of <= 4 096 pixels while pixels x K x M <= 4e9 (K = taps x input channels);
 * larger maps only without spatial taps (kt > 1 or K >= 768) up to 32 768 pixels, and 64-cout convolutions up to 16 384 pixels. */
int ppms_conv_stream_applicable(const ppms_conv* desc);
/* hint: 0 = the library chooses the tile from the grid size; 1 / 2 = 32- / 64-pixel tiles.  Any other value is refused (PPMS_EINVAL). */
int ppms_conv_stream(const ppms_conv* desc, const ppms_conv* dev_desc, int hint, void* stream);
/* sizeof(ppms_sp), sizeof(ppms_epilogue), sizeof(ppms_conv) as compiled: lets a foreign-language binding check its
 * struct layout at load time */
int ppms_struct_sizes(int* sp, int* epilogue, int* conv);

/* ---------------------------------------------------------------- small ops of the update block */
/* depthwise k x k conv + residual GELU of PCBlock4_Deep_nopool_res (ppmtereo_update.py:1026-1027):
 * y = gelu(x + dw(x) + b) on an SP tensor; w: fp32 [C][k*k], b: fp32 [C]; k in {1,7}. */
int ppms_dwconv_gelu(ppms_sp x, ppms_sp y, const float* w, const float* b, int k, int BT, int H, int W, void* stream);
/* im2col of the 2-channel flow for convf1 (7x7, ppmtereo_update.py:452,477): patch[pixel][tap*2+c], 98 -> 128 zero padded */
int ppms_flow_patch7(const float* flow_nhwc, ppms_sp patch, int BT, int H, int W, void* stream);
/* uncertainty tail: sigmoid(w . x + b) per pixel (ppmtereo_update.py:891-892) + per-frame partial sums for the
 * QAM frame confidence (ppmstereo.py:506).  unc: fp32 [pixel]; partial: fp32 [BT][nblk], nblk = ceil(H*W/256). */
int ppms_unc_tail(ppms_sp x, const float* w, float bias, float* unc, float* partial, int BT, int HW, void* stream);

/* layout converters between the reference's NCHW fp32 tensors and SP / channel-last fp32 */
int ppms_nchw_to_sp(const float* src, ppms_sp dst, int BT, int C, int HW, void* stream);
int ppms_sp_to_nchw(ppms_sp src, float* dst, int BT, int C, int HW, void* stream);
int ppms_nchw_to_nhwc(const float* src, float* dst, int dst_ld, int BT, int C, int HW, void* stream);
int ppms_nhwc_to_nchw(const float* src, int src_ld, float* dst, int BT, int C, int HW, void* stream);
int ppms_f32_to_sp(const float* src, int src_ld, ppms_sp dst, int64_t pixels, void* stream);
int ppms_sp_to_f32(ppms_sp src, float* dst, int dst_ld, int64_t pixels, void* stream);

/* Tail of a few-output-channel conv evaluated as a 1x1 GEMM to (taps*cout) channels followed by a shifted sum:
 * out[p][c] = bias[c] + sum_tap y[p + d(tap)][tap*cout + c] (zero padded).  FlowHead3D.conv2, ppmtereo_update.py:674.
 * accum (may be NULL; fp32 [pixel][accum_ld]): accum[p][c] += out[p][c] in the same launch -- flow = flow + delta_flow, ppmstereo.py:571. */
int ppms_tap_gather_sum(const float* y, int y_ld, const float* bias, float* out, int out_ld, float* accum, int accum_ld, int cout, int kt, int kh,
                        int kw, int T, int H, int W, int t_halo, void* stream);   /* t_halo: frames of y readable beyond [0, T) (see ppms_conv) */
/* flow = flow + delta_flow (ppmstereo.py:571); flow: fp32 [pixel][2], dflow: fp32 [pixel][dflow_ld] */
int ppms_flow_add(float* flow_nhwc, const float* dflow, int dflow_ld, int64_t pixels, void* stream);
/* PPMStereo.convex_upsample, ppmstereo.py:185-197.  flow: fp32 [pixel][2]; mask: fp32 [pixel][mask_ld] (144 used);
 * out: fp32 NCHW (BT,2,4H,4W). */
int ppms_convex_upsample(const float* flow_nhwc, const float* mask, int mask_ld, float* out, int BT, int H, int W, void* stream);
/* PPMStereo.convex_upsample_3d, ppmstereo.py:199-228 (use_convex_3d=True): 27 spatio-temporal neighbours of one window of T
 * frames; mask: fp32 [pixel][mask_ld] (432 = 27 * 16 used, channel 16 k + 4 i + j, k = (kt*3 + ky)*3 + kx);
 * out: fp32 NCHW (T,2,4H,4W).  unfoldNd (absent third-party module) restated from its published semantics (zero padding). */
int ppms_convex_upsample_3d(const float* flow_nhwc, const float* mask, int mask_ld, float* out, int T, int H, int W, int t_halo, void* stream);
/* F.interpolate(mode="bilinear") on NCHW fp32, both align_corners modes (ppmstereo.py:578,580-587; utils.py:10-16);
 * out = mul * interp(src). */
int ppms_bilinear(const float* src, float* dst, int N, int C, int H, int W, int OH, int OW, int align_corners, float mul, void* stream);

/* Scale-to-scale hand-over of the cascade without leaving the SP format (ppmstereo.py:726-732, 763-767): per frame,
 * dst = a * dst + b * interp(src), interp = F.interpolate(mode="bilinear", align_corners=True) from HxW to OHxOW.
 * src, dst: SP views with the same channel count (multiple of 8). */
int ppms_sp_resize_blend(ppms_sp src, ppms_sp dst, int N, int H, int W, int OH, int OW, float a, float b, void* stream);

/* Pre-loop glue of PPMStereo.forward (ppmstereo.py:620-682), NCHW fp32: avg_pool2d(k, stride k) of `planes` HxW planes;
 * out = a x + b y[i mod period] (plain or frame-broadcast axpby); net = tanh((f[:, :128] + c[:, :128]) / 2),
 * inp = relu((f[:, 128:] + c[:, 128:]) / 2) for f, c of shape (N, 256, HW). */
int ppms_avgpool(const float* src, float* dst, int planes, int H, int W, int k, void* stream);
int ppms_axpby(const float* x, const float* y, float* out, float a, float b, int64_t period, int64_t n, void* stream);
int ppms_ctx_mix(const float* fmap, const float* ctx, float* net, float* inp, int N, int HW, void* stream);

/* ---------------------------------------------------------------- feature encoder (fnet) pieces
 * BasicEncoder(output_dim=256, norm_fn="instance"), extractor.py:302-423.  Its convolutions run on
 * ppms_conv_gemm2 / ppms_conv_gemm5; the stride-2 layers (extractor.py:306-308, 341-343, 366) as stride-1 convolutions over a
 * 2x2 space-to-depth copy of their input: dst[(n, i, j)][(2 dy + dx) * C + c] = src[(n, 2 i + dy, 2 j + dx)][c].
 * ppms_img_s2d: src = the NCHW fp32 image batch (N, C, H, W) handed to the encoders (extractor.py:400-405, convnext.py:257),
 * factor k (phase = k dy + dx); channels >= k*k*C of dst are zeroed.  ppms_sp_s2d: src, dst split-plane views, factor 2,
 * dst.c == 4 * src.c.  H, W multiples of the factor. */
int ppms_img_s2d(const float* img_nchw, ppms_sp dst, int N, int C, int H, int W, int k, void* stream);   /* k = 2 (fnet), 4 (cnet stem) */
int ppms_sp_s2d(ppms_sp src, ppms_sp dst, int N, int H, int W, void* stream);
/* nn.InstanceNorm2d(affine=False) (extractor.py:326-329, 364): per (sample, channel) mean and 1 / sqrt(biased var + eps) over the
 * HW pixels of x (channel-last fp32 [N * HW][ld], a convolution's fp32 output) -> stats[N][C][2] (pixel slices merged in fixed
 * order: deterministic; caller-owned workspace of ppms_instnorm_workspace_bytes); then
 * out = relu?( (x - mean) * rstd + res? ) as split planes (res: optional residual view, e.g. relu(x + y) of extractor.py:345;
 * channels >= C of out are zeroed). */
int64_t ppms_instnorm_workspace_bytes(int N, int HW, int C);       /* per-slice partial statistics of ppms_instnorm_stats */
int ppms_instnorm_stats(const float* x, int ld, int N, int HW, int C, float eps, float* stats, void* workspace, void* stream);
int ppms_instnorm_apply(const float* x, int ld, const float* stats, ppms_sp res, int relu, ppms_sp out, int N, int HW, int C, void* stream);

/* ---------------------------------------------------------------- context encoder (cnet) pieces
 * Feature("tiny", 256) = frozen ConvNeXt-V2-tiny + FPN decoder, convnext.py:50-264.  Its Linear / conv layers
 * run on ppms_conv_gemm2 (1x1 over channel-last data; the patchify stem 4x4 s4 and the 2x2 s2 downsamplers over space-to-depth
 * copies); these are the rest:
 * ppms_dwconv: depthwise k x k conv + bias (Block.dwconv :60; k = 7), weights [C][k*k]; x split planes -> y fp32 [pixel][ldy].
 * ppms_layernorm_any: LayerNorm over the C channels of a pixel, eps given (:11-35, both data formats; 1e-6), fp32 -> split planes.
 * ppms_grn: GRN (:37-48): out = gamma * (x * Nx) + beta + x, Nx = ||x||_2 over a sample's pixels / (its mean over channels + 1e-6);
 *   x fp32 [N * HW][ld] (the GELU output), caller-owned workspace of ppms_grn_workspace_bytes; deterministic.
 * ppms_sp_upsample2: nn.Upsample(scale_factor=2) (nearest, :226-238) of a split-plane view into another (e.g. a channel range of the
 *   concatenation buffer of :259-261). */
int ppms_dwconv(ppms_sp x, float* y, int ldy, const float* w, const float* b, int k, int N, int H, int W, void* stream);
int ppms_layernorm_any(const float* x, int ld, const float* w, const float* b, float eps, ppms_sp out, int64_t pixels, int C, void* stream);
int64_t ppms_grn_workspace_bytes(int N, int HW, int C);
int ppms_grn(const float* x, int ld, const float* gamma, const float* beta, ppms_sp out, int N, int HW, int C, void* workspace, void* stream);
int ppms_sp_upsample2(ppms_sp src, ppms_sp dst, int N, int H, int W, void* stream);

/* ---------------------------------------------------------------- pick-and-play memory attention */
/* PPMStereo.compute_qk_similarity, ppmstereo.py:397-423.  q,k: fp32 [T][H*W][ld] channel-last (128 channels);
 * pooled: workspace fp32 [2][T][(H/4)*(W/4)]; sim: fp32 [T][T], sim[i][j] = cos(kbar_i, qbar_j). */
int ppms_qk_similarity(const float* q, const float* k, int ld, float* pooled, float* sim, int T, int H, int W, void* stream);
/* its two halves, for a window whose frames are sharded over GPUs: pooled descriptors of this rank's frames
 * (pooled: fp32 [2][T][(H/4)*(W/4)]), then -- after an all-gather of the descriptors -- the T x T cosine matrix */
int ppms_qk_pool(const float* q, const float* k, int ld, float* pooled, int T, int H, int W, void* stream);
int ppms_qk_cos(const float* pooled, float* sim, int T, int cells, void* stream);
/* QAM score + top-k pick + usage counter, ppmstereo.py:501-513, and the normalised play scores of :532-533.
 * conf_partial: output of ppms_unc_tail; strive: fp32 [T][T] in/out; sel: int32 [T][5] ascending frame ids;
 * shat: fp32 [T][5]; score (optional): fp32 [T][T]. */
int ppms_qam_select(const float* sim, float* strive, const float* conf_partial, int nblk, int HW, int32_t* sel, float* shat,
                    float* score, int T, void* stream);
/* Q = bf16(q_i + PE_i) (ppmstereo.py:518-522); qb: bf16 [T][n][128] */
int ppms_attn_prep_q(const float* q, int ld, const float* pe, void* qb, int T, int n, void* stream);
/* K' = bf16(K_j * s_hat_ij + PE_j) for the picked frames (ppmstereo.py:541,547); kb: bf16 [T][ksel][n][128] */
int ppms_attn_prep_k(const float* key, int ld, const float* pe, const int32_t* sel, const float* shat, void* kb, int T, int ksel,
                     int n, void* stream);
enum { PPMS_ATTN_P_BF16 = 0, PPMS_ATTN_P_FP16 = 1 };   /* ppms_mem_attn: p_format */
/* flash_attn_func call of ppmstereo.py:550 for all T clips + the aggregation of :552:
 * hid = bf16(softmax(Q K'^T * scale) V); mfg = mf + beta * hid.
 * qb: bf16 [T][n][128]; kb: bf16 [T][ksel][n][128]; vt: 16-bit [T][128][n] (per-frame transposed values, picked through
 * sel; format: p_format); mf / mfg: SP views (128 channels).  out_bf16 (optional): bf16 [T][n][128] raw attention output.
 * p_format: the 16-bit format in which the unnormalised probabilities P~ = exp(S - m) enter the P~ V product on the matrix cores, and the
 * format of vt.  The reference's shim evaluates that product with fp32 P and bf16-rounded V (tools/gen_golden.py:89-95; flash-attention itself
 * rounds P~ to bf16).  PPMS_ATTN_P_BF16: P~ and vt in bf16 (8 significand bits on P~).  PPMS_ATTN_P_FP16: P~ in fp16 (11 bits), vt = the fp16
 * image of the bf16-rounded values (ppms_epilogue.vt_f16 = 1): V is still "cast to bf16" as ppmstereo.py:550 does, Q K'^T is computed on
 * bf16 operands either way, the MFMA count is the same -- an eighth of the P~ rounding error (what keeps the iters = 20 cascade inside 1e-3 px).
 * mf.hi == NULL: no aggregation -- the `mfg` view receives hid itself (hi plane = hid, lo plane = 0: bf16-exact channels, see
 * ppms_conv.lo_zero_from; the caller then feeds the convolutions [mf | hid] with weights (W_mf + W_mfg | beta W_mfg), the same sum). */
int ppms_mem_attn(const void* qb, const void* kb, const void* vt, const int32_t* sel, int ksel, float scale, const float* beta,
                  ppms_sp mf, ppms_sp mfg, void* out_bf16, int T, int n, void* split_ws, int frames_per_workgroup, int p_format, void* stream);
/* split_ws (optional, caller-owned, ppms_mem_attn_workspace_bytes(T, ksel, n) bytes, contents need not be
 * initialised): when given, every picked frame (or pair of picked frames) is processed by its own workgroups, which leave fp32 partials
 * (O, m, l) in the workspace, and a combine kernel merges them; with n % 64 == 0 this is the 64-queries-per-wave
 * LDS-DMA kernel (rescale-free accumulation + a fix-up pass for tiles it flags -- scores more than 2^60 (bf16 P~) / 2^16 (fp16 P~) above
 * the softmax reference of their query, see mem_attn.hip).  NULL = one fused
 * launch of the 32-query online-softmax kernel.
 * frames_per_workgroup: the 64-query kernel gives a workgroup ONE picked frame, or TWO consecutive ones where the one-frame grid would
 * run at least two rounds on the chip (fewer partials to write and combine).  0 = the library's choice (1 or 2), 1 .. 5 = this call uses that
 * many (5 = all picked frames in one workgroup: one partial set per clip; a per-call argument: the library keeps no mutable state).  The
 * same softmax either way (different summation order). */
int64_t ppms_mem_attn_workspace_bytes(int T, int ksel, int n);

/* Fused chain of up to three per-pixel (1x1, <= 64 input channels) layers with GELU, optional residual from the chain
 * input and optional depthwise-1x1 post step: the ffn1 / pw / ffn2 parts of PCBlock4_Deep_nopool_res
 * (ppmtereo_update.py:1024-1030).  dev_params: device copy of the parameter block built by the host
 * (ppmstereo_amd/engine.py: PwChain; layout checked with ppms_pwchain_param_bytes). */
int ppms_pwchain(const void* dev_params, int64_t pixels, void* stream);
int ppms_pwchain_param_bytes(void);

/* ---------------------------------------------------------------- update_block16 time / space attention pieces */
/* TimeAttnBlock core (ppmtereo_update.py:603-606 with Attention.forward :409-417): per pixel, tokens = its T frames;
 * y = LayerNorm(x); out_t = sum_t2 softmax(y_t . y_t2 / sqrt(48)) y_t2 per head (q = k = v = y).  x, out: SP, 384 channels
 * (update_block16) or 256 (the SST block's TimeAttnBlock, ppmstereo.py:392-393). */
int ppms_time_attn(ppms_sp x, const float* ln_w, const float* ln_b, ppms_sp out, int T, int n, int heads, void* stream);
/* out = resid + LayerNorm(x) (resid.hi == NULL: no residual); x fp32 [pixel][ld]; C = 384 or 256 (attention.py:186-190) */
int ppms_layernorm(const float* x, int ld, const float* w, const float* b, ppms_sp resid, ppms_sp out, int64_t pixels, int C, void* stream);
/* LinearAttention.forward (attention.py:73-100) per frame and head: Q, K already elu()+1, V already / n;
 * kv_ws: fp32 workspace of ppms_linear_attention_workspace_floats(T, n, heads, dh) floats (the per-pixel-split partial sums);
 * out (SP) = (Q KV) / (Q . sum K + 1e-6) * n */
int64_t ppms_linear_attention_workspace_floats(int T, int n, int heads, int dh);
int ppms_linear_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* kv_ws, ppms_sp out, int T, int n,
                          int heads, int dh, void* stream);

#ifdef __cplusplus
}
#endif
#endif
