/*
 * srbh.h -- C ABI of libsrbh (MI355X / gfx950 native kernels for the SR building-height hot path).
 *
 * The reference (lauraset/Super-resolution-building-height-estimation) has NO native/FFI
 * interface: its boundary for this path is Python nn.Module duck typing + state_dict keys
 * (SURVEY.md 8b).  This header is therefore the build-defined C boundary that the Python mirror
 * modules (`super-resolution-building-height-estimation_amd/ *.py`) bind with ctypes; each entry
 * point names the reference call it stands in for (paths relative to /root/reference).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated;
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream);
 *   - every function returns 0 on success, a negative srbh error or a positive hipError_t;
 *     srbh_last_error() returns a thread-local message for the last failure;
 *   - the library allocates nothing persistent: weights/workspaces are caller-owned buffers.
 *
 * Device data layouts (see DESIGN.md "Data layout in HBM")
 *   ACT16  : fp16 activations, chunk-planar, zero-bordered:  [B][C/32][H+2][W+2][32]
 *            (one 64-byte record per pixel per 32-channel chunk; the 1-pixel border is kept zero
 *             by construction so a 3x3 conv never bounds-checks)
 *   RES32  : fp32 residual stream, [B][H][W][64]
 *   NHWC32 : fp32 output, [B][H][W][C]  (== a torch channels_last (B,C,H,W) tensor)
 *   WPACK16: fp16 weights in MFMA A-fragment order: [Cin/32][tap 9][kstep 2][Cout/32][lane 64][8]
 */
#ifndef SRBH_H
#define SRBH_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRBH_OK 0
#define SRBH_ERR_ARG (-1)        /* bad argument / unsupported shape (reference: assert / NotImplementedError) */
#define SRBH_ERR_WORKSPACE (-2)  /* workspace too small */

/* ---- library ------------------------------------------------------------------------------ */
int srbh_version(void);
const char* srbh_last_error(void);
/* Diagnostics: which form of the head entry points ran since the last reset (SURVEY 8b "no silent fallback": the choice between a
 * specialised persistent kernel and the general template, or one fused pass and two launches, depends on shapes / element types and is
 * invisible in the results).  out[i] for i < n in the order below; returns the number of counters. */
#define SRBH_PATH_HCONV16 0             /* srbh_hconv_h16 -> persistent 16 -> 16 3x3 kernel */
#define SRBH_PATH_HCONV_TEMPLATE 1      /* srbh_hconv_f32 / _h16 -> one-tile-per-workgroup template */
#define SRBH_PATH_ENTRY_FUSED 2         /* srbh_hconv_entry_h16 -> one pass */
#define SRBH_PATH_ENTRY_SPLIT 3         /* ... -> two template launches */
#define SRBH_PATH_WGRAD16 4             /* srbh_hconv_wgrad_b16 -> persistent 16 -> 16 kernel */
#define SRBH_PATH_WGRAD_B16_GENERIC 5
#define SRBH_PATH_WGRAD_F32 6
#define SRBH_PATH_WGRAD_ENTRY_FUSED 7   /* srbh_hconv_wgrad_entry_b16 -> one pass */
#define SRBH_PATH_WGRAD_ENTRY_SPLIT 8
#define SRBH_PATH_HCONV_UP 9             /* srbh_hconv_h16, pixelshuffle2 == 2 -> persistent Upsampler 16 -> 64 kernel */
#define SRBH_PATH_HBLOCK16 11            /* srbh_hblock16_eval: a plain BasicBlock of the inference head in one pass */
#define SRBH_PATH_HBWD16 10              /* srbh_hbwd16: BatchNorm apply + weight gradient + data gradient of a 16 -> 16 conv in one pass */
int srbh_path_counters(unsigned long long* out, int n, int reset);

/* ---- layout helpers (used by tests and by the Python mirror at module boundaries) ------------ */
/* bytes of an ACT16 buffer including the read slack the tiled kernels need */
size_t srbh_act16_bytes(int B, int C, int H, int W);
/* NCHW fp32 -> ACT16 (interior only; border must already be zero). C is padded up to a multiple of 32. */
int srbh_nchw32_to_act16(const float* src, void* dst, int B, int C, int H, int W, void* stream);
/* ACT16 -> NCHW fp32 (first C channels) */
int srbh_act16_to_nchw32(const void* src, float* dst, int B, int C, int H, int W, void* stream);

/* ---- weights ------------------------------------------------------------------------------ */
/* bytes of a WPACK16 image of an OIHW 3x3 weight (Cin, Cout padded up to multiples of 32) */
size_t srbh_wpack16_bytes(int cout, int cin);
/* OIHW fp32 [cout][cin][3][3] -> WPACK16.  Stands in for nn.Conv2d's weight tensor
 * (SR/rrdbnet_arch.py:125-129,197-204): same values, rounded to fp16, re-ordered once. */
int srbh_pack_conv3x3_f16(const float* w_oihw, int cout, int cin, void* packed, void* stream);

/* ---- generic fused 3x3 convolution (the hot kernel) ----------------------------------------
 * y = conv3x3(in, w) + bias, stride 1, zero pad 1  (nn.Conv2d(...,3,1,1): SR/rrdbnet_arch.py:125-129)
 * with the reference's surrounding elementwise ops fused into the epilogue:
 *   upsample2x : the input is read through F.interpolate(scale_factor=2, mode='nearest')
 *                (SR/rrdbnet_arch.py:236-237): in[y>>1][x>>1]; H,W below are the OUTPUT size
 *   res_scale  : y *= res_scale (0.2 of SR/rrdbnet_arch.py:143) before adding residuals
 *   res1       : y += res1 (RES32), then res1 := y when res1_update   (x5*0.2 + x, :143)
 *   res2       : y = y*res2_scale + res2 (RES32), res2 := y when res2_update (out*0.2 + x, :167)
 *   skip       : y += skip (RES32, read only)                    (feat + body_feat, :234)
 *   lrelu      : LeakyReLU(0.2)                                  (:131,206)
 *   out16      : store fp16 into chunks [out16_chunk0, +Cout/32) of an ACT16 buffer
 *   out32      : store fp32 NHWC with out32_c channels (only the first out32_c of Cout are stored)
 */
typedef struct srbh_conv3x3_args {
    const void* in;         /* ACT16, C_in = 32*in_chunks channels starting at chunk in_chunk0 */
    int in_chunks_total;    /* chunk planes per image in the `in` buffer */
    int in_chunk0;
    int in_chunks;          /* K = in_chunks*32*9 */
    const void* w;          /* WPACK16 for (cout, 32*in_chunks) */
    const float* bias;      /* [cout] fp32 or NULL */
    int cout;               /* 32 or 64 */
    int B, H, W;            /* output geometry (input is H/2 x W/2 when upsample2x) */
    int upsample2x;
    int lrelu;
    float res_scale;        /* multiplies (conv+bias) when res1 is given; ignored otherwise */
    float* res1;
    int res1_update;
    float res2_scale;
    float* res2;
    int res2_update;
    const float* skip;
    void* out16;            /* ACT16 or NULL */
    int out16_chunks_total;
    int out16_chunk0;
    float* out32;           /* NHWC32 or NULL */
    int out32_c;
    int out16_nhwc;         /* 1: `out16` is a dense fp16 NHWC tensor [B][H][W][32*out16_chunks_total] (no border), this conv's channels
                             * at chunk out16_chunk0 -- the hand-off format of the head's 16-bit kernels (SRBH_IO_SRC0_H16).  64 -> 64
                             * convs without residual epilogue only (the persistent tail kernel); anything else is an error */
} srbh_conv3x3_args;

int srbh_conv3x3_f16(const srbh_conv3x3_args* a, void* stream);
/* The same convolution for the GRADIENT side of the RRDBNet training path (reference SR/rrdbnet_arch.py:538-592 differentiates the
 * generator; round 3, SURVEY 8f-4 second slice).  bf16 != 0: the ACT16 input / output planes and the WPACK16 weights hold bf16
 * (srbh_pack_conv3x3_b16; gradients need fp32's exponent range), products on v_mfma_f32_32x32x16_bf16.  mask16 != NULL: the output is
 * multiplied by the LeakyReLU derivative taken from the SAVED fp16 activation plane(s) -- chunks mask_chunk0.. of an ACT16 buffer with
 * mask_chunks_total planes: post-activation > 0 ? 1 : 0.2 (torch's leaky_relu backward) -- before the 16-bit / fp32 stores.  No
 * nearest-x2 read in these forms. */
int srbh_conv3x3_x16(const srbh_conv3x3_args* a, int bf16, const void* mask16, int mask_chunks_total, int mask_chunk0, void* stream);
int srbh_pack_conv3x3_b16(const float* w_oihw, int cout, int cin, void* packed, void* stream);
/* the same packs (SR/rrdbnet_arch.py:136-167's conv weights as WPACK16) for MANY convs in ONE launch: table_dev = n descriptors in DEVICE memory;
 * bias_dst != NULL also copies the conv's cout bias values (the padded bias table of srbh_rrdbnet_desc); max_elems = the largest
 * srbh_wpack16_bytes(cout, cin) / 2 among them.  (A generator in training repacks every conv each iteration: 351 + 345 launches otherwise.) */
typedef struct srbh_pack3x3_desc {
    const float* w;          /* OIHW fp32 [cout][cin][3][3] */
    void* packed;
    const float* bias_src;   /* or NULL */
    float* bias_dst;         /* or NULL */
    int cout, cin, bf16, pad_;
} srbh_pack3x3_desc;
int srbh_pack_conv3x3_many(const srbh_pack3x3_desc* table_dev, int n, long max_elems, void* stream);
/* NHWC fp32 [B][H][W][C] (C % 32 == 0) * scale -> chunk planes chunk0.. of an ACT16 buffer with chunks_total planes, as fp16 or bf16 */
int srbh_nhwc32_to_act16(const float* src, void* dst, int B, int C, int H, int W, int chunks_total, int chunk0, float scale, int bf16,
                         void* stream);
/* out[nchunk*32] = per-channel sums over (B,H,W) of nchunk ACT16 planes starting at chunk0 (bias gradients); fp16 or bf16 elements */
int srbh_act16_channel_sum(const void* src, int B, int H, int W, int chunks_total, int chunk0, int nchunk, int bf16, float* out,
                           void* stream);
/* weight gradient of a 3x3 conv on ACT16 tensors: x fp16 planes (channels 0..cin-1), dy bf16 planes (channels dy_ch0..+cout);
 * cin, cout, dy_ch0 multiples of 16, cout <= 64; dw OIHW fp32; ws = srbh_hwgrad_ws_bytes(cout, cin, 3) bytes (declared below) */
int srbh_act16_wgrad_b16(const void* x, int x_chunks_total, int cin, const void* dy, int dy_chunks_total, int dy_ch0, int cout,
                         int B, int H, int W, float* dw, float* ws, void* stream);
/* dst = a * x + b * y on fp32 vectors (y may be NULL; dst may alias x or y); n % 4 == 0 */
int srbh_axpby_f32(float* dst, float a, const float* x, float b, const float* y, long n, void* stream);
/* g *= (y > 0 ? 1 : slope), in place: the backward of F.leaky_relu (SR/rrdbnet_arch.py:234-239's activations) from the saved OUTPUT y; n % 4 == 0 */
int srbh_lrelu_bwd_f32(float* g, const float* y, float slope, long n, void* stream);
/* the adjoint of F.interpolate(scale_factor=2, mode='nearest') (SR/rrdbnet_arch.py:236-237) on an NHWC fp32 tensor: g [B][2Ho][2Wo][C] -> out [B][Ho][Wo][C],
 * the sum of each 2 x 2 block ((a + b) + (c + d)); C % 4 == 0 */
int srbh_up2_bwd_nhwc_f32(const float* g, float* out, int B, int Ho, int Wo, int C, void* stream);
/* F.interpolate(scale_factor=2, mode="bilinear", align_corners=False) on an NHWC fp32 tensor (UNetDiscriminatorSN, SR/rrdbnet_arch.py:285-297):
 * backward == 0: x [B][H][W][C] -> out [B][2H][2W][C]; backward != 0: x = the gradient [B][2H][2W][C] -> out = its adjoint [B][H][W][C] (a gather:
 * no atomics, fixed summation order).  C % 4 == 0. */
int srbh_bilinear2x_nhwc_f32(const float* x, float* out, int B, int H, int W, int C, int backward, void* stream);

/* conv_first (SR/rrdbnet_arch.py:197,232): 3x3 conv on the NCHW fp32 network input with few input
 * channels (3, 12 or 48), computed in fp32 on the vector ALUs.  Writes the 64-channel result to up to
 * three RES32 buffers (feat, rdb residual, rrdb residual) and as fp16 to chunks 0..1 of `out16`. */
int srbh_conv_first_f32(const float* x_nchw, const float* w_oihw, const float* bias, int B, int cin, int H, int W,
                        float* res_a, float* res_b, float* res_c, void* out16, int out16_chunks_total, void* stream);

/* ---- whole network: RRDBNet.forward_feature / forward (SR/rrdbnet_arch.py:208-240) -------------- */
typedef struct srbh_conv_w {
    const void* w;      /* WPACK16 */
    const float* bias;  /* fp32 [cout] */
} srbh_conv_w;

typedef struct srbh_rrdbnet_desc {
    int num_in_ch;              /* channels of x as seen by conv_first (after pixel_unshuffle) */
    int num_block;              /* RRDB blocks (23) */
    const float* conv_first_w;  /* OIHW fp32, used by srbh_conv_first_f32 */
    const float* conv_first_b;
    const srbh_conv_w* rdb;     /* [num_block*3*5] in body.{i}.rdb{r}.conv{k} order */
    srbh_conv_w conv_body, conv_up1, conv_up2, conv_hr;
    srbh_conv_w conv_last;      /* cout padded to 32; only used when want_forward */
    int num_out_ch;
} srbh_rrdbnet_desc;

size_t srbh_rrdbnet_workspace_bytes(int B, int H, int W, int want_forward);
/* x: NCHW fp32 (B,num_in_ch,H,W).  out: NHWC32 (B,4H,4W,64) for forward_feature (want_forward=0,
 * no activation after conv_hr -- SR/rrdbnet_arch.py:238) or (B,4H,4W,num_out_ch) for forward
 * (want_forward=1, lrelu(conv_hr) then conv_last -- :221-222).  want_forward=2: forward_feature with `out` a dense fp16 NHWC
 * tensor (B,4H,4W,64) -- the same values rounded once (RNE) in conv_hr's epilogue: what the head's fp16-operand kernels would make
 * of the fp32 tensor while staging it, so the height maps of the fp16 head mode are bit-identical and conv_hr writes / the block
 * entry reads half the bytes (harness paths only; the nn.Module API returns fp32).
 * ws: srbh_rrdbnet_workspace_bytes() bytes that were ZERO when first handed to this (B,H,W) geometry
 * (the kernels never write the zero borders, so the same workspace can be reused across calls). */
int srbh_rrdbnet_forward(const srbh_rrdbnet_desc* d, const float* x, float* out, int B, int H, int W,
                         int want_forward, void* ws, size_t ws_bytes, void* stream);


/* Training path of the 3 * num_block dense blocks (SR/rrdbnet_arch.py:538-592 differentiates the generator), the launch
 * loops of rrdbnet_autograd.py's "fast" mode on the device side of the C ABI.
 * forward: xr and xrr hold (copies of) conv_first's output (B,H,W,64) fp32 on entry and the trunk output on return; dense_all = 3 *
 * num_block + 1 zero-bordered ACT16 buffers of 6 planes, dense_stride bytes apart: buffer i keeps [x | x1 | x2 | x3 | x4] of RDB i.
 * backward: g_a = gradient of the trunk output on entry, g_b / g_c scratch of the same size, *g_out = which of the three holds the
 * gradient of the trunk input; packs = per RDB the five bf16 gradient-conv packs (srbh_pack_conv3x3_b16 of the stacked transposed +
 * flipped weight slices, at pack_off[0..4] inside a pack_stride-byte record); G = zero-bordered 6-plane ACT16 buffer(s) for the bf16
 * gradients [g5 | g4 | g3 | g2 | g1]: g_stride > 0 = two of them, g_stride bytes apart -- the weight / bias gradients of an RDB then
 * run on an internal side stream next to the gradient convs of the RDB below (joined before the call returns control of `stream`); dw_all = per RDB 239 616 floats (conv1..conv5, OIHW each), db_all = per RDB 192 floats in G's
 * channel order; wgrad_ws = srbh_rrdbnet_trunk_wgrad_ws_bytes() bytes. */
int srbh_rrdbnet_trunk_train_forward(const srbh_rrdbnet_desc* d, float* xr, float* xrr, void* dense_all, size_t dense_stride, int B, int H,
                                     int W, void* stream);
/* the same forward as ONE launch of the persistent trunk kernel (SR/rrdbnet_arch.py:136-167 x 69, as the inference trunk runs it) over a row
 * of dense buffers, every plane stored whole for the backward, the fp32 output in xr (pixel order); bit-identical to the per-layer call.
 * aux = srbh_rrdbnet_trunk_train_aux_bytes(B, H, W) bytes of scratch (0: geometry not taken); *used = 0: nothing ran, use the call above. */
size_t srbh_rrdbnet_trunk_train_aux_bytes(int B, int H, int W);
int srbh_rrdbnet_trunk_train_forward_persistent(const srbh_rrdbnet_desc* d, float* xr, float* xrr, void* dense_all, size_t dense_stride, int B, int H,
                                                int W, void* aux, void* stream, int* used);
size_t srbh_rrdbnet_trunk_wgrad_ws_bytes(void);   /* size of wgrad_ws below: the five weight-gradient workspaces of a dense block side by side (their reduces run as one pair of launches) */
int srbh_rrdbnet_trunk_train_backward(int num_block, const void* dense_all, size_t dense_stride, const void* packs, size_t pack_stride,
                                      const size_t* pack_off, float* g_a, float* g_b, float* g_c, float** g_out, void* G, size_t g_stride,
                                      float* dw_all, float* db_all, float* wgrad_ws, int B, int H, int W, void* stream);
/* the same backward (SR/rrdbnet_arch.py:538-592's l_g_total.backward() through the 69 dense blocks, SR/rrdbnet_arch.py:136-167) with the 345
 * data-gradient convs as ONE launch of the persistent trunk kernel's bf16 form: G_all = a ROW of num_block * 3 + 1 zero-bordered gradient buffers
 * g_stride bytes apart (buffer k = the k-th RDB from the end), the saved forward planes as LeakyReLU masks; the weight / bias gradients follow on
 * two streams (wgrad_ws = TWO workspaces of srbh_rrdbnet_trunk_wgrad_ws_bytes()) or, given trunk_wgrad_ws, as one launch (srbh_trunk_wgrad).  zero_bias = 64 zero floats; aux = the persistent forward's
 * scratch.  g_a is only read; *g_out is g_b or g_c.  Same operands and per-conv summation order as the call above, the fp32 streams agree with it
 * to the last bits.  *used = 0: nothing ran, use the call above. */
int srbh_rrdbnet_trunk_train_backward_persistent(int num_block, const void* dense_all, size_t dense_stride, const void* packs, size_t pack_stride,
                                                 const size_t* pack_off, const float* zero_bias, const float* g_a, float* g_b, float* g_c,
                                                 float** g_out, void* G_all, size_t g_stride, float* dw_all, float* db_all, float* wgrad_ws,
                                                 void* trunk_wgrad_ws, int B, int H, int W, void* aux, void* stream, int* used);
/* Weight and bias gradients of ALL dense blocks in one launch + one reduce (the gradients of SR/rrdbnet_arch.py:136-167's five convs w.r.t. their
 * parameters, for every RDB): dense_all = the forward's row of saved buffers (fp16 planes), G_all = the backward's row of gradient buffers (bf16
 * planes, buffer k = the k-th RDB from the end), dw_all / db_all as srbh_rrdbnet_trunk_train_backward; ws = srbh_trunk_wgrad_ws_bytes() bytes
 * (0: geometry not taken -- 64-pixel-wide images, H % 8 == 0).  bf16 operands (the saved planes rounded RNE while staged), fp32 accumulation,
 * fixed summation order.  trunk_wgrad_ws above = this workspace (NULL there = the general kernel, RDB by RDB). */
size_t srbh_trunk_wgrad_ws_bytes(int num_block, int B, int H, int W);
int srbh_trunk_wgrad(int num_block, const void* dense_all, size_t dense_stride, const void* G_all, size_t g_stride, int B, int H, int W, float* dw_all,
                     float* db_all, void* ws, void* stream);

/* Synchronises `stream` and returns 0 if the last srbh_rrdbnet_forward on this workspace completed normally, or a
 * negative code if the persistent trunk kernel gave up waiting for a neighbour workgroup (its spins are bounded so a
 * scheduling problem shows up as an error here instead of a hung GPU).  Set SRBH_PERSISTENT=0 to force per-layer launches. */
int srbh_rrdbnet_last_status(const void* ws, int B, int H, int W, int want_forward, void* stream);

/* Workgroups per launch of the persistent tail convs (conv_up1 / conv_up2 / conv_hr, SR/rrdbnet_arch.py:234-239) issued by the calling host
 * thread: 0 = one per CU (default), n = at most n.  Each workgroup holds a CU's whole LDS for its walk; a caller that runs the feature
 * extractor BESIDE other work (harness.TrainStep's prefetch stream) leaves CUs free with this.  Returns the previous value. */
int srbh_ptail_wgs_cap(int cap);

/* Measurement hook used by bench.py: when on, srbh_rrdbnet_forward() brackets the persistent trunk kernel (the dominant
 * kernel: the 345 dense-block convs, reference SR/rrdbnet_arch.py:136-167) with HIP events on `stream`;
 * srbh_trunk_last_ms() synchronises on the closing event and returns that launch's duration in milliseconds. */
int srbh_trunk_timing(int on);
int srbh_trunk_last_ms(float* ms);
/* Name of the device kernel the last persistent-trunk launch of this process used ("none" before the first one):
 * lets bench.py label its roofline with the kernel that actually ran (it is what rocprofv3 lists). */
const char* srbh_trunk_kernel_name(void);

/* ==== head: HR feature / fusion / regression modules (SR/HRfuse.py), fp32 ==========================
 * Tensors are NHWC fp32 ([B][H][W][C]; a torch channels_last (B,C,H,W) tensor has exactly this memory).
 * HWPACK32: fp32 weights in v_mfma_f32_16x16x4_f32 A-fragment order [Cin/16][tap][4][Cout/16][lane 64]. */
size_t srbh_hpack_bytes(int cout, int cin, int ksize);
/* OIHW fp32 -> HWPACK32 (nn.Conv2d weight of conv3x3/conv1x1/default_conv, SR/HRfuse.py:11-14,95-109).
 * transpose_flip=1 packs the weight of the data-gradient convolution instead (cout/cin are then the
 * LOGICAL counts of that gradient conv: cout = original in_channels, cin = original out_channels). */
int srbh_hpack_conv_f32(const float* w_oihw, int cout, int cin, int ksize, int transpose_flip, float* packed, void* stream);

/* bytes of the per-channel sum / sum-of-squares partial buffer written by srbh_hconv_f32 (C = cout padded to 16) */
size_t srbh_bn_stats_bytes(int C);
/* 1 when srbh_hconv_h16 takes pixelshuffle2 == 2 at this input size (W % 64 == 0, H % 4 == 0) */
int srbh_hconv_up_supported(int H, int W);

/* y = conv_{ksize}(cat(pre(src0), src1)) + bias, stride 1, zero padding ksize/2
 *   pre(x) = relu?(x*pre_scale[c] + pre_shift[c])  -- BatchNorm(+ReLU) of the producer folded into this consumer
 *            (SR/HRfuse.py:146-148); pre_scale NULL = identity
 *   src1   : optional second source, concatenated after src0's channels (torch.cat, SR/HRfuse.py:187)
 *   pixelshuffle2: store through nn.PixelShuffle(2) (SR/HRfuse.py:23): out is [B][2H][2W][cout/4]
 *   stats  : if non-NULL (srbh_bn_stats_bytes), receives per-channel partial sums of y and y^2 over all
 *            B*H*W pixels (training-mode BatchNorm statistics); it is zeroed by this call */
typedef struct srbh_hconv_args {
    const float* src0; int c0;
    const float* pre_scale; const float* pre_shift; int pre_relu;
    const float* src1; int c1;
    const float* w;       /* HWPACK32 for (cout, c0+c1, ksize) */
    const float* bias;    /* [cout padded to 16] or NULL */
    int cout;             /* 1..32 or 49..64 */
    int ksize;            /* 3 or 1 */
    int B, H, W;
    int pixelshuffle2;    /* 1: PixelShuffle(2) store (SR/HRfuse.py:23), out = (B, cout/4, 2H, 2W) NHWC.  2 (srbh_hconv_h16, fp16 operands, 16 -> 64, 3x3, no
                           * pre / post ops, srbh_hconv_up_supported(H, W)): the same, with `w` / `bias` packed SUB-PIXEL-MAJOR -- row ob*16 + kk*4 + q of the
                           * pack (srbh_hpack_conv_h16 of the permuted weight) holds conv channel (kk*4 + ob)*4 + q -- for the persistent Upsampler kernel */
    float* out;
    double* stats;
    /* optional extensions (0 / NULL = off), used by the strict fp32 trunk: strided views into wider NHWC buffers,
     * LeakyReLU(0.2) and the two residual forms y*s1 + res1, (..)*s2 + res2 (SR/rrdbnet_arch.py:143,167) */
    int src0_ld, src1_ld;      /* floats between consecutive pixels of src0 / src1 (default c0 / c1) */
    int out_ld, out_coff;      /* pixel stride and first channel of the output view (default cout, 0) */
    int post_lrelu;
    const float* res1; int res1_ld; float res1_scale;
    const float* res2; int res2_ld; float res2_scale;
    /* inference-mode BasicBlock fusion (SR/HRfuse.py:146-157 with BatchNorm in eval mode): per-output-channel affine
     * y = y*post_scale[c] + post_shift[c] (applied after the bias, before res1) and a plain ReLU at the very end */
    const float* post_scale; const float* post_shift;   /* [cout padded to 16] or NULL */
    int post_relu;
    /* srbh_hconv_h16(bf16 = 0) only -- fp16 ACTIVATIONS in memory (inference: the producing epilogue rounds once, the consumer
     * stages the 8-byte channel quads as they are): bit 0 = src0 holds fp16, bit 1 = src1, bit 2 = res1, bit 3 = out is written as
     * fp16.  Such a tensor is NHWC with 2-byte elements; its *_ld stays in ELEMENTS.  An fp16 src0 takes no pre_scale / pre_relu (the
     * producer applied them), an fp16 out no PixelShuffle store and no statistics; channel counts and strides multiples of 4. */
    int io_h16;
    /* backward-statistics epilogue (srbh_hconv_h16, the persistent 16 -> 16 3x3 kernel only; SR/HRfuse.py:146-157 in reverse): the conv
     * output is the gradient da of a = relu(bn(c)).  With bstat_c (the BatchNorm input, NHWC fp32 [B][H][W][16]), the batch mean / invstd
     * and the folded affine (ms, mh: a > 0 <=> c*ms + mh > 0; both NULL = no ReLU) given, `stats` receives sum(dz) and sum(dz * xhat),
     * dz = da where the ReLU was active -- exactly what srbh_bn_bwd_reduce computes in a pass of its own (pass the same buffer to
     * srbh_bn_bwd_finalize).  No res1; cout == 16.  With a bf16 output (SRBH_IO_OUT_H16) the sums are taken from the fp32 accumulators
     * BEFORE the one rounding of the store, whereas the separate reduce pass (srbh_bn_bwd_reduce_io) sums the rounded tensor: dgamma /
     * dbeta / the mean terms of the two paths differ by bf16 rounding noise averaged over B*H*W elements (not bit-comparable;
     * tests/test_gpu_io16.py bounds it) -- fp16 also drops the out "no PixelShuffle" rule for the full 16 -> 64 conv (see pixelshuffle2). */
    const float* bstat_c; const float* bstat_mean; const float* bstat_invstd; const float* bstat_ms; const float* bstat_mh;
    /* nonzero: the caller guarantees that `stats` is all zero already (a buffer last consumed by srbh_bn_finalize_clear /
     * srbh_bn_bwd_finalize_clear, or freshly zeroed): this call then launches no zero fill of its own (round 5: 49 one-line fills per
     * training step sat as dependent launches in front of their convolutions) */
    int stats_clean;
} srbh_hconv_args;
#define SRBH_IO_SRC0_H16 1
#define SRBH_IO_SRC1_H16 2
#define SRBH_IO_RES1_H16 4
#define SRBH_IO_OUT_H16 8
int srbh_hconv_f32(const srbh_hconv_args* a, void* stream);
/* The same convolution with fp16 OPERANDS (staged activations and weights rounded to fp16, fp32 accumulate on
 * v_mfma_f32_16x16x16_f16; inputs / outputs / BatchNorm statistics stay fp32 in memory): 1/8 of the fp32 matrix-core time,
 * so the kernel runs at its HBM traffic (the head is HBM-bound; BASELINE configs[4] asks for fp16 MFMA).
 * `w` is the fp16 pack of srbh_hpack_conv_h16.  srbh_hconv_f32 remains the strict mode (<= 2e-5 against the reference). */
size_t srbh_hpack_h16_bytes(int cout, int cin, int ksize);
/* bf16 = 0: fp16 operands (forward: activations and weights are O(1));  bf16 = 1: bfloat16 operands on
 * v_mfma_f32_16x16x16_bf16 (data gradients: per-pixel gradients of a mean-reduced loss sit far below fp16's normal range,
 * bf16 keeps fp32's exponent).  The pack and the conv must use the same setting. */
int srbh_hpack_conv_h16(const float* w_oihw, int cout, int cin, int ksize, int transpose_flip, int bf16, void* packed, void* stream);
/* n such packs in ONE launch: table_dev[i] = the arguments of call i (w, cout, cin, ksize, transpose_flip, bf16, packed), in device memory;
 * max_elems = the largest srbh_hpack_h16_bytes(...) / 2 among them.  (hrfuse.py refreshes every registered head pack right behind
 * optimizer.step(): the weights the reference's torch.optim.Adam has just changed, train.py:256.) */
typedef struct srbh_hpack_desc {
    const float* w;
    void* out;
    int cout, cin, ksize, transpose_flip, bf16, pad_;
} srbh_hpack_desc;
int srbh_hpack_conv_h16_many(const srbh_hpack_desc* table_dev, int n, long max_elems, void* stream);
int srbh_hconv_h16(const srbh_hconv_args* a, int bf16, void* stream);
/* The entry of a BasicBlock with a downsample branch (SR/HRfuse.py:142-159): conv1 (3x3, `c1`) and downsample[0] (1x1, `ds`) read the
 * SAME input -- the widest tensor of each head.  One fused pass (the 1x1 is the centre tap with other weights) when both produce 16
 * channels from the same sources with 16-aligned channel counts, no pre-affine, W % 64 == 0, H % 4 == 0; otherwise exactly the two
 * srbh_hconv_h16 launches.  Same results either way (same operand rounding, fp32 accumulation). */
int srbh_hconv_entry_h16(const srbh_hconv_args* c1, const srbh_hconv_args* ds, int bf16, void* stream);

/* training-mode nn.BatchNorm2d statistics (SR/HRfuse.py:124,132,135): partial sums -> biased batch variance ->
 * scale = gamma/sqrt(var+eps), shift = beta - mean*scale; running stats updated with `momentum` and the unbiased
 * variance exactly as torch does; save_mean/save_invstd (optional) are kept for the backward pass. */
int srbh_bn_finalize(const double* stats, int C, double count, const float* gamma, const float* beta, float eps,
                     float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                     float* save_mean, float* save_invstd, void* stream);
/* The same, and the partial-sum buffer is ZEROED behind the read (self-cleaning: the next producer may pass stats_clean = 1).  One
 * launch, one block: the zero fill is ordered behind this kernel's own reads of every slot. */
int srbh_bn_finalize_clear(double* stats, int C, double count, const float* gamma, const float* beta, float eps,
                           float momentum, float* running_mean, float* running_var, float* scale, float* shift,
                           float* save_mean, float* save_invstd, void* stream);
/* eval-mode BatchNorm folded to scale/shift from the running statistics */
int srbh_bn_eval_scale_shift(int C, const float* gamma, const float* beta, const float* running_mean,
                             const float* running_var, float eps, float* scale, float* shift, void* stream);
/* out = relu(a*a_scale + a_shift + (idt*i_scale + i_shift))   (BasicBlock tail, SR/HRfuse.py:150-157);
 * i_scale NULL = identity path without downsample */
int srbh_bn_add_relu(const float* a, const float* a_scale, const float* a_shift, const float* idt,
                     const float* i_scale, const float* i_shift, float* out, long npix, int C, void* stream);
/* The same pass with fp16 tensors in memory (round 3: the training step keeps the block-internal activations c1 / c2 / downsample
 * output as fp16): io bit SRBH_BAR_A_H16 = `a` holds fp16 elements, SRBH_BAR_IDT_H16 = `idt` does; `out` stays fp32. */
#define SRBH_BAR_A_H16 1
#define SRBH_BAR_IDT_H16 2
int srbh_bn_add_relu_io(const void* a, const float* a_scale, const float* a_shift, const void* idt,
                        const float* i_scale, const float* i_shift, float* out, long npix, int C, int io, void* stream);
/* ... and ALSO writing the ReLU's activity pattern as bits (round 4): per 64 consecutive 4-channel groups four 64-bit words (word j bit l:
 * element j of group 64 k + l is > 0), srbh_relu_bits_bytes(npix, C) bytes, C % 4 == 0.  srbh_bn_bwd_reduce_io(SRBH_BN_REF_BITS) takes it
 * in place of the fp32 block output: the backward's reduce pass reads 1 bit per element instead of 4 bytes. */
size_t srbh_relu_bits_bytes(long npix, int C);
int srbh_bn_add_relu_bits(const void* a, const float* a_scale, const float* a_shift, const void* idt, const float* i_scale,
                          const float* i_shift, float* out, void* bits, long npix, int C, int io, void* stream);
/* aggregate_torch (aggregate_utils.py:29-41): data [N][H][W] fp32 -> out [N][H/step][W/step] */
int srbh_aggregate(const float* data, float* out, int N, int H, int W, int step, void* stream);
/* F.interpolate(scale_factor=2, mode='nearest') on NHWC fp32 (SR/rrdbnet_arch.py:236-237); H, W = output size */
int srbh_nearest2x_f32(const float* src, float* dst, int B, int H, int W, int C, void* stream);
/* NCHW fp32 -> NHWC fp32 */
int srbh_nchw_to_nhwc_f32(const float* src, float* dst, int B, int C, int H, int W, void* stream);

/* ---- head backward (training; the reference relies on torch autograd over SR/HRfuse.py, train.py:254-256) ---- */
/* dW[oc][ci][tap] = sum_px dy[px][oc] * X[px+tap][ci] with X = cat(pre(src0), src1) exactly as srbh_hconv_f32 reads it.
 * dw is OIHW fp32 [cout][c0+c1][ksize][ksize]; it is zeroed by this call. */
typedef struct srbh_hwgrad_args {
    const float* src0; int c0;
    const float* pre_scale; const float* pre_shift; int pre_relu;
    const float* src1; int c1;
    const float* dy;      /* NHWC [B][H][W][cout] */
    int cout; int ksize;
    int B, H, W;
    float* dw;            /* OIHW [cout][c0+c1][k][k], fully written (deterministic: fixed summation order) */
    float* ws;            /* scratch of srbh_hwgrad_ws_bytes(cout, c0 + c1, ksize) bytes: per-workgroup partial sums */
    int src0_ld, src1_ld; /* floats between consecutive pixels of src0 / src1 (0 = c0 / c1): strided views into wider NHWC
                           * buffers, e.g. the first 64 + 32k channels of a dense block's 192-channel buffer */
    int io;               /* srbh_hconv_wgrad_b16 only: SRBH_WG_SRC0_H16 = src0 holds fp16 elements (16 -> 16 3x3 form),
                           * SRBH_WG_DY_B16 = dy holds bf16 elements (its bits are the MFMA operand: no rounding) */
} srbh_hwgrad_args;
#define SRBH_WG_SRC0_H16 1
#define SRBH_WG_DY_B16 2
size_t srbh_hwgrad_ws_bytes(int cout, int cin, int ksize);
int srbh_hconv_wgrad_f32(const srbh_hwgrad_args* a, void* stream);
/* Mixed-precision form of the same gradient (torch.cuda.amp-style training; the reference itself trains in fp32,
 * train.py:254-256): X and dY are rounded to bf16 (RNE) while staged, products on v_mfma_f32_16x16x16_bf16, fp32 accumulation,
 * same workspace and fixed summation order.  Needs c0, c1, the pixel strides % 4 == 0, cout % 16 == 0 and 16-byte aligned
 * pointers; a layer outside that (the 1- / 7-channel output convs) is computed by the fp32 kernel. */
int srbh_hconv_wgrad_b16(const srbh_hwgrad_args* a, void* stream);
/* The two weight gradients of a BasicBlock ENTRY (SR/HRfuse.py:142-159: conv1 = 3x3, downsample[0] = 1x1, same input) in ONE pass over
 * that input: a3 / a1 = the arguments the two srbh_hconv_wgrad_b16 calls would take.  Fused when they describe the same sources, shape
 * and element types (cout a multiple of 16, 4-aligned channels); otherwise it runs the two calls.  Same results either way. */
int srbh_hconv_wgrad_entry_b16(const srbh_hwgrad_args* a3, const srbh_hwgrad_args* a1, void* stream);
/* The backward of ONE 3x3, 16 -> 16 convolution of a BasicBlock behind its BatchNorm, in one pass (round 5; reference graph:
 * SR/HRfuse.py:142-159 through torch autograd).  With y = bn(conv(x')), g = dL/dy and the constants of srbh_bn_bwd_finalize,
 *     dc = coef * (g' - k1 - xhat * k2),  g' = g where c*mask_scale + mask_shift > 0 (mask optional),  xhat = (c - mean) * invstd
 * is formed while the window is staged (rounded once to bf16, as srbh_bn_bwd_apply_io's bf16 store did) and feeds BOTH
 *     dw[oc][ci][tap] = sum_px dc[px][oc] * x'[px + tap][ci],   x' = relu?(x*pre_scale + pre_shift) rounded to bf16   (srbh_hconv_wgrad_b16)
 *     dx[px][ci]      = conv^T(dc, W) [+ res]                                                                          (srbh_hconv_h16, bf16)
 * without dc ever being written: 160 - 256 bytes per pixel instead of the 352 of apply + weight gradient + data gradient.
 * g / res: bf16 NHWC [B][H][W][16];  c / x / bstat_c: fp32 NHWC [B][H][W][16];  w: srbh_hpack_conv_h16(transpose_flip = 1, bf16 = 1);
 * dx: bf16 (dx_b16) or fp32;  dw: OIHW fp32 [16][16][3][3];  ws: srbh_hwgrad_ws_bytes(16, 16, 3).
 * stats (optional, srbh_bn_stats_bytes(16)): the BatchNorm-backward sums of dx as the gradient of relu?(bn'(bstat_c)) -- srbh_hconv_args'
 * bstat epilogue -- taken from the fp32 accumulators; no res then (but see relu_bits).  W % 64 == 0, H % 4 == 0 (srbh_hbwd16_supported). */
typedef struct srbh_hbwd16_args {
    const void* g; const float* c;
    const float* mean; const float* invstd; const float* coef; const float* k1; const float* k2;
    const float* mask_scale; const float* mask_shift;
    const float* x; const float* pre_scale; const float* pre_shift; int pre_relu;
    const void* w;
    int B, H, W;
    void* dx; int dx_b16;
    const void* res;
    const float* bstat_c; const float* bstat_mean; const float* bstat_invstd; const float* bstat_ms; const float* bstat_mh;
    double* stats; int stats_clean;
    float* dw; float* ws;
    /* optional (conv1's use inside a chain of blocks, hrfuse_autograd._BlockChainFn): dx (+ res) is the gradient of the PREVIOUS block's output
     * out' = relu(bn2'(c2') + idt').  With relu_bits (srbh_bn_add_relu_bits' buffer of that block) dx is masked with that ReLU, written as
     * bf16 (dx_b16 must be 1) and `stats` receives the BatchNorm-backward sums of bn2' (bstat_c = c2', bstat_mean / bstat_invstd its batch
     * statistics, no bstat_ms) over the rounded values: the previous block's srbh_bn_bwd_reduce_io pass, without the fp32 tensor. */
    const void* relu_bits;
} srbh_hbwd16_args;
int srbh_hbwd16_supported(int H, int W);
int srbh_hbwd16(const srbh_hbwd16_args* a, void* stream);

/* A whole PLAIN BasicBlock of the inference head in one pass (round 6; reference: SR/HRfuse.py:142-159 in eval mode):
 *     out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x),   conv1 / conv2 = 3x3, 16 -> 16, no bias; bn folded to (scale, shift) per channel
 * with fp16 operands and the fp16 intermediate of the two-launch chain (srbh_hconv_h16 x 2: post_scale / post_relu / out fp16, then res1 = x)
 * kept inside the compute unit: 32 bytes read and 32 (out_h16) / 64 written per pixel instead of 160.  Bit-identical to that chain.
 * x: fp16 NHWC [B][H][W][16];  w1 / w2: srbh_hpack_conv_h16(ksize 3, transpose_flip 0, bf16 0);  out: fp16 (out_h16) or fp32 NHWC.
 * W % 64 == 0, H % 4 == 0 (srbh_hblock16_supported); 8-byte aligned x / fp16 out, 16-byte aligned fp32 out and packs. */
typedef struct srbh_hblock16_args {
    const void* x;
    const void* w1; const void* w2;
    const float* scale1; const float* shift1; const float* scale2; const float* shift2;
    void* out; int out_h16;
    int B, H, W;
} srbh_hblock16_args;
int srbh_hblock16_supported(int H, int W);
int srbh_hblock16_eval(const srbh_hblock16_args* a, void* stream);

/* Deferred weight-gradient reduces (round 5; hrfuse_autograd._BlockChainFn.backward).  The reference gets dW from torch autograd
 * (train.py:254-256); nothing reads it before the optimizer / the gradient all-reduce, so the ordered reduce of the per-workgroup partial sums
 * need not sit between the chip-filling kernels of the head's backward.  Between srbh_hwgrad_defer(1) and srbh_hwgrad_flush the entry points
 * srbh_hconv_wgrad_f32 / _b16 / _entry_b16 and srbh_hbwd16 launch their kernels but QUEUE that reduce (host thread local, any number of
 * jobs); srbh_hwgrad_flush(stream) runs all queued reduces as one pair of launches (same order of additions per element) and ends the
 * deferral.  The caller keeps every `ws` / `dw` of a queued job alive until the flush. */
int srbh_hwgrad_defer(int on);
int srbh_hwgrad_flush(void* stream);
/* out = g where ref > 0 else 0   (ReLU backward with the saved output, SR/HRfuse.py:157) */
int srbh_relu_mask_mul(const float* g, const float* ref, float* out, long n, void* stream);
int srbh_add_inplace(float* a, const float* b, long n, void* stream);
/* per-channel partial sums (into a srbh_bn_stats_bytes(C) buffer, zeroed here) of dy and dy*xhat,
 * dy = g * [c*mask_scale + mask_shift > 0] (mask optional), xhat = (c - mean)*invstd (c optional: bias gradients) */
int srbh_bn_bwd_reduce(const float* g, const float* c, const float* mean, const float* invstd, const float* mask_scale,
                       const float* mask_shift, long npix, int C, double* stats, void* stream);
/* The same reduction with the block-closing ReLU backward folded in (SR/HRfuse.py:157): dy = g where relu_ref > 0, else 0; dy is also
 * written to dz_out (may be NULL) for the skip / downsample branches.  C % 4 == 0, 16-byte aligned tensors. */
int srbh_bn_bwd_reduce_relu(const float* g, const float* relu_ref, float* dz_out, const float* c, const float* mean,
                            const float* invstd, long npix, int C, double* stats, void* stream);
/* dgamma = sum dy*xhat, dbeta = sum dy, and the constants of dc = coef*(dy - k1 - xhat*k2) (coef may be NULL) */
int srbh_bn_bwd_finalize(const double* stats, int C, double count, const float* gamma, const float* invstd,
                         float* dgamma, float* dbeta, float* coef, float* k1, float* k2, void* stream);
/* srbh_bn_bwd_finalize + zero fill of `stats` behind the read (see srbh_bn_finalize_clear) */
int srbh_bn_bwd_finalize_clear(double* stats, int C, double count, const float* gamma, const float* invstd,
                               float* dgamma, float* dbeta, float* coef, float* k1, float* k2, void* stream);
int srbh_bn_bwd_apply(const float* g, const float* c, const float* mean, const float* invstd, const float* mask_scale,
                      const float* mask_shift, const float* coef, const float* k1, const float* k2, float* out, long npix,
                      int C, void* stream);
/* The two BatchNorm-backward passes with 16-bit tensors in memory (round 3; vector form only: C % 4 == 0, 256 % (C/4) == 0): the
 * gradient tensors that live INSIDE one BasicBlock's backward are bf16 (SRBH_BN_G_B16: g; SRBH_BN_OUT_B16: dz_out / out -- their
 * consumers round their operands to bf16 anyway), the saved activation c is fp16 (SRBH_BN_C_H16); relu_ref stays fp32; the
 * arithmetic is fp32, and with a bf16 dz_out the sums are taken over the rounded values (what the consumers read).
 * srbh_bn_bwd_reduce_io: relu_ref / dz_out as srbh_bn_bwd_reduce_relu (may be NULL), mask_* as srbh_bn_bwd_reduce. */
#define SRBH_BN_OUT_B16 1
#define SRBH_BN_C_H16 2
#define SRBH_BN_G_B16 4
#define SRBH_BN_REF_BITS 8   /* srbh_bn_bwd_reduce_io: relu_ref points at srbh_bn_add_relu_bits' bit buffer, not at an fp32 tensor */
#define SRBH_BN_STATS_CLEAN 16   /* srbh_bn_bwd_reduce_io: `stats` is all zero already (see srbh_hconv_args.stats_clean): no zero fill here */
int srbh_bn_bwd_reduce_io(const void* g, const float* relu_ref, void* dz_out, const void* c, const float* mean, const float* invstd,
                          const float* mask_scale, const float* mask_shift, long npix, int C, double* stats, int io, void* stream);
int srbh_bn_bwd_apply_io(const void* g, const void* c, const float* mean, const float* invstd, const float* mask_scale,
                         const float* mask_shift, const float* coef, const float* k1, const float* k2, void* out, long npix, int C,
                         int io, void* stream);
/* inverse of the PixelShuffle(2) store map: g_ps [B][2H][2W][C] -> g [B][H][W][4C] */
int srbh_ps2_inverse(const float* g_ps, float* g, int B, int H, int W, int C, void* stream);

/* ---- depthwise KxK convolution (K = 3|5, stride 1|2), fp32 NCHW, zero padding folded in (pad_t / pad_l given, bottom /
 * right implied by OH / OW): the EfficientNet encoder's `_depthwise_conv` (third-party efficientnet_pytorch called from
 * mymodels.py:242-248), for which MIOpen only has its naive fp32 kernels.  x [B][C][H][W], w [C][1][K][K],
 * y / dy [B][C][OH][OW]; bwd_weight sums in a fixed order (deterministic). */
int srbh_dwconv_fwd(const float* x, const float* w, float* y, int B, int C, int H, int W, int K, int stride, int pad_t,
                    int pad_l, int OH, int OW, void* stream);
int srbh_dwconv_bwd_data(const float* dy, const float* w, float* dx, int B, int C, int H, int W, int K, int stride,
                         int pad_t, int pad_l, int OH, int OW, void* stream);
/* ws: caller-provided scratch of srbh_dwconv_bwd_weight_splits(B, C) * C * K * K floats (per-batch-slice partial sums) */
int srbh_dwconv_bwd_weight_splits(int B, int C);
int srbh_dwconv_bwd_weight(const float* x, const float* dy, float* dw, float* ws, int B, int C, int H, int W, int K, int stride,
                           int pad_t, int pad_l, int OH, int OW, void* stream);

/* inference BatchNorm folded to a per-channel affine (srbh_bn_eval_scale_shift) + activation on NCHW fp32:
 * y = act(x * scale[c] + shift[c]), act 0 none | 1 SiLU | 2 ReLU; y may alias x. */
/* The encoder's stem at inference (round 6; smp EfficientNetEncoder.forward behind mymodels.py:276: _swish(_bn0(_conv_stem(x)))): a 3x3 conv with
 * stride `stride` and the static "same" zero padding (pt rows on top, pl columns on the left; whatever the window needs beyond the image on the
 * other sides is zero as well) + the folded BatchNorm (scale, shift per output channel) + activation (0 none, 1 SiLU, 2 ReLU) in ONE pass over
 * NCHW fp32 tensors.  w: OIHW fp32.  Cin <= 16, Cout in {32, 40, 48} (srbh_stem_conv_eval_supported: the stems of efficientnet-b0..b5; wider stems keep the stock conv). */
int srbh_stem_conv_eval_supported(int Cin, int Cout, int K);
int srbh_stem_conv_eval(const float* x, const float* w, const float* scale, const float* shift, float* y, int B, int Cin, int H, int W,
                        int Cout, int stride, int pt, int pl, int OH, int OW, int act, void* stream);
int srbh_affine_act_nchw(const float* x, const float* scale, const float* shift, float* y, int B, int C, int HW, int act,
                         void* stream);
/* y = act(x * scale[c] + shift[c]) + res: the closing BatchNorm of an MBConv block together with its skip connection
 * (efficientnet_pytorch MBConvBlock.forward: x = bn2(project_conv(x)); x = x + inputs), one pass instead of two. */
int srbh_affine_act_add_nchw(const float* x, const float* scale, const float* shift, const float* res, float* y, int B, int C, int HW,
                             int act, void* stream);

/* squeeze-and-excitation of an MBConv block at inference (efficientnet_pytorch MBConvBlock.forward, called through
 * mymodels.py:242-248): (1) affine + activation as above, also writing the per-plane mean pooled [B][C];
 * (2) hidden [B][SQ] = swish(b1 + w1 [SQ][C] . pooled); (3) y[plane (b,c)] *= sigmoid(b2[c] + w2 [C][SQ] . hidden[b]), in place. */
int srbh_affine_act_pool_nchw(const float* x, const float* scale, const float* shift, float* y, float* pooled, int B, int C,
                              int HW, int act, void* stream);
int srbh_se_hidden(const float* pooled, const float* w1, const float* b1, float* hidden, int B, int C, int SQ, void* stream);
int srbh_se_gate_scale(float* y, const float* hidden, const float* w2, const float* b2, int B, int C, int SQ, int HW,
                       void* stream);
/* (3') the gate on its own, gate [B][C] = sigmoid(b2[c] + w2 [C][SQ] . hidden[b]): the fused inference block hands it to the project conv
 * (srbh_pwconv_fwd_epi) instead of rewriting the expanded tensor. */
int srbh_se_gate(const float* hidden, const float* w2, const float* b2, float* gate, int B, int C, int SQ, void* stream);
/* MBConv middle at inference as ONE launch (efficientnet_pytorch MBConvBlock.forward through mymodels.py:242-248, eval mode):
 * y = swish(bn1(depthwise(swish(bn0(x))))) with both BatchNorms folded to (scale, shift) [C] (pre_scale / pre_shift null: the block has
 * no expand conv, x is used as it is), pooled [B][C] = per-plane mean of y.  Geometry as srbh_dwconv_fwd; _supported tells whether
 * the LDS-staged form takes the plane (else: the separate launches). */
int srbh_dwconv_eval_supported(int B, int C, int H, int W, int K, int stride, int pad_t, int pad_l, int OH, int OW);
int srbh_dwconv_eval_fwd(const float* x, const float* w, const float* pre_scale, const float* pre_shift, const float* scale,
                         const float* shift, float* y, float* pooled, int B, int C, int H, int W, int K, int stride, int pad_t,
                         int pad_l, int OH, int OW, void* stream);

/* ---- TRAINING-mode BatchNorm + activation and squeeze-and-excitation of the MBConv / U-Net decoder blocks (csrc/srbh_mbconv.hip) ----
 * Replaces, for the encoder / decoders the reference builds at mymodels.py:242-258 and runs at mymodels.py:276-287, the stock
 * F.batch_norm(training=True) + SiLU/ReLU (+ adaptive_avg_pool2d, + drop-connect multiply and skip add) launches and their autograd
 * backward.  fp32 NCHW; planes of HW = 1, 4, 16, 64 or 256 elements with B * max(64, HW) floats within LDS, or large planes
 * (srbh_bn_act_train_supported; anything else stays on the stock ops).
 *   forward :  mean / biased variance over (B, HW) in two passes, running statistics updated as nn.BatchNorm2d does (momentum, unbiased
 *              variance), y = act(gamma (x - mean) invstd + beta) [* drop[b]] [+ res],  pooled[b][c] = mean over the plane of y
 *   backward:  dy_eff = dy [* gate[b][c] + dpooled[b][c] / HW] [* drop[b]];  dz = dy_eff act'(z);  dbeta = sum dz;  dgamma = sum dz xhat;
 *              dx = gamma invstd (dz - mean(dz) - xhat mean(dz xhat))   (dx may be NULL: parameter gradients only)            */
typedef struct srbh_bnact_args {
    const float* x;            /* [B][C][HW] the convolution output */
    float* y;                  /* [B][C][HW] */
    const float* gamma;        /* [C] */
    const float* beta;         /* [C] */
    float* running_mean;       /* [C], updated in place; NULL (both) = no running statistics */
    float* running_var;
    float* save_mean;          /* [C] out: batch mean */
    float* save_invstd;        /* [C] out: 1 / sqrt(biased variance + eps) */
    float* pooled;             /* optional [B][C] out: plane means of y (the squeeze of squeeze-and-excitation) */
    const float* res;          /* optional [B][C][HW]: added after the activation (MBConv skip connection) */
    const float* drop;         /* optional [B]: per-sample drop-connect factor floor(keep + u) / keep, applied before `res` */
    float momentum, eps;
    int B, C, HW;
    int act;                   /* 0 none | 1 SiLU | 2 ReLU */
    void* ws;                  /* srbh_bn_act_train_ws_bytes(B, C, HW) bytes of scratch (large planes only; may be NULL when that is 0) */
} srbh_bnact_args;
typedef struct srbh_bnact_bwd_args {
    const float* dy;           /* [B][C][HW] gradient of the output */
    const float* x;            /* the forward's x */
    const float* gamma;
    const float* beta;
    const float* save_mean;
    const float* save_invstd;
    const float* gate;         /* optional [B][C] (with dpooled): the squeeze-excite gate the forward output was scaled by */
    const float* dpooled;      /* optional [B][C]: gradient of the plane means */
    const float* drop;         /* optional [B] */
    float* dx;                 /* [B][C][HW] or NULL */
    float* dgamma;             /* [C] */
    float* dbeta;              /* [C] */
    int B, C, HW;
    int act;
    void* ws;                  /* as in the forward */
} srbh_bnact_bwd_args;
/* 0 = not taken; 1 = small planes (one launch each way); 2 = large planes, HW a multiple of 4 from 256 up (two launches each way:
 * per-(channel, image subset) partial sums in double in fixed slots of `ws`, then one workgroup per plane) */
int srbh_bn_act_train_supported(int B, int C, int HW);
size_t srbh_bn_act_train_ws_bytes(int B, int C, int HW);
int srbh_bn_act_train_fwd(const srbh_bnact_args* a, void* stream);
int srbh_bn_act_train_bwd(const srbh_bnact_bwd_args* a, void* stream);
/* squeeze-and-excitation in training (efficientnet_pytorch MBConvBlock.forward: x_squeezed = avg_pool(x); se_expand(swish(se_reduce(.)));
 * x = sigmoid(.) * x): forward = hidden_pre [B][SQ] = b1 + w1 [SQ][C] . pooled, hidden = swish(hidden_pre), gate [B][C] =
 * sigmoid(b2 + w2 [C][SQ] . hidden), y[plane (b, c)] *= gate in place (two launches); the three extra outputs are what the backward reads.
 * backward (three launches; ws: srbh_se_train_bwd_ws_floats): from dout = gradient of the scaled output and the BatchNorm's saved x /
 * statistics (y = act(bn(x)) is recomputed, it was scaled in place) -> dpooled [B][C] (feed it with `gate` to srbh_bn_act_train_bwd) and the
 * gradients of the four squeeze-excite parameters, batch sums in a fixed order. */
int srbh_se_train_fwd(float* y, const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2, float* hidden,
                      float* hidden_pre, float* gate, int B, int C, int SQ, int HW, void* stream);
int srbh_se_train_bwd(const float* dout, const float* x, const float* gamma, const float* beta, const float* save_mean,
                      const float* save_invstd, const float* pooled, const float* hidden, const float* hidden_pre, const float* gate,
                      const float* w1, const float* w2, float* ws, float* dpooled, float* dw1, float* db1, float* dw2, float* db2, int B,
                      int C, int SQ, int HW, int act, void* stream);
size_t srbh_se_train_bwd_ws_floats(int B, int C, int SQ);

/* U-Net decoder block entry (smp DecoderBlock.forward, reached through mymodels.py:279,287): out [B][Cx+Cs][2H][2W] =
 * cat(nearest-x2(x [B][Cx][H][W]), skip [B][Cs][2H][2W]) in one launch (skip may be NULL with Cs = 0); backward: dx = 2x2 sums of the first
 * Cx channels of dout, dskip = the remaining channels as a contiguous tensor (either output may be NULL).  fp32 NCHW, W even. */
int srbh_up2_cat_fwd(const float* x, const float* skip, float* out, int B, int Cx, int Cs, int H, int W, void* stream);
int srbh_up2_cat_bwd(const float* dout, float* dx, float* dskip, int B, int Cx, int Cs, int H, int W, void* stream);

/* ---- 1x1 convolutions of the MBConv blocks (expand / project: no bias, stride 1, no padding), fp32 NCHW (csrc/srbh_pwconv.hip) ----
 * What F.conv2d and its autograd do for efficientnet_pytorch's _expand_conv / _project_conv (reached through mymodels.py:276), as ONE launch
 * per product on the fp32 matrix unit (true fp32, fixed summation order):  x [B][Cin][HW], w [Cout][Cin], y [B][Cout][HW].
 * The weight gradient splits over images when its tile grid is small: partials in `ws` (srbh_pwconv_bwd_weight_ws_floats floats, 0 = not
 * needed) + one ordered reduce.  srbh_pwconv_supported: HW == 4 or a multiple of 16. */
int srbh_pwconv_supported(int B, int Cin, int Cout, int HW);
int srbh_pwconv_fwd(const float* x, const float* w, float* y, int B, int Cin, int Cout, int HW, void* stream);
/* the same forward reading W^T [Cin][Cout] (coalesced along Cout, as the input gradient reads W): 2x faster on the deep products.
 * srbh_transpose_many makes the transposed copies of a whole table of matrices (device-resident descriptors) in one launch. */
typedef struct srbh_transpose_desc {
    const float* src;          /* [rows][cols] */
    float* dst;                /* [cols][rows] */
    int rows, cols;
} srbh_transpose_desc;
int srbh_transpose_many(const srbh_transpose_desc* table_dev, int n, void* stream);
int srbh_pwconv_fwd_wt(const float* x, const float* wt, float* y, int B, int Cin, int Cout, int HW, void* stream);
/* forward with a fused prologue / epilogue (inference MBConv project conv): y = act((W . (x * gate)) * scale[co] + shift[co]) [+ res];
 * gate [B][Cin] (squeeze-excite, srbh_se_gate), scale / shift [Cout] (folded inference BatchNorm), res [B][Cout][HW] (skip connection):
 * each may be null.  w_transposed: w is W^T [Cin][Cout] as for srbh_pwconv_fwd_wt.  act 0 none | 1 SiLU | 2 ReLU. */
int srbh_pwconv_fwd_epi(const float* x, const float* w, int w_transposed, float* y, int B, int Cin, int Cout, int HW, const float* gate,
                        const float* scale, const float* shift, const float* res, int act, void* stream);
int srbh_pwconv_bwd_data(const float* dy, const float* w, float* dx, int B, int Cin, int Cout, int HW, void* stream);
/* the same with the gradient arriving over the MBConv block's skip connection (res: [B][Cin][HW], as dx) added in the store:
 * dX = W^T dY + res (efficientnet_pytorch MBConvBlock.forward's `x = x + inputs`, differentiated) */
int srbh_pwconv_bwd_data_res(const float* dy, const float* w, const float* res, float* dx, int B, int Cin, int Cout, int HW, void* stream);
size_t srbh_pwconv_bwd_weight_ws_floats(int B, int Cin, int Cout, int HW);
int srbh_pwconv_bwd_weight(const float* x, const float* dy, float* dw, float* ws, int B, int Cin, int Cout, int HW, void* stream);

/* ---- inference epilogue: quantise + integer mosaic (predict_realesanet_feature_globe.py:172-204) ------------------
 * accumulate: height [B][th][tw] fp32 (model output, C=1), build logits NHWC [B][th][tw][C] fp32, pos [B][4] int32
 *   = (xoff, yoff, xcount, ycount) already multiplied by 4 (predict...py:182); adds round(max(h,0)*10) and
 *   round(softmax*255) (both round-half-even, uint16 range) into uint32 mosaics [H][W], [C][H][W] and 1 into the
 *   weight mosaic, with atomics.  finalize: build class = argmax of the class sums (mod 2^16), height =
 *   round(sum/weight) where weight (mod 2^8) > 0 -- bit-identical to the reference's uint16/uint8 numpy arrays for any
 *   tile order or sharding (mosaics of different ranks add). */
int srbh_mosaic_accumulate(const float* height, const float* build, int C, int B, int th, int tw, const int* pos,
                           unsigned* res_height, unsigned* res_build, unsigned* res_weight, int H, int W, void* stream);
int srbh_mosaic_finalize(const unsigned* res_height, const unsigned* res_build, const unsigned* res_weight, int C, int H,
                         int W, unsigned short* height_out, unsigned char* build_out, void* stream);

/* ---- loader-side tensor math (BH_loader.py:30-61,326-329,361-392) --------------------------------
 * label_prep: height uint8 [B][H][W] -> build (int64 class = buildhir_lut[height]), height as float,
 *   weight = class_weight[build], and per 4x4 cell height_aggre = aggregate_torch(height, 0.25),
 *   weight_aggre = class_weight[buildhir_lut[(long)height_aggre]]  ([B][H/4][W/4]).
 * normalize_clamp: dst = clip((src - mins[c]) / ranges[c], lo, hi) on NCHW fp32 (clip only when clamp != 0). */
int srbh_label_prep(const unsigned char* height, int B, int H, int W, const unsigned char* buildhir_lut,
                    const float* class_weight, long long* build, float* height_f, float* weight, float* height_aggre,
                    float* weight_aggre, void* stream);
int srbh_normalize_clamp(const float* src, float* dst, int B, int C, int H, int W, const float* mins, const float* ranges,
                         float lo, float hi, int clamp, void* stream);

/* ---- loss and metric reductions ------------------------------------------------------------
 * Replace the elementwise / reduction passes of reference losses_pytorch/selfloss.py and metrics.py.  All outputs are
 * ACCUMULATED into caller-zeroed device buffers; fp64 accumulation.
 *
 * srbh_wmse_sum:  *out_sum += sum_i w_i (pred_i - target_i)^2      (selfloss.py:87-88; weight may be NULL, :75)
 * srbh_wmse_grad: grad_i = gscale[0] * 2 w_i (pred_i - target_i)   (gscale: device scalar, d loss / d sum)          */
int srbh_wmse_sum(const float* pred, const float* target, const float* weight, long n, double* out_sum, void* stream);
int srbh_wmse_grad(const float* pred, const float* target, const float* weight, long n, const float* gscale,
                   float* grad, void* stream);
/* srbh_cedice_sums: logits (B,C,HW) fp32 with element strides (b,c,p) -- NCHW and channels_last both work --,
 * labels (B*HW) int64 in [0,C), weight (B*HW) fp32 or NULL.  out4 += [ sum w*CE, sum pb*tb, sum pb, sum tb ] with
 * CE = -log softmax_y (selfloss.py:149,157), pb = softmax[1:].sum (:160-161), tb = (y > 0) (:162).
 * srbh_cedice_grad: dlogits (same strides) = g3[0]*d(sum w*CE) + g3[1]*d(sum pb*tb) + g3[2]*d(sum pb).              */
int srbh_cedice_sums(const float* logits, int B, int C, long HW, long b_stride, long c_stride, long p_stride,
                     const long long* labels, const float* weight, double* out4, void* stream);
int srbh_cedice_grad(const float* logits, int B, int C, long HW, long b_stride, long c_stride, long p_stride,
                     const long long* labels, const float* weight, const float* g3, float* dlogits, void* stream);
/* srbh_height_metric_sums: out[k*4 + {0,1,2,3}] += { sum d^2, sum |d|, sum d, count } over the pixels with cls == k,
 * d = pred - ref (metrics.py:186-200 computes rmse/mae/me per class from exactly these).
 * srbh_confusion_add: cm[label*num_class + pred] += 1 (metrics.py:71-73); *bad_flag = 1 if any value is out of range. */
int srbh_height_metric_sums(const float* pred, const float* ref, const long long* cls, long n, int num_class,
                            double* out, void* stream);
int srbh_confusion_add(const long long* pred, const long long* label, long n, int num_class,
                       unsigned long long* cm, int* bad_flag, void* stream);

/* ---- The middle of an MBConv block in ONE launch per direction (round 4; efficientnet_pytorch MBConvBlock.forward between the expand and
 * the project conv, run by smp's encoder at mymodels.py:276): BatchNorm0 (training) + SiLU -> depthwise KxK, stride 1, "same" zero padding ->
 * BatchNorm1 (training) + SiLU (+ the plane means squeeze-and-excitation pools).  fp32 NCHW, square planes 2x2 / 4x4 / 8x8, K = 3 | 5.
 * forward: reads e_pre (the expand conv's output), writes d_pre (the depthwise output, kept for the backward), y and pooled, both
 * BatchNorms' batch statistics and running statistics.  backward: dout = gradient of y * gate (the excite gate [B][C]), dpooled = gradient
 * of the plane means [B][C]; writes de_pre (may be NULL), the depthwise weight gradient [C][K][K] and the four affine gradients. */
typedef struct srbh_mbmid_args {
    const float* e_pre; const float* wdw;
    const float* gamma0; const float* beta0; float* running_mean0; float* running_var0; float* mean0; float* invstd0;
    const float* gamma1; const float* beta1; float* running_mean1; float* running_var1; float* mean1; float* invstd1;
    float* d_pre; float* y; float* pooled;
    float momentum0, eps0, momentum1, eps1;
    int B, C, H, W, K;
} srbh_mbmid_args;
typedef struct srbh_mbmid_bwd_args {
    const float* dout; const float* gate; const float* dpooled;
    const float* d_pre; const float* e_pre; const float* wdw;
    const float* gamma0; const float* beta0; const float* mean0; const float* invstd0;
    const float* gamma1; const float* beta1; const float* mean1; const float* invstd1;
    float* de_pre; float* dwdw; float* dgamma0; float* dbeta0; float* dgamma1; float* dbeta1;
    int B, C, H, W, K;
} srbh_mbmid_bwd_args;
int srbh_mbconv_mid_supported(int B, int C, int H, int W, int K, int stride);
int srbh_mbconv_mid_fwd(const srbh_mbmid_args* a, void* stream);
int srbh_mbconv_mid_bwd(const srbh_mbmid_bwd_args* a, void* stream);

/* ---- 3x3 convolutions of the two U-Net decoders (mymodels.py:245-258 builds them, :279 / :287 call them; smp UnetDecoder blocks:
 * nearest x2 -> concat skip -> [Conv3x3 (no bias) + BatchNorm + ReLU] x 2) on NCHW fp32 tensors, square planes 4x4 ... 64x64, 16-bit
 * matrix-core operands, fp32 accumulation (csrc/srbh_dconv.hip).  Weights: srbh_hpack_conv_h16(w, Cout, Cin, 3, transpose_flip, bf16, ...).
 *   srbh_dconv_fwd(bf16 = 0): y = conv3x3(x, w), fp16 operands;
 *   srbh_dconv_fwd(bf16 = 1) with the transposed + flipped pack (hpack cout := forward Cin, cin := forward Cout): dX = conv^T(dY, W);
 *   srbh_dconv_wgrad: dW = sum_pixels dY (x) shifted X, bf16 operands, deterministic (ws: srbh_dconv_wgrad_ws_floats floats). */
typedef struct srbh_dconv_pack_desc {
    const float* w;      /* OIHW fp32 (cout, cin, 3, 3) */
    void* fwd;           /* fp16 image for srbh_dconv_fwd(bf16 = 0) */
    void* bwd;           /* bf16 transposed + flipped image for the data gradient */
    int cout, cin;
} srbh_dconv_pack_desc;
/* both images of n weights in ONE launch (table in device memory); each image = srbh_hpack_h16_bytes(cout, cin, 3) bytes */
int srbh_dconv_pack_many(const srbh_dconv_pack_desc* table, int n, void* stream);
int srbh_dconv_supported(int B, int Cin, int Cout, int H, int W);
int srbh_dconv_fwd(const float* x, const void* wpack, float* y, int B, int Cin, int Cout, int H, int W, int bf16, void* stream);
/* the same conv with y = act(conv * scale[co] + shift[co]) in its store: a decoder block's inference BatchNorm (folded, srbh_bn_eval_scale_shift)
 * and ReLU (act 2; 0 = none) without a pass of their own (smp DecoderBlock's Conv2dReLU, mymodels.py:245-258 in eval mode) */
int srbh_dconv_fwd_epi(const float* x, const void* wpack, float* y, int B, int Cin, int Cout, int H, int W, int bf16, const float* scale,
                       const float* shift, int act, void* stream);
size_t srbh_dconv_wgrad_ws_floats(int B, int Cin, int Cout, int H, int W);
int srbh_dconv_wgrad(const float* x, const float* dy, float* dw, float* ws, int B, int Cin, int Cout, int H, int W, void* stream);

/* ---- Adam (train.py:170-179,254-256: torch.optim.Adam over the network's parameters + the loss log_vars) as ONE launch over every parameter
 * tensor (csrc/srbh_optim.hip).  `table` (device memory): one entry per tensor -- p, m (exp_avg), v (exp_avg_sq) updated in place from the
 * gradient g (NULL: the tensor is skipped this step), n elements, the tensor's learning rate and (L2) weight decay.  `chunks` (device memory,
 * nchunks x 2 ints): (table index, slice) pairs, slice s covering elements [s * srbh_adam_chunk(), ...) of that tensor.  Arithmetic of
 * torch/optim/adam.py (amsgrad = maximize = False): g += wd * p; m = lerp(m, g, 1 - beta1); v = beta2 v + (1 - beta2) g^2;
 * p -= lr / bias_correction1 * m / (sqrt(v) / sqrt(bias_correction2) + eps), the bias corrections 1 - beta^t per tensor in the table. */
typedef struct srbh_adam_entry {
    float* p; const float* g; float* m; float* v;
    long n;
    float lr, wd;
    float inv_bc1, inv_sqrt_bc2;     /* 1 / (1 - beta1^t), 1 / sqrt(1 - beta2^t) with t = THIS tensor's step count (torch counts steps per parameter) */
} srbh_adam_entry;
int srbh_adam_chunk(void);
int srbh_adam_step(const srbh_adam_entry* table, const int* chunks, int nchunks, double beta1, double beta2, double eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SRBH_H */
