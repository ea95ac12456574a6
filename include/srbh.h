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
} srbh_conv3x3_args;

int srbh_conv3x3_f16(const srbh_conv3x3_args* a, void* stream);

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
 * (want_forward=1, lrelu(conv_hr) then conv_last -- :221-222).
 * ws: srbh_rrdbnet_workspace_bytes() bytes that were ZERO when first handed to this (B,H,W) geometry
 * (the kernels never write the zero borders, so the same workspace can be reused across calls). */
int srbh_rrdbnet_forward(const srbh_rrdbnet_desc* d, const float* x, float* out, int B, int H, int W,
                         int want_forward, void* ws, size_t ws_bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SRBH_H */
