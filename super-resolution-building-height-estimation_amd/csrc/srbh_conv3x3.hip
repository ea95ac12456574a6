// srbh_conv3x3.hip -- fused 3x3 convolution for gfx950 (MI355X), fp16 operands / fp32 accumulate.
//
// Stands in for every nn.Conv2d(.., 3, 1, 1) of the RRDBNet trunk and upsampler
// (reference SR/rrdbnet_arch.py:125-143,197-204,232-238) with the surrounding elementwise ops
// (bias, LeakyReLU 0.2, x5*0.2+x, out*0.2+x, feat+body_feat, nearest-x2 read) fused in.
//
// Formulation: implicit GEMM on the matrix cores, D[oc][px] = sum_k W[oc][k] * X[k][px], with
//   A (32 rows)  = 32 output channels x 16 input channels of one filter tap  (WPACK16 fragment, 1 KiB)
//   B (32 cols)  = 32 consecutive output pixels of one row x the same 16 input channels
//   v_mfma_f32_32x32x16_f16, K loop = (input chunk of 32 ch) x (9 taps) x (2 k-steps)
// The im2col is never materialised: the input tile (+1 pixel halo, zero border kept in HBM by the
// ACT16 layout) is staged once per 32-channel chunk into LDS by LDS-DMA (global_load_lds_dwordx4)
// and every tap is just a different LDS address of the same tile.
//
// Workgroup = 256 threads = 4 waves (one per SIMD) -> 8 rows x 64 cols of output, all `cout` channels.
//   wave (wr, wc) owns rows wr*4..+3, cols wc*32..+31: 4 (x cout/32) accumulators of 32x32.
//   per (k-step, dx): 6 pixel fragments (rows -1..4) + 3 weight fragments feed 12 (x cout/32) MFMAs.
// LDS: 2 stages x (input tile 10x66x64 B + weight chunk 18 KiB x cout/32), double buffered, one barrier per chunk.
// LDS bank conflicts: pixel records are 64 B, so the 16-B k-slot inside a record is XOR-swizzled with
// (pixel_col>>2)&3; the swizzle is applied on the *source* address of the LDS-DMA (LDS-DMA writes are
// lane-linear) and on the ds_read_b128 address -- same involution on both sides.
#include "srbh_internal.h"

#include "srbh_conv3x3_kernel.h"

namespace {
using namespace srbh;
using namespace srbh_k;

template <int CB, int UPS, int BF = 0>
int launch(const KParams& p, hipStream_t stream) {
    constexpr int LDS_B = lds_bytes<CB, UPS>();
    SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)conv3x3_f16_kernel<CB, UPS, 0, BF>,
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B)));
    hipLaunchKernelGGL((conv3x3_f16_kernel<CB, UPS, 0, BF>), dim3(p.nblocks), dim3(256), LDS_B, stream, p);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

}  // namespace

static int conv3x3_impl(const srbh_conv3x3_args* a, const int bf16, const void* mask16, const int mask_chunks_total, const int mask_chunk0,
                        void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const bool ext = bf16 || mask16;
    SRBH_REQUIRE(a != nullptr, "srbh_conv3x3_f16: null args");
    SRBH_REQUIRE(a->in && a->w, "srbh_conv3x3_f16: null input/weight pointer");
    SRBH_REQUIRE(a->cout == 32 || a->cout == 64, "srbh_conv3x3_f16: cout must be 32 or 64 (got %d)", a->cout);
    SRBH_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0, "srbh_conv3x3_f16: bad geometry B=%d H=%d W=%d", a->B, a->H, a->W);
    SRBH_REQUIRE(a->in_chunks >= 1 && a->in_chunk0 >= 0 && a->in_chunk0 + a->in_chunks <= a->in_chunks_total,
                 "srbh_conv3x3_f16: input chunk range [%d,+%d) outside %d planes", a->in_chunk0, a->in_chunks,
                 a->in_chunks_total);
    SRBH_REQUIRE(!a->upsample2x || (a->H % 2 == 0 && a->W % 2 == 0), "srbh_conv3x3_f16: upsample2x needs even H,W");
    SRBH_REQUIRE(a->out16 || a->out32, "srbh_conv3x3_f16: no output requested");
    SRBH_REQUIRE(!a->out16 || (a->out16_chunk0 >= 0 && a->out16_chunk0 + a->cout / 32 <= a->out16_chunks_total),
                 "srbh_conv3x3_f16: output chunk range outside buffer");
    SRBH_REQUIRE(!a->out32 || (a->out32_c >= 1 && a->out32_c <= a->cout), "srbh_conv3x3_f16: out32_c=%d invalid",
                 a->out32_c);
    SRBH_REQUIRE(!(a->res2 && !a->res1), "srbh_conv3x3_f16: res2 requires res1");
    SRBH_REQUIRE(!(a->res1 || a->res2 || a->skip) || a->cout == 64,
                 "srbh_conv3x3_f16: residual/skip epilogues need cout == 64");

    if (!ext) {   // many-tile 64 -> 64 convs (the up-sampler tail) have a persistent form, see srbh_ptail.hip
        int used = 0;
        const int rc = ptail_run(a, stream, &used);
        if (rc != SRBH_OK || used) return rc;
    }
    SRBH_REQUIRE(!a->out16_nhwc, "srbh_conv3x3_f16: out16_nhwc (dense fp16 NHWC output) exists for 64 -> 64 convs without residual / skip "
                                 "epilogue and without an fp32 output only (the persistent tail kernel)");
    SRBH_REQUIRE(!ext || !a->upsample2x, "srbh_conv3x3_x16: the gradient forms have no nearest-x2 read");
    SRBH_REQUIRE(!mask16 || (mask_chunk0 >= 0 && mask_chunk0 + a->cout / 32 <= mask_chunks_total), "srbh_conv3x3_x16: mask chunk range outside its buffer");

    const int inH = a->upsample2x ? a->H / 2 : a->H, inW = a->upsample2x ? a->W / 2 : a->W;
    const Act16Geo gi = act16_geo(a->B, a->in_chunks_total, inH, inW);
    KParams p;
    p.in = (const char*)a->in + (long)a->in_chunk0 * gi.plane_b;
    p.in_img_b = gi.img_b;
    p.in_plane_b = gi.plane_b;
    p.in_row_b = gi.row_b;
    p.nchunk = a->in_chunks;
    p.w = (const char*)a->w;
    p.bias = a->bias;
    p.H = a->H;
    p.W = a->W;
    p.tiles_x = (a->W + TILE_W - 1) / TILE_W;
    p.tiles_per_img = p.tiles_x * ((a->H + TILE_H - 1) / TILE_H);
    p.nblocks = p.tiles_per_img * a->B;
    p.lrelu = a->lrelu;
    p.res_scale = a->res_scale;
    p.res2_scale = a->res2_scale;
    p.res1 = a->res1;
    p.res2 = a->res2;
    p.skip = a->skip;
    p.res1_update = a->res1_update;
    p.res2_update = a->res2_update;
    p.out16 = nullptr;
    p.out16_img_b = 0;
    p.out16_plane_b = 0;
    p.out16_row_b = 0;
    if (a->out16) {
        const Act16Geo go = act16_geo(a->B, a->out16_chunks_total, a->H, a->W);
        p.out16 = (char*)a->out16 + (long)a->out16_chunk0 * go.plane_b;
        p.out16_img_b = go.img_b;
        p.out16_plane_b = go.plane_b;
        p.out16_row_b = go.row_b;
    }
    p.out32 = a->out32;
    p.out32_c = a->out32_c;
    p.prof = nullptr;
    p.mask16 = nullptr; p.mask_img_b = 0; p.mask_plane_b = 0; p.mask_row_b = 0;
    if (mask16) {
        const Act16Geo gm = act16_geo(a->B, mask_chunks_total, a->H, a->W);
        p.mask16 = (const char*)mask16 + (long)mask_chunk0 * gm.plane_b;
        p.mask_img_b = gm.img_b; p.mask_plane_b = gm.plane_b; p.mask_row_b = gm.row_b;
    }
    if (bf16) return a->cout == 64 ? launch<2, 0, 1>(p, stream) : launch<1, 0, 1>(p, stream);

    if (a->upsample2x) return a->cout == 64 ? launch<2, 1>(p, stream) : launch<1, 1>(p, stream);
    return a->cout == 64 ? launch<2, 0>(p, stream) : launch<1, 0>(p, stream);
}

extern "C" int srbh_conv3x3_f16(const srbh_conv3x3_args* a, void* stream) { return conv3x3_impl(a, 0, nullptr, 0, 0, stream); }

/* The same convolution for the GRADIENT side of the RRDBNet training path (SR/rrdbnet_arch.py:538-592 differentiates the generator):
 * bf16 != 0: the ACT16 input / output planes and the WPACK16 weights hold bf16 (srbh_pack_conv3x3_b16), products on
 * v_mfma_f32_32x32x16_bf16; mask16 != NULL: the output is multiplied by the LeakyReLU derivative taken from the SAVED fp16
 * activation plane(s) mask16[chunk mask_chunk0 ..] (post-activation > 0 ? 1 : 0.2), before the 16-bit / fp32 stores. */
extern "C" int srbh_conv3x3_x16(const srbh_conv3x3_args* a, int bf16, const void* mask16, int mask_chunks_total, int mask_chunk0,
                                void* stream) {
    return conv3x3_impl(a, bf16, mask16, mask_chunks_total, mask_chunk0, stream);
}
