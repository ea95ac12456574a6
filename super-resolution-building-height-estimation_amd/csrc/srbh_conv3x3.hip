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

namespace {
using namespace srbh;

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define GPTR(p) ((const __attribute__((address_space(1))) void*)(p))
#define LPTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int UPS>
struct TileGeo {
    static constexpr int ROWS = UPS ? (TILE_H / 2 + 2) : (TILE_H + 2);
    static constexpr int COLS = UPS ? (TILE_W / 2 + 2) : (TILE_W + 2);
    static constexpr int ROW_B = COLS * PIX_B;
    static constexpr int UNITS = ROWS * COLS * 4;           // 16-byte units in the tile
    static constexpr int NJ = (UNITS + 255) / 256;          // LDS-DMA instructions per thread per chunk
    static constexpr int IN_B = UNITS * 16;
    static constexpr int NP = UPS ? 4 : 6;                  // distinct pixel-fragment rows per wave
};

struct KParams {
    const char* in;
    long in_img_b;
    int in_plane_b;
    int in_row_b;
    int nchunk;
    const char* w;
    const float* bias;
    int H, W;
    int tiles_x, tiles_per_img, nblocks;
    int lrelu;
    float res_scale, res2_scale;
    float* res1;
    float* res2;
    const float* skip;
    int res1_update, res2_update;
    char* out16;
    long out16_img_b;
    int out16_plane_b;
    int out16_row_b;
    float* out32;
    int out32_c;
};

// XCD-aware bijective remap: hardware places block b on XCD b%8 (speed only, never correctness);
// give every XCD a contiguous range of tiles so the row-blocks of one image share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = bid & 7, j = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

template <int CB, int UPS>
__global__ __launch_bounds__(256, 1) void conv3x3_f16_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = TileGeo<UPS>;
    constexpr int W_B = 18 * 1024 * CB;  // weight bytes per input chunk
    constexpr int STAGE_B = G::IN_B + W_B;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;

    const int t = xcd_remap(blockIdx.x, p.nblocks);
    const int img = t / p.tiles_per_img;
    const int trem = t - img * p.tiles_per_img;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int Y0 = ty * TILE_H, X0 = tx * TILE_W;

    // ---- staging set-up: unit u of the LDS tile <- 16 bytes of the padded source plane
    const char* src0 = p.in + (long)img * p.in_img_b + (long)(UPS ? (Y0 >> 1) : Y0) * p.in_row_b +
                       (UPS ? (X0 >> 1) : X0) * PIX_B;
    int goff[G::NJ];
#pragma unroll
    for (int j = 0; j < G::NJ; ++j) {
        const int u = j * 256 + tid;
        const int trow = u / (G::COLS * 4);
        const int rem = u - trow * (G::COLS * 4);
        const int pc = rem >> 2, ps = rem & 3;
        goff[j] = trow * p.in_row_b + pc * PIX_B + ((ps ^ ((pc >> 2) & 3)) << 4);
    }
    const char* wsrc = p.w + lane * 16;

    auto stage = [&](int chunk, int buf) {
        char* dst = smem + buf * STAGE_B;
        const char* s = src0 + (long)chunk * p.in_plane_b;
#pragma unroll
        for (int j = 0; j < G::NJ; ++j) {
            if (j * 256 + 255 < G::UNITS || j * 256 + tid < G::UNITS)
                __builtin_amdgcn_global_load_lds(GPTR(s + goff[j]), LPTR(dst + (j * 256 + wave * 64) * 16), 16, 0, 0);
        }
        const char* ws = wsrc + (long)chunk * W_B;
        char* wdst = dst + G::IN_B;
        for (int f = wave; f < 18 * CB; f += 4)
            __builtin_amdgcn_global_load_lds(GPTR(ws + f * 1024), LPTR(wdst + f * 1024), 16, 0, 0);
    };

    // ---- per-lane operand addresses inside a stage
    int aoff[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int pc = UPS ? (((wc * 32 + l31 + dx - 1) >> 1) + 1) : (wc * 32 + l31 + dx);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            aoff[dx][ks] = wr * (UPS ? 2 : 4) * G::ROW_B + pc * PIX_B + (((ks * 2 + hi) ^ ((pc >> 2) & 3)) << 4);
    }
    const int woff = G::IN_B + lane * 16;

    floatx16 acc[CB][4];
#pragma unroll
    for (int mb = 0; mb < CB; ++mb)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mb][i][r] = 0.f;

    stage(0, 0);
    for (int c = 0; c < p.nchunk; ++c) {
        __syncthreads();  // chunk c landed (the compiler drains the LDS-DMA with vmcnt(0) here); buf (c+1)&1 is free
        if (c + 1 < p.nchunk) stage(c + 1, (c + 1) & 1);
        const char* sb = smem + (c & 1) * STAGE_B;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                half8 P[G::NP];
#pragma unroll
                for (int r = 0; r < G::NP; ++r) P[r] = *(const half8*)(sb + aoff[dx][ks] + r * G::ROW_B);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    half8 A[CB];
#pragma unroll
                    for (int mb = 0; mb < CB; ++mb)
                        A[mb] = *(const half8*)(sb + woff + ((((dy * 3 + dx) * 2 + ks) * CB + mb) << 10));
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int pr = UPS ? (((i + dy - 1) >> 1) + 1) : (i + dy);
#pragma unroll
                        for (int mb = 0; mb < CB; ++mb)
                            acc[mb][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[mb], P[pr], acc[mb][i], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue.  D layout: lane holds pixel X = l31, channels 8*g + 4*hi + q  (g = reg>>2, q = reg&3)
    const int X = X0 + wc * 32 + l31;
    if (X >= p.W) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int Y = Y0 + wr * 4 + i;
        if (Y >= p.H) continue;
        const long pix = ((long)img * p.H + Y) * p.W + X;
#pragma unroll
        for (int mb = 0; mb < CB; ++mb) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int oc = mb * 32 + g * 8 + hi * 4;
                floatx4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = acc[mb][i][g * 4 + q];
                if (p.bias) v += *(const floatx4*)(p.bias + oc);
                if (p.res1) {
                    float* r1 = p.res1 + pix * 64 + oc;
                    v = v * p.res_scale + *(const floatx4*)r1;
                    if (p.res2) {
                        float* r2 = p.res2 + pix * 64 + oc;
                        v = v * p.res2_scale + *(const floatx4*)r2;
                        if (p.res2_update) *(floatx4*)r2 = v;
                    }
                    if (p.res1_update) *(floatx4*)r1 = v;
                }
                if (p.skip) v += *(const floatx4*)(p.skip + pix * 64 + oc);
                if (p.lrelu) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = v[q] >= 0.f ? v[q] : v[q] * 0.2f;
                }
                if (p.out16) {
                    half4 hv;
#pragma unroll
                    for (int q = 0; q < 4; ++q) hv[q] = (_Float16)v[q];
                    char* o = p.out16 + (long)img * p.out16_img_b + (long)mb * p.out16_plane_b +
                              (long)(Y + 1) * p.out16_row_b + (X + 1) * PIX_B + (g * 8 + hi * 4) * 2;
                    *(half4*)o = hv;
                }
                if (p.out32) {
                    float* o = p.out32 + pix * p.out32_c + oc;
                    if (p.out32_c == 64) {
                        *(floatx4*)o = v;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (oc + q < p.out32_c) o[q] = v[q];
                    }
                }
            }
        }
    }
}

template <int CB, int UPS>
int launch(const KParams& p, hipStream_t stream) {
    using G = TileGeo<UPS>;
    constexpr int LDS_B = 2 * (G::IN_B + 18 * 1024 * CB);
    static bool attr_set = false;
    if (!attr_set) {
        SRBH_HIP(hipFuncSetAttribute((const void*)conv3x3_f16_kernel<CB, UPS>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B));
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_f16_kernel<CB, UPS>), dim3(p.nblocks), dim3(256), LDS_B, stream, p);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

}  // namespace

extern "C" int srbh_conv3x3_f16(const srbh_conv3x3_args* a, void* stream_) {
    hipStream_t stream = (hipStream_t)stream_;
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

    if (a->upsample2x) return a->cout == 64 ? launch<2, 1>(p, stream) : launch<1, 1>(p, stream);
    return a->cout == 64 ? launch<2, 0>(p, stream) : launch<1, 0>(p, stream);
}
