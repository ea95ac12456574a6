// srbh_conv3x3_kernel.h -- device code of the fused 3x3 convolution (see srbh_conv3x3.hip for the design notes).
#pragma once
#include <type_traits>
#include "srbh_internal.h"

#ifndef SRBH_PIN_ACC
#define SRBH_PIN_ACC 0
#endif
#ifndef SRBH_SCHED_HINTS
#define SRBH_SCHED_HINTS 1
#endif

namespace srbh_k {
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
    static constexpr int IN_B = NJ * 256 * 16;              // padded: the tail units land in the pad
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
    unsigned long long* prof;   // bench-only (ABL==9): per-workgroup s_memtime stamps
    // gradient convs of the RRDBNet training path (round 3, srbh_conv3x3_x16): LeakyReLU backward folded into the epilogue --
    // out *= (saved activation > 0 ? 1 : 0.2), the saved activation being the fp16 ACT16 plane(s) the forward wrote
    const char* mask16;
    long mask_img_b;
    int mask_plane_b;
    int mask_row_b;
};

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// dynamic LDS of one workgroup: two pipeline stages, or the epilogue transpose slices, whichever is larger
template <int CB, int UPS>
constexpr int lds_bytes() {
    constexpr int stages = 2 * (TileGeo<UPS>::IN_B + 18 * 1024 * CB);
    constexpr int epi = 4 * 4 * 32 * (32 * CB * 4 + 16);
    return stages > epi ? stages : epi;
}

// XCD-aware bijective remap: hardware places block b on XCD b%8 (speed only, never correctness);
// give every XCD a contiguous range of tiles so the row-blocks of one image share an L2.
__device__ __forceinline__ int xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = bid & 7, j = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

// ABL (bench-only ablations, 0 in the library): 1 = no MFMA/ds_read, 2 = no LDS-DMA after the first chunk,
// 3 = no epilogue stores, 4 = MFMA only (no ds_read), 9 = s_memtime stamps.
// PERSIST = 1 (persistent trunk kernel): activations are read with sc1 LDS-DMA (L1 bypass) and written with
// write-through (sc1) stores so that a neighbouring workgroup can consume them inside the same launch.
// BF = 1: bf16 operands (gradients: fp32's exponent range, no loss scaling) -- the staged 16-bit records and the packed weights are
// bf16, the products run on v_mfma_f32_32x32x16_bf16 and a 16-bit output is rounded to bf16 (RNE); everything else is identical.
template <int CB, int UPS, int ABL, int PERSIST, int BF = 0>
__device__ __forceinline__ void conv_tile(const KParams& p, char* smem, const int img, const int Y0, const int X0,
                                          unsigned long long* tstamp) {
    using G = TileGeo<UPS>;
    constexpr int W_B = 18 * 1024 * CB;  // weight bytes per input chunk
    constexpr int STAGE_B = G::IN_B + W_B;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;

    // ---- staging set-up: unit u of the LDS tile <- 16 bytes of the padded source plane
    const char* src0 = p.in + (long)img * p.in_img_b + (long)(UPS ? (Y0 >> 1) : Y0) * p.in_row_b +
                       (UPS ? (X0 >> 1) : X0) * PIX_B;
    int goff[G::NJ];
#pragma unroll
    for (int j = 0; j < G::NJ; ++j) {
        const int u0 = j * 256 + tid;
        const int u = u0 < G::UNITS ? u0 : 0;               // tail units re-read unit 0 into the LDS pad
        const int trow = u / (G::COLS * 4);
        const int rem = u - trow * (G::COLS * 4);
        const int pc = rem >> 2, ps = rem & 3;
        goff[j] = trow * p.in_row_b + pc * PIX_B + ((ps ^ ((pc >> 2) & 3)) << 4);
    }
    const char* wsrc = p.w + lane * 16;

    // weights: 18*CB fragments of 1 KiB per chunk, fragment f is copied by wave f%4
    constexpr int WFR = (18 * CB + 3) / 4;       // fragments per wave per chunk
    constexpr int WPP = (WFR + 5) / 6;           // ... per pipeline part
    constexpr int JPP = (G::NJ + 5) / 6;         // input LDS-DMA instructions per pipeline part

    // one sixth of the LDS-DMA traffic of a chunk; the six parts are interleaved with the six MFMA groups
    auto stage_part = [&](int chunk, int buf, int part) {
        char* dst = smem + buf * STAGE_B;
        const char* s = src0 + (long)chunk * p.in_plane_b;
#pragma unroll
        for (int jj = 0; jj < JPP; ++jj) {
            const int j = part * JPP + jj;
            if (j < G::NJ)
                __builtin_amdgcn_global_load_lds(GPTR(s + goff[j < G::NJ ? j : 0]),
                                                 LPTR(dst + (j * 256 + wave * 64) * 16), 16, 0, PERSIST ? 16 : 0);
        }
        const char* ws = wsrc + (long)chunk * W_B;
        char* wdst = dst + G::IN_B;
#pragma unroll
        for (int kk = 0; kk < WPP; ++kk) {
            const int f = wave + 4 * (part * WPP + kk);
            if (part * WPP + kk < WFR && f < 18 * CB)
                __builtin_amdgcn_global_load_lds(GPTR(ws + f * 1024), LPTR(wdst + f * 1024), 16, 0, 0);
        }
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

    // fragment registers, double buffered over the six (k-step, dx) groups of a chunk
    half8 P[2][G::NP];
    half8 A[2][3][CB];
    auto load_group = [&](const char* sb, int g, int set) {
        const int ks = g / 3, dx = g - ks * 3;
#pragma unroll
        for (int r = 0; r < G::NP; ++r) {
            if (ABL == 4) {
                P[set][r] = half8{(_Float16)lane, 1, 2, 3, 4, 5, 6, (_Float16)(r + dx)};
                asm volatile("" : "+v"(P[set][r]));
            } else {
                P[set][r] = *(const half8*)(sb + aoff[dx][ks] + r * G::ROW_B);
            }
        }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int mb = 0; mb < CB; ++mb) {
                if (ABL == 4) {
                    A[set][dy][mb] = half8{(_Float16)lane, 1, 2, 3, 4, 5, 6, (_Float16)(mb + dy)};
                    asm volatile("" : "+v"(A[set][dy][mb]));
                } else {
                    A[set][dy][mb] = *(const half8*)(sb + woff + ((((dy * 3 + dx) * 2 + ks) * CB + mb) << 10));
                }
            }
    };

    constexpr int NREAD = G::NP + 3 * CB;   // ds_read_b128 per group
    constexpr int NMFMA = 12 * CB;          // MFMAs per group
    // ONE code path for every chunk: per-variant copies of this body made the compiler shuffle all accumulators at the
    // join.  The LDS-DMA slices of chunk c+1 sit behind a wave-uniform branch at the head of each MFMA group.
    auto chunk_body = [&](const bool more, int c) {
        const char* sb = smem + (c & 1) * STAGE_B;
        load_group(sb, 0, 0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            // all six DMA slices of chunk c+1 go out in the first two MFMA groups: a chunk cannot end before its LAST
            // slice has landed (issue time + L2 latency), spreading them over all six groups made every chunk latency-bound
            if (more && ABL != 2 && g < 2) {
                stage_part(c + 1, (c + 1) & 1, 3 * g);
                stage_part(c + 1, (c + 1) & 1, 3 * g + 1);
                stage_part(c + 1, (c + 1) & 1, 3 * g + 2);
            }
            if (g + 1 < 6) load_group(sb, g + 1, (g + 1) & 1);   // next group's LDS reads fly under this group's MFMAs
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int pr = UPS ? (((i + dy - 1) >> 1) + 1) : (i + dy);
#pragma unroll
                    for (int mb = 0; mb < CB; ++mb)
                        if constexpr (BF)
                            acc[mb][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, A[g & 1][dy][mb]),
                                                                                 __builtin_bit_cast(bf16x8, P[g & 1][pr]), acc[mb][i], 0, 0, 0);
                        else
                            acc[mb][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g & 1][dy][mb], P[g & 1][pr], acc[mb][i], 0, 0, 0);
                }
            }
#if SRBH_SCHED_HINTS
            // pin the interleave: one ds_read behind each of the first NREAD MFMAs, then the rest
            if (g == 0) __builtin_amdgcn_sched_group_barrier(0x100, NREAD, 0);
            if (g + 1 < 6) {
#pragma unroll
                for (int k = 0; k < NREAD; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - NREAD, 0);
            } else {
                __builtin_amdgcn_sched_group_barrier(0x008, NMFMA, 0);
            }
#endif
        }
    };

#pragma unroll
    for (int part = 0; part < 6; ++part) stage_part(0, 0, part);
    for (int c = 0; c < p.nchunk; ++c) {
        // explicit drain: hipcc usually emits vmcnt(0) for pending LDS-DMA before a barrier, but not in every loop shape
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // chunk c landed on every wave; buf (c+1)&1 is free
        if (ABL == 9 && c < 3) tstamp[1 + c] = __builtin_amdgcn_s_memtime();
        if (ABL == 1) {
            if (c + 1 < p.nchunk)
#pragma unroll
                for (int part = 0; part < 6; ++part) stage_part(c + 1, (c + 1) & 1, part);
            continue;
        }
#if SRBH_PIN_ACC
#pragma unroll
        for (int mb = 0; mb < CB; ++mb)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(acc[mb][i]));
#endif
        chunk_body(c + 1 < p.nchunk, c);
#if SRBH_PIN_ACC
#pragma unroll
        for (int mb = 0; mb < CB; ++mb)
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+a"(acc[mb][i]));
#endif
    }

    if (ABL == 9) tstamp[4] = __builtin_amdgcn_s_memtime();
    // ---- epilogue --------------------------------------------------------------------------------------
    if (ABL == 3) {  // keep the accumulators alive, store (practically) nothing
        float sum = 0.f;
#pragma unroll
        for (int mb = 0; mb < CB; ++mb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[mb][i][r];
        if (sum == 1.2345e30f && p.out32) p.out32[0] = sum;
        return;
    }
    // The MFMA D layout gives a lane 4 channels (8g+4hi+q) of ONE pixel (l31): storing it directly means 8-byte
    // pieces at a 64-byte stride.  Instead each wave transposes one output row (32 px x 32*CB ch, fp32) through its
    // own LDS slice so that every lane ends up with 8 consecutive channels of a pixel and consecutive lanes cover
    // consecutive bytes of the pixel record: residual read-modify-write, fp16 and fp32 stores are all whole lines.
    constexpr int NCH = 32 * CB;
    constexpr int EP_STRIDE = NCH * 4 + 16;        // +16 B pad: conflict-free ds_write_b128
    constexpr int LPP = NCH / 8;                   // lanes per pixel (8 channels each)
    constexpr int PPP = 64 / LPP;                  // pixels per pass
    constexpr int NPASS = 32 / PPP;
    __syncthreads();                               // every wave is done with the stage buffers
    if (ABL == 9) tstamp[6] = __builtin_amdgcn_s_memtime();
    char* ep0 = smem + wave * (4 * 32 * EP_STRIDE);   // 4 rows x 32 px per wave (<= 34.8 KiB, 139 KiB per workgroup)
    const int c8 = lane % LPP;                     // this lane's channel octet
    const int pxl = lane / LPP;
    floatx4 bias_lo = {0.f, 0.f, 0.f, 0.f}, bias_hi = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
        bias_lo = *(const floatx4*)(p.bias + c8 * 8);
        bias_hi = *(const floatx4*)(p.bias + c8 * 8 + 4);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {                  // all LDS writes first, then one streaming pass over the 4 rows
        char* ep = ep0 + i * (32 * EP_STRIDE);
#pragma unroll
        for (int mb = 0; mb < CB; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                floatx4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = acc[mb][i][g * 4 + q];
                *(floatx4*)(ep + l31 * EP_STRIDE + (mb * 32 + g * 8 + hi * 4) * 4) = v;
            }
    }
    if (ABL == 9) tstamp[7] = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int Y = Y0 + wr * 4 + i;             // wave-uniform
        const char* ep = ep0 + i * (32 * EP_STRIDE);
        if (Y < p.H) {
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int px = ps * PPP + pxl;
                const int X = X0 + wc * 32 + px;
                floatx4 v0 = *(const floatx4*)(ep + px * EP_STRIDE + c8 * 32) + bias_lo;
                floatx4 v1 = *(const floatx4*)(ep + px * EP_STRIDE + c8 * 32 + 16) + bias_hi;
                if (X < p.W) {
                    const long pix = ((long)img * p.H + Y) * p.W + X;
                    if (p.res1) {
                        float* r1 = p.res1 + pix * 64 + c8 * 8;
                        v0 = v0 * p.res_scale + *(const floatx4*)r1;
                        v1 = v1 * p.res_scale + *(const floatx4*)(r1 + 4);
                        if (p.res2) {
                            float* r2 = p.res2 + pix * 64 + c8 * 8;
                            v0 = v0 * p.res2_scale + *(const floatx4*)r2;
                            v1 = v1 * p.res2_scale + *(const floatx4*)(r2 + 4);
                            if (p.res2_update) {
                                *(floatx4*)r2 = v0;
                                *(floatx4*)(r2 + 4) = v1;
                            }
                        }
                        if (p.res1_update) {
                            *(floatx4*)r1 = v0;
                            *(floatx4*)(r1 + 4) = v1;
                        }
                    }
                    if (p.skip) {
                        const float* sk = p.skip + pix * 64 + c8 * 8;
                        v0 += *(const floatx4*)sk;
                        v1 += *(const floatx4*)(sk + 4);
                    }
                    if (p.lrelu) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v0[q] = v0[q] >= 0.f ? v0[q] : v0[q] * 0.2f;
                            v1[q] = v1[q] >= 0.f ? v1[q] : v1[q] * 0.2f;
                        }
                    }
                    if (p.mask16) {   // LeakyReLU backward with the SAVED post-activation (y > 0 <=> z > 0; slope 0.2 at z <= 0 as torch)
                        const half8 mv = *(const half8*)(p.mask16 + (long)img * p.mask_img_b + (long)(c8 >> 2) * p.mask_plane_b +
                                                         (long)(Y + 1) * p.mask_row_b + (X + 1) * PIX_B + (c8 & 3) * 16);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v0[q] = (float)mv[q] > 0.f ? v0[q] : v0[q] * 0.2f;
                            v1[q] = (float)mv[4 + q] > 0.f ? v1[q] : v1[q] * 0.2f;
                        }
                    }
                    if (p.out16) {
                        half8 hv;
                        if constexpr (BF) {
                            typedef unsigned uint4e __attribute__((ext_vector_type(4)));
                            const uint4e u0 = __builtin_bit_cast(uint4e, v0), u1 = __builtin_bit_cast(uint4e, v1);
                            unsigned r[8];
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                r[q] = (u0[q] + 0x7fffu + ((u0[q] >> 16) & 1u)) >> 16;
                                r[4 + q] = (u1[q] + 0x7fffu + ((u1[q] >> 16) & 1u)) >> 16;
                            }
                            const uint4e pk = {r[0] | (r[1] << 16), r[2] | (r[3] << 16), r[4] | (r[5] << 16), r[6] | (r[7] << 16)};
                            hv = __builtin_bit_cast(half8, pk);
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                hv[q] = (_Float16)v0[q];
                                hv[4 + q] = (_Float16)v1[q];
                            }
                        }
                        char* o = p.out16 + (long)img * p.out16_img_b + (long)(c8 >> 2) * p.out16_plane_b +
                                  (long)(Y + 1) * p.out16_row_b + (X + 1) * PIX_B + (c8 & 3) * 16;
                        if (PERSIST) {
                            floatx4 raw = __builtin_bit_cast(floatx4, hv);
                            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(o), "v"(raw) : "memory");
                        } else {
                            *(half8*)o = hv;
                        }
                    }
                    if (p.out32) {
                        float* o = p.out32 + pix * p.out32_c + c8 * 8;
                        if (p.out32_c == 64) {
                            *(floatx4*)o = v0;
                            *(floatx4*)(o + 4) = v1;
                        } else {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                if (c8 * 8 + q < p.out32_c) o[q] = v0[q];
                                if (c8 * 8 + 4 + q < p.out32_c) o[4 + q] = v1[q];
                            }
                        }
                    }
                }
            }
        }
    }
}

template <int CB, int UPS, int ABL = 0, int BF = 0>
__global__ __launch_bounds__(256, 1) void conv3x3_f16_kernel(const KParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned long long tstamp[8];
    if (ABL == 9) tstamp[0] = __builtin_amdgcn_s_memtime();
    const int t = xcd_remap(blockIdx.x, p.nblocks);
    const int img = t / p.tiles_per_img;
    const int trem = t - img * p.tiles_per_img;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    conv_tile<CB, UPS, ABL, 0, BF>(p, smem, img, ty * TILE_H, tx * TILE_W, tstamp);
    if (ABL == 9 && p.prof) {
        tstamp[5] = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0)
            for (int k = 0; k < 8; ++k) p.prof[blockIdx.x * 8 + k] = tstamp[k];
    }
}

}  // namespace srbh_k
