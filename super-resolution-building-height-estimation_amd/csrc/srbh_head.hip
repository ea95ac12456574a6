// srbh_head.hip -- fp32 kernels of the HR feature / fusion / regression head for gfx950 (MI355X).
//
// Stands in for the torch ops inside the reference's SR/HRfuse.py modules (Upsampler :17-44, BasicBlock :109-159,
// HRfeature :164-169, HRfuse_residual :173-190) and aggregate_utils.py:29-41.  The head runs at 256x256 with only
// 16 channels, i.e. it is HBM-bound; everything stays fp32 (the reference is fp32 and the head is trained).
//
//  * hconv_f32_kernel: KSxKS (3 or 1) convolution as implicit GEMM on the fp32 matrix cores
//    (v_mfma_f32_16x16x4_f32: exact fp32 FMA chain, 157 TF peak).  A = 16 output channels x 4 input channels of
//    one tap, B = 16 consecutive pixels of one output row x the same 4 channels.  Fused around it:
//      - input side : channel concat of two NHWC sources (torch.cat at SR/HRfuse.py:187), per-channel
//                     scale/shift + ReLU applied while staging (= the BatchNorm + ReLU that precede conv2,
//                     SR/HRfuse.py:146-148, folded into the consumer), zero padding by bounds check;
//      - output side: bias, PixelShuffle(2) folded into the store index (SR/HRfuse.py:23), per-channel
//                     sum / sum-of-squares partials for training-mode BatchNorm statistics.
//  * bn_finalize_kernel : partial sums -> batch mean/var -> scale/shift (+ running-stat update, momentum 0.1).
//  * bn_add_relu_kernel : out = relu(bn2(c2) + identity) (SR/HRfuse.py:150-157), identity optionally BN'd (downsample).
//  * aggregate_kernel   : aggregate_torch (aggregate_utils.py:29-41).
#include <stdlib.h>
#include "srbh_internal.h"

#include <type_traits>

namespace {
using namespace srbh;

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef float float2v __attribute__((ext_vector_type(2)));
// operand rounding of the 16-bit forms: 1 = fp16 (forward: activations and weights are O(1)), 2 = bf16 (data gradients:
// per-pixel gradients of a mean loss sit far below fp16's normal range, bf16 keeps fp32's exponent); both RNE
template <int OPT>
__device__ __forceinline__ short4v round4(const float (&v)[4]) {
    short4v r;
    if constexpr (OPT == 1) {
        half4 h;
#pragma unroll
        for (int j = 0; j < 4; ++j) h[j] = (_Float16)v[j];
        r = __builtin_bit_cast(short4v, h);
    } else {
        typedef unsigned uint2p __attribute__((ext_vector_type(2)));
        const uint2p pk = {bf16x2_rne(v[0], v[1]), bf16x2_rne(v[2], v[3])};        // round to nearest even
        r = __builtin_bit_cast(short4v, pk);
    }
    return r;
}

// 4 elements of a 16-bit tensor in memory (fp16 for OPT 1, bf16 for OPT 2; raw bits in a float2v) -> fp32
template <int OPT>
__device__ __forceinline__ floatx4 widen4(const float2v raw) {
    // (whole-vector bit casts: bit_cast(unsigned, raw[1]) of a single ext-vector element returned element 0 with this hipcc)
    typedef unsigned uint2q __attribute__((ext_vector_type(2)));
    const uint2q rw = __builtin_bit_cast(uint2q, raw);
    const unsigned lo = rw[0], hi = rw[1];
    if constexpr (OPT == 1) {
        const half4 h = __builtin_bit_cast(half4, raw);
        return floatx4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
    } else {
        return floatx4{__builtin_bit_cast(float, lo << 16), __builtin_bit_cast(float, lo & 0xffff0000u),
                       __builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
    }
}

constexpr int HT_W = 64;                    // output tile of one workgroup: (4 * RPW) rows x 64 columns, RPW rows per wave
constexpr int HC = 16;                      // input channels per LDS chunk
// dwords per channel plane (multiple of 32; >= rows*66 + 24 stagger slack): 10 rows (RPW 2) / 6 rows (RPW 1)
constexpr int plane_dw(int rpw) { return rpw == 2 ? 704 : 448; }
constexpr int RS = 66;                      // row stride inside a plane (3x3: 64 + 2 halo)
constexpr int NSLOT = 64;                   // stat partial slots (spreads atomic contention)

template <int RPW>
__device__ __forceinline__ int plane_base(int q) {   // bank staggering, see DESIGN.md "head conv LDS layout"
    return q * plane_dw(RPW) + (q & 1) * 16 + (q >> 2) * 8;
}
constexpr int in_dw(int rpw) { return 16 * plane_dw(rpw) + 64; }
// H16 form (fp16 operands, srbh_hconv_h16): the staged tile is pixel-major, one 32-byte record of 16 fp16 channels per
// pixel, the four 8-byte channel quads XOR-swizzled with bit 3 of the column so that the 16-pixel x 4-quad ds_read_b64
// of a B fragment (16 records at a 32-byte pitch = twice the 256-byte bank width) is conflict free for any dx shift
__device__ __forceinline__ int h16_off(int r, int col, int quad, int cols) {
    return (r * cols + col) * 32 + ((quad ^ (((col >> 3) & 1) << 1)) << 3);
}

struct HParams {
    const float* src0; const float* src1;
    int c0, c1;
    const float* pre_scale; const float* pre_shift;   // applied to src0 channels (nullptr = identity)
    int pre_relu;
    const float* w; const float* bias;
    int cout, cout_store;                              // padded (multiple of 16) and real output channels
    int B, H, W;
    int ps2;
    float* out;
    double* stats;                                     // [NSLOT][2][cout] or nullptr
    int tiles_x, tiles_per_img, ntiles, tiles_per_xcd;
    int ld0, ld1, out_ld, out_coff;                    // pixel strides (floats) of src0 / src1 / out, channel offset of out
    int post_lrelu;
    const float* res1; int res1_ld; float res1_scale;  // y = y*res1_scale + res1 ; then y = y*res2_scale + res2
    const float* res2; int res2_ld; float res2_scale;
    const float* post_scale; const float* post_shift; int post_relu;
    int io_h16;                                        // SRBH_IO_*: which of src0 / src1 / res1 / out hold fp16 elements (OPT 1 only)
    const float* bstat_c; const float* bstat_mean; const float* bstat_invstd; const float* bstat_ms; const float* bstat_mh;   // hconv16_kernel: backward-statistics epilogue
};

// NOB = cout/16 (1, 2 or 4), KS = 3 or 1, RPW = output rows per wave (2: 8-row tiles, 54 KiB of LDS, 2 workgroups per CU;
// 1: 4-row tiles, 38 KiB, ~half the registers -> 4 workgroups per CU: more tiles in flight to hide the staging latency
// of a one-tile-per-workgroup kernel, for 20 % more halo reads)
// H16 = 1 (srbh_hconv_h16): same kernel, but the staged tile is rounded to fp16 and the contraction runs on
// v_mfma_f32_16x16x16_f16 (A = 16 out-channels x the chunk's 16 in-channels of one tap, held in registers; B = 16
// pixels x the same 16 channels; fp32 accumulate; D layout identical to the fp32 form, so the epilogue is shared): at
// 1/8 of the fp32 matrix-core time the kernel is bound by its HBM traffic, which is what SURVEY 8d prescribes for the head.
// PixelShuffle(2) store staging of the 16 -> 64 up-sampler convs (NOB = 4): per wave [2 sub-rows][32 output pixels][16 channels + 1 pad] floats
#ifndef HCONV_PS2_STAGE
#define HCONV_PS2_STAGE 1        // (0: the direct scattered store, for same-box A/B builds)
#endif
constexpr int PS2_PITCH = 17, PS2_WAVE_DW = 2 * 32 * PS2_PITCH, PS2_STAGE_B = 4 * PS2_WAVE_DW * 4;

template <int NOB, int KS, int RPW, int OPT>
__global__ __launch_bounds__(256) void hconv_f32_kernel(const HParams p) {
    constexpr bool H16 = OPT != 0;
    constexpr int HT_H = 4 * RPW, IN_DW = H16 ? ((HT_H + 2) * (HT_W + 2) * 8) : in_dw(RPW), NI = 4 * RPW;
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    constexpr int TAPS = KS * KS;
    constexpr int HALO = KS / 2;
    constexpr int ROWS = HT_H + 2 * HALO, COLS = HT_W + 2 * HALO;
    constexpr int W_DW = TAPS * 4 * NOB * 64;          // weight floats per 16-channel chunk
    float* s_in = hsm;
    float* s_w = hsm + IN_DW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kk = lane >> 4;
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs (blockIdx % 8), each with its own L2.  Consecutive tiles
    // (x-neighbours, then the next tile row) share halo rows -- 55 % more input rows than a tile owns at 4-row tiles -- so every
    // XCD walks its own contiguous run of tiles and the halo re-reads hit that XCD's L2 instead of going out to the fabric.
    // (A PERSISTENT walk -- a few workgroups per CU looping over their XCD's tiles, BatchNorm statistics flushed once per workgroup,
    // optionally with the next tile's loads issued ahead of this tile's MFMAs -- was measured in round 2: the loop lets the compiler
    // keep ~80 more loop-invariant values in registers (182 VGPRs, occupancy 2: 180 us vs 160; pipelined 255 + 52 AGPRs, occupancy
    // 1: 280 us).  At EQUAL occupancy the persistent form is 23 % faster than one workgroup per tile, so it is the right shape --
    // for a kernel written around it, not for this template.  DESIGN.md 5.0b / 8.)
    const int t = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3);
    if (t >= p.ntiles) return;
    const int img = t / p.tiles_per_img;
    const int trem = t - img * p.tiles_per_img;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int Y0 = ty * HT_H, X0 = tx * HT_W;
    const int cin = p.c0 + p.c1;
    const int nchunk = (cin + HC - 1) / HC;
    const bool vec0 = (p.c0 & 3) == 0 && (p.ld0 & 3) == 0,
               vec1 = p.c1 > 0 && (p.c1 & 3) == 0 && (p.c0 & 3) == 0 && (p.ld1 & 3) == 0;

    floatx4 acc[NOB][NI];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int i = 0; i < NI; ++i) acc[ob][i] = floatx4{0.f, 0.f, 0.f, 0.f};

    for (int c = 0; c < nchunk; ++c) {
        __syncthreads();
        // ---- weights of this chunk (already in A-fragment order).  H16: 8 bytes per lane and (tap, ob), straight into registers --
        // issued BEFORE the input loads: they are L2 hits that arrive during the staging (loaded after it, the first MFMA of every
        // tile waited a full L2 round trip behind the barrier)
        short4v wa[H16 ? TAPS : 1][H16 ? NOB : 1];
        if constexpr (H16) {
            const short4v* wp = (const short4v*)p.w + (long)c * (TAPS * NOB * 64) + lane;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap)
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) wa[tap][ob] = wp[(tap * NOB + ob) * 64];
        }
        // ---- stage the input chunk: channel-major planes, transform + zero padding applied here.
        // Fast path (every 4-channel group is one aligned 16-byte load from src0 or src1): ALL global loads of the chunk
        // are issued before the first LDS store -- the one-load-per-iteration loop below exposes a full memory latency
        // per iteration (11 of them per chunk) and made the kernel staging-latency bound.
        constexpr int NIT = (ROWS * COLS * 4 + 255) / 256;
        const bool fast = vec0 && (p.c1 == 0 || vec1) && (cin & 3) == 0;
        if (fast) {
            // Address arithmetic of the walk: unit u = tid + 256*it is pixel (tid >> 2) + 64*it of the staged window and channel
            // group tid & 3 (the same for every iteration), so the thread forms ONE pointer to its channel group at the window's
            // origin and then only adds 32-bit element offsets that advance by constants (no 64-bit multiply, no division by the
            // window width per load: those quarter-rate integer ops were half of the kernel's VALU time).
            floatx4 ld[NIT];
            int loff[NIT];
            const int cg = tid & 3;
            const int ch = c * HC + cg * 4;
            const bool chok = ch < cin, in0 = ch < p.c0;
            const int ldp = in0 ? p.ld0 : p.ld1;
            // fp16 activations in memory (inference, SRBH_IO_*): element size 2, the quad is staged as it is (8-byte load, no rounding)
            const bool srch = OPT != 0 && (p.io_h16 & (in0 ? SRBH_IO_SRC0_H16 : SRBH_IO_SRC1_H16)) != 0;
            const int esz = srch ? 2 : 4;
            const char* tp = (const char*)(in0 ? p.src0 : p.src1) +
                             ((((long)img * p.H + (Y0 - HALO)) * p.W + (X0 - HALO)) * ldp + (in0 ? ch : ch - p.c0)) * esz;
            floatx4 psc = {1.f, 1.f, 1.f, 1.f}, psh = {0.f, 0.f, 0.f, 0.f};
            bool relu_t = false;                                   // (a select, not max(a, -inf): a NaN-poisoned input must stay NaN)
            if (in0 && chok) {
                if (p.pre_scale) { psc = *(const floatx4*)(p.pre_scale + ch); psh = *(const floatx4*)(p.pre_shift + ch); }
                relu_t = p.pre_relu != 0;
            }
            int col = tid >> 2, y = Y0 - HALO, x = X0 - HALO + col;
            int voff = col * ldp, lo_px = col;                     // (r*W + col) * ld ; r*RS + col
            const int vstep = 64 * ldp, vwrap = (p.W + 64 - COLS) * ldp;
            unsigned okmask = 0;                                   // loaded units (the others are the zero padding: no transform)
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                bool ok = chok && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
                bool unit = true;
                if ((it + 1) * 256 > ROWS * COLS * 4) unit = tid + it * 256 < ROWS * COLS * 4;
                // (only the load here: the transform sits in the store loop below, so that all loads are in flight before the first use)
                ld[it] = floatx4{0.f, 0.f, 0.f, 0.f};
                if (ok && unit) {
                    if (srch) {
                        const float2v h = *(const float2v*)(tp + (long)voff * 2);      // 4 fp16 channels, kept as raw bits in ld[it][0..1]
                        ld[it][0] = h[0]; ld[it][1] = h[1];
                    } else {
                        ld[it] = *(const floatx4*)(tp + (long)voff * 4);
                    }
                    okmask |= 1u << it;
                }
                if constexpr (H16) loff[it] = unit ? ((tid >> 2) + 64 * it) * 32 + ((cg ^ ((col >> 2) & 2)) << 3) : -1;   // = h16_off(r, col, cg, COLS)
                else loff[it] = unit ? lo_px : -1;
                const bool wrapped = col + 64 >= COLS;
                col += wrapped ? 64 - COLS : 64;
                x += wrapped ? 64 - COLS : 64;
                y += wrapped ? 1 : 0;
                voff += wrapped ? vwrap : vstep;
                lo_px += wrapped ? RS + 64 - COLS : 64;
            }
#pragma unroll
            for (int it = 0; it < NIT; ++it) {
                if (loff[it] >= 0) {
                    if ((okmask & (1u << it)) && !srch) {
                        ld[it] = ld[it] * psc + psh;
#pragma unroll
                        for (int j = 0; j < 4; ++j) ld[it][j] = relu_t ? fmaxf(ld[it][j], 0.f) : ld[it][j];
                    }
                    if constexpr (H16) {
                        const float t4[4] = {ld[it][0], ld[it][1], ld[it][2], ld[it][3]};
                        if (srch) *(float2v*)((char*)s_in + loff[it]) = float2v{ld[it][0], ld[it][1]};
                        else *(short4v*)((char*)s_in + loff[it]) = round4<OPT>(t4);
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j) s_in[plane_base<RPW>(cg * 4 + j) + loff[it]] = ld[it][j];
                    }
                }
            }
        } else
        for (int u = tid; u < ROWS * COLS * 4; u += 256) {
            const int cg = u & 3, pix = u >> 2;
            const int r = pix / COLS, col = pix - r * COLS;
            const int y = Y0 + r - HALO, x = X0 + col - HALO;
            const int ch = c * HC + cg * 4;                 // first of 4 channels
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (y >= 0 && y < p.H && x >= 0 && x < p.W && ch < cin) {
                const long pixi = ((long)img * p.H + y) * p.W + x;
                if (vec0 && ch + 3 < p.c0) {            // 16-byte path: the 4 channels sit in src0
                    floatx4 a = *(const floatx4*)(p.src0 + pixi * p.ld0 + ch);
                    if (p.pre_scale) a = a * *(const floatx4*)(p.pre_scale + ch) + *(const floatx4*)(p.pre_shift + ch);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = p.pre_relu ? fmaxf(a[j], 0.f) : a[j];
                } else if (vec1 && ch >= p.c0 && ch + 3 < cin) {   // ... or in src1
                    const floatx4 a = *(const floatx4*)(p.src1 + pixi * p.ld1 + (ch - p.c0));
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = a[j];
                } else
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int cc = ch + j;
                    if (cc < p.c0) {
                        float a = p.src0[pixi * p.ld0 + cc];
                        if (p.pre_scale) a = a * p.pre_scale[cc] + p.pre_shift[cc];
                        if (p.pre_relu) a = fmaxf(a, 0.f);
                        v[j] = a;
                    } else if (cc < cin) {
                        v[j] = p.src1[pixi * p.ld1 + (cc - p.c0)];
                    }
                }
            }
            if constexpr (H16) {
                *(short4v*)((char*)s_in + h16_off(r, col, cg, COLS)) = round4<OPT>(v);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) s_in[plane_base<RPW>(cg * 4 + j) + r * RS + col] = v[j];
            }
        }
        if constexpr (!H16) {
            for (int u = tid; u < W_DW / 4; u += 256)
                ((floatx4*)s_w)[u] = ((const floatx4*)(p.w + (long)c * W_DW))[u];
        }
        __syncthreads();
        if constexpr (H16) {
            // wave owns RPW rows; 4 column tiles of 16 pixels each; one MFMA per (tap, tile, ob) over the chunk's 16 channels
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int dy = tap / KS, dx = tap - dy * KS;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const short4v b = *(const short4v*)((const char*)s_in + h16_off(wave * RPW + dy + (i >> 2), dx + (i & 3) * 16 + l15, kk, COLS));
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob) {
                        if constexpr (OPT == 1)
                            acc[ob][i] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4, wa[tap][ob]), __builtin_bit_cast(half4, b), acc[ob][i], 0, 0, 0);
                        else
                            acc[ob][i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wa[tap][ob], b, acc[ob][i], 0, 0, 0);
                    }
                }
            }
            continue;
        }
        // ---- MFMA: wave owns RPW rows; 4 column tiles of 16 px each
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            const int dy = tap / KS, dx = tap - dy * KS;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                float a[NOB];
#pragma unroll
                for (int ob = 0; ob < NOB; ++ob) a[ob] = s_w[((tap * 4 + s) * NOB + ob) * 64 + lane];
                const float* bp = s_in + plane_base<RPW>(s * 4 + kk) + (wave * RPW + dy) * RS + dx + l15;
#pragma unroll
                for (int i = 0; i < NI; ++i) {
                    const float b = bp[(i >> 2) * RS + (i & 3) * 16];
#pragma unroll
                    for (int ob = 0; ob < NOB; ++ob)
                        acc[ob][i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ob], b, acc[ob][i], 0, 0, 0);
                }
            }
        }
    }

    // ---- epilogue: lane holds channels ob*16 + kk*4 + (0..3) of pixel (row, col0 + l15)
    float ssum[NOB][4], ssq[NOB][4];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int q = 0; q < 4; ++q) ssum[ob][q] = ssq[ob][q] = 0.f;
    // per-thread constants of the epilogue hoisted out of the pixel loop: channel parameters, and ONE pointer per tensor to this lane's
    // first pixel (row Y0 + wave*RPW, column X0 + l15, channels kk*4..); the pixels of the loop are uniform element offsets from it
    floatx4 e_bias[NOB], e_sc[NOB], e_sh[NOB];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        const int oc = ob * 16 + kk * 4;
        e_bias[ob] = p.bias ? *(const floatx4*)(p.bias + oc) : floatx4{0.f, 0.f, 0.f, 0.f};
        e_sc[ob] = p.post_scale ? *(const floatx4*)(p.post_scale + oc) : floatx4{1.f, 1.f, 1.f, 1.f};
        e_sh[ob] = p.post_scale ? *(const floatx4*)(p.post_shift + oc) : floatx4{0.f, 0.f, 0.f, 0.f};
    }
    const long pix0 = ((long)img * p.H + Y0 + wave * RPW) * p.W + X0 + l15;
    const bool out16 = OPT != 0 && (p.io_h16 & SRBH_IO_OUT_H16) != 0, res16 = OPT != 0 && (p.io_h16 & SRBH_IO_RES1_H16) != 0;
    float* const o0 = (float*)((char*)p.out + (pix0 * p.out_ld + p.out_coff + kk * 4) * (out16 ? 2 : 4));
    const float* const r1p = p.res1 ? (const float*)((const char*)p.res1 + (pix0 * p.res1_ld + kk * 4) * (res16 ? 2 : 4)) : nullptr;
    const float* const r2p = p.res2 ? p.res2 + pix0 * p.res2_ld + kk * 4 : nullptr;
    const bool vec_out = (p.cout_store & 3) == 0 && (p.out_ld & 3) == 0 && (p.out_coff & 3) == 0;
    // PixelShuffle(2) store (SR/HRfuse.py:23) of a full 16 -> 64 conv: a lane's 4 accumulators are the 2x2 sub-pixels of ONE output
    // channel, so storing them directly is 64 scattered 4-byte stores per lane and tile (this kernel ran at 0.19 of the HBM peak).
    // Instead each wave passes its 16-pixel group through an LDS slice in output order and writes it back as 16-byte quads of
    // consecutive channels: 64 lanes x 16 bytes = 1 KiB contiguous per instruction.
    const bool ps2_staged = HCONV_PS2_STAGE && NOB == 4 && p.ps2 && p.cout_store == 64 && !p.res1;
    float* const ps2_lds = hsm + wave * PS2_WAVE_DW;
    if (NOB == 4 && ps2_staged) __syncthreads();                 // every wave is past its last read of the input tile
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int Y = Y0 + wave * RPW + (i >> 2), X = X0 + (i & 3) * 16 + l15;
        const bool ok = Y < p.H && X < p.W;
        const int dpx = (i >> 2) * p.W + (i & 3) * 16;          // uniform pixel offset from pix0
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            const int oc = ob * 16 + kk * 4;
            floatx4 v = acc[ob][i];
            if (p.bias) v += e_bias[ob];
            if (p.post_scale) v = v * e_sc[ob] + e_sh[ob];   // eval-mode BatchNorm
            if (ok && p.res1) {      // residual epilogues of the strict fp32 trunk (x5*0.2 + x, out*0.2 + x)
                if (res16) {
                    const float2v r = *(const float2v*)((const char*)r1p + (long)(dpx * p.res1_ld + ob * 16) * 2);
                    v = v * p.res1_scale + widen4<OPT == 0 ? 1 : OPT>(r);
                } else {
                    v = v * p.res1_scale + *(const floatx4*)(r1p + dpx * p.res1_ld + ob * 16);
                }
                if (p.res2) v = v * p.res2_scale + *(const floatx4*)(r2p + dpx * p.res2_ld + ob * 16);
            }
            if (p.post_lrelu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = v[q] >= 0.f ? v[q] : v[q] * 0.2f;
            }
            if (p.post_relu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
            }
            if (ok) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ssum[ob][q] += v[q];
                    ssq[ob][q] += v[q] * v[q];
                }
                if (NOB == 4 && ps2_staged) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) ps2_lds[((q >> 1) * 32 + 2 * l15 + (q & 1)) * PS2_PITCH + ob * 4 + kk] = v[q];
                } else if (p.ps2) {
                    // PixelShuffle(2): out[b, oc>>2, 2Y + ((oc>>1)&1), 2X + (oc&1)] (SR/HRfuse.py:23); lane's 4 channels
                    // are the 2x2 sub-pixels of output channel oc>>2
                    const int co = p.cout_store >> 2, cq = oc >> 2;
                    if (oc < p.cout_store) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const long o = (((long)img * 2 * p.H + 2 * Y + (q >> 1)) * (2 * p.W) + 2 * X + (q & 1)) * co + cq;
                            p.out[o] = v[q];
                        }
                    }
                } else {
                    float* o = o0 + dpx * p.out_ld + ob * 16;
                    if (out16) {          // (host: 4-aligned channels) one rounding here, none in the consumer
                        if (oc < p.cout_store) {
                            const float t4[4] = {v[0], v[1], v[2], v[3]};
                            *(short4v*)((char*)o0 + (long)(dpx * p.out_ld + ob * 16) * 2) = round4<OPT == 0 ? 1 : OPT>(t4);
                        }
                    } else if (vec_out) {
                        if (oc < p.cout_store) *(floatx4*)o = v;
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (oc + q < p.cout_store) o[q] = v[q];
                    }
                }
            }
        }
        if (NOB == 4 && ps2_staged) {
            // the wave's own slice: LDS operations of one wave complete in order, the waitcnt + memory clobber keep the compiler
            // from moving the reads above the writes
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int Xg = X0 + (i & 3) * 16;
            if (Y < p.H) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = lane + 64 * r, dy = c >> 7, px2 = (c >> 2) & 31, c4 = (c & 3) * 4;
                    const float* s = ps2_lds + (dy * 32 + px2) * PS2_PITCH + c4;
                    floatx4 t;
                    t[0] = s[0]; t[1] = s[1]; t[2] = s[2]; t[3] = s[3];
                    if (Xg + (px2 >> 1) < p.W) {
                        const long o = (((long)img * 2 * p.H + 2 * Y + dy) * (2 * p.W) + 2 * Xg + px2) * 16 + c4;
                        if (out16) {      // fp16 NHWC output (inference chain, round 4): the one rounding the consumer's staging would do
                            const float t4[4] = {t[0], t[1], t[2], t[3]};
                            *(short4v*)((short*)p.out + o) = round4<OPT == 0 ? 1 : OPT>(t4);
                        } else {
                            *(floatx4*)(p.out + o) = t;
                        }
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // reads done before the next group's writes
        }
    }
    if (p.stats) {
        // reduce over the 16 pixel lanes, then over the 4 waves through LDS (the input tile is dead by now), then ONE double
        // atomic per (workgroup, channel, moment) into a slot: 4x fewer atomics than one per wave -- with 4-row tiles there
        // are twice as many workgroups, and the atomics had grown to 60 us of a 335 us conv
        __syncthreads();                       // every wave is past its last LDS read of the tile
        float* red = hsm;                      // [4 waves][2 moments][NOB * 16 channels]
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float a = ssum[ob][q], b = ssq[ob][q];
#pragma unroll
                for (int m = 1; m < 16; m <<= 1) {
                    a += __shfl_xor(a, m);
                    b += __shfl_xor(b, m);
                }
                if (l15 == 0) {
                    const int oc = ob * 16 + kk * 4 + q;
                    red[(wave * 2 + 0) * (NOB * 16) + oc] = a;
                    red[(wave * 2 + 1) * (NOB * 16) + oc] = b;
                }
            }
        __syncthreads();
        if (tid < 2 * NOB * 16) {
            const int mom = tid / (NOB * 16), oc = tid - mom * (NOB * 16);
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += (double)red[(w * 2 + mom) * (NOB * 16) + oc];
            double* slot = p.stats + (long)(blockIdx.x % NSLOT) * 2 * p.cout;
            atomicAdd(slot + mom * p.cout + oc, v);
        }
    }
}

// ---- weights: OIHW fp32 -> [chunk][tap][s][ob][lane 64]  (A fragment of v_mfma_f32_16x16x4_f32: row = lane&15, k = lane>>4)
// transpose_flip: pack the weight of the corresponding data-gradient conv (swap I/O, flip taps) instead
__global__ void hpack_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int ks, int nchunk,
                             int nob, int transpose_flip) {
    const int taps = ks * ks;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)nchunk * taps * 4 * nob * 64;
    if (idx >= total) return;
    int lane = idx & 63;
    long f = idx >> 6;
    int ob = f % nob; f /= nob;
    int s = f & 3; f >>= 2;
    int tap = f % taps;
    int chunk = f / taps;
    int oc = ob * 16 + (lane & 15);
    int ic = chunk * 16 + s * 4 + (lane >> 4);
    float v = 0.f;
    if (!transpose_flip) {
        if (oc < cout && ic < cin) v = w[((long)oc * cin + ic) * taps + tap];
    } else {
        // logical conv: out channel `oc` ranges over the ORIGINAL input channels, `ic` over the original outputs
        // here cout/cin are the logical (already swapped) counts; original tensor is [cin][cout][ks][ks]
        if (oc < cout && ic < cin) v = w[((long)ic * cout + oc) * taps + (taps - 1 - tap)];
    }
    out[idx] = v;
}

// fp16 form: [chunk][tap][ob][lane 64][4]  (A fragment of v_mfma_f32_16x16x16_f16: row = lane&15, k = 4*(lane>>4) + 0..3)
__global__ void hpack_h16_kernel(const float* __restrict__ w, short* __restrict__ out, int cout, int cin, int ks, int nchunk,
                                 int nob, int transpose_flip, int bf16) {
    const int taps = ks * ks;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)nchunk * taps * nob * 64 * 4;
    if (idx >= total) return;
    int j = idx & 3;
    int lane = (idx >> 2) & 63;
    long f = idx >> 8;
    int ob = f % nob; f /= nob;
    int tap = f % taps;
    int chunk = f / taps;
    int oc = ob * 16 + (lane & 15);
    int ic = chunk * 16 + (lane >> 4) * 4 + j;
    float v = 0.f;
    if (oc < cout && ic < cin)
        v = transpose_flip ? w[((long)ic * cout + oc) * taps + (taps - 1 - tap)] : w[((long)oc * cin + ic) * taps + tap];
    if (bf16) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        out[idx] = (short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    } else {
        out[idx] = __builtin_bit_cast(short, (_Float16)v);
    }
}

// ---- BatchNorm: partial sums -> scale/shift (+ running stats) ------------------------------------------------------
// clear != 0 (one block of 256 threads, C <= 64): the slots a thread summed are zeroed behind the read (self-cleaning buffer)
// (round 5: the NSLOT x 2 C partial sums are folded by the whole block -- group g of 256 / (2 C) takes the slots g, g + G, ... of one (moment,
//  channel) value, all of a thread's loads in flight at once, then a fixed-order fold through LDS -- instead of C threads walking 128
//  dependent-latency loads each: these one-block kernels sit between the head's chip-filling kernels 49 times per training step)
__global__ __launch_bounds__(256) void bn_finalize_kernel(double* __restrict__ stats, int C, double count, const float* gamma,
                                   const float* beta, float eps, float momentum, float* running_mean,
                                   float* running_var, float* scale, float* shift, float* save_mean, float* save_invstd, int clear) {
    __shared__ double red[256];
    double s = 0, q = 0;
    srbh::fold_stat_slots(stats, C, clear, red, s, q);
    const int c = threadIdx.x;
    if (c >= C) return;
    double mean = s / count;
    double var = q / count - mean * mean;   // biased, as used for normalisation
    if (var < 0) var = 0;
    float invstd = (float)(1.0 / sqrt(var + (double)eps));
    float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale[c] = g * invstd;
    shift[c] = b - (float)mean * g * invstd;
    if (save_mean) save_mean[c] = (float)mean;
    if (save_invstd) save_invstd[c] = invstd;
    if (running_mean) {
        double unbiased = count > 1 ? var * count / (count - 1) : var;
        running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

// eval mode: scale/shift from running statistics
__global__ void bn_eval_kernel(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                               float* scale, float* shift) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float invstd = 1.f / sqrtf(rv[c] + eps);
    float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale[c] = g * invstd;
    shift[c] = b - rm[c] * g * invstd;
}

// out = relu(a*sa + ha + (idt*si + hi))   (C multiple of 4; NHWC fp32)
// AH / IH: `a` / `idt` hold fp16 elements in memory (the saved activations of the training step, round 3)
// bits (may be NULL): the ReLU's activity pattern for the backward pass -- per 64 consecutive 4-channel groups four 64-bit words, word j
// bit l = (element j of group 64 k + l is > 0): the reduce pass of the BatchNorm backward then reads 1 bit instead of the fp32 output
template <int AH, int IH>
__global__ void bn_add_relu_kernel(const void* __restrict__ a, const float* sa, const float* ha,
                                   const void* __restrict__ idt, const float* si, const float* hi, floatx4* out,
                                   long n4, int C, unsigned long long* __restrict__ bits = nullptr) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    // (one unit per trip: four units per trip with all eight loads in front measured SLOWER, 179-183 against 166-170 us, profiles/r06q_ab_bar_unroll.txt)
    for (; i < n4; i += stride) {
        int c = (int)((i * 4) % C);
        floatx4 av, d;
        if constexpr (AH != 0) av = widen4<1>(((const float2v*)a)[i]);
        else av = ((const floatx4*)a)[i];
        if constexpr (IH != 0) d = widen4<1>(((const float2v*)idt)[i]);
        else d = ((const floatx4*)idt)[i];
        floatx4 v = av * *(const floatx4*)(sa + c) + *(const floatx4*)(ha + c);
        if (si) d = d * *(const floatx4*)(si + c) + *(const floatx4*)(hi + c);
        v += d;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
        out[i] = v;
        if (bits) {          // (grid stride and block size are multiples of 64: a wave's 64 groups are 64 k .. 64 k + 63, lane = bit)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned long long m = __ballot(v[q] > 0.f);
                if ((threadIdx.x & 63) == 0) bits[(i >> 6) * 4 + q] = m;
            }
        }
    }
}

// aggregate_torch (aggregate_utils.py:29-41): step x step block sums of data and of (data >= 0)
__global__ void aggregate_kernel(const float* __restrict__ data, float* __restrict__ out, int N, int H, int W, int step) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    int oh = H / step, ow = W / step;
    long total = (long)N * oh * ow;
    if (idx >= total) return;
    int ox = idx % ow;
    long r = idx / ow;
    int oy = r % oh;
    int n = r / oh;
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < step; ++i)
        for (int j = 0; j < step; ++j) {
            float v = data[((long)n * H + oy * step + i) * W + ox * step + j];
            s1 += v;
            s2 += (v >= 0.f) ? 1.f : 0.f;
        }
    out[idx] = s1 / (s2 + 1e-10f);
}

// NCHW <-> NHWC fp32 (module boundaries of the head)
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int H, int W) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over dst
    long total = (long)B * C * H * W;
    if (idx >= total) return;
    int c = idx % C;
    long r = idx / C;
    int x = r % W; r /= W;
    int y = r % H;
    int b = r / H;
    dst[idx] = src[(((long)b * C + c) * H + y) * W + x];
}

// F.interpolate(scale_factor=2, mode='nearest') on NHWC fp32: out[y][x] = in[y>>1][x>>1] (SR/rrdbnet_arch.py:236-237)
__global__ void nearest2x_kernel(const floatx4* __restrict__ src, floatx4* __restrict__ dst, int B, int H, int W, int C4) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over dst float4s, H,W = OUTPUT size
    long total = (long)B * H * W * C4;
    if (idx >= total) return;
    int c = idx % C4;
    long r = idx / C4;
    int x = r % W; r /= W;
    int y = r % H;
    int b = r / H;
    dst[idx] = src[(((long)b * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) * C4 + c];
}

#include "srbh_hconv16_kernel.h"
#include "srbh_hconv_entry_kernel.h"
#include "srbh_hconv_up_kernel.h"
#include "srbh_hblock16_kernel.h"

template <int NOB, int KS, int RPW, int OPT = 0>
int launch_hconv(HParams& p, int B, int H, int W, hipStream_t st) {
    constexpr bool H16 = OPT != 0;
    constexpr int LDS_T = H16 ? ((4 * RPW + 2) * (HT_W + 2) * 32 > 4 * 2 * NOB * 16 * 4 ? (4 * RPW + 2) * (HT_W + 2) * 32 : 4 * 2 * NOB * 16 * 4)
                              : (in_dw(RPW) + KS * KS * 4 * NOB * 64) * 4;
    constexpr int LDS_B = (NOB == 4 && LDS_T < PS2_STAGE_B) ? PS2_STAGE_B : LDS_T;     // (PixelShuffle store staging, see the epilogue)
    p.tiles_x = (W + HT_W - 1) / HT_W;
    p.tiles_per_img = p.tiles_x * ((H + 4 * RPW - 1) / (4 * RPW));
    p.ntiles = p.tiles_per_img * B;
    p.tiles_per_xcd = (p.ntiles + 7) / 8;
    const int nblocks = p.tiles_per_xcd * 8;
    static const int lds_floor = getenv("SRBH_HCONV_LDS") ? atoi(getenv("SRBH_HCONV_LDS")) : 0;     // developer aid: occupancy A/B
    const int lds_b = H16 && lds_floor > LDS_B ? lds_floor : LDS_B;
    if (lds_b > 65536 || LDS_B > 65536) {
        SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)hconv_f32_kernel<NOB, KS, RPW, OPT>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds_b > LDS_B ? lds_b : LDS_B)));
    }
    hipLaunchKernelGGL((hconv_f32_kernel<NOB, KS, RPW, OPT>), dim3(nblocks), dim3(256), lds_b, st, p);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

// tile height: 4-row tiles (RPW 1) by default; SRBH_HCONV_RPW=2 selects the 8-row tiles (A/B aid)
int hconv_rpw_small() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("SRBH_HCONV_RPW");
        v = (e && atoi(e) == 2) ? 2 : 1;
    }
    return v;
}

}  // namespace

extern "C" size_t srbh_hpack_bytes(int cout, int cin, int ksize) {
    if (cout <= 0 || cin <= 0 || (ksize != 1 && ksize != 3)) return 0;
    return (size_t)((cin + 15) / 16) * ksize * ksize * 4 * ((cout + 15) / 16) * 64 * sizeof(float);
}

extern "C" int srbh_hpack_conv_f32(const float* w, int cout, int cin, int ksize, int transpose_flip, float* packed,
                                   void* stream) {
    SRBH_REQUIRE(w && packed && cout > 0 && cin > 0 && (ksize == 1 || ksize == 3), "srbh_hpack_conv_f32: bad arguments");
    int nchunk = (cin + 15) / 16, nob = (cout + 15) / 16;
    long total = (long)nchunk * ksize * ksize * 4 * nob * 64;
    hipLaunchKernelGGL(hpack_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, packed, cout, cin,
                       ksize, nchunk, nob, transpose_flip);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_hconv_up_supported(int H, int W) { return H > 0 && W > 0 && (W & 63) == 0 && (H & 3) == 0; }

extern "C" size_t srbh_bn_stats_bytes(int C) { return C > 0 ? (size_t)NSLOT * 2 * C * sizeof(double) : 0; }

extern "C" size_t srbh_hpack_h16_bytes(int cout, int cin, int ksize) {
    if (cout <= 0 || cin <= 0 || (ksize != 1 && ksize != 3)) return 0;
    return (size_t)((cin + 15) / 16) * ksize * ksize * ((cout + 15) / 16) * 64 * 4 * sizeof(_Float16);
}

extern "C" int srbh_hpack_conv_h16(const float* w, int cout, int cin, int ksize, int transpose_flip, int bf16, void* packed, void* stream) {
    SRBH_REQUIRE(w && packed && cout > 0 && cin > 0 && (ksize == 1 || ksize == 3), "srbh_hpack_conv_h16: bad arguments");
    int nchunk = (cin + 15) / 16, nob = (cout + 15) / 16;
    long total = (long)nchunk * ksize * ksize * nob * 64 * 4;
    hipLaunchKernelGGL(hpack_h16_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (short*)packed, cout,
                       cin, ksize, nchunk, nob, transpose_flip, bf16);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

// several 16-bit weight packs in ONE launch (round 5): a training step re-packs every head conv weight twice (forward fp16, gradient bf16
// transposed + flipped) because the optimizer has just changed it -- 56 launches of ~5 us between the head's kernels; hrfuse.py registers the
// packs it makes and refreshes them all right behind the optimizer's step.  Same arithmetic per element as hpack_h16_kernel.
__global__ __launch_bounds__(256) void hpack_h16_many_kernel(const srbh_hpack_desc* __restrict__ table) {
    const srbh_hpack_desc d = table[blockIdx.y];
    const int taps = d.ksize * d.ksize, nchunk = (d.cin + 15) / 16, nob = (d.cout + 15) / 16;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)nchunk * taps * nob * 64 * 4;
    if (idx >= total) return;
    const int j = idx & 3, lane = (idx >> 2) & 63;
    long f = idx >> 8;
    const int ob = f % nob; f /= nob;
    const int tap = f % taps;
    const int chunk = (int)(f / taps);
    const int oc = ob * 16 + (lane & 15), ic = chunk * 16 + (lane >> 4) * 4 + j;
    float v = 0.f;
    if (oc < d.cout && ic < d.cin)
        v = d.transpose_flip ? d.w[((long)ic * d.cout + oc) * taps + (taps - 1 - tap)] : d.w[((long)oc * d.cin + ic) * taps + tap];
    short* out = (short*)d.out;
    if (d.bf16) {
        const unsigned u = __builtin_bit_cast(unsigned, v);
        out[idx] = (short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    } else {
        out[idx] = __builtin_bit_cast(short, (_Float16)v);
    }
}

extern "C" int srbh_hpack_conv_h16_many(const srbh_hpack_desc* table_dev, int n, long max_elems, void* stream) {
    SRBH_REQUIRE(table_dev && n > 0 && max_elems > 0, "srbh_hpack_conv_h16_many: bad arguments");
    hipLaunchKernelGGL(hpack_h16_many_kernel, dim3((unsigned)((max_elems + 255) / 256), n), dim3(256), 0, (hipStream_t)stream, table_dev);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

static int hconv_impl(const srbh_hconv_args* a, void* stream, const int opt) {
    const bool h16 = opt != 0;
    SRBH_REQUIRE(a && a->src0 && a->w && a->out, "srbh_hconv_f32: null pointer");
    SRBH_REQUIRE(a->c0 > 0 && a->c1 >= 0 && (a->c1 == 0 || a->src1), "srbh_hconv_f32: bad channel split %d+%d", a->c0, a->c1);
    SRBH_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0, "srbh_hconv_f32: bad geometry");
    SRBH_REQUIRE(a->ksize == 3 || a->ksize == 1, "srbh_hconv_f32: ksize must be 1 or 3 (got %d)", a->ksize);
    SRBH_REQUIRE(a->cout >= 1 && a->cout <= 64, "srbh_hconv_f32: cout must be in 1..64 (got %d)", a->cout);
    SRBH_REQUIRE(!a->pixelshuffle2 || a->cout % 4 == 0, "srbh_hconv_f32: PixelShuffle(2) needs cout %% 4 == 0");
    const int nob = (a->cout + 15) / 16;
    SRBH_REQUIRE(nob == 1 || nob == 2 || nob == 4, "srbh_hconv_f32: cout must be <=32 or in 49..64 (got %d)", a->cout);
    HParams p;
    p.src0 = a->src0; p.src1 = a->src1; p.c0 = a->c0; p.c1 = a->c1;
    p.pre_scale = a->pre_scale; p.pre_shift = a->pre_shift; p.pre_relu = a->pre_relu;
    p.w = a->w; p.bias = a->bias;
    p.cout = nob * 16; p.cout_store = a->cout;
    p.ld0 = a->src0_ld > 0 ? a->src0_ld : a->c0;
    p.ld1 = a->src1_ld > 0 ? a->src1_ld : a->c1;
    p.out_ld = a->out_ld > 0 ? a->out_ld : a->cout;
    p.out_coff = a->out_coff;
    p.post_lrelu = a->post_lrelu;
    p.res1 = a->res1; p.res1_ld = a->res1_ld; p.res1_scale = a->res1_scale;
    p.res2 = a->res2; p.res2_ld = a->res2_ld; p.res2_scale = a->res2_scale;
    p.post_scale = a->post_scale; p.post_shift = a->post_shift; p.post_relu = a->post_relu;
    p.io_h16 = a->io_h16;
    p.bstat_c = a->bstat_c; p.bstat_mean = a->bstat_mean; p.bstat_invstd = a->bstat_invstd; p.bstat_ms = a->bstat_ms; p.bstat_mh = a->bstat_mh;
    if (a->bstat_c)
        SRBH_REQUIRE(a->stats && a->bstat_mean && a->bstat_invstd && !a->res1 && a->cout == 16 && (a->bstat_ms == nullptr) == (a->bstat_mh == nullptr) &&
                     ((uintptr_t)a->bstat_c & 15) == 0, "srbh_hconv: backward-statistics epilogue needs stats, mean, invstd, no residual, cout == 16");
    if (a->io_h16) {
        SRBH_REQUIRE(opt != 0, "srbh_hconv: 16-bit tensors in memory (io_h16) need srbh_hconv_h16 (element type = operand type: fp16 / bf16)");
        SRBH_REQUIRE((a->io_h16 & ~15) == 0, "srbh_hconv_h16: unknown io_h16 bits");
        SRBH_REQUIRE(!(a->io_h16 & SRBH_IO_OUT_H16) || ((!a->pixelshuffle2 || (HCONV_PS2_STAGE && a->cout == 64 && !a->res1)) && a->cout % 4 == 0 && p.out_ld % 4 == 0 && a->out_coff % 4 == 0),
                     "srbh_hconv_h16: a 16-bit output needs 4-aligned channels and no PixelShuffle store");
        SRBH_REQUIRE(!(a->io_h16 & SRBH_IO_RES1_H16) || (a->res1 && !a->res2), "srbh_hconv_h16: fp16 residual: res1 only");
        // the 16-bit staging of fp16 sources exists in the vectorised path only
        SRBH_REQUIRE((a->c0 & 3) == 0 && (a->c1 & 3) == 0 && (p.ld0 & 3) == 0 && (a->c1 == 0 || (p.ld1 & 3) == 0),
                     "srbh_hconv_h16: fp16 activations need channel counts and strides that are multiples of 4");
    }
    SRBH_REQUIRE(!a->post_scale || a->post_shift, "srbh_hconv_f32: post_scale needs post_shift");
    SRBH_REQUIRE(!a->pixelshuffle2 || (a->out_ld <= 0 && a->out_coff == 0), "srbh_hconv_f32: PixelShuffle store needs a dense output");
    SRBH_REQUIRE(!a->res1 || (a->cout % 4 == 0 && a->res1_ld % 4 == 0), "srbh_hconv_f32: residual epilogue needs 4-aligned channels");
    SRBH_REQUIRE(!a->res2 || a->res1, "srbh_hconv_f32: res2 requires res1");
    p.B = a->B; p.H = a->H; p.W = a->W; p.ps2 = a->pixelshuffle2;
    p.out = a->out; p.stats = a->stats;
    hipStream_t st = (hipStream_t)stream;
    if (a->stats && !a->stats_clean) { if (int rc = zero_async(a->stats, srbh_bn_stats_bytes(p.cout), st)) return rc; }
    const int B = a->B, H = a->H, W = a->W;
    // 16-bit operand forms: 4-row tiles (12.7 KiB staged tile; measured level with the 8-row form on single-chunk convs and
    // 10-25 % ahead on the multi-chunk ones)
#define SRBH_H16_DISPATCH(OPT_)                                                                                             \
    do {                                                                                                                    \
        if (a->ksize == 3)                                                                                                  \
            return nob == 1 ? launch_hconv<1, 3, 1, OPT_>(p, B, H, W, st)                                                   \
                            : (nob == 2 ? launch_hconv<2, 3, 1, OPT_>(p, B, H, W, st) : launch_hconv<4, 3, 1, OPT_>(p, B, H, W, st)); \
        return nob == 1 ? launch_hconv<1, 1, 1, OPT_>(p, B, H, W, st)                                                       \
                        : (nob == 2 ? launch_hconv<2, 1, 1, OPT_>(p, B, H, W, st) : launch_hconv<4, 1, 1, OPT_>(p, B, H, W, st)); \
    } while (0)
    if (a->pixelshuffle2 == 2) {      // the Upsampler conv with sub-pixel-major weight rows: its own persistent kernel (srbh_hconv_up_kernel.h)
        const bool s16 = (a->io_h16 & SRBH_IO_SRC0_H16) != 0, o16u = (a->io_h16 & SRBH_IO_OUT_H16) != 0;
        SRBH_REQUIRE(opt == 1 && a->ksize == 3 && a->c0 == 16 && a->c1 == 0 && a->cout == 64 && !a->pre_scale && !a->pre_relu && !a->res1 && !a->res2 &&
                     !a->stats && !a->post_scale && !a->post_relu && !a->post_lrelu && !a->bstat_c && srbh_hconv_up_supported(H, W) && (p.ld0 & 3) == 0 &&
                     ((uintptr_t)a->src0 & (s16 ? 7 : 15)) == 0 && ((uintptr_t)a->out & (o16u ? 7 : 15)) == 0 && (a->io_h16 & ~(SRBH_IO_SRC0_H16 | SRBH_IO_OUT_H16)) == 0,
                     "srbh_hconv_h16: pixelshuffle2 == 2 (sub-pixel-major pack) is the fp16 16 -> 64 3x3 form without pre / post ops, W %% 64 == 0, H %% 4 == 0");
        static const int up_wgs = getenv("SRBH_HCONV_UP_WGS") ? atoi(getenv("SRBH_HCONV_UP_WGS")) : 768;
        p.tiles_x = W / 64;
        p.tiles_per_img = p.tiles_x * (H / 4);
        p.ntiles = p.tiles_per_img * B;
        p.tiles_per_xcd = (p.ntiles + 7) / 8;
        const int per_xcd = p.tiles_per_xcd < up_wgs / 8 ? p.tiles_per_xcd : up_wgs / 8;
        constexpr int LDSUP = 2 * 6 * 66 * 32 + 36 * 64 * 8;
        if (s16 && o16u) hipLaunchKernelGGL((hconv_up_kernel<1, 1>), dim3(per_xcd * 8), dim3(256), LDSUP, st, p);
        else if (s16) hipLaunchKernelGGL((hconv_up_kernel<1, 0>), dim3(per_xcd * 8), dim3(256), LDSUP, st, p);
        else if (o16u) hipLaunchKernelGGL((hconv_up_kernel<0, 1>), dim3(per_xcd * 8), dim3(256), LDSUP, st, p);
        else hipLaunchKernelGGL((hconv_up_kernel<0, 0>), dim3(per_xcd * 8), dim3(256), LDSUP, st, p);
        SRBH_HIP(hipGetLastError());
        count_path(PATH_HCONV_UP);
        return SRBH_OK;
    }
    // the dominant layer shape has its own persistent, double-buffered kernel (srbh_hconv16_kernel.h)
    static const int k16_wgs = getenv("SRBH_HCONV16_WGS") ? atoi(getenv("SRBH_HCONV16_WGS")) : 768;     // 0 = always the template
    const bool src16 = (a->io_h16 & SRBH_IO_SRC0_H16) != 0, o16 = (a->io_h16 & SRBH_IO_OUT_H16) != 0, r16 = (a->io_h16 & SRBH_IO_RES1_H16) != 0;
    const bool full16 = a->cout == 16 && (p.out_ld & 3) == 0 && (p.out_coff & 3) == 0 && ((uintptr_t)a->out & (o16 ? 7 : 15)) == 0;
    const bool narrow = a->cout < 16 && !a->stats && !a->res1 && !o16;         // conv_last (1 / 7 channels): scalar stores
    // narrow INPUT (the data gradients of the 1- / 7-channel output convs): fp32 source of < 16 channels, any pixel stride
    const bool nin = a->c0 < 16 && a->c1 == 0 && !src16 && !a->pre_scale && !a->pre_relu && !a->bstat_c && full16 && ((uintptr_t)a->src0 & 3) == 0;
    if (opt != 0 && k16_wgs >= 8 && a->ksize == 3 && (full16 || narrow) && (a->c0 == 16 || nin) && a->c1 == 0 && (W & 63) == 0 && (H & 3) == 0 &&
        !a->pixelshuffle2 && !a->res2 && !a->post_lrelu && (nin || (p.ld0 & 3) == 0) &&
        (!a->res1 || (a->res1_ld & 3) == 0) && (nin || ((uintptr_t)a->src0 & (src16 ? 7 : 15)) == 0) && ((uintptr_t)a->res1 & (r16 ? 7 : 15)) == 0) {
        count_path(PATH_HCONV16);
        p.tiles_x = W / 64;
        p.tiles_per_img = p.tiles_x * (H / 4);
        p.ntiles = p.tiles_per_img * B;
        p.tiles_per_xcd = (p.ntiles + 7) / 8;
        const int per_xcd = p.tiles_per_xcd < k16_wgs / 8 ? p.tiles_per_xcd : k16_wgs / 8;
        constexpr int LDS16 = 2 * 6 * 66 * 32;
        const int io = (r16 && a->res1 ? 1 : 0) | (o16 ? 2 : 0);
#define SRBH_K16(O_, S_, I_) hipLaunchKernelGGL((hconv16_kernel<O_, S_, I_>), dim3(per_xcd * 8), dim3(256), LDS16, st, p)
#define SRBH_K16_IO(O_, S_) do { switch (io) { case 0: SRBH_K16(O_, S_, 0); break; case 1: SRBH_K16(O_, S_, 1); break; \
                                               case 2: SRBH_K16(O_, S_, 2); break; default: SRBH_K16(O_, S_, 3); } } while (0)
        if (a->bstat_c) {       // backward-statistics epilogues: the bf16-operand (data gradient) forms only
            SRBH_REQUIRE(opt == 2, "srbh_hconv_h16: the backward-statistics epilogue belongs to the bf16 data-gradient form");
#define SRBH_K16_BS(S_, I_) hipLaunchKernelGGL((hconv16_kernel<2, S_, I_, 1>), dim3(per_xcd * 8), dim3(256), LDS16, st, p)
            if (src16 && o16) SRBH_K16_BS(1, 2);
            else if (src16) SRBH_K16_BS(1, 0);
            else if (o16) SRBH_K16_BS(0, 2);
            else SRBH_K16_BS(0, 0);
#undef SRBH_K16_BS
            SRBH_HIP(hipGetLastError());
            return SRBH_OK;
        }
        if (nin) {
#define SRBH_K16_NIN(O_, I_) hipLaunchKernelGGL((hconv16_kernel<O_, 0, I_, 0, 1>), dim3(per_xcd * 8), dim3(256), LDS16, st, p)
            if (opt == 1) { switch (io) { case 0: SRBH_K16_NIN(1, 0); break; case 1: SRBH_K16_NIN(1, 1); break; case 2: SRBH_K16_NIN(1, 2); break; default: SRBH_K16_NIN(1, 3); } }
            else { switch (io) { case 0: SRBH_K16_NIN(2, 0); break; case 1: SRBH_K16_NIN(2, 1); break; case 2: SRBH_K16_NIN(2, 2); break; default: SRBH_K16_NIN(2, 3); } }
#undef SRBH_K16_NIN
        } else
        if (opt == 1 && !src16) SRBH_K16_IO(1, 0);
        else if (opt == 1) SRBH_K16_IO(1, 1);
        else if (!src16) SRBH_K16_IO(2, 0);
        else SRBH_K16_IO(2, 1);
#undef SRBH_K16_IO
#undef SRBH_K16
        SRBH_HIP(hipGetLastError());
        return SRBH_OK;
    }
    count_path(PATH_HCONV_TEMPLATE);
    SRBH_REQUIRE(!a->bstat_c, "srbh_hconv: the backward-statistics epilogue exists in the persistent 16 -> 16 3x3 kernel only (16-bit operand modes, W %% 64 == 0, H %% 4 == 0)");
    // (the template stages a 16-bit source as it is: no transform on the way)
    SRBH_REQUIRE(!src16 || (!a->pre_scale && !a->pre_relu), "srbh_hconv_h16: a 16-bit src0 takes a pre-affine / ReLU only in the 16 -> 16 3x3 form");
    if (opt == 1) SRBH_H16_DISPATCH(1);
    if (opt == 2) SRBH_H16_DISPATCH(2);
#undef SRBH_H16_DISPATCH
    if (hconv_rpw_small() == 1) {
        if (a->ksize == 3)
            return nob == 1 ? launch_hconv<1, 3, 1>(p, B, H, W, st)
                            : (nob == 2 ? launch_hconv<2, 3, 1>(p, B, H, W, st) : launch_hconv<4, 3, 1>(p, B, H, W, st));
        return nob == 1 ? launch_hconv<1, 1, 1>(p, B, H, W, st)
                        : (nob == 2 ? launch_hconv<2, 1, 1>(p, B, H, W, st) : launch_hconv<4, 1, 1>(p, B, H, W, st));
    }
    if (a->ksize == 3)
        return nob == 1 ? launch_hconv<1, 3, 2>(p, B, H, W, st)
                        : (nob == 2 ? launch_hconv<2, 3, 2>(p, B, H, W, st) : launch_hconv<4, 3, 2>(p, B, H, W, st));
    return nob == 1 ? launch_hconv<1, 1, 2>(p, B, H, W, st)
                    : (nob == 2 ? launch_hconv<2, 1, 2>(p, B, H, W, st) : launch_hconv<4, 1, 2>(p, B, H, W, st));
}

extern "C" int srbh_hconv_f32(const srbh_hconv_args* a, void* stream) { return hconv_impl(a, stream, 0); }
extern "C" int srbh_hconv_h16(const srbh_hconv_args* a, int bf16, void* stream) { return hconv_impl(a, stream, bf16 ? 2 : 1); }

// conv1 (3x3) + downsample[0] (1x1) of a BasicBlock entry over the same input: one fused pass when the shapes allow
// (srbh_hconv_entry_kernel.h), otherwise the two template launches -- the results are the same either way.
extern "C" int srbh_hconv_entry_h16(const srbh_hconv_args* c1, const srbh_hconv_args* ds, int bf16, void* stream) {
    SRBH_REQUIRE(c1 && ds, "srbh_hconv_entry_h16: null arguments");
    static const int wgs = getenv("SRBH_HCONV_ENTRY_WGS") ? atoi(getenv("SRBH_HCONV_ENTRY_WGS")) : 768;      // 0 = never fuse
    const int cin = c1->c0 + c1->c1;
    const int ld0 = c1->src0_ld > 0 ? c1->src0_ld : c1->c0, ld1 = c1->src1_ld > 0 ? c1->src1_ld : c1->c1;
    const int dld0 = ds->src0_ld > 0 ? ds->src0_ld : ds->c0, dld1 = ds->src1_ld > 0 ? ds->src1_ld : ds->c1;
    const bool same = c1->src0 == ds->src0 && c1->src1 == ds->src1 && c1->c0 == ds->c0 && c1->c1 == ds->c1 && ld0 == dld0 && ld1 == dld1 &&
                      c1->B == ds->B && c1->H == ds->H && c1->W == ds->W;
    const int out1_ld = c1->out_ld > 0 ? c1->out_ld : c1->cout, out2_ld = ds->out_ld > 0 ? ds->out_ld : ds->cout;
    const bool plain = !c1->pre_scale && !c1->pre_relu && !ds->pre_scale && !ds->pre_relu && !c1->pixelshuffle2 && !ds->pixelshuffle2 &&
                       !c1->res1 && !ds->res1 && !c1->res2 && !ds->res2 && !c1->post_lrelu && !ds->post_lrelu && !ds->post_relu &&
                       !(c1->io_h16 & SRBH_IO_RES1_H16) && c1->io_h16 == ds->io_h16 && (!c1->stats == !ds->stats);
    // 16-bit sources: every source the entry reads holds fp16 (the element type of the fp16-operand form), staged verbatim
    const int srcbits = c1->io_h16 & (SRBH_IO_SRC0_H16 | SRBH_IO_SRC1_H16);
    const bool es16 = srcbits != 0;
    const bool src_ok = !es16 || (!bf16 && srcbits == (SRBH_IO_SRC0_H16 | (c1->c1 ? SRBH_IO_SRC1_H16 : 0)));
    const bool shape = c1->ksize == 3 && ds->ksize == 1 && c1->cout == 16 && ds->cout == 16 && c1->c0 > 0 && (c1->c0 & 15) == 0 &&
                       (c1->c1 & 15) == 0 && cin <= 80 && (c1->c1 == 0 || c1->src1) && c1->B > 0 && (c1->W & 63) == 0 && (c1->H & 3) == 0 &&
                       (ld0 & 3) == 0 && (c1->c1 == 0 || (ld1 & 3) == 0) && (out1_ld & 3) == 0 && (out2_ld & 3) == 0 &&
                       (c1->out_coff & 3) == 0 && (ds->out_coff & 3) == 0 &&
                       (((uintptr_t)c1->src0 | (uintptr_t)c1->src1) & (es16 ? 7 : 15)) == 0 && (((uintptr_t)c1->out | (uintptr_t)ds->out) & 7) == 0;
    if (!(wgs >= 8 && same && plain && shape && src_ok && c1->src0 && c1->w && ds->w && c1->out && ds->out)) {
        count_path(PATH_ENTRY_SPLIT);
        if (int rc = hconv_impl(c1, stream, bf16 ? 2 : 1)) return rc;
        return hconv_impl(ds, stream, bf16 ? 2 : 1);
    }
    count_path(PATH_ENTRY_FUSED);
    SRBH_REQUIRE(!c1->post_scale || c1->post_shift, "srbh_hconv_entry_h16: post_scale needs post_shift");
    SRBH_REQUIRE(!ds->post_scale || ds->post_shift, "srbh_hconv_entry_h16: post_scale needs post_shift");
    hipStream_t st = (hipStream_t)stream;
    EParams e;
    HParams& p = e.a;
    p.src0 = c1->src0; p.src1 = c1->src1; p.c0 = c1->c0; p.c1 = c1->c1;
    p.pre_scale = nullptr; p.pre_shift = nullptr; p.pre_relu = 0;
    p.w = c1->w; p.bias = c1->bias; p.cout = 16; p.cout_store = 16;
    p.B = c1->B; p.H = c1->H; p.W = c1->W; p.ps2 = 0;
    p.out = c1->out; p.stats = c1->stats;
    p.ld0 = ld0; p.ld1 = ld1; p.out_ld = out1_ld; p.out_coff = c1->out_coff;
    p.post_lrelu = 0; p.res1 = nullptr; p.res1_ld = 0; p.res1_scale = 1.f; p.res2 = nullptr; p.res2_ld = 0; p.res2_scale = 1.f;
    p.post_scale = c1->post_scale; p.post_shift = c1->post_shift; p.post_relu = c1->post_relu; p.io_h16 = c1->io_h16 & SRBH_IO_OUT_H16;
    p.bstat_c = p.bstat_mean = p.bstat_invstd = p.bstat_ms = p.bstat_mh = nullptr;
    p.tiles_x = c1->W / 64;
    p.tiles_per_img = p.tiles_x * (c1->H / 4);
    p.ntiles = p.tiles_per_img * c1->B;
    p.tiles_per_xcd = (p.ntiles + 7) / 8;
    e.w2 = ds->w; e.bias2 = ds->bias; e.post2_scale = ds->post_scale; e.post2_shift = ds->post_shift;
    e.out2 = ds->out; e.out2_ld = out2_ld; e.out2_coff = ds->out_coff; e.stats2 = ds->stats;
    e.nchunk = cin / 16;
    if (c1->stats) {
        if (!c1->stats_clean) { if (int rc = zero_async(c1->stats, srbh_bn_stats_bytes(16), st)) return rc; }
        if (!ds->stats_clean) { if (int rc = zero_async(ds->stats, srbh_bn_stats_bytes(16), st)) return rc; }
    }
    const int per_xcd = p.tiles_per_xcd < wgs / 8 ? p.tiles_per_xcd : wgs / 8;
    const int lds_b = 2 * 6 * 66 * 32 + e.nchunk * 640 * 8;
    const bool eo16 = (c1->io_h16 & SRBH_IO_OUT_H16) != 0;
    static const int e64 = getenv("SRBH_HCONV_ENTRY64") ? atoi(getenv("SRBH_HCONV_ENTRY64")) : 1;     // 0: the chunked kernel (A/B aid)
    if (e64 && es16 && !bf16 && c1->c0 == 64 && c1->c1 == 0 && ld0 == 64 && ((uintptr_t)c1->src0 & 15) == 0) {
        // HRfeature's entry on whole 128-byte pixel rows, one workgroup per CU (srbh_hconv_entry_kernel.h)
        constexpr int LDS64 = 2 * 4 * 6 * 66 * 32 + 4 * 640 * 8;
        const int px64 = p.tiles_per_xcd < 32 ? p.tiles_per_xcd : 32;
        if (eo16) {
            SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)hconv_entry64_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS64)));
            hipLaunchKernelGGL((hconv_entry64_kernel<1>), dim3(px64 * 8), dim3(256), LDS64, st, e);
        } else {
            SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)hconv_entry64_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS64)));
            hipLaunchKernelGGL((hconv_entry64_kernel<0>), dim3(px64 * 8), dim3(256), LDS64, st, e);
        }
        SRBH_HIP(hipGetLastError());
        return SRBH_OK;
    }
    if (es16 && eo16) hipLaunchKernelGGL((hconv_entry_kernel<1, 1, 1>), dim3(per_xcd * 8), dim3(256), lds_b, st, e);
    else if (es16) hipLaunchKernelGGL((hconv_entry_kernel<1, 0, 1>), dim3(per_xcd * 8), dim3(256), lds_b, st, e);
    else if (bf16 && eo16) hipLaunchKernelGGL((hconv_entry_kernel<2, 1>), dim3(per_xcd * 8), dim3(256), lds_b, st, e);
    else if (bf16) hipLaunchKernelGGL((hconv_entry_kernel<2, 0>), dim3(per_xcd * 8), dim3(256), lds_b, st, e);
    else if (eo16) hipLaunchKernelGGL((hconv_entry_kernel<1, 1>), dim3(per_xcd * 8), dim3(256), lds_b, st, e);
    else hipLaunchKernelGGL((hconv_entry_kernel<1, 0>), dim3(per_xcd * 8), dim3(256), lds_b, st, e);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_hblock16_supported(int H, int W) { return H > 0 && W > 0 && (W & 63) == 0 && (H & 3) == 0; }

/* A plain BasicBlock of the inference head (eval mode, fp16 chain) as one pass: srbh_hblock16_kernel.h */
extern "C" int srbh_hblock16_eval(const srbh_hblock16_args* a, void* stream) {
    SRBH_REQUIRE(a && a->x && a->w1 && a->w2 && a->scale1 && a->shift1 && a->scale2 && a->shift2 && a->out, "srbh_hblock16_eval: null pointer");
    SRBH_REQUIRE(a->B > 0 && srbh_hblock16_supported(a->H, a->W), "srbh_hblock16_eval: W %% 64 == 0 and H %% 4 == 0 (srbh_hblock16_supported)");
    SRBH_REQUIRE(((uintptr_t)a->x & 7) == 0 && (((uintptr_t)a->w1 | (uintptr_t)a->w2) & 7) == 0 && ((uintptr_t)a->out & (a->out_h16 ? 7 : 15)) == 0 &&
                 (((uintptr_t)a->scale1 | (uintptr_t)a->shift1 | (uintptr_t)a->scale2 | (uintptr_t)a->shift2) & 15) == 0, "srbh_hblock16_eval: misaligned tensor");
    static const int wgs = getenv("SRBH_HBLOCK16_WGS") ? atoi(getenv("SRBH_HBLOCK16_WGS")) : 512;      // two per CU
    HBlkParams p;
    p.x = a->x; p.w1 = a->w1; p.w2 = a->w2; p.s1 = a->scale1; p.h1 = a->shift1; p.s2 = a->scale2; p.h2 = a->shift2; p.out = a->out;
    p.B = a->B; p.H = a->H; p.W = a->W;
    p.tiles_x = a->W / 64;
    p.tiles_per_img = p.tiles_x * (a->H / 4);
    p.ntiles = p.tiles_per_img * a->B;
    p.tiles_per_xcd = (p.ntiles + 7) / 8;
    const int per_xcd = p.tiles_per_xcd < wgs / 8 ? p.tiles_per_xcd : (wgs >= 8 ? wgs / 8 : 1);
    constexpr int LDS_B = 2 * 8 * 68 * 32 + 6 * 66 * 32;
    hipStream_t st = (hipStream_t)stream;
    count_path(PATH_HBLOCK16);
    if (a->out_h16) hipLaunchKernelGGL((hblock16_kernel<1>), dim3(per_xcd * 8), dim3(256), LDS_B, st, p);
    else hipLaunchKernelGGL((hblock16_kernel<0>), dim3(per_xcd * 8), dim3(256), LDS_B, st, p);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_bn_finalize(const double* stats, int C, double count, const float* gamma, const float* beta,
                                float eps, float momentum, float* running_mean, float* running_var, float* scale,
                                float* shift, float* save_mean, float* save_invstd, void* stream) {
    SRBH_REQUIRE(stats && C > 0 && C <= 64 && count > 0 && scale && shift, "srbh_bn_finalize: bad arguments");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (double*)stats, C, count, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean, save_invstd, 0);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_bn_finalize_clear(double* stats, int C, double count, const float* gamma, const float* beta,
                                      float eps, float momentum, float* running_mean, float* running_var, float* scale,
                                      float* shift, float* save_mean, float* save_invstd, void* stream) {
    SRBH_REQUIRE(stats && C > 0 && C <= 64 && count > 0 && scale && shift, "srbh_bn_finalize_clear: bad arguments");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stats, C, count, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, save_mean, save_invstd, 1);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_bn_eval_scale_shift(int C, const float* gamma, const float* beta, const float* running_mean,
                                        const float* running_var, float eps, float* scale, float* shift, void* stream) {
    SRBH_REQUIRE(C > 0 && running_mean && running_var && scale && shift, "srbh_bn_eval_scale_shift: bad arguments");
    hipLaunchKernelGGL(bn_eval_kernel, dim3((C + 63) / 64), dim3(64), 0, (hipStream_t)stream, C, gamma, beta,
                       running_mean, running_var, eps, scale, shift);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

static int bn_add_relu_impl(const void* a, const float* a_scale, const float* a_shift, const void* idt, const float* i_scale,
                            const float* i_shift, float* out, long npix, int C, int io, void* stream, unsigned long long* bits = nullptr) {
    SRBH_REQUIRE(a && a_scale && a_shift && idt && out && npix > 0 && C > 0 && C % 4 == 0, "srbh_bn_add_relu: bad arguments");
    SRBH_REQUIRE((io & ~3) == 0, "srbh_bn_add_relu_io: unknown io bits");
    long n4 = npix * C / 4;
    int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
#define SRBH_BAR(A_, I_) hipLaunchKernelGGL((bn_add_relu_kernel<A_, I_>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, a_scale, a_shift, \
                                            idt, i_scale, i_shift, (floatx4*)out, n4, C, bits)
    switch (io) { case 0: SRBH_BAR(0, 0); break; case 1: SRBH_BAR(1, 0); break; case 2: SRBH_BAR(0, 1); break; default: SRBH_BAR(1, 1); break; }
#undef SRBH_BAR
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_bn_add_relu(const float* a, const float* a_scale, const float* a_shift, const float* idt,
                                const float* i_scale, const float* i_shift, float* out, long npix, int C, void* stream) {
    return bn_add_relu_impl(a, a_scale, a_shift, idt, i_scale, i_shift, out, npix, C, 0, stream);
}

/* io: SRBH_BAR_A_H16 (1) = `a` holds fp16 elements, SRBH_BAR_IDT_H16 (2) = `idt` does */
extern "C" int srbh_bn_add_relu_io(const void* a, const float* a_scale, const float* a_shift, const void* idt,
                                   const float* i_scale, const float* i_shift, float* out, long npix, int C, int io, void* stream) {
    return bn_add_relu_impl(a, a_scale, a_shift, idt, i_scale, i_shift, out, npix, C, io, stream);
}

/* the same pass, also writing the ReLU's activity bits (srbh_relu_bits_bytes(npix, C) bytes) for srbh_bn_bwd_reduce_io(SRBH_BN_REF_BITS) */
extern "C" size_t srbh_relu_bits_bytes(long npix, int C) {
    if (npix <= 0 || C <= 0 || (C & 3)) return 0;
    return (size_t)((npix * (C >> 2) + 63) / 64) * 4 * sizeof(unsigned long long);
}
extern "C" int srbh_bn_add_relu_bits(const void* a, const float* a_scale, const float* a_shift, const void* idt,
                                     const float* i_scale, const float* i_shift, float* out, void* bits, long npix, int C, int io, void* stream) {
    SRBH_REQUIRE(bits && ((uintptr_t)bits & 7) == 0, "srbh_bn_add_relu_bits: bits buffer missing / unaligned");
    return bn_add_relu_impl(a, a_scale, a_shift, idt, i_scale, i_shift, out, npix, C, io, stream, (unsigned long long*)bits);
}

extern "C" int srbh_aggregate(const float* data, float* out, int N, int H, int W, int step, void* stream) {
    SRBH_REQUIRE(data && out && N > 0 && step > 0 && H >= step && W >= step, "srbh_aggregate: bad arguments");
    long total = (long)N * (H / step) * (W / step);
    hipLaunchKernelGGL(aggregate_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, data, out, N, H, W, step);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_nchw_to_nhwc_f32(const float* src, float* dst, int B, int C, int H, int W, void* stream) {
    SRBH_REQUIRE(src && dst && B > 0 && C > 0 && H > 0 && W > 0, "srbh_nchw_to_nhwc_f32: bad arguments");
    long total = (long)B * C * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, dst, B, C, H, W);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_nearest2x_f32(const float* src, float* dst, int B, int H, int W, int C, void* stream) {
    SRBH_REQUIRE(src && dst && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && H % 2 == 0 && W % 2 == 0,
                 "srbh_nearest2x_f32: bad arguments (H, W are the OUTPUT size, C %% 4 == 0)");
    long total = (long)B * H * W * (C / 4);
    hipLaunchKernelGGL(nearest2x_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const floatx4*)src,
                       (floatx4*)dst, B, H, W, C / 4);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}
