// srbh_head_bwd.hip -- backward kernels of the HR feature / fusion head (training: SURVEY.md 3.1, train.py:254-256).
//
// The reference gets these from torch autograd over SR/HRfuse.py; here each is a hand-written gfx950 kernel:
//  * hwgrad_f32_kernel : weight gradient of a 3x3 / 1x1 conv as a GEMM over pixels on the fp32 matrix cores
//                        dW[oc][ci][tap] = sum_px dY[px][oc] * X[px + tap][ci]   (v_mfma_f32_16x16x4_f32, K = 4 pixels),
//                        X read with the same concat / folded BN+ReLU transform as the forward conv.
//  * data gradients reuse the forward conv kernel (srbh_hconv_f32) with transposed + flipped weights
//    (srbh_hpack_conv_f32(transpose_flip=1)).
//  * BatchNorm backward in three steps: per-channel reductions (sum dy, sum dy*xhat) -> per-channel constants ->
//    dc = gamma*invstd * (dy - mean(dy) - xhat * mean(dy*xhat)); the ReLU mask of the block is applied on the fly.
//  * small elementwise helpers: relu mask, in-place add, per-channel sum (bias grad), PixelShuffle(2) inverse.
#include "srbh_internal.h"

#include <type_traits>
#include <vector>
#include <algorithm>

namespace {
using namespace srbh;

typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float float2w __attribute__((ext_vector_type(2)));
typedef _Float16 half4w __attribute__((ext_vector_type(4)));
constexpr int HT_H = 8, HT_W = 64;
constexpr int NSLOT = 64;
constexpr int WS_SLOTS = 768;                  // per-workgroup partial-sum slots of the weight-gradient workspace (max grid.x)

struct WGParams {
    const float* src0; const float* src1;
    int c0, c1;
    const float* pre_scale; const float* pre_shift;
    int pre_relu;
    const float* dy;      // NHWC [B][H][W][cout_total]
    int cout_total;
    float* dw;            // OIHW fp32 [cout_total][cin][KS][KS]
    float* ws;            // [gridDim.x][nob][nchunk][TAPS*256] per-workgroup partial sums (hwgrad_reduce_kernel adds them)
    int B, H, W;
    int tiles_x, tiles_per_img, ntiles, tiles_per_xcd;
    int ld0, ld1;         // pixel strides (floats) of src0 / src1
    int io;               // SRBH_WG_SRC0_H16: src0 holds fp16 elements; SRBH_WG_DY_B16: dy holds bf16 elements (16-bit forms only)
    // ACT16 addressing (hwgrad_b16_kernel<.., XM = 1>, srbh_act16_wgrad_b16: the RRDBNet training path): src0 = fp16 chunk planes
    // [B][chunks][H+2][W+2][32] (channel ch lives in plane ch / 32), dy = bf16 chunk planes starting at channel dy_ch0
    long x_img_b, dy_img_b;
    int x_plane_b, x_row_b, dy_plane_b, dy_row_b, dy_ch0;
    int zchunk;           // 1: the 16-channel input chunk is blockIdx.z (few tiles per launch: one (tile set, oc block, chunk) per workgroup)
    // fused block-entry form (hwgrad_b16_kernel<3, DS, 0, 1>): the 1x1 downsample conv's dY over the SAME input -- one more accumulator on
    // the centre-tap fragment; its partial sums go to ws2 (layout of a ksize = 1 call)
    const float* dy2; float* ws2;
};

// 4 elements of a 16-bit tensor (raw bits in a float2w) -> fp32: fp16 activations / bf16 gradients
__device__ __forceinline__ floatx4 widen_h4(const float2w raw) {
    const half4w h = __builtin_bit_cast(half4w, raw);
    return floatx4{(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
}
// (bit casts go through WHOLE vectors: __builtin_bit_cast of one ext-vector element -- bit_cast(unsigned, v[j]) -- returned element 0
//  for every j with this hipcc: every quad came back as (e0, e1, e0, e1).  Found by tools/io16_unit.py; now a test.)
typedef unsigned uint2q __attribute__((ext_vector_type(2)));
typedef unsigned uint4q __attribute__((ext_vector_type(4)));
__device__ __forceinline__ floatx4 widen_b4(const float2w raw) {
    const uint2q rw = __builtin_bit_cast(uint2q, raw);
    const unsigned lo = rw[0], hi = rw[1];
    return floatx4{__builtin_bit_cast(float, lo << 16), __builtin_bit_cast(float, lo & 0xffff0000u),
                   __builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
}
__device__ __forceinline__ float2w narrow_b4(const floatx4 v) {      // fp32 -> bf16 (RNE), 4 elements as raw bits
    const uint2q o = {bf16x2_rne(v[0], v[1]), bf16x2_rne(v[2], v[3])};
    return __builtin_bit_cast(float2w, o);
}
// the 16-bit element j (0..3) of two raw quads a, b -> one dword (a's element in the low half)
__device__ __forceinline__ unsigned b16_field_pair(const float2w a, const float2w b, const int j) {
    const uint2q ua = __builtin_bit_cast(uint2q, a), ub = __builtin_bit_cast(uint2q, b);
    const unsigned wa = ua[j >> 1], wb = ub[j >> 1];
    return (j & 1) ? ((wa >> 16) | (wb & 0xffff0000u)) : ((wa & 0xffffu) | (wb << 16));
}

// One workgroup walks tiles t = blockIdx.x, +gridDim.x, ... and keeps the 16(oc) x 16(ci) x taps partial sums of
// one (oc block = blockIdx.y, ci chunk) in registers; flushed once per chunk through LDS with one atomic per value.
template <int KS>
__global__ __launch_bounds__(256) void hwgrad_f32_kernel(const WGParams p) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    constexpr int TAPS = KS * KS, HALO = KS / 2;
    constexpr int ROWS = HT_H + 2 * HALO, COLS = HT_W + 2 * HALO;
    float* s_x = wsm;                          // [ROWS*COLS][16]
    float* s_dy = wsm + ROWS * COLS * 16;      // [8*64][16]
    float* s_red = s_dy;                       // reused for the cross-wave reduction at flush time
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kk = lane >> 4;
    const int cin = p.c0 + p.c1;
    const int nchunk = (cin + 15) / 16;
    const int ob = blockIdx.y;
    const bool vec0 = (p.c0 & 3) == 0 && (p.ld0 & 3) == 0, vec1 = p.c1 > 0 && (p.c1 & 3) == 0 && (p.c0 & 3) == 0 && (p.ld1 & 3) == 0;

    for (int c = 0; c < nchunk; ++c) {
        floatx4 acc[TAPS];
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) acc[tp] = floatx4{0.f, 0.f, 0.f, 0.f};
        floatx4 acc2 = {0.f, 0.f, 0.f, 0.f};
        // XCD-aware walk (see hconv_f32_kernel): XCD x = blockIdx % 8 owns the contiguous tiles [x*per_xcd, (x+1)*per_xcd), its
        // gridDim/8 workgroups sweep them side by side, so the tiles' shared halo rows are re-read from that XCD's L2
        const int t_end = min((int)(blockIdx.x & 7) * p.tiles_per_xcd + p.tiles_per_xcd, p.ntiles);
        for (int t = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3); t < t_end; t += gridDim.x >> 3) {
            const int img = t / p.tiles_per_img;
            const int trem = t - img * p.tiles_per_img;
            const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
            const int Y0 = ty * HT_H, X0 = tx * HT_W;
            __syncthreads();
            // Fast path (all 4-channel groups are aligned 16-byte loads): every global load of the tile pair is issued
            // before the first LDS store; the one-load-per-iteration loops below expose one memory latency per
            // iteration (19 per tile) and made the kernel staging-latency bound.
            constexpr int NIX = (ROWS * COLS * 4 + 255) / 256, NID = HT_H * HT_W * 4 / 256;
            const bool fast = vec0 && (p.c1 == 0 || vec1) && (cin & 3) == 0 && (p.cout_total & 3) == 0 &&
                              ob * 16 + 15 < p.cout_total;
            if (fast) {
                floatx4 lx[NIX], ld[NID];
#pragma unroll
                for (int it = 0; it < NIX; ++it) {
                    const int u = tid + it * 256;
                    lx[it] = floatx4{0.f, 0.f, 0.f, 0.f};
                    if (u < ROWS * COLS * 4) {
                        const int cg = u & 3, pix = u >> 2;
                        const int r = pix / COLS, col = pix - r * COLS;
                        const int y = Y0 + r - HALO, x = X0 + col - HALO;
                        const int ch = c * 16 + cg * 4;
                        if (y >= 0 && y < p.H && x >= 0 && x < p.W && ch < cin) {
                            const long pixi = ((long)img * p.H + y) * p.W + x;
                            if (ch < p.c0) {
                                floatx4 a = *(const floatx4*)(p.src0 + pixi * p.ld0 + ch);
                                if (p.pre_scale) a = a * *(const floatx4*)(p.pre_scale + ch) + *(const floatx4*)(p.pre_shift + ch);
                                if (p.pre_relu) {
#pragma unroll
                                    for (int j = 0; j < 4; ++j) a[j] = fmaxf(a[j], 0.f);
                                }
                                lx[it] = a;
                            } else {
                                lx[it] = *(const floatx4*)(p.src1 + pixi * p.ld1 + (ch - p.c0));
                            }
                        }
                    }
                }
#pragma unroll
                for (int it = 0; it < NID; ++it) {
                    const int u = tid + it * 256;
                    const int cg = u & 3, pix = u >> 2;
                    const int y = Y0 + (pix >> 6), x = X0 + (pix & 63);
                    ld[it] = floatx4{0.f, 0.f, 0.f, 0.f};
                    if (y < p.H && x < p.W)
                        ld[it] = *(const floatx4*)(p.dy + (((long)img * p.H + y) * p.W + x) * p.cout_total + ob * 16 + cg * 4);
                }
#pragma unroll
                for (int it = 0; it < NIX; ++it) {
                    const int u = tid + it * 256;
                    if (u < ROWS * COLS * 4) *(floatx4*)(s_x + (u >> 2) * 16 + (u & 3) * 4) = lx[it];
                }
#pragma unroll
                for (int it = 0; it < NID; ++it) {
                    const int u = tid + it * 256;
                    *(floatx4*)(s_dy + (u >> 2) * 16 + (u & 3) * 4) = ld[it];
                }
            } else {
            for (int u = tid; u < ROWS * COLS * 4; u += 256) {   // X tile, pixel-major, 4 channels per thread
                const int cg = u & 3, pix = u >> 2;
                const int r = pix / COLS, col = pix - r * COLS;
                const int y = Y0 + r - HALO, x = X0 + col - HALO;
                const int ch = c * 16 + cg * 4;
                floatx4 v = {0.f, 0.f, 0.f, 0.f};
                if (y >= 0 && y < p.H && x >= 0 && x < p.W && ch < cin) {
                    const long pixi = ((long)img * p.H + y) * p.W + x;
                    if (vec0 && ch + 3 < p.c0) {
                        floatx4 a = *(const floatx4*)(p.src0 + pixi * p.ld0 + ch);
                        if (p.pre_scale) a = a * *(const floatx4*)(p.pre_scale + ch) + *(const floatx4*)(p.pre_shift + ch);
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = p.pre_relu ? fmaxf(a[j], 0.f) : a[j];
                    } else if (vec1 && ch >= p.c0 && ch + 3 < cin) {
                        v = *(const floatx4*)(p.src1 + pixi * p.ld1 + (ch - p.c0));
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
                *(floatx4*)(s_x + pix * 16 + cg * 4) = v;
            }
            for (int u = tid; u < HT_H * HT_W * 4; u += 256) {   // dY tile (16 channels of block `ob`)
                const int cg = u & 3, pix = u >> 2;
                const int r = pix >> 6, col = pix & 63;
                const int y = Y0 + r, x = X0 + col;
                floatx4 v = {0.f, 0.f, 0.f, 0.f};
                if (y < p.H && x < p.W) {
                    const float* s = p.dy + (((long)img * p.H + y) * p.W + x) * p.cout_total + ob * 16 + cg * 4;
                    if ((p.cout_total & 3) == 0 && ob * 16 + cg * 4 + 3 < p.cout_total) {
                        v = *(const floatx4*)s;
                    } else {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            if (ob * 16 + cg * 4 + j < p.cout_total) v[j] = s[j];
                    }
                }
                *(floatx4*)(s_dy + pix * 16 + cg * 4) = v;
            }
            }
            __syncthreads();
            // 2 rows x 16 groups of 4 pixels per wave; the operands of k-step ks+1 are read while the MFMAs of ks run
            float a_cur, b_cur[TAPS];
            auto fetch = [&](const int ks, float& a, float (&b)[TAPS]) {
                const int row = wave * 2 + (ks >> 4), col = (ks & 15) * 4 + kk;
                a = s_dy[(row * 64 + col) * 16 + l15];                     // A[oc = l15][k = kk]
                // one base address per k-step, the taps are immediate offsets of the LDS reads (left to the compiler, every
                // tap's address was rebuilt with 3 VALU instructions: 25 per 9 MFMAs)
                const float* bp = s_x + (row * COLS + col) * 16 + l15;
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp) {
                    const int dy = tp / KS, dx = tp - dy * KS;
                    b[tp] = bp[(dy * COLS + dx) * 16];                     // B[k = kk][ci = l15]
                }
            };
            fetch(0, a_cur, b_cur);
#pragma unroll 2
            for (int ks = 0; ks < 32; ++ks) {
                float a_nxt, b_nxt[TAPS];
                fetch(ks + 1 < 32 ? ks + 1 : ks, a_nxt, b_nxt);
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp) acc[tp] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur, b_cur[tp], acc[tp], 0, 0, 0);
                a_cur = a_nxt;
#pragma unroll
                for (int tp = 0; tp < TAPS; ++tp) b_cur[tp] = b_nxt[tp];
            }
        }
        // flush: D[row = oc = kk*4 + r][col = ci = l15]
        __syncthreads();
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_red[((wave * TAPS + tp) * 16 + kk * 4 + r) * 16 + l15] = acc[tp][r];
        __syncthreads();
        for (int u = tid; u < TAPS * 256; u += 256) {
            const float v = s_red[u] + s_red[TAPS * 256 + u] + s_red[2 * TAPS * 256 + u] + s_red[3 * TAPS * 256 + u];
            // no atomics: 512 workgroups adding into the same 2 304 addresses cost 60 us of a 318 us layer (and made the
            // gradient order-dependent); every workgroup stores its partial, a second tiny kernel adds them in order
            p.ws[(((long)blockIdx.x * gridDim.y + ob) * nchunk + c) * (TAPS * 256) + u] = v;
        }
    }
}

// ---- 16-bit-operand weight gradient (mixed-precision training: hrfuse.set_head_precision("f16")) ----------------------------------
// Same GEMM over pixels, same tile walk, workspace and deterministic two-stage reduction as hwgrad_f32_kernel, but the products run
// on v_mfma_f32_16x16x16_bf16 (K = 16 pixels per instruction; the fp32 form's K = 4 at 32 cycles made the fp32 kernel MFMA-bound at
// ~2x its HBM time).  Both operands are rounded to bf16 (RNE) while staged -- dY needs bf16's exponent range, see srbh_head.hip --
// and accumulated in fp32.  K is the pixel axis, so the 16-bit operands must be contiguous along PIXELS: the staging transposes
// 4 pixels x 4 channels in registers and writes channel-major rows ([channel][row][pixel], 2 pixels per dword; channel stride
// = 4 mod 64 dwords: the 8-byte fragment reads and the staging writes are bank-conflict free).  A tap's dx = -1/+1 fragments are
// funnel-shifted (v_alignbit) out of the aligned quad and one dword of its neighbour.
typedef short short4w __attribute__((ext_vector_type(4)));
typedef unsigned uint2w __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned bf16_pair(float lo, float hi) { return bf16x2_rne(lo, hi); }

template <int KS>
struct WG16 {
    static constexpr int TAPS = KS * KS, HALO = KS / 2;
    static constexpr int ROWS = HT_H + 2 * HALO;
    static constexpr int QX = KS == 3 ? 18 : 16;       // staged 4-pixel groups per row: image columns X0-4 .. X0+67 (3x3) / X0 .. X0+63
    static constexpr int XOFF = KS == 3 ? 4 : 0;       // staged column of image column X0
    static constexpr int SX = KS == 3 ? 388 : 260;     // dwords per staged X channel (>= ROWS*QX*2, = 4 mod 64)
    static constexpr int SD = 260;                     // dwords per staged dY channel (8 rows x 64 pixels / 2 + 4)
    static constexpr int LDS_B = (16 * SX + 16 * SD) * 4;   // >= the flush buffer (4 waves x TAPS x 256 floats)
    static constexpr int LDS_B2 = (16 * SX + 32 * SD) * 4;  // + the second dY tile of the fused block-entry form (>= 4 x (TAPS + 1) x 256 floats)
};

// DS = 1: dY holds bf16 elements in memory (an internal gradient tensor of the training step): its bits are the operand
// XM = 1: both tensors are ACT16 chunk planes (WGParams::x_* / dy_*): X fp16, dY bf16 (DS must be 1)
// D2 = 1 (KS = 3, XM = 0): fused BasicBlock entry (SR/HRfuse.py:142-159): conv1 (3x3) and downsample[0] (1x1) read the same input, so
//   their weight gradients share the staged X tile -- the 1x1 gradient is one more MFMA per K step on the centre-tap fragment with its
//   own dY (p.dy2, same element type and channel count as dy); what bounds these kernels is the X staging (fp32 / fp16 -> bf16, 4x4
//   register transposes into channel-major rows), which the separate 1x1 launch repeated in full.
template <int KS, int DS, int XM = 0, int D2 = 0>
__global__ __launch_bounds__(256) void hwgrad_b16_kernel(const WGParams p) {
    static_assert(D2 == 0 || (KS == 3 && XM == 0), "the fused entry form is the 3x3 NHWC kernel");
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    using G = WG16<KS>;
    constexpr int TAPS = G::TAPS, HALO = G::HALO, ROWS = G::ROWS, QX = G::QX, SX = G::SX, SD = G::SD;
    static_assert(G::LDS_B >= 4 * TAPS * 256 * 4, "flush buffer must fit");
    unsigned* s_x = (unsigned*)wsm;                 // [16 ci][SX]
    unsigned* s_dy = s_x + 16 * SX;                 // [16 oc][SD]
    unsigned* s_dy2 = s_dy + 16 * SD;               // [16 oc][SD] (D2)
    float* s_red = wsm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kk = lane >> 4;
    const int cin = p.c0 + p.c1;
    const int nchunk = (cin + 15) / 16;
    const int ob = blockIdx.y;
    // (zchunk: with few tiles -- the RRDBNet training path at batch 8 has 64 -- the chunk loop is spread over blockIdx.z: a workgroup's
    //  chunk iterations are a serial load -> LDS -> MFMA chain of ~4 us each, 12 of them for a 192-channel conv on a quarter-filled GPU)
    const int c_lo = p.zchunk ? (int)blockIdx.z : 0, c_hi = p.zchunk ? (int)blockIdx.z + 1 : nchunk;

    for (int c = c_lo; c < c_hi; ++c) {
        floatx4 acc[TAPS];
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) acc[tp] = floatx4{0.f, 0.f, 0.f, 0.f};
        floatx4 acc2 = {0.f, 0.f, 0.f, 0.f};
        // XCD-aware walk (see hconv_f32_kernel): XCD x = blockIdx % 8 owns the contiguous tiles [x*per_xcd, (x+1)*per_xcd), its
        // gridDim/8 workgroups sweep them side by side, so the tiles' shared halo rows are re-read from that XCD's L2
        const int t_end = min((int)(blockIdx.x & 7) * p.tiles_per_xcd + p.tiles_per_xcd, p.ntiles);
        for (int t = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3); t < t_end; t += gridDim.x >> 3) {
            const int img = t / p.tiles_per_img;
            const int trem = t - img * p.tiles_per_img;
            const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
            const int Y0 = ty * HT_H, X0 = tx * HT_W;
            // ---- stage: every global load of the tile is issued before the first LDS store.  (Issuing the NEXT tile's loads before this
            // tile's MFMAs -- a software pipeline over the walk -- needs 281 registers, one workgroup per CU: 152 -> 232 us.)
            constexpr int NIX = (ROWS * QX * 4 + 255) / 256, NID = HT_H * 16 * 4 / 256;
            typedef typename std::conditional<DS != 0, float2w, floatx4>::type ldv_t;
            floatx4 lx[NIX][4];
            ldv_t ld[NID][4];
            ldv_t ld2[D2 ? NID : 1][4];
#pragma unroll
            for (int it = 0; it < NIX; ++it) {
                const int u = tid + it * 256;
                const int cg = u & 3, q = u >> 2;
                const int r = q / QX, qc = q - r * QX;
                const int y = Y0 + r - HALO, x0 = X0 - G::XOFF + qc * 4;
                const int ch = c * 16 + cg * 4;
                const bool rowok = u < ROWS * QX * 4 && y >= 0 && y < p.H && ch < cin;
                const long rowbase = ((long)img * p.H + y) * p.W;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    floatx4 a = {0.f, 0.f, 0.f, 0.f};
                    const int x = x0 + i;
                    if (rowok && x >= 0 && x < p.W) {
                        if constexpr (XM != 0) {
                            a = widen_h4(*(const float2w*)((const char*)p.src0 + (long)img * p.x_img_b + (long)(ch >> 5) * p.x_plane_b +
                                                          (long)(y + 1) * p.x_row_b + (x + 1) * 64 + (ch & 31) * 2));
                        } else if (ch < p.c0) {
                            if (p.io & SRBH_WG_SRC0_H16)     // (uniform: fp16 elements in memory, e.g. RRDBNet features handed over as fp16)
                                a = widen_h4(*(const float2w*)((const short*)p.src0 + (rowbase + x) * p.ld0 + ch));
                            else
                                a = *(const floatx4*)(p.src0 + (rowbase + x) * p.ld0 + ch);
                            if (p.pre_scale) a = a * *(const floatx4*)(p.pre_scale + ch) + *(const floatx4*)(p.pre_shift + ch);
                            if (p.pre_relu) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) a[j] = fmaxf(a[j], 0.f);
                            }
                        } else {
                            a = *(const floatx4*)(p.src1 + (rowbase + x) * p.ld1 + (ch - p.c0));
                        }
                    }
                    lx[it][i] = a;
                }
            }
#pragma unroll
            for (int it = 0; it < NID; ++it) {
                const int u = tid + it * 256;
                const int cg = u & 3, q = u >> 2;
                const int y = Y0 + (q >> 4), x0 = X0 + (q & 15) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ldv_t a = ldv_t{};
                    if (y < p.H && x0 + i < p.W) {
                        if constexpr (XM != 0) {
                            const int dch = p.dy_ch0 + ob * 16 + cg * 4;
                            a = *(const ldv_t*)((const char*)p.dy + (long)img * p.dy_img_b + (long)(dch >> 5) * p.dy_plane_b + (long)(y + 1) * p.dy_row_b +
                                                (x0 + i + 1) * 64 + (dch & 31) * 2);
                        } else {
                            a = *(const ldv_t*)((const char*)p.dy + ((((long)img * p.H + y) * p.W + x0 + i) * p.cout_total + ob * 16 + cg * 4) * (DS ? 2 : 4));
                        }
                    }
                    ld[it][i] = a;
                    if constexpr (D2 != 0) {
                        ldv_t a2 = ldv_t{};
                        if (y < p.H && x0 + i < p.W)
                            a2 = *(const ldv_t*)((const char*)p.dy2 + ((((long)img * p.H + y) * p.W + x0 + i) * p.cout_total + ob * 16 + cg * 4) * (DS ? 2 : 4));
                        ld2[it][i] = a2;
                    }
                }
            }
            __syncthreads();                       // the previous tile's fragment reads are done
#pragma unroll
            for (int it = 0; it < NIX; ++it) {
                const int u = tid + it * 256;
                if (u < ROWS * QX * 4) {
                    const int cg = u & 3, q = u >> 2;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *(uint2w*)(s_x + (cg * 4 + j) * SX + q * 2) =
                            uint2w{bf16_pair(lx[it][0][j], lx[it][1][j]), bf16_pair(lx[it][2][j], lx[it][3][j])};
                }
            }
#pragma unroll
            for (int it = 0; it < NID; ++it) {
                const int u = tid + it * 256;
                const int cg = u & 3, q = u >> 2;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (DS != 0)
                        *(uint2w*)(s_dy + (cg * 4 + j) * SD + q * 2) = uint2w{b16_field_pair(ld[it][0], ld[it][1], j), b16_field_pair(ld[it][2], ld[it][3], j)};
                    else
                        *(uint2w*)(s_dy + (cg * 4 + j) * SD + q * 2) =
                            uint2w{bf16_pair(ld[it][0][j], ld[it][1][j]), bf16_pair(ld[it][2][j], ld[it][3][j])};
                    if constexpr (D2 != 0) {
                        if constexpr (DS != 0)
                            *(uint2w*)(s_dy2 + (cg * 4 + j) * SD + q * 2) = uint2w{b16_field_pair(ld2[it][0], ld2[it][1], j), b16_field_pair(ld2[it][2], ld2[it][3], j)};
                        else
                            *(uint2w*)(s_dy2 + (cg * 4 + j) * SD + q * 2) =
                                uint2w{bf16_pair(ld2[it][0][j], ld2[it][1][j]), bf16_pair(ld2[it][2][j], ld2[it][3][j])};
                    }
                }
            }
            __syncthreads();
            // ---- 2 rows x 4 groups of 16 pixels per wave
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int row = wave * 2 + (ks >> 2), g = ks & 3;
                const uint2w a2 = *(const uint2w*)(s_dy + l15 * SD + (row * 16 + g * 4 + kk) * 2);
                const short4w a = __builtin_bit_cast(short4w, a2);
                const unsigned* bp = s_x + l15 * SX + (row * QX + (G::XOFF >> 2) + g * 4 + kk) * 2;
#pragma unroll
                for (int dy = 0; dy < KS; ++dy) {
                    const unsigned* rp = bp + dy * QX * 2;
                    const uint2w cur = *(const uint2w*)rp;
                    if constexpr (KS == 3) {
                        const unsigned pv = rp[-1], nx = rp[2];
                        const unsigned mid = __builtin_amdgcn_alignbit(cur[1], cur[0], 16);
                        const uint2w b0 = {__builtin_amdgcn_alignbit(cur[0], pv, 16), mid};
                        const uint2w b2 = {mid, __builtin_amdgcn_alignbit(nx, cur[1], 16)};
                        acc[dy * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4w, b0), acc[dy * 3 + 0], 0, 0, 0);
                        acc[dy * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4w, cur), acc[dy * 3 + 1], 0, 0, 0);
                        acc[dy * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4w, b2), acc[dy * 3 + 2], 0, 0, 0);
                        if constexpr (D2 != 0) {
                            if (dy == 1) {
                                const uint2w d2 = *(const uint2w*)(s_dy2 + l15 * SD + (row * 16 + g * 4 + kk) * 2);
                                acc2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(short4w, d2), __builtin_bit_cast(short4w, cur), acc2, 0, 0, 0);
                            }
                        }
                    } else {
                        acc[0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4w, cur), acc[0], 0, 0, 0);
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_red[((wave * TAPS + tp) * 16 + kk * 4 + r) * 16 + l15] = acc[tp][r];
        __syncthreads();
        for (int u = tid; u < TAPS * 256; u += 256) {
            const float v = s_red[u] + s_red[TAPS * 256 + u] + s_red[2 * TAPS * 256 + u] + s_red[3 * TAPS * 256 + u];
            p.ws[(((long)blockIdx.x * gridDim.y + ob) * nchunk + c) * (TAPS * 256) + u] = v;
        }
        __syncthreads();                           // s_red aliases the staging buffers of the next chunk
        if constexpr (D2 != 0) {                   // the 1x1 gradient's partial sums, in the layout of a ksize = 1 call
#pragma unroll
            for (int r = 0; r < 4; ++r) s_red[(wave * 16 + kk * 4 + r) * 16 + l15] = acc2[r];
            __syncthreads();
            p.ws2[(((long)blockIdx.x * gridDim.y + ob) * nchunk + c) * 256 + tid] = s_red[tid] + s_red[256 + tid] + s_red[512 + tid] + s_red[768 + tid];
            __syncthreads();
        }
    }
}

// Fused BasicBlock-entry weight gradient with the CHUNK loop inside the tile walk (round 4).  hwgrad_b16_kernel<3, DS, 0, 1> walks all its
// tiles once per 16-channel chunk of the input: the 64-channel entry (HRfeature: cin = 64, fp16 features) re-staged both dY tiles four times
// and touched 32 bytes of every 128-byte pixel row per pass -- 891 us for 805 MB (0.11 of the HBM peak, profiles/r04p).  Here a tile's dY /
// dY2 are staged once, the NC chunks of X follow one another through the same LDS buffer (chunk c + 1's global loads are issued before
// chunk c's MFMAs: the rows' other three 32-byte quarters come out of L2 while they are hot), and NC x 10 accumulators stay in registers.
// Same tile walk, same wave -> row assignment, same flush: every partial sum is the bit pattern the chunk-outer kernel writes.
template <int DS, int NC>
__global__ __launch_bounds__(256) void hwgrad_entry_b16_kernel(const WGParams p) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    using G = WG16<3>;
    constexpr int TAPS = G::TAPS, HALO = G::HALO, ROWS = G::ROWS, QX = G::QX, SX = G::SX, SD = G::SD;
    unsigned* s_x = (unsigned*)wsm;                 // [16 ci][SX]
    unsigned* s_dy = s_x + 16 * SX;                 // [16 oc][SD]
    unsigned* s_dy2 = s_dy + 16 * SD;               // [16 oc][SD]
    float* s_red = wsm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kk = lane >> 4;
    const int cin = p.c0 + p.c1;
    const int ob = blockIdx.y;
    floatx4 acc[NC][TAPS];
    floatx4 acc2[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) acc[c][tp] = floatx4{0.f, 0.f, 0.f, 0.f};
        acc2[c] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    constexpr int NIX = (ROWS * QX * 4 + 255) / 256, NID = HT_H * 16 * 4 / 256;
    typedef typename std::conditional<DS != 0, float2w, floatx4>::type ldv_t;
    const int t_end = min((int)(blockIdx.x & 7) * p.tiles_per_xcd + p.tiles_per_xcd, p.ntiles);
    for (int t = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3); t < t_end; t += gridDim.x >> 3) {
        const int img = t / p.tiles_per_img;
        const int trem = t - img * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const int Y0 = ty * HT_H, X0 = tx * HT_W;
        floatx4 lx[NIX][4];
        auto load_x = [&](const int c) {
#pragma unroll
            for (int it = 0; it < NIX; ++it) {
                const int u = tid + it * 256;
                const int cg = u & 3, q = u >> 2;
                const int r = q / QX, qc = q - r * QX;
                const int y = Y0 + r - HALO, x0 = X0 - G::XOFF + qc * 4;
                const int ch = c * 16 + cg * 4;
                const bool rowok = u < ROWS * QX * 4 && y >= 0 && y < p.H && ch < cin;
                const long rowbase = ((long)img * p.H + y) * p.W;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    floatx4 a = {0.f, 0.f, 0.f, 0.f};
                    const int x = x0 + i;
                    if (rowok && x >= 0 && x < p.W) {
                        if (ch < p.c0) {
                            if (p.io & SRBH_WG_SRC0_H16)
                                a = widen_h4(*(const float2w*)((const short*)p.src0 + (rowbase + x) * p.ld0 + ch));
                            else
                                a = *(const floatx4*)(p.src0 + (rowbase + x) * p.ld0 + ch);
                            if (p.pre_scale) a = a * *(const floatx4*)(p.pre_scale + ch) + *(const floatx4*)(p.pre_shift + ch);
                            if (p.pre_relu) {
#pragma unroll
                                for (int j = 0; j < 4; ++j) a[j] = fmaxf(a[j], 0.f);
                            }
                        } else {
                            a = *(const floatx4*)(p.src1 + (rowbase + x) * p.ld1 + (ch - p.c0));
                        }
                    }
                    lx[it][i] = a;
                }
            }
        };
        auto store_x = [&]() {
#pragma unroll
            for (int it = 0; it < NIX; ++it) {
                const int u = tid + it * 256;
                if (u < ROWS * QX * 4) {
                    const int cg = u & 3, q = u >> 2;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *(uint2w*)(s_x + (cg * 4 + j) * SX + q * 2) =
                            uint2w{bf16_pair(lx[it][0][j], lx[it][1][j]), bf16_pair(lx[it][2][j], lx[it][3][j])};
                }
            }
        };
        {
            ldv_t ld[NID][4], ld2[NID][4];
            load_x(0);
#pragma unroll
            for (int it = 0; it < NID; ++it) {
                const int u = tid + it * 256;
                const int cg = u & 3, q = u >> 2;
                const int y = Y0 + (q >> 4), x0 = X0 + (q & 15) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ldv_t a = ldv_t{}, a2 = ldv_t{};
                    if (y < p.H && x0 + i < p.W) {
                        const long off = ((((long)img * p.H + y) * p.W + x0 + i) * p.cout_total + ob * 16 + cg * 4) * (DS ? 2 : 4);
                        a = *(const ldv_t*)((const char*)p.dy + off);
                        a2 = *(const ldv_t*)((const char*)p.dy2 + off);
                    }
                    ld[it][i] = a;
                    ld2[it][i] = a2;
                }
            }
            __syncthreads();                       // the previous tile's fragment reads are done
            store_x();
#pragma unroll
            for (int it = 0; it < NID; ++it) {
                const int u = tid + it * 256;
                const int cg = u & 3, q = u >> 2;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (DS != 0) {
                        *(uint2w*)(s_dy + (cg * 4 + j) * SD + q * 2) = uint2w{b16_field_pair(ld[it][0], ld[it][1], j), b16_field_pair(ld[it][2], ld[it][3], j)};
                        *(uint2w*)(s_dy2 + (cg * 4 + j) * SD + q * 2) = uint2w{b16_field_pair(ld2[it][0], ld2[it][1], j), b16_field_pair(ld2[it][2], ld2[it][3], j)};
                    } else {
                        *(uint2w*)(s_dy + (cg * 4 + j) * SD + q * 2) = uint2w{bf16_pair(ld[it][0][j], ld[it][1][j]), bf16_pair(ld[it][2][j], ld[it][3][j])};
                        *(uint2w*)(s_dy2 + (cg * 4 + j) * SD + q * 2) = uint2w{bf16_pair(ld2[it][0][j], ld2[it][1][j]), bf16_pair(ld2[it][2][j], ld2[it][3][j])};
                    }
                }
            }
            __syncthreads();
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c + 1 < NC) load_x(c + 1);          // (in flight under this chunk's MFMAs)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int row = wave * 2 + (ks >> 2), g = ks & 3;
                const uint2w a2 = *(const uint2w*)(s_dy + l15 * SD + (row * 16 + g * 4 + kk) * 2);
                const short4w a = __builtin_bit_cast(short4w, a2);
                const unsigned* bp = s_x + l15 * SX + (row * QX + (G::XOFF >> 2) + g * 4 + kk) * 2;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const unsigned* rp = bp + dy * QX * 2;
                    const uint2w cur = *(const uint2w*)rp;
                    const unsigned pv = rp[-1], nx = rp[2];
                    const unsigned mid = __builtin_amdgcn_alignbit(cur[1], cur[0], 16);
                    const uint2w b0 = {__builtin_amdgcn_alignbit(cur[0], pv, 16), mid};
                    const uint2w b2 = {mid, __builtin_amdgcn_alignbit(nx, cur[1], 16)};
                    acc[c][dy * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4w, b0), acc[c][dy * 3 + 0], 0, 0, 0);
                    acc[c][dy * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4w, cur), acc[c][dy * 3 + 1], 0, 0, 0);
                    acc[c][dy * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4w, b2), acc[c][dy * 3 + 2], 0, 0, 0);
                    if (dy == 1) {
                        const uint2w d2 = *(const uint2w*)(s_dy2 + l15 * SD + (row * 16 + g * 4 + kk) * 2);
                        acc2[c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(short4w, d2), __builtin_bit_cast(short4w, cur), acc2[c], 0, 0, 0);
                    }
                }
            }
            if (c + 1 < NC) {
                __syncthreads();                   // every wave has read chunk c's fragments
                store_x();
                __syncthreads();
            }
        }
    }
    // flush, chunk by chunk, in the layout of the chunk-outer kernel
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        __syncthreads();
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_red[((wave * TAPS + tp) * 16 + kk * 4 + r) * 16 + l15] = acc[c][tp][r];
        __syncthreads();
        for (int u = tid; u < TAPS * 256; u += 256) {
            const float v = s_red[u] + s_red[TAPS * 256 + u] + s_red[2 * TAPS * 256 + u] + s_red[3 * TAPS * 256 + u];
            p.ws[(((long)blockIdx.x * gridDim.y + ob) * NC + c) * (TAPS * 256) + u] = v;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) s_red[(wave * 16 + kk * 4 + r) * 16 + l15] = acc2[c][r];
        __syncthreads();
        p.ws2[(((long)blockIdx.x * gridDim.y + ob) * NC + c) * 256 + tid] = s_red[tid] + s_red[256 + tid] + s_red[512 + tid] + s_red[768 + tid];
    }
}

// Weight gradient of a conv with ONE input chunk and NOB output blocks (the Upsampler's 16 -> 64 convs, SR/HRfuse.py:17-44: dY = the
// PixelShuffle-inverted gradient, 64 channels): hwgrad_b16_kernel runs grid.y = NOB workgroups per tile, each staging the same X tile
// (the fp32 -> bf16 4x4 register transposes that bound these kernels).  Here the X tile is staged once and the NOB dY blocks follow one
// another through the dY buffer (block ob + 1's loads issued before block ob's MFMAs), NOB x 9 accumulators in registers.  Same walk,
// wave -> row assignment, products and flush layout as the chunk-outer kernel with grid.y = NOB: bit-identical partial sums.
template <int DS, int NOB>
__global__ __launch_bounds__(256) void hwgrad_ob_b16_kernel(const WGParams p) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    using G = WG16<3>;
    constexpr int TAPS = G::TAPS, HALO = G::HALO, ROWS = G::ROWS, QX = G::QX, SX = G::SX, SD = G::SD;
    unsigned* s_x = (unsigned*)wsm;                 // [16 ci][SX]
    unsigned* s_dy = s_x + 16 * SX;                 // [16 oc][SD]
    float* s_red = wsm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kk = lane >> 4;
    const int cin = p.c0;
    floatx4 acc[NOB][TAPS];
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob)
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) acc[ob][tp] = floatx4{0.f, 0.f, 0.f, 0.f};
    constexpr int NIX = (ROWS * QX * 4 + 255) / 256, NID = HT_H * 16 * 4 / 256;
    typedef typename std::conditional<DS != 0, float2w, floatx4>::type ldv_t;
    const int t_end = min((int)(blockIdx.x & 7) * p.tiles_per_xcd + p.tiles_per_xcd, p.ntiles);
    for (int t = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3); t < t_end; t += gridDim.x >> 3) {
        const int img = t / p.tiles_per_img;
        const int trem = t - img * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const int Y0 = ty * HT_H, X0 = tx * HT_W;
        ldv_t ld[NID][4];
        auto load_dy = [&](const int ob) {
#pragma unroll
            for (int it = 0; it < NID; ++it) {
                const int u = tid + it * 256;
                const int cg = u & 3, q = u >> 2;
                const int y = Y0 + (q >> 4), x0 = X0 + (q & 15) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    ldv_t a = ldv_t{};
                    if (y < p.H && x0 + i < p.W)
                        a = *(const ldv_t*)((const char*)p.dy + ((((long)img * p.H + y) * p.W + x0 + i) * p.cout_total + ob * 16 + cg * 4) * (DS ? 2 : 4));
                    ld[it][i] = a;
                }
            }
        };
        auto store_dy = [&]() {
#pragma unroll
            for (int it = 0; it < NID; ++it) {
                const int u = tid + it * 256;
                const int cg = u & 3, q = u >> 2;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if constexpr (DS != 0)
                        *(uint2w*)(s_dy + (cg * 4 + j) * SD + q * 2) = uint2w{b16_field_pair(ld[it][0], ld[it][1], j), b16_field_pair(ld[it][2], ld[it][3], j)};
                    else
                        *(uint2w*)(s_dy + (cg * 4 + j) * SD + q * 2) = uint2w{bf16_pair(ld[it][0][j], ld[it][1][j]), bf16_pair(ld[it][2][j], ld[it][3][j])};
                }
            }
        };
        {
            floatx4 lx[NIX][4];
#pragma unroll
            for (int it = 0; it < NIX; ++it) {
                const int u = tid + it * 256;
                const int cg = u & 3, q = u >> 2;
                const int r = q / QX, qc = q - r * QX;
                const int y = Y0 + r - HALO, x0 = X0 - G::XOFF + qc * 4;
                const int ch = cg * 4;
                const bool rowok = u < ROWS * QX * 4 && y >= 0 && y < p.H && ch < cin;
                const long rowbase = ((long)img * p.H + y) * p.W;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    floatx4 a = {0.f, 0.f, 0.f, 0.f};
                    const int x = x0 + i;
                    if (rowok && x >= 0 && x < p.W) {
                        if (p.io & SRBH_WG_SRC0_H16)
                            a = widen_h4(*(const float2w*)((const short*)p.src0 + (rowbase + x) * p.ld0 + ch));
                        else
                            a = *(const floatx4*)(p.src0 + (rowbase + x) * p.ld0 + ch);
                        if (p.pre_scale) a = a * *(const floatx4*)(p.pre_scale + ch) + *(const floatx4*)(p.pre_shift + ch);
                        if (p.pre_relu) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) a[j] = fmaxf(a[j], 0.f);
                        }
                    }
                    lx[it][i] = a;
                }
            }
            load_dy(0);
            __syncthreads();                       // the previous tile's fragment reads are done
#pragma unroll
            for (int it = 0; it < NIX; ++it) {
                const int u = tid + it * 256;
                if (u < ROWS * QX * 4) {
                    const int cg = u & 3, q = u >> 2;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *(uint2w*)(s_x + (cg * 4 + j) * SX + q * 2) =
                            uint2w{bf16_pair(lx[it][0][j], lx[it][1][j]), bf16_pair(lx[it][2][j], lx[it][3][j])};
                }
            }
            store_dy();
            __syncthreads();
        }
#pragma unroll
        for (int ob = 0; ob < NOB; ++ob) {
            if (ob + 1 < NOB) load_dy(ob + 1);      // (in flight under this block's MFMAs)
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int row = wave * 2 + (ks >> 2), g = ks & 3;
                const uint2w a2 = *(const uint2w*)(s_dy + l15 * SD + (row * 16 + g * 4 + kk) * 2);
                const short4w a = __builtin_bit_cast(short4w, a2);
                const unsigned* bp = s_x + l15 * SX + (row * QX + (G::XOFF >> 2) + g * 4 + kk) * 2;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const unsigned* rp = bp + dy * QX * 2;
                    const uint2w cur = *(const uint2w*)rp;
                    const unsigned pv = rp[-1], nx = rp[2];
                    const unsigned mid = __builtin_amdgcn_alignbit(cur[1], cur[0], 16);
                    const uint2w b0 = {__builtin_amdgcn_alignbit(cur[0], pv, 16), mid};
                    const uint2w b2 = {mid, __builtin_amdgcn_alignbit(nx, cur[1], 16)};
                    acc[ob][dy * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4w, b0), acc[ob][dy * 3 + 0], 0, 0, 0);
                    acc[ob][dy * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4w, cur), acc[ob][dy * 3 + 1], 0, 0, 0);
                    acc[ob][dy * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4w, b2), acc[ob][dy * 3 + 2], 0, 0, 0);
                }
            }
            if (ob + 1 < NOB) {
                __syncthreads();                   // every wave has read block ob's dY fragments
                store_dy();
                __syncthreads();
            }
        }
    }
#pragma unroll
    for (int ob = 0; ob < NOB; ++ob) {
        __syncthreads();
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_red[((wave * TAPS + tp) * 16 + kk * 4 + r) * 16 + l15] = acc[ob][tp][r];
        __syncthreads();
        for (int u = tid; u < TAPS * 256; u += 256) {
            const float v = s_red[u] + s_red[TAPS * 256 + u] + s_red[2 * TAPS * 256 + u] + s_red[3 * TAPS * 256 + u];
            p.ws[((long)blockIdx.x * NOB + ob) * (TAPS * 256) + u] = v;          // (= the chunk-outer layout with grid.y = NOB, nchunk = 1)
        }
    }
}

// HRfeature's entry (cin = 64, the RRDBNet features handed over as fp16 NHWC: 128 bytes per pixel): the chunked kernels above read 32 bytes
// of every pixel row per pass -- each load instruction touches 16 different 128-byte lines -- and ran at 0.11 of the HBM peak.  Here a lane
// loads 16 bytes (8 channels) and 8 lanes cover a pixel's whole row, ALL FOUR chunks of a tile are staged at once (64 channel rows in LDS,
// 130 KB with the two dY tiles: one workgroup per CU), and the next tile's global loads are issued before this tile's MFMAs (register
// prefetch: 96 + 32 registers) -- the one workgroup per CU has nothing else to hide the load latency behind.  Same walk, wave -> row
// assignment, products and flush as hwgrad_entry_b16_kernel<DS, 4>: bit-identical partial sums.
struct WG64 {
    using G = WG16<3>;
    static constexpr int NIT = (G::ROWS * G::QX * 8 + 255) / 256;           // 16-byte units of the X tile per thread
    static constexpr int LDS_B = (64 * G::SX + 32 * G::SD) * 4;
};
typedef unsigned uint4w __attribute__((ext_vector_type(4)));
typedef _Float16 half8w __attribute__((ext_vector_type(8)));
template <int DS>
__global__ __launch_bounds__(256) void hwgrad_entry64_b16_kernel(const WGParams p) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    using G = WG16<3>;
    constexpr int TAPS = G::TAPS, ROWS = G::ROWS, QX = G::QX, SX = G::SX, SD = G::SD, NIT = WG64::NIT, NID = HT_H * 16 * 4 / 256, NC = 4;
    static_assert(WG64::LDS_B >= 4 * TAPS * 256 * 4, "flush buffer must fit");
    unsigned* s_x = (unsigned*)wsm;                 // [64 ci][SX]
    unsigned* s_dy = s_x + 64 * SX;                 // [16 oc][SD]
    unsigned* s_dy2 = s_dy + 16 * SD;
    float* s_red = wsm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kk = lane >> 4;
    const int ob = blockIdx.y;
    typedef typename std::conditional<DS != 0, float2w, floatx4>::type ldv_t;
    floatx4 acc[NC][TAPS];
    floatx4 acc2[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) acc[c][tp] = floatx4{0.f, 0.f, 0.f, 0.f};
        acc2[c] = floatx4{0.f, 0.f, 0.f, 0.f};
    }
    uint4w lx[NIT][4];
    ldv_t ld[NID][4], ld2[NID][4];
    auto load_tile = [&](const int t) {
        const int img = t / p.tiles_per_img;
        const int trem = t - img * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const int Y0 = ty * HT_H, X0 = tx * HT_W;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int u = tid + it * 256;
            const int pc = u & 7, q = u >> 3;
            const int r = q / QX, qc = q - r * QX;
            const int y = Y0 + r - 1, x0 = X0 - G::XOFF + qc * 4;
            const bool rowok = u < ROWS * QX * 8 && y >= 0 && y < p.H;
            const short* rowp = (const short*)p.src0 + (((long)img * p.H + y) * p.W) * 64 + pc * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                uint4w a = {0u, 0u, 0u, 0u};
                const int x = x0 + i;
                if (rowok && x >= 0 && x < p.W) a = *(const uint4w*)(rowp + (long)x * 64);
                lx[it][i] = a;
            }
        }
#pragma unroll
        for (int it = 0; it < NID; ++it) {
            const int u = tid + it * 256;
            const int cg = u & 3, q = u >> 2;
            const int y = Y0 + (q >> 4), x0 = X0 + (q & 15) * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ldv_t a = ldv_t{}, a2 = ldv_t{};
                if (y < p.H && x0 + i < p.W) {
                    const long off = ((((long)img * p.H + y) * p.W + x0 + i) * p.cout_total + ob * 16 + cg * 4) * (DS ? 2 : 4);
                    a = *(const ldv_t*)((const char*)p.dy + off);
                    a2 = *(const ldv_t*)((const char*)p.dy2 + off);
                }
                ld[it][i] = a;
                ld2[it][i] = a2;
            }
        }
    };
    const int t_end = min((int)(blockIdx.x & 7) * p.tiles_per_xcd + p.tiles_per_xcd, p.ntiles);
    const int t_step = gridDim.x >> 3;
    int t = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3);
    if (t < t_end) load_tile(t);
    for (; t < t_end; t += t_step) {
        __syncthreads();                           // the previous tile's fragment reads are done
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int u = tid + it * 256;
            if (u < ROWS * QX * 8) {
                const int pc = u & 7, q = u >> 3;
                const half8w h0 = __builtin_bit_cast(half8w, lx[it][0]), h1 = __builtin_bit_cast(half8w, lx[it][1]);
                const half8w h2 = __builtin_bit_cast(half8w, lx[it][2]), h3 = __builtin_bit_cast(half8w, lx[it][3]);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    *(uint2w*)(s_x + (pc * 8 + j) * SX + q * 2) =
                        uint2w{bf16_pair((float)h0[j], (float)h1[j]), bf16_pair((float)h2[j], (float)h3[j])};
            }
        }
#pragma unroll
        for (int it = 0; it < NID; ++it) {
            const int u = tid + it * 256;
            const int cg = u & 3, q = u >> 2;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (DS != 0) {
                    *(uint2w*)(s_dy + (cg * 4 + j) * SD + q * 2) = uint2w{b16_field_pair(ld[it][0], ld[it][1], j), b16_field_pair(ld[it][2], ld[it][3], j)};
                    *(uint2w*)(s_dy2 + (cg * 4 + j) * SD + q * 2) = uint2w{b16_field_pair(ld2[it][0], ld2[it][1], j), b16_field_pair(ld2[it][2], ld2[it][3], j)};
                } else {
                    *(uint2w*)(s_dy + (cg * 4 + j) * SD + q * 2) = uint2w{bf16_pair(ld[it][0][j], ld[it][1][j]), bf16_pair(ld[it][2][j], ld[it][3][j])};
                    *(uint2w*)(s_dy2 + (cg * 4 + j) * SD + q * 2) = uint2w{bf16_pair(ld2[it][0][j], ld2[it][1][j]), bf16_pair(ld2[it][2][j], ld2[it][3][j])};
                }
            }
        }
        __syncthreads();
        if (t + t_step < t_end) load_tile(t + t_step);      // in flight under this tile's 4 x 80 MFMAs
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const int row = wave * 2 + (ks >> 2), g = ks & 3;
                const uint2w a2 = *(const uint2w*)(s_dy + l15 * SD + (row * 16 + g * 4 + kk) * 2);
                const short4w a = __builtin_bit_cast(short4w, a2);
                const unsigned* bp = s_x + (c * 16 + l15) * SX + (row * QX + (G::XOFF >> 2) + g * 4 + kk) * 2;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const unsigned* rp = bp + dy * QX * 2;
                    const uint2w cur = *(const uint2w*)rp;
                    const unsigned pv = rp[-1], nx = rp[2];
                    const unsigned mid = __builtin_amdgcn_alignbit(cur[1], cur[0], 16);
                    const uint2w b0 = {__builtin_amdgcn_alignbit(cur[0], pv, 16), mid};
                    const uint2w b2 = {mid, __builtin_amdgcn_alignbit(nx, cur[1], 16)};
                    acc[c][dy * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4w, b0), acc[c][dy * 3 + 0], 0, 0, 0);
                    acc[c][dy * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4w, cur), acc[c][dy * 3 + 1], 0, 0, 0);
                    acc[c][dy * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4w, b2), acc[c][dy * 3 + 2], 0, 0, 0);
                    if (dy == 1) {
                        const uint2w d2 = *(const uint2w*)(s_dy2 + l15 * SD + (row * 16 + g * 4 + kk) * 2);
                        acc2[c] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(short4w, d2), __builtin_bit_cast(short4w, cur), acc2[c], 0, 0, 0);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        __syncthreads();
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
            for (int r = 0; r < 4; ++r) s_red[((wave * TAPS + tp) * 16 + kk * 4 + r) * 16 + l15] = acc[c][tp][r];
        __syncthreads();
        for (int u = tid; u < TAPS * 256; u += 256) {
            const float v = s_red[u] + s_red[TAPS * 256 + u] + s_red[2 * TAPS * 256 + u] + s_red[3 * TAPS * 256 + u];
            p.ws[(((long)blockIdx.x * gridDim.y + ob) * NC + c) * (TAPS * 256) + u] = v;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r) s_red[(wave * 16 + kk * 4 + r) * 16 + l15] = acc2[c][r];
        __syncthreads();
        p.ws2[(((long)blockIdx.x * gridDim.y + ob) * NC + c) * 256 + tid] = s_red[tid] + s_red[256 + tid] + s_red[512 + tid] + s_red[768 + tid];
    }
}

#include "srbh_hwgrad16_kernel.h"
#include "srbh_hbwd16_kernel.h"

// Sum of the workgroups' partials in a fixed order (deterministic), two stages so that enough loads are in flight:
//   stage 1: tmp[s][u] = sum of the partials x in slice s (x = s*per .. s*per+per-1), grid (U/256, slices);
//   stage 2: dw[oc][ci][tap] = sum over s of tmp[s][u(oc, ci, tap)]
__global__ __launch_bounds__(256) void hwgrad_reduce1_kernel(const float* __restrict__ ws, float* __restrict__ tmp, long U, int gx, int per) {
    const long u = (long)blockIdx.x * 256 + threadIdx.x;
    if (u >= U) return;
    const int x0 = blockIdx.y * per;
    const int x1 = x0 + per < gx ? x0 + per : gx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int x = x0;
    for (; x + 3 < x1; x += 4) {
        a0 += ws[(long)x * U + u];
        a1 += ws[(long)(x + 1) * U + u];
        a2 += ws[(long)(x + 2) * U + u];
        a3 += ws[(long)(x + 3) * U + u];
    }
    for (; x < x1; ++x) a0 += ws[(long)x * U + u];
    tmp[(long)blockIdx.y * U + u] = (a0 + a1) + (a2 + a3);
}

__global__ void hwgrad_reduce2_kernel(const float* __restrict__ tmp, float* __restrict__ dw, long U, int slices, int nchunk, int taps,
                                      int cout, int cin) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = cout * cin * taps;
    if (idx >= total) return;
    const int tp = idx % taps;
    const int ci = (idx / taps) % cin;
    const int oc = idx / (taps * cin);
    const long u = ((long)(oc >> 4) * nchunk + (ci >> 4)) * (taps * 256) + tp * 256 + (oc & 15) * 16 + (ci & 15);
    float v = 0.f;
    for (int sidx = 0; sidx < slices; ++sidx) v += tmp[(long)sidx * U + u];
    dw[idx] = v;
}

// ---- the same two stages for SEVERAL weight gradients in one launch each (round 5).  The partial sums of a weight gradient are not needed
// before the optimizer (or a gradient all-reduce) reads dW, so the producers of one autograd node's backward can queue their reduce jobs
// (srbh_hwgrad_defer) and ONE pair of launches does them all (srbh_hwgrad_flush): in the training step these 2 x 28 launches of ~6 us each sat
// between the chip-filling kernels of the head's backward.  Same order of additions per element as the per-gradient kernels.
struct RedJob {
    const float* ws; float* tmp; float* dw;
    long U;
    int gx, nchunk, taps, cout, cin, per;
};
constexpr int RED_MAX = 24;
struct RedMany { RedJob j[RED_MAX]; int n; };
__global__ __launch_bounds__(256) void hwgrad_reduce1_many_kernel(const RedMany m) {
    const RedJob& J = m.j[blockIdx.z];
    const long u = (long)blockIdx.x * 256 + threadIdx.x;
    if (u >= J.U) return;
    const int x0 = blockIdx.y * J.per;
    const int x1 = x0 + J.per < J.gx ? x0 + J.per : J.gx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int x = x0;
    for (; x + 3 < x1; x += 4) {
        a0 += J.ws[(long)x * J.U + u];
        a1 += J.ws[(long)(x + 1) * J.U + u];
        a2 += J.ws[(long)(x + 2) * J.U + u];
        a3 += J.ws[(long)(x + 3) * J.U + u];
    }
    for (; x < x1; ++x) a0 += J.ws[(long)x * J.U + u];
    J.tmp[(long)blockIdx.y * J.U + u] = (a0 + a1) + (a2 + a3);
}
__global__ __launch_bounds__(256) void hwgrad_reduce2_many_kernel(const RedMany m, int slices) {
    const RedJob& J = m.j[blockIdx.y];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    const int total = J.cout * J.cin * J.taps;
    if (idx >= total) return;
    const int tp = idx % J.taps;
    const int ci = (idx / J.taps) % J.cin;
    const int oc = idx / (J.taps * J.cin);
    const long u = ((long)(oc >> 4) * J.nchunk + (ci >> 4)) * (J.taps * 256) + tp * 256 + (oc & 15) * 16 + (ci & 15);
    float v = 0.f;
    for (int sidx = 0; sidx < slices; ++sidx) v += J.tmp[(long)sidx * J.U + u];
    J.dw[idx] = v;
}
constexpr int RED_SLICES = 16;
thread_local bool g_red_defer = false;
thread_local std::vector<RedJob> g_red_queue;
int reduce_flush(hipStream_t st) {
    for (size_t b = 0; b < g_red_queue.size(); b += RED_MAX) {
        RedMany m = {};
        m.n = (int)std::min<size_t>(RED_MAX, g_red_queue.size() - b);
        long umax = 0;
        int tmax = 0;
        for (int k = 0; k < m.n; ++k) {
            m.j[k] = g_red_queue[b + k];
            umax = std::max(umax, m.j[k].U);
            tmax = std::max(tmax, m.j[k].cout * m.j[k].cin * m.j[k].taps);
        }
        hipLaunchKernelGGL(hwgrad_reduce1_many_kernel, dim3((unsigned)((umax + 255) / 256), RED_SLICES, m.n), dim3(256), 0, st, m);
        SRBH_HIP(hipGetLastError());
        hipLaunchKernelGGL(hwgrad_reduce2_many_kernel, dim3((tmax + 255) / 256, m.n), dim3(256), 0, st, m, RED_SLICES);
        SRBH_HIP(hipGetLastError());
    }
    g_red_queue.clear();
    return SRBH_OK;
}
// the two-stage ordered reduce of one weight gradient: now, or queued for srbh_hwgrad_flush
int reduce_partials(const float* ws, float* dw, long U, int gx, int nchunk, int taps, int cout, int cin, hipStream_t st) {
    float* tmp = (float*)ws + (long)WS_SLOTS * U;
    const int per = (gx + RED_SLICES - 1) / RED_SLICES;
    if (g_red_defer) {
        g_red_queue.push_back(RedJob{ws, tmp, dw, U, gx, nchunk, taps, cout, cin, per});
        return SRBH_OK;
    }
    hipLaunchKernelGGL(hwgrad_reduce1_kernel, dim3((unsigned)((U + 255) / 256), RED_SLICES), dim3(256), 0, st, ws, tmp, U, gx, per);
    SRBH_HIP(hipGetLastError());
    const int total = cout * cin * taps;
    hipLaunchKernelGGL(hwgrad_reduce2_kernel, dim3((total + 255) / 256), dim3(256), 0, st, tmp, dw, U, RED_SLICES, nchunk, taps, cout, cin);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

// one-stage form for few partial slots (gx <= 128: small batches, where the two-stage form is two launch latencies for 28 MB of L2 reads):
// dw[oc][ci][tap] = sum over x = 0 .. gx-1 of ws[x][u], in that fixed order
__global__ void hwgrad_reduce_direct_kernel(const float* __restrict__ ws, float* __restrict__ dw, long U, int gx, int nchunk, int taps, int cout,
                                            int cin) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int total = cout * cin * taps;
    if (idx >= total) return;
    const int tp = idx % taps;
    const int ci = (idx / taps) % cin;
    const int oc = idx / (taps * cin);
    const long u = ((long)(oc >> 4) * nchunk + (ci >> 4)) * (taps * 256) + tp * 256 + (oc & 15) * 16 + (ci & 15);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};      // 8 loads in flight; the order of the additions is fixed all the same
    int x = 0;
    for (; x + 8 <= gx; x += 8) {
        float t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = ws[(long)(x + k) * U + u];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] += t[k];
    }
    for (; x < gx; ++x) v[0] += ws[(long)x * U + u];
    dw[idx] = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
}

// ---- elementwise / reduction helpers (NHWC fp32, C % 4 == 0 unless noted) ------------------------------------------
__global__ void relu_mask_mul_kernel(const floatx4* __restrict__ g, const floatx4* __restrict__ ref, floatx4* out, long n4) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) {
        floatx4 a = g[i], r = ref[i];
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = r[q] > 0.f ? a[q] : 0.f;
        out[i] = a;
    }
}

__global__ void add_inplace_kernel(floatx4* a, const floatx4* __restrict__ b, long n4) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    for (; i < n4; i += stride) a[i] += b[i];
}

// per-channel sums of dy (and dy*xhat when c != nullptr); dy = g * [c*ms+mh > 0] when ms != nullptr.
// generic C (scalar loads); block = 256 threads over pixels, thread handles all channels of a pixel subset
__global__ void bn_bwd_reduce_kernel(const float* __restrict__ g, const float* __restrict__ c, const float* mean,
                                     const float* invstd, const float* ms, const float* mh, long npix, int C,
                                     double* stats /* [NSLOT][2][C] */) {
    extern __shared__ float red[];   // [2][C]
    for (int k = threadIdx.x; k < 2 * C; k += blockDim.x) red[k] = 0.f;
    __syncthreads();
    // thread layout: channel = threadIdx.x % C (C <= 64 divides 256 or not: use modulo walk over flat index)
    long total = npix * C;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    // stride is a multiple of C when C divides blockDim.x*gridDim.x; otherwise accumulate per element via LDS atomics
    const bool fixed = (stride % C) == 0;
    float s = 0.f, q = 0.f;
    int ch = (int)(i % C);
    for (; i < total; i += stride) {
        if (!fixed) ch = (int)(i % C);
        float dy = g[i];
        float cv = c ? c[i] : 0.f;
        if (ms && !(cv * ms[ch] + mh[ch] > 0.f)) dy = 0.f;
        const float xh = c ? (cv - mean[ch]) * invstd[ch] : 0.f;
        if (fixed) {
            s += dy;
            q += dy * xh;
        } else {
            atomicAdd(&red[ch], dy);
            atomicAdd(&red[C + ch], dy * xh);
        }
    }
    if (fixed) {
        atomicAdd(&red[ch], s);
        atomicAdd(&red[C + ch], q);
    }
    __syncthreads();
    double* slot = stats + (long)(blockIdx.x % NSLOT) * 2 * C;
    for (int k = threadIdx.x; k < 2 * C; k += blockDim.x) atomicAdd(slot + k, (double)red[k]);
}

// per-channel constants of the BatchNorm backward: dgamma, dbeta and (coef, k1, k2) with
// dc = coef * (dy - k1 - xhat * k2);   training: coef = gamma*invstd, k1 = sum(dy)/N, k2 = sum(dy*xhat)/N
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(double* stats, int C, double count, const float* gamma, const float* invstd,
                                       float* dgamma, float* dbeta, float* coef, float* k1, float* k2, int clear) {
    __shared__ double red[256];
    double s = 0, q = 0;
    srbh::fold_stat_slots(stats, C, clear, red, s, q);   // (one block of 256 threads, C <= 64; clear: the slots are zeroed behind the read)
    const int c = threadIdx.x;
    if (c >= C) return;
    if (dgamma) dgamma[c] = (float)q;
    if (dbeta) dbeta[c] = (float)s;
    if (coef) {
        coef[c] = (gamma ? gamma[c] : 1.f) * invstd[c];
        k1[c] = (float)(s / count);
        k2[c] = (float)(q / count);
    }
}

__global__ void bn_bwd_apply_kernel(const float* __restrict__ g, const float* __restrict__ c, const float* mean,
                                    const float* invstd, const float* ms, const float* mh, const float* coef,
                                    const float* k1, const float* k2, float* out, long total, int C) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int ch = (int)(i % C);
        float dy = g[i];
        const float cv = c[i];
        if (ms && !(cv * ms[ch] + mh[ch] > 0.f)) dy = 0.f;
        const float xh = (cv - mean[ch]) * invstd[ch];
        out[i] = coef[ch] * (dy - k1[ch] - xh * k2[ch]);
    }
}

// ---- 16-byte forms of the two BatchNorm-backward passes (C % 4 == 0: every head layer).  A thread owns the 4-channel group
// (flat index % (C/4)) of the pixels it walks, its per-channel constants live in registers, partial sums are flushed once per thread.
// `relu_ref` / `dz_out` fold the block-closing ReLU backward into the reduce pass (dz = g where ref > 0; it is written for the
// skip / downsample branches and read back by the apply pass): one 3-tensor elementwise launch per BasicBlock less.
// 16-bit tensors in memory (round 3; template flags): GB = g holds bf16 elements, CH = c holds fp16 elements, OB = dz_out / out is written
// as bf16.  One 4-channel group is then an 8-byte access; the arithmetic stays fp32.
template <int B16, typename T>
__device__ __forceinline__ floatx4 ld4(const T* base, const long i) {
    if constexpr (B16 == 0) return ((const floatx4*)base)[i];
    else if constexpr (B16 == 1) return widen_h4(((const float2w*)base)[i]);
    else return widen_b4(((const float2w*)base)[i]);
}
// RB = 1: relu_ref is the ReLU's activity bit pattern written by bn_add_relu (4 words per 64 groups) instead of the fp32 block output
template <int GB, int CH, int OB, int RB = 0>
__global__ __launch_bounds__(256) void bn_bwd_reduce4_kernel(const void* __restrict__ g, const void* __restrict__ c, const float* mean,
                                                              const float* invstd, const float* ms, const float* mh,
                                                              const floatx4* __restrict__ relu_ref, void* __restrict__ dz_out,
                                                              long n4, int C, double* stats /* [NSLOT][2][C] */) {
    extern __shared__ float red[];   // [2][C]
    for (int k = threadIdx.x; k < 2 * C; k += blockDim.x) red[k] = 0.f;
    __syncthreads();
    const int C4 = C >> 2;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;          // host: a multiple of C/4 -> the channel group is fixed per thread
    const int ch = (int)(i % C4) * 4;
    floatx4 mn = {0.f, 0.f, 0.f, 0.f}, is = mn, msv = mn, mhv = mn;
    if (c) { mn = *(const floatx4*)(mean + ch); is = *(const floatx4*)(invstd + ch); }
    if (ms) { msv = *(const floatx4*)(ms + ch); mhv = *(const floatx4*)(mh + ch); }
    floatx4 s = {0.f, 0.f, 0.f, 0.f}, q = s;
    for (; i < n4; i += stride) {
        floatx4 dy = ld4<GB ? 2 : 0>(g, i);
        if (relu_ref) {
            if constexpr (RB != 0) {
                const unsigned long long* mw = (const unsigned long long*)relu_ref + (i >> 6) * 4;
                const int sh = (int)(i & 63);
#pragma unroll
                for (int j = 0; j < 4; ++j) dy[j] = ((mw[j] >> sh) & 1ull) ? dy[j] : 0.f;
            } else {
                const floatx4 r = relu_ref[i];
#pragma unroll
                for (int j = 0; j < 4; ++j) dy[j] = r[j] > 0.f ? dy[j] : 0.f;
            }
            if (dz_out) {
                if constexpr (OB != 0) {
                    const float2w nb = narrow_b4(dy);
                    ((float2w*)dz_out)[i] = nb;
                    dy = widen_b4(nb);          // the sums are taken over the values the consumers will read
                } else {
                    ((floatx4*)dz_out)[i] = dy;
                }
            }
        }
        if (c) {
            const floatx4 cv = ld4<CH ? 1 : 0>(c, i);
            if (ms) {
#pragma unroll
                for (int j = 0; j < 4; ++j) dy[j] = cv[j] * msv[j] + mhv[j] > 0.f ? dy[j] : 0.f;
            }
            q += dy * ((cv - mn) * is);
        }
        s += dy;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        atomicAdd(&red[ch + j], s[j]);
        atomicAdd(&red[C + ch + j], q[j]);
    }
    __syncthreads();
    double* slot = stats + (long)(blockIdx.x % NSLOT) * 2 * C;
    for (int k = threadIdx.x; k < 2 * C; k += blockDim.x) atomicAdd(slot + k, (double)red[k]);
}

template <int GB, int CH, int OB>
__global__ __launch_bounds__(256) void bn_bwd_apply4_kernel(const void* __restrict__ g, const void* __restrict__ c, const float* mean,
                                                             const float* invstd, const float* ms, const float* mh, const float* coef,
                                                             const float* k1, const float* k2, void* __restrict__ out, long n4, int C) {
    const int C4 = C >> 2;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    const int ch = (int)(i % C4) * 4;
    const floatx4 mn = *(const floatx4*)(mean + ch), is = *(const floatx4*)(invstd + ch), cf = *(const floatx4*)(coef + ch),
                  a1 = *(const floatx4*)(k1 + ch), a2 = *(const floatx4*)(k2 + ch);
    floatx4 msv = {0.f, 0.f, 0.f, 0.f}, mhv = msv;
    if (ms) { msv = *(const floatx4*)(ms + ch); mhv = *(const floatx4*)(mh + ch); }
    for (; i < n4; i += stride) {
        floatx4 dy = ld4<GB ? 2 : 0>(g, i);
        const floatx4 cv = ld4<CH ? 1 : 0>(c, i);
        if (ms) {
#pragma unroll
            for (int j = 0; j < 4; ++j) dy[j] = cv[j] * msv[j] + mhv[j] > 0.f ? dy[j] : 0.f;
        }
        const floatx4 r = cf * (dy - a1 - ((cv - mn) * is) * a2);
        if constexpr (OB != 0) ((float2w*)out)[i] = narrow_b4(r);
        else ((floatx4*)out)[i] = r;
    }
}

// grid for the 16-byte forms: a multiple of C/4 threads in total (blockDim 256, C/4 in {1, 2, 4, 8, 16} divides it)
int grid4_for(long n4) {
    long b = (n4 + 255) / 256;
    return (int)(b < 2048 ? (b > 0 ? b : 1) : 2048);
}

// PixelShuffle(2) inverse: g_ps [B][2H][2W][C] -> g [B][H][W][4C], channel 4c + 2i + j <- (2y+i, 2x+j, c)
__global__ void ps2_inverse_kernel(const float* __restrict__ gps, float* __restrict__ g, int B, int H, int W, int C) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)B * H * W * 4 * C;
    if (idx >= total) return;
    int oc = idx % (4 * C);
    long r = idx / (4 * C);
    int x = r % W; r /= W;
    int y = r % H;
    int b = r / H;
    int cc = oc >> 2, i = (oc >> 1) & 1, j = oc & 1;
    g[idx] = gps[(((long)b * 2 * H + 2 * y + i) * (2 * W) + 2 * x + j) * C + cc];
}

// the same permutation, 16 elements per thread (C % 4 == 0): a thread takes 4 channels of the 4 sub-pixels of one output pixel as four
// 16-byte loads, transposes 4x4 in registers and writes 16 consecutive output channels as four 16-byte stores (the one-element form
// spends ~30 integer instructions per 4 bytes: 226 us for 537 MB)
__global__ __launch_bounds__(256) void ps2_inverse4_kernel(const float* __restrict__ gps, float* __restrict__ g, long total, int H, int W, int C) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int C4 = C >> 2;
    const int cq = (int)(idx % C4);
    long r = idx / C4;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const long b = r / H;
    floatx4 v[4];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
        v[sub] = *(const floatx4*)(gps + (((b * 2 * H + 2 * y + (sub >> 1)) * (2 * W) + 2 * x + (sub & 1)) * C + 4 * cq));
    float* o = g + (((b * H + y) * W + x) * 4L * C + 16 * cq);
#pragma unroll
    for (int k = 0; k < 4; ++k) *(floatx4*)(o + 4 * k) = floatx4{v[0][k], v[1][k], v[2][k], v[3][k]};
}

int grid_for(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 2048 ? (b > 0 ? b : 1) : 2048);
}

}  // namespace

namespace {
int wgrad_impl(const srbh_hwgrad_args* a, void* stream, bool b16, const char* who) {
    SRBH_REQUIRE(a && a->src0 && a->dy && a->dw, "%s: null pointer", who);
    SRBH_REQUIRE(a->c0 > 0 && a->c1 >= 0 && (a->c1 == 0 || a->src1), "srbh_hconv_wgrad: bad channel split");
    SRBH_REQUIRE(a->ksize == 3 || a->ksize == 1, "srbh_hconv_wgrad: ksize must be 1 or 3");
    SRBH_REQUIRE(a->cout >= 1 && a->cout <= 64 && a->B > 0 && a->H > 0 && a->W > 0, "srbh_hconv_wgrad: bad shape");
    WGParams p;
    p.src0 = a->src0; p.src1 = a->src1; p.c0 = a->c0; p.c1 = a->c1;
    p.ld0 = a->src0_ld > 0 ? a->src0_ld : a->c0;
    p.ld1 = a->src1_ld > 0 ? a->src1_ld : a->c1;
    p.pre_scale = a->pre_scale; p.pre_shift = a->pre_shift; p.pre_relu = a->pre_relu;
    p.dy = a->dy; p.cout_total = a->cout; p.dw = a->dw; p.ws = a->ws;
    p.io = a->io;
    p.zchunk = 0;
    SRBH_REQUIRE(a->ws, "srbh_hconv_wgrad: workspace missing (srbh_hwgrad_ws_bytes)");
    SRBH_REQUIRE((a->io & ~(SRBH_WG_SRC0_H16 | SRBH_WG_DY_B16)) == 0, "srbh_hconv_wgrad: unknown io bits");
    SRBH_REQUIRE(!a->io || b16, "srbh_hconv_wgrad_f32: 16-bit tensors in memory need the bf16-operand form (srbh_hconv_wgrad_b16)");
    const bool xs16 = (a->io & SRBH_WG_SRC0_H16) != 0, ds16 = (a->io & SRBH_WG_DY_B16) != 0;
    p.B = a->B; p.H = a->H; p.W = a->W;
    p.tiles_x = (a->W + HT_W - 1) / HT_W;
    p.tiles_per_img = p.tiles_x * ((a->H + HT_H - 1) / HT_H);
    p.ntiles = p.tiles_per_img * a->B;
    hipStream_t st = (hipStream_t)stream;
    const int cin = a->c0 + a->c1;
    const int nob = (a->cout + 15) / 16;
    int gx = p.ntiles < 512 ? (p.ntiles + 7) / 8 * 8 : 512;                // (a multiple of 8: the same number of workgroups per XCD)
    p.tiles_per_xcd = (p.ntiles + 7) / 8;
    // the bf16 form moves whole 4-channel groups with 16-byte loads and whole 16-channel output blocks; the few layers outside
    // that (the 1- and 7-channel output convs) keep the fp32 kernel
    const bool can16 = (a->c0 & 3) == 0 && (a->c1 & 3) == 0 && (p.ld0 & 3) == 0 && (a->c1 == 0 || (p.ld1 & 3) == 0) && (a->cout & 15) == 0 &&
                       ((uintptr_t)a->src0 & 15) == 0 && ((uintptr_t)a->src1 & 15) == 0 && ((uintptr_t)a->dy & (ds16 ? 7 : 15)) == 0;
    // the dominant layer shape has its own double-buffered kernel (srbh_hwgrad16_kernel.h)
    static const int k16_wgs = getenv("SRBH_HWGRAD16_WGS") ? atoi(getenv("SRBH_HWGRAD16_WGS")) : 768;   // 0 = never
    const bool narrow = a->cout < 16 && !ds16;          // conv_last (1 / 7 output channels): zero-padded to one 16-channel block while staged
    const bool k16 = b16 && k16_wgs >= 8 && k16_wgs <= WS_SLOTS && a->ksize == 3 && a->c0 == 16 && a->c1 == 0 && (a->cout == 16 || narrow) &&
                     (a->W & 63) == 0 && (a->H & 3) == 0 && (p.ld0 & 3) == 0 && ((uintptr_t)a->src0 & (xs16 ? 7 : 15)) == 0 &&
                     ((uintptr_t)a->dy & (narrow ? 3 : ds16 ? 7 : 15)) == 0;
    SRBH_REQUIRE(!xs16 || k16 || (b16 && can16 && ((uintptr_t)a->src0 & 7) == 0),
                 "srbh_hconv_wgrad_b16: an fp16 source tensor needs the bf16-operand forms (4-aligned channels, 16-channel output blocks)");
    SRBH_REQUIRE(!ds16 || k16 || can16, "srbh_hconv_wgrad_b16: a bf16 dY needs 4-aligned channels / 16-channel output blocks");
    count_path(k16 ? PATH_WGRAD16 : (b16 && can16) ? PATH_WGRAD_B16_GENERIC : PATH_WGRAD_F32);
    static const int ob_inner = getenv("SRBH_WGRAD_OB_INNER") ? atoi(getenv("SRBH_WGRAD_OB_INNER")) : 1;      // 0: grid.y = output blocks (A/B aid)
    if (k16) {
        p.tiles_x = a->W / 64;
        p.tiles_per_img = p.tiles_x * (a->H / 4);
        p.ntiles = p.tiles_per_img * a->B;
        p.tiles_per_xcd = (p.ntiles + 7) / 8;
        const int per_xcd = p.tiles_per_xcd < k16_wgs / 8 ? p.tiles_per_xcd : k16_wgs / 8;
        gx = per_xcd * 8;
#define SRBH_WG16(X_, D_)                                                                                                              \
    do {                                                                                                                          \
        SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)hwgrad16_kernel<X_, D_>, hipFuncAttributeMaxDynamicSharedMemorySize, WG16T::LDS_B))); \
        hipLaunchKernelGGL((hwgrad16_kernel<X_, D_>), dim3(gx), dim3(256), WG16T::LDS_B, st, p);                                   \
    } while (0)
        if (narrow && xs16) SRBH_WG16(1, 2);
        else if (narrow) SRBH_WG16(0, 2);
        else if (xs16 && ds16) SRBH_WG16(1, 1);
        else if (xs16) SRBH_WG16(1, 0);
        else if (ds16) SRBH_WG16(0, 1);
        else SRBH_WG16(0, 0);
#undef SRBH_WG16
    } else
    if (b16 && can16 && a->ksize == 3 && a->c1 == 0 && cin <= 16 && nob == 4 && ob_inner) {
        // one input chunk, four output blocks (the Upsampler's 16 -> 64 convs): X staged once per tile, the dY blocks inside the walk
        if (ds16) {
            SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)hwgrad_ob_b16_kernel<1, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, WG16<3>::LDS_B)));
            hipLaunchKernelGGL((hwgrad_ob_b16_kernel<1, 4>), dim3(gx), dim3(256), WG16<3>::LDS_B, st, p);
        } else {
            SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)hwgrad_ob_b16_kernel<0, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, WG16<3>::LDS_B)));
            hipLaunchKernelGGL((hwgrad_ob_b16_kernel<0, 4>), dim3(gx), dim3(256), WG16<3>::LDS_B, st, p);
        }
    } else
    if (b16 && can16) {
#define SRBH_WGB(K_, D_)                                                                                                               \
    do {                                                                                                                          \
        SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)hwgrad_b16_kernel<K_, D_>, hipFuncAttributeMaxDynamicSharedMemorySize, WG16<K_>::LDS_B))); \
        hipLaunchKernelGGL((hwgrad_b16_kernel<K_, D_>), dim3(gx, nob), dim3(256), WG16<K_>::LDS_B, st, p);                          \
    } while (0)
        if (a->ksize == 3) {
            if (ds16) SRBH_WGB(3, 1); else SRBH_WGB(3, 0);
        } else {
            if (ds16) SRBH_WGB(1, 1); else SRBH_WGB(1, 0);
        }
#undef SRBH_WGB
    } else if (a->ksize == 3) {
        constexpr int LDS_B = (10 * 66 * 16 + 4 * 9 * 256) * 4;   // X tile + max(dY tile, flush buffer)
        SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)hwgrad_f32_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B)));
        hipLaunchKernelGGL(hwgrad_f32_kernel<3>, dim3(gx, nob), dim3(256), LDS_B, st, p);
    } else {
        constexpr int LDS_B = (8 * 64 * 16 + 8 * 64 * 16) * 4;
        SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)hwgrad_f32_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B)));
        hipLaunchKernelGGL(hwgrad_f32_kernel<1>, dim3(gx, nob), dim3(256), LDS_B, st, p);
    }
    SRBH_HIP(hipGetLastError());
    const int taps = a->ksize * a->ksize, nchunk = (cin + 15) / 16;
    const long U = (long)nob * nchunk * taps * 256;
    return reduce_partials(a->ws, a->dw, U, gx, nchunk, taps, a->cout, cin, st);
}
}  // namespace

/* Weight gradients of a BasicBlock ENTRY (SR/HRfuse.py:142-159: conv1 = 3x3 and downsample[0] = 1x1 over the same input) in one pass
 * over that input: a3 = the 3x3 call's arguments, a1 = the 1x1 call's (same sources, shape, cout, element types; its own dy / dw / ws).
 * bf16 operands.  Shapes the fused kernel does not take run as the two separate calls -- same results either way. */
extern "C" int srbh_hconv_wgrad_entry_b16(const srbh_hwgrad_args* a3, const srbh_hwgrad_args* a1, void* stream) {
    SRBH_REQUIRE(a3 && a1, "srbh_hconv_wgrad_entry_b16: null arguments");
    static const int fuse = getenv("SRBH_WGRAD_ENTRY_FUSE") ? atoi(getenv("SRBH_WGRAD_ENTRY_FUSE")) : 1;
    const int ld0 = a3->src0_ld > 0 ? a3->src0_ld : a3->c0, ld1 = a3->src1_ld > 0 ? a3->src1_ld : a3->c1;
    const int dld0 = a1->src0_ld > 0 ? a1->src0_ld : a1->c0, dld1 = a1->src1_ld > 0 ? a1->src1_ld : a1->c1;
    const bool ds16 = (a3->io & SRBH_WG_DY_B16) != 0;
    const bool same = a3->src0 == a1->src0 && a3->src1 == a1->src1 && a3->c0 == a1->c0 && a3->c1 == a1->c1 && ld0 == dld0 && ld1 == dld1 &&
                      a3->B == a1->B && a3->H == a1->H && a3->W == a1->W && a3->cout == a1->cout && a3->io == a1->io &&
                      a3->pre_scale == a1->pre_scale && a3->pre_shift == a1->pre_shift && a3->pre_relu == a1->pre_relu;
    const bool shape = a3->ksize == 3 && a1->ksize == 1 && a3->src0 && a3->dy && a1->dy && a3->dw && a1->dw && a3->ws && a1->ws &&
                       a3->c0 > 0 && (a3->c0 & 3) == 0 && (a3->c1 & 3) == 0 && (a3->c1 == 0 || a3->src1) && (ld0 & 3) == 0 && (a3->c1 == 0 || (ld1 & 3) == 0) &&
                       a3->cout > 0 && a3->cout <= 64 && (a3->cout & 15) == 0 && a3->B > 0 && a3->H > 0 && a3->W > 0 &&
                       !(a3->c0 == 16 && a3->c1 == 0) &&                /* (the 16 -> 16 layer has its own kernel) */
                       (a3->io & ~(SRBH_WG_SRC0_H16 | SRBH_WG_DY_B16)) == 0 &&
                       ((uintptr_t)a3->src0 & ((a3->io & SRBH_WG_SRC0_H16) ? 7 : 15)) == 0 && ((uintptr_t)a3->src1 & 15) == 0 &&
                       (((uintptr_t)a3->dy | (uintptr_t)a1->dy) & (ds16 ? 7 : 15)) == 0;
    if (!(fuse && same && shape)) {
        count_path(PATH_WGRAD_ENTRY_SPLIT);
        if (int rc = wgrad_impl(a3, stream, true, "srbh_hconv_wgrad_b16")) return rc;
        return wgrad_impl(a1, stream, true, "srbh_hconv_wgrad_b16");
    }
    count_path(PATH_WGRAD_ENTRY_FUSED);
    WGParams p = {};
    p.src0 = a3->src0; p.src1 = a3->src1; p.c0 = a3->c0; p.c1 = a3->c1; p.ld0 = ld0; p.ld1 = ld1;
    p.pre_scale = a3->pre_scale; p.pre_shift = a3->pre_shift; p.pre_relu = a3->pre_relu;
    p.dy = a3->dy; p.cout_total = a3->cout; p.dw = a3->dw; p.ws = a3->ws; p.io = a3->io; p.zchunk = 0;
    p.dy2 = a1->dy; p.ws2 = a1->ws;
    p.B = a3->B; p.H = a3->H; p.W = a3->W;
    p.tiles_x = (a3->W + HT_W - 1) / HT_W;
    p.tiles_per_img = p.tiles_x * ((a3->H + HT_H - 1) / HT_H);
    p.ntiles = p.tiles_per_img * a3->B;
    p.tiles_per_xcd = (p.ntiles + 7) / 8;
    hipStream_t st = (hipStream_t)stream;
    const int cin = a3->c0 + a3->c1, nob = a3->cout / 16, nchunk = (cin + 15) / 16;
    const int gx = p.ntiles < 512 ? (p.ntiles + 7) / 8 * 8 : 512;
    // (fuse == 2: the chunk-outer kernel also where the chunk-inner one applies -- same-box A/B aid)
#define SRBH_ENTRY_INNER(DS_, NC_)                                                                                                              \
    do {                                                                                                                                       \
        SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)hwgrad_entry_b16_kernel<DS_, NC_>, hipFuncAttributeMaxDynamicSharedMemorySize, WG16<3>::LDS_B2))); \
        hipLaunchKernelGGL((hwgrad_entry_b16_kernel<DS_, NC_>), dim3(gx, nob), dim3(256), WG16<3>::LDS_B2, st, p);                                \
    } while (0)
    // (fuse == 3: without the whole-row 64-channel kernel -- A/B aid)
    if ((fuse == 1) && a3->c0 == 64 && a3->c1 == 0 && ld0 == 64 && (a3->io & SRBH_WG_SRC0_H16) && !a3->pre_scale && !a3->pre_relu &&
        ((uintptr_t)a3->src0 & 15) == 0) {
        if (ds16) {
            SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)hwgrad_entry64_b16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, WG64::LDS_B)));
            hipLaunchKernelGGL((hwgrad_entry64_b16_kernel<1>), dim3(gx, nob), dim3(256), WG64::LDS_B, st, p);
        } else {
            SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)hwgrad_entry64_b16_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, WG64::LDS_B)));
            hipLaunchKernelGGL((hwgrad_entry64_b16_kernel<0>), dim3(gx, nob), dim3(256), WG64::LDS_B, st, p);
        }
    } else if ((fuse == 1 || fuse == 3) && (nchunk == 2 || nchunk == 4)) {
        if (ds16) { if (nchunk == 4) SRBH_ENTRY_INNER(1, 4); else SRBH_ENTRY_INNER(1, 2); }
        else { if (nchunk == 4) SRBH_ENTRY_INNER(0, 4); else SRBH_ENTRY_INNER(0, 2); }
    } else if (ds16) {
        SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)hwgrad_b16_kernel<3, 1, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, WG16<3>::LDS_B2)));
        hipLaunchKernelGGL((hwgrad_b16_kernel<3, 1, 0, 1>), dim3(gx, nob), dim3(256), WG16<3>::LDS_B2, st, p);
    } else {
        SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)hwgrad_b16_kernel<3, 0, 0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, WG16<3>::LDS_B2)));
        hipLaunchKernelGGL((hwgrad_b16_kernel<3, 0, 0, 1>), dim3(gx, nob), dim3(256), WG16<3>::LDS_B2, st, p);
    }
    SRBH_HIP(hipGetLastError());
    for (int k = 0; k < 2; ++k) {           // the two ordered reduces (3x3, then 1x1): as behind the separate calls
        const int taps = k == 0 ? 9 : 1;
        const srbh_hwgrad_args* a = k == 0 ? a3 : a1;
        const long U = (long)nob * nchunk * taps * 256;
        if (int rc = reduce_partials(a->ws, a->dw, U, gx, nchunk, taps, a->cout, cin, st)) return rc;
    }
    return SRBH_OK;
}

extern "C" int srbh_hconv_wgrad_f32(const srbh_hwgrad_args* a, void* stream) { return wgrad_impl(a, stream, false, "srbh_hconv_wgrad_f32"); }

extern "C" int srbh_hconv_wgrad_b16(const srbh_hwgrad_args* a, void* stream) { return wgrad_impl(a, stream, true, "srbh_hconv_wgrad_b16"); }

/* Weight gradient of a 3x3 conv of the RRDBNet training path on ACT16 tensors: x = fp16 chunk planes (the saved dense buffer: channels
 * 0 .. cin-1, cin % 16 == 0), dy = bf16 chunk planes (channels dy_ch0 .. dy_ch0 + cout - 1 of a dy_chunks_total-plane buffer, cout % 16 == 0);
 * dw OIHW fp32 [cout][cin][3][3]; ws as srbh_hwgrad_ws_bytes(cout, cin, 3).  bf16 operands (x rounded while staged), fp32 accumulate,
 * fixed summation order. */
extern "C" int srbh_act16_wgrad_b16(const void* x, int x_chunks_total, int cin, const void* dy, int dy_chunks_total, int dy_ch0, int cout,
                                    int B, int H, int W, float* dw, float* ws, void* stream) {
    SRBH_REQUIRE(x && dy && dw && ws && B > 0 && H > 0 && W > 0, "srbh_act16_wgrad_b16: bad arguments");
    SRBH_REQUIRE(cin > 0 && (cin & 15) == 0 && cin <= x_chunks_total * 32 && cout > 0 && (cout & 15) == 0 && cout <= 64 && (dy_ch0 & 15) == 0 &&
                 dy_ch0 + cout <= dy_chunks_total * 32, "srbh_act16_wgrad_b16: channel ranges (multiples of 16, inside the buffers)");
    WGParams p = {};
    p.src0 = (const float*)x; p.src1 = nullptr; p.c0 = cin; p.c1 = 0; p.ld0 = 0; p.ld1 = 0;
    p.dy = (const float*)dy; p.cout_total = cout; p.dw = dw; p.ws = ws; p.io = 0;
    const Act16Geo gx_ = act16_geo(B, x_chunks_total, H, W), gd = act16_geo(B, dy_chunks_total, H, W);
    p.x_img_b = gx_.img_b; p.x_plane_b = gx_.plane_b; p.x_row_b = gx_.row_b;
    p.dy_img_b = gd.img_b; p.dy_plane_b = gd.plane_b; p.dy_row_b = gd.row_b; p.dy_ch0 = dy_ch0;
    p.B = B; p.H = H; p.W = W;
    p.tiles_x = (W + HT_W - 1) / HT_W;
    p.tiles_per_img = p.tiles_x * ((H + HT_H - 1) / HT_H);
    p.ntiles = p.tiles_per_img * B;
    p.tiles_per_xcd = (p.ntiles + 7) / 8;
    hipStream_t st = (hipStream_t)stream;
    const int nob = cout / 16;
    int gx = p.ntiles < 512 ? (p.ntiles + 7) / 8 * 8 : 512;
    SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)hwgrad_b16_kernel<3, 1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, WG16<3>::LDS_B)));
    static const int zchunk_max = getenv("SRBH_WG_ZCHUNK_MAX") ? atoi(getenv("SRBH_WG_ZCHUNK_MAX")) : 256;
    p.zchunk = p.ntiles <= zchunk_max ? 1 : 0;
    if (p.zchunk) {
        // Every workgroup column (blockIdx.x) owns a full set of partial sums: with one column per tile the reduce reads gx * |dW| floats
        // -- 56 MB for the 192 -> 64 conv at batch 8, 13 us per call on the stream that bounds the backward.  The grid already has
        // nob * cin/16 workgroups per column, so columns walk several tiles each as long as ~768 workgroups remain (swept at batch 8: 22.5 ms per generator step with one column per tile, 22.1 / 20.7 / 21.1 / 22.2 at 1 536 / 768 / 512 / 256).
        static const int want_wgs = getenv("SRBH_WG_COLUMNS_WGS") ? atoi(getenv("SRBH_WG_COLUMNS_WGS")) : 768;
        const int per_col = nob * (cin / 16);
        int cols = ((want_wgs + per_col - 1) / per_col + 7) / 8 * 8;
        cols = cols < 8 ? 8 : cols;
        if (cols < gx) gx = cols;
    }
    hipLaunchKernelGGL((hwgrad_b16_kernel<3, 1, 1>), dim3(gx, nob, p.zchunk ? cin / 16 : 1), dim3(256), WG16<3>::LDS_B, st, p);
    SRBH_HIP(hipGetLastError());
    constexpr int SLICES = 16;
    const int taps = 9, nchunk = cin / 16;
    const long U = (long)nob * nchunk * taps * 256;
    const int total = cout * cin * taps;
    if (g_red_defer)          // (srbh_hwgrad_defer: the two-stage ordered reduce, queued -- srbh_rrdbnet_trunk_train_backward batches a dense block's five)
        return reduce_partials(ws, dw, U, gx, nchunk, taps, cout, cin, st);
    if (gx <= 128) {
        hipLaunchKernelGGL(hwgrad_reduce_direct_kernel, dim3((total + 255) / 256), dim3(256), 0, st, ws, dw, U, gx, nchunk, taps, cout, cin);
        SRBH_HIP(hipGetLastError());
        return SRBH_OK;
    }
    float* tmp = ws + (long)WS_SLOTS * U;
    const int per = (gx + SLICES - 1) / SLICES;
    hipLaunchKernelGGL(hwgrad_reduce1_kernel, dim3((unsigned)((U + 255) / 256), SLICES), dim3(256), 0, st, ws, tmp, U, gx, per);
    SRBH_HIP(hipGetLastError());
    hipLaunchKernelGGL(hwgrad_reduce2_kernel, dim3((total + 255) / 256), dim3(256), 0, st, tmp, dw, U, SLICES, nchunk, taps, cout, cin);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_hbwd16_supported(int H, int W) { return H > 0 && W > 0 && (W & 63) == 0 && (H & 3) == 0; }

/* One pass for the backward of a 3x3, 16 -> 16 conv behind its BatchNorm (srbh_hbwd16_kernel.h): dc never reaches memory. */
extern "C" int srbh_hbwd16(const srbh_hbwd16_args* a, void* stream) {
    SRBH_REQUIRE(a && a->g && a->c && a->mean && a->invstd && a->coef && a->k1 && a->k2 && a->x && a->w && a->dx && a->dw && a->ws,
                 "srbh_hbwd16: null pointer");
    SRBH_REQUIRE(a->B > 0 && srbh_hbwd16_supported(a->H, a->W), "srbh_hbwd16: W %% 64 == 0 and H %% 4 == 0 (srbh_hbwd16_supported)");
    SRBH_REQUIRE((a->mask_scale == nullptr) == (a->mask_shift == nullptr) && (a->pre_scale == nullptr) == (a->pre_shift == nullptr),
                 "srbh_hbwd16: scale / shift come in pairs");
    SRBH_REQUIRE(!a->stats || (a->bstat_c && a->bstat_mean && a->bstat_invstd && (a->relu_bits || !a->res) && (a->bstat_ms == nullptr) == (a->bstat_mh == nullptr)),
                 "srbh_hbwd16: the statistics epilogue needs bstat_c / mean / invstd and takes no skip gradient (unless relu_bits)");
    SRBH_REQUIRE(!a->relu_bits || (a->stats && a->dx_b16 && !a->bstat_ms && ((uintptr_t)a->relu_bits & 7) == 0),
                 "srbh_hbwd16: relu_bits needs stats + bstat_c / mean / invstd of the previous block's bn2, a bf16 dx and no bstat mask");
    SRBH_REQUIRE((((uintptr_t)a->g | (uintptr_t)a->res) & 7) == 0 && (((uintptr_t)a->c | (uintptr_t)a->x | (uintptr_t)a->bstat_c | (uintptr_t)a->w) & 15) == 0 &&
                 ((uintptr_t)a->dx & (a->dx_b16 ? 7 : 15)) == 0, "srbh_hbwd16: misaligned tensor");
    hipStream_t st = (hipStream_t)stream;
    HBParams p;
    p.g = a->g; p.c = a->c; p.mean = a->mean; p.invstd = a->invstd; p.coef = a->coef; p.k1 = a->k1; p.k2 = a->k2;
    p.ms = a->mask_scale; p.mh = a->mask_shift;
    p.x = a->x; p.pre_scale = a->pre_scale; p.pre_shift = a->pre_shift; p.pre_relu = a->pre_relu;
    p.w = a->w; p.dx = a->dx; p.dx_b16 = a->dx_b16; p.res = a->res;
    p.bstat_c = a->bstat_c; p.bstat_mean = a->bstat_mean; p.bstat_invstd = a->bstat_invstd; p.bstat_ms = a->bstat_ms; p.bstat_mh = a->bstat_mh;
    p.stats = a->stats; p.ws = a->ws; p.relu_bits = (const unsigned long long*)a->relu_bits;
    p.B = a->B; p.H = a->H; p.W = a->W;
    p.tiles_x = a->W / 64;
    p.tiles_per_img = p.tiles_x * (a->H / 4);
    p.ntiles = p.tiles_per_img * a->B;
    p.tiles_per_xcd = (p.ntiles + 7) / 8;
    static const int wgs = getenv("SRBH_HBWD16_WGS") ? atoi(getenv("SRBH_HBWD16_WGS")) : 512;
    SRBH_REQUIRE(wgs >= 8 && wgs <= WS_SLOTS, "SRBH_HBWD16_WGS must be 8 .. %d", WS_SLOTS);
    const int per_xcd = p.tiles_per_xcd < wgs / 8 ? p.tiles_per_xcd : wgs / 8;
    const int gx = per_xcd * 8;
    if (a->stats && !a->stats_clean) { if (int rc = zero_async(a->stats, (size_t)NSLOT * 2 * 16 * sizeof(double), st)) return rc; }
#define SRBH_HB(B_, M_)                                                                                                            \
    do {                                                                                                                      \
        SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)hbwd16_kernel<B_, M_>, hipFuncAttributeMaxDynamicSharedMemorySize, HB16::LDS_B))); \
        hipLaunchKernelGGL((hbwd16_kernel<B_, M_>), dim3(gx), dim3(256), HB16::LDS_B, st, p);                                   \
    } while (0)
    if (a->relu_bits && a->mask_scale) SRBH_HB(2, 1);
    else if (a->relu_bits) SRBH_HB(2, 0);
    else if (a->stats && a->mask_scale) SRBH_HB(1, 1);
    else if (a->stats) SRBH_HB(1, 0);
    else if (a->mask_scale) SRBH_HB(0, 1);
    else SRBH_HB(0, 0);
#undef SRBH_HB
    SRBH_HIP(hipGetLastError());
    count_path(PATH_HBWD16);
    // the workgroups' weight-gradient partials -> dW (the ordered two-stage reduce of srbh_hconv_wgrad_b16; queued under srbh_hwgrad_defer)
    return reduce_partials(a->ws, a->dw, 9 * 256, gx, 1, 9, 16, 16, st);
}

/* Between srbh_hwgrad_defer(1) and srbh_hwgrad_flush the weight-gradient entry points above (srbh_hconv_wgrad_f32 / _b16 / _entry_b16,
 * srbh_hbwd16) run their kernels but QUEUE the ordered reduce of their partial sums; srbh_hwgrad_flush does every queued reduce in one pair
 * of launches and ends the deferral.  The caller keeps every `ws` (and `dw`) alive until the flush.  Per host thread. */
extern "C" int srbh_hwgrad_defer(int on) {
    SRBH_REQUIRE(on || g_red_queue.empty(), "srbh_hwgrad_defer(0) with queued reduce jobs: call srbh_hwgrad_flush");
    g_red_defer = on != 0;
    return SRBH_OK;
}
extern "C" int srbh_hwgrad_flush(void* stream) {
    g_red_defer = false;
    return reduce_flush((hipStream_t)stream);
}

extern "C" size_t srbh_hwgrad_ws_bytes(int cout, int cin, int ksize) {
    if (cout <= 0 || cin <= 0 || (ksize != 1 && ksize != 3)) return 0;
    return (size_t)(WS_SLOTS + 16) * ((cout + 15) / 16) * ((cin + 15) / 16) * ksize * ksize * 256 * sizeof(float);
}

extern "C" int srbh_relu_mask_mul(const float* g, const float* ref, float* out, long n, void* stream) {
    SRBH_REQUIRE(g && ref && out && n > 0 && n % 4 == 0, "srbh_relu_mask_mul: bad arguments");
    hipLaunchKernelGGL(relu_mask_mul_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const floatx4*)g,
                       (const floatx4*)ref, (floatx4*)out, n / 4);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_add_inplace(float* a, const float* b, long n, void* stream) {
    SRBH_REQUIRE(a && b && n > 0 && n % 4 == 0, "srbh_add_inplace: bad arguments");
    hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (floatx4*)a,
                       (const floatx4*)b, n / 4);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

static int bn_bwd_reduce_impl(const void* g, const float* relu_ref, void* dz_out, const void* c, const float* mean, const float* invstd,
                              const float* mask_scale, const float* mask_shift, long npix, int C, double* stats, void* stream, int io = 0) {
    hipStream_t st = (hipStream_t)stream;
    if (!(io & SRBH_BN_STATS_CLEAN)) { if (int rc = zero_async(stats, (size_t)NSLOT * 2 * C * sizeof(double), st)) return rc; }
    io &= ~SRBH_BN_STATS_CLEAN;
    const bool gb = (io & SRBH_BN_G_B16) != 0, ch = (io & SRBH_BN_C_H16) != 0, ob = (io & SRBH_BN_OUT_B16) != 0, rb = (io & SRBH_BN_REF_BITS) != 0;
    const bool v4 = (C & 3) == 0 && (256 % (C >> 2)) == 0 && ((uintptr_t)g & (gb ? 7 : 15)) == 0 && ((uintptr_t)c & (ch ? 7 : 15)) == 0 &&
                    ((uintptr_t)dz_out & (ob ? 7 : 15)) == 0 && ((uintptr_t)relu_ref & (rb ? 7 : 15)) == 0 && (((uintptr_t)mean | (uintptr_t)invstd |
                     (uintptr_t)mask_scale | (uintptr_t)mask_shift) & 15) == 0;
    SRBH_REQUIRE(!rb || (relu_ref && v4), "srbh_bn_bwd_reduce_io: SRBH_BN_REF_BITS needs relu_ref (the bit buffer) and the vector form");
    SRBH_REQUIRE(!io || v4, "srbh_bn_bwd_reduce: 16-bit tensors need the vector form (C %% 4 == 0, 256 %% (C/4) == 0, aligned)");
    if (v4) {
        const long n4 = npix * (C >> 2);
#define SRBH_BNR(G_, C_, O_) hipLaunchKernelGGL((bn_bwd_reduce4_kernel<G_, C_, O_>), dim3(grid4_for(n4)), dim3(256), 2 * C * sizeof(float), st, \
                                                g, c, mean, invstd, mask_scale, mask_shift, (const floatx4*)relu_ref, dz_out, n4, C, stats)
        if (rb) {
#define SRBH_BNRB(G_, C_, O_) hipLaunchKernelGGL((bn_bwd_reduce4_kernel<G_, C_, O_, 1>), dim3(grid4_for(n4)), dim3(256), 2 * C * sizeof(float), st, \
                                                 g, c, mean, invstd, mask_scale, mask_shift, (const floatx4*)relu_ref, dz_out, n4, C, stats)
            switch ((gb ? 4 : 0) | (ch ? 2 : 0) | (ob ? 1 : 0)) {
                case 0: SRBH_BNRB(0, 0, 0); break; case 1: SRBH_BNRB(0, 0, 1); break; case 2: SRBH_BNRB(0, 1, 0); break; case 3: SRBH_BNRB(0, 1, 1); break;
                case 4: SRBH_BNRB(1, 0, 0); break; case 5: SRBH_BNRB(1, 0, 1); break; case 6: SRBH_BNRB(1, 1, 0); break; default: SRBH_BNRB(1, 1, 1); break;
            }
#undef SRBH_BNRB
        } else
        switch ((gb ? 4 : 0) | (ch ? 2 : 0) | (ob ? 1 : 0)) {
            case 0: SRBH_BNR(0, 0, 0); break; case 1: SRBH_BNR(0, 0, 1); break; case 2: SRBH_BNR(0, 1, 0); break; case 3: SRBH_BNR(0, 1, 1); break;
            case 4: SRBH_BNR(1, 0, 0); break; case 5: SRBH_BNR(1, 0, 1); break; case 6: SRBH_BNR(1, 1, 0); break; default: SRBH_BNR(1, 1, 1); break;
        }
#undef SRBH_BNR
    } else {
        SRBH_REQUIRE(!relu_ref, "srbh_bn_bwd_reduce_relu: needs the 16-byte form (C %% 4 == 0, aligned)");
        // a block size that is a multiple of C keeps every thread on ONE channel (register partial sums); with 256 threads and e.g.
        // C = 7 (the bias gradient of the 7-class output conv) every element went through an LDS atomic: 295 us for 117 MB
        const int bd = (256 % C == 0) ? 256 : (256 / C) * C;
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(grid_for(npix * C)), dim3(bd), 2 * C * sizeof(float), st, (const float*)g, (const float*)c, mean,
                           invstd, mask_scale, mask_shift, npix, C, stats);
    }
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_bn_bwd_reduce(const float* g, const float* c, const float* mean, const float* invstd,
                                  const float* mask_scale, const float* mask_shift, long npix, int C, double* stats,
                                  void* stream) {
    SRBH_REQUIRE(g && stats && npix > 0 && C > 0 && C <= 64, "srbh_bn_bwd_reduce: bad arguments");
    SRBH_REQUIRE(!c || (mean && invstd), "srbh_bn_bwd_reduce: c needs mean/invstd");
    SRBH_REQUIRE(!mask_scale || (c && mask_shift), "srbh_bn_bwd_reduce: mask needs c and mask_shift");
    return bn_bwd_reduce_impl(g, nullptr, nullptr, c, mean, invstd, mask_scale, mask_shift, npix, C, stats, stream);
}

extern "C" int srbh_bn_bwd_reduce_relu(const float* g, const float* relu_ref, float* dz_out, const float* c, const float* mean,
                                       const float* invstd, long npix, int C, double* stats, void* stream) {
    SRBH_REQUIRE(g && relu_ref && stats && npix > 0 && C > 0 && C <= 64 && (C & 3) == 0, "srbh_bn_bwd_reduce_relu: bad arguments (C % 4 == 0)");
    SRBH_REQUIRE(!c || (mean && invstd), "srbh_bn_bwd_reduce_relu: c needs mean/invstd");
    SRBH_REQUIRE((((uintptr_t)g | (uintptr_t)relu_ref | (uintptr_t)dz_out | (uintptr_t)c) & 15) == 0, "srbh_bn_bwd_reduce_relu: 16-byte aligned tensors");
    return bn_bwd_reduce_impl(g, relu_ref, dz_out, c, mean, invstd, nullptr, nullptr, npix, C, stats, stream);
}

/* the same two reductions with 16-bit tensors in memory (io: SRBH_BN_G_B16 | SRBH_BN_C_H16 | SRBH_BN_OUT_B16); relu_ref / dz_out optional */
extern "C" int srbh_bn_bwd_reduce_io(const void* g, const float* relu_ref, void* dz_out, const void* c, const float* mean, const float* invstd,
                                     const float* mask_scale, const float* mask_shift, long npix, int C, double* stats, int io, void* stream) {
    SRBH_REQUIRE(g && stats && npix > 0 && C > 0 && C <= 64, "srbh_bn_bwd_reduce_io: bad arguments");
    SRBH_REQUIRE(!c || (mean && invstd), "srbh_bn_bwd_reduce_io: c needs mean/invstd");
    SRBH_REQUIRE(!mask_scale || (c && mask_shift), "srbh_bn_bwd_reduce_io: mask needs c and mask_shift");
    SRBH_REQUIRE(!(relu_ref && mask_scale), "srbh_bn_bwd_reduce_io: either the block-closing ReLU (relu_ref) or the bn1 mask");
    SRBH_REQUIRE((io & ~31) == 0, "srbh_bn_bwd_reduce_io: unknown io bits");
    return bn_bwd_reduce_impl(g, relu_ref, dz_out, c, mean, invstd, mask_scale, mask_shift, npix, C, stats, stream, io);
}

extern "C" int srbh_bn_bwd_apply_io(const void* g, const void* c, const float* mean, const float* invstd, const float* mask_scale,
                                    const float* mask_shift, const float* coef, const float* k1, const float* k2, void* out, long npix, int C,
                                    int io, void* stream) {
    SRBH_REQUIRE(g && c && mean && invstd && coef && k1 && k2 && out && npix > 0 && C > 0, "srbh_bn_bwd_apply_io: bad arguments");
    const bool gb = (io & SRBH_BN_G_B16) != 0, ch = (io & SRBH_BN_C_H16) != 0, ob = (io & SRBH_BN_OUT_B16) != 0;
    SRBH_REQUIRE((io & ~7) == 0 && (C & 3) == 0 && (256 % (C >> 2)) == 0 && ((uintptr_t)g & (gb ? 7 : 15)) == 0 && ((uintptr_t)c & (ch ? 7 : 15)) == 0 &&
                 ((uintptr_t)out & (ob ? 7 : 15)) == 0, "srbh_bn_bwd_apply_io: needs the vector form (C %% 4 == 0, 256 %% (C/4) == 0, aligned)");
    const long n4 = npix * (C >> 2);
#define SRBH_BNA(G_, C_, O_) hipLaunchKernelGGL((bn_bwd_apply4_kernel<G_, C_, O_>), dim3(grid4_for(n4)), dim3(256), 0, (hipStream_t)stream, \
                                                g, c, mean, invstd, mask_scale, mask_shift, coef, k1, k2, out, n4, C)
    switch ((gb ? 4 : 0) | (ch ? 2 : 0) | (ob ? 1 : 0)) {
        case 0: SRBH_BNA(0, 0, 0); break; case 1: SRBH_BNA(0, 0, 1); break; case 2: SRBH_BNA(0, 1, 0); break; case 3: SRBH_BNA(0, 1, 1); break;
        case 4: SRBH_BNA(1, 0, 0); break; case 5: SRBH_BNA(1, 0, 1); break; case 6: SRBH_BNA(1, 1, 0); break; default: SRBH_BNA(1, 1, 1); break;
    }
#undef SRBH_BNA
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_bn_bwd_finalize(const double* stats, int C, double count, const float* gamma, const float* invstd,
                                    float* dgamma, float* dbeta, float* coef, float* k1, float* k2, void* stream) {
    SRBH_REQUIRE(stats && C > 0 && C <= 64 && count > 0, "srbh_bn_bwd_finalize: bad arguments");
    SRBH_REQUIRE(!coef || (invstd && k1 && k2), "srbh_bn_bwd_finalize: coef needs invstd, k1, k2");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (double*)stats, C, count, gamma, invstd,
                       dgamma, dbeta, coef, k1, k2, 0);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_bn_bwd_finalize_clear(double* stats, int C, double count, const float* gamma, const float* invstd,
                                          float* dgamma, float* dbeta, float* coef, float* k1, float* k2, void* stream) {
    SRBH_REQUIRE(stats && C > 0 && C <= 64 && count > 0, "srbh_bn_bwd_finalize_clear: bad arguments");
    SRBH_REQUIRE(!coef || (invstd && k1 && k2), "srbh_bn_bwd_finalize_clear: coef needs invstd, k1, k2");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, stats, C, count, gamma, invstd,
                       dgamma, dbeta, coef, k1, k2, 1);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_bn_bwd_apply(const float* g, const float* c, const float* mean, const float* invstd,
                                 const float* mask_scale, const float* mask_shift, const float* coef, const float* k1,
                                 const float* k2, float* out, long npix, int C, void* stream) {
    SRBH_REQUIRE(g && c && mean && invstd && coef && k1 && k2 && out && npix > 0 && C > 0, "srbh_bn_bwd_apply: bad arguments");
    if ((C & 3) == 0 && (256 % (C >> 2)) == 0 && (((uintptr_t)g | (uintptr_t)c | (uintptr_t)out | (uintptr_t)mean | (uintptr_t)invstd | (uintptr_t)coef |
                                                    (uintptr_t)k1 | (uintptr_t)k2 | (uintptr_t)mask_scale | (uintptr_t)mask_shift) & 15) == 0) {
        const long n4 = npix * (C >> 2);
        hipLaunchKernelGGL((bn_bwd_apply4_kernel<0, 0, 0>), dim3(grid4_for(n4)), dim3(256), 0, (hipStream_t)stream, (const void*)g, (const void*)c,
                           mean, invstd, mask_scale, mask_shift, coef, k1, k2, (void*)out, n4, C);
    } else
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid_for(npix * C)), dim3(256), 0, (hipStream_t)stream, g, c, mean,
                       invstd, mask_scale, mask_shift, coef, k1, k2, out, npix * C, C);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_ps2_inverse(const float* g_ps, float* g, int B, int H, int W, int C, void* stream) {
    SRBH_REQUIRE(g_ps && g && B > 0 && H > 0 && W > 0 && C > 0, "srbh_ps2_inverse: bad arguments");
    long total = (long)B * H * W * 4 * C;
    if ((C & 3) == 0 && (((uintptr_t)g_ps | (uintptr_t)g) & 15) == 0) {
        const long n16 = total / 16;
        hipLaunchKernelGGL(ps2_inverse4_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g_ps, g, n16, H, W, C);
        SRBH_HIP(hipGetLastError());
        return SRBH_OK;
    }
    hipLaunchKernelGGL(ps2_inverse_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, g_ps, g, B, H, W, C);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}
