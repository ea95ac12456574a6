// srbh_dconv.hip -- the 3x3 convolutions of the two U-Net decoders (reference mymodels.py:245-258 builds them with
// segmentation_models_pytorch's UnetDecoder: per block nearest x2 -> concat skip -> [Conv3x3 (no bias) + BatchNorm + ReLU] x 2; called at
// mymodels.py:279,287) on NCHW fp32 tensors, 16-bit matrix-core operands with fp32 accumulation: forward (fp16), data gradient (bf16: the
// forward kernel over the transposed + flipped weights) and weight gradient (bf16).
//
// Why it exists: in the training step MIOpen serves these 20 convolutions per direction with fp32 Winograd kernels (47 us per call on
// 2x2 ... 64x64 planes), NHWC implicit-GEMM weight gradients wrapped in batched transposes and zero fills -- ~130 launches and 3.3 ms of
// a 37.7 ms step (profiles/r04e_train_steady_kernel_stats.txt), 1.75 ms of a 28.5 ms inference batch -- for 0.33 GFLOP per tile.
//
// Shapes (B = 64 training / 128 inference): 608->256 and 256->256 at 4x4, 312->128 / 128->128 at 8x8, 160->64 / 64->64 at 16x16,
// 112->32 / 32->32 at 32x32, 32->16 / 16->16 at 64x64: implicit GEMMs M = Cout, N = B H W, K = 9 Cin with N from 1 024 to 262 144 and K
// from 144 to 5 472, so ONE tiling cannot fill 256 CUs everywhere: the workgroup tile is NPX = 64 or 256 pixels x 16 MB output channels
// (host picks per shape so that the grid has >= 256 workgroups where the problem allows).
//
// forward / data gradient (dconv_kernel): v_mfma_f32_16x16x16 (f16 | bf16), A = 16 output channels x 16 input channels of one tap straight
// from global memory in fragment order (srbh_hpack_conv_h16's layout: 8 bytes per lane, prefetched one chunk ahead), B = 16 pixels x the
// same 16 channels from LDS.  Per 16-channel chunk the input tile (+ 1-pixel halo, zero where the image ends) is staged pixel-major --
// one 32-byte record per pixel -- converted on the way: a thread loads the same pixel of four channel planes (coalesced along the
// pixels of NCHW) and writes one 8-byte quad; loads of chunk c + 1 are in flight in registers while chunk c is multiplied, two LDS
// stages, one barrier per chunk.  A tap is a constant LDS offset.  Output straight from the D layout: 16 consecutive pixels per channel.
//
// weight gradient (dconv_wgrad_kernel): K is the pixel axis and NCHW is pixel-contiguous, so both operands come straight from global
// memory: A = dY[16 output channels][4 consecutive pixels per lane group], B = X[16 input channels][the same 4 pixels shifted by the tap]
// (row shift = another row; column shift = the aligned quad funnelled with its left / right neighbour, zero at the row ends).  One
// workgroup = one (16 MO x 16) block of dW for all nine taps, its four waves and grid.y further workgroups split the pixels; partials in
// fixed slots, one ordered reduce (deterministic, nothing to zero).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "srbh.h"
#include "srbh_internal.h"

namespace {
using namespace srbh;
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef short short4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ short bf16_rne(float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    return (short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
template <int BF16>
__device__ __forceinline__ short4v cvt4(const float a, const float b, const float c, const float d) {
    if constexpr (BF16 != 0) {
        typedef unsigned uint2p __attribute__((ext_vector_type(2)));
        const uint2p pk = {bf16x2_rne(a, b), bf16x2_rne(c, d)};
        return __builtin_bit_cast(short4v, pk);
    } else {
        const half4 h = {(_Float16)a, (_Float16)b, (_Float16)c, (_Float16)d};
        return __builtin_bit_cast(short4v, h);
    }
}
template <int BF16>
__device__ __forceinline__ floatx4 mma(const short4v a, const short4v b, const floatx4 c) {
    if constexpr (BF16 != 0) return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(half4, a), __builtin_bit_cast(half4, b), c, 0, 0, 0);
}

struct DCParams {
    const float* x;       // [B][Cin][W][W]
    const short* w;       // [chunk][tap][ob][lane][4]
    float* y;             // [B][Cout][W][W]
    int B, Cin, Cout, nchunk, nob;
    const float* scale;   // [Cout] folded inference BatchNorm (null: none)
    const float* shift;
    int act;              // 0 none | 2 ReLU (after the affine)
};

// square planes W x W, W = 1 << LOGW (4 ... 64); NPX pixels per workgroup (64 | 256), 16 MB output channels per workgroup
template <int LOGW, int NPX>
struct DCGeo {
    static constexpr int W = 1 << LOGW, PP = W * W;
    static constexpr bool ROWT = PP >= NPX;                       // a tile is TR rows of one image, else NI whole images
    static constexpr int TR = ROWT ? NPX / W : W, NI = ROWT ? 1 : NPX / PP;
    static constexpr int RH = TR + 2, RW = W + 2, NREC = NI * RH * RW;
    static constexpr int S = ROWT ? (TR + 2) * W : NPX;           // staged source pixels of a tile (row tiles: with the halo rows)
    static constexpr int NITEM = (4 * S + 255) / 256;             // (quad, pixel) items per thread and chunk
    static constexpr int STAGE_B = NREC * 32;
    static constexpr int NBW = NPX / 64;                          // 16-pixel blocks per wave
    static_assert(!ROWT || (W % TR == 0 && TR >= 1), "row tiles must divide the plane");
};

template <int LOGW, int NPX, int MB, int BF16>
__global__ __launch_bounds__(256) void dconv_kernel(const DCParams p) {
    using G = DCGeo<LOGW, NPX>;
    constexpr int W = G::W, PP = G::PP, RW = G::RW, RH = G::RH, NBW = G::NBW;
    extern __shared__ __attribute__((aligned(16))) char dsm[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kg = lane >> 4;
    // ---- tile origin
    int b0, y0;
    if constexpr (G::ROWT) {
        constexpr int TPI = W / G::TR;
        b0 = blockIdx.x / TPI;
        y0 = (blockIdx.x - b0 * TPI) * G::TR;
    } else {
        b0 = blockIdx.x * G::NI;
        y0 = 0;
    }
    const int ob0 = blockIdx.y * MB;
    // ---- staging items of this thread: (quad q, source pixel s) -> global offset inside a channel plane set, LDS byte offset, validity
    int goff[G::NITEM], loff[G::NITEM];
    unsigned vmask = 0, qsel = 0;
#pragma unroll
    for (int it = 0; it < G::NITEM; ++it) {
        const int item = tid + it * 256;
        const int q = item / G::S, s = item - q * G::S;
        bool ok = item < 4 * G::S;
        int g, rec;
        if constexpr (G::ROWT) {
            const int r = s >> LOGW, x = s & (W - 1), y = y0 - 1 + r;
            ok = ok && y >= 0 && y < W;
            g = y * W + x;
            rec = r * RW + x + 1;
        } else {
            const int il = s / PP, rem = s - il * PP, yy = rem >> LOGW, x = rem & (W - 1);
            ok = ok && b0 + il < p.B;
            g = il * p.Cin * PP + rem;
            rec = (il * RH + yy + 1) * RW + x + 1;
        }
        goff[it] = g;
        loff[it] = rec * 32 + (q & 3) * 8;
        if (ok) vmask |= 1u << it;
        qsel |= (unsigned)(q & 3) << (2 * it);
    }
    const float* xb = p.x + (long)b0 * p.Cin * PP;
    // ---- B-fragment base addresses of this lane: pixel (wave, nb, l15) -> record
    int rbase[NBW];
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int px = wave * (NPX / 4) + nb * 16 + l15;
        int rec;
        if constexpr (G::ROWT) rec = ((px >> LOGW) + 1) * RW + (px & (W - 1)) + 1;
        else rec = ((px / PP) * RH + ((px & (PP - 1)) >> LOGW) + 1) * RW + (px & (W - 1)) + 1;
        rbase[nb] = rec * 32 + kg * 8;
    }
    // ---- zero both stages once: halo records and out-of-image pixels are never written afterwards
    for (int i = tid; i < 2 * G::STAGE_B / 16; i += 256) ((floatx4*)dsm)[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    floatx4 acc[MB][NBW];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int nb = 0; nb < NBW; ++nb) acc[mb][nb] = floatx4{0.f, 0.f, 0.f, 0.f};
    float ld[G::NITEM][4];
    short4v wn[9][MB];
    auto issue = [&](const int c) {
#pragma unroll
        for (int it = 0; it < G::NITEM; ++it) {
            const int ch = c * 16 + (int)((qsel >> (2 * it)) & 3u) * 4;
            const float* src = xb + (long)ch * PP + goff[it];
            const bool ok = (vmask >> it) & 1u;
#pragma unroll
            for (int j = 0; j < 4; ++j) ld[it][j] = (ok && ch + j < p.Cin) ? src[(long)j * PP] : 0.f;
        }
        const short4v* wp = (const short4v*)p.w + ((long)c * 9 * p.nob + ob0) * 64 + lane;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
                wn[tap][mb] = ob0 + mb < p.nob ? wp[((long)tap * p.nob + mb) * 64] : short4v{0, 0, 0, 0};
    };
    auto commit = [&](char* stage) {
#pragma unroll
        for (int it = 0; it < G::NITEM; ++it)
            if ((vmask >> it) & 1u) *(short4v*)(stage + loff[it]) = cvt4<BF16>(ld[it][0], ld[it][1], ld[it][2], ld[it][3]);
    };
    issue(0);
    __syncthreads();                       // the zero fill is complete
    commit(dsm);
    short4v wc[9][MB];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) wc[tap][mb] = wn[tap][mb];
    __syncthreads();
    for (int c = 0; c < p.nchunk; ++c) {
        const char* stage = dsm + (c & 1) * G::STAGE_B;
        if (c + 1 < p.nchunk) issue(c + 1);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = ((tap / 3 - 1) * RW + (tap % 3 - 1)) * 32;
#pragma unroll
            for (int nb = 0; nb < NBW; ++nb) {
                const short4v bfrag = *(const short4v*)(stage + rbase[nb] + toff);
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) acc[mb][nb] = mma<BF16>(wc[tap][mb], bfrag, acc[mb][nb]);
            }
        }
        if (c + 1 < p.nchunk) {
            commit(dsm + ((c + 1) & 1) * G::STAGE_B);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) wc[tap][mb] = wn[tap][mb];
        }
        __syncthreads();                   // stage (c+1)&1 is complete; everybody is done reading stage c&1
    }
    // ---- epilogue: lane holds output channels 4 kg .. 4 kg + 3 of pixel l15 of each block
#pragma unroll
    for (int nb = 0; nb < NBW; ++nb) {
        const int px = wave * (NPX / 4) + nb * 16 + l15;
        long obase;
        bool ok;
        if constexpr (G::ROWT) {
            obase = (long)b0 * p.Cout * PP + (y0 + (px >> LOGW)) * W + (px & (W - 1));
            ok = true;
        } else {
            const int il = px / PP;
            obase = (long)(b0 + il) * p.Cout * PP + (px & (PP - 1));
            ok = b0 + il < p.B;
        }
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int co = (ob0 + mb) * 16 + kg * 4 + r;
                if (ok && co < p.Cout) {
                    float v = acc[mb][nb][r];
                    if (p.scale) v = fmaf(v, p.scale[co], p.shift[co]);      // (inference: BatchNorm + ReLU of the decoder block in the conv's store)
                    if (p.act == 2) v = fmaxf(v, 0.f);
                    p.y[obase + (long)co * PP] = v;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------- weight gradient
struct DWGParams {
    const float* x; const float* dy; float* part;
    int B, Cin, Cout, cib, nsplit;
};

template <int LOGW, int MO>
__global__ __launch_bounds__(256) void dconv_wgrad_kernel(const DWGParams p) {
    constexpr int W = 1 << LOGW, PP = W * W, SPI = PP / 16;      // 16-pixel K steps per image
    __shared__ float red[4][9][4][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kg = lane >> 4;
    const int tco = blockIdx.x / p.cib, tci = blockIdx.x - tco * p.cib;
    const int ci = tci * 16 + l15;
    const bool ciok = ci < p.Cin;
    const int KS = p.nsplit * 4, ks = blockIdx.y * 4 + wave;
    const long steps = (long)p.B * SPI;
    floatx4 acc[MO][9];
#pragma unroll
    for (int mo = 0; mo < MO; ++mo)
#pragma unroll
        for (int t = 0; t < 9; ++t) acc[mo][t] = floatx4{0.f, 0.f, 0.f, 0.f};
    struct Ld { floatx4 a[MO]; floatx4 q[3]; float l[3], r[3]; };
    auto load = [&](const long st, Ld& v) {
        const long b = st / SPI;
        const int p0 = (int)(st - b * SPI) * 16 + kg * 4;
        const int y = p0 >> LOGW, x0 = p0 & (W - 1);
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) {
            const int co = (tco * MO + mo) * 16 + l15;
            v.a[mo] = co < p.Cout ? *(const floatx4*)(p.dy + ((long)b * p.Cout + co) * PP + p0) : floatx4{0.f, 0.f, 0.f, 0.f};
        }
        const float* xb = p.x + ((long)b * p.Cin + ci) * PP + x0;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int yy = y + d - 1;
            const bool rok = ciok && yy >= 0 && yy < W;
            v.q[d] = rok ? *(const floatx4*)(xb + yy * W) : floatx4{0.f, 0.f, 0.f, 0.f};
            v.l[d] = (rok && x0 > 0) ? xb[yy * W - 1] : 0.f;
            v.r[d] = (rok && x0 + 4 < W) ? xb[yy * W + 4] : 0.f;
        }
    };
    // a wave walks a CONTIGUOUS range of K steps: consecutive 16-pixel groups of a row share cache lines, and the rows y - 1 / y / y + 1 of
    // one step are y / y + 1 / y + 2 of a later one (L1 / L2 hits instead of three far-apart fetches per step)
    const long per = (steps + KS - 1) / KS, s_lo = ks * per, s_hi = s_lo + per < steps ? s_lo + per : steps;
    Ld cur, nxt;
    if (s_lo < s_hi) load(s_lo, cur);
    for (long st = s_lo; st < s_hi; ++st) {
        const bool more = st + 1 < s_hi;
        if (more) load(st + 1, nxt);
        short4v a[MO];
#pragma unroll
        for (int mo = 0; mo < MO; ++mo) a[mo] = cvt4<1>(cur.a[mo][0], cur.a[mo][1], cur.a[mo][2], cur.a[mo][3]);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const floatx4 q = cur.q[d];
            const short4v b0 = cvt4<1>(cur.l[d], q[0], q[1], q[2]), b1 = cvt4<1>(q[0], q[1], q[2], q[3]), b2 = cvt4<1>(q[1], q[2], q[3], cur.r[d]);
#pragma unroll
            for (int mo = 0; mo < MO; ++mo) {
                acc[mo][d * 3 + 0] = mma<1>(a[mo], b0, acc[mo][d * 3 + 0]);
                acc[mo][d * 3 + 1] = mma<1>(a[mo], b1, acc[mo][d * 3 + 1]);
                acc[mo][d * 3 + 2] = mma<1>(a[mo], b2, acc[mo][d * 3 + 2]);
            }
        }
        if (more) cur = nxt;
    }
    // ---- fold the four waves (fixed order), write this workgroup's partial: [split][tile][mo][tap][co 16][ci 16]
    float* out = p.part + ((long)blockIdx.y * gridDim.x + blockIdx.x) * (MO * 9 * 256);
#pragma unroll
    for (int mo = 0; mo < MO; ++mo) {
        if (mo) __syncthreads();
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave][t][r][lane] = acc[mo][t][r];
        __syncthreads();
        for (int u = tid; u < 9 * 256; u += 256) {
            const int ln = u & 63, r = (u >> 6) & 3, t = u >> 8;
            const float v = (red[0][t][r][ln] + red[1][t][r][ln]) + (red[2][t][r][ln] + red[3][t][r][ln]);
            // D layout: lane ln holds row (co) 4 (ln / 16) + r, column (ci) ln % 16
            out[(mo * 9 + t) * 256 + ((ln >> 4) * 4 + r) * 16 + (ln & 15)] = v;
        }
    }
}

// partial [split][tile = tco * cib + tci][mo][tap][16 co][16 ci] -> dW [Cout][Cin][3][3], splits summed in a fixed order.
// Thread = one element of the partial layout (coalesced reads; the scattered 4-byte writes are few); the four waves of a workgroup take
// every fourth split with two independent chains each and fold through LDS -- a single chain over 512 splits was a 100 us latency chain
// of its own (the 16 -> 16 conv at 64x64: one dW block, 512 K-splits).
template <int MO>
__global__ __launch_bounds__(256) void dconv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int Cout, int Cin, int cib,
                                                                 int ntile, int nsplit) {
    __shared__ float red[4][64];
    const int ln = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const long stride = (long)ntile * MO * 9 * 256;
    const long j = (long)blockIdx.x * 64 + ln;
    float a0 = 0.f, a1 = 0.f;
    if (j < stride) {
        int s = sg;
        for (; s + 4 < nsplit; s += 8) {
            a0 += part[s * stride + j];
            a1 += part[(s + 4) * stride + j];
        }
        if (s < nsplit) a0 += part[s * stride + j];
    }
    red[sg][ln] = a0 + a1;
    __syncthreads();
    if (sg == 0 && j < stride) {
        const float v = (red[0][ln] + red[1][ln]) + (red[2][ln] + red[3][ln]);
        const int cil = (int)(j & 15), col = (int)((j >> 4) & 15);
        long f = j >> 8;
        const int tap = (int)(f % 9);
        f /= 9;
        const int mo = (int)(f % MO);
        const int tile = (int)(f / MO);
        const int tco = tile / cib, tci = tile - tco * cib;
        const int co = (tco * MO + mo) * 16 + col, ci = tci * 16 + cil;
        if (co < Cout && ci < Cin) dw[((long)co * Cin + ci) * 9 + tap] = v;
    }
}

// both 16-bit images of MANY decoder conv weights in one launch (a training step changes all of them: 40 convs x 2 packs were 80
// launches of ~4.6 us): blockIdx.y = conv, fragment order of srbh_hpack_conv_h16 ([chunk][tap][ob][lane][4]);
// fwd = fp16 of W[oc][ic][tap], bwd = bf16 of the transposed + flipped weight (its "cout" = Cin, "cin" = Cout)
__device__ __forceinline__ short pack_elem(const float* __restrict__ w, long idx, int cout, int cin, int nob, int transpose_flip, int bf16) {
    const int j = idx & 3, lane = (idx >> 2) & 63;
    long f = idx >> 8;
    const int ob = (int)(f % nob);
    f /= nob;
    const int tap = (int)(f % 9), chunk = (int)(f / 9);
    const int oc = ob * 16 + (lane & 15), ic = chunk * 16 + (lane >> 4) * 4 + j;
    float v = 0.f;
    if (oc < cout && ic < cin) v = transpose_flip ? w[((long)ic * cout + oc) * 9 + (8 - tap)] : w[((long)oc * cin + ic) * 9 + tap];
    if (bf16) return bf16_rne(v);
    return __builtin_bit_cast(short, (_Float16)v);
}
__global__ __launch_bounds__(256) void dconv_pack_many_kernel(const srbh_dconv_pack_desc* __restrict__ table) {
    const srbh_dconv_pack_desc d = table[blockIdx.y];
    const int nob_f = (d.cout + 15) / 16, nch_f = (d.cin + 15) / 16;
    const long tot = (long)nch_f * 9 * nob_f * 256;           // (the same count for both images: chunks x blocks swap roles)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < tot; i += (long)gridDim.x * 256) {
        ((short*)d.fwd)[i] = pack_elem(d.w, i, d.cout, d.cin, nob_f, 0, 0);
        ((short*)d.bwd)[i] = pack_elem(d.w, i, d.cin, d.cout, nch_f, 1, 1);
    }
}

int log2_exact(int v) {
    for (int l = 0; l < 16; ++l)
        if ((1 << l) == v) return l;
    return -1;
}

struct FwdPlan { int npx, mb; };
FwdPlan plan_fwd(int B, int Cout, int W) {
    // >= 256 workgroups where the problem allows; wide tiles (256 pixels) amortise the halo and the weight fragments
    const int nob = (Cout + 15) / 16;
    const long px = (long)B * W * W;
    // (MB <= 2: the weight fragments of two chunks live in registers -- 9 taps x MB x 2 registers x 2)
    const int cand[4][2] = {{256, 2}, {256, 1}, {64, 2}, {64, 1}};
    for (int i = 0; i < 4; ++i) {
        const int npx = cand[i][0], mb = cand[i][1];
        if (mb > nob) continue;
        if (nob % mb != 0 && mb > 1) continue;        // (keeps every workgroup's channel blocks real)
        const long wgs = ((px + npx - 1) / npx) * ((nob + mb - 1) / mb);
        if (wgs >= 256) return {npx, mb};
    }
    return {64, 1};
}

template <int LOGW, int NPX, int MB>
int launch_fwd(const DCParams& p, int bf16, hipStream_t st) {
    using G = DCGeo<LOGW, NPX>;
    const int tiles = G::ROWT ? p.B * ((1 << LOGW) / G::TR) : (p.B + G::NI - 1) / G::NI;
    const dim3 grid(tiles, (p.nob + MB - 1) / MB);
    constexpr int lds = 2 * G::STAGE_B;
    if (bf16) hipLaunchKernelGGL((dconv_kernel<LOGW, NPX, MB, 1>), grid, dim3(256), lds, st, p);
    else hipLaunchKernelGGL((dconv_kernel<LOGW, NPX, MB, 0>), grid, dim3(256), lds, st, p);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

template <int LOGW>
int dispatch_fwd(const DCParams& p, const FwdPlan pl, int bf16, hipStream_t st) {
    if (pl.npx == 256) {
        if (pl.mb == 2) return launch_fwd<LOGW, 256, 2>(p, bf16, st);
        return launch_fwd<LOGW, 256, 1>(p, bf16, st);
    }
    if (pl.mb == 2) return launch_fwd<LOGW, 64, 2>(p, bf16, st);
    return launch_fwd<LOGW, 64, 1>(p, bf16, st);
}

int wgrad_split(int B, int Cin, int Cout, int W, int mo) {
    // workgroups per dW block: enough to fill the chip (~1 024 waves), at least ~8 K steps of 16 pixels per wave
    const long tiles = (long)((Cout + 16 * mo - 1) / (16 * mo)) * ((Cin + 15) / 16);
    const long steps = (long)B * W * W / 16;
    long s = (512 + tiles - 1) / tiles;
    const long smax = steps / 32 > 0 ? steps / 32 : 1;
    if (s > smax) s = smax;
    if (s < 1) s = 1;
    if (s > 1024) s = 1024;
    return (int)s;
}

}  // namespace

/* table: n descriptors in DEVICE memory; fwd / bwd: srbh_hpack_h16_bytes(cout, cin, 3) bytes each (the two images have the same size) */
extern "C" int srbh_dconv_pack_many(const srbh_dconv_pack_desc* table, int n, void* stream) {
    SRBH_REQUIRE(table && n > 0, "srbh_dconv_pack_many: bad arguments");
    hipLaunchKernelGGL(dconv_pack_many_kernel, dim3(64, n), dim3(256), 0, (hipStream_t)stream, table);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_dconv_supported(int B, int Cin, int Cout, int H, int W) {
    const int lw = log2_exact(W);
    return B > 0 && Cin > 0 && Cout > 0 && H == W && lw >= 2 && lw <= 6 && (long)B * (Cin > Cout ? Cin : Cout) * H * W < (1L << 31);
}

/* y = conv3x3(x, w), stride 1, zero padding 1, no bias.  x (B,Cin,H,W) / y (B,Cout,H,W) NCHW fp32, H == W in {4, 8, 16, 32, 64};
 * wpack = srbh_hpack_conv_h16(w, Cout, Cin, 3, transpose_flip, bf16, ...).  bf16 = 0: fp16 operands (forward); bf16 = 1: bf16
 * operands -- with the transposed + flipped pack of the forward weight (cout := forward Cin, cin := forward Cout) this is the data
 * gradient dX = conv^T(dY, W). */
static int dconv_fwd_impl(const float* x, const void* wpack, float* y, int B, int Cin, int Cout, int H, int W, int bf16, const float* scale,
                          const float* shift, int act, void* stream);
extern "C" int srbh_dconv_fwd(const float* x, const void* wpack, float* y, int B, int Cin, int Cout, int H, int W, int bf16, void* stream) {
    return dconv_fwd_impl(x, wpack, y, B, Cin, Cout, H, W, bf16, nullptr, nullptr, 0, stream);
}
/* the same conv with y = act(conv * scale[co] + shift[co]) in its store (inference: the decoder block's BatchNorm folded to an affine, + ReLU) */
extern "C" int srbh_dconv_fwd_epi(const float* x, const void* wpack, float* y, int B, int Cin, int Cout, int H, int W, int bf16, const float* scale,
                                  const float* shift, int act, void* stream) {
    SRBH_REQUIRE((scale == nullptr) == (shift == nullptr) && (act == 0 || act == 2), "srbh_dconv_fwd_epi: bad epilogue");
    return dconv_fwd_impl(x, wpack, y, B, Cin, Cout, H, W, bf16, scale, shift, act, stream);
}
static int dconv_fwd_impl(const float* x, const void* wpack, float* y, int B, int Cin, int Cout, int H, int W, int bf16, const float* scale,
                          const float* shift, int act, void* stream) {
    SRBH_REQUIRE(x && wpack && y, "srbh_dconv_fwd: null pointer");
    SRBH_REQUIRE(srbh_dconv_supported(B, Cin, Cout, H, W), "srbh_dconv_fwd: unsupported geometry B=%d Cin=%d Cout=%d H=%d W=%d (square planes 4..64)", B, Cin, Cout, H, W);
    DCParams p;
    p.x = x; p.w = (const short*)wpack; p.y = y;
    p.B = B; p.Cin = Cin; p.Cout = Cout; p.nchunk = (Cin + 15) / 16; p.nob = (Cout + 15) / 16;
    p.scale = scale; p.shift = shift; p.act = act;
    const FwdPlan pl = plan_fwd(B, Cout, W);
    hipStream_t st = (hipStream_t)stream;
    switch (log2_exact(W)) {
        case 2: return dispatch_fwd<2>(p, pl, bf16, st);
        case 3: return dispatch_fwd<3>(p, pl, bf16, st);
        case 4: return dispatch_fwd<4>(p, pl, bf16, st);
        case 5: return dispatch_fwd<5>(p, pl, bf16, st);
        default: return dispatch_fwd<6>(p, pl, bf16, st);
    }
}

extern "C" size_t srbh_dconv_wgrad_ws_floats(int B, int Cin, int Cout, int H, int W) {
    if (!srbh_dconv_supported(B, Cin, Cout, H, W)) return 0;
    const int mo = (Cout % 32 == 0) ? 2 : 1;
    const long tiles = (long)((Cout + 16 * mo - 1) / (16 * mo)) * ((Cin + 15) / 16);
    return (size_t)wgrad_split(B, Cin, Cout, W, mo) * tiles * mo * 9 * 256;
}

/* dW (Cout,Cin,3,3) fp32 = sum over (b, y, x) of dY[b][co][y][x] * X[b][ci][y+dy-1][x+dx-1], bf16 operands, fp32 accumulation, fixed
 * summation order.  ws: srbh_dconv_wgrad_ws_floats() floats. */
extern "C" int srbh_dconv_wgrad(const float* x, const float* dy, float* dw, float* ws, int B, int Cin, int Cout, int H, int W, void* stream) {
    SRBH_REQUIRE(x && dy && dw && ws, "srbh_dconv_wgrad: null pointer");
    SRBH_REQUIRE(srbh_dconv_supported(B, Cin, Cout, H, W), "srbh_dconv_wgrad: unsupported geometry B=%d Cin=%d Cout=%d H=%d W=%d", B, Cin, Cout, H, W);
    const int mo = (Cout % 32 == 0) ? 2 : 1;
    DWGParams p;
    p.x = x; p.dy = dy; p.part = ws;
    p.B = B; p.Cin = Cin; p.Cout = Cout; p.cib = (Cin + 15) / 16;
    const int tco = (Cout + 16 * mo - 1) / (16 * mo);
    const int ntile = tco * p.cib;
    p.nsplit = wgrad_split(B, Cin, Cout, W, mo);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(ntile, p.nsplit);
#define SRBH_DWG(L_)                                                                                       \
    do {                                                                                                   \
        if (mo == 2) hipLaunchKernelGGL((dconv_wgrad_kernel<L_, 2>), grid, dim3(256), 0, st, p);            \
        else hipLaunchKernelGGL((dconv_wgrad_kernel<L_, 1>), grid, dim3(256), 0, st, p);                    \
    } while (0)
    switch (log2_exact(W)) {
        case 2: SRBH_DWG(2); break;
        case 3: SRBH_DWG(3); break;
        case 4: SRBH_DWG(4); break;
        case 5: SRBH_DWG(5); break;
        default: SRBH_DWG(6); break;
    }
#undef SRBH_DWG
    SRBH_HIP(hipGetLastError());
    const long total = (long)ntile * mo * 9 * 256;
    if (mo == 2) hipLaunchKernelGGL(dconv_wgrad_reduce_kernel<2>, dim3((total + 63) / 64), dim3(256), 0, st, ws, dw, Cout, Cin, p.cib, ntile, p.nsplit);
    else hipLaunchKernelGGL(dconv_wgrad_reduce_kernel<1>, dim3((total + 63) / 64), dim3(256), 0, st, ws, dw, Cout, Cin, p.cib, ntile, p.nsplit);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}
