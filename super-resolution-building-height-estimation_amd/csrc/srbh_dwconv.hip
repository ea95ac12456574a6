// srbh_dwconv.hip -- depthwise KxK convolution (K = 3 or 5, stride 1 or 2), fp32 NCHW, forward / input gradient / weight
// gradient, with the zero padding folded in (top/left pad given; bottom/right implied by the output size).
//
// Why it exists: the EfficientNet-B4 encoder the reference instantiates (mymodels.py:242-248) is stock PyTorch ops in
// this build, but MIOpen has no fp32 solver for its 5x5 / strided depthwise convolutions and falls back
// to `naive_conv_ab_nonpacked_{fwd,bwd,wrw}` -- 10.7 % of the tiled-inference path and 5.6 ms of the 81 ms training step
// (profiles/r01e_*).  These are HBM-bound element-wise kernels: planes are at most 32x32, every tap re-read hits L1/L2.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "srbh.h"
#include "srbh_internal.h"

namespace {
using namespace srbh;
typedef float floatx4 __attribute__((ext_vector_type(4)));

struct DWParams {
    const float* x;     // [B][C][H][W]
    const float* w;     // [C][1][K][K]
    const float* dy;    // [B][C][OH][OW]
    float* out;         // y, dx or dw
    int B, C, H, W, OH, OW, stride, pad_t, pad_l;
};

template <int K>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const DWParams p) {
    const long total = (long)p.B * p.C * p.OH * p.OW;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int ox = idx % p.OW;
        long r = idx / p.OW;
        const int oy = r % p.OH;
        r /= p.OH;                                  // r = b * C + c
        const int c = r % p.C;
        const float* xp = p.x + r * p.H * p.W;
        const float* wp = p.w + (long)c * K * K;
        const int y0 = oy * p.stride - p.pad_t, x0 = ox * p.stride - p.pad_l;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int y = y0 + i;
            if (y < 0 || y >= p.H) continue;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int x = x0 + j;
                if (x >= 0 && x < p.W) acc = fmaf(xp[y * p.W + x], wp[i * K + j], acc);
            }
        }
        p.out[idx] = acc;
    }
}

// dx[y][x] = sum over taps (i, j) with (y + pad_t - i) = s * oy, (x + pad_l - j) = s * ox of dy[oy][ox] * w[i][j]
template <int K>
__global__ __launch_bounds__(256) void dw_bwd_data_kernel(const DWParams p) {
    const long total = (long)p.B * p.C * p.H * p.W;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int x = idx % p.W;
        long r = idx / p.W;
        const int y = r % p.H;
        r /= p.H;
        const int c = r % p.C;
        const float* gp = p.dy + r * p.OH * p.OW;
        const float* wp = p.w + (long)c * K * K;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int ty = y + p.pad_t - i;
            if (ty < 0 || ty % p.stride) continue;
            const int oy = ty / p.stride;
            if (oy >= p.OH) continue;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int tx = x + p.pad_l - j;
                if (tx < 0 || tx % p.stride) continue;
                const int ox = tx / p.stride;
                if (ox < p.OW) acc = fmaf(gp[oy * p.OW + ox], wp[i * K + j], acc);
            }
        }
        p.out[idx] = acc;
    }
}

// dw[c][i][j] = sum over (b, oy, ox) of dy[b][c][oy][ox] * x[b][c][oy*s - pad_t + i][ox*s - pad_l + j]: workgroup (c, s)
// sums the images of batch slice s in a fixed tree (thread-strided partial sums -> wave shuffles -> LDS across the 4
// waves) into ws[s][c][..]; dw_reduce_kernel adds the slices in order: deterministic, and >= ~1000 workgroups even for the
// 144-channel layers
template <int K>
__global__ __launch_bounds__(256) void dw_bwd_weight_kernel(const DWParams p, float* __restrict__ ws, int splits) {
    const int c = blockIdx.x, sp = blockIdx.y;
    const int per_img = p.OH * p.OW;
    const int b0 = (int)((long)p.B * sp / splits), b1 = (int)((long)p.B * (sp + 1) / splits);
    const long n = (long)(b1 - b0) * per_img;
    float acc[K * K];
#pragma unroll
    for (int t = 0; t < K * K; ++t) acc[t] = 0.f;
    for (long e = threadIdx.x; e < n; e += 256) {
        const int b = b0 + (int)(e / per_img);
        const int q = (int)(e % per_img);
        const int oy = q / p.OW, ox = q - oy * p.OW;
        const float g = p.dy[((long)b * p.C + c) * per_img + q];
        const float* xp = p.x + ((long)b * p.C + c) * p.H * p.W;
        const int y0 = oy * p.stride - p.pad_t, x0 = ox * p.stride - p.pad_l;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const int y = y0 + i;
            if (y < 0 || y >= p.H) continue;
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const int x = x0 + j;
                if (x >= 0 && x < p.W) acc[i * K + j] = fmaf(g, xp[y * p.W + x], acc[i * K + j]);
            }
        }
    }
    __shared__ float red[4][K * K];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < K * K; ++t) {
        float v = acc[t];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) red[wave][t] = v;
    }
    __syncthreads();
    if (threadIdx.x < K * K)
        ws[((long)sp * p.C + c) * K * K + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---- LDS-staged forms (round 3).  The one-thread-per-output kernels above spend ~150 instructions per output on 25 predicated taps with 64-bit
// address arithmetic (13-25 us per call on tensors that take 3-5 us to stream); here a workgroup stages P whole planes ZERO-PADDED in LDS
// (coalesced reads of P contiguous planes), so every tap is one unconditional ds_read + fma.
//   forward        : out[oy][ox] = sum_ij xpad[oy s + i][ox s + j] w[i][j]                      (x placed at (pad_t, pad_l))
//   input gradient : dx[y][x]   = sum_ij dyup[y + pad_t + K-1 - i][x + pad_l + K-1 - j] w[i][j]  (dy placed at (K-1 + oy s, K-1 + ox s): zero-upsampled)
struct DWLds {
    const float* src; const float* w; float* out;
    long planes;
    int C, SH, SW, OUTH, OUTW, PH, PW, ss, soy, sox, os, by, bx, P;
};
template <int K, int FLIP>
__global__ __launch_bounds__(256) void dw_lds_kernel(const DWLds p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int t = threadIdx.x, plane_sz = p.PH * p.PW;
    float* wl = sm + p.P * plane_sz;
    const long p0 = (long)blockIdx.x * p.P;
    const int np = (int)((p.planes - p0) < p.P ? (p.planes - p0) : p.P);
    // (zero fill, then scatter the source elements: a single pass over the PADDED index space -- a division and a predicated gather per
    //  padded element -- was measured slower on every shape)
    for (int e = t; e < p.P * plane_sz; e += 256) sm[e] = 0.f;
    for (int e = t; e < np * K * K; e += 256) wl[e] = p.w[(long)((p0 + e / (K * K)) % p.C) * K * K + e % (K * K)];
    __syncthreads();
    const int ssz = p.SH * p.SW;
    const float* sp = p.src + p0 * ssz;
    for (int e = t; e < np * ssz; e += 256) {
        const int pl = e / ssz, q = e - pl * ssz, u = q / p.SW, v = q - u * p.SW;
        const int r = u * p.ss + p.soy, c = v * p.ss + p.sox;
        if (r < p.PH && c < p.PW) sm[pl * plane_sz + r * p.PW + c] = sp[e];
    }
    __syncthreads();
    const int osz = p.OUTH * p.OUTW;
    float* op = p.out + p0 * osz;
    for (int e = t; e < np * osz; e += 256) {
        const int pl = e / osz, q = e - pl * osz, oy = q / p.OUTW, ox = q - oy * p.OUTW;
        const float* base = sm + pl * plane_sz + (oy * p.os + p.by) * p.PW + ox * p.os + p.bx;
        const float* wp = wl + pl * K * K;
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int i = 0; i < K; ++i)
#pragma unroll
            for (int j = 0; j < K; ++j) {
                const float v = FLIP ? base[-(i * p.PW + j)] : base[i * p.PW + j];
                if ((i * K + j) & 1) a1 = fmaf(v, wp[i * K + j], a1);
                else a0 = fmaf(v, wp[i * K + j], a0);
            }
        op[e] = a0 + a1;
    }
}

// ---- MBConv middle at inference, ONE launch (round 4): y = swish(bn1(dw(swish(bn0(e))))) plus the per-plane mean squeeze-excite pools,
// with both inference BatchNorms folded to (scale, shift).  bn0 + swish is applied while the expand conv's output is staged (the zero
// padding is the padding of the ACTIVATED tensor, as in the module chain), bn1 + swish + the pool on the accumulators: the three
// elementwise passes of the unfused chain (affine_act, affine_act_pool over 6x-expanded tensors) and their two intermediates go away.
// G = 2^k threads share one plane (G = min(256, outputs per plane)); the pool is a fixed butterfly over those lanes (+ an ordered LDS sum
// across waves when G > 64): the same bits every run.
struct DWEval {
    const float* a0; const float* b0;      // bn0 folded (null: the block has no expand conv)
    const float* a1; const float* b1;      // bn1 folded
    float* pooled;                         // [planes] means of the activated output
    int G;
};
__device__ __forceinline__ float silu_f(float u) { return u / (1.f + __expf(-u)); }
template <int K>
__global__ __launch_bounds__(256) void dw_lds_eval_kernel(const DWLds p, const DWEval q) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[4];
    const int t = threadIdx.x, plane_sz = p.PH * p.PW;
    float* wl = sm + p.P * plane_sz;
    const long p0 = (long)blockIdx.x * p.P;
    const int np = (int)((p.planes - p0) < p.P ? (p.planes - p0) : p.P);
    for (int e = t; e < p.P * plane_sz; e += 256) sm[e] = 0.f;
    for (int e = t; e < np * K * K; e += 256) wl[e] = p.w[(long)((p0 + e / (K * K)) % p.C) * K * K + e % (K * K)];
    __syncthreads();
    const int ssz = p.SH * p.SW;
    const float* sp = p.src + p0 * ssz;
    for (int e = t; e < np * ssz; e += 256) {
        const int pl = e / ssz, r0 = e - pl * ssz, u = r0 / p.SW, v = r0 - u * p.SW;
        const int r = u * p.ss + p.soy, c = v * p.ss + p.sox;
        float x = sp[e];
        if (q.a0) {
            const int ch = (int)((p0 + pl) % p.C);
            x = silu_f(fmaf(x, q.a0[ch], q.b0[ch]));
        }
        if (r < p.PH && c < p.PW) sm[pl * plane_sz + r * p.PW + c] = x;
    }
    __syncthreads();
    const int osz = p.OUTH * p.OUTW, G = q.G, g = t & (G - 1), slot = t / G, R = 256 / G;
    float* op = p.out + p0 * osz;
    const float inv = 1.f / (float)osz;
    for (int pl0 = 0; pl0 < p.P; pl0 += R) {                       // (uniform trip count: the barriers below are taken by every thread)
        const int pl = pl0 + slot;
        const bool ok = pl < np;
        float sum = 0.f;
        if (ok) {
            const int ch = (int)((p0 + pl) % p.C);
            const float s = q.a1[ch], h = q.b1[ch];
            const float* wp = wl + pl * K * K;
            for (int o = g; o < osz; o += G) {
                const int oy = o / p.OUTW, ox = o - oy * p.OUTW;
                const float* base = sm + pl * plane_sz + (oy * p.os + p.by) * p.PW + ox * p.os + p.bx;
                float c0 = 0.f, c1 = 0.f;
#pragma unroll
                for (int i = 0; i < K; ++i)
#pragma unroll
                    for (int j = 0; j < K; ++j) {
                        if ((i * K + j) & 1) c1 = fmaf(base[i * p.PW + j], wp[i * K + j], c1);
                        else c0 = fmaf(base[i * p.PW + j], wp[i * K + j], c0);
                    }
                const float y = silu_f(fmaf(c0 + c1, s, h));
                op[(long)pl * osz + o] = y;
                sum += y;
            }
        }
        for (int o = (G < 64 ? G : 64) >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
        if (G <= 64) {
            if (ok && g == 0) q.pooled[p0 + pl] = sum * inv;
        } else {
            if ((t & 63) == 0) red[t >> 6] = sum;
            __syncthreads();
            if (ok && g == 0) {
                float tot = 0.f;
                for (int w = 0; w < G / 64; ++w) tot += red[(t >> 6) + w];
                q.pooled[p0 + pl] = tot * inv;
            }
            __syncthreads();
        }
    }
}

//  gate[b][c] = sigmoid(b2[c] + sum_j w2[c][j] * hidden[b][j]) on its own (the fused inference block multiplies it into the project conv's
//  operand instead of rewriting the expanded tensor).  16 lanes share one channel: a wave walks 4 contiguous rows of w2 (coalesced; one
//  thread per (b, c) read 64 different rows per load and ran 15 us), keeps its slice of the row in registers and loops over a chunk of
//  images; fixed butterfly over the 16 lanes.
constexpr int SE_GATE_IB = 8;          // images per workgroup row
__global__ __launch_bounds__(256) void se_gate_kernel(const float* __restrict__ hidden, const float* __restrict__ w2, const float* __restrict__ b2,
                                                     float* __restrict__ gate, int B, int C, int SQ) {
    const int t = threadIdx.x, sub = t & 15;
    const int c = blockIdx.x * 16 + (t >> 4);
    const bool ok = c < C;
    const float* wr = w2 + (long)(ok ? c : 0) * SQ;
    const int b0 = blockIdx.y * SE_GATE_IB, b1 = b0 + SE_GATE_IB < B ? b0 + SE_GATE_IB : B;
    const float bias = ok ? b2[c] : 0.f;
    if (SQ <= 128) {
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = (ok && sub + 16 * u < SQ) ? wr[sub + 16 * u] : 0.f;
        for (int b = b0; b < b1; ++b) {
            const float* hr = hidden + (long)b * SQ;
            float a0 = 0.f, a1 = 0.f;
#pragma unroll
            for (int u = 0; u < 8; u += 2) {
                a0 = fmaf(wv[u], sub + 16 * u < SQ ? hr[sub + 16 * u] : 0.f, a0);
                a1 = fmaf(wv[u + 1], sub + 16 * (u + 1) < SQ ? hr[sub + 16 * (u + 1)] : 0.f, a1);
            }
            float acc = a0 + a1;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) acc += __shfl_xor(acc, o, 64);
            if (ok && sub == 0) gate[(long)b * C + c] = 1.f / (1.f + __expf(-(acc + bias)));
        }
    } else {
        for (int b = b0; b < b1; ++b) {
            const float* hr = hidden + (long)b * SQ;
            float acc = 0.f;
            if (ok)
                for (int j = sub; j < SQ; j += 16) acc = fmaf(wr[j], hr[j], acc);
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) acc += __shfl_xor(acc, o, 64);
            if (ok && sub == 0) gate[(long)b * C + c] = 1.f / (1.f + __expf(-(acc + bias)));
        }
    }
}

// weight gradient, LDS-staged: workgroup (c, slice) stages P images' zero-padded x planes and dy planes per round; thread = one output pixel
template <int K>
__global__ __launch_bounds__(256) void dw_lds_wgrad_kernel(const DWParams p, float* __restrict__ ws, int splits, int PH, int PW, int P) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int c = blockIdx.x, sp = blockIdx.y, t = threadIdx.x;
    const int per_img = p.OH * p.OW, plane_sz = PH * PW, xsz = p.H * p.W;
    float* dyl = sm + P * plane_sz;
    const int b0 = (int)((long)p.B * sp / splits), b1 = (int)((long)p.B * (sp + 1) / splits);
    float acc[K * K];
#pragma unroll
    for (int q = 0; q < K * K; ++q) acc[q] = 0.f;
    for (int e = t; e < P * plane_sz; e += 256) sm[e] = 0.f;          // the halo stays zero for every round
    for (int br = b0; br < b1; br += P) {
        const int np = b1 - br < P ? b1 - br : P;
        __syncthreads();                                               // previous round's reads are done
        for (int e = t; e < np * xsz; e += 256) {
            const int pl = e / xsz, q = e - pl * xsz, u = q / p.W, v = q - u * p.W;
            const int r = u + p.pad_t, cc = v + p.pad_l;
            if (r < PH && cc < PW) sm[pl * plane_sz + r * PW + cc] = p.x[((long)(br + pl) * p.C + c) * xsz + q];
        }
        for (int e = t; e < np * per_img; e += 256) {
            const int pl = e / per_img, q = e - pl * per_img;
            dyl[e] = p.dy[((long)(br + pl) * p.C + c) * per_img + q];
        }
        __syncthreads();
        for (int e = t; e < np * per_img; e += 256) {
            const int pl = e / per_img, q = e - pl * per_img, oy = q / p.OW, ox = q - oy * p.OW;
            const float g = dyl[e];
            const float* base = sm + pl * plane_sz + oy * p.stride * PW + ox * p.stride;
#pragma unroll
            for (int i = 0; i < K; ++i)
#pragma unroll
                for (int j = 0; j < K; ++j) acc[i * K + j] = fmaf(g, base[i * PW + j], acc[i * K + j]);
        }
    }
    __shared__ float red[4][K * K];
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int q = 0; q < K * K; ++q) {
        float v = acc[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
        if (lane == 0) red[wave][q] = v;
    }
    __syncthreads();
    if (t < K * K) ws[((long)sp * p.C + c) * K * K + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}

__global__ void dw_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int n, int splits) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += ws[(long)s * n + i];
    dw[i] = v;
}

// inference BatchNorm (+ activation) on NCHW fp32: y = act(x * scale[c] + shift[c]); act 0 none, 1 SiLU, 2 ReLU.  One pass,
// float4 when the plane size allows (MIOpen's inference-BatchNorm kernel takes ~39 us per call whatever the tensor size,
// 115 calls per batch in the encoder / decoders).
template <int VEC>
__global__ __launch_bounds__(256) void affine_act_nchw_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const float* __restrict__ res,
                                                             float* __restrict__ y, long planes, int C, int hw, int act) {
    const int per = hw / VEC;
    const long total = planes * per;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long pl = idx / per;
        const int c = (int)(pl % C);
        const float s = scale[c], h = shift[c];
        float v[VEC];
        if (VEC == 4) {
            const floatx4 t = ((const floatx4*)x)[idx];
            v[0] = t[0]; v[1 % VEC] = t[1]; v[2 % VEC] = t[2]; v[3 % VEC] = t[3];
        } else {
            v[0] = x[idx];
        }
#pragma unroll
        for (int q = 0; q < VEC; ++q) {
            float u = fmaf(v[q], s, h);
            if (act == 1) u = u / (1.f + __expf(-u));
            else if (act == 2) u = fmaxf(u, 0.f);
            v[q] = u;
        }
        if (VEC == 4) {
            floatx4 t = {v[0], v[1 % VEC], v[2 % VEC], v[3 % VEC]};
            if (res) t += ((const floatx4*)res)[idx];          // the block's skip connection (MBConv: x = bn2(project(x)) + inputs)
            ((floatx4*)y)[idx] = t;
        } else {
            y[idx] = res ? v[0] + res[idx] : v[0];
        }
    }
}

// ---- the encoder's STEM at inference (round 6): conv3x3 stride 2 (static "same" padding) Cin <= 16 -> Cout <= 64 + the folded BatchNorm + SiLU in
// one pass on NCHW fp32 (smp EfficientNetEncoder.forward through mymodels.py:276: `_swish(_bn0(_conv_stem(x)))`).  MIOpen's solver search
// (benchmark mode of the tiled-inference path) settles on an asm kernel that takes ~0.6 ms per 128-tile batch for this 0.9-GFLOP layer
// (profiles/r05cd_predict_steady_kernel_stats.txt), followed by the affine pass.  Here: one thread per output pixel, all Cout accumulators
// in registers, the weights transposed into LDS as [ci][tap][co] and read as wave-uniform ds_read_b128 broadcasts, input taps straight from
// L1 (a 64 x 64 plane is 16 KB), outputs written plane by plane (coalesced along x).
template <int CO4>
__global__ __launch_bounds__(256) void stem_conv_eval_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, float* __restrict__ y, int B, int Cin, int H, int W,
                                                            int stride, int pt, int pl, int OH, int OW, int act) {
    constexpr int CO = CO4 * 4;
    extern __shared__ __attribute__((aligned(16))) float wl[];          // [Cin][9][CO]
    const int Cout = CO;
    for (int i = threadIdx.x; i < Cout * Cin * 9; i += 256) {
        const int co = i / (Cin * 9), r = i - co * Cin * 9;             // OIHW: r = ci * 9 + tap
        wl[r * CO + co] = w[i];
    }
    __syncthreads();
    const long total = (long)B * OH * OW;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int ox = (int)(idx % OW);
    const long t = idx / OW;
    const int oy = (int)(t % OH), b = (int)(t / OH);
    floatx4 acc[CO4];
#pragma unroll
    for (int q = 0; q < CO4; ++q) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};
    const int iy0 = oy * stride - pt, ix0 = ox * stride - pl;
    const float* xb = x + (long)b * Cin * H * W;
    for (int ci = 0; ci < Cin; ++ci) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int iy = iy0 + tap / 3, ix = ix0 + tap % 3;
            float v = 0.f;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) v = xb[((long)ci * H + iy) * W + ix];
            const floatx4* wq = (const floatx4*)(wl + (ci * 9 + tap) * CO);
#pragma unroll
            for (int q = 0; q < CO4; ++q) acc[q] += wq[q] * v;
        }
    }
    float* yb = y + ((long)b * Cout * OH + oy) * OW + ox;
#pragma unroll
    for (int q = 0; q < CO4; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int co = q * 4 + j;
            float u = fmaf(acc[q][j], scale[co], shift[co]);
            if (act == 1) u = u / (1.f + __expf(-u));
            else if (act == 2) u = fmaxf(u, 0.f);
            yb[(long)co * OH * OW] = u;
        }
}

// ---- squeeze-and-excitation of an MBConv block at inference (efficientnet_pytorch MBConvBlock.forward: avg-pool -> 1x1
// reduce -> swish -> 1x1 expand -> sigmoid -> scale), three launches instead of ~11 stock-op ones per block:
//  (1) inference BatchNorm + SiLU of the depthwise output WITH the per-plane mean (one wave per (b, c) plane, <= 1024 px)
__global__ __launch_bounds__(256) void affine_act_pool_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, float* __restrict__ y,
                                                             float* __restrict__ pooled, long planes, int C, int hw, int act) {
    const int lane = threadIdx.x & 63;
    const long pl = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pl >= planes) return;
    const int c = (int)(pl % C);
    const float s = scale[c], h = shift[c];
    const float* xp = x + pl * hw;
    float* yp = y + pl * hw;
    float sum = 0.f;
    for (int i = lane; i < hw; i += 64) {
        float u = fmaf(xp[i], s, h);
        if (act == 1) u = u / (1.f + __expf(-u));
        else if (act == 2) u = fmaxf(u, 0.f);
        yp[i] = u;
        sum += u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_down(sum, o, 64);
    if (lane == 0) pooled[pl] = sum / (float)hw;
}

//  (2) hidden[b][j] = swish(b1[j] + sum_c w1[j][c] * pooled[b][c]): one WAVE per (b, j) dot product (grid = B x SQ/4
//      workgroups: a per-b workgroup looping over j ran 63 us -- a serial chain of dependent global loads)
__global__ __launch_bounds__(256) void se_hidden_kernel(const float* __restrict__ pooled, const float* __restrict__ w1,
                                                       const float* __restrict__ b1, float* __restrict__ hidden, int C, int SQ) {
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const int j = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (j >= SQ) return;
    const float* wr = w1 + (long)j * C;
    const float* pr = pooled + (long)b * C;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int c = lane;
    for (; c + 192 < C; c += 256) {
        a0 = fmaf(wr[c], pr[c], a0);
        a1 = fmaf(wr[c + 64], pr[c + 64], a1);
        a2 = fmaf(wr[c + 128], pr[c + 128], a2);
        a3 = fmaf(wr[c + 192], pr[c + 192], a3);
    }
    for (; c < C; c += 64) a0 = fmaf(wr[c], pr[c], a0);
    float acc = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) {
        const float u = acc + b1[j];
        hidden[(long)b * SQ + j] = u / (1.f + __expf(-u));
    }
}

//  (3) y[plane] *= sigmoid(b2[c] + sum_j w2[c][j] * hidden[b][j]): one wave per (b, c) plane computes its own gate (row c of
//      the expand weight is read coalesced by the wave) and scales the plane in place
__global__ __launch_bounds__(256) void se_gate_scale_kernel(float* __restrict__ y, const float* __restrict__ hidden,
                                                           const float* __restrict__ w2, const float* __restrict__ b2, long planes,
                                                           int C, int SQ, int hw) {
    const int lane = threadIdx.x & 63;
    const long pl = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pl >= planes) return;
    const int c = (int)(pl % C);
    const long b = pl / C;
    const float* wr = w2 + (long)c * SQ;
    const float* hr = hidden + b * SQ;
    float acc = 0.f;
    for (int j = lane; j < SQ; j += 64) acc = fmaf(wr[j], hr[j], acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    const float g = 1.f / (1.f + __expf(-(acc + b2[c])));
    float* yp = y + pl * hw;
    for (int i = lane; i < hw; i += 64) yp[i] *= g;
}

// small planes (hw = 1 .. 32, a power of two): a wave per plane would be mostly empty lanes (4 of 64 at 2x2, 172 k waves per call);
// planes are contiguous, so a wave takes 64 consecutive elements = 64 / hw whole planes and reduces inside hw-lane segments
__global__ __launch_bounds__(256) void affine_act_pool_small_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                                   const float* __restrict__ shift, float* __restrict__ y,
                                                                   float* __restrict__ pooled, long planes, int C, int hw, int act) {
    const int lane = threadIdx.x & 63;
    const long idx = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + lane;
    const long pl = idx / hw;
    float u = 0.f;
    if (pl < planes) {
        const int c = (int)(pl % C);
        u = fmaf(x[idx], scale[c], shift[c]);
        if (act == 1) u = u / (1.f + __expf(-u));
        else if (act == 2) u = fmaxf(u, 0.f);
        y[idx] = u;
    }
    for (int o = 1; o < hw; o <<= 1) u += __shfl_xor(u, o, 64);
    if (pl < planes && (lane & (hw - 1)) == 0) pooled[pl] = u / (float)hw;
}

// gate + scale for planes of 4 / 16 / 64 / 256 elements: a workgroup takes 1024 consecutive elements (a float4 per thread); the hw / 4
// threads that hold one plane compute its gate together (same reason)
__global__ __launch_bounds__(256) void se_gate_scale_quad_kernel(float* __restrict__ y, const float* __restrict__ hidden,
                                                                const float* __restrict__ w2, const float* __restrict__ b2, long planes,
                                                                int C, int SQ, int hw) {
    const int t = threadIdx.x, G = hw >> 2, sub = t & (G - 1);
    const long e = (long)blockIdx.x * 1024 + 4 * t;
    const long pl = e / hw;
    const bool ok = pl < planes;
    const int c = ok ? (int)(pl % C) : 0;
    const long b = ok ? pl / C : 0;
    const float* wr = w2 + (long)c * SQ;
    const float* hr = hidden + b * SQ;
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    int j = sub;
    for (; j + 3 * G < SQ; j += 4 * G) {
#pragma unroll
        for (int u = 0; u < 4; ++u) a[u] = fmaf(wr[j + u * G], hr[j + u * G], a[u]);
    }
    for (; j < SQ; j += G) a[0] = fmaf(wr[j], hr[j], a[0]);
    float acc = (a[0] + a[1]) + (a[2] + a[3]);
    for (int o = 1; o < G; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (!ok) return;
    const float g = 1.f / (1.f + __expf(-(acc + b2[c])));
    floatx4 v = *(floatx4*)(y + e);
    v *= g;
    *(floatx4*)(y + e) = v;
}

int check(const DWParams& p, int K, const char* what) {
    SRBH_REQUIRE(K == 3 || K == 5, "%s: kernel size must be 3 or 5", what);
    SRBH_REQUIRE(p.stride == 1 || p.stride == 2, "%s: stride must be 1 or 2", what);
    SRBH_REQUIRE(p.B > 0 && p.C > 0 && p.H > 0 && p.W > 0 && p.OH > 0 && p.OW > 0, "%s: bad shape", what);
    SRBH_REQUIRE(p.pad_t >= 0 && p.pad_l >= 0 && p.pad_t < K && p.pad_l < K, "%s: bad padding", what);
    // bottom/right padding implied by the output size must be a zero pad, not a crop
    SRBH_REQUIRE((p.OH - 1) * p.stride - p.pad_t + K >= p.H - (p.stride - 1) && (p.OW - 1) * p.stride - p.pad_l + K >= p.W - (p.stride - 1),
                 "%s: output size does not cover the input", what);
    return SRBH_OK;
}

constexpr size_t DW_LDS_MAX = 60 * 1024;
#ifndef DW_LDS_FORMS
#define DW_LDS_FORMS 1          // (0: the one-thread-per-output kernels, for same-box A/B builds)
#endif
// (round 6: up to 8 rounds of planes per workgroup at the tiled prediction's 49 152 planes -- fewer, longer workgroups -- measured: the encoder
//  graph 4.67 ms at one round, 4.57 at two, 4.61 at four, 4.76 at eight (profiles/r06ae_predict_parts.txt): not kept)
int lds_planes(int out_px) {     // planes a workgroup stages per round: ~256 outputs
    int P = 256 / (out_px > 0 ? out_px : 1);
    return P < 1 ? 1 : (P > 64 ? 64 : P);
}
int grid_for(long n) {
    const long b = (n + 255) / 256;
    return (int)(b < 8192 ? (b > 0 ? b : 1) : 8192);
}
}  // namespace

extern "C" int srbh_dwconv_fwd(const float* x, const float* w, float* y, int B, int C, int H, int W, int K, int stride, int pad_t,
                               int pad_l, int OH, int OW, void* stream) {
    SRBH_REQUIRE(x && w && y, "srbh_dwconv_fwd: null pointer");
    DWParams p{x, w, nullptr, y, B, C, H, W, OH, OW, stride, pad_t, pad_l};
    if (int rc = check(p, K, "srbh_dwconv_fwd")) return rc;
    {
        DWLds q;
        q.src = x; q.w = w; q.out = y; q.planes = (long)B * C; q.C = C; q.SH = H; q.SW = W; q.OUTH = OH; q.OUTW = OW;
        q.PH = (OH - 1) * stride + K; q.PW = (OW - 1) * stride + K; q.ss = 1; q.soy = pad_t; q.sox = pad_l; q.os = stride; q.by = 0; q.bx = 0;
        q.P = lds_planes(OH * OW);
        const size_t lds = (size_t)q.P * (q.PH * q.PW + K * K) * 4;
        if (lds <= DW_LDS_MAX && DW_LDS_FORMS) {
            const dim3 grid((unsigned)((q.planes + q.P - 1) / q.P));
            if (K == 3) hipLaunchKernelGGL((dw_lds_kernel<3, 0>), grid, dim3(256), lds, (hipStream_t)stream, q);
            else hipLaunchKernelGGL((dw_lds_kernel<5, 0>), grid, dim3(256), lds, (hipStream_t)stream, q);
            SRBH_HIP(hipGetLastError());
            return SRBH_OK;
        }
    }
    const int g = grid_for((long)B * C * OH * OW);
    if (K == 3)
        hipLaunchKernelGGL(dw_fwd_kernel<3>, dim3(g), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(dw_fwd_kernel<5>, dim3(g), dim3(256), 0, (hipStream_t)stream, p);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

static bool dw_eval_geometry(DWLds& q, int B, int C, int H, int W, int K, int stride, int pad_t, int pad_l, int OH, int OW, size_t& lds, int& G) {
    q.planes = (long)B * C; q.C = C; q.SH = H; q.SW = W; q.OUTH = OH; q.OUTW = OW;
    q.PH = (OH - 1) * stride + K; q.PW = (OW - 1) * stride + K; q.ss = 1; q.soy = pad_t; q.sox = pad_l; q.os = stride; q.by = 0; q.bx = 0;
    q.P = lds_planes(OH * OW);
    lds = (size_t)q.P * (q.PH * q.PW + K * K) * 4;
    G = 1;
    while (G < OH * OW && G < 256) G <<= 1;
    return lds <= DW_LDS_MAX && (K == 3 || K == 5);
}

extern "C" int srbh_dwconv_eval_supported(int B, int C, int H, int W, int K, int stride, int pad_t, int pad_l, int OH, int OW) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || (stride != 1 && stride != 2)) return 0;
    DWLds q;
    size_t lds;
    int G;
    return dw_eval_geometry(q, B, C, H, W, K, stride, pad_t, pad_l, OH, OW, lds, G) ? 1 : 0;
}

extern "C" int srbh_dwconv_eval_fwd(const float* x, const float* w, const float* pre_scale, const float* pre_shift, const float* scale,
                                    const float* shift, float* y, float* pooled, int B, int C, int H, int W, int K, int stride, int pad_t,
                                    int pad_l, int OH, int OW, void* stream) {
    SRBH_REQUIRE(x && w && scale && shift && y && pooled, "srbh_dwconv_eval_fwd: null pointer");
    SRBH_REQUIRE((pre_scale == nullptr) == (pre_shift == nullptr), "srbh_dwconv_eval_fwd: pre_scale and pre_shift go together");
    DWParams p{x, w, nullptr, y, B, C, H, W, OH, OW, stride, pad_t, pad_l};
    if (int rc = check(p, K, "srbh_dwconv_eval_fwd")) return rc;
    DWLds q;
    size_t lds;
    DWEval e{pre_scale, pre_shift, scale, shift, pooled, 1};
    SRBH_REQUIRE(dw_eval_geometry(q, B, C, H, W, K, stride, pad_t, pad_l, OH, OW, lds, e.G), "srbh_dwconv_eval_fwd: plane too large for the "
                 "LDS-staged form (srbh_dwconv_eval_supported)");
    q.src = x; q.w = w; q.out = y;
    const dim3 grid((unsigned)((q.planes + q.P - 1) / q.P));
    if (K == 3) hipLaunchKernelGGL(dw_lds_eval_kernel<3>, grid, dim3(256), lds, (hipStream_t)stream, q, e);
    else hipLaunchKernelGGL(dw_lds_eval_kernel<5>, grid, dim3(256), lds, (hipStream_t)stream, q, e);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_se_gate(const float* hidden, const float* w2, const float* b2, float* gate, int B, int C, int SQ, void* stream) {
    SRBH_REQUIRE(hidden && w2 && b2 && gate, "srbh_se_gate: null pointer");
    SRBH_REQUIRE(B > 0 && C > 0 && SQ > 0, "srbh_se_gate: bad shape");
    SRBH_REQUIRE((B + SE_GATE_IB - 1) / SE_GATE_IB <= 65535, "srbh_se_gate: batch too large");
    hipLaunchKernelGGL(se_gate_kernel, dim3((unsigned)((C + 15) / 16), (unsigned)((B + SE_GATE_IB - 1) / SE_GATE_IB)), dim3(256), 0,
                       (hipStream_t)stream, hidden, w2, b2, gate, B, C, SQ);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_dwconv_bwd_data(const float* dy, const float* w, float* dx, int B, int C, int H, int W, int K, int stride,
                                    int pad_t, int pad_l, int OH, int OW, void* stream) {
    SRBH_REQUIRE(dy && w && dx, "srbh_dwconv_bwd_data: null pointer");
    DWParams p{nullptr, w, dy, dx, B, C, H, W, OH, OW, stride, pad_t, pad_l};
    if (int rc = check(p, K, "srbh_dwconv_bwd_data")) return rc;
    {
        DWLds q;
        q.src = dy; q.w = w; q.out = dx; q.planes = (long)B * C; q.C = C; q.SH = OH; q.SW = OW; q.OUTH = H; q.OUTW = W;
        // (rows / columns of x that no output covers -- possible at stride 2 -- read the zero slack and get dx = 0)
        q.PH = (OH - 1) * stride + 2 * (K - 1) + stride + 1; q.PW = (OW - 1) * stride + 2 * (K - 1) + stride + 1;
        q.ss = stride; q.soy = K - 1; q.sox = K - 1; q.os = 1; q.by = pad_t + K - 1; q.bx = pad_l + K - 1;
        q.P = lds_planes(H * W);
        const size_t lds = (size_t)q.P * (q.PH * q.PW + K * K) * 4;
        if (lds <= DW_LDS_MAX && DW_LDS_FORMS && H - 1 + pad_t + K - 1 < q.PH && W - 1 + pad_l + K - 1 < q.PW) {
            const dim3 grid((unsigned)((q.planes + q.P - 1) / q.P));
            if (K == 3) hipLaunchKernelGGL((dw_lds_kernel<3, 1>), grid, dim3(256), lds, (hipStream_t)stream, q);
            else hipLaunchKernelGGL((dw_lds_kernel<5, 1>), grid, dim3(256), lds, (hipStream_t)stream, q);
            SRBH_HIP(hipGetLastError());
            return SRBH_OK;
        }
    }
    const int g = grid_for((long)B * C * H * W);
    if (K == 3)
        hipLaunchKernelGGL(dw_bwd_data_kernel<3>, dim3(g), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(dw_bwd_data_kernel<5>, dim3(g), dim3(256), 0, (hipStream_t)stream, p);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_dwconv_bwd_weight_splits(int B, int C) {
    int s = (1024 + C - 1) / (C > 0 ? C : 1);
    if (s > B) s = B;
    return s < 1 ? 1 : s;
}

extern "C" int srbh_dwconv_bwd_weight(const float* x, const float* dy, float* dw, float* ws, int B, int C, int H, int W, int K,
                                      int stride, int pad_t, int pad_l, int OH, int OW, void* stream) {
    SRBH_REQUIRE(x && dy && dw && ws, "srbh_dwconv_bwd_weight: null pointer");
    DWParams p{x, nullptr, dy, dw, B, C, H, W, OH, OW, stride, pad_t, pad_l};
    if (int rc = check(p, K, "srbh_dwconv_bwd_weight")) return rc;
    const int splits = srbh_dwconv_bwd_weight_splits(B, C);
    const int PH = (OH - 1) * stride + K, PW = (OW - 1) * stride + K, P = lds_planes(OH * OW);
    const size_t lds = (size_t)P * (PH * PW + OH * OW) * 4;
    float* const part = splits == 1 ? dw : ws;                       // (one slice: its sums ARE the gradient, no reduce launch)
    // (the staged form pays for 5x5 and for small planes; a 3x3 over >= 256-pixel planes is as fast from L1: measured per shape)
    if (lds <= DW_LDS_MAX && DW_LDS_FORMS && (K == 5 || OH * OW < 256)) {
        if (K == 3) hipLaunchKernelGGL(dw_lds_wgrad_kernel<3>, dim3(C, splits), dim3(256), lds, (hipStream_t)stream, p, part, splits, PH, PW, P);
        else hipLaunchKernelGGL(dw_lds_wgrad_kernel<5>, dim3(C, splits), dim3(256), lds, (hipStream_t)stream, p, part, splits, PH, PW, P);
    } else if (K == 3)
        hipLaunchKernelGGL(dw_bwd_weight_kernel<3>, dim3(C, splits), dim3(256), 0, (hipStream_t)stream, p, part, splits);
    else
        hipLaunchKernelGGL(dw_bwd_weight_kernel<5>, dim3(C, splits), dim3(256), 0, (hipStream_t)stream, p, part, splits);
    SRBH_HIP(hipGetLastError());
    if (splits == 1) return SRBH_OK;
    const int n = C * K * K;
    hipLaunchKernelGGL(dw_reduce_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, ws, dw, n, splits);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

static int affine_act_impl(const float* x, const float* scale, const float* shift, const float* res, float* y, int B, int C, int HW,
                           int act, void* stream) {
    SRBH_REQUIRE(x && scale && shift && y, "srbh_affine_act_nchw: null pointer");
    SRBH_REQUIRE(B > 0 && C > 0 && HW > 0 && act >= 0 && act <= 2, "srbh_affine_act_nchw: bad arguments");
    const long planes = (long)B * C;
    const bool v4 = (HW % 4 == 0) && (((uintptr_t)x | (uintptr_t)y | (uintptr_t)res) % 16 == 0);
    const int g = grid_for(planes * (v4 ? HW / 4 : HW));
    if (v4)
        hipLaunchKernelGGL(affine_act_nchw_kernel<4>, dim3(g), dim3(256), 0, (hipStream_t)stream, x, scale, shift, res, y, planes, C, HW, act);
    else
        hipLaunchKernelGGL(affine_act_nchw_kernel<1>, dim3(g), dim3(256), 0, (hipStream_t)stream, x, scale, shift, res, y, planes, C, HW, act);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

// (56 / 64 output channels -- the stems of efficientnet-b6 / b7 -- would need a second accumulator pass: one thread holding 64 accumulators spills)
extern "C" int srbh_stem_conv_eval_supported(int Cin, int Cout, int K) { return K == 3 && Cin > 0 && Cin <= 16 && (Cout == 32 || Cout == 40 || Cout == 48); }

/* act(conv3x3(x, w, stride, static padding top = pt / left = pl, zeros beyond) * scale[c] + shift[c]) on NCHW fp32; w: OIHW; act 0 none / 1 SiLU / 2 ReLU */
extern "C" int srbh_stem_conv_eval(const float* x, const float* w, const float* scale, const float* shift, float* y, int B, int Cin, int H, int W,
                                   int Cout, int stride, int pt, int pl, int OH, int OW, int act, void* stream) {
    SRBH_REQUIRE(x && w && scale && shift && y, "srbh_stem_conv_eval: null pointer");
    SRBH_REQUIRE(B > 0 && H > 0 && W > 0 && OH > 0 && OW > 0 && stride >= 1 && pt >= 0 && pl >= 0 && act >= 0 && act <= 2 && srbh_stem_conv_eval_supported(Cin, Cout, 3),
                 "srbh_stem_conv_eval: bad arguments (3x3, Cin <= 16, Cout in {32, 40, 48})");
    const long total = (long)B * OH * OW;
    const dim3 grid((unsigned)((total + 255) / 256));
    const size_t lds = (size_t)Cin * 9 * Cout * sizeof(float);
    hipStream_t st = (hipStream_t)stream;
#define SRBH_STEM(C4_) hipLaunchKernelGGL((stem_conv_eval_kernel<C4_>), grid, dim3(256), lds, st, x, w, scale, shift, y, B, Cin, H, W, stride, pt, pl, OH, OW, act)
    switch (Cout) { case 32: SRBH_STEM(8); break; case 40: SRBH_STEM(10); break; default: SRBH_STEM(12); }
#undef SRBH_STEM
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_affine_act_nchw(const float* x, const float* scale, const float* shift, float* y, int B, int C, int HW, int act,
                                    void* stream) {
    return affine_act_impl(x, scale, shift, nullptr, y, B, C, HW, act, stream);
}

extern "C" int srbh_affine_act_add_nchw(const float* x, const float* scale, const float* shift, const float* res, float* y, int B, int C,
                                        int HW, int act, void* stream) {
    SRBH_REQUIRE(res, "srbh_affine_act_add_nchw: null residual");
    return affine_act_impl(x, scale, shift, res, y, B, C, HW, act, stream);
}

extern "C" int srbh_affine_act_pool_nchw(const float* x, const float* scale, const float* shift, float* y, float* pooled, int B, int C,
                                         int HW, int act, void* stream) {
    SRBH_REQUIRE(x && scale && shift && y && pooled, "srbh_affine_act_pool_nchw: null pointer");
    SRBH_REQUIRE(B > 0 && C > 0 && HW > 0 && act >= 0 && act <= 2, "srbh_affine_act_pool_nchw: bad arguments");
    const long planes = (long)B * C;
    if (HW < 64 && (HW & (HW - 1)) == 0)
        hipLaunchKernelGGL(affine_act_pool_small_kernel, dim3((unsigned)((planes * HW + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, scale,
                           shift, y, pooled, planes, C, HW, act);
    else
        hipLaunchKernelGGL(affine_act_pool_kernel, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, y,
                           pooled, planes, C, HW, act);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_se_hidden(const float* pooled, const float* w1, const float* b1, float* hidden, int B, int C, int SQ,
                              void* stream) {
    SRBH_REQUIRE(pooled && w1 && b1 && hidden, "srbh_se_hidden: null pointer");
    SRBH_REQUIRE(B > 0 && C > 0 && SQ > 0 && SQ <= 65535 * 4, "srbh_se_hidden: bad shape");
    hipLaunchKernelGGL(se_hidden_kernel, dim3(B, (SQ + 3) / 4), dim3(256), 0, (hipStream_t)stream, pooled, w1, b1, hidden, C, SQ);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_se_gate_scale(float* y, const float* hidden, const float* w2, const float* b2, int B, int C, int SQ, int HW,
                                  void* stream) {
    SRBH_REQUIRE(y && hidden && w2 && b2, "srbh_se_gate_scale: null pointer");
    SRBH_REQUIRE(B > 0 && C > 0 && SQ > 0 && HW > 0, "srbh_se_gate_scale: bad shape");
    const long planes = (long)B * C;
    if ((HW == 4 || HW == 16 || HW == 64 || HW == 256) && ((uintptr_t)y & 15) == 0)
        hipLaunchKernelGGL(se_gate_scale_quad_kernel, dim3((unsigned)((planes * HW + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, y, hidden,
                           w2, b2, planes, C, SQ, HW);
    else
        hipLaunchKernelGGL(se_gate_scale_kernel, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, (hipStream_t)stream, y, hidden, w2, b2,
                           planes, C, SQ, HW);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}
