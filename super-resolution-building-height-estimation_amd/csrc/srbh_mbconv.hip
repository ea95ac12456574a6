// srbh_mbconv.hip -- TRAINING-mode BatchNorm + activation and squeeze-and-excitation of the EfficientNet MBConv blocks and the
// U-Net decoder blocks (fp32 NCHW), forward and backward.
//
// Why it exists: the encoder / decoders the reference instantiates (mymodels.py:242-258, forward at 276-287) are stock ops in this
// build, and at 64x64 tiles their training step is ~2 070 tiny launches (16.3 ms of a 44 ms step, round-2 VERDICT item 4): per
// MBConv block BatchNorm + SiLU, the pooled squeeze-excite branch (avg-pool, two 1x1 convs with their bias adds, SiLU, sigmoid,
// scale) and BatchNorm + drop-connect + skip are ~14 launches forward and ~30 backward, each moving well under a megabyte.  These
// are HBM/L2-resident element-wise + reduction passes; nothing here is a GEMM.
//
// Layout trick: below 32x32 a channel plane is at most 256 floats, so ONE workgroup owns a group of adjacent channels whose planes
// form a contiguous run of RW = 64 or 256 floats per image (CPW = RW / HW channels), holds all B images of that run in LDS
// (B * RW floats: 16 KB at B = 64, RW = 64) and does the whole training-mode BatchNorm in one launch: exact two-pass statistics from
// LDS, running-statistics update, normalise + activation (+ plane means for squeeze-excite, + drop-connect and skip connection),
// ONE global read of x and one write of y.  The backward has the same shape: dz = dy * act'(z) cached in LDS, the two channel sums,
// then dx.  Planes of 32x32 and larger (6 encoder and 8 decoder layers) stay on the stock ops: the host checks
// srbh_bn_act_train_supported.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "srbh.h"
#include "srbh_internal.h"

namespace {
using namespace srbh;

constexpr int MAX_LDS_B = 160 * 1024 - 4096;
constexpr int NW = 16;               // waves per workgroup of the BatchNorm kernels (1024 threads)
constexpr int U = 4;                 // images per wave and round of global loads (16 waves x 4 = the whole batch of 64 in ONE round)

// (v_rcp_f32: 1 ulp; these kernels are bound by the instruction stream of one wave per SIMD, an IEEE division is ~10 instructions)
__device__ __forceinline__ float sigmoidf(float z) { return __builtin_amdgcn_rcpf(1.f + __expf(-z)); }
template <int ACT>
__device__ __forceinline__ float act_f(float z) {
    if (ACT == 1) return z * sigmoidf(z);
    if (ACT == 2) return fmaxf(z, 0.f);
    return z;
}
template <int ACT>
__device__ __forceinline__ float act_grad(float z) {
    if (ACT == 1) {
        const float s = sigmoidf(z);
        return s * (1.f + z * (1.f - s));
    }
    if (ACT == 2) return z > 0.f ? 1.f : 0.f;
    return 1.f;
}

// sum over the lanes that share a channel (a run of `seg` = min(HW, 64) consecutive lanes, seg a power of two), then over the NW waves
// through `red`; every thread returns the total of ITS channel.  RW == 64: a wave spans the CPW channels of the workgroup (channel
// of a lane = lane / HW); RW == 256: the whole workgroup is one channel.  Deterministic order.
__device__ __forceinline__ float channel_sum(float v, int seg, int slot, float (*red)[64], int wave, int lane) {
    for (int o = 1; o < seg; o <<= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();                          // `red` may still be read from the previous reduction
    if ((lane & (seg - 1)) == 0) red[wave][slot] = v;
    __syncthreads();
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < NW; w += 2) {
        a += red[w][slot];
        b += red[w + 1][slot];
    }
    return a + b;
}

typedef float floatx4 __attribute__((ext_vector_type(4)));
template <int VEC> struct Vt { float v[VEC]; };
template <int VEC> __device__ __forceinline__ Vt<VEC> ldv(const float* p) {
    Vt<VEC> r;
    if (VEC == 4) {
        const floatx4 t = *(const floatx4*)p;
        r.v[0] = t[0]; r.v[1 % VEC] = t[1]; r.v[2 % VEC] = t[2]; r.v[3 % VEC] = t[3];
    } else {
        r.v[0] = *p;
    }
    return r;
}
template <int VEC> __device__ __forceinline__ void stv(float* p, const Vt<VEC>& r) {
    if (VEC == 4) *(floatx4*)p = floatx4{r.v[0], r.v[1 % VEC], r.v[2 % VEC], r.v[3 % VEC]};
    else *p = r.v[0];
}
template <int VEC> __device__ __forceinline__ Vt<VEC> zerov() {
    Vt<VEC> r;
#pragma unroll
    for (int k = 0; k < VEC; ++k) r.v[k] = 0.f;
    return r;
}

// Geometry shared by the forward and the backward kernel.  A workgroup owns a run of 64 * VEC contiguous floats per image: VEC = 1 --
// CPW = 64 / HW adjacent channels of HW <= 64 elements; VEC = 4 -- one channel of 256 elements, a float4 per lane.  Wave w takes the
// images w, w + NW, ...; a lane keeps the same position of the run for every image.
struct FwdP {
    const float* x; float* y;
    const float* gamma; const float* beta;
    float* running_mean; float* running_var;
    float* save_mean; float* save_invstd;
    float* pooled; const float* res; const float* drop;
    float momentum, eps;
    int B, C, HW;
};

template <int ACT, int VEC>
__global__ __launch_bounds__(64 * NW) void bn_act_train_fwd_kernel(const FwdP p) {
    extern __shared__ __attribute__((aligned(16))) float cache[];      // [B][64 * VEC]
    __shared__ float red[NW][64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int HW = p.HW, RW = 64 * VEC;
    const int seg = (VEC == 4 || HW >= 64) ? 64 : HW;
    const int slot = VEC == 4 ? 0 : lane / HW;
    const int c0 = blockIdx.x * (RW / HW), c = c0 + slot;
    const bool cok = c < p.C;
    const float invN = 1.f / ((float)p.B * (float)HW);
    const long off = (long)c0 * HW + lane * VEC, bstride = (long)p.C * HW;
    // These kernels are LATENCY chains, not bandwidth: a workgroup moves 16-64 KB.  Every global pass is therefore issued U images at
    // a time (a plain loop leaves one load in flight per lane: measured 12-16 us per launch), and everything the later phases
    // need -- affine parameters, running statistics, the first round of the skip connection -- is requested before the reductions.
    const float gam = cok ? p.gamma[c] : 0.f, bet = cok ? p.beta[c] : 0.f;
    const float rm0 = (cok && p.running_mean) ? p.running_mean[c] : 0.f, rv0 = (cok && p.running_mean) ? p.running_var[c] : 0.f;
    float s = 0.f;
    for (int b0 = wave; b0 < p.B; b0 += NW * U) {
        Vt<VEC> v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + NW * u;
            v[u] = (cok && b < p.B) ? ldv<VEC>(p.x + b * bstride + off) : zerov<VEC>();
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + NW * u;
            if (b < p.B) stv<VEC>(cache + (b * 64 + lane) * VEC, v[u]);
#pragma unroll
            for (int k = 0; k < VEC; ++k) s += v[u].v[k];
        }
    }
    Vt<VEC> r0[U];
    float dr0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int b = wave + NW * u;
        r0[u] = (p.res && cok && b < p.B) ? ldv<VEC>(p.res + b * bstride + off) : zerov<VEC>();
        dr0[u] = (p.drop && b < p.B) ? p.drop[b] : 1.f;
    }
    const float mean = channel_sum(s, seg, slot, red, wave, lane) * invN;
    float q = 0.f;
    for (int b = wave; b < p.B; b += NW) {
        const Vt<VEC> v = ldv<VEC>(cache + (b * 64 + lane) * VEC);
#pragma unroll
        for (int k = 0; k < VEC; ++k) q = fmaf(v.v[k] - mean, v.v[k] - mean, q);
    }
    const float var = channel_sum(q, seg, slot, red, wave, lane) * invN;       // biased, as F.batch_norm normalises
    const float invstd = 1.f / sqrtf(var + p.eps);
    if (cok && wave == 0 && (lane & (seg - 1)) == 0) {
        p.save_mean[c] = mean;
        p.save_invstd[c] = invstd;
        if (p.running_mean) {
            const float n = (float)p.B * (float)HW;
            p.running_mean[c] = (1.f - p.momentum) * rm0 + p.momentum * mean;
            p.running_var[c] = (1.f - p.momentum) * rv0 + p.momentum * (n > 1.f ? var * n / (n - 1.f) : var);
        }
    }
    const float scale = gam * invstd, shift = bet - mean * scale;
    for (int b0 = wave; b0 < p.B; b0 += NW * U) {
        Vt<VEC> r[U];
        float dr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + NW * u;
            if (b0 == wave) {
                r[u] = r0[u];
                dr[u] = dr0[u];
            } else {
                r[u] = (p.res && cok && b < p.B) ? ldv<VEC>(p.res + b * bstride + off) : zerov<VEC>();
                dr[u] = (p.drop && b < p.B) ? p.drop[b] : 1.f;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + NW * u;
            if (b >= p.B) continue;                                // (uniform per wave: b depends on the wave only)
            Vt<VEC> a = ldv<VEC>(cache + (b * 64 + lane) * VEC);
            float w = 0.f;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                a.v[k] = fmaf(act_f<ACT>(fmaf(a.v[k], scale, shift)), dr[u], r[u].v[k]);
                w += a.v[k];
            }
            if (cok) stv<VEC>(p.y + b * bstride + off, a);
            if (p.pooled) {
                for (int o = 1; o < seg; o <<= 1) w += __shfl_xor(w, o, 64);
                if (cok && (lane & (seg - 1)) == 0) p.pooled[(long)b * p.C + c] = w / (float)HW;
            }
        }
    }
}

struct BwdP {
    const float* dy; const float* x;
    const float* gamma; const float* beta; const float* save_mean; const float* save_invstd;
    const float* gate; const float* dpooled; const float* drop;
    float* dx; float* dgamma; float* dbeta;
    int B, C, HW;
};

template <int ACT, int VEC>
__global__ __launch_bounds__(64 * NW) void bn_act_train_bwd_kernel(const BwdP p) {
    extern __shared__ __attribute__((aligned(16))) float cache[];      // [B][64 * VEC] dz | [B][64 * VEC] xhat
    __shared__ float red[NW][64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int HW = p.HW, RW = 64 * VEC;
    const int seg = (VEC == 4 || HW >= 64) ? 64 : HW;
    const int slot = VEC == 4 ? 0 : lane / HW;
    const int c0 = blockIdx.x * (RW / HW), c = c0 + slot;
    const bool cok = c < p.C;
    const float invN = 1.f / ((float)p.B * (float)HW), invHW = 1.f / (float)HW;
    const float mean = cok ? p.save_mean[c] : 0.f, invstd = cok ? p.save_invstd[c] : 0.f;
    const float g = cok ? p.gamma[c] : 0.f, be = cok ? p.beta[c] : 0.f;
    const long off = (long)c0 * HW + lane * VEC, bstride = (long)p.C * HW;
    float* const xhc = cache + p.B * 64 * VEC;
    float s1 = 0.f, s2 = 0.f;
    for (int b0 = wave; b0 < p.B; b0 += NW * U) {
        Vt<VEC> xv[U], dv[U];
        float gt[U], dpo[U], dr[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + NW * u;
            const bool ok = cok && b < p.B;
            xv[u] = ok ? ldv<VEC>(p.x + b * bstride + off) : zerov<VEC>();
            dv[u] = ok ? ldv<VEC>(p.dy + b * bstride + off) : zerov<VEC>();
            gt[u] = (ok && p.gate) ? p.gate[(long)b * p.C + c] : 1.f;
            dpo[u] = (ok && p.gate) ? p.dpooled[(long)b * p.C + c] * invHW : 0.f;
            dr[u] = (ok && p.drop) ? p.drop[b] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0 + NW * u;
            if (b >= p.B) continue;
            Vt<VEC> dz, xh;
#pragma unroll
            for (int k = 0; k < VEC; ++k) {
                xh.v[k] = (xv[u].v[k] - mean) * invstd;
                const float d = fmaf(dv[u].v[k], gt[u], dpo[u]) * dr[u];
                dz.v[k] = cok ? d * act_grad<ACT>(fmaf(xh.v[k], g, be)) : 0.f;
                s2 = fmaf(dz.v[k], xh.v[k], s2);
                s1 += dz.v[k];
            }
            stv<VEC>(cache + (b * 64 + lane) * VEC, dz);
            stv<VEC>(xhc + (b * 64 + lane) * VEC, xh);
        }
    }
    const float sum_dz = channel_sum(s1, seg, slot, red, wave, lane);
    const float sum_dzx = channel_sum(s2, seg, slot, red, wave, lane);
    if (cok && wave == 0 && (lane & (seg - 1)) == 0) {
        p.dbeta[c] = sum_dz;
        p.dgamma[c] = sum_dzx;
    }
    if (!p.dx) return;
    const float k1 = sum_dz * invN, k2 = sum_dzx * invN, gi = g * invstd;
    for (int b = wave; b < p.B; b += NW) {
        if (!cok) continue;
        Vt<VEC> dz = ldv<VEC>(cache + (b * 64 + lane) * VEC);
        const Vt<VEC> xh = ldv<VEC>(xhc + (b * 64 + lane) * VEC);
#pragma unroll
        for (int k = 0; k < VEC; ++k) dz.v[k] = gi * (dz.v[k] - k1 - xh.v[k] * k2);
        stv<VEC>(p.dx + b * bstride + off, dz);
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The MIDDLE of an MBConv block as ONE launch per direction (round 4; efficientnet_pytorch MBConvBlock.forward between the expand and
// the project conv, as smp's encoder runs it at mymodels.py:276): BatchNorm0 + SiLU -> depthwise KxK (stride 1, "same" padding) ->
// BatchNorm1 + SiLU (+ the plane means squeeze-excite pools).  Everything in there is per CHANNEL, and in the geometry above a workgroup
// holds ALL B images of its channels in LDS -- so the three launches of the forward (bn_act, depthwise, bn_act + pool) and the four of
// the backward (bn_act backward with the excite gate, depthwise data and weight gradient, bn_act backward) need nothing from another
// workgroup: one read of the expand conv's output, one write each of the depthwise output (kept for the backward) and of y; the
// backward reads both back, recomputes SiLU(bn0(.)) instead of keeping it, and writes only the gradient of the expand conv's output.
// Planes of 2x2, 4x4 and 8x8 (blocks 6..31 of EfficientNet-B4 at 64x64 tiles, without the three stride-2 blocks).
struct MidFwdP {
    const float* e_pre; const float* wdw;
    const float* g0; const float* b0; float* rm0; float* rv0; float* mean0; float* invstd0;
    const float* g1; const float* b1; float* rm1; float* rv1; float* mean1; float* invstd1;
    float* d_pre; float* y; float* pooled;
    float mom0, eps0, mom1, eps1;
    int B, C, HW, logw;
};

// two-pass statistics of this thread's channel over cache[b][lane], b = wave, wave + NW, ... (as bn_act_train_fwd_kernel)
__device__ __forceinline__ void channel_stats(const float* cache, int B, float s, float invN, float eps, int seg, int slot, float (*red)[64],
                                              int wave, int lane, float& mean, float& var, float& invstd) {
    mean = channel_sum(s, seg, slot, red, wave, lane) * invN;
    float q = 0.f;
    for (int b = wave; b < B; b += NW) {
        const float v = cache[b * 64 + lane];
        q = fmaf(v - mean, v - mean, q);
    }
    var = channel_sum(q, seg, slot, red, wave, lane) * invN;
    invstd = 1.f / sqrtf(var + eps);
}

template <int K>
__global__ __launch_bounds__(64 * NW) void mbconv_mid_fwd_kernel(const MidFwdP p) {
    extern __shared__ __attribute__((aligned(16))) float cache[];      // [B][64] e_pre -> SiLU(bn0) | [B][64] depthwise output
    __shared__ float red[NW][64];
    constexpr int R = K / 2, KK = K * K;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int HW = p.HW, W = 1 << p.logw;
    const int seg = HW >= 64 ? 64 : HW;
    const int slot = lane / HW, pos = lane & (HW - 1), py = pos >> p.logw, px = pos & (W - 1);
    const int c0 = blockIdx.x * (64 / HW), c = c0 + slot;
    const bool cok = c < p.C;
    const float invN = 1.f / ((float)p.B * (float)HW), n = (float)p.B * (float)HW;
    const long off = (long)c0 * HW + lane, bstride = (long)p.C * HW;
    float* const dch = cache + p.B * 64;
    const float g0 = cok ? p.g0[c] : 0.f, b0 = cok ? p.b0[c] : 0.f, g1 = cok ? p.g1[c] : 0.f, b1 = cok ? p.b1[c] : 0.f;
    const float rm0 = cok ? p.rm0[c] : 0.f, rv0 = cok ? p.rv0[c] : 0.f, rm1 = cok ? p.rm1[c] : 0.f, rv1 = cok ? p.rv1[c] : 0.f;
    float wk[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) wk[k] = cok ? p.wdw[(long)c * KK + k] : 0.f;
    // ---- BatchNorm0: statistics of the expand conv's output
    float s = 0.f;
    for (int b0i = wave; b0i < p.B; b0i += NW * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0i + NW * u;
            v[u] = (cok && b < p.B) ? p.e_pre[b * bstride + off] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0i + NW * u;
            if (b < p.B) cache[b * 64 + lane] = v[u];
            s += v[u];
        }
    }
    float mean, var, invstd;
    channel_stats(cache, p.B, s, invN, p.eps0, seg, slot, red, wave, lane, mean, var, invstd);
    if (cok && wave == 0 && pos == 0) {
        p.mean0[c] = mean;
        p.invstd0[c] = invstd;
        p.rm0[c] = (1.f - p.mom0) * rm0 + p.mom0 * mean;
        p.rv0[c] = (1.f - p.mom0) * rv0 + p.mom0 * (n > 1.f ? var * n / (n - 1.f) : var);
    }
    {
        const float sc = g0 * invstd, sh = b0 - mean * sc;
        for (int b = wave; b < p.B; b += NW) cache[b * 64 + lane] = cok ? act_f<1>(fmaf(cache[b * 64 + lane], sc, sh)) : 0.f;
    }
    __syncthreads();
    // ---- depthwise K x K, stride 1, zero padding R: neighbours of this lane's pixel inside its own plane
    s = 0.f;
    for (int b = wave; b < p.B; b += NW) {
        const float* pl = cache + b * 64 + slot * HW;
        float acc = 0.f;
#pragma unroll
        for (int dy = 0; dy < K; ++dy) {
            if (dy - R >= W || R - dy >= W) continue;       // (uniform: a tap further out than the plane is wide never lands inside it -- 16 of the 25 taps at 2x2)
            const int yy = py + dy - R;
#pragma unroll
            for (int dx = 0; dx < K; ++dx) {
                if (dx - R >= W || R - dx >= W) continue;
                const int xx = px + dx - R;
                if ((unsigned)yy < (unsigned)W && (unsigned)xx < (unsigned)W) acc = fmaf(wk[dy * K + dx], pl[(yy << p.logw) + xx], acc);
            }
        }
        dch[b * 64 + lane] = acc;
        if (cok) p.d_pre[b * bstride + off] = acc;
        s += acc;
    }
    // ---- BatchNorm1 + SiLU + plane means
    channel_stats(dch, p.B, s, invN, p.eps1, seg, slot, red, wave, lane, mean, var, invstd);
    if (cok && wave == 0 && pos == 0) {
        p.mean1[c] = mean;
        p.invstd1[c] = invstd;
        p.rm1[c] = (1.f - p.mom1) * rm1 + p.mom1 * mean;
        p.rv1[c] = (1.f - p.mom1) * rv1 + p.mom1 * (n > 1.f ? var * n / (n - 1.f) : var);
    }
    const float sc1 = g1 * invstd, sh1 = b1 - mean * sc1, invHW = 1.f / (float)HW;
    for (int b = wave; b < p.B; b += NW) {
        const float v = act_f<1>(fmaf(dch[b * 64 + lane], sc1, sh1));
        if (cok) p.y[b * bstride + off] = v;
        float w = v;
        for (int o = 1; o < seg; o <<= 1) w += __shfl_xor(w, o, 64);
        if (cok && pos == 0) p.pooled[(long)b * p.C + c] = w * invHW;
    }
}

struct MidBwdP {
    const float* dout; const float* gate; const float* dpooled;      // gradient of y * gate; excite gate [B][C]; gradient of the plane means [B][C]
    const float* d_pre; const float* e_pre; const float* wdw;
    const float* g0; const float* b0; const float* mean0; const float* invstd0;
    const float* g1; const float* b1; const float* mean1; const float* invstd1;
    float* de_pre; float* dwdw; float* dg0; float* db0; float* dg1; float* db1;
    int B, C, HW, logw;
};

template <int K>
__global__ __launch_bounds__(64 * NW) void mbconv_mid_bwd_kernel(const MidBwdP p) {
    extern __shared__ __attribute__((aligned(16))) float cache[];      // A: dz1 -> dd_pre | Bc: xhat1 -> SiLU(bn0) -> dz0 | Cc: xhat0
    __shared__ float red[NW][64];
    __shared__ float redw[NW][K * K][16];
    constexpr int R = K / 2, KK = K * K;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int HW = p.HW, W = 1 << p.logw;
    const int seg = HW >= 64 ? 64 : HW;
    const int slot = lane / HW, pos = lane & (HW - 1), py = pos >> p.logw, px = pos & (W - 1);
    const int cpw = 64 / HW, c0 = blockIdx.x * cpw, c = c0 + slot;
    const bool cok = c < p.C;
    const float invN = 1.f / ((float)p.B * (float)HW), invHW = 1.f / (float)HW;
    const long off = (long)c0 * HW + lane, bstride = (long)p.C * HW;
    float* const A = cache;
    float* const Bc = cache + p.B * 64;
    float* const Cc = cache + 2 * p.B * 64;
    const float g0 = cok ? p.g0[c] : 0.f, b0 = cok ? p.b0[c] : 0.f, g1 = cok ? p.g1[c] : 0.f, b1 = cok ? p.b1[c] : 0.f;
    const float m0 = cok ? p.mean0[c] : 0.f, i0 = cok ? p.invstd0[c] : 0.f, m1 = cok ? p.mean1[c] : 0.f, i1 = cok ? p.invstd1[c] : 0.f;
    float wk[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) wk[k] = cok ? p.wdw[(long)c * KK + k] : 0.f;
    // ---- BatchNorm1 backward through SiLU and the excite gate (bn_act_train_bwd_kernel<1> with gate / dpooled)
    float s1 = 0.f, s2 = 0.f;
    for (int b0i = wave; b0i < p.B; b0i += NW * U) {
        float xv[U], dv[U], ev[U], gt[U], dpo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0i + NW * u;
            const bool ok = cok && b < p.B;
            xv[u] = ok ? p.d_pre[b * bstride + off] : 0.f;
            dv[u] = ok ? p.dout[b * bstride + off] : 0.f;
            ev[u] = ok ? p.e_pre[b * bstride + off] : 0.f;
            gt[u] = ok ? p.gate[(long)b * p.C + c] : 1.f;
            dpo[u] = ok ? p.dpooled[(long)b * p.C + c] * invHW : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int b = b0i + NW * u;
            if (b >= p.B) continue;
            const float xh = (xv[u] - m1) * i1;
            const float dz = cok ? fmaf(dv[u], gt[u], dpo[u]) * act_grad<1>(fmaf(xh, g1, b1)) : 0.f;
            s2 = fmaf(dz, xh, s2);
            s1 += dz;
            A[b * 64 + lane] = dz;
            Bc[b * 64 + lane] = xh;
            Cc[b * 64 + lane] = (ev[u] - m0) * i0;                 // xhat0 (the depthwise conv's input is SiLU(xhat0 g0 + b0))
        }
    }
    float sum_dz = channel_sum(s1, seg, slot, red, wave, lane);
    float sum_dzx = channel_sum(s2, seg, slot, red, wave, lane);
    if (cok && wave == 0 && pos == 0) {
        p.db1[c] = sum_dz;
        p.dg1[c] = sum_dzx;
    }
    {
        const float k1 = sum_dz * invN, k2 = sum_dzx * invN, gi = g1 * i1;
        for (int b = wave; b < p.B; b += NW) {
            const float dd = cok ? gi * (A[b * 64 + lane] - k1 - Bc[b * 64 + lane] * k2) : 0.f;     // gradient of the depthwise output
            A[b * 64 + lane] = dd;
            Bc[b * 64 + lane] = cok ? act_f<1>(fmaf(Cc[b * 64 + lane], g0, b0)) : 0.f;              // the depthwise input, recomputed
        }
    }
    __syncthreads();
    // ---- depthwise weight gradient: dw[c][tap] = sum over (b, p) of dd[b][c][p] * in[b][c][p + tap]
    {
        float pw[KK];
#pragma unroll
        for (int k = 0; k < KK; ++k) pw[k] = 0.f;
        for (int b = wave; b < p.B; b += NW) {
            const float dd = A[b * 64 + lane];
            const float* pl = Bc + b * 64 + slot * HW;
#pragma unroll
            for (int dy = 0; dy < K; ++dy) {
                if (dy - R >= W || R - dy >= W) continue;
                const int yy = py + dy - R;
#pragma unroll
                for (int dx = 0; dx < K; ++dx) {
                    if (dx - R >= W || R - dx >= W) continue;
                    const int xx = px + dx - R;
                    if ((unsigned)yy < (unsigned)W && (unsigned)xx < (unsigned)W) pw[dy * K + dx] = fmaf(dd, pl[(yy << p.logw) + xx], pw[dy * K + dx]);
                }
            }
        }
#pragma unroll
        for (int k = 0; k < KK; ++k) {
            float v = pw[k];
            const int ty = k / K - R, tx = k % K - R;
            if (ty < W && -ty < W && tx < W && -tx < W)          // (uniform; a tap that never lands inside the plane stays 0)
                for (int o = 1; o < seg; o <<= 1) v += __shfl_xor(v, o, 64);
            if (pos == 0) redw[wave][k][slot] = v;
        }
        __syncthreads();
        for (int u = t; u < KK * cpw; u += 64 * NW) {
            const int k = u / cpw, sl = u - k * cpw;
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < NW; w += 2) {
                a += redw[w][k][sl];
                b += redw[w + 1][k][sl];
            }
            if (c0 + sl < p.C) p.dwdw[(long)(c0 + sl) * KK + k] = a + b;
        }
    }
    // ---- depthwise data gradient, then BatchNorm0 backward through SiLU
    s1 = 0.f;
    s2 = 0.f;
    __syncthreads();              // (Bc is overwritten below: every wave is past the weight gradient's reads)
    for (int b = wave; b < p.B; b += NW) {
        const float* pl = A + b * 64 + slot * HW;
        float acc = 0.f;
#pragma unroll
        for (int dy = 0; dy < K; ++dy) {
            if (dy - R >= W || R - dy >= W) continue;
            const int yy = py - dy + R;
#pragma unroll
            for (int dx = 0; dx < K; ++dx) {
                if (dx - R >= W || R - dx >= W) continue;
                const int xx = px - dx + R;
                if ((unsigned)yy < (unsigned)W && (unsigned)xx < (unsigned)W) acc = fmaf(wk[dy * K + dx], pl[(yy << p.logw) + xx], acc);
            }
        }
        const float xh = Cc[b * 64 + lane];
        const float dz = cok ? acc * act_grad<1>(fmaf(xh, g0, b0)) : 0.f;
        s2 = fmaf(dz, xh, s2);
        s1 += dz;
        Bc[b * 64 + lane] = dz;          // (own element)
    }
    sum_dz = channel_sum(s1, seg, slot, red, wave, lane);
    sum_dzx = channel_sum(s2, seg, slot, red, wave, lane);
    if (cok && wave == 0 && pos == 0) {
        p.db0[c] = sum_dz;
        p.dg0[c] = sum_dzx;
    }
    if (!p.de_pre) return;
    {
        const float k1 = sum_dz * invN, k2 = sum_dzx * invN, gi = g0 * i0;
        for (int b = wave; b < p.B; b += NW)
            if (cok) p.de_pre[b * bstride + off] = gi * (Bc[b * 64 + lane] - k1 - Cc[b * 64 + lane] * k2);
    }
}

// squeeze-excite backward, step 1: draw[b][c] = sum over the plane of dout * y, y = act(bn(x)) recomputed from the saved conv output
// (the forward scaled y in place, it is not kept).  Planes are contiguous, so a wave takes 64 consecutive elements = 64 / hw whole
// planes when hw < 64 (a wave per 4-element plane would be 172 k nearly empty waves at 2x2), else one plane per wave.
template <int ACT>
__global__ __launch_bounds__(256) void se_bwd_dgate_kernel(const float* __restrict__ dout, const float* __restrict__ x,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          float* __restrict__ draw, long planes, int C, int hw) {
    const int lane = threadIdx.x & 63;
    const long wv = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (hw < 64) {
        const long idx = wv * 64 + lane;
        const long pl = idx / hw;
        float acc = 0.f;
        if (pl < planes) {
            const int c = (int)(pl % C);
            const float sc = gamma[c] * invstd[c], sh = beta[c] - mean[c] * sc;
            acc = dout[idx] * act_f<ACT>(fmaf(x[idx], sc, sh));
        }
        for (int o = 1; o < hw; o <<= 1) acc += __shfl_xor(acc, o, 64);
        if (pl < planes && (lane & (hw - 1)) == 0) draw[pl] = acc;
        return;
    }
    const long pl = wv;
    if (pl >= planes) return;
    const int c = (int)(pl % C);
    const float sc = gamma[c] * invstd[c], sh = beta[c] - mean[c] * sc;
    const float* xp = x + pl * hw;
    const float* dp = dout + pl * hw;
    float acc = 0.f;
    for (int i = lane; i < hw; i += 64) acc = fmaf(dp[i], act_f<ACT>(fmaf(xp[i], sc, sh)), acc);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) draw[pl] = acc;
}

// step 2a, grid (B, NCH): workgroup (b, k) owns the channels [k CC, (k+1) CC): through the sigmoid, and its share of the expand conv
//   dsig[c] = draw[c] gate[c] (1 - gate[c]);   part[b][k][s] = sum over its c of dsig[c] w2[c][s]
// (one workgroup per image ran a 672-iteration dependent loop per wave at C = 2688; chunks keep every loop under ~30 iterations)
constexpr int SE_MAX_SQ = 256;
__global__ __launch_bounds__(256) void se_bwd_expand_kernel(const float* __restrict__ draw, const float* __restrict__ gate,
                                                           const float* __restrict__ w2, float* __restrict__ dsig_out,
                                                           float* __restrict__ part, int C, int SQ, int CC) {
    __shared__ float dsig[256 * 4];
    __shared__ float red[4][SE_MAX_SQ];
    const int b = blockIdx.x, k = blockIdx.y, nch = gridDim.y, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int cb = k * CC, ce = (cb + CC < C) ? cb + CC : C, n = ce - cb;
    for (int i = t; i < n; i += 256) {
        const float gt = gate[(long)b * C + cb + i];
        const float d = draw[(long)b * C + cb + i] * gt * (1.f - gt);
        dsig[i] = d;
        dsig_out[(long)b * C + cb + i] = d;
    }
    __syncthreads();
    for (int s0 = 0; s0 < SQ; s0 += 64) {
        const int s = s0 + lane;
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        if (s < SQ) {
            int i = wave;
            for (; i + 12 < n; i += 16) {
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = fmaf(dsig[i + 4 * u], w2[(long)(cb + i + 4 * u) * SQ + s], a[u]);
            }
            for (; i < n; i += 4) a[0] = fmaf(dsig[i], w2[(long)(cb + i) * SQ + s], a[0]);
            red[wave][s] = (a[0] + a[1]) + (a[2] + a[3]);
        }
    }
    __syncthreads();
    for (int s = t; s < SQ; s += 256) part[((long)b * nch + k) * SQ + s] = (red[0][s] + red[1][s]) + (red[2][s] + red[3][s]);
}

// step 2b, grid (B, NCH): dh[s] = sum_k part[b][k][s];  dhp[s] = dh[s] silu'(hp[s]);  dpooled[c] = sum_s dhp[s] w1[s][c] for its channels
__global__ __launch_bounds__(256) void se_bwd_reduce_kernel(const float* __restrict__ part, const float* __restrict__ hidden_pre,
                                                           const float* __restrict__ w1, float* __restrict__ dhp_out,
                                                           float* __restrict__ dpooled, int C, int SQ, int CC) {
    __shared__ float dhp[SE_MAX_SQ];
    const int b = blockIdx.x, k = blockIdx.y, nch = gridDim.y, t = threadIdx.x;
    for (int s = t; s < SQ; s += 256) {
        float dh = 0.f;
        for (int j = 0; j < nch; ++j) dh += part[((long)b * nch + j) * SQ + s];
        const float d = dh * act_grad<1>(hidden_pre[(long)b * SQ + s]);
        dhp[s] = d;
        if (k == 0) dhp_out[(long)b * SQ + s] = d;
    }
    __syncthreads();
    const int cb = k * CC, ce = (cb + CC < C) ? cb + CC : C;
    for (int c = cb + t; c < ce; c += 256) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        int s = 0;
        for (; s + 3 < SQ; s += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = fmaf(dhp[s + u], w1[(long)(s + u) * C + c], a[u]);
        }
        for (; s < SQ; ++s) a[0] = fmaf(dhp[s], w1[(long)s * C + c], a[0]);
        dpooled[(long)b * C + c] = (a[0] + a[1]) + (a[2] + a[3]);
    }
}

// step 3: the four parameter gradients, sums over the batch:
//   dw2[c][s] = sum_b dsig[b][c] h[b][s];  db2[c] = sum_b dsig[b][c];  dw1[s][c] = sum_b dhp[b][s] pooled[b][c];  db1[s] = sum_b dhp[b][s]
// Two small "A^T B" products with K = B.  One thread per output element looping over the batch re-read both operands from L2 for
// every element (~300 MB of L1 traffic at C = 2688, 20 us per block); here a wave owns a 32 x 32 tile and feeds the fp32 matrix unit
// straight from global memory -- v_mfma_f32_32x32x2_f32 takes A[i][k] / B[k][j] with i, j = lane % 32 and k = lane / 32, which is
// exactly a coalesced row read of both operands -- in true fp32, fixed order.  The bias gradients are the column sums of A.
typedef float floatx16 __attribute__((ext_vector_type(16)));
struct GemmTN {
    const float* A;        // [K][M]
    const float* Bm;       // [K][N]
    float* out;            // [M][N] = A^T Bm
    float* colsum;         // [M] = sum_k A[k][:]
    int M, N;
};
__global__ __launch_bounds__(256) void se_bwd_params_kernel(const GemmTN g0, const GemmTN g1, int K, int tiles0, int tn0, int tn1) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int tile = blockIdx.x * 4 + wave;
    const bool second = tile >= tiles0;
    if (second) tile -= tiles0;
    const float* __restrict__ A = second ? g1.A : g0.A;
    const float* __restrict__ Bm = second ? g1.Bm : g0.Bm;
    float* __restrict__ out = second ? g1.out : g0.out;
    float* __restrict__ colsum = second ? g1.colsum : g0.colsum;
    const int M = second ? g1.M : g0.M, N = second ? g1.N : g0.N, tn = second ? tn1 : tn0;
    const int ti = tile / tn, tj = tile - ti * tn;
    if (ti * 32 >= M) return;
    const int i = ti * 32 + (lane & 31), j = tj * 32 + (lane & 31), kh = lane >> 5;
    const bool iok = i < M, jok = j < N;
    floatx16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float cs = 0.f;
    for (int k0 = 0; k0 < K; k0 += 64) {
        float a[32], b[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int k = k0 + 2 * u + kh;
            a[u] = (iok && k < K) ? A[(long)k * M + i] : 0.f;
            b[u] = (jok && k < K) ? Bm[(long)k * N + j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[u], acc, 0, 0, 0);
            cs += a[u];
        }
    }
    if (tj == 0) {
        cs += __shfl_xor(cs, 32, 64);
        if (kh == 0 && iok) colsum[i] = cs;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ii = ti * 32 + 8 * (r >> 2) + 4 * kh + (r & 3);
        if (ii < M && jok) out[(long)ii * N + j] = acc[r];
    }
}

// training forms of the two inference squeeze-excite kernels (srbh_dwconv.hip): they also keep what the backward needs
__global__ __launch_bounds__(256) void se_hidden_train_kernel(const float* __restrict__ pooled, const float* __restrict__ w1,
                                                             const float* __restrict__ b1, float* __restrict__ hidden,
                                                             float* __restrict__ hidden_pre, int C, int SQ) {
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
        hidden_pre[(long)b * SQ + j] = u;
        hidden[(long)b * SQ + j] = u * sigmoidf(u);
    }
}

// gate + scale, planes of 4 / 16 / 64 / 256 elements: a workgroup takes 1024 consecutive elements (a float4 per thread); the G = hw / 4
// threads that hold one plane compute its gate together (a wave per 4-element plane would be 172 k nearly empty waves at 2x2)
__global__ __launch_bounds__(256) void se_gate_scale_train_kernel(float* __restrict__ y, const float* __restrict__ hidden,
                                                                 const float* __restrict__ w2, const float* __restrict__ b2,
                                                                 float* __restrict__ gate, long planes, int C, int SQ, int hw) {
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
    const float g = sigmoidf(acc + b2[c]);
    if (sub == 0) gate[pl] = g;
    floatx4 v = *(floatx4*)(y + e);
    v *= g;
    *(floatx4*)(y + e) = v;
}

// (any other plane size: one wave per plane)
__global__ __launch_bounds__(256) void se_gate_scale_train_wave_kernel(float* __restrict__ y, const float* __restrict__ hidden,
                                                                      const float* __restrict__ w2, const float* __restrict__ b2,
                                                                      float* __restrict__ gate, long planes, int C, int SQ, int hw) {
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
    const float g = sigmoidf(acc + b2[c]);
    if (lane == 0) gate[pl] = g;
    float* yp = y + pl * hw;
    for (int i = lane; i < hw; i += 64) yp[i] *= g;
}

// ---- U-Net decoder block entry (smp DecoderBlock.forward: x = F.interpolate(x, scale_factor=2, mode="nearest"); x = torch.cat([x, skip], 1)),
// one launch forward and one backward instead of upsample + cat and their autograd (ATen's nearest backward alone ran 56 us per call).
// A thread owns 4 consecutive output columns: two source pixels of x, or a float4 of skip.
__global__ __launch_bounds__(256) void up2_cat_fwd_kernel(const float* __restrict__ x, const float* __restrict__ skip,
                                                         float* __restrict__ out, long total4, int Cx, int Cs, int H, int W) {
    const int W4 = (2 * W) >> 2, OH = 2 * H, Ct = Cx + Cs;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total4; idx += (long)gridDim.x * 256) {
        const int xq = (int)(idx % W4);
        long r = idx / W4;
        const int y = (int)(r % OH);
        r /= OH;
        const int c = (int)(r % Ct);
        const long b = r / Ct;
        floatx4 v;
        if (c < Cx) {
            const float* s = x + ((b * Cx + c) * H + (y >> 1)) * W + 2 * xq;
            const float s0 = s[0], s1 = s[1];
            v = floatx4{s0, s0, s1, s1};
        } else {
            v = *(const floatx4*)(skip + ((b * Cs + (c - Cx)) * OH + y) * (long)(2 * W) + 4 * xq);
        }
        *(floatx4*)(out + 4 * idx) = v;
    }
}

// backward: dx[b][c][y][x] = the sum of its 2x2 output pixels (a thread makes two dx columns from two float4 of dout),
// dskip = the skip slice of dout as a contiguous tensor
__global__ __launch_bounds__(256) void up2_cat_bwd_kernel(const float* __restrict__ dout, float* __restrict__ dx, float* __restrict__ dskip,
                                                         long n_x2, long n_s4, int Cx, int Cs, int H, int W) {
    const int W2 = W >> 1, OH = 2 * H, OW = 2 * W, Ct = Cx + Cs, W4 = OW >> 2;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < n_x2 + n_s4; idx += (long)gridDim.x * 256) {
        if (idx < n_x2) {
            const int xp = (int)(idx % W2);
            long r = idx / W2;
            const int y = (int)(r % H);
            r /= H;
            const int c = (int)(r % Cx);
            const long b = r / Cx;
            const float* d = dout + ((b * Ct + c) * OH + 2 * y) * (long)OW + 4 * xp;
            const floatx4 a = *(const floatx4*)d, e = *(const floatx4*)(d + OW);
            float* o = dx + ((b * Cx + c) * H + y) * (long)W + 2 * xp;
            o[0] = (a[0] + a[1]) + (e[0] + e[1]);
            o[1] = (a[2] + a[3]) + (e[2] + e[3]);
        } else {
            const long i = idx - n_x2;
            const int xq = (int)(i % W4);
            long r = i / W4;
            const int y = (int)(r % OH);
            r /= OH;
            const int c = (int)(r % Cs);
            const long b = r / Cs;
            *(floatx4*)(dskip + 4 * i) = *(const floatx4*)(dout + ((b * Ct + Cx + c) * OH + y) * (long)OW + 4 * xq);
        }
    }
}

// ---- the same two operations for LARGE planes (32x32 and up: the stem, the first three encoder blocks, the last two decoder levels):
// a channel no longer fits one workgroup's LDS, so each direction is two launches -- per-(channel, image subset) partial sums in
// double (fixed slots, no atomics: deterministic and nothing to zero), then one workgroup per (b, c) plane that folds the S partials
// and applies.  Traffic: forward x twice + y once; backward x and dy twice + dx once (dz is recomputed instead of stored).
__device__ __forceinline__ double wg_sum_double(double v, double* red, int t) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((t & 63) == 0) red[t >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void bn_large_stats_kernel(const float* __restrict__ x, double* __restrict__ part, int B, int C, int HW, int S) {
    __shared__ double red[4];
    const int c = blockIdx.x, sp = blockIdx.y, t = threadIdx.x, n4 = HW >> 2;
    // sums of (x - pivot), pivot = the channel's first element (the same value in every workgroup of the channel): E[x^2] - mean^2 on
    // raw fp32 sums loses (mean/std)^2 * 1e-7 of the variance for channels whose mean is far from 0 (round-3 ADVICE); shifted sums
    // do not, at one subtraction per element
    const float pivot = x[(long)c * HW];
    float s = 0.f, q = 0.f;
#pragma unroll 4
    for (int b = sp; b < B; b += S) {
        const floatx4* pl = (const floatx4*)(x + ((long)b * C + c) * HW);
        for (int i = t; i < n4; i += 256) {
            const floatx4 v = pl[i] - pivot;
            s += (v[0] + v[1]) + (v[2] + v[3]);
            q = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], q))));
        }
    }
    const double ds = wg_sum_double((double)s, red, t), dq = wg_sum_double((double)q, red, t);
    if (t == 0) {
        part[((long)c * S + sp) * 2] = ds;
        part[((long)c * S + sp) * 2 + 1] = dq;
    }
}

template <int ACT>
__global__ __launch_bounds__(256) void bn_large_apply_kernel(const FwdP p, const double* __restrict__ part, int S) {
    __shared__ double red[4];
    const long pl = blockIdx.x;
    const int c = (int)(pl % p.C), t = threadIdx.x, n4 = p.HW >> 2;
    const long b = pl / p.C;
    double ds = 0.0, dq = 0.0;
    for (int k = 0; k < S; ++k) {
        ds += part[((long)c * S + k) * 2];
        dq += part[((long)c * S + k) * 2 + 1];
    }
    const double n = (double)p.B * (double)p.HW, dshift = ds / n, dmean = (double)p.x[(long)c * p.HW] + dshift;      // (pivot: bn_large_stats_kernel)
    double dvar = dq / n - dshift * dshift;
    dvar = dvar > 0.0 ? dvar : 0.0;
    const float mean = (float)dmean, var = (float)dvar, invstd = 1.f / sqrtf(var + p.eps);
    if (b == 0 && t == 0) {
        p.save_mean[c] = mean;
        p.save_invstd[c] = invstd;
        if (p.running_mean) {
            p.running_mean[c] = (1.f - p.momentum) * p.running_mean[c] + p.momentum * mean;
            p.running_var[c] = (1.f - p.momentum) * p.running_var[c] + p.momentum * (float)(n > 1.0 ? dvar * n / (n - 1.0) : dvar);
        }
    }
    const float scale = p.gamma[c] * invstd, shift = p.beta[c] - mean * scale, dr = p.drop ? p.drop[b] : 1.f;
    const floatx4* xp = (const floatx4*)(p.x + pl * p.HW);
    const floatx4* rp = p.res ? (const floatx4*)(p.res + pl * p.HW) : nullptr;
    floatx4* yp = (floatx4*)(p.y + pl * p.HW);
    float w = 0.f;
    for (int i = t; i < n4; i += 256) {
        floatx4 v = xp[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = act_f<ACT>(fmaf(v[k], scale, shift)) * dr;
        if (rp) v += rp[i];
        yp[i] = v;
        w += (v[0] + v[1]) + (v[2] + v[3]);
    }
    if (p.pooled) {
        const double tot = wg_sum_double((double)w, red, t);
        if (t == 0) p.pooled[pl] = (float)(tot / (double)p.HW);
    }
}

template <int ACT>
__device__ __forceinline__ floatx4 large_dz(const floatx4 xv, const floatx4 dv, float mean, float invstd, float g, float be, float gt, float dpo,
                                            floatx4& xh) {
    floatx4 dz;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        xh[k] = (xv[k] - mean) * invstd;
        dz[k] = fmaf(dv[k], gt, dpo) * act_grad<ACT>(fmaf(xh[k], g, be));
    }
    return dz;
}

template <int ACT>
__global__ __launch_bounds__(256) void bn_large_bwd_stats_kernel(const BwdP p, double* __restrict__ part, int S) {
    __shared__ double red[4];
    const int c = blockIdx.x, sp = blockIdx.y, t = threadIdx.x, n4 = p.HW >> 2;
    const float mean = p.save_mean[c], invstd = p.save_invstd[c], g = p.gamma[c], be = p.beta[c], invHW = 1.f / (float)p.HW;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll 2
    for (int b = sp; b < p.B; b += S) {
        const long pl = (long)b * p.C + c;
        const float dr = p.drop ? p.drop[b] : 1.f;
        const float gt = (p.gate ? p.gate[pl] : 1.f) * dr, dpo = (p.gate ? p.dpooled[pl] * invHW : 0.f) * dr;
        const floatx4* xp = (const floatx4*)(p.x + pl * p.HW);
        const floatx4* dp = (const floatx4*)(p.dy + pl * p.HW);
        for (int i = t; i < n4; i += 256) {
            floatx4 xh;
            const floatx4 dz = large_dz<ACT>(xp[i], dp[i], mean, invstd, g, be, gt, dpo, xh);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                s1 += dz[k];
                s2 = fmaf(dz[k], xh[k], s2);
            }
        }
    }
    const double d1 = wg_sum_double((double)s1, red, t), d2 = wg_sum_double((double)s2, red, t);
    if (t == 0) {
        part[((long)c * S + sp) * 2] = d1;
        part[((long)c * S + sp) * 2 + 1] = d2;
    }
}

template <int ACT>
__global__ __launch_bounds__(256) void bn_large_bwd_apply_kernel(const BwdP p, const double* __restrict__ part, int S) {
    const long pl = blockIdx.x;
    const int c = (int)(pl % p.C), t = threadIdx.x, n4 = p.HW >> 2;
    const long b = pl / p.C;
    double d1 = 0.0, d2 = 0.0;
    for (int k = 0; k < S; ++k) {
        d1 += part[((long)c * S + k) * 2];
        d2 += part[((long)c * S + k) * 2 + 1];
    }
    if (b == 0 && t == 0) {
        p.dbeta[c] = (float)d1;
        p.dgamma[c] = (float)d2;
    }
    if (!p.dx) return;
    const double n = (double)p.B * (double)p.HW;
    const float k1 = (float)(d1 / n), k2 = (float)(d2 / n);
    const float mean = p.save_mean[c], invstd = p.save_invstd[c], g = p.gamma[c], be = p.beta[c], gi = g * invstd, invHW = 1.f / (float)p.HW;
    const float dr = p.drop ? p.drop[b] : 1.f;
    const float gt = (p.gate ? p.gate[pl] : 1.f) * dr, dpo = (p.gate ? p.dpooled[pl] * invHW : 0.f) * dr;
    const floatx4* xp = (const floatx4*)(p.x + pl * p.HW);
    const floatx4* dp = (const floatx4*)(p.dy + pl * p.HW);
    floatx4* op = (floatx4*)(p.dx + pl * p.HW);
    for (int i = t; i < n4; i += 256) {
        floatx4 xh;
        floatx4 dz = large_dz<ACT>(xp[i], dp[i], mean, invstd, g, be, gt, dpo, xh);
#pragma unroll
        for (int k = 0; k < 4; ++k) dz[k] = gi * (dz[k] - k1 - xh[k] * k2);
        op[i] = dz;
    }
}

int large_splits(int B, int C) {
    int S = 1024 / (C > 0 ? C : 1);
    S = S < 1 ? 1 : S;
    S = S > 32 ? 32 : S;
    return S > B ? B : S;
}
bool large_ok(int B, int C, int HW) { return B > 0 && C > 0 && HW >= 256 && (HW & 3) == 0; }

int se_chunks(int C) {                      // channel chunks of the squeeze-excite backward: <= 1024 channels each (LDS), ~16 at most
    int n = (C + 127) / 128;
    return n < 1 ? 1 : (n > 16 ? ((C + 1023) / 1024 > 16 ? (C + 1023) / 1024 : 16) : n);
}
int row_width(int HW) { return HW == 256 ? 256 : ((HW == 64 || HW == 16 || HW == 4 || HW == 1) ? 64 : 0); }
size_t fwd_lds(int B, int RW) { return (size_t)B * RW * 4; }
}  // namespace

static int mid_logw(int H, int W) { return (H == W && (W == 2 || W == 4 || W == 8)) ? (W == 2 ? 1 : W == 4 ? 2 : 3) : -1; }

/* 1 when the fused MBConv middle kernels take the shape: square planes of 2x2, 4x4 or 8x8, depthwise kernel 3 or 5 with stride 1, and
 * three copies of the workgroup's B x 64-float run within LDS */
extern "C" int srbh_mbconv_mid_supported(int B, int C, int H, int W, int K, int stride) {
    return B > 0 && C > 0 && mid_logw(H, W) > 0 && (K == 3 || K == 5) && stride == 1 && (size_t)3 * B * 64 * 4 <= (size_t)MAX_LDS_B - 32 * 1024;
}

extern "C" int srbh_mbconv_mid_fwd(const srbh_mbmid_args* a, void* stream) {
    SRBH_REQUIRE(a && a->e_pre && a->wdw && a->gamma0 && a->beta0 && a->gamma1 && a->beta1 && a->running_mean0 && a->running_var0 &&
                 a->running_mean1 && a->running_var1 && a->mean0 && a->invstd0 && a->mean1 && a->invstd1 && a->d_pre && a->y && a->pooled,
                 "srbh_mbconv_mid_fwd: null pointer");
    SRBH_REQUIRE(srbh_mbconv_mid_supported(a->B, a->C, a->H, a->W, a->K, 1), "srbh_mbconv_mid_fwd: unsupported shape B=%d C=%d %dx%d k%d", a->B, a->C, a->H, a->W, a->K);
    MidFwdP p;
    p.e_pre = a->e_pre; p.wdw = a->wdw;
    p.g0 = a->gamma0; p.b0 = a->beta0; p.rm0 = a->running_mean0; p.rv0 = a->running_var0; p.mean0 = a->mean0; p.invstd0 = a->invstd0;
    p.g1 = a->gamma1; p.b1 = a->beta1; p.rm1 = a->running_mean1; p.rv1 = a->running_var1; p.mean1 = a->mean1; p.invstd1 = a->invstd1;
    p.d_pre = a->d_pre; p.y = a->y; p.pooled = a->pooled;
    p.mom0 = a->momentum0; p.eps0 = a->eps0; p.mom1 = a->momentum1; p.eps1 = a->eps1;
    p.B = a->B; p.C = a->C; p.HW = a->H * a->W; p.logw = mid_logw(a->H, a->W);
    const int cpw = 64 / p.HW;
    const size_t lds = (size_t)2 * p.B * 64 * 4;
    const dim3 grid((p.C + cpw - 1) / cpw);
    if (a->K == 3) {
        SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)mbconv_mid_fwd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_B - 32 * 1024)));
        hipLaunchKernelGGL(mbconv_mid_fwd_kernel<3>, grid, dim3(64 * NW), lds, (hipStream_t)stream, p);
    } else {
        SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)mbconv_mid_fwd_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_B - 32 * 1024)));
        hipLaunchKernelGGL(mbconv_mid_fwd_kernel<5>, grid, dim3(64 * NW), lds, (hipStream_t)stream, p);
    }
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_mbconv_mid_bwd(const srbh_mbmid_bwd_args* a, void* stream) {
    SRBH_REQUIRE(a && a->dout && a->gate && a->dpooled && a->d_pre && a->e_pre && a->wdw && a->gamma0 && a->beta0 && a->mean0 && a->invstd0 &&
                 a->gamma1 && a->beta1 && a->mean1 && a->invstd1 && a->dwdw && a->dgamma0 && a->dbeta0 && a->dgamma1 && a->dbeta1,
                 "srbh_mbconv_mid_bwd: null pointer");
    SRBH_REQUIRE(srbh_mbconv_mid_supported(a->B, a->C, a->H, a->W, a->K, 1), "srbh_mbconv_mid_bwd: unsupported shape B=%d C=%d %dx%d k%d", a->B, a->C, a->H, a->W, a->K);
    MidBwdP p;
    p.dout = a->dout; p.gate = a->gate; p.dpooled = a->dpooled; p.d_pre = a->d_pre; p.e_pre = a->e_pre; p.wdw = a->wdw;
    p.g0 = a->gamma0; p.b0 = a->beta0; p.mean0 = a->mean0; p.invstd0 = a->invstd0;
    p.g1 = a->gamma1; p.b1 = a->beta1; p.mean1 = a->mean1; p.invstd1 = a->invstd1;
    p.de_pre = a->de_pre; p.dwdw = a->dwdw; p.dg0 = a->dgamma0; p.db0 = a->dbeta0; p.dg1 = a->dgamma1; p.db1 = a->dbeta1;
    p.B = a->B; p.C = a->C; p.HW = a->H * a->W; p.logw = mid_logw(a->H, a->W);
    const int cpw = 64 / p.HW;
    const size_t lds = (size_t)3 * p.B * 64 * 4;
    const dim3 grid((p.C + cpw - 1) / cpw);
    if (a->K == 3) {
        SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)mbconv_mid_bwd_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_B - 32 * 1024)));
        hipLaunchKernelGGL(mbconv_mid_bwd_kernel<3>, grid, dim3(64 * NW), lds, (hipStream_t)stream, p);
    } else {
        SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)mbconv_mid_bwd_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_B - 32 * 1024)));
        hipLaunchKernelGGL(mbconv_mid_bwd_kernel<5>, grid, dim3(64 * NW), lds, (hipStream_t)stream, p);
    }
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_bn_act_train_supported(int B, int C, int HW) {
    const int RW = row_width(HW);
    if (RW != 0 && B > 0 && C > 0 && 2 * fwd_lds(B, RW) <= (size_t)MAX_LDS_B) return 1;       // (the backward keeps dz and xhat)
    return large_ok(B, C, HW) ? 2 : 0;
}

extern "C" size_t srbh_bn_act_train_ws_bytes(int B, int C, int HW) {
    return srbh_bn_act_train_supported(B, C, HW) == 2 ? (size_t)C * large_splits(B, C) * 2 * sizeof(double) : 0;
}

extern "C" int srbh_bn_act_train_fwd(const srbh_bnact_args* a, void* stream) {
    SRBH_REQUIRE(a && a->x && a->y && a->gamma && a->beta && a->save_mean && a->save_invstd, "srbh_bn_act_train_fwd: null pointer");
    SRBH_REQUIRE(a->act >= 0 && a->act <= 2, "srbh_bn_act_train_fwd: act must be 0 (none), 1 (SiLU) or 2 (ReLU)");
    SRBH_REQUIRE(srbh_bn_act_train_supported(a->B, a->C, a->HW), "srbh_bn_act_train_fwd: unsupported shape B=%d C=%d HW=%d (planes of 1, 4, 16, 64 or 256 floats with B * row within LDS, or a multiple of 4 from 256 up)", a->B, a->C, a->HW);
    SRBH_REQUIRE((a->running_mean == nullptr) == (a->running_var == nullptr), "srbh_bn_act_train_fwd: running_mean / running_var go together");
    FwdP p;
    p.x = a->x; p.y = a->y; p.gamma = a->gamma; p.beta = a->beta; p.running_mean = a->running_mean; p.running_var = a->running_var;
    p.save_mean = a->save_mean; p.save_invstd = a->save_invstd; p.pooled = a->pooled; p.res = a->res; p.drop = a->drop;
    p.momentum = a->momentum; p.eps = a->eps; p.B = a->B; p.C = a->C; p.HW = a->HW;
    if (srbh_bn_act_train_supported(a->B, a->C, a->HW) == 2) {
        SRBH_REQUIRE(a->ws, "srbh_bn_act_train_fwd: planes of %d elements need the workspace (srbh_bn_act_train_ws_bytes)", a->HW);
        const int S = large_splits(p.B, p.C);
        double* part = (double*)a->ws;
        hipLaunchKernelGGL(bn_large_stats_kernel, dim3(p.C, S), dim3(256), 0, (hipStream_t)stream, p.x, part, p.B, p.C, p.HW, S);
        const dim3 gp((unsigned)((long)p.B * p.C));
        if (a->act == 0) hipLaunchKernelGGL(bn_large_apply_kernel<0>, gp, dim3(256), 0, (hipStream_t)stream, p, part, S);
        else if (a->act == 1) hipLaunchKernelGGL(bn_large_apply_kernel<1>, gp, dim3(256), 0, (hipStream_t)stream, p, part, S);
        else hipLaunchKernelGGL(bn_large_apply_kernel<2>, gp, dim3(256), 0, (hipStream_t)stream, p, part, S);
        SRBH_HIP(hipGetLastError());
        return SRBH_OK;
    }
    const int RW = row_width(a->HW), cpw = RW / p.HW;
    const size_t lds = fwd_lds(p.B, RW);
    const dim3 grid((p.C + cpw - 1) / cpw);
#define SRBH_FWD(A_, V_)                                                                                                                  \
    do {                                                                                                                                  \
        SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)bn_act_train_fwd_kernel<A_, V_>, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_B))); \
        hipLaunchKernelGGL((bn_act_train_fwd_kernel<A_, V_>), grid, dim3(64 * NW), lds, (hipStream_t)stream, p);                              \
    } while (0)
#define SRBH_FWD_V(A_)          \
    do {                        \
        if (RW == 256) SRBH_FWD(A_, 4); \
        else SRBH_FWD(A_, 1);   \
    } while (0)
    if (a->act == 0) SRBH_FWD_V(0);
    else if (a->act == 1) SRBH_FWD_V(1);
    else SRBH_FWD_V(2);
#undef SRBH_FWD_V
#undef SRBH_FWD
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_bn_act_train_bwd(const srbh_bnact_bwd_args* a, void* stream) {
    SRBH_REQUIRE(a && a->dy && a->x && a->gamma && a->beta && a->save_mean && a->save_invstd && a->dgamma && a->dbeta,
                 "srbh_bn_act_train_bwd: null pointer");
    SRBH_REQUIRE(a->act >= 0 && a->act <= 2, "srbh_bn_act_train_bwd: act must be 0 (none), 1 (SiLU) or 2 (ReLU)");
    SRBH_REQUIRE(srbh_bn_act_train_supported(a->B, a->C, a->HW), "srbh_bn_act_train_bwd: unsupported shape B=%d C=%d HW=%d", a->B, a->C, a->HW);
    SRBH_REQUIRE((a->gate == nullptr) == (a->dpooled == nullptr), "srbh_bn_act_train_bwd: gate / dpooled go together");
    BwdP p;
    p.dy = a->dy; p.x = a->x; p.gamma = a->gamma; p.beta = a->beta; p.save_mean = a->save_mean; p.save_invstd = a->save_invstd;
    p.gate = a->gate; p.dpooled = a->dpooled; p.drop = a->drop; p.dx = a->dx; p.dgamma = a->dgamma; p.dbeta = a->dbeta;
    p.B = a->B; p.C = a->C; p.HW = a->HW;
    if (srbh_bn_act_train_supported(a->B, a->C, a->HW) == 2) {
        SRBH_REQUIRE(a->ws, "srbh_bn_act_train_bwd: planes of %d elements need the workspace (srbh_bn_act_train_ws_bytes)", a->HW);
        const int S = large_splits(p.B, p.C);
        double* part = (double*)a->ws;
        const dim3 gs(p.C, S), gp((unsigned)((long)p.B * p.C));
        if (a->act == 0) {
            hipLaunchKernelGGL(bn_large_bwd_stats_kernel<0>, gs, dim3(256), 0, (hipStream_t)stream, p, part, S);
            hipLaunchKernelGGL(bn_large_bwd_apply_kernel<0>, gp, dim3(256), 0, (hipStream_t)stream, p, part, S);
        } else if (a->act == 1) {
            hipLaunchKernelGGL(bn_large_bwd_stats_kernel<1>, gs, dim3(256), 0, (hipStream_t)stream, p, part, S);
            hipLaunchKernelGGL(bn_large_bwd_apply_kernel<1>, gp, dim3(256), 0, (hipStream_t)stream, p, part, S);
        } else {
            hipLaunchKernelGGL(bn_large_bwd_stats_kernel<2>, gs, dim3(256), 0, (hipStream_t)stream, p, part, S);
            hipLaunchKernelGGL(bn_large_bwd_apply_kernel<2>, gp, dim3(256), 0, (hipStream_t)stream, p, part, S);
        }
        SRBH_HIP(hipGetLastError());
        return SRBH_OK;
    }
    const int RW = row_width(a->HW), cpw = RW / p.HW;
    const size_t lds = 2 * fwd_lds(p.B, RW);
    const dim3 grid((p.C + cpw - 1) / cpw);
#define SRBH_BWD(A_, V_)                                                                                                                  \
    do {                                                                                                                                  \
        SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)bn_act_train_bwd_kernel<A_, V_>, hipFuncAttributeMaxDynamicSharedMemorySize, MAX_LDS_B))); \
        hipLaunchKernelGGL((bn_act_train_bwd_kernel<A_, V_>), grid, dim3(64 * NW), lds, (hipStream_t)stream, p);                              \
    } while (0)
#define SRBH_BWD_V(A_)          \
    do {                        \
        if (RW == 256) SRBH_BWD(A_, 4); \
        else SRBH_BWD(A_, 1);   \
    } while (0)
    if (a->act == 0) SRBH_BWD_V(0);
    else if (a->act == 1) SRBH_BWD_V(1);
    else SRBH_BWD_V(2);
#undef SRBH_BWD_V
#undef SRBH_BWD
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_se_train_fwd(float* y, const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2,
                                 float* hidden, float* hidden_pre, float* gate, int B, int C, int SQ, int HW, void* stream) {
    SRBH_REQUIRE(y && pooled && w1 && b1 && w2 && b2 && hidden && hidden_pre && gate, "srbh_se_train_fwd: null pointer");
    SRBH_REQUIRE(B > 0 && C > 0 && SQ > 0 && SQ <= 65535 * 4 && HW > 0, "srbh_se_train_fwd: bad shape");
    const long planes = (long)B * C;
    hipLaunchKernelGGL(se_hidden_train_kernel, dim3(B, (SQ + 3) / 4), dim3(256), 0, (hipStream_t)stream, pooled, w1, b1, hidden,
                       hidden_pre, C, SQ);
    if (HW == 4 || HW == 16 || HW == 64 || HW == 256)
        hipLaunchKernelGGL(se_gate_scale_train_kernel, dim3((unsigned)((planes * HW + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, y,
                           hidden, w2, b2, gate, planes, C, SQ, HW);
    else
        hipLaunchKernelGGL(se_gate_scale_train_wave_kernel, dim3((unsigned)((planes + 3) / 4)), dim3(256), 0, (hipStream_t)stream, y,
                           hidden, w2, b2, gate, planes, C, SQ, HW);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_se_train_bwd(const float* dout, const float* x, const float* gamma, const float* beta, const float* save_mean,
                                 const float* save_invstd, const float* pooled, const float* hidden, const float* hidden_pre,
                                 const float* gate, const float* w1, const float* w2, float* ws, float* dpooled, float* dw1, float* db1,
                                 float* dw2, float* db2, int B, int C, int SQ, int HW, int act, void* stream) {
    SRBH_REQUIRE(dout && x && gamma && beta && save_mean && save_invstd && pooled && hidden && hidden_pre && gate && w1 && w2 && ws &&
                 dpooled && dw1 && db1 && dw2 && db2, "srbh_se_train_bwd: null pointer");
    SRBH_REQUIRE(B > 0 && C > 0 && SQ > 0 && HW > 0 && act >= 0 && act <= 2, "srbh_se_train_bwd: bad arguments");
    SRBH_REQUIRE(SQ <= SE_MAX_SQ, "srbh_se_train_bwd: SQ = %d exceeds %d", SQ, SE_MAX_SQ);
    const int nch = se_chunks(C), CC = (C + nch - 1) / nch;
    float* draw = ws;                       // [B][C]
    float* dsig = ws + (size_t)B * C;       // [B][C]
    float* dhp = dsig + (size_t)B * C;      // [B][SQ]
    float* part = dhp + (size_t)B * SQ;     // [B][nch][SQ]
    const long planes = (long)B * C;
    const long waves = HW < 64 ? (planes * HW + 63) / 64 : planes;
    const dim3 g1((unsigned)((waves + 3) / 4));
    SRBH_REQUIRE(HW >= 64 || (HW & (HW - 1)) == 0, "srbh_se_train_bwd: planes below 64 elements must be a power of two (HW = %d)", HW);
    if (act == 0) hipLaunchKernelGGL(se_bwd_dgate_kernel<0>, g1, dim3(256), 0, (hipStream_t)stream, dout, x, gamma, beta, save_mean, save_invstd, draw, planes, C, HW);
    else if (act == 1) hipLaunchKernelGGL(se_bwd_dgate_kernel<1>, g1, dim3(256), 0, (hipStream_t)stream, dout, x, gamma, beta, save_mean, save_invstd, draw, planes, C, HW);
    else hipLaunchKernelGGL(se_bwd_dgate_kernel<2>, g1, dim3(256), 0, (hipStream_t)stream, dout, x, gamma, beta, save_mean, save_invstd, draw, planes, C, HW);
    hipLaunchKernelGGL(se_bwd_expand_kernel, dim3(B, nch), dim3(256), 0, (hipStream_t)stream, draw, gate, w2, dsig, part, C, SQ, CC);
    hipLaunchKernelGGL(se_bwd_reduce_kernel, dim3(B, nch), dim3(256), 0, (hipStream_t)stream, part, hidden_pre, w1, dhp, dpooled, C, SQ, CC);
    GemmTN ga, gb;                 // dw2 [C][SQ] = dsig^T hidden (+ db2);  dw1 [SQ][C] = dhp^T pooled (+ db1)
    ga.A = dsig; ga.Bm = hidden; ga.out = dw2; ga.colsum = db2; ga.M = C; ga.N = SQ;
    gb.A = dhp; gb.Bm = pooled; gb.out = dw1; gb.colsum = db1; gb.M = SQ; gb.N = C;
    const int tn0 = (SQ + 31) / 32, tn1 = (C + 31) / 32;
    const int tiles0 = ((C + 31) / 32) * tn0, tiles1 = ((SQ + 31) / 32) * tn1;
    hipLaunchKernelGGL(se_bwd_params_kernel, dim3((unsigned)((tiles0 + tiles1 + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ga, gb, B,
                       tiles0, tn0, tn1);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" size_t srbh_se_train_bwd_ws_floats(int B, int C, int SQ) {
    return 2 * (size_t)B * C + (size_t)B * SQ + (size_t)B * se_chunks(C) * SQ;
}

extern "C" int srbh_up2_cat_fwd(const float* x, const float* skip, float* out, int B, int Cx, int Cs, int H, int W, void* stream) {
    SRBH_REQUIRE(x && out && (skip || Cs == 0), "srbh_up2_cat_fwd: null pointer");
    SRBH_REQUIRE(B > 0 && Cx > 0 && Cs >= 0 && H > 0 && W > 0 && (W & 1) == 0, "srbh_up2_cat_fwd: bad shape (W must be even)");
    const long total4 = (long)B * (Cx + Cs) * 2 * H * (2 * W / 4);
    const long blocks = (total4 + 255) / 256;
    hipLaunchKernelGGL(up2_cat_fwd_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, x, skip, out,
                       total4, Cx, Cs, H, W);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_up2_cat_bwd(const float* dout, float* dx, float* dskip, int B, int Cx, int Cs, int H, int W, void* stream) {
    SRBH_REQUIRE(dout && (dx || dskip), "srbh_up2_cat_bwd: null pointer");
    SRBH_REQUIRE(B > 0 && Cx > 0 && Cs >= 0 && H > 0 && W > 0 && (W & 1) == 0, "srbh_up2_cat_bwd: bad shape (W must be even)");
    const long n_x2 = dx ? (long)B * Cx * H * (W / 2) : 0, n_s4 = (dskip && Cs) ? (long)B * Cs * 2 * H * (2 * W / 4) : 0;
    if (n_x2 + n_s4 == 0) return SRBH_OK;
    const long blocks = (n_x2 + n_s4 + 255) / 256;
    hipLaunchKernelGGL(up2_cat_bwd_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, dout, dx, dskip,
                       n_x2, n_s4, Cx, Cs, H, W);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}
