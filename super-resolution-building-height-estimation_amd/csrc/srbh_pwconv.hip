// srbh_pwconv.hip -- the 1x1 ("pointwise") convolutions of the MBConv blocks (expand / project, no bias, stride 1), fp32 NCHW:
// forward, input gradient, weight gradient.
//
// Why it exists: in the training step MIOpen serves these 62 convolutions with NHWC implicit-GEMM / Tensile kernels wrapped in batched
// transposes, zero fills and split-K atomics: ~390 launches and ~4.1 ms of a 39.5 ms step (profiles/r03c), for 23 GFLOP of work.
// The shapes are small GEMMs -- M, K = 24 ... 2688 channels, N = B * HW = 256 ... 16 384 pixels -- so what matters is (a) ONE launch
// per product, straight on the NCHW tensors, and (b) enough waves: a 32x32 tile leaves most of the 1 024 SIMDs idle at 2x2 and 4x4,
// so the tile is 16x16 (v_mfma_f32_16x16x4_f32, true fp32, fixed summation order) and grows to 16x32 / 32x32 only when the grid stays
// above ~1 500 waves.
//
//   forward : Y[b][co][p] = sum_ci W[co][ci] X[b][ci][p]          M = Cout, K = Cin,  A(m, k) = W[m K + k]
//   dgrad   : dX[b][ci][p] = sum_co W[co][ci] dY[b][co][p]        M = Cin,  K = Cout, A(m, k) = W[k M + m]
//   wgrad   : dW[co][ci] = sum_{b, p} dY[b][co][p] X[b][ci][p]    M = Cout, N = Cin,  K = B * HW, split over images when the tile grid is
//             small (partials in fixed slots + one ordered reduce: deterministic, nothing to zero)
//
// Operands go from global memory (L2-resident: the largest weight is 4.8 MB) straight into the MFMA: lane l holds A[i = l % 16][k = l / 16]
// and B[k = l / 16][j = l % 16].  The K index a lane group walks is free as long as A and B agree, so group kq takes the contiguous
// quarter [kq K/4, (kq+1) K/4): every lane then walks its own row / plane sequentially.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <mutex>
#include <vector>
#include "srbh.h"
#include "srbh_internal.h"

namespace {
using namespace srbh;
typedef float floatx4 __attribute__((ext_vector_type(4)));
#ifndef SRBH_PW_UK
#define SRBH_PW_UK 8
#endif
constexpr int UK = SRBH_PW_UK;      // K steps per round of loads (a multiple of 4)

// KW waves share one tile and split its K range between them (partials folded through LDS in a fixed order): the deep products at 2x2 and
// 4x4 (K up to 2688 with only ~300 tiles) are otherwise one long chain of load rounds on a quarter of the SIMDs.  KW = 1: the 4 waves
// of a workgroup own 4 tiles.
template <int NREG>
__device__ __forceinline__ void fold_k_slices(floatx4 (&acc)[NREG], float* red, int KW, int ks, int lane) {
    // red: [KW][NREG * 4][64]
#pragma unroll
    for (int q = 0; q < NREG; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(ks * NREG * 4 + q * 4 + r) * 64 + lane] = acc[q][r];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NREG; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = 0.f;
            if ((q * 4 + r) % KW == ks)                          // output register o belongs to wave o % KW
                for (int k = 0; k < KW; ++k) v += red[(k * NREG * 4 + q * 4 + r) * 64 + lane];
            acc[q][r] = v;
        }
}

// EPI (inference MBConv blocks, round 4): the operand is multiplied by a per-(image, input channel) gate while it is loaded (squeeze-excite
// folded into the project conv) and the result goes through y = act(acc * scale[m] + shift[m]) [+ res] (the folded inference BatchNorm, the
// block's skip connection) before the one store.
struct PwEpi {
    const float* gate;       // [B][K] or null
    const float* scale;      // [M] (null: no affine)
    const float* shift;
    const float* res;        // [B][M][HW] or null
    int act;                 // 0 none, 1 SiLU, 2 ReLU
};
template <int MI, int NI, int TRANS_A, int KW, int EPI = 0>
__global__ __launch_bounds__(KW == 16 ? 1024 : 256) void pw_gemm_kernel(const float* __restrict__ W, const float* __restrict__ In,
                                                                      float* __restrict__ Out, int M, int K, int HW, long ncols,
                                                                      int tiles_m, int tiles_n, const PwEpi ep) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    const int lane = threadIdx.x & 63, l16 = lane & 15, kq = lane >> 4, wave = threadIdx.x >> 6;
    const int ks = KW == 1 ? 0 : wave;
    const long wid = KW == 1 ? (long)blockIdx.x * 4 + wave : (long)blockIdx.x;
    const int tm = (int)(wid / tiles_n), tn = (int)(wid - (long)tm * tiles_n);
    if (tm >= tiles_m) return;                                     // (KW == 1 only: whole waves; with KW > 1 the grid is exact)
    const int Kw = KW == 1 ? K : (((K + KW - 1) / KW + 3) & ~3);  // K range of one wave
    const int kend = (ks + 1) * Kw < K ? (ks + 1) * Kw : K;
    const int K4 = (Kw + 3) >> 2, kbeg = ks * Kw + kq * K4;
    int kcnt = kend - kbeg;
    kcnt = kcnt < 0 ? 0 : (kcnt > K4 ? K4 : kcnt);
    const float* ap[MI];
    bool aok[MI];
    const long astride = TRANS_A ? M : 1;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int m = (tm * MI + mi) * 16 + l16;
        aok[mi] = m < M;
        ap[mi] = TRANS_A ? W + (long)kbeg * M + m : W + (long)m * K + kbeg;
    }
    const float* bp[NI];
    const float* gp[NI];
    bool bok[NI];
    long obase[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const long n = ((long)tn * NI + ni) * 16 + l16;
        bok[ni] = n < ncols;
        const long b = n / HW;
        const int hw = (int)(n - b * HW);
        bp[ni] = In + (b * K + kbeg) * HW + hw;
        gp[ni] = (EPI && ep.gate) ? ep.gate + (bok[ni] ? b : 0) * K + kbeg : nullptr;
        obase[ni] = b * M * HW + hw;
    }
    floatx4 acc[MI * NI];
#pragma unroll
    for (int q = 0; q < MI * NI; ++q) acc[q] = floatx4{0.f, 0.f, 0.f, 0.f};
    for (int u0 = 0; u0 < K4; u0 += UK) {
        float a[MI][UK], b[NI][UK];
#pragma unroll
        for (int u = 0; u < UK; ++u) {
            const bool kok = u0 + u < kcnt;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi][u] = (kok && aok[mi]) ? ap[mi][(u0 + u) * astride] : 0.f;
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni][u] = (kok && bok[ni]) ? bp[ni][(long)(u0 + u) * HW] : 0.f;
        }
        if (EPI && ep.gate) {
            // the gates of this lane's K run are contiguous floats: two (unaligned) 16-byte loads per round instead of one load per element
            typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) {
                float gv[UK];
                if (u0 + UK <= kcnt) {
#pragma unroll
                    for (int v = 0; v < UK / 4; ++v) {
                        const f4u g4 = *(const f4u*)(gp[ni] + u0 + 4 * v);
                        gv[4 * v] = g4[0]; gv[4 * v + 1] = g4[1]; gv[4 * v + 2] = g4[2]; gv[4 * v + 3] = g4[3];
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < UK; ++u) gv[u] = u0 + u < kcnt ? gp[ni][u0 + u] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < UK; ++u) b[ni][u] *= gv[u];
            }
        }
#pragma unroll
        for (int u = 0; u < UK; ++u)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi * NI + ni] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mi][u], b[ni][u], acc[mi * NI + ni], 0, 0, 0);
    }
    if (KW > 1) fold_k_slices<MI * NI>(acc, red, KW, ks, lane);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (KW > 1 && ((mi * NI + ni) * 4 + r) % KW != ks) continue;
                const int m = (tm * MI + mi) * 16 + 4 * kq + r;
                if (m < M && bok[ni]) {
                    float v = acc[mi * NI + ni][r];
                    if (EPI) {
                        if (ep.scale) v = fmaf(v, ep.scale[m], ep.shift[m]);
                        if (ep.act == 1) v = v / (1.f + __expf(-v));
                        else if (ep.act == 2) v = fmaxf(v, 0.f);
                        if (ep.res) v += ep.res[obase[ni] + (long)m * HW];
                    }
                    Out[obase[ni] + (long)m * HW] = v;
                }
            }
}

// weight gradient.  K = (image, pixel); lane group kq walks the quarter [kq HW/4, (kq+1) HW/4) of every plane of its images, so at 2x2 the
// 16 lanes x 4 groups of one load cover 16 whole planes = 256 contiguous bytes, and from 4x4 up a lane reads float4 runs of its plane.
// The KW waves of a workgroup share one 16x16 tile and take every KW-th image of the split (folded through LDS); grid.y = S further splits
// over images write partials for the ordered reduce.
template <int VEC, int KW>
__device__ __forceinline__ void pw_wgrad_body(const float* __restrict__ dY, const float* __restrict__ X, float* __restrict__ out, int M, int N,
                                              int HW, int B, int S, int tiles_n, const int bx, const int by, float* red) {
    const int lane = threadIdx.x & 63, l16 = lane & 15, kq = lane >> 4, ks = threadIdx.x >> 6;
    const int tm = bx / tiles_n, tn = bx - tm * tiles_n;
    const int first = by * KW + ks, stride = S * KW, HQ = HW >> 2;      // images first, first + stride, ...
    const int m = tm * 16 + l16, n = tn * 16 + l16;
    const bool aok = m < M, bok = n < N;
    const float* ap = dY + (long)m * HW + kq * HQ;
    const float* bp = X + (long)n * HW + kq * HQ;
    const long sa = (long)M * HW, sb = (long)N * HW;
    floatx4 acc[1] = {floatx4{0.f, 0.f, 0.f, 0.f}};
    if (VEC == 1) {                     // HQ == 1 (2x2 planes): one K step per image, UK images per round of loads
        for (int b0 = first; b0 < B; b0 += stride * UK) {
            float a[UK], b[UK];
#pragma unroll
            for (int u = 0; u < UK; ++u) {
                const long bi = b0 + (long)u * stride;
                a[u] = (bi < B && aok) ? ap[bi * sa] : 0.f;
                b[u] = (bi < B && bok) ? bp[bi * sb] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < UK; ++u) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], b[u], acc[0], 0, 0, 0);
        }
    } else {                            // HQ a multiple of 4: float4 runs, 4 runs (16 K steps) per round
        // K slots = (image, float4 run of the lane's quarter); slice `first` of `stride` takes every stride-th slot, so that a product
        // with few tiles and large planes (48 -> 24 at 32x32: 6 tiles, K = 65 536) still spreads over the chip
        const int runs = HQ >> 2;
        const long slots = (long)B * runs;
        for (long q0 = first; q0 < slots; q0 += 4L * stride) {
            floatx4 a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long q = q0 + (long)u * stride;
                const long bi = q / runs;
                const int off = (int)(q - bi * runs) * 4;
                a[u] = (q < slots && aok) ? *(const floatx4*)(ap + bi * sa + off) : floatx4{0.f, 0.f, 0.f, 0.f};
                b[u] = (q < slots && bok) ? *(const floatx4*)(bp + bi * sb + off) : floatx4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][e], b[u][e], acc[0], 0, 0, 0);
        }
    }
    fold_k_slices<1>(acc, red, KW, ks, lane);
    float* o = out + (long)by * M * N;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (r % KW != ks) continue;
        const int mm = tm * 16 + 4 * kq + r;
        if (mm < M && bok) o[(long)mm * N + n] = acc[0][r];
    }
}
template <int VEC, int KW>
__global__ __launch_bounds__(KW == 16 ? 1024 : 256) void pw_wgrad_kernel(const float* __restrict__ dY, const float* __restrict__ X,
                                                                       float* __restrict__ out, int M, int N, int HW, int B, int S,
                                                                       int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) float red[];
    pw_wgrad_body<VEC, KW>(dY, X, out, M, N, HW, B, S, tiles_n, blockIdx.x, blockIdx.y, red);
}

__global__ __launch_bounds__(256) void pw_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, long n, int S) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float a0 = 0.f, a1 = 0.f;
    int s = 0;
    for (; s + 1 < S; s += 2) {
        a0 += part[(long)s * n + i];
        a1 += part[(long)(s + 1) * n + i];
    }
    if (s < S) a0 += part[(long)s * n + i];
    dw[i] = a0 + a1;
}

// W [rows][cols] -> W^T [cols][rows] for a table of matrices in one launch (the forward reads the weights as the input gradient does:
// coalesced along the output channel; with A(m, k) = W[m K + k] every lane walks its own row and the forward ran 2x slower than dgrad)
__global__ __launch_bounds__(256) void transpose_many_kernel(const srbh_transpose_desc* __restrict__ table) {
    __shared__ float tile[32][33];
    const srbh_transpose_desc d = table[blockIdx.y];
    const int tr = (d.rows + 31) / 32, tc = (d.cols + 31) / 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int t = blockIdx.x; t < tr * tc; t += gridDim.x) {
        const int r0 = (t / tc) * 32, c0 = (t % tc) * 32;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int r = r0 + ty + 8 * k, c = c0 + tx;
            tile[ty + 8 * k][tx] = (r < d.rows && c < d.cols) ? d.src[(long)r * d.cols + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = c0 + ty + 8 * k, r = r0 + tx;
            if (r < d.rows && c < d.cols) d.dst[(long)c * d.rows + r] = tile[tx][ty + 8 * k];
        }
    }
}

constexpr long WAVES_WANTED = 1536;
int pick_tile(long m16, long n16) {               // -> 0: 16x16, 1: 16x32, 2: 32x32 (the largest whose grid still fills the chip)
    if (((m16 + 1) / 2) * ((n16 + 1) / 2) >= WAVES_WANTED) return 2;
    if (m16 * ((n16 + 1) / 2) >= WAVES_WANTED) return 1;
    return 0;
}
int pick_kw(long tiles, int K) {                  // waves per tile: split K inside the workgroup when tiles are few and K is deep
    const int rounds = (K / 4 + UK - 1) / UK;
    if (tiles >= WAVES_WANTED || rounds <= 2) return 1;
    if (tiles * 4 >= WAVES_WANTED || rounds <= 8) return 4;
    return 16;
}
struct WgradPlan { int KW, S; };
WgradPlan wgrad_plan(int M, int N, int B, int HW) {
    const long tiles = (long)((M + 15) / 16) * ((N + 15) / 16);
    WgradPlan p;
    p.KW = (tiles >= 1024 || B < 16) ? 4 : 16;
    long S = (3 * WAVES_WANTED + tiles * p.KW - 1) / (tiles * p.KW);    // (~4 waves per SIMD: the K walk is a chain of scattered plane reads)
    const long slots = HW == 4 ? B : (long)B * (HW >> 4);            // K slots a slice strides over (images, or float4 runs)
    const long smax = slots / (p.KW * 4L) > 0 ? slots / (p.KW * 4L) : 1;   // (leave every wave at least ~4 slots)
    S = S < 1 ? 1 : (S > smax ? smax : S);
    S = S > 64 ? 64 : S;
    p.S = (int)S;
    return p;
}

template <int MI, int NI, int TRANS_A>
int gemm_launch_kw(int kw, const float* W, const float* In, float* Out, int M, int K, int HW, long ncols, int tm, int tn, hipStream_t st) {
    const long tiles = (long)tm * tn;
    if (kw == 1)
        hipLaunchKernelGGL((pw_gemm_kernel<MI, NI, TRANS_A, 1>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, W, In, Out, M, K, HW, ncols, tm, tn, PwEpi{});
    else if (kw == 4)
        hipLaunchKernelGGL((pw_gemm_kernel<MI, NI, TRANS_A, 4>), dim3((unsigned)tiles), dim3(256), 4 * MI * NI * 4 * 64 * 4, st, W, In, Out, M, K, HW, ncols, tm, tn, PwEpi{});
    else
        hipLaunchKernelGGL((pw_gemm_kernel<MI, NI, TRANS_A, 16>), dim3((unsigned)tiles), dim3(1024), 16 * MI * NI * 4 * 64 * 4, st, W, In, Out, M, K, HW, ncols, tm, tn, PwEpi{});
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

template <int MI, int NI, int TRANS_A>
int gemm_launch_epi_kw(int kw, const float* W, const float* In, float* Out, int M, int K, int HW, long ncols, int tm, int tn, const PwEpi& ep,
                       hipStream_t st) {
    const long tiles = (long)tm * tn;
    if (kw == 1)
        hipLaunchKernelGGL((pw_gemm_kernel<MI, NI, TRANS_A, 1, 1>), dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st, W, In, Out, M, K, HW, ncols, tm, tn, ep);
    else if (kw == 4)
        hipLaunchKernelGGL((pw_gemm_kernel<MI, NI, TRANS_A, 4, 1>), dim3((unsigned)tiles), dim3(256), 4 * MI * NI * 4 * 64 * 4, st, W, In, Out, M, K, HW, ncols, tm, tn, ep);
    else
        hipLaunchKernelGGL((pw_gemm_kernel<MI, NI, TRANS_A, 16, 1>), dim3((unsigned)tiles), dim3(1024), 16 * MI * NI * 4 * 64 * 4, st, W, In, Out, M, K, HW, ncols, tm, tn, ep);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

#include "srbh_pwgemm_lds_kernel.h"      // the LDS-tiled form for wide products (the tiled prediction's batches)

template <int TRANS_A>
int gemm_launch_epi(const float* W, const float* In, float* Out, int M, int K, int HW, int B, const PwEpi& ep, hipStream_t st) {
    const long ncols = (long)B * HW, m16 = (M + 15) / 16, n16 = (ncols + 15) / 16;
    const PwLdsPlan lp = pw_lds_plan(M, K, HW, ncols);
    if (lp.form) return pw_lds_launch<TRANS_A, 1>(lp, W, In, Out, M, K, HW, ncols, ep, st);
    const int t = pick_tile(m16, n16);
    const int MI = t == 2 ? 2 : 1, NI = t >= 1 ? 2 : 1;
    const int tm = (int)((m16 + MI - 1) / MI), tn = (int)((n16 + NI - 1) / NI);
    const int kw = t == 0 ? pick_kw((long)tm * tn, K) : 1;
    if (t == 2) return gemm_launch_epi_kw<2, 2, TRANS_A>(1, W, In, Out, M, K, HW, ncols, tm, tn, ep, st);
    if (t == 1) return gemm_launch_epi_kw<1, 2, TRANS_A>(1, W, In, Out, M, K, HW, ncols, tm, tn, ep, st);
    return gemm_launch_epi_kw<1, 1, TRANS_A>(kw, W, In, Out, M, K, HW, ncols, tm, tn, ep, st);
}

template <int TRANS_A>
int gemm_launch(const float* W, const float* In, float* Out, int M, int K, int HW, int B, hipStream_t st) {
    const long ncols = (long)B * HW, m16 = (M + 15) / 16, n16 = (ncols + 15) / 16;
    const PwLdsPlan lp = pw_lds_plan(M, K, HW, ncols);
    if (lp.form) return pw_lds_launch<TRANS_A, 0>(lp, W, In, Out, M, K, HW, ncols, PwEpi{}, st);
    const int t = pick_tile(m16, n16);
    const int MI = t == 2 ? 2 : 1, NI = t >= 1 ? 2 : 1;
    const int tm = (int)((m16 + MI - 1) / MI), tn = (int)((n16 + NI - 1) / NI);
    const int kw = t == 0 ? pick_kw((long)tm * tn, K) : 1;
    if (t == 2) return gemm_launch_kw<2, 2, TRANS_A>(1, W, In, Out, M, K, HW, ncols, tm, tn, st);
    if (t == 1) return gemm_launch_kw<1, 2, TRANS_A>(1, W, In, Out, M, K, HW, ncols, tm, tn, st);
    return gemm_launch_kw<1, 1, TRANS_A>(kw, W, In, Out, M, K, HW, ncols, tm, tn, st);
}
}  // namespace

extern "C" int srbh_pwconv_fwd(const float* x, const float* w, float* y, int B, int Cin, int Cout, int HW, void* stream) {
    SRBH_REQUIRE(x && w && y, "srbh_pwconv_fwd: null pointer");
    SRBH_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && HW > 0, "srbh_pwconv_fwd: bad shape");
    return gemm_launch<0>(w, x, y, Cout, Cin, HW, B, (hipStream_t)stream);
}

extern "C" int srbh_pwconv_fwd_epi(const float* x, const float* w, int w_transposed, float* y, int B, int Cin, int Cout, int HW,
                                   const float* gate, const float* scale, const float* shift, const float* res, int act, void* stream) {
    SRBH_REQUIRE(x && w && y, "srbh_pwconv_fwd_epi: null pointer");
    SRBH_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && HW > 0, "srbh_pwconv_fwd_epi: bad shape");
    SRBH_REQUIRE((scale == nullptr) == (shift == nullptr) && act >= 0 && act <= 2, "srbh_pwconv_fwd_epi: bad epilogue");
    const PwEpi ep{gate, scale, shift, res, act};
    return w_transposed ? gemm_launch_epi<1>(w, x, y, Cout, Cin, HW, B, ep, (hipStream_t)stream)
                        : gemm_launch_epi<0>(w, x, y, Cout, Cin, HW, B, ep, (hipStream_t)stream);
}

extern "C" int srbh_pwconv_fwd_wt(const float* x, const float* wt, float* y, int B, int Cin, int Cout, int HW, void* stream) {
    SRBH_REQUIRE(x && wt && y, "srbh_pwconv_fwd_wt: null pointer");
    SRBH_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && HW > 0, "srbh_pwconv_fwd_wt: bad shape");
    return gemm_launch<1>(wt, x, y, Cout, Cin, HW, B, (hipStream_t)stream);
}

extern "C" int srbh_transpose_many(const srbh_transpose_desc* table_dev, int n, void* stream) {
    SRBH_REQUIRE(table_dev && n > 0 && n <= 65535, "srbh_transpose_many: bad arguments");
    hipLaunchKernelGGL(transpose_many_kernel, dim3(256, n), dim3(256), 0, (hipStream_t)stream, table_dev);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_pwconv_bwd_data(const float* dy, const float* w, float* dx, int B, int Cin, int Cout, int HW, void* stream) {
    SRBH_REQUIRE(dy && w && dx, "srbh_pwconv_bwd_data: null pointer");
    SRBH_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && HW > 0, "srbh_pwconv_bwd_data: bad shape");
    return gemm_launch<1>(w, dy, dx, Cin, Cout, HW, B, (hipStream_t)stream);
}

/* dX = W^T dY + res: the input gradient with the gradient that arrives over the block's skip connection added in the store (the two used
 * to meet in an element-wise add launched by autograd: ~25 launches of 7 us per step) */
extern "C" int srbh_pwconv_bwd_data_res(const float* dy, const float* w, const float* res, float* dx, int B, int Cin, int Cout, int HW, void* stream) {
    SRBH_REQUIRE(dy && w && res && dx, "srbh_pwconv_bwd_data_res: null pointer");
    SRBH_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && HW > 0, "srbh_pwconv_bwd_data_res: bad shape");
    PwEpi ep = {};
    ep.res = res;
    return gemm_launch_epi<1>(w, dy, dx, Cin, Cout, HW, B, ep, (hipStream_t)stream);
}

extern "C" size_t srbh_pwconv_bwd_weight_ws_floats(int B, int Cin, int Cout, int HW) {
    const WgradPlan p = wgrad_plan(Cout, Cin, B, HW);
    return p.S > 1 ? (size_t)p.S * Cout * Cin : 0;
}

extern "C" int srbh_pwconv_bwd_weight(const float* x, const float* dy, float* dw, float* ws, int B, int Cin, int Cout, int HW, void* stream) {
    SRBH_REQUIRE(x && dy && dw, "srbh_pwconv_bwd_weight: null pointer");
    SRBH_REQUIRE(B > 0 && Cin > 0 && Cout > 0 && HW > 0, "srbh_pwconv_bwd_weight: bad shape");
    SRBH_REQUIRE(HW == 4 || (HW & 15) == 0, "srbh_pwconv_bwd_weight: planes of 4 or a multiple of 16 elements (HW = %d)", HW);
    const WgradPlan p = wgrad_plan(Cout, Cin, B, HW);
    SRBH_REQUIRE(p.S == 1 || ws, "srbh_pwconv_bwd_weight: this shape needs the workspace (srbh_pwconv_bwd_weight_ws_floats)");
    float* out = p.S > 1 ? ws : dw;
    const int tm = (Cout + 15) / 16, tn = (Cin + 15) / 16;
    const dim3 grid((unsigned)(tm * tn), p.S);
    hipStream_t st = (hipStream_t)stream;
#define SRBH_WG(V_, K_) hipLaunchKernelGGL((pw_wgrad_kernel<V_, K_>), grid, dim3(64 * K_), K_ * 4 * 64 * 4, st, dy, x, out, Cout, Cin, HW, B, p.S, tn)
    if (HW == 4) {
        if (p.KW == 16) SRBH_WG(1, 16);
        else SRBH_WG(1, 4);
    } else {
        if (p.KW == 16) SRBH_WG(4, 16);
        else SRBH_WG(4, 4);
    }
#undef SRBH_WG
    if (p.S > 1) {
        const long n = (long)Cout * Cin;
        hipLaunchKernelGGL(pw_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, ws, dw, n, p.S);
    }
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_pwconv_supported(int B, int Cin, int Cout, int HW) {
    return B > 0 && Cin > 0 && Cout > 0 && (HW == 4 || (HW > 0 && (HW & 15) == 0));
}
