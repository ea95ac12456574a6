// srbh_loss.hip -- loss and metric reductions of the training / validation loop on the device.
//
// Stands in for reference losses_pytorch/selfloss.py (MSE_adapt[_weight] :70-91, CE_DICE_adapt[_weight] :124-168, Dice
// :6-17) and metrics.py (SegmentationMetric.genConfusionMatrix :67-74, HeightMetric.addBatch :186-200).  The reference
// runs each loss as 6-12 full-resolution elementwise / reduction passes over (B,7,256,256) logits; here every loss is ONE
// read of its inputs for the forward sums and one read + one write for the gradient.  The device kernels only produce the
// *sums*; the handful of scalar operations around them (mean, Dice ratio, exp(-log_var) weighting) stay differentiable
// torch scalars in the Python mirror, so log_var keeps its gradient without any host synchronisation.
// All kernels are HBM-bound; reductions accumulate in fp64 (block tree in LDS, one atomic per block and quantity).
#include "srbh_internal.h"

namespace {
using namespace srbh;

template <int NQ>
__device__ __forceinline__ void block_reduce_add(double (&v)[NQ], double* out) {
    __shared__ double red[NQ][4];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        double x = v[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) x += __shfl_down(x, o, 64);
        if ((threadIdx.x & 63) == 0) red[q][threadIdx.x >> 6] = x;
    }
    __syncthreads();
    if (threadIdx.x < NQ) {
        double x = 0;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) x += red[threadIdx.x][w];
        atomicAdd(out + threadIdx.x, x);
    }
}

// sum_i w_i (p_i - t_i)^2
__global__ __launch_bounds__(256) void wmse_sum_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                       const float* __restrict__ w, long n, double* out) {
    double acc[1] = {0.0};
    float a = 0.f;
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    int k = 0;
    for (; i < n; i += stride) {
        const float d = p[i] - t[i];
        a += (w ? w[i] : 1.f) * d * d;
        if (++k == 64) { acc[0] += (double)a; a = 0.f; k = 0; }   // bounded fp32 partials
    }
    acc[0] += (double)a;
    block_reduce_add<1>(acc, out);
}

// grad_i = g * 2 w_i (p_i - t_i),  g = *gscale (device scalar: upstream gradient of the sum)
__global__ __launch_bounds__(256) void wmse_grad_kernel(const float* __restrict__ p, const float* __restrict__ t,
                                                        const float* __restrict__ w, long n, const float* __restrict__ gscale,
                                                        float* __restrict__ grad) {
    const float g = 2.f * gscale[0];
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) grad[i] = g * (w ? w[i] : 1.f) * (p[i] - t[i]);
}

struct CEGeo {
    int B, C;
    long HW, bs, cs, ps;   // element strides of the logits (batch, channel, pixel)
};
constexpr int MAXC = 16;

// per pixel: softmax over C, nll = -log s_y;  sums: [ sum w*nll, sum pb*tb, sum pb, sum tb ]  with pb = 1 - s_0 (the
// reference's softmax[:,1:].sum(1)) and tb = (y > 0)
template <bool GRAD>
__global__ __launch_bounds__(256) void cedice_kernel(const float* __restrict__ z, CEGeo g, const long long* __restrict__ y,
                                                     const float* __restrict__ w, double* sums, const float* __restrict__ gv,
                                                     float* __restrict__ dz) {
    const long total = (long)g.B * g.HW;
    double acc[4] = {0, 0, 0, 0};
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    if (GRAD) { g0 = gv[0]; g1 = gv[1]; g2 = gv[2]; }
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long b = i / g.HW, px = i - b * g.HW;
        const float* zp = z + b * g.bs + px * g.ps;
        float v[MAXC];
        float m = -3.0e38f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < g.C) { v[c] = zp[c * g.cs]; m = fmaxf(m, v[c]); }
        const int yy = (int)y[i];
        // nll = log(sum exp(z - m)) - (z_y - m): the log-softmax form (what nn.CrossEntropyLoss computes); taking
        // -log(exp(z_y - m) / sum) instead underflows to +inf once the target logit is ~87 below the maximum.
        // A label outside [0, C) raises in the reference; a kernel cannot, so it poisons the CE sum with NaN (loud).
        float zy = __builtin_nanf("");
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < g.C) {
                const float d = v[c] - m;
                if (c == yy) zy = d;
                v[c] = expf(d);
                se += v[c];
            }
        const float inv = 1.f / se;
        const float wi = w ? w[i] : 1.f;
        const float s0 = v[0] * inv;
        const float tb = yy > 0 ? 1.f : 0.f;
        if (!GRAD) {
            const float pb = 1.f - s0;
            acc[0] += (double)(wi * (logf(se) - zy));
            acc[1] += (double)(pb * tb);
            acc[2] += (double)pb;
            acc[3] += (double)tb;
        } else {
            // d(sum_ce)/dz_c = w (s_c - [c==y]);   d pb / dz_c = s_0 s_c - s_0 [c==0]
            const float gd = (g1 * tb + g2) * s0;
            float* dp = dz + b * g.bs + px * g.ps;
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < g.C) {
                    const float sc = v[c] * inv;
                    dp[c * g.cs] = g0 * wi * (sc - (c == yy ? 1.f : 0.f)) + gd * (sc - (c == 0 ? 1.f : 0.f));
                }
        }
    }
    if (!GRAD) block_reduce_add<4>(acc, sums);
}

// per class k: [ sum d^2, sum |d|, sum d, count ],  d = pred - ref over the pixels with cls == k
__global__ __launch_bounds__(256) void height_metric_kernel(const float* __restrict__ p, const float* __restrict__ r,
                                                            const long long* __restrict__ cls, long n, int nc, double* out) {
    __shared__ double s[MAXC * 4];
    for (int k = threadIdx.x; k < nc * 4; k += blockDim.x) s[k] = 0.0;
    __syncthreads();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int k = (int)cls[i];
        if (k < 0 || k >= nc) continue;
        const double d = (double)p[i] - (double)r[i];
        atomicAdd(&s[k * 4 + 0], d * d);
        atomicAdd(&s[k * 4 + 1], fabs(d));
        atomicAdd(&s[k * 4 + 2], d);
        atomicAdd(&s[k * 4 + 3], 1.0);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nc * 4; k += blockDim.x)
        if (s[k] != 0.0) atomicAdd(out + k, s[k]);
}

// cm[label][pred] += 1   (metrics.py:71-73: bincount(numClass * label + pred))
__global__ __launch_bounds__(256) void confusion_kernel(const long long* __restrict__ pred, const long long* __restrict__ label,
                                                        long n, int nc, unsigned long long* cm, int* bad) {
    __shared__ unsigned int s[MAXC * MAXC];
    for (int k = threadIdx.x; k < nc * nc; k += blockDim.x) s[k] = 0u;
    __syncthreads();
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const long long a = label[i], b = pred[i];
        if (a < 0 || a >= nc || b < 0 || b >= nc) { *bad = 1; continue; }
        atomicAdd(&s[(int)a * nc + (int)b], 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < nc * nc; k += blockDim.x)
        if (s[k]) atomicAdd(cm + k, (unsigned long long)s[k]);
}

inline int grid_for(long n) {
    long b = (n + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int srbh_wmse_sum(const float* pred, const float* target, const float* weight, long n, double* out_sum,
                             void* stream) {
    SRBH_REQUIRE(pred && target && out_sum && n > 0, "srbh_wmse_sum: bad arguments");
    hipLaunchKernelGGL(wmse_sum_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, pred, target, weight, n, out_sum);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_wmse_grad(const float* pred, const float* target, const float* weight, long n, const float* gscale,
                              float* grad, void* stream) {
    SRBH_REQUIRE(pred && target && gscale && grad && n > 0, "srbh_wmse_grad: bad arguments");
    hipLaunchKernelGGL(wmse_grad_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, pred, target, weight, n, gscale, grad);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_cedice_sums(const float* logits, int B, int C, long HW, long b_stride, long c_stride, long p_stride,
                                const long long* labels, const float* weight, double* out4, void* stream) {
    SRBH_REQUIRE(logits && labels && out4 && B > 0 && HW > 0, "srbh_cedice_sums: bad arguments");
    SRBH_REQUIRE(C >= 2 && C <= MAXC, "srbh_cedice_sums: 2..%d classes supported (got %d)", MAXC, C);
    CEGeo g{B, C, HW, b_stride, c_stride, p_stride};
    hipLaunchKernelGGL(cedice_kernel<false>, dim3(grid_for((long)B * HW)), dim3(256), 0, (hipStream_t)stream, logits, g, labels,
                       weight, out4, (const float*)nullptr, (float*)nullptr);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_cedice_grad(const float* logits, int B, int C, long HW, long b_stride, long c_stride, long p_stride,
                                const long long* labels, const float* weight, const float* g3, float* dlogits, void* stream) {
    SRBH_REQUIRE(logits && labels && g3 && dlogits && B > 0 && HW > 0, "srbh_cedice_grad: bad arguments");
    SRBH_REQUIRE(C >= 2 && C <= MAXC, "srbh_cedice_grad: 2..%d classes supported (got %d)", MAXC, C);
    CEGeo g{B, C, HW, b_stride, c_stride, p_stride};
    hipLaunchKernelGGL(cedice_kernel<true>, dim3(grid_for((long)B * HW)), dim3(256), 0, (hipStream_t)stream, logits, g, labels,
                       weight, (double*)nullptr, g3, dlogits);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_height_metric_sums(const float* pred, const float* ref, const long long* cls, long n, int num_class,
                                       double* out, void* stream) {
    SRBH_REQUIRE(pred && ref && cls && out && n > 0, "srbh_height_metric_sums: bad arguments");
    SRBH_REQUIRE(num_class >= 1 && num_class <= MAXC, "srbh_height_metric_sums: 1..%d classes supported (got %d)", MAXC, num_class);
    hipLaunchKernelGGL(height_metric_kernel, dim3(grid_for(n) > 512 ? 512 : grid_for(n)), dim3(256), 0, (hipStream_t)stream, pred,
                       ref, cls, n, num_class, out);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

extern "C" int srbh_confusion_add(const long long* pred, const long long* label, long n, int num_class,
                                  unsigned long long* cm, int* bad_flag, void* stream) {
    SRBH_REQUIRE(pred && label && cm && bad_flag && n > 0, "srbh_confusion_add: bad arguments");
    SRBH_REQUIRE(num_class >= 1 && num_class <= MAXC, "srbh_confusion_add: 1..%d classes supported (got %d)", MAXC, num_class);
    hipLaunchKernelGGL(confusion_kernel, dim3(grid_for(n) > 512 ? 512 : grid_for(n)), dim3(256), 0, (hipStream_t)stream, pred,
                       label, n, num_class, cm, bad_flag);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}
