// srbh_optim.hip -- the Adam update of the training step (reference: train.py:170-179 builds torch.optim.Adam(net.parameters(), lr, weight_decay=1e-4)
// plus a parameter group for the three loss log_vars; train.py:254-256 steps it) as ONE launch over all parameter tensors.
//
// torch's fused Adam reaches the ~700 tensors of SRRegress_Cls_feature through multi_tensor_apply: 17 + 8 launches whose kernel arguments
// carry the tensor pointers, 0.55 ms per step for 23 M parameters (644 MB of traffic: 0.12 ms at the HBM rate); the update sits between the
// end of backward and the next step's first kernel.  Here the pointers live in a device table (refreshed per step: the gradient tensors are
// new allocations) and one launch walks a static chunk list: chunk c = (tensor, 4096-element slice).  Same arithmetic as torch/optim/adam.py
// (weight decay added to the gradient, exp_avg by lerp, bias corrections on the host in double): parameters agree with torch.optim.Adam to
// fp32 rounding (tests/test_gpu_adam.py).
#include "srbh_internal.h"

namespace {
using namespace srbh;
typedef float floatx4 __attribute__((ext_vector_type(4)));
constexpr int CHUNK = 4096;

__global__ __launch_bounds__(256) void adam_kernel(const srbh_adam_entry* __restrict__ tab, const int* __restrict__ chunks /* [n][2]: tensor, slice */,
                                                    float beta1, float beta2, float omb1, float omb2, float eps) {
    const int ti = chunks[2 * blockIdx.x], sl = chunks[2 * blockIdx.x + 1];
    const srbh_adam_entry e = tab[ti];
    if (!e.g || e.n <= 0) return;                       // (a parameter that received no gradient this step)
    const long base = (long)sl * CHUNK;
    const long end = base + CHUNK < e.n ? base + CHUNK : e.n;
    const float step_size = e.lr * e.inv_bc1, inv_sqrt_bc2 = e.inv_sqrt_bc2;
    const bool vec = ((((uintptr_t)e.p | (uintptr_t)e.g | (uintptr_t)e.m | (uintptr_t)e.v) & 15) == 0);
    auto upd = [&](float p, float g, float& m, float& v) {
        g = fmaf(e.wd, p, g);                           // grad = grad + weight_decay * param
        m = m + (g - m) * omb1;                         // exp_avg.lerp_(grad, 1 - beta1)   (1 - beta as the host's double rounds it)
        v = fmaf(v, beta2, omb2 * g * g);               // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
        const float denom = sqrtf(v) * inv_sqrt_bc2 + eps;
        return p - step_size * (m / denom);
    };
    if (vec) {
        for (long i = base + threadIdx.x * 4; i < end; i += 256 * 4) {
            if (i + 4 <= end) {
                floatx4 p = *(const floatx4*)(e.p + i), m = *(const floatx4*)(e.m + i), v = *(const floatx4*)(e.v + i);
                const floatx4 g = *(const floatx4*)(e.g + i);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float mj = m[j], vj = v[j];
                    p[j] = upd(p[j], g[j], mj, vj);
                    m[j] = mj; v[j] = vj;
                }
                *(floatx4*)(e.p + i) = p; *(floatx4*)(e.m + i) = m; *(floatx4*)(e.v + i) = v;
            } else {
                for (long k = i; k < end; ++k) { float m = e.m[k], v = e.v[k]; e.p[k] = upd(e.p[k], e.g[k], m, v); e.m[k] = m; e.v[k] = v; }
            }
        }
    } else {
        for (long k = base + threadIdx.x; k < end; k += 256) { float m = e.m[k], v = e.v[k]; e.p[k] = upd(e.p[k], e.g[k], m, v); e.m[k] = m; e.v[k] = v; }
    }
}
}  // namespace

extern "C" int srbh_adam_chunk(void) { return CHUNK; }

extern "C" int srbh_adam_step(const srbh_adam_entry* table_dev, const int* chunks_dev, int nchunks, double beta1, double beta2, double eps,
                              void* stream) {
    SRBH_REQUIRE(table_dev && chunks_dev && nchunks > 0, "srbh_adam_step: null table / no chunks");
    SRBH_REQUIRE(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0, "srbh_adam_step: bad hyper-parameters");
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)nchunks), dim3(256), 0, (hipStream_t)stream, table_dev, chunks_dev, (float)beta1, (float)beta2,
                       (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}
