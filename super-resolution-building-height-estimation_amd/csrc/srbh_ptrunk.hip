// srbh_ptrunk.hip -- persistent kernel for the 23 x 3 x 5 dense-block convolutions of the RRDBNet trunk
// (reference SR/rrdbnet_arch.py:136-167, the 92 % of forward_feature's FLOPs).
//
// One launch runs every layer: a workgroup owns (image, 8 output rows) for the whole trunk and walks a device-side
// layer table.  What a per-layer launch pays on every conv (launch gap, cold prologue, store-drain tail) is paid once.
// Row-block neighbours exchange their 1-row halos INSIDE the launch:
//   producer : activations are stored write-through (global_store ... sc1), every wave drains vmcnt(0), barrier,
//              one lane publishes prog[tile] = layers completed (relaxed, agent scope);
//   consumer : one lane polls the two neighbours' prog words (relaxed agent loads + s_sleep, bounded), barrier,
//              then reads the activations with sc1 LDS-DMA (L1 bypass) -- placement independent (no reliance on
//              which XCD a workgroup landed on; the XCD-aware tile map only helps L2 locality).
// Skew between neighbours is <= 1 layer (a layer cannot start before both neighbours finished the previous one),
// while any plane is re-written no earlier than 5 layers after its last read, so there is no WAR hazard.
// All workgroups must be co-resident (1 per CU: the kernel uses the whole 160 KiB LDS): the host launches at most
// multiProcessorCount workgroups per call (sub-batches of images) and every spin is bounded -- on timeout the launch
// sets an error word and drains instead of hanging.
#include "srbh_conv3x3_kernel.h"

namespace {
using namespace srbh;
using namespace srbh_k;

struct PLayer {
    const char* w;
    const float* bias;
    int nchunk, cb, in_sel, out_sel, out_chunk0, flags;   // flags: 1 lrelu, 2 res1 (rdb stream), 4 res2 (rrdb stream)
};

struct PParams {
    char* dense[2];
    long img_b;
    int plane_b, row_b;
    float* xr;
    float* xrr;
    const PLayer* layers;
    int nlayers;
    int H, W, tiles_per_img, nblocks;
    int* prog;
    int* err;
};

constexpr unsigned SPIN_LIMIT = 4u << 20;

__global__ __launch_bounds__(256, 1) void ptrunk_kernel(const PParams pp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int t = xcd_remap(blockIdx.x, pp.nblocks);
    const int img = t / pp.tiles_per_img;
    const int ty = t - img * pp.tiles_per_img;
    const int up = ty > 0 ? t - 1 : -1, dn = ty + 1 < pp.tiles_per_img ? t + 1 : -1;
    for (int L = 0; L < pp.nlayers; ++L) {
        const PLayer lay = pp.layers[L];
        if (L > 0) {
            if (threadIdx.x == 0) {
                int bad = 0;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int nb = k ? dn : up;
                    if (nb < 0) continue;
                    unsigned spins = 0;
                    while (__hip_atomic_load(pp.prog + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < L) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > SPIN_LIMIT || __hip_atomic_load(pp.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            bad = 1;
                            break;
                        }
                    }
                }
                if (bad) __hip_atomic_store(pp.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *(volatile int*)smem = bad;
            }
            __syncthreads();
            const int bad = *(volatile int*)smem;
            __syncthreads();
            if (bad) return;   // uniform: the whole launch drains, the host sees err != 0
        }
        KParams p;
        p.in = pp.dense[lay.in_sel];
        p.in_img_b = pp.img_b;
        p.in_plane_b = pp.plane_b;
        p.in_row_b = pp.row_b;
        p.nchunk = lay.nchunk;
        p.w = lay.w;
        p.bias = lay.bias;
        p.H = pp.H;
        p.W = pp.W;
        p.tiles_x = 1;
        p.tiles_per_img = pp.tiles_per_img;
        p.nblocks = pp.nblocks;
        p.lrelu = lay.flags & 1;
        p.res_scale = 0.2f;
        p.res2_scale = 0.2f;
        p.res1 = (lay.flags & 2) ? pp.xr : nullptr;
        p.res2 = (lay.flags & 4) ? pp.xrr : nullptr;
        p.skip = nullptr;
        p.res1_update = 1;
        p.res2_update = 1;
        p.out16 = pp.dense[lay.out_sel] + (long)lay.out_chunk0 * pp.plane_b;
        p.out16_img_b = pp.img_b;
        p.out16_plane_b = pp.plane_b;
        p.out16_row_b = pp.row_b;
        p.out32 = nullptr;
        p.out32_c = 0;
        p.prof = nullptr;
        if (lay.cb == 1)
            conv_tile<1, 0, 0, 1>(p, smem, img, ty * TILE_H, 0, nullptr);
        else
            conv_tile<2, 0, 0, 1>(p, smem, img, ty * TILE_H, 0, nullptr);
        // publish: all of this workgroup's stores are complete (write-through) before the counter moves
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(pp.prog + t, L + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace

namespace srbh {

constexpr int MAX_BLOCKS = 64;   // layer-table capacity (RRDB blocks)
static size_t table_bytes() { return ((size_t)MAX_BLOCKS * 15 * sizeof(PLayer) + 255) & ~(size_t)255; }
static size_t prog_bytes(int B, int tpi) { return ((size_t)B * tpi * sizeof(int) + 255) & ~(size_t)255; }

size_t ptrunk_aux_bytes(int B, int tiles_per_img) { return table_bytes() + prog_bytes(B, tiles_per_img) + 256; }
size_t ptrunk_err_offset(int B, int tiles_per_img) { return table_bytes() + prog_bytes(B, tiles_per_img); }

// returns SRBH_OK and sets *used = 1 when the persistent path ran, *used = 0 when the shape is not eligible
int ptrunk_run(const srbh_rrdbnet_desc* d, void* dense0, void* dense1, float* xr, float* xrr, int B, int H, int W,
               void* aux, hipStream_t stream, int* used, int* final_cur) {
    *used = 0;
    if (W > TILE_W || d->num_block <= 0 || d->num_block > MAX_BLOCKS) return SRBH_OK;
    int dev = 0;
    SRBH_HIP(hipGetDevice(&dev));
    int ncu = 0;
    SRBH_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const int tpi = (H + TILE_H - 1) / TILE_H;
    if (tpi > ncu) return SRBH_OK;
    constexpr int LDS_B = lds_bytes<2, 0>() > lds_bytes<1, 0>() ? lds_bytes<2, 0>() : lds_bytes<1, 0>();
    static bool attr_set = false;
    if (!attr_set) {
        SRBH_HIP(hipFuncSetAttribute((const void*)ptrunk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B));
        attr_set = true;
    }
    int per_cu = 0;
    SRBH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ptrunk_kernel, 256, LDS_B));
    if (per_cu < 1) return SRBH_OK;

    const int nl = d->num_block * 15;
    std::vector<PLayer> tab(nl);
    int cur = 0, li = 0;
    for (int blk = 0; blk < d->num_block; ++blk)
        for (int r = 0; r < 3; ++r) {
            const srbh_conv_w* cw = d->rdb + (blk * 3 + r) * 5;
            for (int k = 0; k < 4; ++k) tab[li++] = PLayer{(const char*)cw[k].w, cw[k].bias, 2 + k, 1, cur, cur, 2 + k, 1};
            tab[li++] = PLayer{(const char*)cw[4].w, cw[4].bias, 6, 2, cur, cur ^ 1, 0, 2 | (r == 2 ? 4 : 0)};
            cur ^= 1;
        }
    *final_cur = cur;
    char* a = (char*)aux;
    PLayer* d_tab = (PLayer*)a;
    int* d_prog = (int*)(a + table_bytes());
    int* d_err = (int*)(a + ptrunk_err_offset(B, tpi));
    SRBH_HIP(hipMemcpyAsync(d_tab, tab.data(), (size_t)nl * sizeof(PLayer), hipMemcpyHostToDevice, stream));
    SRBH_HIP(hipMemsetAsync(d_prog, 0, (size_t)B * tpi * sizeof(int) , stream));
    SRBH_HIP(hipMemsetAsync(d_err, 0, sizeof(int), stream));
    const Act16Geo g = act16_geo(B, 6, H, W);
    const int imgs_per_launch = ncu / tpi;
    for (int b0 = 0; b0 < B; b0 += imgs_per_launch) {
        const int nb = (B - b0) < imgs_per_launch ? (B - b0) : imgs_per_launch;
        PParams pp;
        pp.dense[0] = (char*)dense0 + (long)b0 * g.img_b;
        pp.dense[1] = (char*)dense1 + (long)b0 * g.img_b;
        pp.img_b = g.img_b;
        pp.plane_b = g.plane_b;
        pp.row_b = g.row_b;
        pp.xr = xr + (long)b0 * H * W * 64;
        pp.xrr = xrr + (long)b0 * H * W * 64;
        pp.layers = d_tab;
        pp.nlayers = nl;
        pp.H = H;
        pp.W = W;
        pp.tiles_per_img = tpi;
        pp.nblocks = nb * tpi;
        pp.prog = d_prog + b0 * tpi;
        pp.err = d_err;
        hipLaunchKernelGGL(ptrunk_kernel, dim3(pp.nblocks), dim3(256), LDS_B, stream, pp);
        SRBH_HIP(hipGetLastError());
    }
    *used = 1;
    return SRBH_OK;
}

}  // namespace srbh
