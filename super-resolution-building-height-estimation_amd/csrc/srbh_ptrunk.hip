// srbh_ptrunk.hip -- persistent kernel for the 23 x 3 x 5 dense-block convolutions of the RRDBNet trunk
// (reference SR/rrdbnet_arch.py:136-167, the 92 % of forward_feature's FLOPs).
//
// One launch runs every layer: a workgroup owns (image, 8 output rows) for the whole trunk and walks a device-side
// layer table.  What a per-layer launch pays on every conv (launch gap, cold prologue, store-drain tail) is paid once.
// Row-block neighbours exchange their 1-row halos INSIDE the launch:
//   producer : activations are stored, every wave drains vmcnt(0), barrier, one lane publishes
//              prog[tile] = layers completed (relaxed atomic);
//   consumer : one lane polls the two neighbours' prog words (relaxed agent loads + s_sleep, bounded), barrier,
//              then reads the activations with sc1 LDS-DMA (L1 bypass).
// The tile map (xcd_remap) puts the row blocks of one image on one XCD, so the exchange normally stays inside that
// XCD's L2: plain stores, complete once they are in L2.  Placement is verified at kernel start (XCC_ID handshake with
// both neighbours); a workgroup whose neighbour sits on another XCD falls back to write-through (sc1) stores and
// agent-scope publishes, which are placement independent (measured: +6 % whole-forward for the in-L2 exchange).
// Skew between neighbours is <= 1 layer (a layer cannot start before both neighbours finished the previous one),
// while any plane is re-written no earlier than 5 layers after its last read, so there is no WAR hazard.
// All workgroups must be co-resident (1 per CU: the kernel uses the whole 160 KiB LDS): the host launches at most
// multiProcessorCount workgroups per call (sub-batches of images) and every spin is bounded -- on timeout the launch
// sets an error word and drains instead of hanging.
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include "srbh_conv3x3_kernel.h"


namespace {
using namespace srbh;
using namespace srbh_k;

struct PLayer {
    const char* w;
    const float* bias;
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
    int* xcc;                   // [nblocks] XCC_ID + 1 of every workgroup (placement handshake)
    int force_wt;               // 1: always use write-through stores (debugging aid, env SRBH_PT_WT=1)
    unsigned long long* prof;   // debug (tools/convbench): [block][layer][4] s_memtime stamps, nullptr in production
    // TRAINING forward (ptrunk3_kernel only; srbh_rrdbnet_trunk_train_forward_persistent): every RDB keeps its own dense buffer for the backward
    long dense_stride;          // > 0: RDB i reads / writes dense[0] + i * dense_stride (its output x goes to RDB i + 1's buffer); 0: the two buffers alternate
    int keep_all;               // 1: every plane is stored whole (the backward reads them), not only the rows a neighbour reads
    int out_pixel;              // 1: the trunk's fp32 output goes to `xr` in pixel order (NHWC) behind the last RDB
    // BACKWARD of the dense blocks (ptrunk3_kernel<., 1>; srbh_rrdbnet_trunk_train_backward_persistent): the saved forward planes are the LeakyReLU masks
    const char* mask;           // dense buffer of the forward RDB whose gradient runs first (the LAST forward RDB)
    long mask_stride;           // bytes from one running RDB's forward buffer to the next one's (negative: the RDBs run in reverse)
};

constexpr unsigned SPIN_LIMIT = 4u << 20;
using G = TileGeo<0>;
// LDS map (160 KiB).  The cout-32 layers of an RDB (conv1-4, "phase A") are bound by the L2-miss bandwidth of their input
// staging, not by the matrix cores, so they keep plane 0 of the dense buffer RESIDENT for the whole RDB (it is read by
// every conv) and stage only planes 1..k: 10 instead of 14 plane reads per RDB in those layers.
//   phase A (cout 32):  [R: plane 0][stage 0: input + 18 KiB weights][stage 1][bias 128 B][flag word]
//   phase B (cout 64):  [stage 0: input + 36 KiB weights][stage 1][bias 256 B] ... [flag word]
// Input areas are exact (the tail lanes of the last DMA instruction are masked off), weight areas too (no over-read).
constexpr int IN_EX = G::UNITS * 16;
constexpr int A_STAGE_B = IN_EX + 18 * 1024, B_STAGE_B = IN_EX + 36 * 1024;
constexpr int A_BASE = IN_EX;
constexpr int A_BIAS_OFF = A_BASE + 2 * A_STAGE_B, B_BIAS_OFF = 2 * B_STAGE_B;
constexpr int P_WORD_OFF = A_BIAS_OFF + 128;
constexpr int P_LDS_B = 163840;
static_assert(P_WORD_OFF + 4 <= P_LDS_B && B_BIAS_OFF + 256 <= P_WORD_OFF, "LDS map must fit 160 KiB");
__device__ __forceinline__ int stage_off(int cb, int idx) { return cb == 1 ? A_BASE + idx * A_STAGE_B : idx * B_STAGE_B; }

#include "srbh_ptrunk3_kernel.h"

}  // namespace

namespace srbh {

unsigned long long* g_ptrunk_prof = nullptr;   // set by tools/convbench only
static int g_trunk_timing = 0;                 // srbh_trunk_timing(): HIP events around the trunk launch(es), on their stream
static hipEvent_t g_trunk_ev[2];
static int g_trunk_ev_recorded = 0;       // a timed persistent launch has recorded both events since srbh_trunk_timing(1)
static const char* g_trunk_kernel = "none";    // srbh_trunk_kernel_name()

constexpr int MAX_BLOCKS = 64;   // layer-table capacity (RRDB blocks)
static size_t table_bytes() { return ((size_t)MAX_BLOCKS * 15 * sizeof(PLayer) + 255) & ~(size_t)255; }
static size_t prog_bytes(int B, int tpi) { return ((size_t)B * 2 * tpi * sizeof(int) + 255) & ~(size_t)255; }

size_t ptrunk_aux_bytes(int B, int tiles_per_img) { return table_bytes() + 2 * prog_bytes(B, tiles_per_img) + 256; }
// The launch's progress counters, XCC words and error word are cleared by a KERNEL, not by hipMemsetAsync: inside a replayed HIP
// graph the runtime does not keep memset nodes in stream order with the kernels of the PREVIOUS replay (measured: back-to-back
// replays of a captured training step timed out in the halo exchange -- the next replay's memsets had zeroed the running launch's
// progress counters; one replay at a time, or eager launches, never did).  A kernel node is ordered like any other launch.
__global__ void ptrunk_reset_kernel(int* __restrict__ prog, int* __restrict__ xcc, int* __restrict__ err, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { prog[i] = 0; xcc[i] = 0; }
    if (i == 0) *err = 0;
}

size_t ptrunk_err_offset(int B, int tiles_per_img) { return table_bytes() + prog_bytes(B, tiles_per_img); }

// The layer table depends only on the packed-weight pointers and num_block: it is uploaded ONCE per (device, contents)
// into a library-owned buffer and reused by every later forward (the per-call hipMemcpyAsync from pageable memory
// stalled the host on every forward and made forward_feature uncapturable in a HIP graph).  A handful of entries
// (several nets alive at once); a miss uploads synchronously.
struct TabEntry {
    int dev;
    std::vector<char> host;
    PLayer* dptr;
};
static std::mutex g_tab_mu;
static std::vector<TabEntry> g_tabs;

static int device_table(const std::vector<PLayer>& tab, PLayer** out) {
    int dev = 0;
    SRBH_HIP(hipGetDevice(&dev));
    const size_t nb = tab.size() * sizeof(PLayer);
    std::lock_guard<std::mutex> lock(g_tab_mu);
    for (size_t i = 0; i < g_tabs.size(); ++i)
        if (g_tabs[i].dev == dev && g_tabs[i].host.size() == nb && memcmp(g_tabs[i].host.data(), tab.data(), nb) == 0) {
            *out = g_tabs[i].dptr;
            return SRBH_OK;
        }
    if (g_tabs.size() >= 16) {   // evict the oldest (hipFree synchronises the device: nobody still reads it)
        SRBH_HIP(hipFree(g_tabs.front().dptr));
        g_tabs.erase(g_tabs.begin());
    }
    TabEntry e;
    e.dev = dev;
    e.host.assign((const char*)tab.data(), (const char*)tab.data() + nb);
    SRBH_HIP(hipMalloc(&e.dptr, nb));
    SRBH_HIP(hipMemcpy(e.dptr, tab.data(), nb, hipMemcpyHostToDevice));
    *out = e.dptr;
    g_tabs.push_back(std::move(e));
    return SRBH_OK;
}

// returns SRBH_OK and sets *used = 1 when the persistent path ran, *used = 0 when the shape is not the kernel's (full 8 x 64 tiles: W == 64,
// H a multiple of 8, at most one workgroup per CU per image column; the caller then issues the per-layer launch sequence, srbh_rrdbnet.hip)
// train_stride > 0 = the TRAINING forward (srbh_rrdbnet_trunk_train_forward_persistent): dense0 is RDB 0's buffer of a row of buffers train_stride
// bytes apart (dense1 ignored), every plane is stored whole, and the trunk's fp32 output goes to `xr` in pixel order.
// mask != nullptr = the BACKWARD of the dense blocks (srbh_rrdbnet_trunk_train_backward_persistent; needs train_stride > 0): `d` holds the gradient convs'
// bf16 packs in running (reverse) order, mask / mask_stride walk the saved forward buffers.
int ptrunk_run(const srbh_rrdbnet_desc* d, void* dense0, void* dense1, float* xr, float* xrr, int B, int H, int W,
               void* aux, hipStream_t stream, int* used, int* final_cur, long train_stride, const void* mask, long mask_stride) {
    *used = 0;
    if (mask && train_stride <= 0) return SRBH_OK;
    if (W != TILE_W || (H % TILE_H) != 0 || d->num_block <= 0 || d->num_block > MAX_BLOCKS) return SRBH_OK;
    int dev = 0;
    SRBH_HIP(hipGetDevice(&dev));
    int ncu = 0;
    SRBH_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const int tpi = H / TILE_H;
    if (tpi > ncu) return SRBH_OK;
    constexpr int LDS_B = P_LDS_B;
    SRBH_ONCE_PER_DEVICE({
        SRBH_HIP(hipFuncSetAttribute((const void*)ptrunk3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B));
        SRBH_HIP(hipFuncSetAttribute((const void*)ptrunk3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B));
        SRBH_HIP(hipFuncSetAttribute((const void*)ptrunk3_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B));
    });
    int per_cu = 0;
    SRBH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ptrunk3_kernel<0>, 256, LDS_B));
    if (per_cu < 1) return SRBH_OK;

    // layer table: the five convs of every RDB in running order (the kernel knows their shapes: four of cout 32, one of cout 64)
    const int nl = d->num_block * 15;
    std::vector<PLayer> tab(nl);
    int cur = 0, li = 0;
    for (int blk = 0; blk < d->num_block; ++blk)
        for (int r = 0; r < 3; ++r) {
            const srbh_conv_w* cw = d->rdb + (blk * 3 + r) * 5;
            for (int k = 0; k < 5; ++k) tab[li++] = PLayer{(const char*)cw[k].w, cw[k].bias};
            cur ^= 1;
        }
    *final_cur = cur;
    char* a = (char*)aux;
    PLayer* d_tab = nullptr;
    if (int rc = device_table(tab, &d_tab)) return rc;
    int* d_prog = (int*)(a + table_bytes());
    int* d_err = (int*)(a + ptrunk_err_offset(B, tpi));
    int* d_xcc = (int*)(a + ptrunk_err_offset(B, tpi) + 256);
    hipLaunchKernelGGL(ptrunk_reset_kernel, dim3((B * tpi + 255) / 256), dim3(256), 0, stream, d_prog, d_xcc, d_err, B * tpi);
    SRBH_HIP(hipGetLastError());
    const Act16Geo g = act16_geo(B, 6, H, W);
    // One workgroup per CU at most (co-residency).  SRBH_PT_IMAGES caps the images of one launch below that (developer / harness knob: a launch
    // that leaves CUs free lets kernels of ANOTHER stream run beside the trunk -- it holds every byte of LDS of the CUs it sits on); the
    // batch is then split evenly over the launches.
    int imgs_per_launch = ncu / tpi;
    if (const char* ie = getenv("SRBH_PT_IMAGES")) {
        const int cap = atoi(ie);
        if (cap > 0 && cap < imgs_per_launch) {
            const int nl_ = (B + cap - 1) / cap;
            imgs_per_launch = (B + nl_ - 1) / nl_;
        }
    }
    for (int b0 = 0; b0 < B; b0 += imgs_per_launch) {
        const int nb = (B - b0) < imgs_per_launch ? (B - b0) : imgs_per_launch;
        PParams pp;
        pp.dense_stride = train_stride;
        pp.keep_all = pp.out_pixel = train_stride > 0;
        pp.mask = mask ? (const char*)mask + (long)b0 * g.img_b : nullptr;
        pp.mask_stride = mask_stride;
        pp.dense[0] = (char*)dense0 + (long)b0 * g.img_b;
        pp.dense[1] = (train_stride > 0 ? (char*)dense0 + train_stride : (char*)dense1) + (long)b0 * g.img_b;
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
        pp.xcc = d_xcc + b0 * tpi;
        {
            const char* e = getenv("SRBH_PT_WT");
            pp.force_wt = (e && atoi(e) == 1) ? 1 : 0;
        }
        if (getenv("SRBH_PT_PROF") && !g_ptrunk_prof)
            SRBH_HIP(hipMalloc(&g_ptrunk_prof, (size_t)ncu * MAX_BLOCKS * 15 * 6 * 8));
        pp.prof = g_ptrunk_prof;
        if (g_trunk_timing && b0 == 0) SRBH_HIP(hipEventRecord(g_trunk_ev[0], stream));
        g_trunk_kernel = "ptrunk3_kernel";
        if (mask)
            hipLaunchKernelGGL((ptrunk3_kernel<0, 1>), dim3(pp.nblocks), dim3(256), LDS_B, stream, pp);
        else if (pp.prof)
            hipLaunchKernelGGL((ptrunk3_kernel<1>), dim3(pp.nblocks), dim3(256), LDS_B, stream, pp);
        else
            hipLaunchKernelGGL((ptrunk3_kernel<0>), dim3(pp.nblocks), dim3(256), LDS_B, stream, pp);
        SRBH_HIP(hipGetLastError());
    }
    if (g_trunk_timing) { SRBH_HIP(hipEventRecord(g_trunk_ev[1], stream)); g_trunk_ev_recorded = 1; }
    if (getenv("SRBH_PT_PROF") && g_ptrunk_prof) {   // developer aid: cycles vs wall clock of the real forward
        SRBH_HIP(hipStreamSynchronize(stream));
        const int nblk = (B < imgs_per_launch ? B : imgs_per_launch) * tpi;
        std::vector<unsigned long long> h((size_t)nblk * nl * 6);
        SRBH_HIP(hipMemcpy(h.data(), g_ptrunk_prof, h.size() * 8, hipMemcpyDeviceToHost));
        double cyc = 0;
        for (int b = 0; b < nblk; ++b) cyc += (double)(h[((size_t)b * nl + nl - 1) * 6 + 2] - h[(size_t)b * nl * 6]);
        fprintf(stderr, "[srbh] ptrunk: avg %.0f shader cycles per workgroup (first layer start -> last epilogue)\n", cyc / nblk);
        double loop[5] = {0}, epi[5] = {0}, pub[5] = {0}, wait[5] = {0}, tot5[5] = {0}, dma[5] = {0}, bar[5] = {0};
        for (int b = 0; b < nblk; ++b)
            for (int L = 1; L + 1 < nl; ++L) {
                const unsigned long long* q = &h[((size_t)b * nl + L) * 6];
                const unsigned long long* qn = &h[((size_t)b * nl + L + 1) * 6];
                const int k = L % 5;
                loop[k] += (double)(q[1] - q[0]); epi[k] += (double)(q[2] - q[1]); pub[k] += (double)(q[3] >> 32);
                wait[k] += (double)(q[3] & 0xffffffffu); tot5[k] += (double)(qn[0] - q[0]);
                bar[k] += (double)q[4]; dma[k] += (double)q[5];
            }
        const double cnt = (double)nblk * (nl - 2) / 5.0;
        for (int k = 0; k < 5; ++k)
            fprintf(stderr, "[srbh]   conv%d: loop %.0f (of which: waiting for the own LDS-DMA %.0f, at the step barriers %.0f; prologue %.0f) | "
                    "epilogue %.0f | publish/seam %.0f | start-to-start %.0f\n",
                    k + 1, loop[k] / cnt, dma[k] / cnt, bar[k] / cnt, wait[k] / cnt, epi[k] / cnt, pub[k] / cnt, tot5[k] / cnt);
        // conv5's epilogue / seam split by RDB kind: every third RDB also closes an RRDB (second residual from memory)
        double e5[2] = {0, 0}, s5[2] = {0, 0};
        long n5[2] = {0, 0};
        for (int b = 0; b < nblk; ++b)
            for (int L = 4; L + 1 < nl; L += 5) {
                const unsigned long long* q = &h[((size_t)b * nl + L) * 6];
                const int kind = ((L / 5) % 3) == 2;
                e5[kind] += (double)(q[2] - q[1]); s5[kind] += (double)(q[3] >> 32); n5[kind]++;
            }
        fprintf(stderr, "[srbh]   conv5 epilogue: %.0f plain / %.0f RRDB-closing; seam %.0f / %.0f\n", e5[0] / (n5[0] ? n5[0] : 1),
                e5[1] / (n5[1] ? n5[1] : 1), s5[0] / (n5[0] ? n5[0] : 1), s5[1] / (n5[1] ? n5[1] : 1));
    }
    *used = 1;
    return SRBH_OK;
}

}  // namespace srbh

// ---- measurement hook (bench.py): duration of the persistent trunk kernel itself, taken with HIP events recorded on the
// stream it is launched on.  Off by default; not thread safe (one forward at a time while it is on).
extern "C" int srbh_trunk_timing(int on) {
    using namespace srbh;
    if (on && !g_trunk_timing) {
        SRBH_HIP(hipEventCreate(&g_trunk_ev[0]));
        SRBH_HIP(hipEventCreate(&g_trunk_ev[1]));
    } else if (!on && g_trunk_timing) {
        SRBH_HIP(hipEventDestroy(g_trunk_ev[0]));
        SRBH_HIP(hipEventDestroy(g_trunk_ev[1]));
    }
    g_trunk_timing = on ? 1 : 0;
    g_trunk_ev_recorded = 0;
    return SRBH_OK;
}

extern "C" const char* srbh_trunk_kernel_name(void) { return srbh::g_trunk_kernel; }

extern "C" int srbh_trunk_last_ms(float* ms) {
    using namespace srbh;
    SRBH_REQUIRE(ms && g_trunk_timing, "srbh_trunk_last_ms: timing is off (srbh_trunk_timing(1) first)");
    // (never hand unrecorded events to hipEventElapsedTime: it leaves a sticky 'invalid resource handle' behind that the next,
    // unrelated HIP call of the process reports)
    SRBH_REQUIRE(g_trunk_ev_recorded, "srbh_trunk_last_ms: no persistent trunk launch was timed (per-layer path, SRBH_PERSISTENT=0?)");
    SRBH_HIP(hipEventSynchronize(g_trunk_ev[1]));
    SRBH_HIP(hipEventElapsedTime(ms, g_trunk_ev[0], g_trunk_ev[1]));
    return SRBH_OK;
}
