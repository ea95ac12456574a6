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


#ifndef PT_LOAD_SCOPE
#define PT_LOAD_SCOPE " sc1"   // cache-coherence bits of the activation LDS-DMA loads
#endif
#ifndef PT_DEFAULT_VARIANT
#define PT_DEFAULT_VARIANT 3
#endif
#ifndef PT_DMA_GROUPS
#define PT_DMA_GROUPS 3
#endif
#ifndef PT_DEFER_FLAGS
#define PT_DEFER_FLAGS 1
#endif

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
    int* xcc;                   // [nblocks] XCC_ID + 1 of every workgroup (placement handshake)
    int force_wt;               // 1: always use write-through stores (debugging aid, env SRBH_PT_WT=1)
    int frag_res;               // 1: fp32 residual streams in fragment order inside the launch (W == TILE_W)
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
constexpr int JPP = (G::NJ + 5) / 6;
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
__device__ __forceinline__ int bias_off(int cb) { return cb == 1 ? A_BIAS_OFF : B_BIAS_OFF; }

// PLayer.flags bit 3 is set by the host when the layer's FIRST input chunk is produced by the previous layer
// (conv1 of an RDB reads the x written by the previous conv5): it can be neither prefetched nor published lazily.
__device__ __forceinline__ int first_new_chunk(const PLayer& l) { return (l.flags & 8) ? 0 : l.nchunk - 1; }

// PROF (developer builds of the timeline only): s_memtime stamps per layer; compiled out of the production kernel -- the
// stamps and their pointer cost ~16 SGPRs in a kernel that already spills scalars
template <bool PROF, bool REGRES>
__global__ __launch_bounds__(256, 1) void ptrunk_kernel(const PParams pp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int t = xcd_remap(blockIdx.x, pp.nblocks);
    const int img = t / pp.tiles_per_img;
    const int ty = t - img * pp.tiles_per_img;
    const int Y0 = ty * TILE_H;
    const int up = ty > 0 ? t - 1 : -1, dn = ty + 1 < pp.tiles_per_img ? t + 1 : -1;

    // ---- geometry that is identical for every layer
    int goff[G::NJ];
#pragma unroll
    for (int j = 0; j < G::NJ; ++j) {
        const int u0 = j * 256 + tid;
        const int u = u0 < G::UNITS ? u0 : 0;
        const int trow = u / (G::COLS * 4);
        const int rem = u - trow * (G::COLS * 4);
        const int pc = rem >> 2, ps = rem & 3;
        goff[j] = trow * pp.row_b + pc * PIX_B + ((ps ^ ((pc >> 2) & 3)) << 4);
    }
    const bool tail_ok = (G::NJ - 1) * 256 + tid < G::UNITS;   // lanes of the last DMA instruction that carry tile data
    int aoff[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        const int pc = wc * 32 + l31 + dx;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
            aoff[dx][ks] = wr * 4 * G::ROW_B + pc * PIX_B + (((ks * 2 + hi) ^ ((pc >> 2) & 3)) << 4);
    }
    const int woff = lane * 16;
    const long tile_off = (long)img * pp.img_b + (long)Y0 * pp.row_b;

    // ---- one 16-B-per-lane LDS-DMA under an explicit EXEC mask (all-ones, none, or the tail lanes): predication without a
    // branch.  SC1 = bypass this CU's L1 (activations written by other CUs inside this launch); weights may hit L1.
    const unsigned long long tail_mask = __builtin_amdgcn_ballot_w64(tail_ok);
    auto lds_addr = [](const char* p) { return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p; };
    auto uni64 = [](unsigned long long m) {   // make uniformity visible to the compiler ("s" operands must be SGPRs)
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)m), hi = __builtin_amdgcn_readfirstlane((unsigned)(m >> 32));
        return ((unsigned long long)hi << 32) | lo;
    };
    auto dma16 = [&](auto sc1_tag, const char* gaddr, const unsigned lds_off_v, const unsigned long long mask) {
        unsigned long long sv;
        const unsigned lds_off = __builtin_amdgcn_readfirstlane(lds_off_v);
        if constexpr (decltype(sc1_tag)::value)
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %3, off" PT_LOAD_SCOPE "\n\ts_mov_b64 exec, %0"
                         : "=&s"(sv) : "s"(mask), "s"(lds_off), "v"(gaddr) : "memory", "m0");
        else
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %3, off\n\ts_mov_b64 exec, %0"
                         : "=&s"(sv) : "s"(mask), "s"(lds_off), "v"(gaddr) : "memory", "m0");
    };

    // ---- cold staging of one step (used where nothing could be prefetched): input plane (if any) to `din`, the
    // 18 * ncb weight fragments to `dw`
    auto stage_cold = [&](const char* src, char* din, const char* wsrc, char* dw, const int ncb) {
        const unsigned din_l = lds_addr(din), dw_l = lds_addr(dw);
        if (src) {
#pragma unroll
            for (int j = 0; j < G::NJ; ++j)
                dma16(std::true_type{}, src + goff[j], din_l + (j * 256 + wave * 64) * 16, j < G::NJ - 1 ? ~0ull : tail_mask);
        }
        const char* ws = wsrc + lane * 16;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int f = wave + 4 * k;
            dma16(std::false_type{}, ws + f * 1024, dw_l + f * 1024, uni64(f < 18 * ncb ? ~0ull : 0ull));
        }
    };
    auto chunk_src = [&](const PLayer& l, int c) { return pp.dense[l.in_sel] + tile_off + (long)c * pp.plane_b; };
    auto chunk_w = [&](const PLayer& l, int c) { return l.w + (long)c * (18 * 1024 * l.cb); };

    // ---- neighbour progress: thread 0 keeps the last values it loaded; a blocking (bounded) poll only if they are stale
    int f_up = up < 0 ? 0x7fffffff : 0, f_dn = dn < 0 ? 0x7fffffff : 0;
    int gs = 0;                  // global step counter: stage buffer = gs & 1
    bool aborted = false;
    auto ensure_flags = [&](int need) {
        auto* word = (__attribute__((address_space(3))) int*)(smem + P_WORD_OFF);   // (a generic pointer would become flat_*)
        if (tid == 0) {
            int bad = 0;
            unsigned spins = 0;
            while (f_up < need || f_dn < need) {
                if (up >= 0) f_up = __hip_atomic_load(pp.prog + up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (dn >= 0) f_dn = __hip_atomic_load(pp.prog + dn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (f_up >= need && f_dn >= need) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > SPIN_LIMIT || __hip_atomic_load(pp.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    bad = 1;
                    break;
                }
            }
            if (bad) __hip_atomic_store(pp.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *word = bad;
        }
        __syncthreads();
        const int bad = *word;
        __syncthreads();
        if (bad) aborted = true;
    };

    // ---- where do my halo partners run?  Workgroups are dealt round-robin to the 8 XCDs and xcd_remap() puts the tiles
    // of one image on one XCD, so normally both neighbours share this workgroup's L2 and the exchange never has to leave
    // it: plain stores are complete (vmcnt) once they are in L2, and the neighbours' L1-bypassing reads find them there.
    // That placement is a dispatcher habit, not a contract, so it is verified: every workgroup posts its XCC_ID and
    // compares it with its neighbours'; any mismatch (or a neighbour that never answers) selects write-through (sc1)
    // stores, which are placement independent.
    bool wt = true;
    {
        int my_xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(my_xcc));
        auto* word = (__attribute__((address_space(3))) int*)(smem);
        if (tid == 0) {
            __hip_atomic_store(pp.xcc + t, my_xcc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int diff = pp.force_wt;
            for (int s = 0; s < 2 && !diff; ++s) {
                const int nb = s ? dn : up;
                if (nb < 0) continue;
                int v = 0;
                for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
                    v = __hip_atomic_load(pp.xcc + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                if (v != my_xcc + 1) diff = 1;
            }
            *word = diff;
        }
        __syncthreads();
        wt = __builtin_amdgcn_readfirstlane(*word) != 0;
        __syncthreads();
    }
    auto publish = [&](int v) {
        if (tid == 0) {
            if (wt)
                __hip_atomic_store(pp.prog + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                __hip_atomic_store(pp.prog + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // reaches L2, no further
        }
    };

    bool pending_pub = false;
    int pub_val = 0;

    // The RDB-level residual stream (the fp32 input x of the current RDB) of this wave's 4 rows x 32 pixels x 64 channels
    // lives in 128 REGISTERS per lane, in exactly the accumulator layout of conv5 ([mb][row][channel group]): with one
    // wave per SIMD half of the 512-entry register file is otherwise idle, and conv5's epilogue stops being an fp32
    // read-modify-write of 128 KiB per workgroup through HBM.  Only the RRDB-level stream (every third RDB) stays in memory.
    floatx4 xres[2][4][4];
    if (REGRES) {   // the launch's input x, written by conv_first in pixel order
        const int X = wc * 32 + l31;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int Y = Y0 + wr * 4 + i;
            const bool ok = (Y < pp.H) && (X < pp.W);
            const float* q = pp.xrr + (((long)img * pp.H + (ok ? Y : 0)) * pp.W + (ok ? X : 0)) * 64 + hi * 4;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g) xres[mb][i][g] = *(const floatx4*)(q + mb * 32 + g * 8);
        }
    }

    // ---- one layer: CB = cout/32
    auto run_layer = [&](auto cb_tag, const int L, const PLayer& lay) {
        constexpr int CB = decltype(cb_tag)::value;
        constexpr int NREAD = G::NP + 3 * CB, NMFMA = 12 * CB;
        // ---- layer prologue (no accumulator is live here)
        unsigned long long p0 = 0;
        if (PROF) p0 = __builtin_amdgcn_s_memtime();
        // LDS-DMA completion is NOT reliably waited for by hipcc before a barrier (seen: no vmcnt at all in this loop
        // shape) -> always drain explicitly.  vmcnt(0) also covers this workgroup's write-through stores of layer L-1.
        // The layer's bias goes through LDS: its global-load latency hides under the drain below instead of
        // opening the epilogue, and no registers are held across the chunk loop.
        float bias_v = 0.f;
        if (tid < lay.cb * 32) bias_v = lay.bias[tid];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();   // step (L,0) landed on every wave, and every wave is past the previous epilogue (bias slot free)
        if (tid < lay.cb * 32) ((float*)(smem + bias_off(lay.cb)))[tid] = bias_v;   // read after >= 1 more barrier
        if (pending_pub) {
            publish(pub_val);
            pending_pub = false;
        }
        // Layer L's only NEW input plane is its last chunk (except at an RDB seam, flag 8, whose wait happened in the
        // previous epilogue): the neighbour-flag wait (one L2 round trip, ~2.5 k cycles when taken here) is deferred
        // behind step 0, which reads planes verified layers ago and stages another such plane.
        const bool defer_flags = PT_DEFER_FLAGS && !(lay.flags & 8) && lay.nchunk >= 3;
        if (L > 0 && !defer_flags) ensure_flags(L);   // every input plane of layer L is complete on both neighbours
        if (aborted) return;
        const bool has_next_prefetch = (L + 1 < pp.nlayers) && !(pp.layers[L + 1 < pp.nlayers ? L + 1 : L].flags & 8);
        const PLayer nlay = pp.layers[L + 1 < pp.nlayers ? L + 1 : L];
        unsigned long long p1 = 0;
        if (PROF) p1 = __builtin_amdgcn_s_memtime();
        floatx16 acc[CB][4];
#pragma unroll
        for (int mb = 0; mb < CB; ++mb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][i][r] = 0.f;
        half8 P[2][G::NP];
        half8 A[2][3][CB];
        auto load_group = [&](const char* sbi, const char* sbw, int g, int set) {
            const int ks = g / 3, dx = g - ks * 3;
#pragma unroll
            for (int r = 0; r < G::NP; ++r) P[set][r] = *(const half8*)(sbi + aoff[dx][ks] + r * G::ROW_B);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int mb = 0; mb < CB; ++mb)
                    A[set][dy][mb] = *(const half8*)(sbw + woff + ((((dy * 3 + dx) * 2 + ks) * CB + mb) << 10));
        };
        // One code path for every chunk (duplicating it per staging variant made the compiler shuffle all accumulators
        // through VGPRs at the join): the LDS-DMA slices of the next step sit behind tiny wave-uniform branches at the
        // head of each MFMA group.  next_cb = cout/32 of the layer that owns the staged step, 0 = nothing to stage.
        auto compute = [&](const char* sbi, const char* sbw, const int next_cb, const char* nsrc, const char* nw, char* dst) {
            const char* ws = nw + lane * 16;
            load_group(sbi, sbw, 0, 0);
            // LDS-DMA issue order: the instructions that carry the two halo rows (tile rows 0 and 9: j = 0, 1, 9, 10) go last
            constexpr int JORD[12] = {2, 3, 4, 5, 6, 7, 8, 0, 1, 9, 10, 11};
            static_assert(G::NJ == 11 && JPP == 2, "DMA issue order is written for 11 instructions, 2 per group");
            // Every staging decision is an EXEC mask, not a branch (dma16): the whole step is one basic block, so the
            // DMA set-up (address, M0) is scheduled into the shadow of the MFMAs instead of idling the matrix core between
            // groups (a one-wave-per-SIMD kernel has nobody else to fill that gap).
            const unsigned long long m_all = ~0ull;
            const unsigned long long m_in = uni64((next_cb && nsrc) ? m_all : 0ull);      // nullptr: the step reads the resident plane
            const unsigned long long m_tail = uni64((next_cb && nsrc) ? tail_mask : 0ull);
            const unsigned long long m_w = uni64(next_cb ? m_all : 0ull);
            const unsigned long long m_w4 = uni64((next_cb == 2 || (next_cb == 1 && wave < 2)) ? m_all : 0ull);   // fragments 16..19: 18 or 36 in total
            const unsigned long long m_w2 = uni64(next_cb == 2 ? m_all : 0ull);
            // The 20 DMA instructions of the next step are issued in the first MFMA groups: a step cannot end before its
            // LAST DMA has landed (issue time + ~1.5 us of L2-miss latency), so spreading them over all six groups made
            // every step latency-bound.
            const unsigned dst_w = __builtin_amdgcn_readfirstlane(lds_addr(dst) + wave * 1024);       // inputs: + j * 4096
            const unsigned wdst_w = dst_w + IN_EX;                                                      // weights: + k * 4096
            const unsigned long long ibase = uni64((unsigned long long)nsrc);
            const unsigned long long wbase = uni64((unsigned long long)(nw + wave * 1024));
            const unsigned wl = lane * 16;
            // one asm statement per DMA (the scheduler interleaves them with the MFMAs: blocks of 5 back-to-back DMAs stall
            // the wave on the VMEM issue queue and measured 5 % slower), scalar base + 32-bit VGPR offset addressing
            // (EXEC is all-ones everywhere in this kernel's compute region: restored with the constant, no save; the EXEC
            // write doubles as the wait state the M0 write needs before an LDS-DMA.  The cout-32 steps carry ~4 non-MFMA
            // instructions per MFMA, about what one wave per SIMD can hide: every instruction less in here is time.)
            auto dma_s = [&](auto sc1_tag, const unsigned long long base, const unsigned voff, const unsigned lds_off,
                             const unsigned long long mask) {
                if constexpr (decltype(sc1_tag)::value)
                    asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 exec, %0\n\tglobal_load_lds_dwordx4 %2, %3 sc1\n\ts_mov_b64 exec, -1"
                                 :: "s"(mask), "s"(lds_off), "v"(voff), "s"(base) : "memory", "m0");
                else
                    asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 exec, %0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, -1"
                                 :: "s"(mask), "s"(lds_off), "v"(voff), "s"(base) : "memory", "m0");
            };
            auto issue = [&](const int i) {
                if (i < 11) {
                    const int j = JORD[i];
                    dma_s(std::true_type{}, ibase, goff[j], dst_w + j * 4096, j < G::NJ - 1 ? m_in : m_tail);
                } else if (i < 16) {
                    const int k = i - 11;
                    dma_s(std::false_type{}, wbase + k * 4096, wl, wdst_w + k * 4096, k < 4 ? m_w : m_w4);
                } else if (CB == 2) {        // fragments 20..35 only exist for a cout-64 step; a cout-32 layer stages one
                    const int k = i - 16;    // only in its very last step (conv4 -> conv5), see below the group loop
                    dma_s(std::false_type{}, wbase + 20 * 1024 + k * 4096, wl, wdst_w + 20 * 1024 + k * 4096, m_w2);
                }
            };
#pragma unroll
            for (int g = 0; g < 6; ++g) {
#pragma unroll
                for (int i = 0; i < 20; ++i)
                    if (i * PT_DMA_GROUPS / 20 == g) issue(i);
                if (g + 1 < 6) load_group(sbi, sbw, g + 1, (g + 1) & 1);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int mb = 0; mb < CB; ++mb)
                            acc[mb][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g & 1][dy][mb], P[g & 1][i + dy], acc[mb][i], 0, 0, 0);
                if (g == 0) __builtin_amdgcn_sched_group_barrier(0x100, NREAD, 0);
                if (g + 1 < 6) {
#pragma unroll
                    for (int k = 0; k < NREAD; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, NMFMA - NREAD, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, NMFMA, 0);
                }
            }
            if (CB == 1 && next_cb == 2) {   // (behind the MFMAs: one step per RDB)
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    dma_s(std::false_type{}, wbase + 20 * 1024 + k * 4096, wl, wdst_w + 20 * 1024 + k * 4096, ~0ull);
            }
        };

        const int n = lay.nchunk;
        unsigned long long ts0 = p1, ts1 = 0, ts2 = 0, tw = p1 - p0, tb = 0;
        // The chunk loop below must stay as plain as the per-layer kernel's (barrier + one compute body): with any
        // extra control flow inside it hipcc parks the loop-carried accumulators in VGPRs and copies all of them back
        // into AGPRs at the top of every chunk.  So everything protocol-related happened in the layer prologue.
        auto step = [&](const int c) {
            const char* st = smem + stage_off(CB, gs & 1);
            const char* sbi = (CB == 1 && c == 0) ? smem : st;   // phase A: plane 0 is resident
            int next_cb = 0;
            const char* nsrc = nullptr;
            const char* nw = nullptr;
            if (c + 1 < n) {
                next_cb = CB;
                nsrc = chunk_src(lay, c + 1);
                nw = chunk_w(lay, c + 1);
            } else if (has_next_prefetch) {
                next_cb = nlay.cb;
                nsrc = nlay.cb == 1 ? nullptr : chunk_src(nlay, 0);   // a cout-32 layer's step 0 reads the resident plane
                nw = chunk_w(nlay, 0);
            }
            compute(sbi, st + IN_EX, next_cb, nsrc, nw, smem + stage_off(next_cb ? next_cb : CB, (gs + 1) & 1));
            ++gs;
        };
        step(0);
        if (defer_flags) ensure_flags(L);   // (once per layer, outside the chunk loop: see the note on control flow above)
        if (aborted) return;
        unsigned long long t_dma = 0;   // PROF: cycles wave 0 waits for its own LDS-DMA / at the barrier, summed over the steps
        for (int c = 1; c < n; ++c) {
            unsigned long long w0 = 0, w1 = 0;
            if (PROF) w0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA of step gs has landed ...
            if (PROF) w1 = __builtin_amdgcn_s_memtime();
            __syncthreads();                                   // ... and everybody else's; all waves are past step gs-1
            if (PROF) {
                t_dma += w1 - w0;
                tb += __builtin_amdgcn_s_memtime() - w1;
            }
            step(c);
        }
        if (PROF) ts1 = __builtin_amdgcn_s_memtime();

        // ---- epilogue, straight from the MFMA D layout (no LDS round trip): lane (l31, hi) holds, for every row i and
        // channel group g, the 4 consecutive channels 8g + 4hi + (0..3) of pixel l31.  fp32 residual traffic is already
        // 16 B per lane; the fp16 output is widened from 8 to 16 B per lane with v_permlane32_swap (the two half-waves
        // exchange one 4-channel packet so that each ends up with 8 consecutive channels).
        // The trunk has exactly two layer shapes and ptrunk_run() builds nothing else: cout 32 + leaky ReLU (conv1-4) and
        // cout 64 + residual(s) (conv5).  Tying the epilogue variant to CB at compile time keeps selects out of a
        // VALU-bound epilogue (flags 1 and 2 of the table are implied; flag 4 stays a run-time property).
        constexpr bool lrelu = (CB == 1), r1 = (CB == 2);
        const bool r2 = r1 && (lay.flags & 4);
        char* obase = pp.dense[lay.out_sel] + (long)img * pp.img_b + (long)lay.out_chunk0 * pp.plane_b;
        floatx4 bias4[CB][4];
#pragma unroll
        for (int mb = 0; mb < CB; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                bias4[mb][g] = *(const floatx4*)((const float*)(smem + bias_off(CB)) + mb * 32 + g * 8 + hi * 4);
        const int X = wc * 32 + l31;
        // fp32 residual streams.  They are private to this workgroup -- each lane re-reads exactly the values it wrote one
        // RDB earlier -- so inside the launch they live in "fragment order": within the wave's 32-pixel segment of a
        // row, instruction (mb, g) owns 1 KiB and lane l its 16 B at l*16 (whole cache lines per instruction instead of
        // 32 B pieces of 32 different lines).  Only the first read of each stream (written by conv_first) is in pixel
        // order (flags 16 / 32); widths other than TILE_W keep pixel order throughout.  Float offsets:
        struct ResForm { int lane, sm, sg; };
        const ResForm pixel_form{X * 64 + hi * 4, 32, 8}, frag_form{wc * 2048 + lane * 4, 1024, 256};
        const ResForm s1 = (!pp.frag_res || (lay.flags & 16)) ? pixel_form : frag_form;
        const ResForm s2 = (!pp.frag_res || (lay.flags & 32)) ? pixel_form : frag_form;
        const ResForm sd = pp.frag_res ? frag_form : pixel_form;
        const float* res1_src = (lay.flags & 64) ? pp.xrr : pp.xr;
        // rows are processed NR at a time: all residual loads of the group are issued before any of them is consumed
        // (per-row processing left only 8-16 loads in flight per wave and made the fp32 residual RMW latency-bound)
        auto process_rows = [&](auto nr_tag, auto i0_tag) {
            constexpr int NR = decltype(nr_tag)::value, i0 = decltype(i0_tag)::value;
            floatx4 a1[NR][CB][4], a2[NR][CB][4];
            bool valid[NR];
            long rowb[NR];   // float offset of image row Y in the RES32 streams
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                const int Y = Y0 + wr * 4 + i0 + k;
                valid[k] = (Y < pp.H) && (X < pp.W);
                rowb[k] = ((long)img * pp.H + Y) * pp.W * 64;
                if (r1 && valid[k]) {
                    if (!REGRES) {
                        const float* q1 = res1_src + rowb[k] + s1.lane;
#pragma unroll
                        for (int mb = 0; mb < CB; ++mb)
#pragma unroll
                            for (int g = 0; g < 4; ++g) a1[k][mb][g] = *(const floatx4*)(q1 + mb * s1.sm + g * s1.sg);
                    }
                    if (r2) {
                        const float* q2 = pp.xrr + rowb[k] + s2.lane;
#pragma unroll
                        for (int mb = 0; mb < CB; ++mb)
#pragma unroll
                            for (int g = 0; g < 4; ++g) a2[k][mb][g] = *(const floatx4*)(q2 + mb * s2.sm + g * s2.sg);
                    }
                }
            }
            floatx4 vv[NR][CB][4];
#pragma unroll
            for (int k = 0; k < NR; ++k) {
                const int i = i0 + k;
                const int Y = Y0 + wr * 4 + i;
#pragma unroll
                for (int mb = 0; mb < CB; ++mb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        floatx4 t;
#pragma unroll
                        for (int q = 0; q < 4; ++q) t[q] = acc[mb][i][g * 4 + q];
                        t += bias4[mb][g];
                        if (r1) {
                            if (REGRES)
                                t = t * 0.2f + xres[mb][i][g];
                            else
                                t = t * 0.2f + a1[k][mb][g];
                            if (r2) t = t * 0.2f + a2[k][mb][g];
                            if (REGRES) xres[mb][i][g] = t;
                        }
                        vv[k][mb][g] = t;
                    }
#pragma unroll
                for (int mb = 0; mb < CB; ++mb) {
                    unsigned hp[4][2];   // packed fp16 pairs of the 4 channel groups
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        floatx4 w = vv[k][mb][g];
                        // leaky ReLU as max(x, 0.2x) with a bare v_max_f32: fmaxf() costs two extra canonicalising
                        // v_max per value and this epilogue is VALU-bound
                        if (lrelu) {
                            const floatx4 ws = w * 0.2f;
#pragma unroll
                            for (int q = 0; q < 4; ++q) asm("v_max_f32 %0, %1, %2" : "=v"(w[q]) : "v"(w[q]), "v"(ws[q]));
                        }
                        half4 h4;
#pragma unroll
                        for (int q = 0; q < 4; ++q) h4[q] = (_Float16)w[q];
                        const uint2 u = __builtin_bit_cast(uint2, h4);
                        hp[g][0] = u.x;
                        hp[g][1] = u.y;
                    }
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        // lower half-wave ends with channels 16m + 0..7, upper half-wave with 16m + 8..15
                        auto s0 = __builtin_amdgcn_permlane32_swap(hp[2 * m][0], hp[2 * m + 1][0], false, false);
                        auto s1 = __builtin_amdgcn_permlane32_swap(hp[2 * m][1], hp[2 * m + 1][1], false, false);
                        typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
                        const uintx4 raw = {s0[0], s1[0], s0[1], s1[1]};
                        if (valid[k]) {
                            char* o = obase + (long)mb * pp.plane_b + (long)(Y + 1) * pp.row_b + (X + 1) * PIX_B + m * 32 + hi * 16;
                            if (wt)
                                asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(o), "v"(raw) : "memory");
                            else
                                asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(o), "v"(raw) : "memory");
                        }
                    }
                }
            }
            // the private fp32 residual streams go out LAST: at an RDB seam only the write-through fp16 stores above have
            // to be complete before the progress counter moves (see `seam` below), these may still be in flight
            if (r1 && (r2 || !REGRES)) {   // REGRES: the RDB-level stream stays in registers, nothing to store
#pragma unroll
                for (int k = 0; k < NR; ++k) {
                    if (!valid[k]) continue;
                    // an RRDB-closing layer leaves xr == xrr: only xrr is written, the next RDB reads its res1 from there
                    // scalar base + 32-bit lane offset, formed at the point of use: left to the compiler, the 32 64-bit
                    // store addresses are computed early, spilled, and every store then waits for its address reload
                    const float* q = (r2 ? pp.xrr : pp.xr) + rowb[k];
                    const unsigned vo = (unsigned)sd.lane * 4u;
#pragma unroll
                    for (int mb = 0; mb < CB; ++mb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const unsigned long long sb = uni64((unsigned long long)(q + mb * sd.sm + g * sd.sg));
                            // Hazards the compiler cannot see through inline asm: the scalar base is typically a fresh
                            // v_readlane (SGPR spill reload) and a VALU-written SGPR needs 5 wait states before VMEM reads
                            // it; a >8-byte store needs wait states before a VALU write of its data registers.
                            asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(vo), "v"(vv[k][mb][g]), "s"(sb) : "memory");
                        }
                }
            }
        };
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        using I3 = std::integral_constant<int, 3>;
        using I4 = std::integral_constant<int, 4>;
        if (r2) {
            process_rows(I2{}, I0{});
            process_rows(I2{}, I2{});
        } else if (REGRES && r1) {   // no residual loads to batch: row by row keeps the register pressure down
            process_rows(I1{}, I0{});
            process_rows(I1{}, I1{});
            process_rows(I1{}, I2{});
            process_rows(I1{}, I3{});
        } else {
            process_rows(I4{}, I0{});
        }
        if (PROF) ts2 = __builtin_amdgcn_s_memtime();
        // ---- publication of "layer L complete"
        const bool seam = (L + 1 == pp.nlayers) || (pp.layers[L + 1 < pp.nlayers ? L + 1 : L].flags & 8);
        if (seam) {
            // the next layer's first chunk is THIS layer's output on the neighbours: publish now, then wait for them
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            publish(L + 1);
            if (L + 1 < pp.nlayers) {
                const PLayer& nl = pp.layers[L + 1];
                ensure_flags(L + 1);   // its first chunk is this layer's output on the neighbours: wait before staging it
                if (aborted) return;
                // ring restart: the seam layer's step 0 reads the resident plane + the weights in stage 0; with 14 phase-A
                // steps per RDB the last one then sits in stage 1, clear of the phase-B stage 0 it prefetches into
                gs = 0;
                if (nl.cb == 1)
                    stage_cold(chunk_src(nl, 0), smem, chunk_w(nl, 0), smem + stage_off(1, 0) + IN_EX, 1);
                else
                    stage_cold(chunk_src(nl, 0), smem + stage_off(2, 0), chunk_w(nl, 0), smem + stage_off(2, 0) + IN_EX, 2);
            }
        } else {
            pending_pub = true;   // published behind the next top-of-step barrier (whose vmcnt(0) covers these stores)
            pub_val = L + 1;
        }
        if (PROF && tid == 0) {
            unsigned long long* q = pp.prof + ((long)blockIdx.x * pp.nlayers + L) * 6;
            q[0] = ts0; q[1] = ts1; q[2] = ts2; q[3] = tw | ((unsigned long long)(__builtin_amdgcn_s_memtime() - ts2) << 32);
            q[4] = tb;
            q[5] = t_dma;
        }
    };

    // ---- prologue: layer 0's inputs were written by the previous kernel (conv_first): no flag needed
    {
        const PLayer& l0 = pp.layers[0];
        if (l0.cb == 1)
            stage_cold(chunk_src(l0, 0), smem, chunk_w(l0, 0), smem + stage_off(1, 0) + IN_EX, 1);
        else
            stage_cold(chunk_src(l0, 0), smem + stage_off(2, 0), chunk_w(l0, 0), smem + stage_off(2, 0) + IN_EX, 2);
    }
    for (int L = 0; L < pp.nlayers && !aborted; ++L) {
        const PLayer lay = pp.layers[L];
        if (lay.cb == 1)
            run_layer(std::integral_constant<int, 1>{}, L, lay);
        else
            run_layer(std::integral_constant<int, 2>{}, L, lay);
    }
}


#include "srbh_ptrunk3_kernel.h"

// =====================================================================================================================
// Variant 2: TWO workgroups per CU.
//
// With one 4-wave workgroup per CU every SIMD runs a single wave: nothing fills the matrix core while that wave sits in
// an epilogue, a barrier, a flag round trip or the first LDS reads of a step (~40 % of the time in variant 1).  Here a
// workgroup owns (image, FOUR rows), keeps at most 256 registers per wave and ~61 KiB of LDS, so two workgroups --
// normally in different phases of the layer sequence -- share a CU and fill each other's gaps.  To fit:
//   * the K pipeline advances in HALF chunks (16 input channels = one MFMA K step): a stage is a 6 x 66 pixel tile with
//     32 B per pixel (12.4 KiB, stored as two 16-B planes so that ds_read_b128 is lane-linear) + the 9 * cb weight
//     fragments of that k-step (<= 18 KiB);
//   * a wave owns 2 rows x 32 pixels (32 * cb accumulator registers), 4 pixel fragments + 3 * cb weight fragments feed
//     6 * cb MFMAs per (dx) group.
// Everything else (layer table, progress counters, in-L2 exchange with XCC handshake, fragment-order residual streams,
// exec-masked DMA issued at the head of a step) is the protocol of variant 1; results are bit-identical to it.
constexpr int T2_H = 4;
constexpr int T2_ROWS = T2_H + 2, T2_COLS = TILE_W + 2;
constexpr int T2_PIX = T2_ROWS * T2_COLS;                  // 396 pixels
constexpr int T2_UNITS = 2 * T2_PIX;                       // 16-B units, [k-half][row][col]
constexpr int T2_IN_B = T2_UNITS * 16;                     // 12 672 B
constexpr int T2_NJ = (T2_UNITS + 255) / 256;              // 4 DMA instructions (the last one: 24 lanes)
constexpr int T2_STAGE_B = T2_IN_B + 18 * 1024;
constexpr int T2_BIAS_OFF = 2 * T2_STAGE_B;
constexpr int T2_WORD_OFF = T2_BIAS_OFF + 256;
constexpr int T2_LDS_B = T2_WORD_OFF + 64;
static_assert(2 * T2_LDS_B <= 163840, "two workgroups must fit one CU's LDS");

struct Step2 {
    const char* src;   // input plane, tile origin (nullptr: nothing to stage)
    const char* w;     // packed weights of the chunk
    int ks, cb;
};

__global__ __launch_bounds__(256, 2) void ptrunk2_kernel(const PParams pp) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wr = wave >> 1, wc = wave & 1;
    const int t = xcd_remap(blockIdx.x, pp.nblocks);
    const int img = t / pp.tiles_per_img;
    const int ty = t - img * pp.tiles_per_img;
    const int Y0 = ty * T2_H;
    const int up = ty > 0 ? t - 1 : -1, dn = ty + 1 < pp.tiles_per_img ? t + 1 : -1;
    const int X = wc * 32 + l31;

    // ---- DMA geometry: unit u -> (k-half h, tile row, column); source = k-slot 2*ks + h of the 64-B pixel record
    int goff0[T2_NJ], goff1[T2_NJ];
    unsigned long long jmask[T2_NJ];
#pragma unroll
    for (int j = 0; j < T2_NJ; ++j) {
        const int u0 = j * 256 + tid;
        const int u = u0 < T2_UNITS ? u0 : 0;
        const int h = u / T2_PIX, pix = u - h * T2_PIX;
        const int trow = pix / T2_COLS, pc = pix - trow * T2_COLS;
        const int base = trow * pp.row_b + pc * PIX_B;   // (the XOR swizzle of variant 1 is an LDS-side trick only)
        goff0[j] = base + ((0 + h) << 4);
        goff1[j] = base + ((2 + h) << 4);
        jmask[j] = __builtin_amdgcn_ballot_w64(u0 < T2_UNITS);
    }
    int poff[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) poff[dx] = (hi * T2_PIX + (wr * 2) * T2_COLS + wc * 32 + l31 + dx) * 16;
    const int woff = T2_IN_B + lane * 16;
    const long tile_off = (long)img * pp.img_b + (long)Y0 * pp.row_b;

    auto uni64 = [](unsigned long long m) {
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)m), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(m >> 32));
        return ((unsigned long long)hi32 << 32) | lo;
    };
    auto lds_addr = [](const char* p) { return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p; };
    auto dma16 = [&](auto sc1_tag, const char* gaddr, const unsigned lds_off_v, const unsigned long long mask) {
        unsigned long long sv;
        const unsigned lds_off = __builtin_amdgcn_readfirstlane(lds_off_v);
        if constexpr (decltype(sc1_tag)::value)
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %3, off" PT_LOAD_SCOPE "\n\ts_mov_b64 exec, %0"
                         : "=&s"(sv) : "s"(mask), "s"(lds_off), "v"(gaddr) : "memory", "m0");
        else
            asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %3, off\n\ts_mov_b64 exec, %0"
                         : "=&s"(sv) : "s"(mask), "s"(lds_off), "v"(gaddr) : "memory", "m0");
    };
    // the 4 + 5 DMA instructions of one half step (input tile; weight fragment f' = tap * cb + mb of k-step ks)
    auto stage = [&](const Step2& d, const unsigned dst_l) {
        const unsigned long long on = uni64(d.src ? ~0ull : 0ull);
#pragma unroll
        for (int j = 0; j < T2_NJ; ++j)
            dma16(std::true_type{}, d.src + (d.ks ? goff1[j] : goff0[j]), dst_l + (j * 256 + wave * 64) * 16, uni64(jmask[j]) & on);
        const int sh = d.cb - 1;   // cb is 1 or 2
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int f = wave + 4 * k;
            const int tap = f >> sh, mb = f & sh;
            const char* src = d.w + ((((tap * 2 + d.ks) << sh) + mb) << 10) + lane * 16;
            dma16(std::false_type{}, src, dst_l + T2_IN_B + f * 1024, uni64((d.src && f < 9 * d.cb) ? ~0ull : 0ull));
        }
    };
    auto step_of = [&](const PLayer& l, int s) {
        Step2 d;
        const int c = s >> 1;
        d.src = pp.dense[l.in_sel] + tile_off + (long)c * pp.plane_b;
        d.w = l.w + (long)c * (18 * 1024 * l.cb);
        d.ks = s & 1;
        d.cb = l.cb;
        return d;
    };

    int f_up = up < 0 ? 0x7fffffff : 0, f_dn = dn < 0 ? 0x7fffffff : 0;
    int gs = 0;
    bool aborted = false;
    auto ensure_flags = [&](int need) {
        auto* word = (__attribute__((address_space(3))) int*)(smem + T2_WORD_OFF);
        if (tid == 0) {
            int bad = 0;
            unsigned spins = 0;
            while (f_up < need || f_dn < need) {
                if (up >= 0) f_up = __hip_atomic_load(pp.prog + up, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (dn >= 0) f_dn = __hip_atomic_load(pp.prog + dn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (f_up >= need && f_dn >= need) break;
                __builtin_amdgcn_s_sleep(2);
                if (++spins > SPIN_LIMIT || __hip_atomic_load(pp.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                    bad = 1;
                    break;
                }
            }
            if (bad) __hip_atomic_store(pp.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *word = bad;
        }
        __syncthreads();
        const int bad = *word;
        __syncthreads();
        if (bad) aborted = true;
    };
    bool wt = true;
    {
        int my_xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(my_xcc));
        auto* word = (__attribute__((address_space(3))) int*)(smem + T2_WORD_OFF);
        if (tid == 0) {
            __hip_atomic_store(pp.xcc + t, my_xcc + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int diff = pp.force_wt;
            for (int s = 0; s < 2 && !diff; ++s) {
                const int nb = s ? dn : up;
                if (nb < 0) continue;
                int v = 0;
                for (unsigned spins = 0; spins < SPIN_LIMIT; ++spins) {
                    v = __hip_atomic_load(pp.xcc + nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                if (v != my_xcc + 1) diff = 1;
            }
            *word = diff;
        }
        __syncthreads();
        wt = __builtin_amdgcn_readfirstlane(*word) != 0;
        __syncthreads();
    }
    auto publish = [&](int v) {
        if (tid == 0) {
            if (wt)
                __hip_atomic_store(pp.prog + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                __hip_atomic_store(pp.prog + t, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };
    bool pending_pub = false;
    int pub_val = 0;

    auto run_layer = [&](auto cb_tag, const int L, const PLayer& lay) {
        constexpr int CB = decltype(cb_tag)::value;
        constexpr int NREAD = 4 + 3 * CB, NMFMA = 6 * CB;
        unsigned long long p0 = 0;
        if (pp.prof) p0 = __builtin_amdgcn_s_memtime();
        float bias_v = 0.f;
        if (tid < CB * 32) bias_v = lay.bias[tid];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < CB * 32) ((float*)(smem + T2_BIAS_OFF))[tid] = bias_v;
        if (pending_pub) {
            publish(pub_val);
            pending_pub = false;
        }
        if (L > 0) ensure_flags(L);
        if (aborted) return;
        const bool last_layer = L + 1 >= pp.nlayers;
        const PLayer nlay = pp.layers[last_layer ? L : L + 1];
        const bool has_next_prefetch = !last_layer && !(nlay.flags & 8);
        unsigned long long p1 = 0;
        if (pp.prof) p1 = __builtin_amdgcn_s_memtime();
        floatx16 acc[CB][2];
#pragma unroll
        for (int mb = 0; mb < CB; ++mb)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][i][r] = 0.f;
        half8 P[2][4];
        half8 A[2][3][CB];
        auto load_group = [&](const char* sb, int dx, int set) {
#pragma unroll
            for (int r = 0; r < 4; ++r) P[set][r] = *(const half8*)(sb + poff[dx] + r * (T2_COLS * 16));
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int mb = 0; mb < CB; ++mb) A[set][dy][mb] = *(const half8*)(sb + woff + (((dy * 3 + dx) * CB + mb) << 10));
        };
        auto compute = [&](const Step2& nx) {
            const char* sb = smem + (gs & 1) * T2_STAGE_B;
            load_group(sb, 0, 0);
            stage(nx, lds_addr(smem + ((gs + 1) & 1) * T2_STAGE_B));   // everything up front: see variant 1
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                if (dx + 1 < 3) load_group(sb, dx + 1, (dx + 1) & 1);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int mb = 0; mb < CB; ++mb)
                            acc[mb][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[dx & 1][dy][mb], P[dx & 1][i + dy], acc[mb][i], 0, 0, 0);
                if (dx == 0) __builtin_amdgcn_sched_group_barrier(0x100, NREAD, 0);
                if (dx + 1 < 3) {
#pragma unroll
                    for (int k = 0; k < NMFMA; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (k < NREAD) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    if (NREAD > NMFMA) __builtin_amdgcn_sched_group_barrier(0x100, NREAD - NMFMA, 0);
                } else {
                    __builtin_amdgcn_sched_group_barrier(0x008, NMFMA, 0);
                }
            }
            ++gs;
        };
        const int ns = 2 * lay.nchunk;
        unsigned long long ts0 = p1, ts1 = 0, ts2 = 0, tw = p1 - p0;
        for (int s = 0; s < ns; ++s) {
            if (s > 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            Step2 nx;
            if (s + 1 < ns) {
                nx = step_of(lay, s + 1);
            } else {
                nx = step_of(nlay, 0);
                if (!has_next_prefetch) nx.src = nullptr;
            }
            compute(nx);
        }
        if (pp.prof) ts1 = __builtin_amdgcn_s_memtime();

        // ---- epilogue (D layout: lane (l31, hi) holds channels 8g + 4hi + (0..3) of pixel l31 for row i, group g)
        char* obase = pp.dense[lay.out_sel] + (long)img * pp.img_b + (long)lay.out_chunk0 * pp.plane_b;
        floatx4 bias4[CB][4];
#pragma unroll
        for (int mb = 0; mb < CB; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) bias4[mb][g] = *(const floatx4*)((const float*)(smem + T2_BIAS_OFF) + mb * 32 + g * 8 + hi * 4);
        constexpr bool lrelu = (CB == 1), r1 = (CB == 2);
        const bool r2 = r1 && (lay.flags & 4);
        struct ResForm { int lane, sm, sg; };
        const ResForm pixel_form{X * 64 + hi * 4, 32, 8}, frag_form{wc * 2048 + lane * 4, 1024, 256};
        const ResForm s1 = (!pp.frag_res || (lay.flags & 16)) ? pixel_form : frag_form;
        const ResForm s2 = (!pp.frag_res || (lay.flags & 32)) ? pixel_form : frag_form;
        const ResForm sd = pp.frag_res ? frag_form : pixel_form;
        const float* res1_src = (lay.flags & 64) ? pp.xrr : pp.xr;
        // one row at a time: the register budget is 256 per wave here (two waves per SIMD hide the load latency instead)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int Y = Y0 + wr * 2 + k;
            const bool valid = (Y < pp.H) && (X < pp.W);
            const long rowb = ((long)img * pp.H + Y) * pp.W * 64;
            floatx4 a1[CB][4], a2[CB][4], vv[CB][4];
            if (r1 && valid) {
                const float* q1 = res1_src + rowb + s1.lane;
#pragma unroll
                for (int mb = 0; mb < CB; ++mb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) a1[mb][g] = *(const floatx4*)(q1 + mb * s1.sm + g * s1.sg);
                if (r2) {
                    const float* q2 = pp.xrr + rowb + s2.lane;
#pragma unroll
                    for (int mb = 0; mb < CB; ++mb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) a2[mb][g] = *(const floatx4*)(q2 + mb * s2.sm + g * s2.sg);
                }
            }
#pragma unroll
            for (int mb = 0; mb < CB; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    floatx4 v;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = acc[mb][k][g * 4 + q];
                    v += bias4[mb][g];
                    if (r1) {
                        v = v * 0.2f + a1[mb][g];
                        if (r2) v = v * 0.2f + a2[mb][g];
                    }
                    vv[mb][g] = v;
                }
#pragma unroll
            for (int mb = 0; mb < CB; ++mb) {
                unsigned hp[4][2];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    floatx4 w = vv[mb][g];
                    if (lrelu) {
                        const floatx4 ws = w * 0.2f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) asm("v_max_f32 %0, %1, %2" : "=v"(w[q]) : "v"(w[q]), "v"(ws[q]));
                    }
                    half4 h4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) h4[q] = (_Float16)w[q];
                    const uint2 u = __builtin_bit_cast(uint2, h4);
                    hp[g][0] = u.x;
                    hp[g][1] = u.y;
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    auto s0 = __builtin_amdgcn_permlane32_swap(hp[2 * m][0], hp[2 * m + 1][0], false, false);
                    auto s1v = __builtin_amdgcn_permlane32_swap(hp[2 * m][1], hp[2 * m + 1][1], false, false);
                    typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
                    const uintx4 raw = {s0[0], s1v[0], s0[1], s1v[1]};
                    if (valid) {
                        char* o = obase + (long)mb * pp.plane_b + (long)(Y + 1) * pp.row_b + (X + 1) * PIX_B + m * 32 + hi * 16;
                        if (wt)
                            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(o), "v"(raw) : "memory");
                        else
                            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(o), "v"(raw) : "memory");
                    }
                }
            }
            if (r1 && valid) {
                float* q = (r2 ? pp.xrr : pp.xr) + rowb + sd.lane;
#pragma unroll
                for (int mb = 0; mb < CB; ++mb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) *(floatx4*)(q + mb * sd.sm + g * sd.sg) = vv[mb][g];
            }
        }
        if (pp.prof) ts2 = __builtin_amdgcn_s_memtime();
        const bool seam = last_layer || (nlay.flags & 8);
        if (seam) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            publish(L + 1);
            if (!last_layer) {
                ensure_flags(L + 1);
                if (aborted) return;
                stage(step_of(nlay, 0), lds_addr(smem + (gs & 1) * T2_STAGE_B));
            }
        } else {
            pending_pub = true;
            pub_val = L + 1;
        }
        if (pp.prof && tid == 0) {
            unsigned long long* q = pp.prof + ((long)blockIdx.x * pp.nlayers + L) * 6;
            q[0] = ts0; q[1] = ts1; q[2] = ts2; q[3] = tw | ((unsigned long long)(__builtin_amdgcn_s_memtime() - ts2) << 32);
            q[4] = 0;
        }
    };

    stage(step_of(pp.layers[0], 0), lds_addr(smem));
    for (int L = 0; L < pp.nlayers && !aborted; ++L) {
        const PLayer lay = pp.layers[L];
        if (lay.cb == 1)
            run_layer(std::integral_constant<int, 1>{}, L, lay);
        else
            run_layer(std::integral_constant<int, 2>{}, L, lay);
    }
}

}  // namespace

namespace srbh {

unsigned long long* g_ptrunk_prof = nullptr;   // set by tools/convbench only
static int g_trunk_timing = 0;                 // srbh_trunk_timing(): HIP events around the trunk launch(es), on their stream
static hipEvent_t g_trunk_ev[2];
static int g_trunk_ev_recorded = 0;       // a timed persistent launch has recorded both events since srbh_trunk_timing(1)
static const char* g_trunk_kernel = "none";    // srbh_trunk_kernel_name()

constexpr int MAX_BLOCKS = 64;   // layer-table capacity (RRDB blocks)
static size_t table_bytes() { return ((size_t)MAX_BLOCKS * 15 * sizeof(PLayer) + 255) & ~(size_t)255; }
static size_t prog_bytes(int B, int tpi) { return ((size_t)B * 2 * tpi * sizeof(int) + 255) & ~(size_t)255; }   // (x2: 4-row tiles of variant 2)

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

static void build_table(const srbh_rrdbnet_desc* d, std::vector<PLayer>& tab, int* final_cur) {
    int cur = 0, li = 0;
    for (int blk = 0; blk < d->num_block; ++blk)
        for (int r = 0; r < 3; ++r) {
            const srbh_conv_w* cw = d->rdb + (blk * 3 + r) * 5;
            for (int k = 0; k < 4; ++k)
                tab[li++] = PLayer{(const char*)cw[k].w, cw[k].bias, 2 + k, 1, cur, cur, 2 + k, 1 | (k == 0 ? 8 : 0)};
            // conv5: 64 = res1 comes from the xrr stream (first RDB of a block: xr == xrr there and the closing layer of
            // the previous block wrote only xrr); 16 / 32 = that stream still holds conv_first's pixel-order data
            tab[li++] = PLayer{(const char*)cw[4].w, cw[4].bias, 6, 2, cur, cur ^ 1, 0,
                               2 | (r == 2 ? 4 : 0) | (r == 0 ? 64 : 0) | (blk == 0 && r == 0 ? 16 : 0) | (blk == 0 && r == 2 ? 32 : 0)};
            cur ^= 1;
        }
    *final_cur = cur;
}

static int ptrunk2_run(const srbh_rrdbnet_desc* d, void* dense0, void* dense1, float* xr, float* xrr, int B, int H, int W,
                       void* aux, hipStream_t stream, int* used, int* final_cur) {
    *used = 0;
    if (W > TILE_W || d->num_block <= 0 || d->num_block > MAX_BLOCKS) return SRBH_OK;
    int dev = 0;
    SRBH_HIP(hipGetDevice(&dev));
    int ncu = 0;
    SRBH_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    const int tpi = (H + T2_H - 1) / T2_H;
    SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)ptrunk2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T2_LDS_B)));
    int per_cu = 0;
    SRBH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ptrunk2_kernel, 256, T2_LDS_B));
    if (per_cu < 2 || tpi > per_cu * ncu) return SRBH_OK;
    const int slots = 2 * ncu;   // co-resident workgroups the protocol may rely on
    const int nl = d->num_block * 15;
    std::vector<PLayer> tab(nl);
    build_table(d, tab, final_cur);
    const int tpi8 = (H + TILE_H - 1) / TILE_H;
    char* a = (char*)aux;
    PLayer* d_tab = nullptr;
    if (int rc = device_table(tab, &d_tab)) return rc;
    int* d_prog = (int*)(a + table_bytes());
    int* d_err = (int*)(a + ptrunk_err_offset(B, tpi8));
    int* d_xcc = (int*)(a + ptrunk_err_offset(B, tpi8) + 256);
    hipLaunchKernelGGL(ptrunk_reset_kernel, dim3((B * tpi + 255) / 256), dim3(256), 0, stream, d_prog, d_xcc, d_err, B * tpi);
    SRBH_HIP(hipGetLastError());
    const Act16Geo g = act16_geo(B, 6, H, W);
    const int imgs_per_launch = slots / tpi;
    for (int b0 = 0; b0 < B; b0 += imgs_per_launch) {
        const int nb = (B - b0) < imgs_per_launch ? (B - b0) : imgs_per_launch;
        PParams pp{};
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
        pp.xcc = d_xcc + b0 * tpi;
        const char* e = getenv("SRBH_PT_WT");
        pp.force_wt = (e && atoi(e) == 1) ? 1 : 0;
        if (getenv("SRBH_PT_PROF") && !g_ptrunk_prof)
            SRBH_HIP(hipMalloc(&g_ptrunk_prof, (size_t)slots * MAX_BLOCKS * 15 * 6 * 8));
        pp.prof = g_ptrunk_prof;
        const char* fr = getenv("SRBH_PT_FRAGRES");
        pp.frag_res = (W == TILE_W) && !(fr && atoi(fr) == 0);
        if (g_trunk_timing && b0 == 0) SRBH_HIP(hipEventRecord(g_trunk_ev[0], stream));
        g_trunk_kernel = "ptrunk2_kernel";
        hipLaunchKernelGGL(ptrunk2_kernel, dim3(pp.nblocks), dim3(256), T2_LDS_B, stream, pp);
        SRBH_HIP(hipGetLastError());
    }
    if (g_trunk_timing) { SRBH_HIP(hipEventRecord(g_trunk_ev[1], stream)); g_trunk_ev_recorded = 1; }
    if (getenv("SRBH_PT_PROF") && g_ptrunk_prof) {
        SRBH_HIP(hipStreamSynchronize(stream));
        const int nblk = (B < imgs_per_launch ? B : imgs_per_launch) * tpi;
        std::vector<unsigned long long> h((size_t)nblk * nl * 6);
        SRBH_HIP(hipMemcpy(h.data(), g_ptrunk_prof, h.size() * 8, hipMemcpyDeviceToHost));
        double cyc = 0, loop[5] = {0}, epi[5] = {0}, pub[5] = {0}, wait[5] = {0}, tot5[5] = {0};
        for (int b = 0; b < nblk; ++b) cyc += (double)(h[((size_t)b * nl + nl - 1) * 6 + 2] - h[(size_t)b * nl * 6]);
        fprintf(stderr, "[srbh] ptrunk2: avg %.0f shader cycles per workgroup (2 workgroups per CU)\n", cyc / nblk);
        for (int b = 0; b < nblk; ++b)
            for (int L = 1; L + 1 < nl; ++L) {
                const unsigned long long* q = &h[((size_t)b * nl + L) * 6];
                const unsigned long long* qn = &h[((size_t)b * nl + L + 1) * 6];
                const int k = L % 5;
                loop[k] += (double)(q[1] - q[0]); epi[k] += (double)(q[2] - q[1]); pub[k] += (double)(q[3] >> 32);
                wait[k] += (double)(q[3] & 0xffffffffu); tot5[k] += (double)(qn[0] - q[0]);
            }
        const double cnt = (double)nblk * (nl - 2) / 5.0;
        for (int k = 0; k < 5; ++k)
            fprintf(stderr, "[srbh]   conv%d: loop %.0f (flag-wait %.0f) | epilogue %.0f | publish/seam %.0f | start-to-start %.0f\n",
                    k + 1, loop[k] / cnt, wait[k] / cnt, epi[k] / cnt, pub[k] / cnt, tot5[k] / cnt);
    }
    *used = 1;
    return SRBH_OK;
}

// returns SRBH_OK and sets *used = 1 when the persistent path ran, *used = 0 when the shape is not eligible
// train_stride > 0 = the TRAINING forward (srbh_rrdbnet_trunk_train_forward_persistent): dense0 is RDB 0's buffer of a row of buffers train_stride
// bytes apart (dense1 ignored), every plane is stored whole, and the trunk's fp32 output goes to `xr` in pixel order; variant 3 only.
// mask != nullptr = the BACKWARD of the dense blocks (srbh_rrdbnet_trunk_train_backward_persistent; needs train_stride > 0): `d` holds the gradient convs'
// bf16 packs in running (reverse) order, mask / mask_stride walk the saved forward buffers.
int ptrunk_run(const srbh_rrdbnet_desc* d, void* dense0, void* dense1, float* xr, float* xrr, int B, int H, int W,
               void* aux, hipStream_t stream, int* used, int* final_cur, long train_stride, const void* mask, long mask_stride) {
    *used = 0;
    if (mask && train_stride <= 0) return SRBH_OK;
    if (W > TILE_W || d->num_block <= 0 || d->num_block > MAX_BLOCKS) return SRBH_OK;
    int dev = 0;
    SRBH_HIP(hipGetDevice(&dev));
    int ncu = 0;
    SRBH_HIP(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
    // variant 2 (two 4-row workgroups per CU, see ptrunk2_kernel) or variant 1 (one 8-row workgroup per CU)
    const char* ve = getenv("SRBH_PT_VARIANT");
    const int variant = train_stride > 0 ? 3 : (ve ? atoi(ve) : PT_DEFAULT_VARIANT);
    if (train_stride > 0 && !(W == TILE_W && (H % TILE_H) == 0)) return SRBH_OK;
    if (variant == 2) {
        const int rc2 = ptrunk2_run(d, dense0, dense1, xr, xrr, B, H, W, aux, stream, used, final_cur);
        if (rc2 != SRBH_OK || *used) return rc2;
    }
    const int tpi = (H + TILE_H - 1) / TILE_H;
    if (tpi > ncu) return SRBH_OK;
    constexpr int LDS_B = P_LDS_B;
    SRBH_ONCE_PER_DEVICE({
        SRBH_HIP(hipFuncSetAttribute((const void*)ptrunk3_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B));
        SRBH_HIP(hipFuncSetAttribute((const void*)ptrunk3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B));
        SRBH_HIP(hipFuncSetAttribute((const void*)ptrunk3_kernel<0, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B));
        SRBH_HIP(hipFuncSetAttribute((const void*)ptrunk_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B));
        SRBH_HIP(hipFuncSetAttribute((const void*)ptrunk_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B));
        SRBH_HIP(hipFuncSetAttribute((const void*)ptrunk_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B));
    });
    int per_cu = 0;
    SRBH_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, ptrunk_kernel<false, true>, 256, LDS_B));
    if (per_cu < 1) return SRBH_OK;

    const int nl = d->num_block * 15;
    std::vector<PLayer> tab(nl);
    int cur = 0, li = 0;
    // The RDB-level fp32 residual stream lives in registers (ptrunk_kernel<., true>); SRBH_PT_REGRES=0 selects the
    // instantiation that keeps it in memory (A/B aid).
    const char* rr = getenv("SRBH_PT_REGRES");
    const bool reg_res = !(rr && atoi(rr) == 0);
    for (int blk = 0; blk < d->num_block; ++blk)
        for (int r = 0; r < 3; ++r) {
            const srbh_conv_w* cw = d->rdb + (blk * 3 + r) * 5;
            for (int k = 0; k < 4; ++k)
                tab[li++] = PLayer{(const char*)cw[k].w, cw[k].bias, 2 + k, 1, cur, cur, 2 + k, 1 | (k == 0 ? 8 : 0)};
            // conv5: 64 = res1 comes from the xrr stream (first RDB of a block: xr == xrr there and the closing layer of
            // the previous block wrote only xrr); 16 / 32 = that stream still holds conv_first's pixel-order data
            tab[li++] = PLayer{(const char*)cw[4].w, cw[4].bias, 6, 2, cur, cur ^ 1, 0,
                               2 | (r == 2 ? 4 : 0) | (r == 0 ? 64 : 0) | (blk == 0 && r == 0 ? 16 : 0) | (blk == 0 && r == 2 ? 32 : 0)};
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
        {
            const char* e = getenv("SRBH_PT_FRAGRES");   // debugging aid: 0 keeps the residual streams in pixel order
            pp.frag_res = (W == TILE_W) && !(e && atoi(e) == 0);
        }
        if (g_trunk_timing && b0 == 0) SRBH_HIP(hipEventRecord(g_trunk_ev[0], stream));
        // variant 3 (RDB-unrolled instruction stream, see srbh_ptrunk3_kernel.h): full 8 x 64 tiles only
        const bool v3 = variant == 3 && W == TILE_W && (H % TILE_H) == 0 && (reg_res || train_stride > 0);
        g_trunk_kernel = v3 ? "ptrunk3_kernel" : "ptrunk_kernel";
        if (mask) {
            SRBH_REQUIRE(v3, "ptrunk_run: the backward form is ptrunk3_kernel's");
            hipLaunchKernelGGL((ptrunk3_kernel<0, 1>), dim3(pp.nblocks), dim3(256), LDS_B, stream, pp);
        } else if (v3 && pp.prof)
            hipLaunchKernelGGL((ptrunk3_kernel<1>), dim3(pp.nblocks), dim3(256), LDS_B, stream, pp);
        else if (v3)
            hipLaunchKernelGGL((ptrunk3_kernel<0>), dim3(pp.nblocks), dim3(256), LDS_B, stream, pp);
        else if (pp.prof)
            hipLaunchKernelGGL((ptrunk_kernel<true, true>), dim3(pp.nblocks), dim3(256), LDS_B, stream, pp);
        else if (reg_res)
            hipLaunchKernelGGL((ptrunk_kernel<false, true>), dim3(pp.nblocks), dim3(256), LDS_B, stream, pp);
        else
            hipLaunchKernelGGL((ptrunk_kernel<false, false>), dim3(pp.nblocks), dim3(256), LDS_B, stream, pp);
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
