// Internal helpers shared by the libsrbh translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <vector>
#include "srbh.h"

namespace srbh {

constexpr int PIX_B = 64;       // bytes of one ACT16 pixel record (32 fp16 channels)
constexpr int CHUNK_C = 32;     // channels per chunk plane
constexpr int TILE_W = 64;      // output columns per workgroup
constexpr int TILE_H = 8;       // output rows per workgroup

void set_error(const char* fmt, ...);
// Which form of a head entry point ran (srbh_path_counters): several C entry points choose between a specialised persistent kernel and
// the general template (or between one fused pass and two launches) from the shapes they are given -- same results, different speed.
// SURVEY 8(b) "no silent fallback": the choice is counted, bench.py prints the counts (`train_step.head_paths`), so a hot shape
// sliding back to the slow form shows as a count, not only as time.  (Round 4: the fp16 feature hand-off had sent the inference
// chain's 64-channel block entry back to two template launches -- found only in a kernel trace.)
enum PathCounter { PATH_HCONV16 = 0, PATH_HCONV_TEMPLATE, PATH_ENTRY_FUSED, PATH_ENTRY_SPLIT, PATH_WGRAD16, PATH_WGRAD_B16_GENERIC,
                   PATH_WGRAD_F32, PATH_WGRAD_ENTRY_FUSED, PATH_WGRAD_ENTRY_SPLIT, PATH_HCONV_UP, PATH_HBWD16, PATH_HBLOCK16, PATH_N };
extern unsigned long long g_path_counters[PATH_N];
inline void count_path(int i) { ++g_path_counters[i]; }
// Stream-ordered zero fill of `bytes` (a multiple of 4) by a KERNEL.  Not hipMemsetAsync: captured into a HIP graph, a memset node is
// not kept in order with the kernels of the previous replay of the same graph (srbh_ptrunk.hip, ptrunk_reset_kernel), and every
// libsrbh call must stay correct inside back-to-back graph replays (harness.TrainStep(graph=True), predict_tiles).
int zero_async(void* p, size_t bytes, hipStream_t st);
int hip_fail(hipError_t e, const char* what);

#define SRBH_HIP(call)                                         \
    do {                                                       \
        hipError_t e_ = (call);                                \
        if (e_ != hipSuccess) return srbh::hip_fail(e_, #call); \
    } while (0)

#define SRBH_REQUIRE(cond, ...)            \
    do {                                   \
        if (!(cond)) {                     \
            srbh::set_error(__VA_ARGS__);  \
            return SRBH_ERR_ARG;           \
        }                                  \
    } while (0)

// hipFuncSetAttribute applies to the CURRENT device only (one process may drive several GPUs): run `stmt` the first
// time this call site is reached on each device.
#define SRBH_ONCE_PER_DEVICE(stmt)                        \
    do {                                                  \
        static unsigned long long done_ = 0;              \
        int dev_ = 0;                                     \
        SRBH_HIP(hipGetDevice(&dev_));                    \
        if (!((done_ >> (dev_ & 63)) & 1ull)) {           \
            stmt;                                         \
            done_ |= 1ull << (dev_ & 63);                 \
        }                                                 \
    } while (0)

// ACT16 geometry: [B][chunks][H+2][W+2][32] fp16 + read slack so tiled kernels may over-read.
struct Act16Geo {
    int row_b;       // bytes per padded row
    int plane_b;     // bytes per chunk plane
    long img_b;      // bytes per image
    size_t total_b;  // bytes including slack
};
inline Act16Geo act16_geo(int B, int chunks, int H, int W) {
    Act16Geo g;
    g.row_b = (W + 2) * PIX_B;
    g.plane_b = (H + 2) * g.row_b;
    g.img_b = (long)chunks * g.plane_b;
    // slack: a tile may read TILE_H+2 rows and TILE_W+2 columns starting anywhere inside the last plane
    size_t slack = (size_t)(TILE_H + 4) * (size_t)((W + 2 > TILE_W + 2) ? (W + 2) : (TILE_W + 2)) * PIX_B + 8192;
    g.total_b = (size_t)B * g.img_b + slack;
    g.total_b = (g.total_b + 255) & ~(size_t)255;
    return g;
}


// persistent trunk (srbh_ptrunk.hip)
size_t ptrunk_aux_bytes(int B, int tiles_per_img);
size_t ptrunk_err_offset(int B, int tiles_per_img);
int ptail_run(const srbh_conv3x3_args* a, hipStream_t stream, int* used);
int ptrunk_run(const srbh_rrdbnet_desc* d, void* dense0, void* dense1, float* xr, float* xrr, int B, int H, int W,
               void* aux, hipStream_t stream, int* used, int* final_cur, long train_stride = 0, const void* mask = nullptr, long mask_stride = 0);


#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)
// BatchNorm partial sums [NS = 64 slots][2 moments][C] (double) -> the two totals of channel threadIdx.x (threads >= C get garbage), by ONE block
// of 256 threads: thread t owns the value v = t % (2 C) (moment * C + channel) and the slots g, g + G, ... (g = t / (2 C), G = 256 / (2 C)
// groups): all of its loads are in flight together, then the G partials are added in a fixed order through `red` (256 doubles of LDS).
// clear: every slot is zeroed behind its read (self-cleaning buffer).  Deterministic.
__device__ __forceinline__ void fold_stat_slots(double* stats, const int C, const int clear, double* red, double& s, double& q) {
    constexpr int NS = 64;
    const int t = threadIdx.x, V = 2 * C, G = 256 / V, v = t % V, g = t / V;
    double part = 0.0;
    if (g < G) {
        for (int k = g; k < NS; k += G) part += stats[(long)k * V + v];
        if (clear)
            for (int k = g; k < NS; k += G) stats[(long)k * V + v] = 0.0;
        red[g * V + v] = part;
    }
    __syncthreads();
    s = 0.0;
    q = 0.0;
    if (t < C)
        for (int k = 0; k < G; ++k) {
            s += red[k * V + t];
            q += red[k * V + C + t];
        }
}

// two fp32 -> one dword of bf16 (lo in the low half), round to nearest even: ONE v_cvt_pk_bf16_f32 on gfx950 (the integer sequence
// u + 0x7fff + ((u >> 16) & 1) >> 16 it replaces is 4 VALU ops per element in kernels whose staging is VALU-bound; same bits for every
// finite input and the infinities)
__device__ __forceinline__ unsigned bf16x2_rne(float lo, float hi) {
    typedef float f2_t __attribute__((ext_vector_type(2)));
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    const f2_t f = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f, b2_t));
}
#endif
}  // namespace srbh
