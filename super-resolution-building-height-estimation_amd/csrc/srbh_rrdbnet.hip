// srbh_rrdbnet.hip -- host-side driver that runs RRDBNet.forward_feature / forward
// (reference SR/rrdbnet_arch.py:208-240) as a fixed sequence of libsrbh kernel launches on one stream.
//
// Dense-block concat is never materialised: one ACT16 buffer with 6 chunk planes per image holds
// [x | x1 | x2 | x3 | x4] (SR/rrdbnet_arch.py:137-141); conv_k reads planes 0..k and writes plane k+1;
// conv5 writes the next block's x into planes 0..1 of the other (ping-pong) buffer.  The fp32
// residual streams (x5*0.2+x at :143, out*0.2+x at :167, feat+body_feat at :234) live in three RES32 buffers.
#include <stdlib.h>
#include "srbh_internal.h"

using namespace srbh;

namespace {

struct WsLayout {
    size_t d0, d1, feat, xr, xrr, u2, u3, u4, aux, total;
    size_t dense_b;
};

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// The persistent trunk kernel needs every workgroup co-resident; if something else holds CUs / LDS its bounded spins
// time out, it sets an error word and drains, and the kernels behind it would run on stale activations.  This guard
// runs last in the forward: error word clear -> every workgroup returns at once; set -> the whole output becomes NaN,
// so a timed-out launch can never be mistaken for a result (features, mosaics and losses all turn NaN).  The host-side
// check (srbh_rrdbnet_last_status) stays the way to get the reason.
__global__ __launch_bounds__(256) void poison_on_error_kernel(const int* __restrict__ err, float* __restrict__ out, size_t n) {
    if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
    const float nan = __builtin_nanf("");
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = nan;
}

WsLayout ws_layout(int B, int H, int W, int want_forward) {
    WsLayout L;
    size_t off = 0;
    L.dense_b = act16_geo(B, 6, H, W).total_b;
    L.d0 = off; off = align256(off + L.dense_b);
    L.d1 = off; off = align256(off + L.dense_b);
    size_t res_b = (size_t)B * H * W * 64 * sizeof(float);
    L.feat = off; off = align256(off + res_b);
    L.xr = off; off = align256(off + res_b);
    L.xrr = off; off = align256(off + res_b);
    L.u2 = off; off = align256(off + act16_geo(B, 2, 2 * H, 2 * W).total_b);
    L.u3 = off; off = align256(off + act16_geo(B, 2, 4 * H, 4 * W).total_b);
    L.u4 = off;
    if (want_forward == 1) off = align256(off + act16_geo(B, 2, 4 * H, 4 * W).total_b);      // (2 = forward_feature as fp16: no conv_last)
    L.aux = off;   // persistent-trunk layer table, progress counters, error word
    off = align256(off + ptrunk_aux_bytes(B, (H + TILE_H - 1) / TILE_H));
    L.total = off;
    return L;
}

}  // namespace

extern "C" size_t srbh_rrdbnet_workspace_bytes(int B, int H, int W, int want_forward) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return ws_layout(B, H, W, want_forward).total;
}

extern "C" int srbh_rrdbnet_forward(const srbh_rrdbnet_desc* d, const float* x, float* out, int B, int H, int W,
                                    int want_forward, void* ws, size_t ws_bytes, void* stream) {
    SRBH_REQUIRE(d && x && out && ws, "srbh_rrdbnet_forward: null pointer");
    SRBH_REQUIRE(B > 0 && H > 0 && W > 0, "srbh_rrdbnet_forward: bad geometry B=%d H=%d W=%d", B, H, W);
    SRBH_REQUIRE(d->num_block >= 0 && d->rdb != nullptr, "srbh_rrdbnet_forward: bad descriptor");
    const WsLayout L = ws_layout(B, H, W, want_forward);
    if (ws_bytes < L.total) {
        set_error("srbh_rrdbnet_forward: workspace %zu bytes < required %zu", ws_bytes, L.total);
        return SRBH_ERR_WORKSPACE;
    }
    char* base = (char*)ws;
    void* D[2] = {base + L.d0, base + L.d1};
    float* feat = (float*)(base + L.feat);
    float* xr = (float*)(base + L.xr);
    float* xrr = (float*)(base + L.xrr);
    void* U2 = base + L.u2;
    void* U3 = base + L.u3;
    void* U4 = base + L.u4;

    int rc = srbh_conv_first_f32(x, d->conv_first_w, d->conv_first_b, B, d->num_in_ch, H, W, feat, xr, xrr, D[0], 6,
                                 stream);
    if (rc) return rc;

    srbh_conv3x3_args a;
    int cur = 0;
    int used_persistent = 0;
    {
        const char* env = getenv("SRBH_PERSISTENT");
        const bool want = !(env && env[0] == '0');
        if (want) {
            rc = ptrunk_run(d, D[0], D[1], xr, xrr, B, H, W, base + L.aux, (hipStream_t)stream, &used_persistent, &cur);
            if (rc) return rc;
        }
    }
    for (int blk = 0; !used_persistent && blk < d->num_block; ++blk) {
        for (int r = 0; r < 3; ++r) {
            const srbh_conv_w* cw = d->rdb + (blk * 3 + r) * 5;
            for (int k = 0; k < 4; ++k) {  // conv1..conv4: lrelu(conv(cat(x, x1..xk)))
                a = srbh_conv3x3_args{};
                a.in = D[cur]; a.in_chunks_total = 6; a.in_chunk0 = 0; a.in_chunks = 2 + k;
                a.w = cw[k].w; a.bias = cw[k].bias; a.cout = 32;
                a.B = B; a.H = H; a.W = W; a.lrelu = 1;
                a.out16 = D[cur]; a.out16_chunks_total = 6; a.out16_chunk0 = 2 + k;
                if ((rc = srbh_conv3x3_f16(&a, stream))) return rc;
            }
            a = srbh_conv3x3_args{};  // conv5 + x5*0.2 + x (+ out*0.2 + x_rrdb at the end of the RRDB)
            a.in = D[cur]; a.in_chunks_total = 6; a.in_chunk0 = 0; a.in_chunks = 6;
            a.w = cw[4].w; a.bias = cw[4].bias; a.cout = 64;
            a.B = B; a.H = H; a.W = W;
            a.res_scale = 0.2f; a.res1 = xr; a.res1_update = 1;
            if (r == 2) { a.res2 = xrr; a.res2_scale = 0.2f; a.res2_update = 1; }
            a.out16 = D[cur ^ 1]; a.out16_chunks_total = 6; a.out16_chunk0 = 0;
            if ((rc = srbh_conv3x3_f16(&a, stream))) return rc;
            cur ^= 1;
        }
    }
    // conv_body + trunk skip
    a = srbh_conv3x3_args{};
    a.in = D[cur]; a.in_chunks_total = 6; a.in_chunk0 = 0; a.in_chunks = 2;
    a.w = d->conv_body.w; a.bias = d->conv_body.bias; a.cout = 64;
    a.B = B; a.H = H; a.W = W; a.skip = feat;
    a.out16 = D[cur ^ 1]; a.out16_chunks_total = 6; a.out16_chunk0 = 0;
    if ((rc = srbh_conv3x3_f16(&a, stream))) return rc;
    // conv_up1 / conv_up2 read through the nearest-x2 index map
    a = srbh_conv3x3_args{};
    a.in = D[cur ^ 1]; a.in_chunks_total = 6; a.in_chunk0 = 0; a.in_chunks = 2;
    a.w = d->conv_up1.w; a.bias = d->conv_up1.bias; a.cout = 64;
    a.B = B; a.H = 2 * H; a.W = 2 * W; a.upsample2x = 1; a.lrelu = 1;
    a.out16 = U2; a.out16_chunks_total = 2; a.out16_chunk0 = 0;
    if ((rc = srbh_conv3x3_f16(&a, stream))) return rc;
    a = srbh_conv3x3_args{};
    a.in = U2; a.in_chunks_total = 2; a.in_chunk0 = 0; a.in_chunks = 2;
    a.w = d->conv_up2.w; a.bias = d->conv_up2.bias; a.cout = 64;
    a.B = B; a.H = 4 * H; a.W = 4 * W; a.upsample2x = 1; a.lrelu = 1;
    a.out16 = U3; a.out16_chunks_total = 2; a.out16_chunk0 = 0;
    if ((rc = srbh_conv3x3_f16(&a, stream))) return rc;
    // conv_hr
    a = srbh_conv3x3_args{};
    a.in = U3; a.in_chunks_total = 2; a.in_chunk0 = 0; a.in_chunks = 2;
    a.w = d->conv_hr.w; a.bias = d->conv_hr.bias; a.cout = 64;
    a.B = B; a.H = 4 * H; a.W = 4 * W;
    const int* err_word = (const int*)(base + L.aux + ptrunk_err_offset(B, (H + TILE_H - 1) / TILE_H));
    auto guard = [&](int out_c) -> int {
        if (!used_persistent) return SRBH_OK;
        hipLaunchKernelGGL(poison_on_error_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, err_word, out,
                           (size_t)B * 16 * H * W * out_c);
        SRBH_HIP(hipGetLastError());
        return SRBH_OK;
    };
    if (want_forward == 2) {      // forward_feature as a dense fp16 NHWC tensor (the 16-bit head kernels stage it verbatim)
        a.out16 = out; a.out16_chunks_total = 2; a.out16_chunk0 = 0; a.out16_nhwc = 1;
        if ((rc = srbh_conv3x3_f16(&a, stream))) return rc;
        return guard(32);         // (64 halves = 32 words per pixel)
    }
    if (!want_forward) {
        a.out32 = out; a.out32_c = 64;
        if ((rc = srbh_conv3x3_f16(&a, stream))) return rc;
        return guard(64);
    }
    a.lrelu = 1;
    a.out16 = U4; a.out16_chunks_total = 2; a.out16_chunk0 = 0;
    if ((rc = srbh_conv3x3_f16(&a, stream))) return rc;
    SRBH_REQUIRE(d->conv_last.w && d->num_out_ch >= 1 && d->num_out_ch <= 32,
                 "srbh_rrdbnet_forward: conv_last needs 1..32 output channels (got %d)", d->num_out_ch);
    a = srbh_conv3x3_args{};
    a.in = U4; a.in_chunks_total = 2; a.in_chunk0 = 0; a.in_chunks = 2;
    a.w = d->conv_last.w; a.bias = d->conv_last.bias; a.cout = 32;
    a.B = B; a.H = 4 * H; a.W = 4 * W;
    a.out32 = out; a.out32_c = d->num_out_ch;
    if ((rc = srbh_conv3x3_f16(&a, stream))) return rc;
    return guard(d->num_out_ch);
}

extern "C" int srbh_rrdbnet_last_status(const void* ws, int B, int H, int W, int want_forward, void* stream) {
    SRBH_REQUIRE(ws && B > 0 && H > 0 && W > 0, "srbh_rrdbnet_last_status: bad arguments");
    const WsLayout L = ws_layout(B, H, W, want_forward);
    const int tpi = (H + TILE_H - 1) / TILE_H;
    SRBH_HIP(hipStreamSynchronize((hipStream_t)stream));
    int err = 0;
    const size_t eoff = ptrunk_err_offset(B, tpi);
    SRBH_HIP(hipMemcpy(&err, (const char*)ws + L.aux + eoff, sizeof(int), hipMemcpyDeviceToHost));
    if (err) set_error("persistent trunk kernel timed out waiting for a neighbour workgroup (err=%d)", err);
    return err ? -3 : SRBH_OK;
}


// ---- training path of the trunk (second slice; host mirror: rrdbnet_autograd.py "fast") -------------------------------
// The two loops below are the per-layer launch sequence above with every RDB's dense buffer KEPT (forward) and its mirror image on
// gradients (backward); they live here rather than in Python because at batch 8 a step is ~2 400 launches of a few microseconds each
// and the ctypes call overhead (~10 us) was the whole runtime.
extern "C" int srbh_rrdbnet_trunk_train_forward(const srbh_rrdbnet_desc* d, float* xr, float* xrr, void* dense_all, size_t dense_stride,
                                                int B, int H, int W, void* stream) {
    SRBH_REQUIRE(d && d->rdb && xr && xrr && dense_all && B > 0 && H > 0 && W > 0, "srbh_rrdbnet_trunk_train_forward: bad arguments");
    // xr == xrr == feat on entry (the caller's copies); dense buffer 0 receives feat as fp16 planes 0..1
    int rc = srbh_nhwc32_to_act16(xr, dense_all, B, 64, H, W, 6, 0, 1.0f, 0, stream);
    if (rc) return rc;
    srbh_conv3x3_args a;
    const int n_rdb = d->num_block * 3;
    for (int i = 0; i < n_rdb; ++i) {
        char* D = (char*)dense_all + (size_t)i * dense_stride;
        const srbh_conv_w* cw = d->rdb + i * 5;
        for (int k = 0; k < 4; ++k) {
            a = srbh_conv3x3_args{};
            a.in = D; a.in_chunks_total = 6; a.in_chunk0 = 0; a.in_chunks = 2 + k;
            a.w = cw[k].w; a.bias = cw[k].bias; a.cout = 32;
            a.B = B; a.H = H; a.W = W; a.lrelu = 1;
            a.out16 = D; a.out16_chunks_total = 6; a.out16_chunk0 = 2 + k;
            if ((rc = srbh_conv3x3_f16(&a, stream))) return rc;
        }
        a = srbh_conv3x3_args{};
        a.in = D; a.in_chunks_total = 6; a.in_chunk0 = 0; a.in_chunks = 6;
        a.w = cw[4].w; a.bias = cw[4].bias; a.cout = 64;
        a.B = B; a.H = H; a.W = W;
        a.res_scale = 0.2f; a.res1 = xr; a.res1_update = 1;
        if (i % 3 == 2) { a.res2 = xrr; a.res2_scale = 0.2f; a.res2_update = 1; }
        a.out16 = D + dense_stride; a.out16_chunks_total = 6; a.out16_chunk0 = 0;
        if ((rc = srbh_conv3x3_f16(&a, stream))) return rc;
    }
    return SRBH_OK;
}

/* The same forward as ONE launch of the persistent trunk kernel (round 5): the inference trunk's ptrunk3_kernel walking a row of dense buffers
 * (RDB i in dense_all + i * dense_stride) with every plane stored whole, its fp32 output written to xr in pixel order.  Same arithmetic in the
 * same order as the per-layer sequence above: xr and every saved plane come out bit-identical (tests/test_sr_stage.py).  aux: scratch of
 * srbh_rrdbnet_trunk_train_aux_bytes(B, H, W) bytes (0 = this geometry is not the kernel's: 64-pixel-wide tiles, H %% 8 == 0).  *used = 0:
 * nothing was launched, call srbh_rrdbnet_trunk_train_forward.  A halo-exchange timeout (never seen) turns xr into NaN. */
extern "C" size_t srbh_rrdbnet_trunk_train_aux_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W != TILE_W || (H % TILE_H) != 0) return 0;
    return ptrunk_aux_bytes(B, (H + TILE_H - 1) / TILE_H);
}
extern "C" int srbh_rrdbnet_trunk_train_forward_persistent(const srbh_rrdbnet_desc* d, float* xr, float* xrr, void* dense_all, size_t dense_stride,
                                                           int B, int H, int W, void* aux, void* stream, int* used) {
    SRBH_REQUIRE(d && d->rdb && xr && xrr && dense_all && aux && used && B > 0 && H > 0 && W > 0 && dense_stride > 0,
                 "srbh_rrdbnet_trunk_train_forward_persistent: bad arguments");
    *used = 0;
    static const bool off = getenv("SRBH_SR_PTRUNK") && getenv("SRBH_SR_PTRUNK")[0] == '0';
    if (off || srbh_rrdbnet_trunk_train_aux_bytes(B, H, W) == 0) return SRBH_OK;
    int rc = srbh_nhwc32_to_act16(xr, dense_all, B, 64, H, W, 6, 0, 1.0f, 0, stream);      // dense buffer 0 <- feat as fp16 planes 0..1 (xr == xrr == feat)
    if (rc) return rc;
    int cur = 0;
    if ((rc = ptrunk_run(d, dense_all, nullptr, xr, xrr, B, H, W, aux, (hipStream_t)stream, used, &cur, (long)dense_stride))) return rc;
    if (!*used) return SRBH_OK;
    const int* err_word = (const int*)((const char*)aux + ptrunk_err_offset(B, (H + TILE_H - 1) / TILE_H));
    hipLaunchKernelGGL(poison_on_error_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, err_word, xr, (size_t)B * H * W * 64);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

// g_a: gradient of the trunk output on entry (fp32 NHWC64); g_b, g_c: scratch of the same size.  Returns the gradient of the trunk
// input in *g_out (one of the three).  packs: per RDB `pack_stride` bytes, gradient conv j (dX4, dX3, dX2, dX1, dx) at pack_off[j].
// dw_all: per RDB 239 616 floats in conv1..conv5 order (OIHW each); db_all: per RDB 192 floats in G order [g5 (64) | g4 | g3 | g2 | g1].
// The weight / bias gradients of RDB i only READ what the gradient convs of RDB i produced (G) and the saved planes: they run on a
// side stream next to the gradient convs of RDB i-1 -- at small batch a conv launch fills a quarter of the chip (64 workgroups at
// batch 8), so the two chains overlap almost completely.  G is double buffered for that (G and G + g_stride).
namespace {
struct SideStream {
    hipStream_t s = nullptr;
    hipEvent_t ready[2] = {nullptr, nullptr}, done[2] = {nullptr, nullptr};
    int dev = -1;
};
SideStream g_side;
int side_init() {
    int dev = 0;
    SRBH_HIP(hipGetDevice(&dev));
    if (g_side.s && g_side.dev == dev) return SRBH_OK;
    SRBH_HIP(hipStreamCreateWithFlags(&g_side.s, hipStreamNonBlocking));
    for (int k = 0; k < 2; ++k) {
        SRBH_HIP(hipEventCreateWithFlags(&g_side.ready[k], hipEventDisableTiming));
        SRBH_HIP(hipEventCreateWithFlags(&g_side.done[k], hipEventDisableTiming));
    }
    g_side.dev = dev;
    return SRBH_OK;
}
}  // namespace

/* bytes of `wgrad_ws` for srbh_rrdbnet_trunk_train_backward: the partial-sum workspaces of a dense block's five weight gradients side by side */
extern "C" size_t srbh_rrdbnet_trunk_wgrad_ws_bytes(void) {
    static const int COUT[5] = {32, 32, 32, 32, 64}, CIN[5] = {64, 96, 128, 160, 192};
    size_t n = 0;
    for (int k = 0; k < 5; ++k) n += srbh_hwgrad_ws_bytes(COUT[k], CIN[k], 3);
    return n;
}

extern "C" int srbh_rrdbnet_trunk_train_backward(int num_block, const void* dense_all, size_t dense_stride, const void* packs, size_t pack_stride,
                                                 const size_t* pack_off, float* g_a, float* g_b, float* g_c, float** g_out, void* G, size_t g_stride,
                                                 float* dw_all, float* db_all, float* wgrad_ws, int B, int H, int W, void* stream) {
    SRBH_REQUIRE(num_block > 0 && dense_all && packs && pack_off && g_a && g_b && g_c && g_out && G && dw_all && db_all && wgrad_ws,
                 "srbh_rrdbnet_trunk_train_backward: null pointer");
    const long n = (long)B * H * W * 64;
    static const int CH0[5] = {160, 128, 96, 64, 0}, COUT[5] = {32, 32, 32, 32, 64}, CIN[5] = {64, 96, 128, 160, 192};
    static const long DWOFF[5] = {0, 9L * 2048, 9L * (2048 + 3072), 9L * (2048 + 3072 + 4096), 9L * (2048 + 3072 + 4096 + 5120)};
    constexpr long DW_RDB = 9L * 26624;
    static const bool overlap = !(getenv("SRBH_SR_OVERLAP") && getenv("SRBH_SR_OVERLAP")[0] == '0');
    const bool two = overlap && g_stride > 0;
    hipStream_t st = (hipStream_t)stream;
    if (two) { if (int rc0 = side_init()) return rc0; }
    hipStream_t ws_st = two ? g_side.s : st;
    float* gout = g_a;            // gradient of the current RRDB's output
    float* cur = g_b;             // gradient flowing down the RDBs
    float* nxt = g_c;
    int rc;
    int i = num_block * 3;
    int used[2] = {0, 0};
    srbh_conv3x3_args a;
    for (int blk = num_block - 1; blk >= 0; --blk) {
        if ((rc = srbh_axpby_f32(cur, 0.2f, gout, 0.f, nullptr, n, stream))) return rc;        // out = rdb3(.) * 0.2 + x_rrdb
        for (int r = 2; r >= 0; --r) {
            --i;
            const int gb = two ? (i & 1) : 0;
            char* Gi = (char*)G + (size_t)gb * g_stride;
            const char* D = (const char*)dense_all + (size_t)i * dense_stride;
            const char* pk = (const char*)packs + (size_t)i * pack_stride;
            if (two && used[gb]) SRBH_HIP(hipStreamWaitEvent(st, g_side.done[gb], 0));          // the side stream is done reading this G
            if ((rc = srbh_nhwc32_to_act16(cur, Gi, B, 64, H, W, 6, 0, 0.2f, 1, stream))) return rc;        // g5 = 0.2 g (bf16)
            for (int j = 0; j < 4; ++j) {          // g4 .. g1: masked by the saved planes X4 .. X1
                a = srbh_conv3x3_args{};
                a.in = Gi; a.in_chunks_total = 6; a.in_chunk0 = 0; a.in_chunks = 2 + j;
                a.w = pk + pack_off[j]; a.cout = 32; a.B = B; a.H = H; a.W = W;
                a.out16 = Gi; a.out16_chunks_total = 6; a.out16_chunk0 = 2 + j;
                if ((rc = srbh_conv3x3_x16(&a, 1, D, 6, 5 - j, stream))) return rc;
            }
            if (two) {      // G of this RDB is complete: the weight / bias gradients start on the side stream
                SRBH_HIP(hipEventRecord(g_side.ready[gb], st));
                SRBH_HIP(hipStreamWaitEvent(ws_st, g_side.ready[gb], 0));
            }
            if (!two) {
                a = srbh_conv3x3_args{};
                a.in = Gi; a.in_chunks_total = 6; a.in_chunk0 = 0; a.in_chunks = 6;
                a.w = pk + pack_off[4]; a.cout = 64; a.B = B; a.H = H; a.W = W;
                a.skip = cur; a.out32 = nxt; a.out32_c = 64;
                if ((rc = srbh_conv3x3_x16(&a, 1, nullptr, 0, 0, stream))) return rc;
            }
            if ((rc = srbh_act16_channel_sum(Gi, B, H, W, 6, 0, 6, 1, db_all + (long)i * 192, ws_st))) return rc;
            // the five weight gradients of the dense block: each into its own slice of the workspace, their ordered reduces queued and done by
            // ONE pair of launches (round 5: five reduce launches of ~9 us per block sat on the stream that bounds the backward)
            static const bool batch_red = !(getenv("SRBH_SR_BATCH_REDUCE") && getenv("SRBH_SR_BATCH_REDUCE")[0] == '0');
            if (batch_red && (rc = srbh_hwgrad_defer(1))) return rc;
            size_t woff = 0;
            for (int k = 0; k < 5; ++k) {
                rc = srbh_act16_wgrad_b16(D, 6, CIN[k], Gi, 6, CH0[k], COUT[k], B, H, W, dw_all + (long)i * DW_RDB + DWOFF[k], wgrad_ws + woff, ws_st);
                if (rc) { if (batch_red) srbh_hwgrad_flush(ws_st); return rc; }
                if (batch_red) woff += srbh_hwgrad_ws_bytes(COUT[k], CIN[k], 3) / sizeof(float);
            }
            if (batch_red && (rc = srbh_hwgrad_flush(ws_st))) return rc;
            if (two) {
                SRBH_HIP(hipEventRecord(g_side.done[gb], ws_st));
                used[gb] = 1;
                a = srbh_conv3x3_args{};
                a.in = Gi; a.in_chunks_total = 6; a.in_chunk0 = 0; a.in_chunks = 6;
                a.w = pk + pack_off[4]; a.cout = 64; a.B = B; a.H = H; a.W = W;
                a.skip = cur; a.out32 = nxt; a.out32_c = 64;
                if ((rc = srbh_conv3x3_x16(&a, 1, nullptr, 0, 0, stream))) return rc;
            }
            float* t = cur; cur = nxt; nxt = t;
        }
        // the RRDB's skip connection: gradient of the RRDB input = cur + gout; it is the next (lower) RRDB's output gradient
        if ((rc = srbh_axpby_f32(nxt, 1.f, cur, 1.f, gout, n, stream))) return rc;
        float* t = gout; gout = nxt; nxt = t;
    }
    if (two)      // join: everything the side stream wrote (dw_all, db_all) is ordered before what follows on `stream`
        for (int k = 0; k < 2; ++k)
            if (used[k]) SRBH_HIP(hipStreamWaitEvent(st, g_side.done[k], 0));
    *g_out = gout;
    return SRBH_OK;
}

/* The same backward with the 345 data-gradient convs as ONE launch of the persistent trunk kernel (round 5; ptrunk3_kernel<., 1>): the gradient of a
 * dense block is a dense block on gradients, so the launch walks the RDBs in reverse over a ROW of G buffers (G_all + k * g_stride for the k-th RDB
 * from the end; n_rdb + 1 buffers, zero borders) with the saved forward buffers as LeakyReLU masks, and the weight / bias gradients of all RDBs
 * follow on two streams.  The kernel's residual recurrences are the forward's: it runs on x = 0.04 g (g = gradient of the trunk output), where
 *   x' = 0.2 conv5(G) + x         is  0.2 (dx + cur)  with  x = 0.2 cur  (cur = gradient entering the RDB: `a.skip = cur` above), and
 *   x  = 0.2 x + x_rrdb           is  the RRDB's skip  (cur + gout) / 25  with  x_rrdb = gout / 25,
 * so every g5 plane the weight gradients read comes out at its true scale and the result is 25 x the launch's output.  Same operands (bf16, RNE)
 * and the same accumulation order per conv as the per-layer form; the fp32 streams differ from it in the last bit (0.2 applied to conv5's sum,
 * not to the stream).  zero_bias: >= 64 zero floats.  wgrad_ws: TWO workspaces of srbh_rrdbnet_trunk_wgrad_ws_bytes(); trunk_wgrad_ws: srbh_trunk_wgrad_ws_bytes()
 * bytes (the one-launch weight gradients; NULL = the general kernel RDB by RDB through wgrad_ws).  aux: the scratch of the
 * persistent forward.  g_a is read, g_b / g_c are scratch; *g_out = the gradient of the trunk input (one of g_b, g_c).  *used = 0: nothing was
 * launched (geometry not the kernel's, or SRBH_SR_PTRUNK_BWD=0): call srbh_rrdbnet_trunk_train_backward. */
extern "C" int srbh_rrdbnet_trunk_train_backward_persistent(int num_block, const void* dense_all, size_t dense_stride, const void* packs, size_t pack_stride,
                                                            const size_t* pack_off, const float* zero_bias, const float* g_a, float* g_b, float* g_c,
                                                            float** g_out, void* G_all, size_t g_stride, float* dw_all, float* db_all, float* wgrad_ws,
                                                            void* trunk_wgrad_ws, int B, int H, int W, void* aux, void* stream, int* used) {
    SRBH_REQUIRE(num_block > 0 && dense_all && packs && pack_off && zero_bias && g_a && g_b && g_c && g_out && G_all && dw_all && db_all && wgrad_ws && aux && used &&
                 dense_stride > 0 && g_stride > 0, "srbh_rrdbnet_trunk_train_backward_persistent: bad arguments");
    *used = 0;
    static const bool off = getenv("SRBH_SR_PTRUNK_BWD") && getenv("SRBH_SR_PTRUNK_BWD")[0] == '0';
    if (off || srbh_rrdbnet_trunk_train_aux_bytes(B, H, W) == 0) return SRBH_OK;
    const long n = (long)B * H * W * 64;
    const int n_rdb = num_block * 3;
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if ((rc = srbh_axpby_f32(g_b, 0.04f, g_a, 0.f, nullptr, n, stream))) return rc;
    if ((rc = srbh_axpby_f32(g_c, 0.04f, g_a, 0.f, nullptr, n, stream))) return rc;
    if ((rc = srbh_nhwc32_to_act16(g_a, G_all, B, 64, H, W, 6, 0, 0.04f, 1, stream))) return rc;      // g5 of the last RDB (bf16)
    std::vector<srbh_conv_w> cw((size_t)n_rdb * 5);
    for (int k = 0; k < n_rdb; ++k)
        for (int j = 0; j < 5; ++j) {
            cw[(size_t)k * 5 + j] = srbh_conv_w{};
            cw[(size_t)k * 5 + j].w = (const char*)packs + (size_t)(n_rdb - 1 - k) * pack_stride + pack_off[j];
            cw[(size_t)k * 5 + j].bias = zero_bias;
        }
    srbh_rrdbnet_desc dd = {};
    dd.num_block = num_block;
    dd.rdb = cw.data();
    int cur = 0;
    if ((rc = ptrunk_run(&dd, G_all, nullptr, g_b, g_c, B, H, W, aux, st, used, &cur, (long)g_stride,
                         (const char*)dense_all + (size_t)(n_rdb - 1) * dense_stride, -(long)dense_stride))) return rc;
    if (!*used) return SRBH_OK;
    const int* err_word = (const int*)((const char*)aux + ptrunk_err_offset(B, (H + TILE_H - 1) / TILE_H));
    hipLaunchKernelGGL(poison_on_error_kernel, dim3(256), dim3(256), 0, st, err_word, g_b, (size_t)n);
    SRBH_HIP(hipGetLastError());
    if ((rc = srbh_axpby_f32(g_c, 25.f, g_b, 0.f, nullptr, n, stream))) return rc;
    *g_out = g_c;
    // weight / bias gradients of all RDBs: one launch over (RDB, plane pair, tile range) + one reduce (srbh_trunk_wgrad.hip) ...
    static const bool one_launch = !(getenv("SRBH_SR_TRUNK_WGRAD") && getenv("SRBH_SR_TRUNK_WGRAD")[0] == '0');
    if (trunk_wgrad_ws && one_launch) return srbh_trunk_wgrad(num_block, dense_all, dense_stride, G_all, g_stride, B, H, W, dw_all, db_all, trunk_wgrad_ws, stream);
    // ... or (no workspace given / SRBH_SR_TRUNK_WGRAD=0) RDB by RDB with the general kernel, alternating between the caller's stream and the side stream (a weight-gradient launch fills the chip; the
    // small reduces and plane sums of one RDB run beside the next RDB's)
    static const int CH0[5] = {160, 128, 96, 64, 0}, COUT[5] = {32, 32, 32, 32, 64}, CIN[5] = {64, 96, 128, 160, 192};
    static const long DWOFF[5] = {0, 9L * 2048, 9L * (2048 + 3072), 9L * (2048 + 3072 + 4096), 9L * (2048 + 3072 + 4096 + 5120)};
    constexpr long DW_RDB = 9L * 26624;
    static const bool overlap = !(getenv("SRBH_SR_OVERLAP") && getenv("SRBH_SR_OVERLAP")[0] == '0');
    if (overlap) {
        if ((rc = side_init())) return rc;
        SRBH_HIP(hipEventRecord(g_side.ready[0], st));
        SRBH_HIP(hipStreamWaitEvent(g_side.s, g_side.ready[0], 0));
    }
    const size_t ws_floats = srbh_rrdbnet_trunk_wgrad_ws_bytes() / sizeof(float);
    for (int k = 0; k < n_rdb; ++k) {
        const int i = n_rdb - 1 - k;                 // forward index of the RDB whose gradients sit in G buffer k
        const bool on_side = overlap && (k & 1);
        hipStream_t ws_st = on_side ? g_side.s : st;
        float* ws = wgrad_ws + (on_side ? ws_floats : 0);
        const char* Gk = (const char*)G_all + (size_t)k * g_stride;
        const char* D = (const char*)dense_all + (size_t)i * dense_stride;
        if ((rc = srbh_act16_channel_sum(Gk, B, H, W, 6, 0, 6, 1, db_all + (long)i * 192, ws_st))) return rc;
        if ((rc = srbh_hwgrad_defer(1))) return rc;
        size_t woff = 0;
        for (int c = 0; c < 5; ++c) {
            rc = srbh_act16_wgrad_b16(D, 6, CIN[c], Gk, 6, CH0[c], COUT[c], B, H, W, dw_all + (long)i * DW_RDB + DWOFF[c], ws + woff, ws_st);
            if (rc) { srbh_hwgrad_flush(ws_st); return rc; }
            woff += srbh_hwgrad_ws_bytes(COUT[c], CIN[c], 3) / sizeof(float);
        }
        if ((rc = srbh_hwgrad_flush(ws_st))) return rc;
    }
    if (overlap) {
        SRBH_HIP(hipEventRecord(g_side.done[0], g_side.s));
        SRBH_HIP(hipStreamWaitEvent(st, g_side.done[0], 0));
    }
    return SRBH_OK;
}
