// srbh_hblock16_kernel.h -- a whole plain BasicBlock of the inference head as ONE pass (round 6; included by srbh_head.hip inside its
// anonymous namespace, after round4 / widen4 / h16_off).
//
// Reference: SR/HRfuse.py:142-159 in eval mode -- out = relu(bn2(conv2(relu(bn1(conv1(x))))) + x), both convs 3x3, 16 -> 16, no bias, the
// BatchNorms folded to per-channel affines (scale, shift).  In the fp16 inference chain (hrfuse.BasicBlock.forward_nhwc) this used to be two
// launches of hconv16_kernel with the fp16 intermediate a1 = relu(bn1(conv1(x))) written and read back: 32 + 32 bytes per pixel for conv1,
// 32 (a1) + 32 (identity) + 32 (out) for conv2 = 160 bytes per pixel and block for kernels that run at 0.35-0.5 of the HBM roofline
// (profiles/r05cd_predict_steady_kernel_stats.txt: 15 launches, 2.8 ms of a 25 ms batch of 128 tiles).  Here a1 never leaves the CU:
//   * a workgroup owns a 4 x 64 output tile and stages the 8 x 68 window of x (fp16 records as they are: no VALU);
//   * phase 1: conv1 on the 6 x 66 window of a1 the tile's conv2 needs (24 sixteen-pixel units + ONE unit gathering the two right-most
//     columns of all six rows: lanes address LDS individually, so 12 scattered pixels make one MFMA group), bn1 + ReLU + fp16 rounding
//     in the epilogue, ZERO outside the image (conv2's padding is the padding of a1, not conv1 of a padded x), written to LDS;
//   * phase 2: conv2 from that window, bn2, + the identity read from the staged x window (it is the tile's own input), ReLU, one store.
// 32 bytes read + 32 (fp16) / 64 (fp32) written per pixel; the halo ring of a1 is recomputed (1.29 x conv1's work).
// Same operand rounding and the same epilogue expressions as the two-launch chain.  The matrix instruction is v_mfma_f32_16x16x32_f16 with TWO
// taps per instruction (K = 2 x 16 channels): the legacy 16x16x16 form issues in the same 8 passes (tools/mfma_rate.hip), and with
// 90-99 of them per tile and wave this kernel kept the matrix pipe 70 % busy -- its bound, not HBM.  Lane group kk = lane >> 4 holds channels
// 8 (kk & 1) .. + 7 of the pair's first (kk < 2) or second (kk >= 2) tap -- a 16-byte half of the pixel record (the record's swizzle exchanges
// its two halves as wholes).  conv1 pairs the taps (dy 0, dy 1) of a dx and leaves dy 2 as a half-empty instruction (zero weights in the upper
// lane groups): the upper groups then simply read one window row further down, so a fragment row is still read from LDS once and used by two
// output rows -- 6 x 6 instead of 6 x 9 instructions; conv2 and the gathered edge unit have no such reuse and pair taps (0,1) (2,3) (4,5) (6,7)
// (8,-): 5 instead of 9.  The sum over a pixel's 144 products is the same set of products in another order: the outputs equal the
// two-launch chain's up to the last fp16 bit on a small fraction of the elements (tests/test_gpu_hblock16.py), no longer bit for bit.
// The walk, the XCD-contiguous tile ranges and the two-tiles-ahead loads are hconv16_kernel's.
// Restrictions (host: srbh_hblock16_supported): 16 channels, fp16 NHWC input, W % 64 == 0, H % 4 == 0.
struct HBlkParams {
    const void* x;                  // fp16 [B][H][W][16]
    const void* w1; const void* w2; // fp16 HWPACK16 (srbh_hpack_conv_h16) of conv1 / conv2
    const float* s1; const float* h1; const float* s2; const float* h2;      // folded bn1 / bn2
    void* out;                      // fp16 or fp32 [B][H][W][16]
    int B, H, W, tiles_x, tiles_per_img, ntiles, tiles_per_xcd;
};

// Two workgroups per CU (256 registers): at three (168) the kernel spills ~39 registers, and scratch reloads count in vmcnt like any load --
// every one of them turns a counted wait into a wait for the prefetch.
template <int O16>
__global__ __launch_bounds__(256, 2) void hblock16_kernel(const HBlkParams p) {
    constexpr int XR = 8, XC = 68, AR = 6, AC = 66;
    constexpr int NIT = (XR * XC * 4 + 255) / 256;                  // 9 staging units per thread (8-byte quads; the last one partial)
    constexpr int XSTAGE_B = XR * XC * 32;                          // 17 408 bytes
    constexpr int A_B = AR * AC * 32;                               // 12 672 bytes
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    char* const s_x = (char*)hsm;                                   // two x stages
    char* const s_a = s_x + 2 * XSTAGE_B;                           // the a1 window
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kk = lane >> 4;
    const int cg = tid & 3;
    const int t_end = min((int)(blockIdx.x & 7) * p.tiles_per_xcd + p.tiles_per_xcd, p.ntiles);
    const int t_first = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3), t_step = gridDim.x >> 3;

    // ---- per-thread constants
    // A operands: lane (oc = l15, kk) holds input channels 8 (kk & 1) .. + 7 of tap tA (kk < 2) / tB (kk >= 2, zero when the pair has no
    // second tap), gathered from the one-tap-per-instruction pack (lane (q << 4 | oc) of a tap = input channels 4 q .. 4 q + 3)
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    typedef short short8v __attribute__((ext_vector_type(8)));
    const int hi = kk >> 1, oct = kk & 1;
    auto pair_w = [&](const void* w, const int tA, const int tB) {
        const short4v* q = (const short4v*)w;
        const int tap = hi ? tB : tA;
        short8v r = {0, 0, 0, 0, 0, 0, 0, 0};
        if (tap >= 0) {
            const short4v lo4 = q[tap * 64 + ((2 * oct) << 4 | l15)], hi4 = q[tap * 64 + ((2 * oct + 1) << 4 | l15)];
            r = short8v{lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
        }
        return r;
    };
    short8v wp1[3], ws1[3], we1[5], wa2[5];       // conv1: (dy 0, dy 1) pairs and dy 2 singles per dx; the edge unit's and conv2's five pairs
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) { wp1[dx] = pair_w(p.w1, dx, 3 + dx); ws1[dx] = pair_w(p.w1, 6 + dx, -1); }
#pragma unroll
    for (int q = 0; q < 5; ++q) { we1[q] = pair_w(p.w1, 2 * q, q < 4 ? 2 * q + 1 : -1); wa2[q] = pair_w(p.w2, 2 * q, q < 4 ? 2 * q + 1 : -1); }
    const floatx4 sc1 = *(const floatx4*)(p.s1 + kk * 4), sh1 = *(const floatx4*)(p.h1 + kk * 4);
    const floatx4 sc2 = *(const floatx4*)(p.s2 + kk * 4), sh2 = *(const floatx4*)(p.h2 + kk * 4);
    // staging unit `it`: window pixel (tid >> 2) + 64 it = (row, col) of the 8 x 68 window, quad cg; tile-independent
    int uoff[NIT], ulds[NIT];
    unsigned urow = 0, uleft = 0, uright = 0;       // 3 bits per unit: window row; bit it: window column < 2 / >= 66 (outside the image at the image's edge)
    {
        int r = 0, col = tid >> 2;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            uoff[it] = ((r * p.W + col) * 16 + cg * 4) * 2;                    // bytes from the window origin (Y0 - 2, X0 - 2)
            ulds[it] = h16_off(r, col, cg, XC);
            urow |= (unsigned)r << (3 * it);
            if (col < 2) uleft |= 1u << it;
            if (col >= XC - 2) uright |= 1u << it;
            const bool wrapped = col + 64 >= XC;
            col += wrapped ? 64 - XC : 64;
            r += wrapped ? 1 : 0;
        }
    }
    const bool last_unit = tid + (NIT - 1) * 256 < XR * XC * 4;
    // 16-byte half `o` of the record of window pixel (r, col): the swizzle exchanges the halves with bit 3 of the column
    auto off16 = [](const int r, const int col, const int o, const int cols) { return (r * cols + col) * 32 + ((o ^ ((col >> 3) & 1)) << 4); };
    // phase 1 (conv1): wave w computes a1 columns 16 w .. 16 w + 15 of all six rows.  Fragment (xr, dx): lane groups kk < 2 read x window pixel
    // (xr, 16 w + l15 + dx), groups kk >= 2 the pixel one row below (b1); the last window row has no row below: every group reads it (b1u;
    // it only feeds the half-empty dy 2 instructions, whose upper weights are zero -- but zero times whatever LDS holds behind the stage is not)
    int b1[3], b1u[3];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
        b1[dx] = off16(hi, wave * 16 + l15 + dx, oct, XC);
        b1u[dx] = off16(0, wave * 16 + l15 + dx, oct, XC);
    }
    // the edge unit (wave 3): lane l15 < 12 holds a1 (row l15 >> 1, column 64 + (l15 & 1)); the other lanes repeat lane 0's addresses
    const int e_row = l15 < 12 ? (l15 >> 1) : 0, e_col = l15 < 12 ? 64 + (l15 & 1) : 64;
    int be[5], b2[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int tap = (hi && q < 4) ? 2 * q + 1 : 2 * q, dy = tap / 3, dx = tap - dy * 3;      // (the single tap 8: both halves read its pixel)
        be[q] = off16(e_row + dy, e_col + dx, oct, XC);
        // phase 2 (conv2): wave w = output row w; column group i adds 16 i columns (the swizzle bit does not change with 16 columns)
        b2[q] = off16(wave + dy, l15 + dx, oct, AC);
    }
    // a1 stores of phase 1 and the identity reads of phase 2 (swizzle bit 3 of the column: constant over 16 i for the identity, per unit below)
    const int a_st = h16_off(0, wave * 16 + l15, kk, AC);                      // + r * AC * 32
    const int a_se = h16_off(e_row, e_col, kk, AC);
    const int idt = h16_off(wave + 2, 2 + l15, kk, XC);                        // + i * 16 * 32 (16 columns: the swizzle bit flips with bit 3 of the column)

    // The loads of a tile are UNCONDITIONAL (a unit outside the image reads the tile's own first pixel and is zeroed when it goes to LDS): a
    // load behind a branch makes the number of memory operations in flight unknown to the compiler, and every wait for the OLDER slot then
    // becomes s_waitcnt vmcnt(0) -- i.e. a wait for the prefetch issued a moment ago as well (the first version of this kernel ran at the
    // speed of the two launches it replaces for exactly that reason).
    typedef float2v ldv_t;
    ldv_t ld[2][NIT];
    unsigned okmask[2] = {0u, 0u};
    auto issue = [&](auto slot_tag, const int t) {
        constexpr int SL = decltype(slot_tag)::value;
        const int img = t / p.tiles_per_img;
        const int trem = t - img * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const int Y0 = ty * 4, X0 = tx * 64;
        const long org = (((long)img * p.H + (Y0 - 2)) * p.W + (X0 - 2)) * 32;
        const char* tp = (const char*)p.x + org;
        const int safe = (2 * p.W + 2) * 32 + cg * 8;                              // the tile's pixel (Y0, X0): always inside the image
        unsigned m = 0;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int r = (urow >> (3 * it)) & 7;
            bool ok = (unsigned)(Y0 - 2 + r) < (unsigned)p.H;
            if ((uleft >> it) & 1) ok = ok && X0 > 0;
            if ((uright >> it) & 1) ok = ok && X0 + 64 < p.W;
            if (it == NIT - 1) ok = ok && last_unit;
            ld[SL][it] = *(const ldv_t*)(tp + (ok ? uoff[it] : safe));
            m |= ok ? 1u << it : 0u;
        }
        okmask[SL] = m;
    };
    auto commit = [&](auto slot_tag, char* stage) {
        constexpr int SL = decltype(slot_tag)::value;
#pragma unroll
        for (int it = 0; it < NIT; ++it)
            if (it < NIT - 1 || last_unit)
                *(float2v*)(stage + ulds[it]) = ((okmask[SL] >> it) & 1) ? ld[SL][it] : float2v{0.f, 0.f};      // (zero padding = zero bits)
    };
    using SL0 = std::integral_constant<int, 0>;
    using SL1 = std::integral_constant<int, 1>;
    // everything loaded once (weights, folded BatchNorms) is CONSUMED here, in front of the walk: a value whose first use sits inside the
    // loop makes the compiler wait with vmcnt(0) at that use in EVERY iteration (it cannot order the pre-loop load against the loop's
    // prefetches), which again waits for the prefetch issued a moment ago
    asm volatile("" ::"v"(sc1), "v"(sh1), "v"(sc2), "v"(sh2));
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) asm volatile("" ::"v"(wp1[dx]), "v"(ws1[dx]));
#pragma unroll
    for (int q = 0; q < 5; ++q) asm volatile("" ::"v"(we1[q]), "v"(wa2[q]));
    if (t_first >= t_end) return;
    issue(SL0{}, t_first);
    issue(SL1{}, t_first + t_step < t_end ? t_first + t_step : t_first);

    auto mfma = [&](const short8v a, const short8v b, floatx4& acc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), __builtin_bit_cast(half8, b), acc, 0, 0, 0);
    };
    // bn1 + ReLU + fp16 of one a1 unit (hconv16_kernel's epilogue expressions: post_scale, post_relu, out16), zero outside the image
    auto a1_store = [&](const floatx4 acc, const bool inside, const int off) {
        floatx4 v = acc;
        v = v * sc1 + sh1;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
        const float t4[4] = {v[0], v[1], v[2], v[3]};
        short4v h = round4<1>(t4);
        if (!inside) h = short4v{0, 0, 0, 0};
        *(short4v*)(s_a + off) = h;
    };
    auto tile = [&](auto slot_tag, const int t, const int buf) {
        char* const sx = s_x + buf * XSTAGE_B;
        commit(slot_tag, sx);
        const int img = t / p.tiles_per_img;
        const int trem = t - img * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const int Y0 = ty * 4, X0 = tx * 64;
        // (into the registers `commit` has just emptied; ALWAYS issued -- behind the range's end the tile is loaded again and never used -- so that
        //  the number of memory operations between a slot's loads and their use is a compile-time constant: see `issue`)
        issue(slot_tag, t + 2 * t_step < t_end ? t + 2 * t_step : t);
        __syncthreads();           // the x stage is complete; every wave is past phase 2 of the tile before (it read the a1 window and the other x stage)
        // ---- phase 1: a1 = relu(bn1(conv1(x))) on the 6 x 66 window
        {
            floatx4 acc[AR];
#pragma unroll
            for (int r = 0; r < AR; ++r) acc[r] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int xr = 0; xr < XR; ++xr) {          // fragment row xr = x window rows (xr | xr + 1): taps (dy 0 | dy 1) of a1 row xr, tap dy 2 of a1 row xr - 2
                short8v b[3];
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) b[dx] = *(const short8v*)(sx + (xr < XR - 1 ? b1[dx] : b1u[dx]) + xr * XC * 32);
                if (xr < AR) {
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) mfma(wp1[dx], b[dx], acc[xr]);
                }
                if (xr >= 2) {
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) mfma(ws1[dx], b[dx], acc[xr - 2]);
                }
            }
            const int Xc = X0 - 1 + wave * 16 + l15;          // image column of this lane's a1 pixel
            const bool col_in = (unsigned)Xc < (unsigned)p.W;
#pragma unroll
            for (int r = 0; r < AR; ++r) a1_store(acc[r], col_in && (unsigned)(Y0 - 1 + r) < (unsigned)p.H, a_st + r * AC * 32);
            if (wave == 3) {       // the two right-most columns of the six rows as one gathered unit
                floatx4 ae = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 5; ++q) mfma(we1[q], *(const short8v*)(sx + be[q]), ae);
                if (l15 < 12)
                    a1_store(ae, (unsigned)(X0 - 1 + e_col) < (unsigned)p.W && (unsigned)(Y0 - 1 + e_row) < (unsigned)p.H, a_se);
            }
        }
        __syncthreads();           // the a1 window is complete
        // ---- phase 2: out = relu(bn2(conv2(a1)) + x)
        {
            floatx4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 5; ++q) {
#pragma unroll
                for (int i = 0; i < 4; ++i) mfma(wa2[q], *(const short8v*)(s_a + b2[q] + i * 16 * 32), acc[i]);
            }
            const long pix0 = ((long)img * p.H + Y0 + wave) * p.W + X0 + l15;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                floatx4 v = acc[i];
                v = v * sc2 + sh2;
                const float2v raw = *(const float2v*)(sx + idt + i * 16 * 32);
                v = v * 1.0f + widen4<1>(raw);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
                if constexpr (O16) {
                    const float t4[4] = {v[0], v[1], v[2], v[3]};
                    *(short4v*)((char*)p.out + ((pix0 + i * 16) * 16 + kk * 4) * 2) = round4<1>(t4);
                } else {
                    *(floatx4*)((float*)p.out + (pix0 + i * 16) * 16 + kk * 4) = v;
                }
            }
        }
    };
    for (int t = t_first; t < t_end;) {      // two tiles per trip: LDS stage and register slot are compile-time constants
        tile(SL0{}, t, 0);
        t += t_step;
        if (t >= t_end) break;
        tile(SL1{}, t, 1);
        t += t_step;
    }
}
