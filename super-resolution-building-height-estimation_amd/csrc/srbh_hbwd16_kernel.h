// srbh_hbwd16_kernel.h -- the backward of one 3x3, 16 -> 16 convolution of a BasicBlock behind its BatchNorm, as ONE pass (round 5).
// Included by srbh_head_bwd.hip inside its anonymous namespace (after WG16T / bf16_pair / widen_b4 / NSLOT).
//
// Reference graph (SR/HRfuse.py:142-159 through torch autograd): y = bn(conv(x')).  Given g = dL/dy, the three consumers of
//      dc = coef * (g' - k1 - xhat * k2),   g' = g [masked by the ReLU behind the BatchNorm],  xhat = (c - mean) * invstd
// used to be three launches: bn_bwd_apply (g, c -> dc), hwgrad16 (x', dc -> dW) and hconv16 (dc, W^T -> dx'), i.e. 128 + 96 + 128 bytes
// per pixel with dc written once and read twice.  Here dc is never written: a workgroup stages the 6 x 66 window of dc -- computed from
// g (bf16) and c (fp32) while staging, rounded ONCE to bf16 exactly as bn_bwd_apply's store did -- and the window of x' (fp32 -> the
// forward conv's operand transform -> bf16) and runs BOTH contractions on them:
//      dW[oc][ci][tap] += sum_px dc[px][oc] * x'[px + tap][ci]        (hwgrad16's fragments: channel-major copies, K = pixels)
//      dx'[px][ci]      = sum_tap,oc dc[px - tap][oc] * W[oc][ci][tap] (hconv16's fragments: pixel-major copy, K = channels)
// 32 (g) + 64 (c) + 64 (x) bytes read per pixel, 32 / 64 written: 192 - 256 instead of 352.
// Two uses in a block's backward (hrfuse_autograd._BasicBlockFn.backward):
//   conv2: g = dz (bf16, through the block-closing ReLU), c = c2, no mask; x = c1 with the folded bn1 + ReLU; epilogue = the BatchNorm-
//          backward sums of bn1 over da1 (hconv16's BS form: c1 read at the output pixels) and a bf16 store of da1;
//   conv1 (plain 16-channel block): g = da1 (bf16), c = c1 with bn1's ReLU mask; x = the block input (fp32, no transform); epilogue =
//          + the skip gradient dz (bf16) and an fp32 store of dx.
// Same operand rounding and MFMA order per output element as the separate kernels (the walk, the fragments and the accumulation order
// are theirs): results agree with the three-launch path to the last bits of the fp32 apply arithmetic (tests/test_gpu_hbwd16.py).
// Restrictions (host: srbh_hbwd16_supported): 16 channels everywhere, W % 64 == 0, H % 4 == 0, dense NHWC tensors.
#ifndef HB16_LATE
#define HB16_LATE 0          // (measured: 329 / 342 / 338 us for 0 / 1 / 2, profiles/r05k_time_hbwd16_chain.txt -- the spills are not these registers)
#endif
#ifndef HB16_BITS_VEC
#define HB16_BITS_VEC 1      // 1: the ReLU bit words of a tile come by one vector load ahead of the MFMAs (0: 16 scalar loads in the epilogue)
#endif
struct HB16 {
    static constexpr int QX = WG16T::QX, SX = WG16T::SX, SD = WG16T::SD;      // 18 quads per row, channel strides of the two channel-major copies
    static constexpr int ROWS = 6, COLS = 66;
    static constexpr int PM_DW = ROWS * COLS * 8;                             // pixel-major dc window: 32 bytes per pixel
    static constexpr int STAGE_DW = 16 * SX + 16 * SD + PM_DW;                // 9 440 dwords
    static constexpr int LDS_B = 2 * STAGE_DW * 4 + 13 * 16 * 4 + 9 * 64 * 8;  // 80 960 bytes (+ per-channel constants + weights): two workgroups per CU
};

struct HBParams {
    const void* g;            // bf16 [B][H][W][16]
    const float* c;           // fp32 [B][H][W][16]
    const float* mean; const float* invstd; const float* coef; const float* k1; const float* k2;
    const float* ms; const float* mh;             // ReLU mask of the BatchNorm output (c*ms + mh > 0) or null
    const float* x;           // fp32 [B][H][W][16]
    const float* pre_scale; const float* pre_shift; int pre_relu;
    const void* w;            // bf16 data-gradient pack (srbh_hpack_conv_h16(transpose_flip = 1, bf16 = 1))
    void* dx; int dx_b16;
    const void* res;          // bf16 [B][H][W][16] or null
    const float* bstat_c; const float* bstat_mean; const float* bstat_invstd; const float* bstat_ms; const float* bstat_mh;
    double* stats;
    const unsigned long long* relu_bits;   // BS = 2: activity pattern of the ReLU the output gradient passes next (srbh_bn_add_relu_bits' layout)
    float* ws;
    int B, H, W, tiles_x, tiles_per_img, ntiles, tiles_per_xcd;
};

// BS: 0 = none; 1 = BatchNorm-backward sums of the OUTPUT gradient in the epilogue (conv2's use: the gradient of relu(bn'(bstat_c)));
//     2 = the output gradient (+ res) is the gradient of the PREVIOUS block's output out' = relu(bn2'(c2') + idt'): it is masked with that ReLU's
//         bit pattern, written as bf16 and summed for bn2' (sum dz', sum dz' xhat2' over the rounded values) -- exactly what that block's
//         srbh_bn_bwd_reduce_io(SRBH_BN_REF_BITS | SRBH_BN_OUT_B16) pass would compute from the fp32 tensor this kernel then never writes
// MK: 1 = g is masked with c*ms + mh > 0
// RES: the forms that may carry a skip gradient (conv1's uses: BS 0 and 2; conv2's use, BS 1, has none) ALWAYS load four quads for it -- from `g`
// when no `res` is given, and then do not add them: a run-time `if (p.res) load` would hide the number of loads in flight from the compiler
template <int BS, int MK, int RES = (BS != 1)>
__global__ __launch_bounds__(256, 2) void hbwd16_kernel(const HBParams p) {
    extern __shared__ __attribute__((aligned(16))) float hbsm[];
    using G = HB16;
    constexpr int QX = G::QX, SX = G::SX, SD = G::SD, COLS = G::COLS;
    unsigned* const s_base = (unsigned*)hbsm;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, kk = lane >> 4;
    const int cg = tid & 3;
    const int t_end = min((int)(blockIdx.x & 7) * p.tiles_per_xcd + p.tiles_per_xcd, p.ntiles);
    const int t_first = (blockIdx.x & 7) * p.tiles_per_xcd + (blockIdx.x >> 3), t_step = gridDim.x >> 3;

    // ---- per-channel constants live in LDS (13 rows of 16 floats behind the two stages) and are read where they are used: as 52 registers
    // held over the walk they pushed the kernel 15-52 registers over the 256 of two waves per SIMD (scratch spills)
    float* const cst = hbsm + 2 * G::STAGE_DW;
    enum { C_MEAN = 0, C_INV, C_COEF, C_K1, C_K2, C_MS, C_MH, C_PSC, C_PSH, C_BMEAN, C_BINV, C_BMS, C_BMH, C_ROWS };
    if (tid < C_ROWS * 16) {
        const int row = tid >> 4, ch = tid & 15;
        float v = 0.f;
        switch (row) {
            case C_MEAN: v = p.mean[ch]; break;
            case C_INV: v = p.invstd[ch]; break;
            case C_COEF: v = p.coef[ch]; break;
            case C_K1: v = p.k1[ch]; break;
            case C_K2: v = p.k2[ch]; break;
            case C_MS: v = (MK != 0) ? p.ms[ch] : 0.f; break;
            case C_MH: v = (MK != 0) ? p.mh[ch] : 1.f; break;
            case C_PSC: v = p.pre_scale ? p.pre_scale[ch] : 1.f; break;
            case C_PSH: v = p.pre_scale ? p.pre_shift[ch] : 0.f; break;
            case C_BMEAN: v = (BS != 0) ? p.bstat_mean[ch] : 0.f; break;
            case C_BINV: v = (BS != 0) ? p.bstat_invstd[ch] : 0.f; break;
            case C_BMS: v = (BS == 1 && p.bstat_ms) ? p.bstat_ms[ch] : 0.f; break;
            default: v = (BS == 1 && p.bstat_ms) ? p.bstat_mh[ch] : 1.f; break;       // (no mask given: every element passes, 1 > 0)
        }
        cst[row * 16 + ch] = v;
    }
    auto cq = [&](const int row, const int grp) { return *(const floatx4*)(cst + row * 16 + grp * 4); };
    const bool pre_relu = p.pre_relu != 0;
    // data-gradient weights: 9 taps x 8 bytes per lane, staged once per workgroup behind the constants and read per tap (as 18 registers
    // over the walk they were the rest of the spills)
    short4w* const wlds = (short4w*)(cst + C_ROWS * 16);
    for (int u = tid; u < 9 * 64; u += 256) wlds[u] = ((const short4w*)p.w)[u];
    // staging items (tile independent): item it = (window row xr, quad xq): image columns X0 - 4 + 4 xq .. + 3 of row Y0 - 1 + xr
    constexpr int NIX = 2;                            // 6 * 18 * 4 = 432 items: the second iteration is partial
    int xoff[NIX], xlds[NIX], pmlds[NIX], dlds[NIX];
    int xr[NIX], xq[NIX];
#pragma unroll
    for (int it = 0; it < NIX; ++it) {
        const int q = (tid + it * 256) >> 2;
        xr[it] = q / QX;
        xq[it] = q - xr[it] * QX;
        xoff[it] = (xr[it] * p.W + xq[it] * 4) * 16 + cg * 4;              // elements from the window origin (Y0 - 1, X0 - 4)
        xlds[it] = cg * 4 * SX + q * 2;                                    // channel-major x: + j * SX
        // channel-major dc (interior rows 1..4, quads 1..16 only): the 4 x 64 tile as hwgrad16 stages dY
        const bool inner = xr[it] >= 1 && xr[it] <= 4 && xq[it] >= 1 && xq[it] <= 16;
        dlds[it] = inner ? 16 * SX + cg * 4 * SD + ((xr[it] - 1) * 16 + (xq[it] - 1)) * 2 : -1;
        // pixel-major dc: window column of the quad's pixel i = 4 xq + i - 3 (0..65 are staged)
        pmlds[it] = 16 * SX + 16 * SD;                                     // dword offset of the pixel-major area (pixel offsets added per i)
    }
    const bool x1_valid = tid + 256 < 6 * QX * 4;
    const int xsafe = (p.W + 4) * 16 + cg * 4;        // element offset of the tile's first own quad (row Y0, columns X0 .. X0 + 3) from the window origin

    floatx4 accw[9];
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) accw[tp] = floatx4{0.f, 0.f, 0.f, 0.f};
    float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};

    float2w lg[NIX][4];       // g: raw bf16 quads
    floatx4 lc[NIX][4], lx[NIX][4];
    unsigned okx = 0;
    auto issue = [&](const int t) {
        const int img = t / p.tiles_per_img;
        const int trem = t - img * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const int Y0 = ty * 4, X0 = tx * 64;
        const long org = (((long)img * p.H + (Y0 - 1)) * p.W + (X0 - 4)) * 16;
        const char* gp = (const char*)p.g + org * 2;
        const float* cp = p.c + org;
        const float* xp = p.x + org;
        okx = 0;
#pragma unroll
        for (int it = 0; it < NIX; ++it) {
            bool ok = (unsigned)(Y0 - 1 + xr[it]) < (unsigned)p.H;
            if (xq[it] == 0) ok = ok && X0 > 0;
            if (xq[it] == QX - 1) ok = ok && X0 + 64 < p.W;
            if (it == 1) ok = ok && x1_valid;
            // UNCONDITIONAL loads (an item outside the image reads the tile's own first quad; `commit` zeroes it): a load behind a branch hides the
            // number of memory operations in flight from the compiler, and every later counted wait -- the epilogue's wait for THIS tile's c / skip
            // gradient, issued in front of the next tile's 24 window loads -- becomes s_waitcnt vmcnt(0), i.e. a wait for that prefetch
            // (csrc/srbh_hblock16_kernel.h has the measurement of what that costs)
            const int xo = ok ? xoff[it] : xsafe;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lg[it][i] = *(const float2w*)(gp + (long)(xo + i * 16) * 2);
                lc[it][i] = *(const floatx4*)(cp + xo + i * 16);
                lx[it][i] = *(const floatx4*)(xp + xo + i * 16);
            }
            okx |= ok ? 1u << it : 0u;
        }
    };
    auto commit = [&](unsigned* stage) {
#pragma unroll
        for (int it = 0; it < NIX; ++it) {
            if (it == 0 || x1_valid) {
                const bool ok = (okx >> it) & 1u;
                const floatx4 mn = cq(C_MEAN, cg), is = cq(C_INV, cg), cf = cq(C_COEF, cg), a1 = cq(C_K1, cg), a2 = cq(C_K2, cg);
                const floatx4 psc = cq(C_PSC, cg), psh = cq(C_PSH, cg);
                floatx4 dc[4], xv[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    floatx4 dy = widen_b4(lg[it][i]);
                    const floatx4 cv = lc[it][i];
                    if constexpr (MK != 0) {
                        const floatx4 msv = cq(C_MS, cg), mhv = cq(C_MH, cg);
#pragma unroll
                        for (int j = 0; j < 4; ++j) dy[j] = cv[j] * msv[j] + mhv[j] > 0.f ? dy[j] : 0.f;
                    }
                    floatx4 r = cf * (dy - a1 - ((cv - mn) * is) * a2);       // (bn_bwd_apply4_kernel's expression)
                    floatx4 a = lx[it][i] * psc + psh;
                    if (pre_relu) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) a[j] = fmaxf(a[j], 0.f);
                    }
                    if (!ok) { r = floatx4{0.f, 0.f, 0.f, 0.f}; a = r; }      // outside the image: the convs' zero padding
                    dc[i] = r;
                    xv[i] = a;
                }
                // x', channel-major (hwgrad16's B operand)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    *(uint2w*)(stage + xlds[it] + j * SX) = uint2w{bf16_pair(xv[0][j], xv[1][j]), bf16_pair(xv[2][j], xv[3][j])};
                // dc, channel-major (hwgrad16's A operand): the tile's own pixels only
                if (dlds[it] >= 0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        *(uint2w*)(stage + dlds[it] + j * SD) = uint2w{bf16_pair(dc[0][j], dc[1][j]), bf16_pair(dc[2][j], dc[3][j])};
                }
                // dc, pixel-major (hconv16's B operand): window columns 0..65 = image columns X0 - 1 .. X0 + 64
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int col = xq[it] * 4 + i - 3;
                    if (col >= 0 && col < COLS) {
                        const int off = (xr[it] * COLS + col) * 8 + ((cg ^ ((col >> 2) & 2)) << 1);          // dwords (h16_off / 4)
                        *(uint2w*)(stage + pmlds[it] + off) = uint2w{bf16_pair(dc[i][0], dc[i][1]), bf16_pair(dc[i][2], dc[i][3])};
                    }
                }
            }
        }
    };

    if (t_first < t_end) issue(t_first);
    __syncthreads();           // constants and weights are in LDS (commit() of the first tile reads the constants)
    int buf = 0;
    const int abase = 16 * SX + l15 * SD + (wave * 16 + kk) * 2;            // wgrad A (dc) fragment: + g*8 dwords per 16-pixel group
    const int bbase = l15 * SX + (wave * QX + 1 + kk) * 2;                   // wgrad B (x') fragment: + dy*QX*2 + g*8
    int pbase[3];                                                            // dgrad B (dc, pixel-major) fragment per dx, in BYTES
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
        pbase[dx] = (16 * SX + 16 * SD) * 4 + (wave * COLS + dx + l15) * 32 + ((kk ^ ((((dx + l15) >> 3) & 1) << 1)) << 3);
    for (int t = t_first; t < t_end; t += t_step, buf ^= 1) {
        unsigned* const stage = s_base + buf * G::STAGE_DW;
        commit(stage);
        const int img = t / p.tiles_per_img;
        const int trem = t - img * p.tiles_per_img;
        const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const long pix0 = ((long)img * p.H + ty * 4 + wave) * p.W + tx * 64 + l15;
        // epilogue operands of THIS tile first (c of the statistics / the skip gradient), then the next tile's window
        floatx4 rres[BS != 0 ? 4 : 1];
        float2w rraw[RES ? 4 : 1];                // (the skip gradient: raw bf16 quads, widened in the epilogue)
        // HB16_LATE (BS = 2 only, where both are needed: 24 registers across the MFMAs cost 16 spills): 1 = the skip gradient, 2 = the skip
        // gradient and c' are requested BEHIND the MFMAs (their latency is then covered by the other wave of the SIMD only)
        auto load_c = [&]() {
            const float* rp = p.bstat_c + pix0 * 16 + kk * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) rres[i] = *(const floatx4*)(rp + i * 16 * 16);
        };
        auto load_res = [&]() {
            const char* rp = (const char*)(p.res ? p.res : p.g) + (pix0 * 16 + kk * 4) * 2;
#pragma unroll
            for (int i = 0; i < 4; ++i) rraw[i] = *(const float2w*)(rp + (long)i * 16 * 16 * 2);
        };
        constexpr int LATE = BS == 2 ? HB16_LATE : 0;
        // BS = 2: the ReLU bit words of this wave's 4 x 16 pixels are 16 consecutive 64-bit words (128 bytes): ONE dword per lane, requested
        // here with the other epilogue operands and handed out by v_readlane in the epilogue.  (As 16 scalar loads IN the epilogue, each
        // with its own wait, they were most of the 130 us this form took longer than the plain one: 348 vs 214 us.)
        unsigned bitsv = 0;
        if constexpr (BS == 2 && HB16_BITS_VEC) bitsv = ((const unsigned*)(p.relu_bits + ((pix0 - l15) >> 4) * 4))[lane & 31];
        if constexpr (BS != 0 && LATE < 2) load_c();
        if constexpr (RES && LATE < 1) load_res();
        issue(t + t_step < t_end ? t + t_step : t);          // (ALWAYS: behind the range's end the tile is loaded again and never used -- a constant count)
        __syncthreads();           // stage `buf` complete; every wave is past the MFMAs of the tile before (other stage)
        // ---- weight gradient: 4 pixel groups x 3 rows x 3 shifts (hwgrad16_kernel's loop)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint2w a2w = *(const uint2w*)(stage + abase + g * 8);
            const short4w a = __builtin_bit_cast(short4w, a2w);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const unsigned* rp = stage + bbase + dy * QX * 2 + g * 8;
                const uint2w cur = *(const uint2w*)rp;
                const unsigned pv = rp[-1], nx = rp[2];
                const unsigned mid = __builtin_amdgcn_alignbit(cur[1], cur[0], 16);
                const uint2w b0 = {__builtin_amdgcn_alignbit(cur[0], pv, 16), mid};
                const uint2w b2 = {mid, __builtin_amdgcn_alignbit(nx, cur[1], 16)};
                accw[dy * 3 + 0] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4w, b0), accw[dy * 3 + 0], 0, 0, 0);
                accw[dy * 3 + 1] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4w, cur), accw[dy * 3 + 1], 0, 0, 0);
                accw[dy * 3 + 2] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, __builtin_bit_cast(short4w, b2), accw[dy * 3 + 2], 0, 0, 0);
            }
        }
        // ---- data gradient: 9 taps x 4 pixel groups (hconv16_kernel's loop, bf16 operands)
        floatx4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = floatx4{0.f, 0.f, 0.f, 0.f};
        const char* const sb = (const char*)stage;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap - dy * 3;
            const short4w wtap = wlds[tap * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const short4w b = *(const short4w*)(sb + pbase[dx] + (dy * COLS + i * 16) * 32);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wtap, b, acc[i], 0, 0, 0);
            }
        }
        // ---- epilogue of the data gradient
        if constexpr (BS != 0 && LATE >= 2) load_c();
        if constexpr (RES && LATE >= 1) load_res();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            floatx4 v = acc[i];
            if constexpr (BS == 1) {
                const floatx4 b_ms = cq(C_BMS, kk), b_mh = cq(C_BMH, kk), b_mean = cq(C_BMEAN, kk), b_inv = cq(C_BINV, kk);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float c = rres[i][q];
                    const float dz = fmaf(c, b_ms[q], b_mh[q]) > 0.f ? v[q] : 0.f;
                    ssum[q] += dz;
                    ssq[q] = fmaf(dz, (c - b_mean[q]) * b_inv[q], ssq[q]);
                }
            } else if constexpr (BS == 2) {
                if (p.res) v = v + widen_b4(rraw[i]);
                // the 64 4-channel groups of this wave's 16 pixels share four 64-bit words (wave-uniform address: scalar loads)
                const int sh = l15 * 4 + kk;
                if constexpr (HB16_BITS_VEC) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned lo = __builtin_amdgcn_readlane(bitsv, i * 8 + q * 2), hi32 = __builtin_amdgcn_readlane(bitsv, i * 8 + q * 2 + 1);
                        const unsigned w = sh < 32 ? lo : hi32;
                        v[q] = ((w >> (sh & 31)) & 1u) ? v[q] : 0.f;
                    }
                } else {
                    const long grp0 = __builtin_amdgcn_readfirstlane((int)(((pix0 - l15) + i * 16) >> 4));
                    const unsigned long long* mw = p.relu_bits + grp0 * 4;
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = ((mw[q] >> sh) & 1ull) ? v[q] : 0.f;
                }
                const float2w nb = narrow_b4(v);
                *(float2w*)((char*)p.dx + ((pix0 + i * 16) * 16 + kk * 4) * 2) = nb;
                const floatx4 dzr = widen_b4(nb);          // the sums are taken over the values the consumer reads
                const floatx4 b_mean = cq(C_BMEAN, kk), b_inv = cq(C_BINV, kk);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ssum[q] += dzr[q];
                    ssq[q] = fmaf(dzr[q], (rres[i][q] - b_mean[q]) * b_inv[q], ssq[q]);
                }
                continue;
            } else if constexpr (RES) {
                if (p.res) v = v + widen_b4(rraw[i]);
            }
            if (p.dx_b16) *(float2w*)((char*)p.dx + ((pix0 + i * 16) * 16 + kk * 4) * 2) = narrow_b4(v);
            else *(floatx4*)((float*)p.dx + (pix0 + i * 16) * 16 + kk * 4) = v;
        }
    }
    // ---- flush the weight-gradient partials (hwgrad16's layout: D[row = oc = kk*4 + r][col = ci = l15] per tap)
    __syncthreads();
    float* s_red = hbsm;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int r = 0; r < 4; ++r) s_red[((wave * 9 + tp) * 16 + kk * 4 + r) * 16 + l15] = accw[tp][r];
    __syncthreads();
    for (int u = tid; u < 9 * 256; u += 256) {
        const float v = s_red[u] + s_red[9 * 256 + u] + s_red[2 * 9 * 256 + u] + s_red[3 * 9 * 256 + u];
        p.ws[(long)blockIdx.x * (9 * 256) + u] = v;
    }
    if constexpr (BS != 0) {
        __syncthreads();
        float* red = hbsm;                      // [4 waves][2 moments][16 channels]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float a = ssum[q], b = ssq[q];
#pragma unroll
            for (int m = 1; m < 16; m <<= 1) {
                a += __shfl_xor(a, m);
                b += __shfl_xor(b, m);
            }
            if (l15 == 0) {
                red[(wave * 2 + 0) * 16 + kk * 4 + q] = a;
                red[(wave * 2 + 1) * 16 + kk * 4 + q] = b;
            }
        }
        __syncthreads();
        if (tid < 32) {
            const int mom = tid >> 4, oc = tid & 15;
            double v = 0.0;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += (double)red[(w * 2 + mom) * 16 + oc];
            double* slot = p.stats + (long)(blockIdx.x % NSLOT) * 2 * 16;
            atomicAdd(slot + mom * 16 + oc, v);
        }
    }
}
