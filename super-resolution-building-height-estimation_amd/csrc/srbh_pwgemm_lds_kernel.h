// srbh_pwgemm_lds_kernel.h -- the 1x1 convolutions of the MBConv blocks as an LDS-tiled fp32-MFMA GEMM (round 6), included by srbh_pwconv.hip.
//
// pw_gemm_kernel (srbh_pwconv.hip) was sized for the training step's batch of 64: N = B HW of 256 ... 16 384 columns, so 16x16 tiles straight
// from global memory and as many waves as possible.  The tiled prediction runs 256 tiles per batch (N = 1 024 ... 262 144): there the same
// kernel re-reads its operands 16-64 times through L1 / L2 and sits at 15-30 TF/s of the 157 TF/s fp32 matrix rate
// (profiles/r06u_predict_part_kernels.txt: 1.9 ms of the 4.9 ms the encoder / decoders take per batch).  Here a workgroup stages a
// (32 WM TM) x (32 WN TN) tile's operands through LDS, 16 WK values of K at a time, double-buffered (global -> registers while the matrix
// cores work on the other stage -> LDS), and its four waves run v_mfma_f32_32x32x2_f32 on them: lane l feeds A[m = l & 31][k = l >> 5] and
// B[k = l >> 5][n = l & 31] (one ds_read_b32 each, conflict-free along m / n).  Exact fp32 (the instruction is an fmaf chain over k), K walked
// in order.
//   forms: WM x WN x WK waves -- 2x2x1 (64x64 tile, 64x128 with TN = 2): the expand convs, M large and K small; 1x4x1 (32x128): M <= 32.
//   (WK > 1 -- waves splitting every K block and folding through LDS -- is supported by the template; no such form is instantiated, see pw_lds_plan.)
// Layouts as pw_gemm_kernel: In [B][K][HW], Out [B][M][HW], W [M][K] or, TRANS_A, [K][M]; M, K, HW multiples of 4 (every MBConv width is a
// multiple of 8, planes are 4 ... 1 024 pixels), so every global access is a 16-byte one: a column quad n .. n + 3 lies in one image.  Planes of
// 4 pixels walk K fastest in the operand loads and M fastest in the stores (a plane is one 16-byte item; consecutive channels are contiguous).
// The epilogue goes through LDS so that the stores are 16-byte items as well: acc -> Cs[wk][m][n] -> (sum over wk) -> scale / shift /
// activation / skip -> Out.
#pragma once

template <int WM, int WN, int WK, int TM, int TN, int TRANS_A, int EPI>
__global__ __launch_bounds__(64 * WM * WN * WK) void pw_gemm_lds_kernel(const float* __restrict__ W, const float* __restrict__ In, float* __restrict__ Out,
                                                          int M, int K, int HW, long ncols, int tiles_m, int tiles_n, const PwEpi ep) {
    constexpr int NT = 64 * WM * WN * WK;                          // four waves, eight in the deep split-K form
    typedef float floatx16 __attribute__((ext_vector_type(16)));
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN, BK = 16 * WK;
    constexpr int PA = BM + 4, PB = BN + 4, PC = BN + 4;           // row pitches in floats (multiples of 4: 16-byte rows)
    constexpr int STAGE = BK * (PA + PB);
    constexpr int A4 = BK * BM / 4, NA = (A4 + NT - 1) / NT, NB = BK * BN / 4 / NT, O4 = BM * BN / 4, NO = (O4 + NT - 1) / NT;      // 16-byte items (per thread)
    static_assert(BK * BN / 4 % NT == 0, "B tile: whole rounds of loads");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wk = wave / (WM * WN), wm = (wave % (WM * WN)) / WN, wn = wave % WN;
    // consecutive blockIdx values go to consecutive XCDs: give each XCD a contiguous run of tiles (M fastest: the tiles of a run share their
    // columns of In in that XCD's L2)
    const int per_xcd = gridDim.x >> 3;
    const long tile = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= (long)tiles_m * tiles_n) return;
    const int tm = (int)(tile % tiles_m), tn = (int)(tile / tiles_m);
    const bool quad_planes = HW == 4;

    const float* ap[NA];
    int a_lds[NA], a_k[NA];
    bool aok[NA], ast[NA];
#pragma unroll
    for (int j = 0; j < NA; ++j) {
        const int f = (A4 % NT == 0 || t + NT * j < A4) ? t + NT * j : 0;
        ast[j] = A4 % NT == 0 || t + NT * j < A4;              // (the 32x128 form's A tile is 128 items: half of the threads idle)
        if (TRANS_A) {
            const int kk = f / (BM / 4), m4 = f % (BM / 4), m = tm * BM + 4 * m4;
            aok[j] = m < M;
            a_k[j] = kk;
            ap[j] = W + (long)kk * M + (aok[j] ? m : 0);
            a_lds[j] = kk * PA + 4 * m4;
        } else {
            const int ml = f / (BK / 4), k4 = f % (BK / 4), m = tm * BM + ml;
            aok[j] = m < M;
            a_k[j] = 4 * k4;
            ap[j] = W + (long)(aok[j] ? m : 0) * K + 4 * k4;
            a_lds[j] = 4 * k4 * PA + ml;
        }
    }
    const float* bp[NB];
    const float* gp[NB];
    int b_lds[NB], b_k[NB];
    bool bok[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int f = t + NT * j;
        const int kk = quad_planes ? f % BK : f / (BN / 4), c4 = quad_planes ? f / BK : f % (BN / 4);
        const long n = (long)tn * BN + 4 * c4;
        bok[j] = n < ncols;
        const long b = bok[j] ? n / HW : 0;
        const int hw = bok[j] ? (int)(n - b * HW) : 0;
        b_k[j] = kk;
        bp[j] = In + (b * K + kk) * HW + hw;
        gp[j] = (EPI && ep.gate) ? ep.gate + b * K + kk : nullptr;
        b_lds[j] = kk * PB + 4 * c4;
    }
    const long a_step = TRANS_A ? (long)BK * M : BK, b_step = (long)BK * HW;

    floatx4 ra[NA], rb[NB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            ra[j] = (aok[j] && k0 + a_k[j] < K) ? *(const floatx4*)ap[j] : floatx4{0.f, 0.f, 0.f, 0.f};
            ap[j] += a_step;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const bool ok = bok[j] && k0 + b_k[j] < K;
            rb[j] = ok ? *(const floatx4*)bp[j] : floatx4{0.f, 0.f, 0.f, 0.f};
            bp[j] += b_step;
            if (EPI && ep.gate) {
                const float g = ok ? *gp[j] : 0.f;
                rb[j] *= g;
                gp[j] += BK;
            }
        }
    };
    auto commit = [&](int s) {
        float* As = smem + s * STAGE;
        float* Bs = As + BK * PA;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            if (!ast[j]) continue;
            if (TRANS_A) *(floatx4*)(As + a_lds[j]) = ra[j];
            else {
#pragma unroll
                for (int e = 0; e < 4; ++e) As[a_lds[j] + e * PA] = ra[j][e];
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) *(floatx4*)(Bs + b_lds[j]) = rb[j];
    };

    floatx16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nit = (K + BK - 1) / BK;
    fetch(0);
    commit(0);
    __syncthreads();
    const int l32 = lane & 31, kh = lane >> 5;
    for (int it = 0; it < nit; ++it) {
        const bool more = it + 1 < nit;
        if (more) fetch((it + 1) * BK);
        const float* As = smem + (it & 1) * STAGE;
        const float* Bs = As + BK * PA;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            const int kr = wk * 16 + 2 * s + kh;
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[kr * PA + (wm * TM + i) * 32 + l32];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[kr * PB + (wn * TN + j) * 32 + l32];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (more) commit((it + 1) & 1);
        __syncthreads();
    }

    // accumulators -> Cs[wk][m][n]  (C/D map of the 32x32 forms: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5))
    float* Cs = smem;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * kh;
                Cs[(wk * BM + (wm * TM + i) * 32 + row) * PC + (wn * TN + j) * 32 + l32] = acc[i][j][r];
            }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < NO; ++j) {
        const int f = t + NT * j;
        if (O4 % NT != 0 && f >= O4) break;
        const int ml = quad_planes ? f % BM : f / (BN / 4), n4 = quad_planes ? f / BM : f % (BN / 4);
        floatx4 v = *(const floatx4*)(Cs + ml * PC + 4 * n4);
#pragma unroll
        for (int w = 1; w < WK; ++w) v += *(const floatx4*)(Cs + (w * BM + ml) * PC + 4 * n4);
        const int m = tm * BM + ml;
        const long n = (long)tn * BN + 4 * n4;
        if (m < M && n < ncols) {
            const long b = n / HW;
            const long o = (b * M + m) * HW + (n - b * HW);
            if (EPI) {
                if (ep.scale) {
                    const float sc = ep.scale[m], sh = ep.shift[m];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc, sh);
                }
                if (ep.act == 1) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v[e] / (1.f + __expf(-v[e]));
                } else if (ep.act == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (ep.res) v += *(const floatx4*)(ep.res + o);
            }
            *(floatx4*)(Out + o) = v;
        }
    }
}

// which form takes a product (0: none -- pw_gemm_kernel keeps it)
struct PwLdsPlan { int form, tiles_m, tiles_n; };      // form 1: 64x64, 2: 64x128, 4: 32x128
inline PwLdsPlan pw_lds_plan(int M, int K, int HW, long ncols) {
    static const int enabled = getenv("SRBH_PW_LDS") ? atoi(getenv("SRBH_PW_LDS")) : 1;              // 0: always pw_gemm_kernel (A/B aid)
    static const long min_cols = getenv("SRBH_PW_LDS_MINN") ? atol(getenv("SRBH_PW_LDS_MINN")) : 1024;
    PwLdsPlan p{0, 0, 0};
    if (!enabled || ncols < min_cols || (M & 3) || (K & 3) || (HW & 3)) return p;
    auto cdiv = [](long a, long b) { return (a + b - 1) / b; };
    if (M <= 32) p = PwLdsPlan{4, 1, (int)cdiv(ncols, 128)};
    else if (cdiv(M, 64) * cdiv(ncols, 128) >= 1024) p = PwLdsPlan{2, (int)cdiv(M, 64), (int)cdiv(ncols, 128)};
    else if (cdiv(M, 64) * cdiv(ncols, 64) >= 384 || K < 128) p = PwLdsPlan{1, (int)cdiv(M, 64), (int)cdiv(ncols, 64)};
    // (else: few tiles and a deep K -- the project convs at 2x2 / 4x4 planes.  A 32x32 form whose eight waves split every 128-wide K block
    //  was built: 20-30 % faster alone (profiles/r06w_time_pwconv.txt), but its 74 KB of LDS per workgroup beside the head's kernels cost the
    //  tiled prediction 1.5-7 % depending on the box (profiles/r06ac_ab_pw_lds_forms.txt); pw_gemm_kernel's 16x16 tiles with split K keep them)
    return p;
}

template <int WM, int WN, int WK, int TM, int TN, int TRANS_A, int EPI>
int pw_lds_launch_form(const PwLdsPlan& p, const float* W, const float* In, float* Out, int M, int K, int HW, long ncols, const PwEpi& ep,
                       hipStream_t st) {
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN, BK = 16 * WK;
    constexpr int stage = BK * (BM + 4 + BN + 4), cs = WK * BM * (BN + 4);
    constexpr size_t lds = sizeof(float) * (2 * stage > cs ? 2 * stage : cs);
    const long tiles = (long)p.tiles_m * p.tiles_n;
    if (lds > 64 * 1024)
        SRBH_ONCE_PER_DEVICE(SRBH_HIP(hipFuncSetAttribute((const void*)pw_gemm_lds_kernel<WM, WN, WK, TM, TN, TRANS_A, EPI>,
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)));
    hipLaunchKernelGGL((pw_gemm_lds_kernel<WM, WN, WK, TM, TN, TRANS_A, EPI>), dim3((unsigned)((tiles + 7) / 8 * 8)), dim3(64 * WM * WN * WK), lds, st, W, In, Out,
                       M, K, HW, ncols, p.tiles_m, p.tiles_n, ep);
    SRBH_HIP(hipGetLastError());
    return SRBH_OK;
}

template <int TRANS_A, int EPI>
int pw_lds_launch(const PwLdsPlan& p, const float* W, const float* In, float* Out, int M, int K, int HW, long ncols, const PwEpi& ep,
                  hipStream_t st) {
    switch (p.form) {
        case 1: return pw_lds_launch_form<2, 2, 1, 1, 1, TRANS_A, EPI>(p, W, In, Out, M, K, HW, ncols, ep, st);
        case 2: return pw_lds_launch_form<2, 2, 1, 1, 2, TRANS_A, EPI>(p, W, In, Out, M, K, HW, ncols, ep, st);
        default: return pw_lds_launch_form<1, 4, 1, 1, 1, TRANS_A, EPI>(p, W, In, Out, M, K, HW, ncols, ep, st);
    }
}
