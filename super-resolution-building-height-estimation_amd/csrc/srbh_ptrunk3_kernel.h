// srbh_ptrunk3_kernel.h -- persistent trunk kernel, variant 3: the protocol and data layout of ptrunk_kernel (one workgroup
// per (image, 8 rows), in-L2 halo exchange, resident plane 0, register-resident RDB stream), re-cut for the instruction
// stream.  Included by srbh_ptrunk.hip inside its anonymous namespace (shares PLayer / PParams / the LDS map).
//
// What the r01 kernel's ISA showed (one wave per SIMD: every instruction that is not in the shadow of an MFMA is wall clock):
//   * a step's ~20 LDS-DMA statements were emitted in clumps of 6-7 *in front of* a group's MFMAs, each clump idling the
//     matrix core for ~150 cycles; 4-6 of them were EXEC-masked no-ops that still cost their issue slots, and every step
//     recomputed its masks through v_cndmask / v_readfirstlane;
//   * ~70 instructions (descriptor arithmetic, mask selects, the first LDS reads, six DMAs) sat between the step barrier and
//     the first MFMA.
// Here the kind of staging a step performs is a COMPILE-TIME property (KIND), so a step contains exactly the DMA statements
// it needs and no mask arithmetic, and the order of its instructions is pinned with scheduling fences:
//   MFMA ; <= 2 ds_read of the next group  |  MFMA ; one DMA statement  -- every non-MFMA instruction sits behind an MFMA.
// The DMA statement carries its own wait states (M0 write -> LDS-DMA; a VALU-written scalar base -> VMEM), so it is
// hazard-proof by construction wherever the compiler puts the registers (tests/test_isa_hazards.py still scans the ISA).
// Restrictions (the host falls back to ptrunk_kernel otherwise): W == 64, H a multiple of 8 (full tiles: no store
// predication in the epilogues).  Same arithmetic in the same order as ptrunk_kernel: bit-identical results.
#pragma once

#ifndef P3_READS_PER_SHADOW
#define P3_READS_PER_SHADOW 2
#endif

template <int DUMMY>
__global__ __launch_bounds__(256, 1) void ptrunk3_kernel(const PParams pp) {
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

    // ---- geometry that is identical for every layer (as ptrunk_kernel)
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
    const bool tail_ok = (G::NJ - 1) * 256 + tid < G::UNITS;
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

    auto lds_addr = [](const char* p) { return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const char*)p; };
    auto uni64 = [](unsigned long long m) {   // make uniformity visible to the compiler ("s" operands must be SGPRs)
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)m), hi32 = __builtin_amdgcn_readfirstlane((unsigned)(m >> 32));
        return ((unsigned long long)hi32 << 32) | lo;
    };
    const unsigned long long tail_mask = uni64(__builtin_amdgcn_ballot_w64(tail_ok));
    const unsigned long long w4_mask = uni64(wave < 2 ? ~0ull : 0ull);   // weight fragments 16, 17 of an 18-fragment chunk

    // ---- the LDS-DMA statement.  `s_nop 4` = 5 wait states: covers "SALU writes M0 -> LDS-DMA" (1) and "VALU writes an
    // SGPR (v_readfirstlane, spill reload) -> VMEM reads it as scalar base" (5) whatever the compiler emitted just before.
    // The statement is placed in the shadow of an MFMA, where these cycles are free.
    auto dma = [&](auto sc1_tag, const unsigned long long base, const unsigned voff, const unsigned lds_off) {
        if constexpr (decltype(sc1_tag)::value)
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2 sc1" ::"s"(lds_off), "v"(voff), "s"(base) : "memory", "m0");
        else
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_off), "v"(voff), "s"(base) : "memory", "m0");
    };
    auto dma_masked = [&](auto sc1_tag, const unsigned long long base, const unsigned voff, const unsigned lds_off, const unsigned long long mask) {
        if constexpr (decltype(sc1_tag)::value)
            asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 exec, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %2, %3 sc1\n\ts_mov_b64 exec, -1"
                         ::"s"(mask), "s"(lds_off), "v"(voff), "s"(base) : "memory", "m0");
        else
            asm volatile("s_mov_b32 m0, %1\n\ts_mov_b64 exec, %0\n\ts_nop 4\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, -1"
                         ::"s"(mask), "s"(lds_off), "v"(voff), "s"(base) : "memory", "m0");
    };
    using SC1 = std::true_type;
    using NOSC = std::false_type;
    constexpr int JORD[11] = {2, 3, 4, 5, 6, 7, 8, 0, 1, 9, 10};   // input DMA order: the two halo rows last
    static_assert(G::NJ == 11, "DMA order is written for 11 input instructions");
    // DMA statement d of a staging job (NIN input statements first, then NW weight statements)
    auto dma_item = [&](auto nin_tag, auto nw_tag, const int d, const unsigned long long ibase, const unsigned long long wbase,
                        const unsigned din_w, const unsigned dw_w) {
        constexpr int NIN = decltype(nin_tag)::value, NW = decltype(nw_tag)::value;
        if (d < NIN) {
            const int j = JORD[d];
            if (j < G::NJ - 1)
                dma(SC1{}, ibase, goff[j], din_w + j * 4096);
            else
                dma_masked(SC1{}, ibase, goff[j], din_w + j * 4096, uni64(tail_mask));   // (re-formed at the point of use: a long-lived 64-bit uniform may be parked in VGPRs)
        } else {
            const int k = d - NIN;
            if (NW == 5 && k == 4)
                dma_masked(NOSC{}, wbase + k * 4096, woff, dw_w + k * 4096, uni64(w4_mask));
            else
                dma(NOSC{}, wbase + k * 4096, woff, dw_w + k * 4096);
        }
    };
    // cold staging (kernel start and RDB seam: nothing could be prefetched)
    auto stage_cold = [&](const char* src, char* din, const char* wsrc, char* dw, auto nw_tag) {
        const unsigned long long ib = uni64((unsigned long long)src), wb = uni64((unsigned long long)(wsrc + wave * 1024));
        const unsigned din_w = __builtin_amdgcn_readfirstlane(lds_addr(din) + wave * 1024);
        const unsigned dw_w = __builtin_amdgcn_readfirstlane(lds_addr(dw) + wave * 1024);
        constexpr int NW = decltype(nw_tag)::value;
#pragma unroll
        for (int d = 0; d < 11 + NW; ++d) dma_item(std::integral_constant<int, 11>{}, nw_tag, d, ib, wb, din_w, dw_w);
    };

    // ---- neighbour progress (as ptrunk_kernel)
    int f_up = up < 0 ? 0x7fffffff : 0, f_dn = dn < 0 ? 0x7fffffff : 0;
    bool aborted = false;
    auto ensure_flags = [&](int need) {
        auto* word = (__attribute__((address_space(3))) int*)(smem + P_WORD_OFF);
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

    // ---- the RDB-level fp32 stream of this wave's 4 rows x 32 pixels x 64 channels: 128 registers per lane, conv5's
    // accumulator layout; loaded once from conv_first's pixel-order output
    floatx4 xres[2][4][4];
    const int X = wc * 32 + l31;
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int Y = Y0 + wr * 4 + i;
            const float* q = pp.xrr + (((long)img * pp.H + Y) * pp.W + X) * 64 + hi * 4;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int g = 0; g < 4; ++g) xres[mb][i][g] = *(const floatx4*)(q + mb * 32 + g * 8);
        }
    }

    // ---- one K step.  CB = cout/32 of the running layer; KIND = what is staged for the NEXT step:
    //   0 nothing | 1 input plane + 18 weight fragments | 2 18 weight fragments | 3 input plane + 36 weight fragments
    auto run_step = [&](auto cb_tag, auto kind_tag, floatx16 (&acc)[decltype(cb_tag)::value][4], const char* sbi, const char* sbw,
                        const char* nsrc, const char* nw, char* dst) {
        constexpr int CB = decltype(cb_tag)::value, KIND = decltype(kind_tag)::value;
        constexpr int NRD = G::NP + 3 * CB, NMF = 12 * CB;
        constexpr int NIN = (KIND == 1 || KIND == 3) ? 11 : 0;
        constexpr int NW = KIND == 0 ? 0 : (KIND == 3 ? 9 : 5);
        constexpr int ND = NIN + NW;
        constexpr int RSH = (NRD + P3_READS_PER_SHADOW - 1) / P3_READS_PER_SHADOW;   // MFMA shadows of a group that carry LDS reads
        constexpr int DPG = NMF - RSH;                                                 // ... that can carry a DMA statement
        static_assert(ND <= 3 * DPG, "the step's DMA statements must fit the first three groups");
        const unsigned long long ibase = uni64((unsigned long long)nsrc);
        const unsigned long long wbase = uni64((unsigned long long)(nw + wave * 1024));
        const unsigned din_w = __builtin_amdgcn_readfirstlane(lds_addr(dst) + wave * 1024);
        const unsigned dw_w = din_w + IN_EX;
        half8 P[2][G::NP];
        half8 A[2][3][CB];
        auto read_item = [&](const int g, const int r, const int set) {   // LDS read r (0..NRD-1) of group g
            const int ks = g / 3, dx = g - ks * 3;
            if (r < G::NP) {
                P[set][r] = *(const half8*)(sbi + aoff[dx][ks] + r * G::ROW_B);
            } else {
                const int q = r - G::NP, dy = q / CB, mb = q - dy * CB;
                A[set][dy][mb] = *(const half8*)(sbw + woff + ((((dy * 3 + dx) * 2 + ks) * CB + mb) << 10));
            }
        };
#pragma unroll
        for (int r = 0; r < NRD; ++r) read_item(0, r, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 6; ++g) {
#pragma unroll
            for (int m = 0; m < NMF; ++m) {
                const int dy = m / (4 * CB), rem = m - dy * 4 * CB, i = rem / CB, mb = rem - i * CB;
                acc[mb][i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[g & 1][dy][mb], P[g & 1][i + dy], acc[mb][i], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (m < RSH) {
                    if (g + 1 < 6) {
#pragma unroll
                        for (int q = 0; q < P3_READS_PER_SHADOW; ++q)
                            if (m * P3_READS_PER_SHADOW + q < NRD) read_item(g + 1, m * P3_READS_PER_SHADOW + q, (g + 1) & 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                } else {
                    const int d = g * DPG + (m - RSH);
                    if (d < ND) {
                        dma_item(std::integral_constant<int, NIN>{}, std::integral_constant<int, NW>{}, d, ibase, wbase, din_w, dw_w);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    };

    using C1 = std::integral_constant<int, 1>;
    using C2 = std::integral_constant<int, 2>;
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    using K3 = std::integral_constant<int, 3>;

    // ---- layer prologue: drain the own DMA / stores, barrier, bias into LDS, lazy publish
    auto prologue = [&](const float* bias, const int nb, const int bias_lds) {
        float bias_v = 0.f;
        if (tid < nb) bias_v = bias[tid];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < nb) ((float*)(smem + bias_lds))[tid] = bias_v;
        if (pending_pub) {
            publish(pub_val);
            pending_pub = false;
        }
    };
    auto step_sync = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's LDS-DMA of the step has landed ...
        __syncthreads();                                   // ... and everybody else's; all waves are past the previous step
    };

    // ---- epilogue of a cout-32 layer: bias, leaky ReLU, fp16, straight from the MFMA D layout (see ptrunk_kernel)
    auto epi32 = [&](floatx16 (&acc)[1][4], char* oplane) {
        floatx4 bias4[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) bias4[g] = *(const floatx4*)((const float*)(smem + A_BIAS_OFF) + g * 8 + hi * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int Y = Y0 + wr * 4 + i;
            unsigned hp[4][2];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                floatx4 w;
#pragma unroll
                for (int q = 0; q < 4; ++q) w[q] = acc[0][i][g * 4 + q];
                w += bias4[g];
                const floatx4 ws = w * 0.2f;
#pragma unroll
                for (int q = 0; q < 4; ++q) asm("v_max_f32 %0, %1, %2" : "=v"(w[q]) : "v"(w[q]), "v"(ws[q]));
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
                auto s1 = __builtin_amdgcn_permlane32_swap(hp[2 * m][1], hp[2 * m + 1][1], false, false);
                typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
                const uintx4 raw = {s0[0], s1[0], s0[1], s1[1]};
                char* o = oplane + (long)(Y + 1) * pp.row_b + (X + 1) * PIX_B + m * 32 + hi * 16;
                if (wt)
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(o), "v"(raw) : "memory");
                else
                    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(o), "v"(raw) : "memory");
            }
        }
    };
    // ---- epilogue of conv5: x = 0.2 (acc + bias) + x in registers (+ the RRDB-level stream every third RDB), fp16 copy out
    auto epi64 = [&](floatx16 (&acc)[2][4], char* obase, const bool r2, const bool r2_pixel) {
        floatx4 bias4[2][4];
#pragma unroll
        for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int g = 0; g < 4; ++g) bias4[mb][g] = *(const floatx4*)((const float*)(smem + B_BIAS_OFF) + mb * 32 + g * 8 + hi * 4);
        const int frag_lane = wc * 2048 + lane * 4;   // fragment order: instruction (mb, g) owns 1 KiB, lane l its 16 B
        // rows go two at a time: the 16 loads of the RRDB-level stream (every third RDB) are all issued before any is consumed
#pragma unroll
        for (int ih = 0; ih < 4; ih += 2) {
            floatx4 a2[2][2][4];
            if (r2) {
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const long rowb = ((long)img * pp.H + Y0 + wr * 4 + ih + k) * pp.W * 64;
                    const float* q2 = pp.xrr + rowb + (r2_pixel ? X * 64 + hi * 4 : frag_lane);
                    const int sm = r2_pixel ? 32 : 1024, sg = r2_pixel ? 8 : 256;
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) a2[k][mb][g] = *(const floatx4*)(q2 + mb * sm + g * sg);
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int i = ih + k;
                const int Y = Y0 + wr * 4 + i;
                const long rowb = ((long)img * pp.H + Y) * pp.W * 64;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) {
                    unsigned hp[4][2];
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        floatx4 tt;
#pragma unroll
                        for (int q = 0; q < 4; ++q) tt[q] = acc[mb][i][g * 4 + q];
                        tt += bias4[mb][g];
                        tt = tt * 0.2f + xres[mb][i][g];
                        if (r2) tt = tt * 0.2f + a2[k][mb][g];
                        xres[mb][i][g] = tt;
                        half4 h4;
#pragma unroll
                        for (int q = 0; q < 4; ++q) h4[q] = (_Float16)tt[q];
                        const uint2 u = __builtin_bit_cast(uint2, h4);
                        hp[g][0] = u.x;
                        hp[g][1] = u.y;
                    }
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        auto s0 = __builtin_amdgcn_permlane32_swap(hp[2 * m][0], hp[2 * m + 1][0], false, false);
                        auto s1 = __builtin_amdgcn_permlane32_swap(hp[2 * m][1], hp[2 * m + 1][1], false, false);
                        typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
                        const uintx4 raw = {s0[0], s1[0], s0[1], s1[1]};
                        char* o = obase + (long)mb * pp.plane_b + (long)(Y + 1) * pp.row_b + (X + 1) * PIX_B + m * 32 + hi * 16;
                        if (wt)
                            asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(o), "v"(raw) : "memory");
                        else
                            asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(o), "v"(raw) : "memory");
                    }
                }
                if (r2) {   // the RRDB-level stream goes back to memory (fragment order): private to this workgroup
                    const float* q = pp.xrr + rowb;
                    const unsigned vo = (unsigned)frag_lane * 4u;
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const unsigned long long sb = uni64((unsigned long long)(q + mb * 1024 + g * 256));
                            asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(vo), "v"(xres[mb][i][g]), "s"(sb) : "memory");
                        }
                }
            }
        }
    };

    // ---- prologue of the launch: conv1 of RDB 0 reads conv_first's output (no flag needed)
    char* dcur = pp.dense[0] + tile_off;    // tile origin of plane 0 in the running RDB's dense buffer
    char* dnxt = pp.dense[1] + tile_off;
    stage_cold(dcur, smem, pp.layers[0].w, smem + stage_off(1, 0) + IN_EX, std::integral_constant<int, 5>{});
    const int nrdb = pp.nlayers / 5;
    for (int rdb = 0; rdb < nrdb && !aborted; ++rdb) {
        const PLayer* T = pp.layers + rdb * 5;
        const int L0 = rdb * 5;
        int gs = 0;
        // ---------------- conv1..conv4 (cout 32, plane 0 resident, stages of IN_EX + 18 KiB)
        for (int k = 0; k < 4 && !aborted; ++k) {
            const int L = L0 + k, n = k + 2;
            const char* wl = T[k].w;
            prologue(T[k].bias, 32, A_BIAS_OFF);
            // conv1's inputs were verified at the seam; conv2..4: the only NEW input plane is the last chunk -> the neighbour
            // flags are checked behind step 0
            floatx16 acc[1][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][i][r] = 0.f;
            // step 0: resident plane, stages chunk 1
            run_step(C1{}, K1{}, acc, smem, smem + stage_off(1, gs & 1) + IN_EX, dcur + (long)pp.plane_b, wl + 18 * 1024,
                     smem + stage_off(1, (gs + 1) & 1));
            ++gs;
            if (k > 0 && L > 0) {
                ensure_flags(L);
                if (aborted) break;
            }
            for (int c = 1; c + 1 < n; ++c) {
                step_sync();
                const char* st = smem + stage_off(1, gs & 1);
                run_step(C1{}, K1{}, acc, st, st + IN_EX, dcur + (long)(c + 1) * pp.plane_b, wl + (long)(c + 1) * (18 * 1024),
                         smem + stage_off(1, (gs + 1) & 1));
                ++gs;
            }
            step_sync();
            {
                const char* st = smem + stage_off(1, gs & 1);
                if (k < 3)        // last step of conv1..3: the next layer's step 0 reads the resident plane: weights only
                    run_step(C1{}, K2{}, acc, st, st + IN_EX, nullptr, T[k + 1].w, smem + stage_off(1, (gs + 1) & 1));
                else              // last step of conv4: conv5's chunk 0 (plane 0) + 36 KiB of weights into the phase-B stage 0
                    run_step(C1{}, K3{}, acc, st, st + IN_EX, dcur, T[4].w, smem + stage_off(2, 0));
                ++gs;
            }
            epi32(acc, dcur + (long)(2 + k) * pp.plane_b - (long)Y0 * pp.row_b);
            pending_pub = true;   // published behind the next top-of-step barrier (whose vmcnt(0) covers these stores)
            pub_val = L + 1;
        }
        if (aborted) break;
        // ---------------- conv5 (cout 64, stages of IN_EX + 36 KiB, residual epilogue)
        {
            const int L = L0 + 4;
            const char* wl = T[4].w;
            prologue(T[4].bias, 64, B_BIAS_OFF);
            floatx16 acc[2][4];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mb][i][r] = 0.f;
            {
                const char* st = smem + stage_off(2, 0);
                run_step(C2{}, K3{}, acc, st, st + IN_EX, dcur + (long)pp.plane_b, wl + 36 * 1024, smem + stage_off(2, 1));
            }
            ensure_flags(L);
            if (aborted) break;
            for (int c = 1; c < 5; ++c) {
                step_sync();
                const char* st = smem + stage_off(2, c & 1);
                run_step(C2{}, K3{}, acc, st, st + IN_EX, dcur + (long)(c + 1) * pp.plane_b, wl + (long)(c + 1) * (36 * 1024),
                         smem + stage_off(2, (c + 1) & 1));
            }
            step_sync();
            {
                const char* st = smem + stage_off(2, 1);
                run_step(C2{}, K0{}, acc, st, st + IN_EX, nullptr, nullptr, smem);
            }
            const bool r2 = (rdb % 3) == 2;
            epi64(acc, dnxt - (long)Y0 * pp.row_b, r2, rdb == 2);
            // RDB seam: the next conv1's first chunk is THIS layer's output on the neighbours: publish now, then wait for them
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            publish(L + 1);
            if (rdb + 1 < nrdb) {
                ensure_flags(L + 1);
                if (aborted) break;
                stage_cold(dnxt, smem, T[5].w, smem + stage_off(1, 0) + IN_EX, std::integral_constant<int, 5>{});
            }
            char* tmp = dcur;
            dcur = dnxt;
            dnxt = tmp;
        }
    }
}
