// 16x16 Cholesky + inverse of the factor, one wave, DPP row broadcasts.  Shared by
// chol_kernels.hip and the stand-alone micro-benchmark tools/f16bench.hip.
#pragma once
#include <hip/hip_runtime.h>

template <int N_> struct ic { static constexpr int value = N_; };

// compile-time loop: f(ic<I>{}) for I in [I0, N)
template <int I0, int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I0 < N) {
        f(ic<I0>{});
        static_for<I0 + 1, N>(f);
    }
}

// value of lane J of each 16-lane DPP row, broadcast to the whole row (v_mov_b32 row_newbcast)
template <int J>
__device__ __forceinline__ double bcast16(double v)
{
    // one v_mov_b64_dpp; old = 0 with bound_ctrl so hipcc does not copy the source first
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + J, 0xf, 0xf, true);
}

// acc += -(value of `src` in lane J of each 16-lane row) * own, in ONE instruction: v_fmac_f64 is a VOP2
// instruction and takes a DPP row broadcast on its first source (gfx90a+ "DP ALU DPP"), where the
// compiler emits v_mov_b64_dpp + v_fma_f64 -- a dependent pair per update, and these updates are most
// of what the single, in-order factor wave issues per pivot.  The rounding is that of
// fma(-bcast(src), own, acc).  NOP1: the hardware does not interlock a DPP read of a VGPR the
// preceding VALU instruction wrote (2 wait states on gfx9); the hazard recogniser cannot see into
// inline assembly, so the first use after `src` was produced carries its own s_nop (the statements are
// volatile: they keep their program order, the later ones of a group stay behind the first).
template <int J, bool NOP1 = false>
__device__ __forceinline__ void fnmac_bcast16(double &acc, double src, double own)
{
    if constexpr (NOP1)
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own), "n"(J));
    else
        asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(own), "n"(J));
}

// Cholesky of a 16x16 SPD tile held in LDS as [row][col] (lower triangle used) and the
// inverse of its factor.  One wave; every lane keeps matrix row lane&15 in registers (the four
// 16-lane DPP rows hold identical copies of the factor), pivots and multipliers travel by DPP
// row broadcast, so everything stays in VGPRs.  The inverse Y = L16^-1 is split over the DPP
// rows by columns -- lane (r, q) carries Y[r][q + 4 i], i = 0..3, which is exactly the MFMA
// A-operand order it is published in -- and its step k (needs only column k of L and 1/L_kk)
// is issued right behind pivot k, in the shadow of the pivot chain's latencies.
// Out: s_d16 = L16 (upper zeroed), s_inv[kg*64 + l] = Linv[l&15][(l>>4) + 4*kg].  Returns 0 or
// 1 + index of the first non-positive pivot.
__device__ __forceinline__ int factor16(double (*s_d16)[17], double *s_inv, int lane)
{
    const int lr = lane & 15, lq = lane >> 4;
    double row[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) row[c] = s_d16[lr][c];
    double acc[4];          // Y[r][lq + 4 i] before the final scaling by 1 / L_rr
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (lq + 4 * i == lr) ? 1.0 : 0.0;
    // The single factor wave issues in order and is ISSUE-bound (~7 cycles per instruction), so what counts
    // per pivot is the number of instructions: 8 on the chain (broadcast, rsq, cubic step, column scaling),
    // 15 - j row updates and j/4 + 1 inverse updates (one v_fmac_f64_dpp each), 4 for the multiplier of the
    // inverse.  Nothing else: the scaled column IS the factor's column (lane j: d * rsqrt(d) = L_jj, <= 1.5 ulp;
    // no Newton-corrected copy, no select to put it in place), rows of lanes <= j are final and what the later
    // updates do to their (upper-triangular, never read) entries does not matter, and 1 / L_rr of a lane's own
    // row is recomputed ONCE at the end from the factor it has just written to LDS.
    static_for<0, 16>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        // lane masks are recomputed per pivot from an opaque copy of the row index: kept live
        // for all 16 pivots they overflow the SGPR file and get spilled to VGPR lanes
        int lrj = lr;
        asm volatile("" : "+v"(lrj));
        const double d = bcast16<j>(row[j]);
        // The 128-pivot chain is the critical path of the whole panel phase.  No pivot test sits
        // on it: a non-positive (or NaN) pivot turns L_jj into NaN and is found from that after
        // the loop.  rsqrt = v_rsq_f64 (~2^-26) + one cubic step (<= 1 ulp), without the library's
        // zero / infinity special cases.
        const double y0 = __builtin_amdgcn_rsq(d);
        const double e = fma(-(y0 * d), y0, 1.0);
        const double ri = fma(y0 * e, fma(e, 0.375, 0.5), y0);
        const double cj = row[j] * ri;   // lanes r > j: L[r][j]; lane j: L_jj
        row[j] = cj;
        static_for<j + 1, 16>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            fnmac_bcast16<c, c == j + 1>(row[c], cj, cj);  // row[c] -= L[c][j] (lives in lane c) * L[r][j]
        });
        // Inverse, step j: row j of Y is final (acc_j / L_jj), is broadcast from lane j of each
        // DPP row, and every lane r > j subtracts L[r][j] * Y[j][:].  Columns past j are still
        // zero in lane j, so the registers i > j/4 need no work.
        const double m = (lrj > j) ? cj * ri : 0.0;  // L[r][j] / L_jj below the pivot, 0 on and above it
        static_for<0, j / 4 + 1>([&](auto ic_) {
            constexpr int i = decltype(ic_)::value;
            fnmac_bcast16<j, i == 0>(acc[i], acc[i], m);  // acc[i] -= (acc[i] of lane j) * m; lane j itself: m = 0
        });
    });
    if (lq == 0) {
#pragma unroll
        for (int c = 0; c < 16; ++c) s_d16[lr][c] = (c <= lr) ? row[c] : 0.0;
    }
    // 1 / L_rr of this lane's own row: L_rr back from LDS (the lq == 0 lane of the same wave wrote it; LDS
    // operations of one wave complete in order), v_rcp_f64 + two Newton steps
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const double lrr = s_d16[lr][lr];
    double own_dinv = __builtin_amdgcn_rcp(lrr);
    own_dinv = fma(own_dinv, fma(-lrr, own_dinv, 1.0), own_dinv);
    own_dinv = fma(own_dinv, fma(-lrr, own_dinv, 1.0), own_dinv);
    // first row whose pivot was not a positive finite number (lanes 0..15 hold rows 0..15)
    const unsigned long long badmask = __ballot(!(lrr > 0.0) || !(lrr < INFINITY)) & 0xffffull;
    const int bad = badmask ? __builtin_ctzll(badmask) + 1 : 0;
    // A-operand order: register i of lane (r, q) is Linv[r][q + 4 i] (zero above the diagonal)
#pragma unroll
    for (int i = 0; i < 4; ++i) s_inv[i * 64 + lane] = acc[i] * own_dinv;
    return bad;
}
