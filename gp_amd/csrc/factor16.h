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

// Cholesky of a 16x16 SPD tile held in LDS as [row][col] (lower triangle used) and the
// inverse of its factor.  One wave; every lane keeps matrix row lane&15 in registers (the four
// 16-lane DPP rows hold identical copies), pivots and multipliers travel by DPP row
// broadcast, so everything stays in VGPRs.  Out: s_d16 = L16 (upper zeroed), s_inv = L16^-1
// in MFMA A-operand order s_inv[kg*64 + l] = Linv[l&15][(l>>4) + 4*kg].  Returns 0 or
// 1 + index of the first non-positive pivot.
__device__ __forceinline__ int factor16(double (*s_d16)[17], double *s_inv, int lane)
{
    const int lr = lane & 15, lq = lane >> 4;
    double row[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) row[c] = s_d16[lr][c];
    int bad = 0;
    double dinv[16];  // 1 / L_jj (wave-uniform values)
    double own_dinv = 1.0;  // 1 / L_rr of this lane's own row
    static_for<0, 16>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
        double d = bcast16<j>(row[j]);
        const bool neg = !(d > 0.0);
        bad = (neg && !bad) ? j + 1 : bad;
        d = neg ? 1.0 : d;
        // The 128-pivot chain is the critical path of the whole panel phase: the multipliers
        // use 1/sqrt(d) straight from rsqrt (<= 1 ulp); the diagonal entry and its reciprocal
        // (needed only by the inverse, later) get a Newton correction off the chain.
        const double ri = rsqrt(d);
        const double cj = row[j] * ri;
        static_for<j + 1, 16>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            row[c] = fma(-cj, bcast16<c>(cj), row[c]);  // L[c][j] lives in lane c
        });
        double s = d * ri;
        s = fma(0.5 * ri, fma(-s, s, d), s);
        dinv[j] = fma(ri, fma(-s, ri, 1.0), ri);
        own_dinv = (lr == j) ? dinv[j] : own_dinv;
        row[j] = (lr == j) ? s : cj;
    });
    // Inverse by rows: Y = L16^-1, lane r accumulates row r.  At step k row k is final
    // (acc_k * 1/L_kk), is broadcast from lane k, and every lane r > k subtracts L[r][k] * Y[k][:].
    // Each broadcast depends on a value computed in the previous step, so hipcc cannot hoist
    // them all up front (the column-oriented form cost > 256 VGPRs that way).
    double acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = (c == lr) ? 1.0 : 0.0;
    static_for<0, 16>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        const double m = (lr > k) ? row[k] : 0.0;  // L[r][k] for the rows still open
        static_for<0, k + 1>([&](auto cc) {
            constexpr int c = decltype(cc)::value;
            const double y = bcast16<k>(acc[c]) * dinv[k];  // Y[k][c]
            acc[c] = fma(-m, y, acc[c]);
        });
    });
    if (lq == 0) {
#pragma unroll
        for (int c = 0; c < 16; ++c) s_d16[lr][c] = (c <= lr) ? row[c] : 0.0;
        // A-operand order: s_inv[kg*64 + l] = Linv[j = l&15][k = (l>>4) + 4*kg]; lane j = lr holds row j
#pragma unroll
        for (int k = 0; k < 16; ++k) s_inv[(k >> 2) * 64 + lr + 16 * (k & 3)] = (k <= lr) ? acc[k] * own_dinv : 0.0;
    }
    // all lanes saw the same pivots; make the flag wave-uniform for the caller
    return __builtin_amdgcn_readfirstlane(bad);
}
