// Blocked right-looking fp64 Cholesky for gfx950 built on v_mfma_f64_16x16x4_f64.
//
// Everything is expressed in 16x16 tiles held in the MFMA accumulator layout
//   D[i][j]: lane l holds i = (l>>4) + 4*reg, j = l&15       (reg = 0..3)
// whose register `reg` is, lane for lane, also a valid A operand
// (A[i=l&15][k=(l>>4)+4*kg]) of the transposed tile and a valid B operand
// (B[k=(l>>4)+4*kg][j=l&15]) of the tile itself.  A tile therefore feeds the
// next MFMA straight from registers: no LDS round trip, no shuffles.
//
// We always compute the TRANSPOSE of the mathematical block, D[n][m] with m the
// matrix row: for column-major storage a register then covers 16 consecutive
// rows (128 contiguous bytes) of 4 columns.
//
// Kernels
//   k_potrf_diag4 one workgroup: factor a <=128x128 diagonal block, emit its
//                 factors packed in fragment order (Fpack) for the panel solve
//                 (k_potrf_diag: the 5-wave form of round 1, kept as an option)
//   k_trsm_panel  rows below the block: X = A21 L11^-T by blocked substitution,
//                 each wave owns 16 rows and chains MFMAs through registers
//   k_gemm_nt     C -= A B^T / C = A B^T, 128x128 tiles, LDS-staged with
//                 global_load_lds, 2-stage pipeline; SYRK mode walks only the
//                 lower-triangular tiles of the trailing matrix
// Reference: Stan cholesky_decompose (models/fit_hyperparameters.stan:25),
// multi_normal_cholesky (:31), L*z (models/exact_gp.stan:25).
#include "gpmi_internal.h"
#include <math.h>

namespace {

__device__ __forceinline__ d4 mfma(double a, double b, d4 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// C-tile accesses of the update kernels: dbg bit 3 (probe) makes them non-temporal so that the
// streamed C tiles do not displace the panel operands, which every tile re-reads, from L2
__device__ __forceinline__ double ld_c(const double *p, bool nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ void st_c(double *p, double v, bool nt)
{
    if (nt) __builtin_nontemporal_store(v, p);
    else *p = v;
}

__device__ __forceinline__ double readlane64(double v, int l)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, l);
    hi = __builtin_amdgcn_readlane(hi, l);
    return __hiloint2double(hi, lo);
}

// Fpack tile slots (256 doubles each): negated L block (jb,kb), kb<jb, then Linv16 of block jb
__device__ __host__ constexpr int fp_l(int jb, int kb) { return jb * (jb - 1) / 2 + kb; }
__device__ __host__ constexpr int fp_inv(int jb) { return 28 + jb; }


#include "factor16.h"
#include "se_device.h"

#ifdef GPMI_PROBES  // the 5-wave / 3-barrier diagonal-block kernel of round 1 (diag_waves = 5): A/B material only
// ---------------------------------------------------------------------------
// Diagonal block (<= 128 x 128).  One workgroup of 5 waves:
//   waves 0-3 "tile waves": wave w owns block-rows w and 7-w (9 register tiles, balanced),
//             held transposed in the MFMA accumulator layout T[jb][reg] = A[row][jb*16+col];
//             fully unrolled over the 8 block columns (compile-time tile indices);
//   wave 4    "factor wave": runs the sequential 16x16 factor + inverse (factor16) for every
//             block column out of LDS; it holds no tiles, so the register-hungry broadcast
//             code exists once and never forces tile copies at control-flow joins.
// Right-looking over the block columns; three workgroup barriers per column.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(320) void k_potrf_diag(double *__restrict__ A, size_t lda, int nb_act,
                                                    double *__restrict__ Fpack, int *info, int col0)
{
    __shared__ double s_pub[2][8][256];
    __shared__ double s_inv[8][256];  // L16^-1 of every block column (kept: written to Fpack at the end)
    __shared__ double s_d16[16][17];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    // No global store is issued inside the column loop: a __syncthreads() behind a store waits
    // for its acknowledgement (~us).  The packed factors go out after the loop -- the -L tiles
    // ARE the final tile registers, the inverses wait in LDS.

    if (w == 4) {
#pragma unroll 1
        for (int kb = 0; kb < 8; ++kb) {
            __syncthreads();  // B1: the owner's diagonal tile is in s_d16
            const int bad = factor16(s_d16, s_inv[kb], lane);
            if (bad && lane == 0) atomicCAS(info, 0, col0 + kb * 16 + bad);
            __syncthreads();  // B2: L16 in s_d16, L16^-1 in s_inv
            __syncthreads();  // B3: -X tiles published
        }
        return;
    }

    const int ra = w, rb = 7 - w;  // the two block-rows of this wave, ra < rb
    d4 TA[8], TB[8];
#define GPMI_LOAD_ROW(T, br)                                                                           \
    _Pragma("unroll") for (int jb = 0; jb < 8; ++jb) {                                                 \
        T[jb] = d4{0.0, 0.0, 0.0, 0.0};                                                                \
        if (jb <= (br)) {                                                                              \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
                const int col = jb * 16 + lq + 4 * i, row = (br) * 16 + lr;                            \
                const int rr = row > col ? row : col, cc = row > col ? col : row;                      \
                T[jb][i] = (rr < nb_act) ? A[(size_t)rr + (size_t)cc * lda] : (row == col ? 1.0 : 0.0); \
            }                                                                                          \
        }                                                                                              \
    }
    GPMI_LOAD_ROW(TA, ra)
    GPMI_LOAD_ROW(TB, rb)

#define GPMI_SOLVE_ROW(T, br, X)                                                                       \
    if ((br) == kb) {                                                                                  \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) T[kb][i] = s_d16[lr][lq + 4 * i];                \
    } else if ((br) > kb) {                                                                            \
        _Pragma("unroll") for (int kg = 0; kg < 4; ++kg)                                               \
            X = mfma(s_inv[kb][kg * 64 + lane], T[kb][kg], X);                                         \
        T[kb] = X;                                                                                     \
        _Pragma("unroll") for (int kg = 0; kg < 4; ++kg) s_pub[kb & 1][br][kg * 64 + lane] = -X[kg];   \
    }
#define GPMI_UPDATE_ROW(T, br, X)                                                                      \
    _Pragma("unroll") for (int jb = kb + 1; jb < 8; ++jb) {                                            \
        if (jb <= (br)) {                                                                              \
            _Pragma("unroll") for (int kg = 0; kg < 4; ++kg)                                           \
                T[jb] = mfma(s_pub[kb & 1][jb][kg * 64 + lane], X[kg], T[jb]);                         \
        }                                                                                              \
    }

#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        // (a) the owner hands its updated diagonal tile to the factor wave in matrix order
        if (ra == kb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) s_d16[lr][lq + 4 * i] = TA[kb][i];
        } else if (rb == kb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) s_d16[lr][lq + 4 * i] = TB[kb][i];
        }
        __syncthreads();  // B1
        __syncthreads();  // B2: factor wave done
        // (a3) the owner reloads L16 in tile layout; (b) rows below: X = Linv16 * T[kb], publish -X
        d4 XA = d4{0.0, 0.0, 0.0, 0.0}, XB = XA;
        GPMI_SOLVE_ROW(TA, ra, XA)
        GPMI_SOLVE_ROW(TB, rb, XB)
        __syncthreads();  // B3
        // (c) trailing tiles of both block-rows: T[jb] -= L[jb][kb] * X_kb
        if (ra > kb) { GPMI_UPDATE_ROW(TA, ra, XA) }
        if (rb > kb) { GPMI_UPDATE_ROW(TB, rb, XB) }
    }

#define GPMI_STORE_ROW(T, br)                                                                          \
    _Pragma("unroll") for (int jb = 0; jb < 8; ++jb) {                                                 \
        if (jb <= (br)) {                                                                              \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
                const int col = jb * 16 + lq + 4 * i, row = (br) * 16 + lr;                            \
                if (row < nb_act && col <= row) A[(size_t)row + (size_t)col * lda] = T[jb][i];         \
            }                                                                                          \
            if (jb < (br)) {                                                                           \
                _Pragma("unroll") for (int kg = 0; kg < 4; ++kg)                                       \
                    Fpack[(size_t)((br) * ((br) - 1) / 2 + jb) * 256 + kg * 64 + lane] = -T[jb][kg];   \
            }                                                                                          \
        }                                                                                              \
    }
    GPMI_STORE_ROW(TA, ra)
    GPMI_STORE_ROW(TB, rb)
    // the two inverses this wave's block-rows belong to
#pragma unroll
    for (int kg = 0; kg < 4; ++kg) {
        Fpack[(size_t)fp_inv(ra) * 256 + kg * 64 + lane] = s_inv[ra][kg * 64 + lane];
        Fpack[(size_t)fp_inv(rb) * 256 + kg * 64 + lane] = s_inv[rb][kg * 64 + lane];
    }
#undef GPMI_LOAD_ROW
#undef GPMI_SOLVE_ROW
#undef GPMI_UPDATE_ROW
#undef GPMI_STORE_ROW
}

#endif  // GPMI_PROBES

// ---------------------------------------------------------------------------
// Diagonal block, 4-wave variant: the same algorithm with THREE tile waves (block-rows
// {7,2,0}, {6,3,1}, {5,4}: 12 / 13 / 11 register tiles) and the factor wave.  One wave per
// SIMD and ~50 KB of LDS: the workgroup fits into the half of a CU that a retiring
// trailing-update workgroup leaves behind, so next to a running SYRK it starts within
// microseconds instead of waiting for a whole CU to drain by chance (5 waves need two wave
// slots with ~200 registers each on one SIMD, which a resident SYRK wave rules out).
// ---------------------------------------------------------------------------
#ifdef GPMI_PROBES
// where a diagonal-block body spends its cycles (accumulated over all bodies since the last read): [0] block loads
// (drained), [1] the 8-step loop, [2] stores, [3] factor wave inside factor16, [4] factor wave waiting at B1 for the
// next diagonal tile, [5] bodies
__device__ unsigned long long g_body[8];
#define GPMI_BSTAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime();
#define GPMI_BADD(i, d) atomicAdd(&g_body[i], (unsigned long long)(d));
#else
#define GPMI_BSTAMP(v)
#define GPMI_BADD(i, d)
#endif
template <bool COH>
__device__ __forceinline__ double ld_blk(const double *p)
{
    if constexpr (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return *p;
}
constexpr int DIAG4_LDS = 2 * 8 * 256 + 8 * 256 + 2 * 16 * 17 + 2;  // doubles of workgroup memory the body needs (52 KB; the last two: the tile waves' arrival counter)
// COH: the block was written by other workgroups of the same launch with agent-scope stores; read it
// with agent-scope loads (they do not trust this XCD's L2) instead of invalidating caches with a fence
// FULL: the block has all 128 rows and columns (every panel but a ragged last one): the tiles below the
// diagonal are loaded and stored unconditionally -- the guarded form costs a compare, an exec-mask
// save / restore and a branch per ELEMENT (60 per lane), ~3 us per block
// nblk (ragged blocks only): number of 16-column pivot blocks to run, ceil(nb_act / 16) -- a small matrix does not
// pay for the identity padding's pivots (n = 21: 2 of 8 block columns); Fpack slots of the skipped blocks are then
// never read by the consumers, which loop over the same count
template <bool COH = false, bool FULL = false>
__device__ __forceinline__ void potrf_diag4_body(double *__restrict__ sm, double *__restrict__ A, size_t lda, int nb_act,
                                                 double *__restrict__ Fpack, int *info, int col0, int nblk = 8,
                                                 int tid = (int)threadIdx.x)
{
    if (FULL) nblk = 8;
    // ragged blocks: block-rows >= nblk are identity padding -- not loaded, solved, updated or stored (at n = 21 two of
    // eight block-rows exist; carrying the padding through every step was half of the body's time there)
#define GPMI_ACT(br) (FULL || (br) < nblk)
    double (*s_pub)[8][256] = reinterpret_cast<double (*)[8][256]>(sm);
    double (*s_inv)[256] = reinterpret_cast<double (*)[256]>(sm + 2 * 8 * 256);
    // the diagonal tile travels to the factor wave and comes back as L16 through s_d16[kb & 1]: two
    // buffers, so that the owner of block-row kb + 1 can hand over the NEXT diagonal tile while the
    // owner of block-row kb still reads L16 of this step
    double (*s_d16)[16][17] = reinterpret_cast<double (*)[16][17]>(sm + 2 * 8 * 256 + 8 * 256);
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;

    // TWO workgroup barriers per block column.  The chain factor16(kb) -> solve of block-row kb + 1 ->
    // update of its diagonal tile -> factor16(kb + 1) crosses B2 and B1 only: the update of the next
    // diagonal tile needs nothing but its owner's own solve result (register r of X is, lane for lane,
    // the A and the B operand of X X^T), so no barrier stands between the solve and that update (the
    // third barrier of the earlier form cost ~2 k of the ~7.7 k cycles per step).
    // The barriers order LDS traffic only (the waves talk through s_d16 / s_inv / s_pub): they are raw
    // s_waitcnt lgkmcnt(0) + s_barrier, NOT __syncthreads(), whose release fence also drains vmcnt -- so the block's
    // global loads may still be in flight at the first barriers (they are issued in the order they are needed: block
    // column 0 of every row first) and the finished tiles are stored from inside the loop, under the factor wave's
    // time, instead of in a ~6 k-cycle tail behind it.  Nothing in the body reads global memory another wave of the
    // workgroup has written.
#define GPMI_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
    if (w == 3) {
#pragma unroll 1
        for (int kb = 0; kb < nblk; ++kb) {
            GPMI_BSTAMP(f0)
            GPMI_LDS_BARRIER();  // B1: the owner's diagonal tile is in s_d16[kb & 1]; -X tiles of step kb - 1 published
            GPMI_BSTAMP(f1)
            const int bad = factor16(s_d16[kb & 1], s_inv[kb], lane);
            if (bad && lane == 0) atomicCAS(info, 0, col0 + kb * 16 + bad);
            GPMI_BSTAMP(f2)
#ifdef GPMI_PROBES
            if (lane == 0) {
                GPMI_BADD(3, f2 - f1)
                GPMI_BADD(4, f1 - f0)
            }
#endif
            GPMI_LDS_BARRIER();  // B2: L16 in s_d16[kb & 1], L16^-1 in s_inv[kb]
        }
        return;
    }
    GPMI_BSTAMP(b0)

    // block-rows of this wave, ra > rb > rc (rc = -1: none); array sizes cover the largest row of each class
    const int ra = 7 - w, rb = 2 + w, rc = w < 2 ? w : -1;
    d4 TA[8], TB[5], TC[2];
#define GPMI_CL(jb, NJ) ((jb) < (NJ) ? (jb) : 0)  // keeps compile-time indices of never-taken branches in range
    // per block-row: A + (16 br + lr) + lq lda -- element (i) of tile jb is then a wave-uniform multiple of lda away
    // (the general form costs a max / min / 64-bit multiply-add per element: 4.5 k cycles of pure address arithmetic
    // for the 60 loads of a lane)
    const double *const pra = A + (size_t)(ra * 16 + lr) + (size_t)lq * lda;
    const double *const prb = A + (size_t)(rb * 16 + lr) + (size_t)lq * lda;
    const double *const prc = A + (size_t)((rc < 0 ? 0 : rc) * 16 + lr) + (size_t)lq * lda;
#define GPMI_LOAD_TILE(T, br, jb, PR)                                                                  \
    {                                                                                                  \
        T[jb] = d4{0.0, 0.0, 0.0, 0.0};                                                                \
        if (!GPMI_ACT(br)) {                                                                           \
        } else if (FULL && (jb) < (br)) {                                                              \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) T[jb][i] = ld_blk<COH>(PR + (size_t)((jb) * 16 + 4 * i) * lda); \
        } else if ((jb) <= (br)) {                                                                     \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                            \
                const int col = (jb) * 16 + lq + 4 * i, row = (br) * 16 + lr;                          \
                const int rr = row > col ? row : col, cc = row > col ? col : row;                      \
                if (FULL) T[jb][i] = ld_blk<COH>(A + (size_t)rr + (size_t)cc * lda);                   \
                else T[jb][i] = (rr < nb_act) ? ld_blk<COH>(A + (size_t)rr + (size_t)cc * lda) : (row == col ? 1.0 : 0.0); \
            }                                                                                          \
        }                                                                                              \
    }
    // block column by block column, lowest rows first: tile (0, 0) and then the tiles of block column 0 are what the
    // first steps wait for
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) {
        if (jb < 2) GPMI_LOAD_TILE(TC, rc, jb, prc)
        if (jb < 5) GPMI_LOAD_TILE(TB, rb, jb, prb)
        GPMI_LOAD_TILE(TA, ra, jb, pra)
    }

    // step k of block-row br: the diagonal tile comes back from the factor wave as L16; a row below is solved against
    // L16^-1 (4 chained MFMAs), keeps X as its final tile and publishes -X for the other rows' updates
#define GPMI_SOLVE_ROW(T, NJ, br, X, k)                                                                \
    if ((br) == (k)) {                                                                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) T[GPMI_CL(k, NJ)][i] = s_d16[(k) & 1][lr][lq + 4 * i]; \
    } else if ((br) > (k) && GPMI_ACT(br)) {                                                           \
        X = d4{0.0, 0.0, 0.0, 0.0};                                                                    \
        _Pragma("unroll") for (int kg = 0; kg < 4; ++kg)                                               \
            X = mfma(s_inv[k][kg * 64 + lane], T[GPMI_CL(k, NJ)][kg], X);                              \
        T[GPMI_CL(k, NJ)] = X;                                                                         \
        _Pragma("unroll") for (int kg = 0; kg < 4; ++kg) s_pub[(k) & 1][br][kg * 64 + lane] = -X[kg];  \
    }

    // The only tile the next pivot block waits for is the diagonal tile of block-row kb + 1: its
    // owner updates it first (EARLY), straight from the registers of its own solve, and hands it to
    // the factor wave; every other update of step kb (REST) runs in the next iteration between B1
    // and B2, i.e. under the factor wave's 4.4 k cycles.
#define GPMI_UPDATE_EARLY(T, NJ, br, X)                                                                \
    {                                                                                                  \
        _Pragma("unroll") for (int kg = 0; kg < 4; ++kg)                                               \
            T[GPMI_CL(kb + 1, NJ)] = mfma(-X[kg], X[kg], T[GPMI_CL(kb + 1, NJ)]);                      \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) s_d16[(kb + 1) & 1][lr][lq + 4 * i] = T[GPMI_CL(kb + 1, NJ)][i]; \
    }
#define GPMI_UPDATE_REST(T, NJ, br, X)                                                                 \
    _Pragma("unroll") for (int jb = kb; jb < (NJ); ++jb) {                                             \
        if (jb <= (br) && !(jb == kb && (br) == kb)) {                                                 \
            _Pragma("unroll") for (int kg = 0; kg < 4; ++kg)                                           \
                T[jb] = mfma(s_pub[(kb - 1) & 1][jb][kg * 64 + lane], X[kg], T[jb]);                   \
        }                                                                                              \
    }
    // Block column k of block-row br is final once step k has solved it: L tile (from the solve's registers) and its
    // packed negative below the diagonal; on the diagonal the factor's tile and the inverse the factor wave left in
    // s_inv[k].  Issued one step later, behind B1, so that the stores do not sit between B2 and B1 (the critical path).
#define GPMI_STORE_STEP(T, NJ, br, k, PR)                                                              \
    if ((br) > (k) && !GPMI_ACT(br)) {                                                                 \
    } else if ((br) > (k)) {                                                                           \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                \
            if (FULL || (br) * 16 + lr < nb_act)                                                       \
                const_cast<double *>(PR)[(size_t)((k) * 16 + 4 * i) * lda] = T[GPMI_CL(k, NJ)][i];     \
        }                                                                                              \
        _Pragma("unroll") for (int kg = 0; kg < 4; ++kg)                                               \
            Fpack[(size_t)fp_l(br, k) * 256 + kg * 64 + lane] = -T[GPMI_CL(k, NJ)][kg];                \
    } else if ((br) == (k)) {                                                                          \
        _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                \
            const int col = (k) * 16 + lq + 4 * i, row = (br) * 16 + lr;                               \
            if (col <= row && (FULL || row < nb_act)) A[(size_t)row + (size_t)col * lda] = T[GPMI_CL(k, NJ)][i]; \
        }                                                                                              \
        _Pragma("unroll") for (int kg = 0; kg < 4; ++kg)                                               \
            Fpack[(size_t)fp_inv(k) * 256 + kg * 64 + lane] = s_inv[k][kg * 64 + lane];                \
    }

    GPMI_BSTAMP(b1)
    // Between B2 (factor16 of step kb done) and B1 (the next diagonal tile handed over) -- the critical path -- ONLY the
    // owner of block-row kb + 1 works: it solves that row (4 MFMAs), updates the next diagonal tile from its own
    // registers (4 MFMAs) and hands it to the factor wave.  Every other solve of step kb, the publication of the -X
    // tiles, the stores and the REST updates run behind B1, under factor16(kb + 1); the REST updates read the other
    // rows' -X tiles, so the three tile waves meet once more in between, on an arrival counter in LDS (the factor
    // wave, busy on the chain, takes no part).  Before: all solves of a step stood between B2 and B1 (~2.0 k cycles
    // per step against ~0.9 k now).
    // (explicitly an LDS pointer: through a generic one the accesses become FLAT operations, whose completion the
    // compiler can only await with vmcnt(0) -- which would drain the block loads still in flight)
    typedef __attribute__((address_space(3))) int lds_int;
    lds_int *const s_cnt = (lds_int *)(sm + DIAG4_LDS - 2);
    if (tid == 0) *s_cnt = 0;   // ordered before every arrival by the first B1
    auto tile_waves_meet = [&](int target) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's -X tiles are in LDS
        if (lane == 0) __hip_atomic_fetch_add(s_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        while (__hip_atomic_load(s_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
    };
    d4 XA[8], XB[8], XC[8];
    if (rc == 0) {  // block-row 0 hands tile (0, 0) to the factor wave in matrix order
#pragma unroll
        for (int i = 0; i < 4; ++i) s_d16[0][lr][lq + 4 * i] = TC[0][i];
    }
    int last = -1;  // last step whose non-critical solves and stores are still due
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        if (!FULL && kb >= nblk) break;  // workgroup-uniform
        GPMI_LDS_BARRIER();  // B1: diagonal tile kb is in s_d16[kb & 1] (factor16(kb) starts)
        if (kb > 0) {
            // the rest of step kb - 1: its other rows' solves (row kb was solved before B1) ...
            if (ra != kb) { GPMI_SOLVE_ROW(TA, 8, ra, XA[kb - 1], kb - 1) }
            if (rb != kb) { GPMI_SOLVE_ROW(TB, 5, rb, XB[kb - 1], kb - 1) }
            if (rc != kb) { GPMI_SOLVE_ROW(TC, 2, rc, XC[kb - 1], kb - 1) }
            tile_waves_meet(3 * kb);   // ... every row's -X tile of step kb - 1 is published ...
            GPMI_STORE_STEP(TA, 8, ra, kb - 1, pra)
            GPMI_STORE_STEP(TB, 5, rb, kb - 1, prb)
            GPMI_STORE_STEP(TC, 2, rc, kb - 1, prc)
            // ... and REST: tiles jb >= kb of the rows below, except tile (kb, kb) (updated EARLY)
            if (ra >= kb && GPMI_ACT(ra)) { GPMI_UPDATE_REST(TA, 8, ra, XA[kb - 1]) }
            if (rb >= kb && GPMI_ACT(rb)) { GPMI_UPDATE_REST(TB, 5, rb, XB[kb - 1]) }
            if (rc >= kb && GPMI_ACT(rc)) { GPMI_UPDATE_REST(TC, 2, rc, XC[kb - 1]) }
        }
        GPMI_LDS_BARRIER();  // B2: factor16(kb) done: L16 in s_d16[kb & 1], its inverse in s_inv[kb]
        if (kb < 7 && GPMI_ACT(kb + 1)) {     // the critical row kb + 1: solve, EARLY update of the next diagonal tile, hand-over
            if (ra == kb + 1) { GPMI_SOLVE_ROW(TA, 8, ra, XA[kb], kb) GPMI_UPDATE_EARLY(TA, 8, ra, XA[kb]) }
            else if (rb == kb + 1) { GPMI_SOLVE_ROW(TB, 5, rb, XB[kb], kb) GPMI_UPDATE_EARLY(TB, 5, rb, XB[kb]) }
            else if (rc == kb + 1) { GPMI_SOLVE_ROW(TC, 2, rc, XC[kb], kb) GPMI_UPDATE_EARLY(TC, 2, rc, XC[kb]) }
        }
        last = kb;
    }
#undef GPMI_UPDATE_EARLY
#undef GPMI_UPDATE_REST
    GPMI_BSTAMP(b2)
    // the last step: its diagonal tile back from the factor wave (rows below it are padding: nothing to solve), its stores
    // (compile-time step index: one copy per possible last step of a ragged block)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        if (k == last) {
            if (ra == k) { GPMI_SOLVE_ROW(TA, 8, ra, XA[k], k) }
            if (rb == k) { GPMI_SOLVE_ROW(TB, 5, rb, XB[k], k) }
            if (rc == k) { GPMI_SOLVE_ROW(TC, 2, rc, XC[k], k) }
            GPMI_STORE_STEP(TA, 8, ra, k, pra)
            GPMI_STORE_STEP(TB, 5, rb, k, prb)
            GPMI_STORE_STEP(TC, 2, rc, k, prc)
        }
    }
#ifdef GPMI_PROBES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (w == 0 && lane == 0) {
        GPMI_BSTAMP(b3)
        GPMI_BADD(0, b1 - b0)
        GPMI_BADD(1, b2 - b1)
        GPMI_BADD(2, b3 - b2)
        GPMI_BADD(5, 1)
    }
#endif
#undef GPMI_CL
#undef GPMI_ACT
#undef GPMI_LOAD_TILE
#undef GPMI_SOLVE_ROW
#undef GPMI_STORE_STEP
#undef GPMI_LDS_BARRIER
}

__global__ __launch_bounds__(256, 2) void k_potrf_diag4(double *__restrict__ A, size_t lda, int nb_act,
                                                     double *__restrict__ Fpack, int *info, int col0)
{
    __shared__ double sm[DIAG4_LDS];
    if (nb_act == GPMI_NB) potrf_diag4_body<false, true>(sm, A, lda, nb_act, Fpack, info, col0);
    else potrf_diag4_body<false, false>(sm, A, lda, nb_act, Fpack, info, col0);
}

// ---------------------------------------------------------------------------
// Panel solve: rows [row0, M) of the nb_act columns starting at Acol.
// X L11^T = A21 by block forward substitution over the 8 block columns; each
// wave carries its 16 rows through all steps in registers.
// ---------------------------------------------------------------------------
// FULL: all 64 rows of the workgroup and all 128 columns exist -- unconditional, batched loads and
// stores from one running column pointer (the guarded form predicates and branches per element)
// kb0 (wave-uniform): the strip's columns left of block kb0 are zero (rows of the identity / of an upper-triangular
// operand): those block steps produce zeros and are skipped
template <bool FULL>
__device__ __forceinline__ void trsm_panel_body(const double *__restrict__ s_F, double *__restrict__ Acol, size_t lda,
                                                int r, bool rok, int nb_act, int tid = (int)threadIdx.x, int kb0 = 0)
{
    const int lane = tid & 63;
    const int lq = lane >> 4;
    const int nblk = FULL ? 8 : (nb_act + 15) >> 4;
    d4 T[8];
    if constexpr (FULL) {
        const double *p = Acol + (size_t)r + (size_t)lq * lda;
#pragma unroll
        for (int jb = 0; jb < 8; ++jb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                T[jb][i] = *p;
                p += 4 * lda;
            }
    } else {
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = jb * 16 + lq + 4 * i;
                T[jb][i] = (rok && col < nb_act) ? Acol[(size_t)r + (size_t)col * lda] : 0.0;
            }
        }
    }
    __syncthreads();  // packed factors are in s_F

#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        if (kb >= kb0 && kb < nblk) {
            d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) acc = mfma(s_F[fp_inv(kb) * 256 + kg * 64 + lane], T[kb][kg], acc);
            T[kb] = acc;
#pragma unroll
            for (int jb = kb + 1; jb < 8; ++jb) {
                if (jb < nblk) {
#pragma unroll
                    for (int kg = 0; kg < 4; ++kg)
                        T[jb] = mfma(s_F[fp_l(jb, kb) * 256 + kg * 64 + lane], T[kb][kg], T[jb]);
                }
            }
        }
    }

    if constexpr (FULL) {
        double *p = Acol + (size_t)r + (size_t)lq * lda;
#pragma unroll
        for (int jb = 0; jb < 8; ++jb)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *p = T[jb][i];
                p += 4 * lda;
            }
    } else {
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int col = jb * 16 + lq + 4 * i;
                if (rok && col < nb_act) Acol[(size_t)r + (size_t)col * lda] = T[jb][i];
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_trsm_panel(double *__restrict__ Acol, size_t lda, int row0,
                                                    int M, int nb_act, const double *__restrict__ Fpack)
{
    __shared__ __attribute__((aligned(16))) double s_F[GPMI_FPACK];
    {
        const double2 *src = reinterpret_cast<const double2 *>(Fpack);
        double2 *dst = reinterpret_cast<double2 *>(s_F);
#pragma unroll
        for (int q = 0; q < GPMI_FPACK / 2 / 256; ++q) dst[threadIdx.x + 256 * q] = src[threadIdx.x + 256 * q];
    }
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int rbase = row0 + blockIdx.x * 64;
    const int r = rbase + w * 16 + (lane & 15);
    if (nb_act == GPMI_NB && rbase + 64 <= M)  // workgroup-uniform
        trsm_panel_body<true>(s_F, Acol, lda, r, true, nb_act);
    else
        trsm_panel_body<false>(s_F, Acol, lda, r, r < M, nb_act);
}

// ---------------------------------------------------------------------------
// GEMM NT on 128x128 tiles.  A: M x K, B: N x K (both with the tile index
// contiguous), C: M x N.  MODE 0: C -= A B^T; 1: same, A == B panel, lower
// tiles only (SYRK); 2: C = A B^T.
// LDS image per stage and operand: [16 k][144] doubles -- each k-row is one
// 1-KiB global_load_lds write; the 128-B row pad puts k and k+1 on opposite
// halves of the 64 banks so the ds_read_b64 fragment reads are conflict-free.
// ---------------------------------------------------------------------------
constexpr int GT = 128, GK = 16, GP = 144;

// SYRK tile order.  Workgroups are dealt round-robin to the 8 XCDs (observed dispatch
// behaviour; performance only), so block b runs on XCD-group b % 8 as that group's (b / 8)-th
// workgroup.  Each group walks 8x8 SUPER-TILES of the lower triangle: the 64 tiles that are
// resident on one XCD at a time then share 8 row-blocks and 8 column-blocks of the panel
// through that XCD's L2 (16 blocks per 64 tiles instead of ~65 in row-major order), which
// takes the operand stream off HBM / Infinity Cache.  Tiles of a super-tile that fall outside
// the triangle exit at once.
// TN < T: the lower TRAPEZOID of a T x TN tile grid (a block column of the trailing matrix, the
// first part of a look-ahead update): the TN x TN triangle first, then the rows below it.
constexpr int ST = 8;
__device__ __forceinline__ bool syrk_tile(int b, int T, int TN, int order, int &ti, int &tj)
{
    if (order == 0) {  // plain row-major walk of the lower triangle / trapezoid
        const int tri = TN * (TN + 1) / 2;
        if (b >= tri) {
            const int q = b - tri;
            ti = TN + q / TN;
            tj = q % TN;
            return ti < T;
        }
        int i = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);
        while ((i + 1) * (i + 2) / 2 <= b) ++i;
        while (i * (i + 1) / 2 > b) --i;
        ti = i;
        tj = b - i * (i + 1) / 2;
        return ti < T;
    }
#ifndef GPMI_PROBES
    return false;  // the XCD-grouped super-tile walk (measured slower in place) exists in the probe build only
#else
    const int xcd = b & 7, q = b >> 3;
    const int s = (q / (ST * ST)) * 8 + xcd;       // super-tile index, row-major over the triangle
    const int local = q % (ST * ST);
    const int nst = (T + ST - 1) / ST;
    if (s >= nst * (nst + 1) / 2) return false;
    int I = (int)((sqrt(8.0 * (double)s + 1.0) - 1.0) * 0.5);
    while ((I + 1) * (I + 2) / 2 <= s) ++I;
    while (I * (I + 1) / 2 > s) --I;
    const int J = s - I * (I + 1) / 2;
    ti = I * ST + (local % ST);
    tj = J * ST + (local / ST);
    return ti < T && tj <= ti;
#endif
}
__host__ inline int syrk_grid(int T, int TN, int order)
{
    if (order == 0) return TN * (TN + 1) / 2 + (T - TN) * TN;
    const int nst = (T + ST - 1) / ST;
    const int ns = nst * (nst + 1) / 2;
    return ((ns + 7) / 8) * 8 * ST * ST;
}

// XCD-partitioned band walk (order 2) of the lower triangle / trapezoid, WITHOUT a padded grid: the valid tiles are
// enumerated band by band (8 tile rows), inside a band column by column -- 64 consecutive tiles are an 8 x 8 patch --
// and workgroup b takes canonical tile (b % 8) S + b / 8, S = ceil(ntiles / 8): the workgroups the dispatcher deals to
// one XCD (observed: b % 8; speed only, any placement is correct) walk ONE contiguous eighth of that order.  The 64
// tiles resident on an XCD at a time then share 8 row blocks (for a whole band) and 8 column blocks of the panel
// through that XCD's L2 instead of ~5 + 15 that change every round.  Every XCD gets the same number of tiles, so the
// launch ends as evenly as the row-major walk; the last partial round of EACH XCD is what the tail split cuts into
// quadrants, and a thin last tile row (the augmented row) is multiplied as quadrants by its own workgroup.
__device__ __forceinline__ void syrk_tile_bands(int c, int T, int TN, int &ti, int &tj)
{
    for (int r0 = 0;; r0 += 8) {
        const int R = (T - r0 < 8) ? T - r0 : 8;
        const int cf = r0 < TN ? r0 : TN;                          // columns left of the band's diagonal block: R tiles each
        const int dmax = (TN - r0 < R) ? (TN - r0 > 0 ? TN - r0 : 0) : R;  // columns r0 + d of the diagonal block: rows d .. R - 1
        const int cnt = R * cf + dmax * R - dmax * (dmax - 1) / 2;
        if (c < cnt || R <= 0) {
            if (c < R * cf) {
                tj = c / R;
                ti = r0 + c - tj * R;
                return;
            }
            c -= R * cf;
            int d = 0;
            while (d < dmax - 1 && c >= R - d) {
                c -= R - d;
                ++d;
            }
            tj = r0 + d;
            ti = r0 + d + c;
            return;
        }
        c -= cnt;
    }
}

// One 128 x 128 output tile (ti, tj); smem is the workgroup's staging buffer (free on entry:
// every wave has finished reading it).
// MODE 3: the product A B^T is not stored, its elements are consumed where they are (the accumulators never leave their
// registers; C is unused): epi.row(tm, m, ok) announces the lane's four rows, then per column epi.col(n, ok) and
// epi.elem(v, tm) for its four elements -- so that the consumer loads what depends on a row or a column once.
struct NoEpi {
    __device__ void row(int, int, bool) const {}
    __device__ void col(int, bool) const {}
    __device__ void elem(double, int) const {}
};
template <class R, class C, class E>
struct Epi3 {
    R row;
    C col;
    E elem;
};
template <class R, class C, class E>
__device__ __forceinline__ Epi3<R, C, E> make_epi3(R r, C c, E e)
{
    return Epi3<R, C, E>{r, c, e};
}
template <int MODE, bool WHOLE, class EPI = NoEpi>
__device__ __forceinline__ void gemm_tile_k(double (&smem)[2][2][GK][GP], const double *__restrict__ A, size_t lda,
                                            const double *__restrict__ B, size_t ldb, double *__restrict__ C,
                                            size_t ldc, int M, int N, int K, int ti, int tj, int dbg, int tid, EPI &&epi = EPI{})
{
    const int m0 = ti * GT, n0 = tj * GT;
    if (MODE == 1 && n0 >= N) return;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int wm = w & 1, wn = w >> 1;

    // staging: waves 0,1 stream the A tile (m index), waves 2,3 the B tile (n index); 8 k-rows each
    const int op = w >> 1;
    const double *gsrc = (op ? B + n0 : A + m0) + 2 * lane;
    const size_t gld = op ? ldb : lda;
    const int krow0 = (w & 1) * 8;

    d4 acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};

    const int nk = (K + GK - 1) / GK;
    auto issue = [&](int stage, int k0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int kr = krow0 + q;
            int kc = k0 + kr;
            kc = kc < K ? kc : K - 1;  // clamp: never read a column past the operand
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(gsrc + (size_t)kc * gld),
                (__attribute__((address_space(3))) void *)&smem[stage][op][kr][0], 16, 0, 0);
        }
    };
    // Fragments of sub-step kk+1 are requested before the 16 MFMAs of sub-step kk are issued
    // (two register sets), so the LDS latency sits under ~1k cycles of matrix work.  MASK
    // (zeroing of columns past K) is compiled only into the last, possibly partial, k-step:
    // a select on a just-loaded fragment forces the wait in front of the MFMAs.
    auto ldfrag = [&](auto mk, int st, int kk, int klim, double (&af)[4], double (&bf)[4]) {
        constexpr bool MASK = decltype(mk)::value != 0;
        const int kr = kk * 4 + lq;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            af[t] = smem[st][1][kr][wn * 64 + t * 16 + lr];
            bf[t] = smem[st][0][kr][wm * 64 + t * 16 + lr];
        }
        if constexpr (MASK) {
            const bool kv = kr < klim;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                af[t] = kv ? af[t] : 0.0;
                bf[t] = kv ? bf[t] : 0.0;
            }
        }
    };
    auto mm16 = [&](const double (&af)[4], const double (&bf)[4]) {
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) acc[tn][tm] = mfma(af[tn], bf[tm], acc[tn][tm]);
    };
    // mid(): issued between the first fragment reads and the first MFMAs (the DMA of the next stage:
    // its instructions then run under the LDS latency instead of in front of it)
    auto compute = [&](auto mk, int st, int klim, auto &&mid) {
        double a0[4], b0[4], a1[4], b1[4];
        ldfrag(mk, st, 0, klim, a0, b0);
        ldfrag(mk, st, 1, klim, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mid();
        __builtin_amdgcn_sched_barrier(0);
        mm16(a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        ldfrag(mk, st, 2, klim, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mm16(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        ldfrag(mk, st, 3, klim, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mm16(a0, b0);
        mm16(a1, b1);
    };

    if constexpr (WHOLE) {
        const double *g0 = gsrc + (size_t)krow0 * gld;
#pragma unroll
        for (int q = 0; q < 8; ++q)  // stage 0, same lean addressing as the loop
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g0 + (size_t)q * gld),
                                             (__attribute__((address_space(3))) void *)&smem[0][op][krow0 + q][0], 16, 0, 0);
    } else {
        issue(0, 0);
    }
    double a0[4], b0[4];  // whole-k-step form: the fragments of sub-step 0 cross the k-step boundary
    if constexpr (WHOLE) {
        // Whole k-steps only (every launch of a factorisation whose order is a multiple of 16): no
        // clamp, and the eight row addresses of a stage are one running per-lane pointer plus
        // loop-invariant uniform offsets -- one VALU add per DMA instead of the ~12 scalar
        // instructions (min, 64-bit multiply, ...) of the general form.
        // The k-step boundary is software-pipelined (round 3): the barrier that publishes stage t + 1 sits BEFORE the
        // last 16 MFMAs of step t, and the DMA of step t + 2 and the first fragment reads of step t + 1 are issued
        // between those MFMAs -- so what a k-step exposes is the barrier itself, not barrier + DMA issue + LDS latency
        // in front of its first MFMA (4970 cycles per 4096 of MFMA issue for a workgroup alone on a CU before).
        // Stage t & 1 is free for the DMA of step t + 2 at that barrier: every wave has its last fragments of step t
        // in registers (lgkmcnt(0) in front of the barrier).
        const double *gp = gsrc + (size_t)krow0 * gld;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (nk > 1 && !(dbg & 2)) {
            gp += (size_t)GK * gld;
            asm volatile("" : "+v"(gp));
#pragma unroll
            for (int q = 0; q < 8; ++q)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp + (size_t)q * gld),
                                                 (__attribute__((address_space(3))) void *)&smem[1][op][krow0 + q][0], 16, 0, 0);
        }
        ldfrag(ic<0>{}, 0, 0, GK, a0, b0);
        // one k-step that has a successor; DMA: the step after that exists and is requested here
        auto kstep = [&](auto dma, int st) {
            constexpr bool DMA = decltype(dma)::value != 0;
            double a1[4], b1[4];
            // the first reads of a1, b1 go out BEHIND the first four MFMAs: the wait in front of those then covers a0, b0
            // only (issued a quarter of a k-step ago), not an LDS round trip
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) acc[0][tm] = mfma(a0[0], b0[tm], acc[0][tm]);
            __builtin_amdgcn_sched_barrier(0);
            ldfrag(ic<0>{}, st, 1, GK, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tn = 1; tn < 4; ++tn)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) acc[tn][tm] = mfma(a0[tn], b0[tm], acc[tn][tm]);
            __builtin_amdgcn_sched_barrier(0);
            ldfrag(ic<0>{}, st, 2, GK, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            mm16(a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            ldfrag(ic<0>{}, st, 3, GK, a1, b1);
            __builtin_amdgcn_sched_barrier(0);
            mm16(a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
            // keep ONE running per-lane pointer (opaque to the optimiser, which otherwise turns the
            // eight invariant offsets into eight running scalar pointers: 16 SALU per k-step)
            gp += (size_t)GK * gld;
            asm volatile("" : "+v"(gp));
#pragma unroll
            for (int tn = 0; tn < 4; ++tn) {
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) acc[tn][tm] = mfma(a1[tn], b1[tm], acc[tn][tm]);
                __builtin_amdgcn_sched_barrier(0);
                a0[tn] = smem[st ^ 1][1][lq][wn * 64 + tn * 16 + lr];
                b0[tn] = smem[st ^ 1][0][lq][wm * 64 + tn * 16 + lr];
                if constexpr (DMA) if (tn < 2) {  // all eight requests behind the first eight MFMAs: they have until the next barrier
#pragma unroll
                    for (int q = 4 * tn; q < 4 * tn + 4; ++q)
                        __builtin_amdgcn_global_load_lds(
                            (const __attribute__((address_space(1))) void *)(gp + (size_t)q * gld),
                            (__attribute__((address_space(3))) void *)&smem[st][op][krow0 + q][0], 16, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (dbg & 2) {
#pragma unroll 1
            for (int kt = 0; kt < nk - 1; ++kt) kstep(ic<0>{}, kt & 1);
        } else {
#pragma unroll 1
            for (int kt = 0; kt < nk - 2; ++kt) kstep(ic<1>{}, kt & 1);
            if (nk > 1) kstep(ic<0>{}, nk & 1);  // step nk - 2: nothing left to request
        }
    } else {
#pragma unroll 1
        for (int kt = 0; kt < nk - 1; ++kt) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (!(dbg & 2)) issue((kt + 1) & 1, (kt + 1) * GK);
            compute(ic<0>{}, kt & 1, GK, []() {});
        }
    }

    // Last k-step peeled.  For a tile wholly inside the matrix the first half of the C tile
    // (32 loads per lane, uniform offsets from one base) is issued under that step -- the
    // staging loads are all retired by then -- and the second half right after the first
    // half's stores: one memory round trip is exposed per tile instead of four.  For SYRK the
    // diagonal tiles are computed in full (their strictly-upper outputs land in the unused
    // upper triangle of the workspace).
    const bool interior = (m0 + GT <= M) && (n0 + GT <= N);
    double *const cbase = C + (size_t)(m0 + wm * 64 + lr) + (size_t)(n0 + wn * 64 + lq) * ldc;
    double ch[2][4][4];
    const bool cnt = (dbg & 8) != 0;
    if constexpr (!WHOLE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (MODE != 2 && MODE != 3 && interior) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int i = 0; i < 4; ++i) ch[tn][tm][i] = ld_c(cbase + tm * 16 + (size_t)(tn * 16 + 4 * i) * ldc, cnt);
    }
    if constexpr (WHOLE) {  // the fragments of sub-step 0 of the last stage are in registers
        const int st = (nk - 1) & 1;
        double a1[4], b1[4];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) acc[0][tm] = mfma(a0[0], b0[tm], acc[0][tm]);
        __builtin_amdgcn_sched_barrier(0);
        ldfrag(ic<0>{}, st, 1, GK, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int tn = 1; tn < 4; ++tn)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm) acc[tn][tm] = mfma(a0[tn], b0[tm], acc[tn][tm]);
        __builtin_amdgcn_sched_barrier(0);
        ldfrag(ic<0>{}, st, 2, GK, a0, b0);
        __builtin_amdgcn_sched_barrier(0);
        mm16(a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        ldfrag(ic<0>{}, st, 3, GK, a1, b1);
        __builtin_amdgcn_sched_barrier(0);
        mm16(a0, b0);
        mm16(a1, b1);
    } else {
        compute(ic<1>{}, (nk - 1) & 1, K - (nk - 1) * GK, []() {});
    }
    if (dbg & 1) {
        double sacc = 0.0;
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int i = 0; i < 4; ++i) sacc += acc[tn][tm][i];
        if (sacc == 1.2345e300) cbase[0] = sacc;
        return;
    }
    if constexpr (MODE == 3) {
#pragma unroll
        for (int tm = 0; tm < 4; ++tm) {
            const int m = m0 + wm * 64 + tm * 16 + lr;
            epi.row(tm, m, m < M);
        }
#pragma unroll
        for (int tn = 0; tn < 4; ++tn)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + wn * 64 + tn * 16 + lq + 4 * i;
                epi.col(n, n < N);
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) epi.elem(acc[tn][tm][i], tm);
            }
        return;
    }
    if (interior) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    st_c(cbase + tm * 16 + (size_t)(tn * 16 + 4 * i) * ldc,
                         (MODE == 2) ? acc[tn][tm][i] : ch[tn][tm][i] - acc[tn][tm][i], cnt);
        if (MODE != 2) {
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        ch[tn][tm][i] = ld_c(cbase + tm * 16 + (size_t)((tn + 2) * 16 + 4 * i) * ldc, cnt);
        }
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    st_c(cbase + tm * 16 + (size_t)((tn + 2) * 16 + 4 * i) * ldc,
                         (MODE == 2) ? acc[tn + 2][tm][i] : ch[tn][tm][i] - acc[tn + 2][tm][i], cnt);
        return;
    }

    // edge tile: per tn, 16 loads from clamped (always valid) addresses, then guarded stores
#pragma unroll
    for (int tn = 0; tn < 4; ++tn) {
        double ce[4][4];
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int n = n0 + wn * 64 + tn * 16 + lq + 4 * i, m = m0 + wm * 64 + tm * 16 + lr;
                n = n < N ? n : N - 1;
                m = m < M ? m : M - 1;
                if (MODE != 2) ce[tm][i] = C[(size_t)m + (size_t)n * ldc];
            }
#pragma unroll
        for (int tm = 0; tm < 4; ++tm)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int n = n0 + wn * 64 + tn * 16 + lq + 4 * i, m = m0 + wm * 64 + tm * 16 + lr;
                if (m < M && n < N && (MODE != 1 || n <= m))
                    C[(size_t)m + (size_t)n * ldc] = (MODE == 2) ? acc[tn][tm][i] : ce[tm][i] - acc[tn][tm][i];
            }
    }
}


// Whole k-steps (every launch of a factorisation whose order is a multiple of 16) take the software-pipelined form;
// the two forms are separate instantiations so that neither's live ranges weigh on the other's register allocation.
template <int MODE, class EPI = NoEpi>
__device__ __forceinline__ void gemm_tile(double (&smem)[2][2][GK][GP], const double *__restrict__ A, size_t lda,
                                          const double *__restrict__ B, size_t ldb, double *__restrict__ C,
                                          size_t ldc, int M, int N, int K, int ti, int tj, int dbg, int tid, EPI &&epi = EPI{})
{
    if (K % GK == 0 && !(dbg & 16)) gemm_tile_k<MODE, true>(smem, A, lda, B, ldb, C, ldc, M, N, K, ti, tj, dbg, tid, epi);  // workgroup-uniform
    else gemm_tile_k<MODE, false>(smem, A, lda, B, ldb, C, ldc, M, N, K, ti, tj, dbg, tid, epi);
}


// Two workgroups share a CU and its matrix pipes.  All tiles cost the same, so workgroups that
// start together stay in lock-step: both reach their memory-bound epilogue at the same time
// and the pipes idle.  Delaying the second-dispatched workgroup of each CU once, by about one
// epilogue, puts the pair in anti-phase for the rest of the launch: one streams its C tile
// while the other has the pipes to itself.
// stagger = (mode << 16) | sleeps; mode 1: blocks 256..511 (second dispatch round),
// mode 2: odd hardware wave slot of wave 0 (HW_REG_HW_ID[3:0]).
__device__ __forceinline__ void stagger_start(double (&smem)[2][2][GK][GP], int stagger)
{
    if ((stagger & 0xffffff) && blockIdx.x < 512) {
        bool late = false;
        if (((stagger >> 16) & 0xff) == 1) late = blockIdx.x >= 256;
        else {
            if (threadIdx.x == 0) smem[0][0][0][0] = (double)(__builtin_amdgcn_s_getreg((31 << 11) | 4) & 1);
            __syncthreads();
            late = smem[0][0][0][0] != 0.0;
            __syncthreads();
        }
        if (late)
            for (int i = 0; i < (stagger & 0xffff); ++i) __builtin_amdgcn_s_sleep(127);
    }
}

// Fused look-ahead: the workgroup of tile (0, 0) -- the 128 x 128 diagonal block of the NEXT panel
// whenever the update starts at the row of its first column -- factors that block as soon as its
// own tile is stored, while the other workgroups of the launch are still updating.  The
// diagonal-block latency chain (27 us per panel) then runs inside the update instead of after
// it, needs no launch and no free CU of its own, and the panel solve follows directly.
constexpr int SUBWG = 9;  // workgroups (x 4 waves = 36 sub-tiles of 16 x 16) of the sub-tiled diagonal tile of a fused launch
struct FuseDiag {
    double *Fp;   // packed-factor slot of that panel; nullptr: no fusion
    int *info;
    int col0;     // global column of the block (for info)
    int nb;       // active order of the block (<= 128)
    int *ctr;     // zeroed device counter for the sub-tiled variant (order bit 9)
};

// 16 x 16 sub-tile of C -= A B^T at (m0, n0), ONE wave, fragments straight from global memory with EVERY
// load of a 128-wide K range in flight at once (two sets of 16 k-groups = 64 loads per lane): the operands
// were written by the previous kernel on other CUs / XCDs and come from HBM or the Infinity Cache, so
// the cost of a sub-tile is memory round trips, not matrix work -- the 64 x 64 form above (8 k-groups in
// flight behind 8 being multiplied) took 3 dependent round trips at K = 128 (~11 us), this one takes one.
// Two accumulators halve the dependent MFMA chain.  Requires K % 64 == 0; the tile lies wholly inside C.
template <bool COH = false>
__device__ __forceinline__ void gemm_sub16(const double *__restrict__ A, size_t lda, const double *__restrict__ B, size_t ldb,
                                           double *__restrict__ C, size_t ldc, int K, int m0, int n0)
{
    const int lane = threadIdx.x & 63;
    const int lr = lane & 15, lq = lane >> 4;
    const double *pa = A + (size_t)(m0 + lr) + (size_t)lq * lda;  // bf = pa[k * lda]
    const double *pb = B + (size_t)(n0 + lr) + (size_t)lq * ldb;  // af = pb[k * ldb]
    d4 acc0 = d4{0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
    double fa[2][16], fb[2][16];
    // ONE running offset per operand, opaque to the optimiser: with per-load offsets it materialises all 64
    // addresses (128 registers) next to the 128 registers of loaded operands and spills the operands (the
    // offset, not the pointer, is made opaque: the loads stay global_load, not flat_load)
    const size_t sa = 4 * lda, sb = 4 * ldb;
    auto load = [&](int set, int k0) {
        size_t oa = (size_t)k0 * lda, ob = (size_t)k0 * ldb;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            fa[set][g] = pb[ob];
            fb[set][g] = pa[oa];
            oa += sa;
            ob += sb;
            asm volatile("" : "+v"(oa), "+v"(ob));
        }
    };
    auto mul = [&](int set) {
#pragma unroll
        for (int g = 0; g < 16; g += 2) {
            acc0 = mfma(fa[set][g], fb[set][g], acc0);
            acc1 = mfma(fa[set][g + 1], fb[set][g + 1], acc1);
        }
    };
    load(0, 0);
    if (K > 64) load(1, 64);
#pragma unroll 1
    for (int k0 = 0; k0 < K; k0 += 128) {
        mul(0);
        if (k0 + 128 < K) load(0, k0 + 128);
        if (k0 + 64 < K) {
            mul(1);
            if (k0 + 192 < K) load(1, k0 + 192);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double *q = C + (size_t)(m0 + lr) + (size_t)(n0 + lq + 4 * i) * ldc;
        const double v = *q - (acc0[i] + acc1[i]);
        if constexpr (COH) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *q = v;
    }
}

// 64 x 64 tile of C -= A B^T, LDS-staged: the quadrant kernel of the SYRK tail split.  Four waves
// x (2 x 2 MFMA tiles), k-step 16, two LDS stages filled through registers (16-B global loads,
// ds_write_b128; row pad 16 doubles: k and k + 1 on opposite bank halves as in the big tile).
// A, B point at the quadrant's first operand rows; mv / nv valid rows from there (clamped loads,
// dropped outputs).  K % 16 == 0.
constexpr int QP = 80;  // padded row of the quadrant's LDS image
__device__ __forceinline__ void gemm_quad64(double *__restrict__ sm, const double *__restrict__ A, size_t lda,
                                            const double *__restrict__ B, size_t ldb, double *__restrict__ C,
                                            size_t ldc, int K, int mv, int nv, int tid)
{
    double (*q)[2][GK][QP] = reinterpret_cast<double (*)[2][GK][QP]>(sm);  // [stage][op][k][row]
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int mb = (w & 1) * 32, nb = (w >> 1) * 32;
    // staging: thread handles the row pair r2 of k-rows kr and kr + 8 of both operands
    const int r2 = tid & 31, kr = tid >> 5;
    const bool fast = (mv >= 64) && (nv >= 64);
    int ra0 = 2 * r2, ra1 = 2 * r2 + 1, rb0 = ra0, rb1 = ra1;
    ra0 = ra0 < mv ? ra0 : mv - 1;
    ra1 = ra1 < mv ? ra1 : mv - 1;
    rb0 = rb0 < nv ? rb0 : nv - 1;
    rb1 = rb1 < nv ? rb1 : nv - 1;
    double2 va[2], vb[2];
    auto gload = [&](int k0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const size_t ca = (size_t)(k0 + kr + 8 * j) * lda, cb = (size_t)(k0 + kr + 8 * j) * ldb;
            if (fast) {
                va[j] = *reinterpret_cast<const double2 *>(A + 2 * r2 + ca);
                vb[j] = *reinterpret_cast<const double2 *>(B + 2 * r2 + cb);
            } else {
                va[j] = make_double2(A[ra0 + ca], A[ra1 + ca]);
                vb[j] = make_double2(B[rb0 + cb], B[rb1 + cb]);
            }
        }
    };
    auto swrite = [&](int st) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            *reinterpret_cast<double2 *>(&q[st][0][kr + 8 * j][2 * r2]) = va[j];
            *reinterpret_cast<double2 *>(&q[st][1][kr + 8 * j][2 * r2]) = vb[j];
        }
    };
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
    const int nk = K / GK;
    gload(0);
    swrite(0);
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        const int st = kt & 1;
        if (kt + 1 < nk) gload((kt + 1) * GK);
        if (mb < mv && nb < nv) {  // wave-uniform: a wave whose block lies outside the matrix only stages
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                double af[2], bf[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    af[t] = q[st][1][kk * 4 + lq][nb + t * 16 + lr];
                    bf[t] = q[st][0][kk * 4 + lq][mb + t * 16 + lr];
                }
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                    for (int tm = 0; tm < 2; ++tm)
                        if (mb + tm * 16 < mv && nb + tn * 16 < nv) acc[tn][tm] = mfma(af[tn], bf[tm], acc[tn][tm]);
            }
        }
        if (kt + 1 < nk) swrite(st ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = mb + tm * 16 + lr, n = nb + tn * 16 + lq + 4 * i;
                if (m < mv && n < nv) {
                    double *c = C + (size_t)m + (size_t)n * ldc;
                    *c = *c - acc[tn][tm][i];
                }
            }
}

// Tail split of a SYRK launch: the R < slots tiles of the last, partial round would hold R
// workgroup slots for a whole tile time while the rest of the chip idles.  Each of them is cut
// into its four 64 x 64 quadrants (blocks bfull + 4 t + q), computed by gemm_quad64: independent
// outputs, no exchange between workgroups, the same sums in the same order as timing-independent
// code must have.
struct KSplit {
    int bfull;   // blocks below this index are whole tiles
    int S;       // 4: quadrants; <= 1: no split
};

#ifdef GPMI_PROBES
// shader-clock probe: per workgroup of the SYRK kernel, elapsed shader cycles (s_memtime) and elapsed
// 100 MHz reference ticks (s_memrealtime), summed over the launch -> average clock and cycles per tile
__device__ unsigned long long g_clk[4];
__device__ unsigned long long g_fz[8];  // fused in-block launch, block 0: cycles in sub-tile, flag wait, diagonal body; launches
#endif

// Blocks 0 .. SUBWG - 1 of a sub-tiled fused launch (order bit 9): the 36 lower 16 x 16 sub-tiles of the
// diagonal tile (0, 0), one per wave; block 0 then waits for the other eight on fd.ctr and factors the block.
// The SUBWG blocks are the first of the grid -- dispatched together, so the wait cannot deadlock.
__device__ __forceinline__ void fused_subtiles_and_diag(double (&smem)[2][2][GK][GP], int b, const double *__restrict__ A, size_t lda,
                                                        const double *__restrict__ B, size_t ldb, double *__restrict__ C, size_t ldc,
                                                        int K, const FuseDiag &fd)
{
#ifdef GPMI_PROBES
    const unsigned long long fz0 = __builtin_amdgcn_s_memtime();
#endif
    {
        const int t = b * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // sub-tile index, row-major lower triangle
        int tm = 0;
        while ((tm + 1) * (tm + 2) / 2 <= t) ++tm;
        const int tn = t - tm * (tm + 1) / 2;
        gemm_sub16<true>(A, lda, B, ldb, C, ldc, K, tm * 16, tn * 16);
    }
    // the agent-scope stores above are complete (acknowledged by the memory side) once
    // vmcnt drains; no cache-wide fence is needed around this hand-over
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (b) {
        if (threadIdx.x == 0) atomicAdd(fd.ctr, 1);
        return;
    }
#ifdef GPMI_PROBES
    const unsigned long long fz1 = __builtin_amdgcn_s_memtime();
#endif
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(fd.ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < SUBWG - 1)
            __builtin_amdgcn_s_sleep(4);
        __hip_atomic_store(fd.ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // next launch is stream-ordered
    }
    __syncthreads();
#ifdef GPMI_PROBES
    const unsigned long long fz2 = __builtin_amdgcn_s_memtime();
#endif
    if (fd.nb == GPMI_NB) potrf_diag4_body<true, true>(&smem[0][0][0][0], C, ldc, fd.nb, fd.Fp, fd.info, fd.col0);
    else potrf_diag4_body<true, false>(&smem[0][0][0][0], C, ldc, fd.nb, fd.Fp, fd.info, fd.col0);
#ifdef GPMI_PROBES
    if (threadIdx.x == 0) {
        atomicAdd(&g_fz[0], fz1 - fz0);
        atomicAdd(&g_fz[1], fz2 - fz1);
        atomicAdd(&g_fz[2], __builtin_amdgcn_s_memtime() - fz2);
        atomicAdd(&g_fz[3], 1ull);
        atomicAdd(&g_fz[4], (unsigned long long)K);
        if (K == 128) {  // leaf launches: the body IS the critical path there
            atomicAdd(&g_fz[5], __builtin_amdgcn_s_memtime() - fz2);
            atomicAdd(&g_fz[6], fz1 - fz0);
            atomicAdd(&g_fz[7], 1ull);
        }
    }
#endif
}

template <int MODE>
__global__ __launch_bounds__(256, 2) void k_gemm_nt(const double *__restrict__ A, size_t lda,
                                                 const double *__restrict__ B, size_t ldb,
                                                 double *__restrict__ C, size_t ldc, int M, int N, int K, int order,
                                                 int stagger, FuseDiag fd, KSplit ks)
{
    static_assert(2 * 2 * GK * GP >= DIAG4_LDS, "the diagonal-block body runs in the staging buffer");
    __shared__ __attribute__((aligned(16))) double smem[2][2][GK][GP];
    // bits 24+ of `stagger` are probe-only switches (tools/syrk_bench.py): 1 = skip the C
    // epilogue, 2 = skip the operand streaming of the main loop; used for the cost breakdown
    // in DESIGN.md.  They exist in the probe build only; here they fold to nothing.
#ifdef GPMI_PROBES
    const int dbg = stagger >> 24;
    struct ClkProbe {
        unsigned long long t0, r0;
        __device__ ClkProbe() : t0(__builtin_amdgcn_s_memtime()), r0(__builtin_amdgcn_s_memrealtime()) {}
        __device__ ~ClkProbe()
        {
            if (MODE == 1 && threadIdx.x == 0) {
                atomicAdd(&g_clk[0], __builtin_amdgcn_s_memtime() - t0);
                atomicAdd(&g_clk[1], __builtin_amdgcn_s_memrealtime() - r0);
                atomicAdd(&g_clk[2], 1ull);
            }
        }
    } clk_probe;
#else
    constexpr int dbg = 0;
#endif
    stagger_start(smem, stagger);
    int ti, tj;
    if (MODE == 1) {
        int b = blockIdx.x;
        if (order & 0x200) {  // sub-tiled fused launch: tile 0 = the next diagonal block, by the first SUBWG blocks
            if (b < SUBWG) {
                fused_subtiles_and_diag(smem, b, A, lda, B, ldb, C, ldc, K, fd);
                return;
            }
            b -= SUBWG - 1;   // blocks SUBWG, ... are the tiles 1, 2, ...
        }
        const int T = (M + GT - 1) / GT;
        const int TN = (N + GT - 1) / GT < T ? (N + GT - 1) / GT : T;
        if ((order & 0xff) == 2) {  // XCD-partitioned band walk (multi-round launches)
            const int nt = TN * (TN + 1) / 2 + (T - TN) * TN;
            const int S = (nt + 7) >> 3;
            int c, quad = -1;  // quad < 0: the whole tile
            if (b < 8 * S) {
                const int s = b >> 3;
                c = (b & 7) * S + s;
                if (ks.S > 1 && s >= ks.bfull) quad = 0;  // last partial round of this XCD: quadrant 0 here, 1 .. 3 behind the grid
            } else {
                const int e = b - 8 * S, Rx = S - ks.bfull, t = e / 3;
                quad = 1 + e - 3 * t;
                c = (t / Rx) * S + ks.bfull + t % Rx;
            }
            if (c >= nt) return;
            syrk_tile_bands(c, T, TN, ti, tj);
            const bool thin = ti == T - 1 && T > 1 && M - ti * GT <= 64 && K % GK == 0;  // e.g. the augmented row alone
            if (quad >= 0 || thin) {
                for (int q = quad >= 0 ? quad : 0; q < (quad >= 0 ? quad + 1 : 4); ++q) {
                    const int qm = q & 1, qn = (q >> 1) & 1;
                    const int m0 = ti * GT + qm * 64, n0 = tj * GT + qn * 64;
                    if ((ti == tj && qn > qm) || m0 >= M || n0 >= N) continue;  // workgroup-uniform
                    gemm_quad64(&smem[0][0][0][0], A + m0, lda, B + n0, ldb, C + (size_t)m0 + (size_t)n0 * ldc, ldc, K, M - m0,
                                N - n0, (int)threadIdx.x);
                    __syncthreads();
                }
                return;
            }
        } else
        if (ks.S > 1 && b >= ks.bfull) {
            const int q = b - ks.bfull;
            if (!syrk_tile(ks.bfull + (q >> 2), T, TN, order & 0xff, ti, tj)) return;
            const int qm = q & 1, qn = (q >> 1) & 1;
            if (ti == tj && qn > qm) return;  // strictly upper quadrant of a diagonal tile
            const int m0 = ti * GT + qm * 64, n0 = tj * GT + qn * 64;
            if (m0 >= M || n0 >= N) return;
            gemm_quad64(&smem[0][0][0][0], A + m0, lda, B + n0, ldb, C + (size_t)m0 + (size_t)n0 * ldc, ldc, K, M - m0,
                        N - n0, (int)threadIdx.x);
            return;
        }
        if ((order & 0xff) != 2 && !syrk_tile(b, T, TN, order & 0xff, ti, tj)) return;
        // order bit 8: the panel is UPPER TRIANGULAR (P[i][k] = 0 for k < i, e.g. L^-T): the
        // products of tile (ti, tj), tj <= ti, start at column ti * 128 -- a third of the work of
        // the full update; row-major tile order runs the long tiles first
        if (order & 0x100) {
            const int k0 = ti * GT;
            A += (size_t)k0 * lda;
            B += (size_t)k0 * ldb;
            K -= k0;
        }
    } else if (MODE == 0 && (order & 0x200)) {
        // Sub-tiled fused launch (1-D grid): the diagonal tile (0, 0) sits on the critical path of
        // the panel chain and one workgroup needs K * 256 cycles for it, plus the memory latency of
        // operands the previous kernel has just written elsewhere.  Its lower triangle is cut into 36
        // 16 x 16 sub-tiles, one per wave of the first SUBWG = 9 workgroups (dispatched first, so
        // all are resident and the wait below cannot deadlock), each with all its operand loads in
        // flight at once; block 0 waits for the other eight and factors the block.  Blocks >= SUBWG
        // are the tiles 1, 2, ...
        const int b = blockIdx.x;
        if (b < SUBWG) {
            fused_subtiles_and_diag(smem, b, A, lda, B, ldb, C, ldc, K, fd);
            return;
        }
        const int gx = (M + GT - 1) / GT;
        ti = (b - (SUBWG - 1)) % gx;
        tj = (b - (SUBWG - 1)) / gx;
    } else if (MODE == 2 && (order & 0x400)) {
        // C = A B^T for a LOWER-triangular A (M x K, A[i][k] = 0 for k > i) and an UPPER-triangular B (N x K stored
        // with the n index contiguous, B[j][k] = 0 for k < j -- the transpose of a lower-triangular factor): the
        // product is lower triangular, only the tiles ti >= tj are computed (1-D grid, row-major lower triangle) and
        // tile (ti, tj) needs the columns [128 tj, 128 (ti + 1)) only -- a sixth of the full product's work
        if (!syrk_tile(blockIdx.x, (M + GT - 1) / GT, (M + GT - 1) / GT, 0, ti, tj)) return;
        const int k0 = tj * GT, k1 = (ti + 1) * GT < K ? (ti + 1) * GT : K;
        A += (size_t)k0 * lda;
        B += (size_t)k0 * ldb;
        K = k1 - k0;
    } else {
        ti = blockIdx.x;
        tj = blockIdx.y;
        if (MODE == 0 && (order & 0x100)) {  // longest tiles (small ti) first: grid is (tj, ti)
            ti = blockIdx.y;
            tj = blockIdx.x;
        }
        // order bit 8 (plain grid): A is upper-trapezoidal -- A[i][k] = 0 for i > k + koff, koff =
        // fd.col0 (the global column of A's first column; fd carries no fusion here) -- so the
        // products of row-tile ti start at its first non-zero column (X L^-T with triangular X)
        if (MODE == 0 && (order & 0x100)) {
            const int k0 = ti * GT - fd.col0;
            if (k0 > 0) {
                if (k0 >= K) return;
                A += (size_t)k0 * lda;
                B += (size_t)k0 * ldb;
                K -= k0;
            }
        }
    }
    gemm_tile<MODE>(smem, A, lda, B, ldb, C, ldc, M, N, K, ti, tj, dbg, (int)threadIdx.x);
    if (MODE != 2 && fd.Fp && !(order & 0x200) && ti == 0 && tj == 0) {  // workgroup-uniform
        // the tile's stores must be visible to the other waves of this workgroup, which read the
        // block back in the diagonal kernel's register layout
        __threadfence();
        __syncthreads();
        __threadfence();
        if (fd.nb == GPMI_NB) potrf_diag4_body<false, true>(&smem[0][0][0][0], C, ldc, fd.nb, fd.Fp, fd.info, fd.col0);
        else potrf_diag4_body<false, false>(&smem[0][0][0][0], C, ldc, fd.nb, fd.Fp, fd.info, fd.col0);
    }
}

#ifdef GPMI_PROBES  // A/B kernels of tools/ (libgpmi_probes.so); the product library ships without them
// ---------------------------------------------------------------------------
// GEMM NT v2: same 128x128 tile and LDS image, 8 waves (2 m x 4 n, 64 x 32 outputs per
// wave = 8 accumulator tiles) so that 64 VGPRs are free to PREFETCH the wave's share of the
// C tile while the main loop runs: the read-modify-write epilogue of v1 (4 dependent
// load->store round trips per tile, ~40 % of a K=256 tile's time) becomes subtract + store.
// Waits are counted by hand: the C loads are issued in four 8-load pieces AFTER the LDS-DMA
// of the next stage, so `s_waitcnt vmcnt(8)` retires the stage while the piece stays in
// flight across the raw s_barrier (hipcc's __syncthreads would drain everything).
// Register budget <= 168 (3 waves/SIMD): one workgroup per CU leaves a third wave slot per
// SIMD and 87 KB of LDS for the panel kernels of the look-ahead stream.
// ---------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(512, 2) void k_gemm8(const double *__restrict__ A, size_t lda,
                                                  const double *__restrict__ B, size_t ldb,
                                                  double *__restrict__ C, size_t ldc, int M, int N, int K, int order)
{
    __shared__ __attribute__((aligned(16))) double smem[2][2][GK][GP];
    int ti, tj;
    if (MODE == 1) {
        const int Tr = (M + GT - 1) / GT, Tc = (N + GT - 1) / GT;  // M = N + 1 (augmented row): a trapezoid
        if (!syrk_tile(blockIdx.x, Tr, Tc < Tr ? Tc : Tr, order, ti, tj)) return;
    } else {
        ti = blockIdx.x;
        tj = blockIdx.y;
    }
    const int m0 = ti * GT, n0 = tj * GT;
    if (MODE == 1 && n0 >= N) return;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int wm = w & 1, wn = w >> 1;
    // a tile wholly inside the matrix takes the branch-free path; for SYRK the diagonal tiles
    // are computed in full (their strictly-upper outputs land in the unused upper triangle)
    const bool interior = (m0 + GT <= M) && (n0 + GT <= N);

    // staging: waves 0-3 stream the A tile (m index), waves 4-7 the B tile (n index); 4 k-rows each
    const int op = w >> 2;
    const double *gsrc = (op ? B + n0 : A + m0) + 2 * lane;
    const size_t gld = op ? ldb : lda;
    const int krow0 = (w & 3) * 4;

    d4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};

    const int nk = (K + GK - 1) / GK;
    auto issue = [&](int stage, int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kr = krow0 + q;
            int kc = k0 + kr;
            kc = kc < K ? kc : K - 1;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(gsrc + (size_t)kc * gld),
                (__attribute__((address_space(3))) void *)&smem[stage][op][kr][0], 16, 0, 0);
        }
    };

    // this lane's corner of the C tile; element (tn, tm, i) = cbase[tm*16 + (tn*16 + 4 i) * ldc]
    double *const cbase = C + (size_t)(m0 + wm * 64 + lr) + (size_t)(n0 + wn * 32 + lq) * ldc;
    double cv[2][4][4];
    auto prefetch = [&](auto pc) {  // piece P: tn = P>>1, tm in {2(P&1), 2(P&1)+1}: 8 loads
        constexpr int P = decltype(pc)::value;
        constexpr int tn = P >> 1;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                constexpr int dummy = 0;
                (void)dummy;
                const int tm = 2 * (P & 1) + h;
                cv[tn][tm][i] = cbase[tm * 16 + (size_t)(tn * 16 + 4 * i) * ldc];
            }
    };
    auto compute = [&](int st, int klim) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int kr = kk * 4 + lq;
            const bool kv = kr < klim;
            double af[2], bf[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const double a = smem[st][1][kr][wn * 32 + t * 16 + lr];
                af[t] = kv ? a : 0.0;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double b = smem[st][0][kr][wm * 64 + t * 16 + lr];
                bf[t] = kv ? b : 0.0;
            }
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) acc[tn][tm] = mfma(af[tn], bf[tm], acc[tn][tm]);
        }
    };
    // one k-step: retire stage kt (leaving WAITN younger loads in flight), barrier, start the
    // DMA of stage kt+1, optionally issue C-prefetch piece P behind it, multiply stage kt
    auto step = [&](int kt, auto wn_, auto pc) {
        constexpr int WAITN = decltype(wn_)::value;
        constexpr int P = decltype(pc)::value;
        // lgkmcnt(0): every LDS read of the stage about to be overwritten has returned before
        // any wave can start the DMA into it (WAR across the barrier)
        if constexpr (WAITN == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 1 < nk) issue((kt + 1) & 1, (kt + 1) * GK);
        if constexpr (P >= 0) prefetch(ic<P>{});
        compute(kt & 1, K - kt * GK);
    };

    issue(0, 0);
    const bool pf = (MODE != 2) && interior;
    int kt = 0;
    if (pf && nk >= 5) {
        step(0, ic<0>{}, ic<0>{});
        step(1, ic<8>{}, ic<1>{});
        step(2, ic<8>{}, ic<2>{});
        step(3, ic<8>{}, ic<3>{});
        step(4, ic<8>{}, ic<-1>{});
        kt = 5;
    } else if (pf) {
        prefetch(ic<0>{}); prefetch(ic<1>{}); prefetch(ic<2>{}); prefetch(ic<3>{});
    }
#pragma unroll 1
    for (; kt < nk; ++kt) step(kt, ic<0>{}, ic<-1>{});

    if (interior) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    double *p = cbase + tm * 16 + (size_t)(tn * 16 + 4 * i) * ldc;
                    *p = (MODE == 2) ? acc[tn][tm][i] : cv[tn][tm][i] - acc[tn][tm][i];
                }
    } else {
        // edge tile: loads from clamped (always valid) addresses, guarded stores
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            double ce[4][4];
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int n = n0 + wn * 32 + tn * 16 + lq + 4 * i, m = m0 + wm * 64 + tm * 16 + lr;
                    n = n < N ? n : N - 1;
                    m = m < M ? m : M - 1;
                    if (MODE != 2) ce[tm][i] = C[(size_t)m + (size_t)n * ldc];
                }
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = n0 + wn * 32 + tn * 16 + lq + 4 * i, m = m0 + wm * 64 + tm * 16 + lr;
                    if (m < M && n < N && (MODE != 1 || n <= m))
                        C[(size_t)m + (size_t)n * ldc] = (MODE == 2) ? acc[tn][tm][i] : ce[tm][i] - acc[tn][tm][i];
                }
        }
    }
}

// ---------------------------------------------------------------------------
// GEMM NT v3: v2's wave layout and C prefetch with a 3-buffer LDS ring, DMA issued TWO
// k-steps ahead.  A stage has two full k-steps (~8k cycles) to land, so the counted wait at
// the top of a step does not stall on HBM / Infinity-Cache latency.  Issue order per step:
// [DMA stage kt+2] [C piece kt]; the wait lets everything younger than stage kt stay in
// flight (4 DMA + up to two 8-load pieces).  One workgroup per CU (110 KB LDS).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void wait_vm_lgkm0(int n)
{
    // immediates only; n is wave-uniform.  lgkmcnt(0): the buffer the next DMA overwrites was
    // read in the previous step -- those LDS reads must have returned before any wave passes
    // the barrier and starts that DMA.
    if (n >= 20) asm volatile("s_waitcnt vmcnt(20) lgkmcnt(0)" ::: "memory");
    else if (n >= 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
    else if (n >= 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

template <int MODE>
__global__ __launch_bounds__(512, 2) void k_gemm9(const double *__restrict__ A, size_t lda,
                                                  const double *__restrict__ B, size_t ldb,
                                                  double *__restrict__ C, size_t ldc, int M, int N, int K, int order)
{
    __shared__ __attribute__((aligned(16))) double smem[3][2][GK][GP];
    int ti, tj;
    if (MODE == 1) {
        const int Tr = (M + GT - 1) / GT, Tc = (N + GT - 1) / GT;  // M = N + 1 (augmented row): a trapezoid
        if (!syrk_tile(blockIdx.x, Tr, Tc < Tr ? Tc : Tr, order, ti, tj)) return;
    } else {
        ti = blockIdx.x;
        tj = blockIdx.y;
    }
    const int m0 = ti * GT, n0 = tj * GT;
    if (MODE == 1 && n0 >= N) return;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int wm = w & 1, wn = w >> 1;
    const bool interior = (m0 + GT <= M) && (n0 + GT <= N);

    const int op = w >> 2;
    const double *gsrc = (op ? B + n0 : A + m0) + 2 * lane;
    const size_t gld = op ? ldb : lda;
    const int krow0 = (w & 3) * 4;

    d4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};

    const int nk = (K + GK - 1) / GK;
    auto issue = [&](int stage, int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kr = krow0 + q;
            int kc = k0 + kr;
            kc = kc < K ? kc : K - 1;
            __builtin_amdgcn_global_load_lds(
                (const __attribute__((address_space(1))) void *)(gsrc + (size_t)kc * gld),
                (__attribute__((address_space(3))) void *)&smem[stage][op][kr][0], 16, 0, 0);
        }
    };

    double *const cbase = C + (size_t)(m0 + wm * 64 + lr) + (size_t)(n0 + wn * 32 + lq) * ldc;
    double cv[2][4][4];
    auto prefetch = [&](auto pc) {
        constexpr int P = decltype(pc)::value;
        constexpr int tn = P >> 1;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int tm = 2 * (P & 1) + h;
                cv[tn][tm][i] = cbase[tm * 16 + (size_t)(tn * 16 + 4 * i) * ldc];
            }
    };
    auto compute = [&](int st, int klim) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int kr = kk * 4 + lq;
            const bool kv = kr < klim;
            double af[2], bf[4];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const double a = smem[st][1][kr][wn * 32 + t * 16 + lr];
                af[t] = kv ? a : 0.0;
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const double b = smem[st][0][kr][wm * 64 + t * 16 + lr];
                bf[t] = kv ? b : 0.0;
            }
#pragma unroll
            for (int tn = 0; tn < 2; ++tn)
#pragma unroll
                for (int tm = 0; tm < 4; ++tm) acc[tn][tm] = mfma(af[tn], bf[tm], acc[tn][tm]);
        }
    };

    const bool pf = (MODE != 2) && interior;
    int st = 0, st2 = 2;  // buffer of stage kt, buffer of stage kt+2
    // one k-step with compile-time wait count and prefetch piece (straight-line code: hipcc
    // then keeps the prefetched C values in place instead of copying them at branch joins)
    auto step = [&](int kt, auto wn_, auto pc) {
        constexpr int WAITN = decltype(wn_)::value;
        constexpr int P = decltype(pc)::value;
        wait_vm_lgkm0(WAITN);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (kt + 2 < nk) issue(st2, (kt + 2) * GK);
        if constexpr (P >= 0) prefetch(ic<P>{});
        compute(st, K - kt * GK);
        st = (st == 2) ? 0 : st + 1;
        st2 = (st2 == 2) ? 0 : st2 + 1;
    };
    issue(0, 0);
    if (nk > 1) issue(1, GK);
    int kt = 0;
    if (pf && nk >= 8) {
        // younger than stage kt and allowed to stay in flight: DMA kt+1 (4) + the C pieces of
        // steps kt-1 and kt-2 (8 each)
        step(0, ic<4>{}, ic<0>{});
        step(1, ic<12>{}, ic<1>{});
        step(2, ic<20>{}, ic<2>{});
        step(3, ic<20>{}, ic<3>{});
        step(4, ic<20>{}, ic<-1>{});
        step(5, ic<12>{}, ic<-1>{});
        kt = 6;
    } else if (pf) {
        prefetch(ic<0>{}); prefetch(ic<1>{}); prefetch(ic<2>{}); prefetch(ic<3>{});
        step(0, ic<0>{}, ic<-1>{});  // drains the C loads too (short-K path)
        kt = 1;
    }
#pragma unroll 1
    for (; kt + 1 < nk; ++kt) step(kt, ic<4>{}, ic<-1>{});
    if (kt < nk) step(kt, ic<0>{}, ic<-1>{});

    if (interior) {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    double *p = cbase + tm * 16 + (size_t)(tn * 16 + 4 * i) * ldc;
                    *p = (MODE == 2) ? acc[tn][tm][i] : cv[tn][tm][i] - acc[tn][tm][i];
                }
    } else {
#pragma unroll
        for (int tn = 0; tn < 2; ++tn) {
            double ce[4][4];
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    int n = n0 + wn * 32 + tn * 16 + lq + 4 * i, m = m0 + wm * 64 + tm * 16 + lr;
                    n = n < N ? n : N - 1;
                    m = m < M ? m : M - 1;
                    if (MODE != 2) ce[tm][i] = C[(size_t)m + (size_t)n * ldc];
                }
#pragma unroll
            for (int tm = 0; tm < 4; ++tm)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int n = n0 + wn * 32 + tn * 16 + lq + 4 * i, m = m0 + wm * 64 + tm * 16 + lr;
                    if (m < M && n < N && (MODE != 1 || n <= m))
                        C[(size_t)m + (size_t)n * ldc] = (MODE == 2) ? acc[tn][tm][i] : ce[tm][i] - acc[tn][tm][i];
                }
        }
    }
}

#endif  // GPMI_PROBES

// ---------------------------------------------------------------------------
// Small problems: ONE workgroup does a whole evaluation.
//
// The reference's drivers call the path at N = 21 (R/tests.R:5-19), 79 .. 199 (pendulum_fit*.R:206-214) and
// 256 (BASELINE c1); there the chain of launches of the blocked code (build, row, diagonal block, panel solve,
// update, ..., two finalize kernels) is pure launch latency.  Here one workgroup of 4 waves runs the same
// device functions back to back on a matrix that never leaves its CU's L2:
//   build (se_cov_tile: the arithmetic of k_se_cov, bit-identical K)  ->  for every 128-column panel:
//   potrf_diag4_body (packed factors to LDS)  ->  rows below by trsm_panel_body strips  ->  trailing tiles by
//   gemm_tile<1> / gemm_quad64  ->  log-det and quadratic form, reduced in the order of k_logml_partial.
// The right-hand side y rides along as row n; when it is the only row below the last panel
// its solve is a VALU forward substitution from the packed factors in LDS (two barriers per 16 pivots) instead
// of a 144-MFMA strip.  Phase boundaries are __syncthreads(): global memory written by one wave of a workgroup
// is visible to the others behind the barrier (one CU, one L1).  A grid of G hyper-parameter points is G
// workgroups of ONE launch (k_logml_small_batch), each with its own workspace slice.
// ---------------------------------------------------------------------------
#ifdef GPMI_PROBES  // phase stamps of the small kernels (block 0, thread 0): g_fz[0] build, [1] diagonal blocks, [2] rows below, [3] launches, [4] trailing tiles, [5] finalize
#define GPMI_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime();
#define GPMI_STAMP_ADD(i, d) if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(&g_fz[i], (unsigned long long)(d));
#else
#define GPMI_STAMP(v)
#define GPMI_STAMP_ADD(i, d)
#endif
constexpr int FIN_SLICE = 256;  // slice of the diagonal one workgroup of k_logml_partial reduces
struct SmallSe {            // hyper-parameters of one point, in registers
    double a2;
    double inv_ell[GPMI_MAXD];
    int D;
};

// z = L11^-1 r for ONE right-hand row (W[row, 0 .. nb), stride ld) against the packed factors of a <= 128-order
// block in LDS: per 16-pivot block, z_kb = Linv16[kb] r_kb by 16 threads, then every row below subtracts
// L[r][kb] z_kb -- -L tiles and inverses in the fragment order potrf_diag4_body packs them in.
__device__ __forceinline__ void small_row_solve(const double *__restrict__ s_F, double *__restrict__ s_r,
                                                double *__restrict__ s_z, double *__restrict__ Wrow, size_t ld, int nb, int t)
{
    const int nblk = (nb + 15) >> 4;
    if (t < 128) s_r[t] = (t < nb) ? Wrow[(size_t)t * ld] : 0.0;
    __syncthreads();
    // every LDS read of a step is issued before its first use (fully unrolled, two accumulators): a loop with a
    // per-lane trip count made each of the 16 products wait for its own pair of reads (3 k cycles per step)
    for (int kb = 0; kb < nblk; ++kb) {
        if (t < 16) {
            double f[16], r[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                f[c] = s_F[fp_inv(kb) * 256 + (c >> 2) * 64 + (c & 3) * 16 + t];  // Linv16[t][c], zero above the diagonal
                r[c] = s_r[kb * 16 + c];
            }
            double z0 = 0.0, z1 = 0.0;
#pragma unroll
            for (int c = 0; c < 16; c += 2) {
                z0 = fma(f[c], r[c], z0);
                z1 = fma(f[c + 1], r[c + 1], z1);
            }
            s_z[kb * 16 + t] = z0 + z1;
        }
        __syncthreads();
        if (t < 128 && t >= (kb + 1) * 16 && t < nblk * 16) {
            const int jb = t >> 4, lr = t & 15;
            double f[16], z[16];
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                f[c] = s_F[fp_l(jb, kb) * 256 + (c >> 2) * 64 + (c & 3) * 16 + lr];  // -L[t][16 kb + c]
                z[c] = s_z[kb * 16 + c];
            }
            double a0 = s_r[t], a1 = 0.0;
#pragma unroll
            for (int c = 0; c < 16; c += 2) {
                a0 = fma(f[c], z[c], a0);
                a1 = fma(f[c + 1], z[c + 1], a1);
            }
            s_r[t] = a0 + a1;
        }
        __syncthreads();
    }
    if (t < nb) Wrow[(size_t)t * ld] = s_z[t];
}

// Right-looking partial factorisation by ONE workgroup: the first nfac columns of the M x ncol lower trapezoid in
// W are factored, rows below / the trailing block updated (launch_potrf_partial's contract).  one_row: the caller
// promises M == ncol + 1 == nfac + 1 (one augmented row), which lets the last panel use small_row_solve.
// WITH_U (value + gradient kernels): U (same leading dimension, holds the identity on entry) becomes L^-T, block column
// by block column while that panel's packed factors are in LDS: column block k of U holds I - sum_{j < k} U[:, j] L[k, j]^T
// in its rows [0, k + nb) when panel k has been factored; the panel's strips apply L_kk^-T, and the panel's rows below
// (solved next) take the block out of the later column blocks in one product each.
template <bool WITH_U = false>
__device__ __forceinline__ void small_potrf_partial(double (&smem)[2][2][GK][GP], double *__restrict__ s_F,
                                                    double *__restrict__ s_aux, double *__restrict__ W, size_t ld, int M,
                                                    int ncol, int nfac, int *info, bool one_row, double *__restrict__ U = nullptr)
{
    const size_t ld0 = ld;
    // thread index, re-read behind an optimisation barrier in front of every phase: everything a phase derives from it
    // (LDS addresses, lane masks, column offsets: hundreds of values) is then computed where it is used instead of
    // being hoisted in front of the panel loop and kept in scratch memory across all phases
    auto fresh_tid = []() {
        int t = (int)threadIdx.x;
        asm volatile("" : "+v"(t));
        return t;
    };
    for (int k = 0; k < nfac; k += GPMI_NB) {
        const int nb = (nfac - k < GPMI_NB) ? nfac - k : GPMI_NB;
        // the leading dimension is made opaque per panel: otherwise every per-element offset of every phase (hundreds of
        // 64-bit values) is hoisted out of this loop and stays live across all phases -- the kernel then needs 512
        // registers, copies values through AGPRs around factor16's hand-scheduled DPP chain and breaks its hazard
        // assumptions (the hazard recogniser cannot see into the asm statements)
        size_t ld = ld0;
        asm volatile("" : "+s"(ld));
        double *Akk = W + (size_t)k + (size_t)k * ld;
        GPMI_STAMP(ts0)
        if (nb == GPMI_NB) potrf_diag4_body<false, true>(&smem[0][0][0][0], Akk, ld, nb, s_F, info, k, 8, fresh_tid());
        else potrf_diag4_body<false, false>(&smem[0][0][0][0], Akk, ld, nb, s_F, info, k, (nb + 15) >> 4, fresh_tid());
        __syncthreads();
        GPMI_STAMP(ts1)
        GPMI_STAMP_ADD(1, ts1 - ts0)
        if constexpr (WITH_U) {
            for (int rb = 0; rb < k + nb; rb += 64) {
                const int tid = fresh_tid();
                const int r = rb + (tid >> 6) * 16 + (tid & 15);
                // rows inside this panel are rows of the identity: zero left of their own 16-column block
                const int rw = rb + (tid >> 6) * 16 - k;
                const int kb0 = __builtin_amdgcn_readfirstlane(rw > 0 ? rw >> 4 : 0);
                if (nb == GPMI_NB && rb + 64 <= k + nb) trsm_panel_body<true>(s_F, U + (size_t)k * ld, ld, r, true, nb, tid, kb0);
                else trsm_panel_body<false>(s_F, U + (size_t)k * ld, ld, r, r < k + nb, nb, tid, kb0);
            }
            __syncthreads();
        }
        const int r0 = k + nb;
        if (r0 >= M) break;
        if (one_row && M - r0 == 1) {
            small_row_solve(s_F, s_aux, s_aux + 128, W + (size_t)r0 + (size_t)k * ld, ld, nb, fresh_tid());
            __syncthreads();
            GPMI_STAMP(ts2)
            GPMI_STAMP_ADD(2, ts2 - ts1)
            break;
        }
        // rows [r0, M): 16-row strips, one per wave, 64 rows per round
        for (int rb = r0; rb < M; rb += 64) {
            const int tid = fresh_tid();
            const int r = rb + (tid >> 6) * 16 + (tid & 15);
            if (nb == GPMI_NB && rb + 64 <= M) trsm_panel_body<true>(s_F, W + (size_t)k * ld, ld, r, true, nb, tid);
            else trsm_panel_body<false>(s_F, W + (size_t)k * ld, ld, r, r < M, nb, tid);
        }
        __syncthreads();
        GPMI_STAMP(ts2)
        GPMI_STAMP_ADD(2, ts2 - ts1)
        // trailing block: C[r0.., r0..ncol) -= X X^T, lower tiles
        const int mt = M - r0, nt = ncol - r0;
        if constexpr (WITH_U) {
            if (nt > 0) {  // U[0 : r0, r0 : ncol) -= U[0 : r0, k : r0) L[r0 : ncol, k : r0)^T
                for (int ti = 0; ti * GT < r0; ++ti)
                    for (int tj = 0; tj * GT < nt; ++tj) {
                        gemm_tile<0>(smem, U + (size_t)k * ld, ld, W + (size_t)r0 + (size_t)k * ld, ld, U + (size_t)r0 * ld, ld, r0, nt,
                                     nb, ti, tj, 0, fresh_tid());
                        __syncthreads();
                    }
            }
        }
        if (nt <= 0) continue;
        const double *X = W + (size_t)r0 + (size_t)k * ld;
        double *C = W + (size_t)r0 + (size_t)r0 * ld;
        const int T = (mt + GT - 1) / GT, TN = (nt + GT - 1) / GT;
        for (int ti = 0; ti < T; ++ti) {
            const int vr = (mt - ti * GT < GT) ? mt - ti * GT : GT;  // valid rows of this tile row
            if (vr == 1) {
                // ONE row below the square part (the augmented row y^T when the order is a multiple of 128): its update is
                // nt dot products of length nb -- thread = column, the row's panel entries from LDS, eight loads in flight --
                // instead of a 64 x 64 MFMA quadrant per 64 columns (8 k cycles each for one useful row)
                const int tid = fresh_tid(), row = ti * GT;
                __syncthreads();
                if (tid < nb) s_aux[tid] = X[(size_t)row + (size_t)tid * ld];
                __syncthreads();
                for (int j = tid; j < nt && j <= row; j += 256) {
                    double a0 = 0.0, a1 = 0.0;
                    for (int k0 = 0; k0 < nb; k0 += 8) {
                        double u[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) u[q] = X[(size_t)j + (size_t)(k0 + q < nb ? k0 + q : nb - 1) * ld];
#pragma unroll
                        for (int q = 0; q < 8; q += 2) {
                            a0 = fma(u[q], (k0 + q < nb) ? s_aux[k0 + q] : 0.0, a0);
                            a1 = fma(u[q + 1], (k0 + q + 1 < nb) ? s_aux[k0 + q + 1] : 0.0, a1);
                        }
                    }
                    C[(size_t)row + (size_t)j * ld] -= a0 + a1;
                }
                __syncthreads();
                continue;
            }
            for (int tj = 0; tj <= ti && tj < TN; ++tj) {
                if (vr <= 64 && nb % GK == 0) {  // thin tile row (e.g. the augmented row alone): 64-row quadrants
                    for (int qn = 0; qn < 2; ++qn) {
                        const int m0 = ti * GT, n0 = tj * GT + qn * 64;
                        if (n0 >= nt || (ti == tj && qn > 0)) continue;
                        gemm_quad64(&smem[0][0][0][0], X + m0, ld, X + n0, ld, C + (size_t)m0 + (size_t)n0 * ld, ld, nb, mt - m0,
                                    nt - n0, fresh_tid());
                        __syncthreads();
                    }
                } else {
                    gemm_tile<1>(smem, X, ld, X, ld, C, ld, mt, nt, nb, ti, tj, 0, fresh_tid());
                    __syncthreads();
                }
            }
        }
        GPMI_STAMP(ts3)
        GPMI_STAMP_ADD(4, ts3 - ts2)
    }
}

// One evaluation of models/fit_hyperparameters.stan:18-32 at n <= SMALL_N_MAX by one workgroup.
__device__ __forceinline__ void logml_small_body(double (&smem)[2][2][GK][GP], double *__restrict__ s_F, double *__restrict__ s_aux,
                                                 const double *__restrict__ X, int n, int ldx, const double *__restrict__ y,
                                                 const SmallSe &se, double diag_add, double *__restrict__ W, size_t ld,
                                                 double *__restrict__ out3, int *info_out, int *info_w, const ExpC &ec)
{
    const int tid = threadIdx.x;
    GPMI_STAMP(tb0)
    if (tid == 0) *info_w = 0;
    // covariance, lower 64 x 64 tiles, from the scaled coordinates staged ONCE in LDS (the staging buffer of the later
    // phases is free): one global round trip instead of one per tile and operand; y^T as row n
    {
        double *xs = &smem[0][0][0][0];
#pragma unroll
        for (int d = 0; d < GPMI_MAXD; ++d)
            if (d < se.D)
                for (int i = tid; i < n; i += 256) xs[i + d * n] = __dmul_rn(X[(size_t)i + (size_t)d * ldx], se.inv_ell[d]);
        __syncthreads();
        for (int row0 = 0; row0 < n; row0 += SE_TR)
            for (int col0 = 0; col0 < row0 + SE_TR && col0 < n; col0 += SE_TC) {
                switch (se.D) {
                case 1: se_cov_tile<1, true>(xs, n, n, xs, n, n, se, diag_add, 1, 1, W, ld, 1, ec, row0, col0); break;
                case 2: se_cov_tile<2, true>(xs, n, n, xs, n, n, se, diag_add, 1, 1, W, ld, 1, ec, row0, col0); break;
                case 3: se_cov_tile<3, true>(xs, n, n, xs, n, n, se, diag_add, 1, 1, W, ld, 1, ec, row0, col0); break;
                default: se_cov_tile<0, true>(xs, n, n, xs, n, n, se, diag_add, 1, 1, W, ld, 1, ec, row0, col0); break;
                }
            }
    }
    for (int j = tid; j < n; j += 256) W[(size_t)n + (size_t)j * ld] = y[j];
    __syncthreads();
    GPMI_STAMP(tb1)
    GPMI_STAMP_ADD(0, tb1 - tb0)
    GPMI_STAMP_ADD(3, 1)
    small_potrf_partial(smem, s_F, s_aux, W, ld, n + 1, n, n, info_w, true);
    GPMI_STAMP(tb2)
    // sum log L_ii, z'z: the reduction tree of k_logml_partial (slices of 256, thread `slice` keeps its sum) and, for more
    // than one slice, of k_logml_finalize over the slice sums -- the same additions in the same order as the blocked path
    double *s_a = s_aux, *s_b = s_aux + 256;
    const int nslice = (n + FIN_SLICE - 1) / FIN_SLICE;
    double pa = 0.0, pb = 0.0;
    for (int sl = 0; sl < nslice; ++sl) {
        const int i = sl * FIN_SLICE + tid;
        double a = 0.0, b = 0.0;
        if (i < n) {
            a = log(W[(size_t)i * (ld + 1)]);
            const double z = W[(size_t)n + (size_t)i * ld];
            b = z * z;
        }
        s_a[tid] = a;
        s_b[tid] = b;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) {
                s_a[tid] += s_a[tid + st];
                s_b[tid] += s_b[tid + st];
            }
            __syncthreads();
        }
        if (nslice > 1) {
            if (tid == sl) {
                pa = s_a[0];
                pb = s_b[0];
            }
            __syncthreads();
        }
    }
    if (nslice > 1) {
        s_a[tid] = pa;
        s_b[tid] = pb;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) {
                s_a[tid] += s_a[tid + st];
                s_b[tid] += s_b[tid + st];
            }
            __syncthreads();
        }
    }
    if (tid == 0) {
        const int info = __hip_atomic_load(info_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (info_out) *info_out = info;
        if (info) {
            out3[0] = out3[1] = out3[2] = __builtin_nan("");
        } else {
            out3[1] = s_a[0];
            out3[2] = s_b[0];
            out3[0] = -0.5 * s_b[0] - s_a[0] - 0.5 * (double)n * 1.8378770664093454835606594728112;  // log(2 pi)
        }
    }
    GPMI_STAMP(tb3)
    GPMI_STAMP_ADD(5, tb3 - tb2)
}

// Value AND gradient sums of models/fit_hyperparameters.stan:18-32 by ONE workgroup (n <= 256, D <= GPMI_MAXD): what one
// leapfrog step of NUTS asks for at the sizes the reference's fits run (R/tests.R:5 N = 21, pendulum_fit.R 79 .. 199), where
// the launch chain of gpmi_logml_grad (factorisation, identity, L^-T, K^-1, contraction: ~25 launches + 4 copies) costs 150 us.
// d logml / d theta = 1/2 tr((a a' - K^-1) dK/dtheta): U = L^-T rides along in the factorisation (small_potrf_partial<true>),
// a = U z, K^-1 = U U^T by gemm_tile<2> over the lower tiles (only the columns >= the tile row's first: U is upper
// triangular), and the contraction re-evaluates the kernel from the scaled coordinates in LDS, thread = row.
// res: [0..2] logml, sum log L_ii, z'z; [3 + s] the contraction sums in the layout of k_grad_partial (GRAD_NS = 10 slots:
// [0] sum c, [1 + d] sum c (x_id - x_jd)^2, [9] sum_i g_ii) -- the host turns them into the gradient as for the chain.
constexpr int SMALL_GRAD_NS = 2 + GPMI_MAXD, SMALL_GRAD_RES = 3 + SMALL_GRAD_NS;
__device__ __forceinline__ void logml_grad_small_body(double (&smem)[2][2][GK][GP], double *__restrict__ s_F, double *__restrict__ s_aux,
                                                      const double *__restrict__ X, int n, int ldx, const double *__restrict__ y,
                                                      const SmallSe &se, double diag_add, double *__restrict__ W, size_t ld,
                                                      double *__restrict__ U, double *__restrict__ res, int *info_out, int *info_w,
                                                      const ExpC &ec)
{
    const int tid = threadIdx.x;
    GPMI_STAMP(tg0)
    if (tid == 0) *info_w = 0;
    double *xs = &smem[0][0][0][0];
    auto stage_x = [&]() {
#pragma unroll
        for (int d = 0; d < GPMI_MAXD; ++d)
            if (d < se.D)
                for (int i = tid; i < n; i += 256) xs[i + d * n] = __dmul_rn(X[(size_t)i + (size_t)d * ldx], se.inv_ell[d]);
        __syncthreads();
    };
    stage_x();
    for (int row0 = 0; row0 < n; row0 += SE_TR)
        for (int col0 = 0; col0 < row0 + SE_TR && col0 < n; col0 += SE_TC) {
            switch (se.D) {
            case 1: se_cov_tile<1, true>(xs, n, n, xs, n, n, se, diag_add, 1, 1, W, ld, 1, ec, row0, col0); break;
            case 2: se_cov_tile<2, true>(xs, n, n, xs, n, n, se, diag_add, 1, 1, W, ld, 1, ec, row0, col0); break;
            case 3: se_cov_tile<3, true>(xs, n, n, xs, n, n, se, diag_add, 1, 1, W, ld, 1, ec, row0, col0); break;
            default: se_cov_tile<0, true>(xs, n, n, xs, n, n, se, diag_add, 1, 1, W, ld, 1, ec, row0, col0); break;
            }
        }
    for (int j = tid; j < n; j += 256) W[(size_t)n + (size_t)j * ld] = y[j];
    {   // U = I: 16-byte stores (row pair of a thread; ld is even and the slice 16-byte aligned), then the diagonal
        const int rp = 2 * (tid & 127), cp = tid >> 7;
        if (rp < n)
            for (int j = cp; j < n; j += 2) *reinterpret_cast<double2 *>(U + (size_t)rp + (size_t)j * ld) = make_double2(0.0, 0.0);
        __syncthreads();
        for (int i = tid; i < n; i += 256) U[(size_t)i * (ld + 1)] = 1.0;
    }
    __syncthreads();
    GPMI_STAMP(tg1)
    GPMI_STAMP_ADD(0, tg1 - tg0)
    GPMI_STAMP_ADD(3, 1)
    small_potrf_partial<true>(smem, s_F, s_aux, W, ld, n + 1, n, n, info_w, true, U);
    __syncthreads();
    GPMI_STAMP(tg2)
    // value: one slice (n <= 256), the tree of k_logml_partial
    double *s_a = s_aux, *s_b = s_aux + 256, *s_z = s_aux + 512, *s_av = s_aux + 768;
    {
        double a = 0.0, b = 0.0, z = 0.0;
        if (tid < n) {
            a = log(W[(size_t)tid * (ld + 1)]);
            z = W[(size_t)n + (size_t)tid * ld];
            b = z * z;
        }
        s_a[tid] = a;
        s_b[tid] = b;
        s_z[tid] = z;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) {
                s_a[tid] += s_a[tid + st];
                s_b[tid] += s_b[tid + st];
            }
            __syncthreads();
        }
    }
    const double sum_log = s_a[0], zz = s_b[0];
    // a = U z = K^-1 y (U upper triangular: the columns left of a wave's first row are zero); sixteen loads in flight per
    // round trip -- a loop with one load per iteration is a chain of n memory latencies
    {
        double acc0 = 0.0, acc1 = 0.0;
        const int ir = tid < n ? tid : n - 1;
        for (int j0 = tid & ~63; j0 < n; j0 += 16) {
            double u[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int j = j0 + q < n ? j0 + q : n - 1;
                u[q] = U[(size_t)ir + (size_t)j * ld];
            }
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                acc0 = fma(u[q], (j0 + q < n) ? s_z[j0 + q] : 0.0, acc0);
                acc1 = fma(u[q + 1], (j0 + q + 1 < n) ? s_z[j0 + q + 1] : 0.0, acc1);
            }
        }
        s_av[tid] = acc0 + acc1;
    }
    // scaled coordinates for the contraction: in the packed-factor buffer (free now; the staging buffer is the product's)
    double *xg = s_F;
#pragma unroll
    for (int d = 0; d < GPMI_MAXD; ++d)
        if (d < se.D)
            for (int i = tid; i < n; i += 256) xg[i + d * n] = __dmul_rn(X[(size_t)i + (size_t)d * ldx], se.inv_ell[d]);
    __syncthreads();
    GPMI_STAMP(tg3)
    GPMI_STAMP_ADD(5, tg3 - tg2)
    // K^-1 = U U^T tile by tile (lower tiles; only the columns >= the tile row's first), contracted where it is produced:
    // every element (m, n <= m) of a tile goes from its accumulator register into the sums -- K^-1 is never stored
    double acc[SMALL_GRAD_NS];
#pragma unroll
    for (int q = 0; q < SMALL_GRAD_NS; ++q) acc[q] = 0.0;
    const double a2 = se.a2;
    auto kinv_tiles = [&](auto dt) {
        constexpr int DT = decltype(dt)::value;   // compile-time dimension count (0: se.D at run time, <= GPMI_MAXD)
        const int Dn = DT ? DT : se.D;
        constexpr int DH = DT ? DT : 1;        // coordinates kept in registers per row / column (run-time D: re-read from LDS)
        double xm[4][DH], am[4], xn[DH], an = 0.0;
        int mm[4], ncur = 0;
        bool okn = false;
        auto contract = make_epi3(
            [&](int tm, int m, bool ok) {
                mm[tm] = ok ? m : -1;           // a row outside the matrix lies above every column: weight 0
                const int mc = ok ? m : 0;
                am[tm] = s_av[mc];
                if constexpr (DT != 0) {
#pragma unroll
                    for (int d = 0; d < DT; ++d) xm[tm][d] = xg[mc + d * n];
                }
            },
            [&](int nn, bool ok) {
                ncur = ok ? nn : 0;
                okn = ok;
                an = s_av[ncur];
                if constexpr (DT != 0) {
#pragma unroll
                    for (int d = 0; d < DT; ++d) xn[d] = xg[ncur + d * n];
                }
            },
            [&](double kinv, int tm) {
                double e = 0.0, r2[GPMI_MAXD];
                const int mc = mm[tm] < 0 ? 0 : mm[tm];
#pragma unroll
                for (int d = 0; d < (DT ? DT : GPMI_MAXD); ++d) {
                    double r;
                    if constexpr (DT != 0) r = xm[tm][d] - xn[d];
                    else r = d < Dn ? xg[mc + d * n] - xg[ncur + d * n] : 0.0;
                    r2[d] = r * r;
                    e += r2[d];
                }
                const double kse = a2 * exp_nonpos(-0.5 * e, ec);
                const double g = 0.5 * (am[tm] * an - kinv);
                const bool lower = okn && ncur <= mm[tm];
                const double c = lower ? ((ncur == mm[tm]) ? 1.0 : 2.0) * g * kse : 0.0;
                acc[0] += c;
#pragma unroll
                for (int d = 0; d < (DT ? DT : GPMI_MAXD); ++d) acc[1 + d] += c * r2[d];
                acc[1 + GPMI_MAXD] += (lower && ncur == mm[tm]) ? g : 0.0;
            });
        for (int ti = 0; ti * GT < n; ++ti)
            for (int tj = 0; tj <= ti; ++tj) {
                const int k0 = ti * GT;
                gemm_tile<3>(smem, U + (size_t)k0 * ld, ld, U + (size_t)k0 * ld, ld, W, ld, n, n, n - k0, ti, tj, 0, (int)threadIdx.x,
                             contract);
                __syncthreads();
            }
    };
    switch (se.D) {
    case 1: kinv_tiles(ic<1>{}); break;
    case 2: kinv_tiles(ic<2>{}); break;
    case 3: kinv_tiles(ic<3>{}); break;
    default: kinv_tiles(ic<0>{}); break;
    }
    GPMI_STAMP(tg4)
    GPMI_STAMP_ADD(6, tg4 - tg3)
#pragma unroll
    for (int d = 0; d < GPMI_MAXD; ++d)   // the sums are over UNSCALED squared differences (layout of k_grad_partial)
        acc[1 + d] = (d < se.D) ? acc[1 + d] / (se.inv_ell[d] * se.inv_ell[d]) : 0.0;
    __syncthreads();
    const int info = __hip_atomic_load(info_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // fixed-shape reduction (deterministic): butterfly inside every wave, the four wave sums added in wave order
#pragma unroll
    for (int q = 0; q < SMALL_GRAD_NS; ++q) {
        double v = acc[q];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if ((tid & 63) == 0) s_a[(tid >> 6) * SMALL_GRAD_NS + q] = v;
    }
    __syncthreads();
    if (tid < SMALL_GRAD_NS) {
        const double v = ((s_a[tid] + s_a[SMALL_GRAD_NS + tid]) + s_a[2 * SMALL_GRAD_NS + tid]) + s_a[3 * SMALL_GRAD_NS + tid];
        res[3 + tid] = info ? __builtin_nan("") : v;
    }
    GPMI_STAMP(tg5)
    GPMI_STAMP_ADD(7, tg5 - tg4)
    if (tid == 0) {
        if (info_out) *info_out = info;
        if (info) {
            res[0] = res[1] = res[2] = __builtin_nan("");
        } else {
            res[1] = sum_log;
            res[2] = zz;
            res[0] = -0.5 * zz - sum_log - 0.5 * (double)n * 1.8378770664093454835606594728112;  // log(2 pi)
        }
    }
}

// Completion flag of the one-launch host-buffer calls: the results lie in pinned, device-mapped host memory; every thread
// makes its stores visible system-wide, the workgroup meets, and thread 0 publishes `seq` -- the host polls the flag instead of
// paying a stream synchronisation (~8 us of a 30 us call).  done == nullptr: no flag.
__device__ __forceinline__ void small_signal_done(int *done, int seq)
{
    if (!done) return;   // kernel argument: workgroup-uniform
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

constexpr int SMALL_PTS = 128;  // grid points per launch: their hyper-parameters travel as kernel arguments
struct SmallBatch {
    double a2[SMALL_PTS], inv_rho[SMALL_PTS], diag[SMALL_PTS];
};

// Workgroup memory of the small kernels: staging buffer of gemm_tile / workspace of the diagonal-block body, the
// packed factors of the current panel, reduction / substitution scratch.  DYNAMIC: with 148 KB of static LDS the
// compiler knows that one workgroup fits per CU, hands the kernel all 512 registers and moves values through AGPRs
// around factor16's hand-scheduled DPP chain -- whose hazard spacing (the recogniser cannot see into asm statements)
// it thereby breaks (v_accvgpr_read directly in front of a DPP read of the same register: wrong numbers).  With
// the size unknown at compile time __launch_bounds__(256, 2) holds and the kernel is allocated like k_gemm_nt<0>:
// <= 256 registers, no AGPR traffic.
constexpr int SMALL_LDS_DOUBLES = 2 * 2 * GK * GP + GPMI_FPACK + 512;
extern __shared__ __attribute__((aligned(16))) double small_lds[];
#define GPMI_SMALL_LDS                                                                                   \
    double (&smem)[2][2][GK][GP] = *reinterpret_cast<double (*)[2][2][GK][GP]>(small_lds);               \
    double *s_F = small_lds + 2 * 2 * GK * GP;                                                           \
    double *s_aux = s_F + GPMI_FPACK;

// stage (nullable): X and y are host-mapped memory (the host-buffer entry point): they are first copied, one
// coalesced pass with every load in flight (one PCIe round trip), to `stage` in device memory; out3 / info_out
// may likewise be host-mapped -- nothing is copied around the launch
__global__ __launch_bounds__(256, 2) void k_logml_small(const double *__restrict__ X, int n, int ldx, const double *__restrict__ y,
                                                     SeParams p, double diag_add, double *__restrict__ W, size_t ld,
                                                     double *__restrict__ out3, int *info_out, int *info_w, ExpC ec,
                                                     double *__restrict__ stage, int *done, int seq)
{
    GPMI_SMALL_LDS
    if (stage) {
        const int nx = n * p.D;
        for (int e = threadIdx.x; e < nx + n; e += 256) {
            const int d = e / n, i = e - d * n;
            stage[e] = (e < nx) ? X[(size_t)i + (size_t)d * ldx] : y[e - nx];
        }
        __syncthreads();
        X = stage;
        y = stage + nx;
        ldx = n;
    }
    SmallSe se;
    se.a2 = p.a2;
    se.D = p.D;
#pragma unroll
    for (int d = 0; d < GPMI_MAXD; ++d) se.inv_ell[d] = p.inv_ell[d];
    logml_small_body(smem, s_F, s_aux, X, n, ldx, y, se, diag_add, W, ld, out3, info_out, info_w, ec);
    small_signal_done(done, seq);
}

// workgroup g = point g of the batch: isotropic (alpha, rho, sigma) as in gpmi_logml_grid; workspace slice g
__global__ __launch_bounds__(256, 2) void k_logml_small_batch(const double *__restrict__ X, int n, int ldx, int D,
                                                           const double *__restrict__ y, SmallBatch b, double *__restrict__ Wall,
                                                           size_t wstride, size_t ld, double *__restrict__ out3, int *info_out,
                                                           int *info_w, ExpC ec)
{
    GPMI_SMALL_LDS
    const int g = blockIdx.x;
    SmallSe se;
    se.a2 = b.a2[g];
    se.D = D;
#pragma unroll
    for (int d = 0; d < GPMI_MAXD; ++d) se.inv_ell[d] = b.inv_rho[g];
    logml_small_body(smem, s_F, s_aux, X, n, ldx, y, se, b.diag[g], Wall + (size_t)g * wstride, ld, out3 + 3 * (size_t)g,
                     info_out + g, info_w + g, ec);
}

// the same with one length-scale PER DIMENSION and point (ARD grids: QQard takes a vector phi[[2]], R/kernels.R:11-19);
// 32 points per launch (their D <= 8 inverse length-scales travel as kernel arguments too)
constexpr int SMALL_PTS_ARD = 32;
struct SmallBatchArd {
    double a2[SMALL_PTS_ARD], diag[SMALL_PTS_ARD], inv_ell[SMALL_PTS_ARD][GPMI_MAXD];
};
__global__ __launch_bounds__(256, 2) void k_logml_small_batch_ard(const double *__restrict__ X, int n, int ldx, int D,
                                                               const double *__restrict__ y, SmallBatchArd b,
                                                               double *__restrict__ Wall, size_t wstride, size_t ld,
                                                               double *__restrict__ out3, int *info_out, int *info_w, ExpC ec)
{
    GPMI_SMALL_LDS
    const int g = blockIdx.x;
    SmallSe se;
    se.a2 = b.a2[g];
    se.D = D;
#pragma unroll
    for (int d = 0; d < GPMI_MAXD; ++d) se.inv_ell[d] = b.inv_ell[g][d];
    logml_small_body(smem, s_F, s_aux, X, n, ldx, y, se, b.diag[g], Wall + (size_t)g * wstride, ld, out3 + 3 * (size_t)g,
                     info_out + g, info_w + g, ec);
}

// Points whose hyper-parameters lie in DEVICE memory -- any number per launch, isotropic or ARD: par[g] = {alpha^2,
// sigma^2 + jitter, 1 / ell_0 .. 1 / ell_7}.  Used for grids of more than GPMI_SMALL_PTS points and for the mid sizes
// (n <= 1024) at which a grid large enough to give every CU a problem of its own beats the four lanes of the blocked path.
constexpr int SMALL_PAR = 2 + GPMI_MAXD;
__global__ __launch_bounds__(256, 2) void k_logml_small_batch_dev(const double *__restrict__ X, int n, int ldx, int D,
                                                               const double *__restrict__ y, const double *__restrict__ par,
                                                               double *__restrict__ Wall, size_t wstride, size_t ld,
                                                               double *__restrict__ out3, int *info_out, int *info_w, ExpC ec)
{
    GPMI_SMALL_LDS
    const int g = blockIdx.x;
    const double *pg = par + (size_t)g * SMALL_PAR;
    SmallSe se;
    se.a2 = pg[0];
    se.D = D;
#pragma unroll
    for (int d = 0; d < GPMI_MAXD; ++d) se.inv_ell[d] = pg[2 + d];
    logml_small_body(smem, s_F, s_aux, X, n, ldx, y, se, pg[1], Wall + (size_t)g * wstride, ld, out3 + 3 * (size_t)g,
                     info_out + g, info_w + g, ec);
}

// value + gradient sums: one evaluation (host-mapped X, y staged as in k_logml_small) ...
constexpr int SMALL_GRAD_LDS_DOUBLES = SMALL_LDS_DOUBLES + 512;   // s_aux: two reduction arrays + z + a
__global__ __launch_bounds__(256, 2) void k_logml_grad_small(const double *__restrict__ X, int n, int ldx, const double *__restrict__ y,
                                                          SeParams p, double diag_add, double *__restrict__ W, size_t ld,
                                                          double *__restrict__ U, double *__restrict__ res, int *info_out, int *info_w,
                                                          ExpC ec, double *__restrict__ stage, int *done, int seq)
{
    GPMI_SMALL_LDS
    if (stage) {
        const int nx = n * p.D;
        for (int e = threadIdx.x; e < nx + n; e += 256) {
            const int d = e / n, i = e - d * n;
            stage[e] = (e < nx) ? X[(size_t)i + (size_t)d * ldx] : y[e - nx];
        }
        __syncthreads();
        X = stage;
        y = stage + nx;
        ldx = n;
    }
    SmallSe se;
    se.a2 = p.a2;
    se.D = p.D;
#pragma unroll
    for (int d = 0; d < GPMI_MAXD; ++d) se.inv_ell[d] = p.inv_ell[d];
    logml_grad_small_body(smem, s_F, s_aux, X, n, ldx, y, se, diag_add, W, ld, U, res, info_out, info_w, ec);
    small_signal_done(done, seq);
}

// ... and G isotropic points (the chains of a sampler: rstan's default is four), one workgroup each; slice g of Wall holds
// W and, ustride doubles behind it, U
__global__ __launch_bounds__(256, 2) void k_logml_grad_small_batch(const double *__restrict__ X, int n, int ldx, int D,
                                                                const double *__restrict__ y, SmallBatch b,
                                                                double *__restrict__ Wall, size_t wstride, size_t ustride, size_t ld,
                                                                double *__restrict__ res, int *info_out, int *info_w, ExpC ec,
                                                                double *__restrict__ stage, int *done, int seq, int *arrive)
{
    GPMI_SMALL_LDS
    const int g = blockIdx.x;
    if (stage) {   // X, y host-mapped (few chains: every workgroup stages its own copy, one PCIe round trip, side by side)
        double *st = stage + (size_t)g * n * (D + 1);
        const int nx = n * D;
        for (int e = threadIdx.x; e < nx + n; e += 256) {
            const int d = e / n, i = e - d * n;
            st[e] = (e < nx) ? X[(size_t)i + (size_t)d * ldx] : y[e - nx];
        }
        __syncthreads();
        X = st;
        y = st + nx;
        ldx = n;
    }
    SmallSe se;
    se.a2 = b.a2[g];
    se.D = D;
#pragma unroll
    for (int d = 0; d < GPMI_MAXD; ++d) se.inv_ell[d] = b.inv_rho[g];
    double *W = Wall + (size_t)g * wstride;
    logml_grad_small_body(smem, s_F, s_aux, X, n, ldx, y, se, b.diag[g], W, ld, W + ustride, res + (size_t)g * SMALL_GRAD_RES,
                          info_out + g, info_w + g, ec);
    if (done) {   // the LAST workgroup to finish publishes the completion flag (device counter `arrive`, re-armed by it)
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            const int k = __hip_atomic_fetch_add(arrive, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (k == (int)gridDim.x - 1) {
                __hip_atomic_store(arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(done, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
}

// One posterior draw of the derivative process (sample_derivs, pendulum_fit.R:227-255) per workgroup: the loop
// mclapply(s_list[1:100], sample_derivs_both_states, mc.cores = 2) (:261-268) is B independent draws, each with its own
// (l, a, sy) and noisy series, at n = m = 199 -- a chain of ~25 latency-bound launches per draw on the blocked path.  Here
// draw b builds [[a^2 QQ + sy^2 I, .], [a^2 RQ, a^2 RR]] with the row [y^T, 0] (the arithmetic of k_deriv_cov: deriv_val),
// factors the first n columns (Schur complement = cov - jitter I in the trailing block, -mu^T in the last row), adds the
// jitter, factors the m x m block in place and forms mu + chol(cov) z -- the composition of sample_derivs_core.
// par[3 g ..] = (l, a, sy) of draw g, in device memory (any number of draws per launch).
// status: 0, k (K + sy^2 I not PD at order k), n + k (cov).
__global__ __launch_bounds__(256, 2) void k_sample_derivs_small_batch(const double *__restrict__ t, int n, const double *__restrict__ ts,
                                                                   int m, const double *__restrict__ Y, const double *__restrict__ par,
                                                                   double jitter, const double *__restrict__ Z, double *__restrict__ Wall,
                                                                   size_t wstride, size_t ld, double *__restrict__ draws,
                                                                   double *__restrict__ mus, int *__restrict__ status,
                                                                   int *__restrict__ info_w)
{
    GPMI_SMALL_LDS
    const int g = blockIdx.x, tid = threadIdx.x;
    const int nt = n + m;
    double *W = Wall + (size_t)g * wstride;
    const double a2 = par[3 * g + 1] * par[3 * g + 1], l2 = par[3 * g] * par[3 * g], s2 = par[3 * g + 2] * par[3 * g + 2];
    const double *y = Y + (size_t)g * n, *z = Z + (size_t)g * m;
    int *iw = info_w + 2 * g;
    if (tid < 2) iw[tid] = 0;
    // lower triangle of the joint matrix, thread = row, and the augmented row
    for (int i = tid; i < nt; i += 256) {
        const bool star = i >= n;
        const double xi = star ? ts[i - n] : t[i];
        const int jn = i < n ? i + 1 : n;
        for (int j = 0; j < jn; ++j) {
            double v = a2 * deriv_val(star ? GPMI_RQ : GPMI_QQ, xi, t[j], l2);
            if (i == j) v += s2;
            W[(size_t)i + (size_t)j * ld] = v;
        }
        for (int j = n; j <= i; ++j) W[(size_t)i + (size_t)j * ld] = a2 * deriv_val(GPMI_RR, xi, ts[j - n], l2);
    }
    for (int j = tid; j < nt; j += 256) W[(size_t)nt + (size_t)j * ld] = j < n ? y[j] : 0.0;
    __syncthreads();
    small_potrf_partial(smem, s_F, s_aux, W, ld, nt + 1, nt, n, iw, false);
    __syncthreads();
    double *S = W + (size_t)n + (size_t)n * ld;
    for (int j = tid; j < m; j += 256) {
        S[(size_t)j * (ld + 1)] += jitter;
        const double mu = -W[(size_t)nt + (size_t)(n + j) * ld];
        mus[(size_t)g * m + j] = mu;
    }
    __syncthreads();
    small_potrf_partial(smem, s_F, s_aux, S, ld, m, m, m, iw + 1, false);
    __syncthreads();
    // draw = mu + L z: row i, columns 0 .. i in order (the order of k_trmv_lower_part within a chunk), sixteen loads in flight
    for (int i0 = 0; i0 < m; i0 += 256) {
        const int i = i0 + tid, ic = i < m ? i : m - 1;
        double acc = 0.0;
        const int jend = (i0 + 255 < m ? i0 + 255 : m - 1);   // workgroup-uniform bound; columns > i contribute exact zeros
        for (int j0 = 0; j0 <= jend; j0 += 16) {
            double u[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int j = j0 + q <= ic ? j0 + q : ic;
                u[q] = S[(size_t)ic + (size_t)j * ld];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (j0 + q <= ic) acc = fma(u[q], z[j0 + q], acc);
        }
        if (i < m) draws[(size_t)g * m + i] = acc + mus[(size_t)g * m + i];
    }
    __syncthreads();
    if (tid == 0) {
        const int i1 = __hip_atomic_load(iw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int i2 = __hip_atomic_load(iw + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        status[g] = i1 ? i1 : (i2 ? n + i2 : 0);
    }
}

// gpmi_gp_condition (p_Xn / p_dotXn, R/ode_gp.R:1-32; the moments of sample_derivs) at the sizes R/tests.R runs it by ONE
// workgroup: joint matrix [[K + s2 I, .], [Ks, Kss]] with the row [y^T, 0] (deriv_cov_val: the arithmetic of k_deriv_cov),
// partial factorisation of the first n columns, then Kn = Schur complement mirrored + jitter I and mn = minus the last row --
// the launch chain's nine kernels and six copies in one launch.  stage (nullable): t, ts, y are host-mapped and are copied
// to device memory first; Kn / mn / info_out may be host-mapped as well.
struct CondArgs {
    int kindK, kindS, kindSS, compat;
    double a2, l2, s2, jitter;
};
__global__ __launch_bounds__(256, 2) void k_gp_condition_small(const double *__restrict__ t, int n, const double *__restrict__ ts, int m,
                                                            const double *__restrict__ y, CondArgs q, double *__restrict__ W, size_t ld,
                                                            double *__restrict__ Kn, size_t ldo, double *__restrict__ mn, int *info_out,
                                                            int *info_w, double *__restrict__ stage, int *done, int seq)
{
    GPMI_SMALL_LDS
    const int tid = threadIdx.x, nt = n + m;
    if (stage) {
        for (int e = tid; e < 2 * n + m; e += 256) stage[e] = e < n ? t[e] : (e < nt ? ts[e - n] : y[e - nt]);
        __syncthreads();
        t = stage;
        ts = stage + n;
        y = stage + nt;
    }
    if (tid == 0) *info_w = 0;
    for (int i = tid; i < nt; i += 256) {
        const bool star = i >= n;
        const double xi = star ? ts[i - n] : t[i];
        const int jn = i < n ? i + 1 : n;
        for (int j = 0; j < jn; ++j) {
            double v = deriv_cov_val(star ? q.kindS : q.kindK, q.compat, q.a2, xi, t[j], q.l2);
            if (i == j) v += q.s2;
            W[(size_t)i + (size_t)j * ld] = v;
        }
        for (int j = n; j <= i; ++j) W[(size_t)i + (size_t)j * ld] = deriv_cov_val(q.kindSS, q.compat, q.a2, xi, ts[j - n], q.l2);
    }
    for (int j = tid; j < nt; j += 256) W[(size_t)nt + (size_t)j * ld] = j < n ? y[j] : 0.0;
    __syncthreads();
    small_potrf_partial(smem, s_F, s_aux, W, ld, nt + 1, nt, n, info_w, false);
    __syncthreads();
    const double *S = W + (size_t)n + (size_t)n * ld;
    for (int r = tid; r < m; r += 256) {
        for (int c = 0; c < m; ++c) {
            double v = (r >= c) ? S[(size_t)r + (size_t)c * ld] : S[(size_t)c + (size_t)r * ld];
            if (r == c) v += q.jitter;
            Kn[(size_t)r + (size_t)c * ldo] = v;
        }
        mn[r] = -W[(size_t)nt + (size_t)(n + r) * ld];
    }
    if (tid == 0) *info_out = __hip_atomic_load(info_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    small_signal_done(done, seq);
}

// f = chol(cov_exp_quad(X, alpha, ell) + diag_add I) z (models/exact_gp.stan:17-25: the latent exact GP's transform, once per
// leapfrog step with a new length-scale) by ONE workgroup for n <= 256: build (se_cov_tile), factorisation, and the row sums
// f_i = sum_{j <= i} L_ij z_j in column order (the order of k_trmv_lower_part inside its first chunk).  stage (nullable): X, z
// host-mapped -> copied to device memory first; f / info_out may be host-mapped.
__global__ __launch_bounds__(256, 2) void k_exact_gp_small(const double *__restrict__ X, int n, int ldx, const double *__restrict__ z,
                                                        SeParams p, double diag_add, double *__restrict__ W, size_t ld,
                                                        double *__restrict__ f, int *info_out, int *info_w, ExpC ec,
                                                        double *__restrict__ stage, int *done, int seq)
{
    GPMI_SMALL_LDS
    const int tid = threadIdx.x;
    if (stage) {
        const int nx = n * p.D;
        for (int e = tid; e < nx + n; e += 256) {
            const int d = e / n, i = e - d * n;
            stage[e] = (e < nx) ? X[(size_t)i + (size_t)d * ldx] : z[e - nx];
        }
        __syncthreads();
        X = stage;
        z = stage + nx;
        ldx = n;
    }
    if (tid == 0) *info_w = 0;
    SmallSe se;
    se.a2 = p.a2;
    se.D = p.D;
#pragma unroll
    for (int d = 0; d < GPMI_MAXD; ++d) se.inv_ell[d] = p.inv_ell[d];
    {
        double *xs = &smem[0][0][0][0];
#pragma unroll
        for (int d = 0; d < GPMI_MAXD; ++d)
            if (d < se.D)
                for (int i = tid; i < n; i += 256) xs[i + d * n] = __dmul_rn(X[(size_t)i + (size_t)d * ldx], se.inv_ell[d]);
        __syncthreads();
        for (int row0 = 0; row0 < n; row0 += SE_TR)
            for (int col0 = 0; col0 < row0 + SE_TR && col0 < n; col0 += SE_TC) {
                switch (se.D) {
                case 1: se_cov_tile<1, true>(xs, n, n, xs, n, n, se, diag_add, 1, 1, W, ld, 1, ec, row0, col0); break;
                case 2: se_cov_tile<2, true>(xs, n, n, xs, n, n, se, diag_add, 1, 1, W, ld, 1, ec, row0, col0); break;
                case 3: se_cov_tile<3, true>(xs, n, n, xs, n, n, se, diag_add, 1, 1, W, ld, 1, ec, row0, col0); break;
                default: se_cov_tile<0, true>(xs, n, n, xs, n, n, se, diag_add, 1, 1, W, ld, 1, ec, row0, col0); break;
                }
            }
    }
    __syncthreads();
    small_potrf_partial(smem, s_F, s_aux, W, ld, n, n, n, info_w, false);
    __syncthreads();
    if (tid < n) s_aux[tid] = z[tid];
    __syncthreads();
    if (tid < n) {
        const int i = tid;
        double acc = 0.0;
        for (int j0 = 0; j0 <= i; j0 += 16) {
            double u[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int j = j0 + q <= i ? j0 + q : i;
                u[q] = W[(size_t)i + (size_t)j * ld];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q)
                if (j0 + q <= i) acc = fma(u[q], s_aux[j0 + q], acc);
        }
        const int info = __hip_atomic_load(info_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        f[i] = info ? __builtin_nan("") : acc;
    }
    if (tid == 0) *info_out = __hip_atomic_load(info_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    small_signal_done(done, seq);
}

// rbf_cov_chol (covariance.cpp:9-47) by ONE workgroup for n <= 128 (test_interpolate.R:5 runs it at N = 100, P = 10 times):
// Sigma_ij = exp(-(x_i - x_j)^2 / (2 l^2)) + 1e-10 [i == j], L = chol(Sigma), and the forward-mode tangent
// dL/dl = L Phi(L^-1 Sdot L^-T), Sdot_ij = Sigma_ij (x_i - x_j)^2 / l^3, Phi = lower triangle with halved diagonal --
// the launch chain of rbf_cov_chol_core (build, factor, copy, tangent build, two panel solves, two transposes, mask, product:
// ~12 launches) with the same device functions back to back.  Workgroup g handles length-scale ls[g] and writes L (upper
// zeroed) and dL/dl (lower; its upper triangle exact zeros) to Lout + g ostride, dLout + g ostride (leading dimension ldo;
// device or host-mapped memory).  Workspace per workgroup: three slices of small_ws_layout(n) (Sigma / L, S, S2).
struct RbfBatch {
    double l[64];
};
__global__ __launch_bounds__(256, 2) void k_rbf_cov_chol_small(const double *__restrict__ x, int n, RbfBatch ls, double *__restrict__ Wall,
                                                            size_t wstride, size_t ld, double *__restrict__ Lout,
                                                            double *__restrict__ dLout, size_t ostride, size_t ldo, int *info_out,
                                                            int *info_w, ExpC ec, double *__restrict__ stage)
{
    GPMI_SMALL_LDS
    const int g = blockIdx.x, tid = threadIdx.x;
    const double l = ls.l[g];
    double *W = Wall + (size_t)g * 3 * wstride, *S = W + wstride, *S2 = S + wstride;
    double *Lo = Lout + (size_t)g * ostride, *dLo = dLout + (size_t)g * ostride;
    int *iw = info_w + g;
    if (stage) {   // x host-mapped: one copy per workgroup
        double *st = stage + (size_t)g * n;
        for (int i = tid; i < n; i += 256) st[i] = x[i];
        __syncthreads();
        x = st;
    }
    if (tid == 0) *iw = 0;
    SmallSe se;
    se.a2 = 1.0;
    se.D = 1;
#pragma unroll
    for (int d = 0; d < GPMI_MAXD; ++d) se.inv_ell[d] = 1.0 / l;
    double *xs = &smem[0][0][0][0];
    for (int i = tid; i < n; i += 256) xs[i] = __dmul_rn(x[i], se.inv_ell[0]);
    __syncthreads();
    for (int row0 = 0; row0 < n; row0 += SE_TR)
        for (int col0 = 0; col0 < row0 + SE_TR && col0 < n; col0 += SE_TC)
            se_cov_tile<1, true>(xs, n, n, xs, n, n, se, 1e-10, 1, 1, W, ld, 1, ec, row0, col0);
    // Sdot, full (the arithmetic of k_rbf_dsigma), thread = row
    for (int i = tid; i < n; i += 256) {
        const double xi = x[i];
        for (int j = 0; j < n; ++j) {
            const double r = xi - x[j], r2 = r * r;
            S[(size_t)i + (size_t)j * ld] = exp(-r2 / (2 * l * l)) * r2 / (l * l * l);
        }
    }
    __syncthreads();
    const int nblk = (n + 15) >> 4;
    if (n == GPMI_NB) potrf_diag4_body<false, true>(&smem[0][0][0][0], W, ld, n, s_F, iw, 0, 8, tid);
    else potrf_diag4_body<false, false>(&smem[0][0][0][0], W, ld, n, s_F, iw, 0, nblk, tid);
    __syncthreads();
    // (the element-wise passes below keep eight loads in flight per round trip: a loop with one dependent load per iteration
    // is a chain of n memory latencies -- 100 us per pass at n = 100)
    // L out, and its upper triangle zeroed in place: W is the A operand of the last product
    for (int i = tid; i < n; i += 256)
        for (int j0 = 0; j0 < n; j0 += 8) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j = j0 + q < n ? j0 + q : n - 1;
                v[q] = W[(size_t)i + (size_t)j * ld];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j = j0 + q;
                if (j < n) {
                    if (j > i) W[(size_t)i + (size_t)j * ld] = 0.0;
                    Lo[(size_t)i + (size_t)j * ldo] = (j <= i) ? v[q] : 0.0;
                }
            }
        }
    auto solve_rows = [&](double *A) {   // A <- A L^-T, all n rows, 64 per round
        for (int rb = 0; rb < n; rb += 64) {
            const int r = rb + (tid >> 6) * 16 + (tid & 15);
            if (n == GPMI_NB && rb + 64 <= n) trsm_panel_body<true>(s_F, A, ld, r, true, n, tid);
            else trsm_panel_body<false>(s_F, A, ld, r, r < n, n, tid);
        }
        __syncthreads();
    };
    solve_rows(S);                                                  // S = Sdot L^-T
    for (int i = tid; i < n; i += 256)                              // S2 = S^T = L^-1 Sdot
        for (int j0 = 0; j0 < n; j0 += 8) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = S[(size_t)(j0 + q < n ? j0 + q : n - 1) + (size_t)i * ld];
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (j0 + q < n) S2[(size_t)i + (size_t)(j0 + q) * ld] = v[q];
        }
    __syncthreads();
    solve_rows(S2);                                                 // S2 = M = L^-1 Sdot L^-T
    // B operand of the product: row j, column k holds Phi(M)[k][j]  (k >= j; the diagonal halved)
    for (int j = tid; j < n; j += 256)
        for (int k0 = 0; k0 < n; k0 += 8) {
            double v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = S2[(size_t)(k0 + q < n ? k0 + q : n - 1) + (size_t)j * ld];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = k0 + q;
                if (k < n) S[(size_t)j + (size_t)k * ld] = (k > j) ? v[q] : ((k == j) ? 0.5 * v[q] : 0.0);
            }
        }
    __syncthreads();
    gemm_tile<2>(smem, W, ld, S, ld, dLo, ldo, n, n, n, 0, 0, 0, tid);   // dL = L Phi (n <= 128: one tile)
    __syncthreads();
    if (tid == 0) info_out[g] = __hip_atomic_load(iw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// stream-ordered upload of up to PUT_MAX doubles that travel as kernel arguments (no staging buffer whose reuse would
// have to be fenced against an earlier asynchronous call)
constexpr int PUT_MAX = 480;
struct PutArgs {
    double v[PUT_MAX];
};
__global__ __launch_bounds__(256) void k_put_doubles(PutArgs a, double *__restrict__ dst, int count)
{
    for (int i = threadIdx.x; i < count; i += 256) dst[i] = a.v[i];
}

// the factorisation alone (launch_potrf_partial at small sizes: posteriors, rbf_cov_chol, ...)
__global__ __launch_bounds__(256, 2) void k_potrf_small(double *__restrict__ W, size_t ld, int M, int ncol, int nfac, int *info)
{
    GPMI_SMALL_LDS
    small_potrf_partial(smem, s_F, s_aux, W, ld, M, ncol, nfac, info, false);
}
#undef GPMI_SMALL_LDS

// Packed factors (Fpack) of an ALREADY factored diagonal block: -L tiles in fragment
// order and the inverse of every 16x16 diagonal tile.  Used by solves against a given L.
__global__ __launch_bounds__(512) void k_pack_factors(const double *__restrict__ L11, size_t ldl,
                                                      int nb_act, double *__restrict__ Fpack)
{
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lq = lane >> 4;
    const int row = w * 16 + lr;
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
        if (kb < w) {
#pragma unroll
            for (int kg = 0; kg < 4; ++kg) {
                const int col = kb * 16 + lq + 4 * kg;
                const double v = (row < nb_act && col < nb_act) ? L11[(size_t)row + (size_t)col * ldl] : 0.0;
                Fpack[(size_t)(w * (w - 1) / 2 + kb) * 256 + kg * 64 + lane] = -v;
            }
        }
    }
    double rw[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int col = w * 16 + c;
        rw[c] = (c <= lr && row < nb_act) ? L11[(size_t)row + (size_t)col * ldl] : ((c == lr) ? 1.0 : 0.0);
    }
    double x[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        double s = (r == lr) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < r; ++k) s = fma(-readlane64(rw[k], r), x[k], s);
        x[r] = s / readlane64(rw[r], r);
    }
    if (lq == 0) {
#pragma unroll
        for (int j = 0; j < 16; ++j)
            Fpack[(size_t)fp_inv(w) * 256 + (lr >> 2) * 64 + j + 16 * (lr & 3)] = x[j];
    }
}

// ---------------------------------------------------------------------------
// One right-hand side against a given factor: t = L^-1 k (forward substitution), the whitened kernel row of
// the sequential sampler and mdivide_left_tri_low.  The factor is read ONCE (4 n^2 bytes: HBM-bound); what is
// sequential is the chain of the n / 128 diagonal blocks.  Done through the panel kernels a one-row solve is
// 2 n / 128 dependent launches (4.2 ms at n = 16384); here it is ONE launch:
//   workgroup w owns the 128 unknowns of block-row w.  For j = 0 .. w - 1 it takes the 128 x 128 block
//   L[w, j] -- its loads are issued BEFORE it waits for t_j -- multiplies it with t_j as soon as workgroup j
//   has published it, and after j = w - 1 applies the inverse of its diagonal block (k_diag_inv_t) to
//   k_w - sum_j L[w, j] t_j and publishes t_w.
// Publication needs no flag: the output vector is pre-filled with an all-ones bit pattern (a NaN no
// arithmetic produces: a computed NaN is stored as the canonical quiet NaN), every consumer thread polls ITS
// element with agent-scope loads until it is no longer the pattern, and 64-bit agent-scope stores are
// single-copy atomic.  Block-rows are handed out by a device ticket in START order, so a workgroup waits only for
// workgroups that are already running: the waits cannot deadlock, whatever the dispatch order or residency.
// Sums in a fixed order: deterministic.
// ---------------------------------------------------------------------------
constexpr int TSV = 128;
constexpr unsigned long long TSV_EMPTY = ~0ull;

// Dinv[w]: the inverse of the w-th 128 x 128 diagonal block of L, column-major with leading dimension 128
// (element (r, c) at r + 128 c; zero above the diagonal; identity rows / columns past the matrix order).
// From X = I solved against the block by the panel kernel (X L^-T = L^-T), transposed.
__global__ __launch_bounds__(256) void k_eye_blocks(double *__restrict__ X, int nblk)
{
    const int w = blockIdx.x;
    for (int e = threadIdx.x; e < TSV * TSV; e += 256) X[(size_t)w * TSV * TSV + e] = ((e & (TSV - 1)) == (e >> 7)) ? 1.0 : 0.0;
}
__global__ __launch_bounds__(256) void k_transpose_blocks(const double *__restrict__ X, double *__restrict__ D)
{
    __shared__ double t[16][17];
    const size_t base = (size_t)blockIdx.x * TSV * TSV;
    const int bi = (blockIdx.y & 7) * 16, bj = (blockIdx.y >> 3) * 16;
    const int a = threadIdx.x & 15, b = threadIdx.x >> 4;
    t[b][a] = X[base + (bi + a) + (size_t)(bj + b) * TSV];
    __syncthreads();
    D[base + (bj + a) + (size_t)(bi + b) * TSV] = t[a][b];
}

__global__ __launch_bounds__(256) void k_trsv_wave(const double *__restrict__ L, size_t ldl, int n,
                                                   const double *__restrict__ k, double *__restrict__ t,
                                                   const double *__restrict__ Dinv, int *__restrict__ ticket)
{
    __shared__ double ts[TSV];
    __shared__ double red[2][TSV];
    __shared__ int s_w;
    // Block-row by TICKET, not by blockIdx: HIP promises no dispatch order (each XCD deals its share of the grid
    // independently), so "lower blockIdx started earlier" is not a guarantee.  A workgroup that draws ticket w waits
    // only for tickets < w -- workgroups that have already STARTED, are resident and make progress -- whatever the
    // dispatch order and whatever else shares the chip.  The workgroup drawing the last ticket re-arms the counter
    // (the next launch is stream-ordered behind this one).
    if (threadIdx.x == 0) {
        s_w = atomicAdd(ticket, 1);
        if (s_w == (int)gridDim.x - 1) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const int w = s_w, tid = threadIdx.x, r = tid & (TSV - 1), h = tid >> 7;
    const int row = w * TSV + r;
    const bool rok = row < n;
    const size_t rowc = (size_t)(rok ? row : n - 1);
    // this thread's 64 entries of row r of the inverse diagonal block (columns 64 h ...)
    double Dr[64];
    {
        const double *d = Dinv + (size_t)w * TSV * TSV + r + (size_t)(64 * h) * TSV;
#pragma unroll
        for (int q = 0; q < 64; ++q) Dr[q] = d[(size_t)q * TSV];
    }
    const double kv = rok ? k[row] : 0.0;
    double acc = 0.0;
#pragma unroll 1
    for (int j = 0; j < w; ++j) {
        double Lr[64];
        const double *p = L + rowc + (size_t)(j * TSV + 64 * h) * ldl;
#pragma unroll
        for (int q = 0; q < 64; ++q) Lr[q] = p[(size_t)q * ldl];
        if (tid < TSV) {
            const unsigned long long *src = reinterpret_cast<const unsigned long long *>(t) + (size_t)j * TSV + tid;
            unsigned long long u;
            while ((u = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == TSV_EMPTY) __builtin_amdgcn_s_sleep(1);
            ts[tid] = __longlong_as_double((long long)u);
        }
        __syncthreads();
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int q = 0; q < 64; q += 2) {
            a0 = fma(Lr[q], ts[64 * h + q], a0);
            a1 = fma(Lr[q + 1], ts[64 * h + q + 1], a1);
        }
        acc += a0 + a1;
        __syncthreads();  // ts is rewritten in the next round
    }
    red[h][r] = acc;
    __syncthreads();
    if (h == 0) ts[r] = rok ? kv - (red[0][r] + red[1][r]) : 0.0;
    __syncthreads();
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int q = 0; q < 64; q += 2) {  // lower triangle only: a NaN further down the right-hand side stays there
        if (64 * h + q <= r) a0 = fma(Dr[q], ts[64 * h + q], a0);
        if (64 * h + q + 1 <= r) a1 = fma(Dr[q + 1], ts[64 * h + q + 1], a1);
    }
    __syncthreads();
    red[h][r] = a0 + a1;
    __syncthreads();
    if (h == 0 && rok) {
        double v = red[0][r] + red[1][r];
        if (v != v) v = __longlong_as_double(0x7ff8000000000000ll);  // never the "not yet" pattern
        __hip_atomic_store(t + row, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// out[j] = scale * W[row, col0 + j]
__global__ void k_get_row(const double *__restrict__ W, size_t ld, int row, int col0, int m, double scale,
                          double *__restrict__ out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j < m) out[j] = scale * W[(size_t)row + (size_t)(col0 + j) * ld];
}

// sum log L_ii, z'z (z = row zrow of the factor), logml.  Two launches with a fixed shape (the sums do
// not depend on timing): FIN_WG workgroups reduce 256-element slices of the diagonal -- every L_ii
// sits in a cache line of its own, so the loads want many waves in flight; one workgroup doing all of
// them took 60 us at N = 16384 -- and a last workgroup adds the slice sums in slice order.
__global__ __launch_bounds__(FIN_SLICE) void k_logml_partial(const double *__restrict__ W, size_t ld, int n, int zrow,
                                                             double *__restrict__ part)
{
    __shared__ double s_a[FIN_SLICE], s_b[FIN_SLICE];
    const int i = blockIdx.x * FIN_SLICE + threadIdx.x;
    double a = 0.0, b = 0.0;
    if (i < n) {
        a = log(W[(size_t)i * (ld + 1)]);
        const double z = W[(size_t)zrow + (size_t)i * ld];
        b = z * z;
    }
    s_a[threadIdx.x] = a;
    s_b[threadIdx.x] = b;
    __syncthreads();
    for (int s = FIN_SLICE / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            s_a[threadIdx.x] += s_a[threadIdx.x + s];
            s_b[threadIdx.x] += s_b[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = s_a[0];
        part[2 * blockIdx.x + 1] = s_b[0];
    }
}

__global__ __launch_bounds__(256) void k_logml_finalize(const double *__restrict__ part, int nslice, int n,
                                                        const int *d_info, double *__restrict__ out3, int *info_out)
{
    __shared__ double s_a[256], s_b[256];
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < nslice; i += 256) {
        a += part[2 * i];
        b += part[2 * i + 1];
    }
    s_a[threadIdx.x] = a;
    s_b[threadIdx.x] = b;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            s_a[threadIdx.x] += s_a[threadIdx.x + s];
            s_b[threadIdx.x] += s_b[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int info = *d_info;
        if (info_out) *info_out = info;
        if (info) {
            out3[0] = out3[1] = out3[2] = __builtin_nan("");
        } else {
            out3[1] = s_a[0];
            out3[2] = s_b[0];
            out3[0] = -0.5 * s_b[0] - s_a[0] - 0.5 * (double)n * 1.8378770664093454835606594728112;  // log(2 pi)
        }
    }
}

// f = L z, one thread per row (rows coalesced across lanes)
// 512-column chunks on a 2-D grid (a one-thread-per-row loop over all columns is a latency chain at
// large n), chunks wholly above the diagonal skipped, partial sums added in chunk order
constexpr int TMV_COLS = 512;
__global__ __launch_bounds__(256) void k_trmv_lower_part(const double *__restrict__ L, size_t ldl, int n,
                                                         const double *__restrict__ z, double *__restrict__ part)
{
    const int r0 = blockIdx.x * 256, i = r0 + threadIdx.x;
    const int c0 = blockIdx.y * TMV_COLS;
    if (i >= n) return;
    double s = 0.0;
    if (c0 <= r0 + 255) {
        const int c1 = (c0 + TMV_COLS - 1 < i) ? c0 + TMV_COLS - 1 : i;  // last column of this chunk for row i
        for (int k = c0; k <= c1; ++k) s = fma(L[(size_t)i + (size_t)k * ldl], z[k], s);
    }
    part[(size_t)blockIdx.y * n + i] = s;
}

__global__ __launch_bounds__(256) void k_trmv_lower_sum(const double *__restrict__ part, int n, int nchunk,
                                                        double *__restrict__ f)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int q = 0; q < nchunk; ++q) s += part[(size_t)q * n + i];
    f[i] = s;
}

#ifdef GPMI_PROBES
// D = A(16x4) * B(4x16) with the library's operand conventions (layout probe)
__global__ void k_probe_mfma(const double *A, const double *B, double *D)
{
    const int l = threadIdx.x;
    d4 acc = d4{0.0, 0.0, 0.0, 0.0};
    acc = mfma(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) D[((l >> 4) + 4 * i) * 16 + (l & 15)] = acc[i];
}

__global__ __launch_bounds__(256) void k_probe_peak(double *sink, int iters)
{
    const int l = threadIdx.x;
    d4 c0 = d4{0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    double a = 1.0 + 1e-9 * l, b = 1.0 - 1e-9 * l;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    // inline asm keeps the eight accumulators pinned in VGPRs (the builtin form makes hipcc
    // shuffle them through AGPRs every iteration, which is not what we want to time)
#define GPMI_MFMA_ASM(cc) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(cc) : "v"(a), "v"(b))
    for (int it = 0; it < iters; ++it) {
        GPMI_MFMA_ASM(c0); GPMI_MFMA_ASM(c1); GPMI_MFMA_ASM(c2); GPMI_MFMA_ASM(c3);
        GPMI_MFMA_ASM(c4); GPMI_MFMA_ASM(c5); GPMI_MFMA_ASM(c6); GPMI_MFMA_ASM(c7);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#undef GPMI_MFMA_ASM
    d4 s = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (s[0] + s[1] + s[2] + s[3] == 123.456) sink[0] = s[0];
    if (l == 0) {  // shader clock = d(s_memtime) / d(s_memrealtime) * 100 MHz
        sink[8 + 2 * blockIdx.x] = (double)(t1 - t0);
        sink[9 + 2 * blockIdx.x] = (double)(r1 - r0);
    }
}
#endif  // GPMI_PROBES

}  // namespace

// ---------------------------------------------------------------------------
// host-side drivers
// ---------------------------------------------------------------------------
// the small kernels use more dynamic workgroup memory than the default limit: raise it once per device
static void small_lds_attr()
{
    static bool done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || done[dev]) return;
    const int bytes = SMALL_LDS_DOUBLES * (int)sizeof(double);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_logml_small), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_logml_small_batch), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_logml_small_batch_ard), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_logml_small_batch_dev), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_sample_derivs_small_batch), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_gp_condition_small), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_exact_gp_small), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_rbf_cov_chol_small), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    const int gbytes = SMALL_GRAD_LDS_DOUBLES * (int)sizeof(double);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_logml_grad_small), hipFuncAttributeMaxDynamicSharedMemorySize, gbytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_logml_grad_small_batch), hipFuncAttributeMaxDynamicSharedMemorySize, gbytes);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_potrf_small), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done[dev] = true;
}

void gpmi_tuning_defaults(gpmi_tuning *t)
{
    t->syrk_order = 2;
    t->stagger = (2 << 16) | 4;
    t->fuse_diag = 15;
    t->diag_waves = 4;
    t->nb_adapt = 1;
    t->nb_thr[0] = 8192;
    t->nb_thr[1] = 4608;
    t->nb_thr[2] = 3584;
    t->ksplit = 1;
    t->ksplit_max = 100;
    t->block_recursive = 1;
    t->se_nt = 1;
    t->gemm_variant = 3;
    t->rect_auto = 0;
    t->small_n = 256;
    t->small_n1 = 128;
    t->small_m = 160;
    t->small_ng1 = 128;   // tools/grad_small_bench.py: one workgroup 57 / 82 / 109 / 207 / 303 / 337 us at n = 21 / 64 / 128 / 160 / 199 / 256,
                          // the launch chain 151 / 162 / 180 / ~225 / 274 / 272; four chains at once 93 .. 363 us against 377 .. 525
    t->small_ng = 256;
    t->grad_aug_n = 3072;
    t->grad_aug_ng = 2304;
    t->small_gc = 180;    // gpmi_gp_condition by one workgroup up to n + m + 1 rows (tools/cond_bench.py)
    t->small_sd = 640;
    t->small_sdb = 5;     // tools/sample_derivs_bench.py: one workgroup 0.21 / 0.56 / 0.73 ms at n = m = 79 / 199 / 256, the lanes 95 / 136 / 129 us per draw
    t->small_n2 = 1024;
    t->small_g2 = 40;
}

void launch_gemm_nt(const gpmi_ctx *c, hipStream_t s, const double *A, size_t lda, const double *B, size_t ldb,
                    double *C, size_t ldc, int M, int N, int K, int accumulate_minus)
{
    if (M <= 0 || N <= 0 || K <= 0) return;
    dim3 grid((M + GT - 1) / GT, (N + GT - 1) / GT);
#ifdef GPMI_PROBES
    // Launches with fewer tiles than workgroup slots are latency-bound (one tile per CU, every
    // k-step exposes the DMA latency): the 3-buffer-ring kernel (DMA two steps ahead, C tile
    // prefetched) has the shorter per-tile time there; the 2-workgroup-per-CU kernel wins once
    // several rounds of tiles keep each CU's pair of workgroups busy.
    const bool few_tiles = c->tune.rect_auto && (int)(grid.x * grid.y) <= 384;
    if (c->tune.gemm_variant == 2 || (c->tune.gemm_variant == 3 && few_tiles)) {
        if (accumulate_minus)
            hipLaunchKernelGGL(k_gemm9<0>, grid, 512, 0, s, A, lda, B, ldb, C, ldc, M, N, K, 0);
        else
            hipLaunchKernelGGL(k_gemm9<2>, grid, 512, 0, s, A, lda, B, ldb, C, ldc, M, N, K, 0);
        return;
    }
    if (c->tune.gemm_variant == 1) {
        if (accumulate_minus)
            hipLaunchKernelGGL(k_gemm8<0>, grid, 512, 0, s, A, lda, B, ldb, C, ldc, M, N, K, 0);
        else
            hipLaunchKernelGGL(k_gemm8<2>, grid, 512, 0, s, A, lda, B, ldb, C, ldc, M, N, K, 0);
        return;
    }
#else
    (void)c;
#endif
    if (accumulate_minus)
        hipLaunchKernelGGL(k_gemm_nt<0>, grid, 256, 0, s, A, lda, B, ldb, C, ldc, M, N, K, 0, 0, FuseDiag{}, KSplit{});
    else
        hipLaunchKernelGGL(k_gemm_nt<2>, grid, 256, 0, s, A, lda, B, ldb, C, ldc, M, N, K, 0, 0, FuseDiag{}, KSplit{});
}

// C (n x n, lower tiles only; the caller zeroes the rest) = A B^T, A lower triangular, B the transpose of a lower
// triangular matrix: n^3 / 3 flops instead of 2 n^3
void launch_gemm_tri_lower(hipStream_t s, const double *A, size_t lda, const double *B, size_t ldb, double *C, size_t ldc, int n)
{
    if (n <= 0) return;
    const int T = (n + GT - 1) / GT;
    hipLaunchKernelGGL(k_gemm_nt<2>, dim3(syrk_grid(T, T, 0)), 256, 0, s, A, lda, B, ldb, C, ldc, n, n, n, 0x400, 0, FuseDiag{},
                       KSplit{});
}

// the next panel's diagonal block is factored inside the update that completes it -- bit 0: in-block
// GEMMs, bit 1: the trailing SYRK (multi-round launches only).  Measured with 4 grid lanes (N = 16384 /
// 8192, ms per evaluation): off 25.0 / 4.12, GEMMs only 25.25 / 4.12, SYRK only 24.95 / 4.07, both
// 24.55 / 4.02 -- with both, no one-workgroup kernel is left that has to find a free CU next to the
// other lanes' updates.  One evaluation at a time: neutral (the block's tile sits on the critical
// path either way).
// (tune.fuse_diag, default 15; bit 2: sub-tiled diagonal tile in the fused in-block GEMMs; bit 3: the same in
// single-round trailing SYRK launches)

// C -= A B^T with the default kernel and the diagonal block at C's origin factored by the
// workgroup of tile (0, 0).  false: this configuration cannot fuse (caller launches the
// diagonal kernel itself and uses launch_gemm_nt).
static bool launch_gemm_nt_fused(const gpmi_ctx *c, hipStream_t s, const double *A, size_t lda, const double *B, size_t ldb, double *C,
                                 size_t ldc, int M, int N, int K, const FuseDiag &fd)
{
    if (!(c->tune.fuse_diag & 1) || (c->tune.gemm_variant != 3 && c->tune.gemm_variant != 0) || M <= 0 || N <= 0 || K <= 0) return false;
    dim3 grid((M + GT - 1) / GT, (N + GT - 1) / GT);
    if ((c->tune.fuse_diag & 4) && fd.ctr && M >= GT && N >= GT && K % 64 == 0) {  // tile (0, 0) interior: sub-tiled
        hipLaunchKernelGGL(k_gemm_nt<0>, dim3(grid.x * grid.y + SUBWG - 1), 256, 0, s, A, lda, B, ldb, C, ldc, M, N, K, 0x200, 0, fd, KSplit{});
        return true;
    }
    hipLaunchKernelGGL(k_gemm_nt<0>, grid, 256, 0, s, A, lda, B, ldb, C, ldc, M, N, K, 0, 0, fd, KSplit{});
    return true;
}

// tune.diag_waves  4 (default): k_potrf_diag4 (3 tile waves + factor wave, two barriers per block column; fits beside a
//                  resident SYRK workgroup), 5: k_potrf_diag (4 tile waves + factor wave, three barriers)
// tune.nb_adapt    1 (default): the auto outer-block width is re-chosen per block from the order of the matrix still to
//                  update (tune.nb_thr: >= 8192 -> 1024 columns, >= 4608 -> 512, >= 3584 -> 256, below -> 128): the last
//                  blocks of a large matrix are a small matrix.  Pays since single-round trailing updates carry the next
//                  diagonal block (fuse_diag bit 3): N = 8192 5.63 -> 5.35 ms one at a time, 3.54 -> 3.38 on lanes;
//                  N = 4096 1.54 -> 1.47; N = 16384 28.0 -> 27.4 / 23.3 -> 23.1 (thresholds swept in steps of 1024)
// tune.ksplit      quadrant split of the tail-round tiles of a SYRK launch (see KSplit) ...
// tune.ksplit_max  ... when at most this many tiles are left for the last round.  Whole tiles of a round this thin run
// alone on their CUs (~115 us at K = 1024 instead of ~250 us shared); four quadrant workgroups take
// ~75 us as long as they, too, have CUs of their own (4 R <= ~400).  Measured per launch (K = 1024):
// R = 16..92: -0.04 .. -0.065 ms; R = 136..340: +0.03 .. +0.06 ms.

// returns true when fd was given and the launch factors the diagonal block at C's origin
static bool launch_syrk_lower(const gpmi_ctx *c, hipStream_t s, const double *P, size_t ldp, double *C, size_t ldc, int M,
                              int N, int K, int *ctr, int ncu, int tri = 0, const FuseDiag *fd = nullptr)
{
    if (tri && M > 0 && N > 0 && K > 0) {  // upper-triangular panel: plain kernel, row-major tiles, zero K-range skipped
        const int T = (M + GT - 1) / GT;
        hipLaunchKernelGGL(k_gemm_nt<1>, dim3(syrk_grid(T, T, 0)), 256, 0, s, P, ldp, P, ldp, C, ldc, M, N, K, 0x100, 0,
                           FuseDiag{}, KSplit{});
        return false;
    }
    if (M <= 0 || N <= 0 || K <= 0) return false;
    const int T = (M + GT - 1) / GT;
    // N < M: lower trapezoid (block column of a look-ahead update), row-major order only
    const int TNf = (N + GT - 1) / GT, TN = TNf < T ? TNf : T;
    const int slots = 2 * (ncu > 0 ? ncu : 256);
    // tile walk: 0 row-major; 2 XCD-partitioned bands (launches of more than one round of tiles: that is where the
    // operand re-reads that miss the XCD L2s come from); 1 (probe build) the padded 8 x 8 super-tile grid of round 1
    // Walk 2 is the default for an evaluation that has the chip to itself (N = 16384: L2 hit rate 53 -> 79 %, L2-miss
    // traffic 1.27 -> 0.65 GB per launch, 27.31 -> 27.07 ms); under the grid lanes, where four launches share every
    // XCD, its static per-XCD partition loses 1.6 % (23.09 -> 23.45 ms per point) and the row-major walk stays.
    int syrk_order = (c->tune.syrk_order == 2 && !c->lanes_active && syrk_grid(T, TN, 0) > slots) ? 2 : 0;
#ifdef GPMI_PROBES
    if (c->tune.syrk_order == 1 && TN == T) syrk_order = 1;
#endif
    const int ntiles = syrk_grid(T, TN, syrk_order == 1 ? 1 : 0);
    const int stg = ntiles >= 1024 ? c->tune.stagger : 0;  // only when every CU holds two workgroups for many rounds
#ifdef GPMI_PROBES
    if (c->tune.gemm_variant == 2) {
        hipLaunchKernelGGL(k_gemm9<1>, dim3(ntiles), 512, 0, s, P, ldp, P, ldp, C, ldc, M, N, K, syrk_order);
        return false;
    }
    if (c->tune.gemm_variant == 1) {
        hipLaunchKernelGGL(k_gemm8<1>, dim3(ntiles), 512, 0, s, P, ldp, P, ldp, C, ldc, M, N, K, syrk_order);
        return false;
    }
#else
    (void)ctr;
#endif
    // tile (0, 0) is block 0 only in the row-major order
    // only in launches of more than one round of tiles, where workgroup 0's extra 27 us do not
    // lengthen the kernel
    bool fuse = fd && fd->Fp && (c->tune.fuse_diag & 2) && syrk_order != 1 && ntiles > slots;  // tile (0, 0) is block 0 in walks 0 and 2
    // Single-round launches: the next diagonal block rides along as well, sub-tiled like in the in-block products
    // (fused_subtiles_and_diag) -- its 36 sub-tiles and the block's factorisation (~6 + 21 us) run beside the other
    // tiles instead of in a kernel of their own behind the launch (tune.fuse_diag bit 3).
    int ord = syrk_order;
    FuseDiag fdl = fuse ? *fd : FuseDiag{};
    if (!fuse && fd && fd->Fp && (c->tune.fuse_diag & 8) && syrk_order == 0 && M >= GT && N >= GT && K % 64 == 0 && c->d_ctr) {
        fuse = true;
        ord |= 0x200;
        fdl = *fd;
        fdl.ctr = c->d_ctr + 8;
    }
    KSplit ks{};
    int grid = ntiles;
    if (syrk_order == 2) {
        // every XCD walks S tiles on its slots / 8 workgroup slots: its last partial round, when thin, runs as quadrants
        // (quadrant 0 by the tile's own workgroup, 1 .. 3 by workgroups behind the grid)
        const int S = (ntiles + 7) / 8, per = slots / 8;
        const int S_full = (S / per) * per, Rx = S - S_full;
        grid = 8 * S;
        if (c->tune.ksplit && K % GK == 0 && Rx > 0 && 8 * Rx <= c->tune.ksplit_max) {
            ks = KSplit{S_full, 4};
            grid += 3 * 8 * Rx;
        }
    } else
    if (c->tune.ksplit && syrk_order == 0 && K % GK == 0) {
        int first = ntiles;  // first tile computed as quadrants
        const int bfull = (ntiles / slots) * slots, R = ntiles - bfull;
        if (R > 0 && R <= c->tune.ksplit_max) first = bfull;
        // A last tile row with few valid rows -- the augmented row y^T of the marginal likelihood makes
        // every trailing update end in a row of T tiles with ONE valid row -- costs a whole tile time
        // per tile (2.4 % of all tiles at N = 16384); as quadrants only the MFMA tiles that hold
        // valid rows are multiplied (1/8 of the work).
        const int vlast = M - (T - 1) * GT;
        if (T > 1 && vlast <= 64 && ntiles - TN < first) first = ntiles - TN;  // the last tile row has TN tiles
        if ((ord & 0x200) && first < 1) first = 1;  // tile 0 belongs to the sub-tile blocks
        if (first < ntiles) {
            ks = KSplit{first, 4};
            grid = first + 4 * (ntiles - first);
        }
    }
    if (ord & 0x200) grid += SUBWG - 1;
    hipLaunchKernelGGL(k_gemm_nt<1>, dim3(grid), 256, 0, s, P, ldp, P, ldp, C, ldc, M, N, K, ord, stg, fdl, ks);
    return fuse;
}

#ifdef GPMI_PROBES
void launch_syrk_probe(const gpmi_ctx *c, hipStream_t s, const double *P, size_t ldp, double *C, size_t ldc, int m, int k)
{
    launch_syrk_lower(c, s, P, ldp, C, ldc, m, m, k, c->d_ctr, c->ncu);
}
#endif

// C(lower) -= U U^T for an upper-triangular n x n U
void launch_syrk_uut(const gpmi_ctx *c, hipStream_t s, const double *U, size_t ldu, double *C, size_t ldc, int n)
{
    launch_syrk_lower(c, s, U, ldu, C, ldc, n, n, n, nullptr, 0, 1);
}

// tune.block_recursive  1: in-block updates by recursive halving; 0: 128 / 256 / rest-of-block levels

// One 128-column panel [k, k + kb): diagonal block, then the rows [max(k + kb, row_lo), row_hi) below it.
static double *fpack_slot(gpmi_ctx *c, double *Fpack_all, int ko, int k)
{
    return Fpack_all ? Fpack_all + (size_t)(k / GPMI_NB) * GPMI_FPACK
                     : c->Fpack + (size_t)(((k - ko) / GPMI_NB) % GPMI_FPACK_SLOTS) * GPMI_FPACK;
}

static void panel_one(gpmi_ctx *c, double *W, size_t ld, int *d_info, double *Fpack_all, int ko, int k, int kb,
                      int row_lo, int row_hi, bool with_diag, hipStream_t s)
{
    double *Fp = fpack_slot(c, Fpack_all, ko, k);
    if (with_diag) {
#ifdef GPMI_PROBES
        if (c->tune.diag_waves == 5)
            hipLaunchKernelGGL(k_potrf_diag, dim3(1), 320, 0, s, W + (size_t)k + (size_t)k * ld, ld, kb, Fp, d_info, k);
        else
#endif
            hipLaunchKernelGGL(k_potrf_diag4, dim3(1), 256, 0, s, W + (size_t)k + (size_t)k * ld, ld, kb, Fp, d_info, k);
    }
    const int r0 = k + kb;
    const int rlo = r0 > row_lo ? r0 : row_lo;
    if (rlo >= row_hi) return;
    hipLaunchKernelGGL(k_trsm_panel, dim3((row_hi - rlo + 63) / 64), 256, 0, s, W + (size_t)k * ld, ld, rlo, row_hi, kb, Fp);
}

// Columns [k0, k1) by recursive halving at panel boundaries: factor the left half, update the
// right half with it in ONE product (K = width of the left half: 128, 256, 512 for a 1024-column
// block), factor the right half.  Compared with fixed 128 / 256 levels the same flops move from
// K = 256 products to a K = 512 one, whose per-tile prologue and C round trip weigh half as much.
static void panel_rec(gpmi_ctx *c, double *W, size_t ld, int *d_info, double *Fpack_all, int ko, int k0, int k1,
                      int row_lo, int row_hi, bool with_diag, bool first_diag_done, hipStream_t s)
{
    const int NB = GPMI_NB;
    if (k1 - k0 <= NB) {
        panel_one(c, W, ld, d_info, Fpack_all, ko, k0, k1 - k0, row_lo, row_hi, with_diag && !first_diag_done, s);
        return;
    }
    const int npan = (k1 - k0 + NB - 1) / NB;
    const int km = k0 + ((npan + 1) / 2) * NB;
    panel_rec(c, W, ld, d_info, Fpack_all, ko, k0, km, row_lo, row_hi, with_diag, first_diag_done, s);
    const int rlo = km > row_lo ? km : row_lo;
    bool fused = false;
    if (rlo < row_hi) {  // rows [rlo, row_hi) x cols [km, k1) -= A[rows, k0:km] A[km:k1, k0:km]^T
        const double *A = W + (size_t)rlo + (size_t)k0 * ld, *B = W + (size_t)km + (size_t)k0 * ld;
        double *C = W + (size_t)rlo + (size_t)km * ld;
        if (with_diag && rlo == km) {  // C starts at the next panel's diagonal block: factor it in this launch
            const int kb = (k1 - km < NB) ? k1 - km : NB;
            fused = launch_gemm_nt_fused(c, s, A, ld, B, ld, C, ld, row_hi - rlo, k1 - km, km - k0,
                                         FuseDiag{fpack_slot(c, Fpack_all, ko, km), d_info, km, kb, c->d_ctr + 8});
        }
        if (!fused) launch_gemm_nt(c, s, A, ld, B, ld, C, ld, row_hi - rlo, k1 - km, km - k0, 1);
    }
    panel_rec(c, W, ld, d_info, Fpack_all, ko, km, k1, row_lo, row_hi, with_diag, fused, s);
}

// Panel work of one outer block [ko, ke) restricted to the rows [row_lo, row_hi) below each
// panel.  with_diag: also factor the 128 x 128 diagonal blocks (their factors go to the block's
// Fpack slots); without it the slots written by an earlier call are used.
//   (0, M, true)   : the whole panel phase, full height
//   (0, ke, true)  : only the NBO x NBO diagonal block -- the latency chain of the look-ahead
//   (ke, M, false) : the rows below it -- wide, throughput-bound kernels
// tune.block_recursive == 0 keeps the earlier three fixed levels: 128-column panels grouped into
// middle blocks of NBM columns; a panel's K = 128 update reaches only to the end of its middle
// block, the rest of the outer block is updated once per middle block with K = NBM.
static void panel_rows(gpmi_ctx *c, double *W, size_t ld, int *d_info, double *Fpack_all, int ko, int ke, int NBO,
                       int row_lo, int row_hi, bool with_diag, hipStream_t s, bool first_diag_done = false)
{
    if (c->tune.block_recursive) {
        panel_rec(c, W, ld, d_info, Fpack_all, ko, ko, ke, row_lo, row_hi, with_diag, first_diag_done, s);
        return;
    }
    const int NB = GPMI_NB;
    const int NBM = (NBO >= 512) ? 256 : NBO;
    for (int km = ko; km < ke; km += NBM) {
        const int kme = (km + NBM < ke) ? km + NBM : ke;
        for (int k = km; k < kme; k += NB) {
            const int kb = (kme - k < NB) ? kme - k : NB;
            panel_one(c, W, ld, d_info, Fpack_all, ko, k, kb, row_lo, row_hi, with_diag && !(first_diag_done && k == ko), s);
            const int r0 = k + kb;
            const int rlo = r0 > row_lo ? r0 : row_lo;
            if (rlo >= row_hi) continue;
            if (r0 < kme)  // rest of this middle block: rows [rlo, row_hi) x cols [r0, kme), K = kb
                launch_gemm_nt(c, s, W + (size_t)rlo + (size_t)k * ld, ld, W + (size_t)r0 + (size_t)k * ld, ld,
                               W + (size_t)rlo + (size_t)r0 * ld, ld, row_hi - rlo, kme - r0, kb, 1);
        }
        const int rlo = kme > row_lo ? kme : row_lo;
        if (kme < ke && rlo < row_hi)  // rest of the outer block: cols [kme, ke), K = kme - km
            launch_gemm_nt(c, s, W + (size_t)rlo + (size_t)km * ld, ld, W + (size_t)kme + (size_t)km * ld, ld,
                           W + (size_t)rlo + (size_t)kme * ld, ld, row_hi - rlo, ke - kme, kme - km, 1);
    }
}

int launch_potrf_partial(gpmi_ctx *c, double *W, size_t ld, int M, int ncol, int nfac, int *d_info,
                         double *Fpack_all)
{
    // Blocked right-looking factorisation of the leading nfac columns of the M x ncol lower
    // trapezoid in W; the trailing [nfac, ncol) part receives the Schur complement.
    // outer block width: K of the trailing update.  Wider blocks cut the C traffic and the
    // number of epilogues once the trailing matrix is large; 256 keeps the panel phase short.
    // small matrices: the whole partial factorisation in ONE workgroup of one launch (k_potrf_small) instead of a
    // chain of latency-bound launches (tune.small_m; callers that keep the packed factors take the blocked path)
    if (!Fpack_all && M <= c->tune.small_m && M > 0 && nfac > 0) {
        small_lds_attr();
        hipLaunchKernelGGL(k_potrf_small, dim3(1), 256, SMALL_LDS_DOUBLES * sizeof(double), c->stream, W, ld, M, ncol, nfac, d_info);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return gpmi_fail(GPMI_EHIP, "potrf launch failed: %s", hipGetErrorString(e));
        return 0;
    }
    auto nbo_for = [c](int cols) {
        return cols >= c->tune.nb_thr[0] ? 1024 : (cols >= c->tune.nb_thr[1] ? 512 : (cols >= c->tune.nb_thr[2] ? 256 : 128));
    };
    const int NBO = c->nb_outer > 0 ? c->nb_outer : nbo_for(ncol);
#ifdef GPMI_PROBES
    // (probe build) two-stream look-ahead over outer blocks: needs at least three outer blocks to pay, a second
    // stream, and the block's packed factors in the ring (or all kept); measured not to pay, DESIGN section 4
    bool la = c->lookahead > 0 && nfac >= 3 * NBO && (Fpack_all || NBO / GPMI_NB <= GPMI_FPACK_SLOTS);
    if (la) {
        int rc = gpmi_lookahead_streams(c);
        if (rc) return rc;
        la = c->nq >= 2 || c->pstream;
    }
    if (!la)
#endif
    {
        hipStream_t s = c->stream;
        // auto width follows the order of the matrix still to update (tune.nb_adapt): the last blocks of a
        // large matrix are a small matrix, whose few trailing tiles do not fill the chip at K = 1024
        bool diag_done = false;
        for (int ko = 0, nbo = NBO; ko < nfac; ko += nbo) {
            nbo = (c->nb_outer > 0 || !c->tune.nb_adapt) ? NBO : nbo_for(ncol - ko);
            const int ke = (ko + nbo < nfac) ? ko + nbo : nfac;
            kt_begin(c, 2, s);
            panel_rows(c, W, ld, d_info, Fpack_all, ko, ke, nbo, 0, M, true, s, diag_done);
            {   // algorithmic flops of the panel phase: diagonal block w^3/3 + rows below (m - w) w^2
                const double w = (double)(ke - ko), m = (double)(M - ko);
                kt_end(c, 2, w * w * (m - w) + w * w * w / 3.0, s);
            }
            diag_done = false;
            if (ke >= M || ke >= ncol) continue;
            const double mt = (double)(ncol - ke), extra = (double)(M - ncol);
            // the first diagonal block of the next outer block rides in this trailing update
            FuseDiag fd{};
            if (ke < nfac)
                fd = FuseDiag{fpack_slot(c, Fpack_all, ke, ke), d_info, ke, (nfac - ke < GPMI_NB) ? nfac - ke : GPMI_NB, nullptr};
            kt_begin(c, 1, s);
            diag_done = launch_syrk_lower(c, s, W + (size_t)ke + (size_t)ko * ld, ld, W + (size_t)ke + (size_t)ke * ld, ld,
                                          M - ke, ncol - ke, ke - ko, c->d_ctr, c->ncu, 0, &fd);
            // algorithmic flops: lower triangle (incl. diagonal) of the square part + extra rows
            {   // flag: more than one round of tiles (throughput-bound launch)
                const int T = (M - ke + GT - 1) / GT, TNf = (ncol - ke + GT - 1) / GT, TN = TNf < T ? TNf : T;
                kt_end(c, 1, (mt * (mt + 1.0) + 2.0 * extra * mt) * (double)(ke - ko), s,
                       syrk_grid(T, TN, 0) > 2 * (c->ncu > 0 ? c->ncu : 256));
            }
        }
    }
#ifdef GPMI_PROBES
    else {
        // Look-ahead over outer blocks on TWO streams.  With U(j) the trailing update by block j,
        // split by columns into U1(j) (block column j + 1, all rows below) and U2(j) (the rest):
        //   sB (panel stream): P(0) | U1(0) P(1) | U1(1) P(2) | ...   U1(j) waits for U2(j - 1)
        //   sA (bulk stream):        | U2(0)      | U2(1)      | ...   U2(j) waits for P(j)
        // The latency-bound panel phase P(j + 1) -- 8 x (diagonal block, panel solve, in-block
        // product) per 1024 columns, a chain of small kernels -- and the thin U1(j) run while U2(j)
        // fills the chip from the other stream: sA goes from one bulk update straight into the next
        // as long as U1(j) + P(j + 1) take less time than U2(j).  The two streams are the context's
        // calibrated dispatch streams (hardware queues on different command-processor pipes: kernels
        // of queues that share a pipe are time-sliced instead of overlapped).
        hipStream_t const caller = c->stream;
        hipStream_t sA = caller, sB = c->pstream;
        if (c->nq >= 2) {
            sA = c->qstream[0];
            sB = c->qstream[1];
        }
        (void)hipEventRecord(c->evM, caller);
        if (sA != caller) (void)hipStreamWaitEvent(sA, c->evM, 0);
        (void)hipStreamWaitEvent(sB, c->evM, 0);
        {
            const int ke0 = NBO < nfac ? NBO : nfac;
            kt_begin(c, 2, sB);
            panel_rows(c, W, ld, d_info, Fpack_all, 0, ke0, NBO, 0, M, true, sB);
            const double w = (double)ke0, m = (double)M;
            kt_end(c, 2, w * w * (m - w) + w * w * w / 3.0, sB);
            (void)hipEventRecord(c->evP, sB);
        }
        bool u2_pending = false;
        for (int ko = 0; ko < nfac; ko += NBO) {
            const int ke = (ko + NBO < nfac) ? ko + NBO : nfac;
            const int K = ke - ko;
            if (ke >= M || ke >= ncol) break;
            const double *Pj = W + (size_t)ko * ld;   // panel j: rows [ke, M) of columns [ko, ke)
            if (ke >= nfac) {  // last factored block: Schur complement / augmented rows, nothing follows
                (void)hipStreamWaitEvent(sA, c->evP, 0);
                const double mt = (double)(ncol - ke), extra = (double)(M - ncol);
                kt_begin(c, 1, sA);
                launch_syrk_lower(c, sA, Pj + ke, ld, W + (size_t)ke + (size_t)ke * ld, ld, M - ke, ncol - ke, K, c->d_ctr, c->ncu);
                kt_end(c, 1, (mt * (mt + 1.0) + 2.0 * extra * mt) * (double)K, sA);
                break;
            }
            const int ke2 = (ke + NBO < nfac) ? ke + NBO : nfac;
            const int w1 = ke2 - ke;
            // U1(j) on sB: rows [ke, M) x columns [ke, ke2); its tile (0, 0) is the next diagonal block
            if (u2_pending) (void)hipStreamWaitEvent(sB, c->evU, 0);
            FuseDiag fd{fpack_slot(c, Fpack_all, ke, ke), d_info, ke, (nfac - ke < GPMI_NB) ? nfac - ke : GPMI_NB, nullptr};
            kt_begin(c, 1, sB);
            const bool fused = launch_syrk_lower(c, sB, Pj + ke, ld, W + (size_t)ke + (size_t)ke * ld, ld, M - ke, w1, K, c->d_ctr,
                                                 c->ncu, 0, &fd);
            kt_end(c, 1, ((double)w1 * ((double)w1 + 1.0) + 2.0 * (double)(M - ke2) * (double)w1) * (double)K, sB);
            // U2(j) on sA: rows [ke2, M) x columns [ke2, ncol)
            u2_pending = false;
            if (ke2 < M && ke2 < ncol) {
                (void)hipStreamWaitEvent(sA, c->evP, 0);
                const double mt = (double)(ncol - ke2), extra = (double)(M - ncol);
                kt_begin(c, 1, sA);
                launch_syrk_lower(c, sA, Pj + ke2, ld, W + (size_t)ke2 + (size_t)ke2 * ld, ld, M - ke2, ncol - ke2, K, c->d_ctr, c->ncu);
                kt_end(c, 1, (mt * (mt + 1.0) + 2.0 * extra * mt) * (double)K, sA);
                (void)hipEventRecord(c->evU, sA);
                u2_pending = true;
            }
            // P(j + 1) on sB
            kt_begin(c, 2, sB);
            panel_rows(c, W, ld, d_info, Fpack_all, ke, ke2, NBO, 0, M, true, sB, fused);
            {
                const double w = (double)w1, m = (double)(M - ke);
                kt_end(c, 2, w * w * (m - w) + w * w * w / 3.0, sB);
            }
            (void)hipEventRecord(c->evP, sB);
        }
        if (sA != caller) {
            (void)hipEventRecord(c->evM, sA);
            (void)hipStreamWaitEvent(caller, c->evM, 0);
        }
        (void)hipEventRecord(c->evU, sB);
        (void)hipStreamWaitEvent(caller, c->evU, 0);
    }
#endif  // GPMI_PROBES
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return gpmi_fail(GPMI_EHIP, "potrf launch failed: %s", hipGetErrorString(e));
    return 0;
}

// Columns [c0, c1) of X <- X L^-T by recursive halving at panel boundaries: solve the left half, take
// it out of the right half in ONE product (K = width of the left half: the bulk of the N^3 flops runs
// at K in the thousands instead of K = 128), solve the right half.  upper_tri: X[i][j] = 0 for i > j --
// column block [c0, c1) has rows [0, c1) only, and the product skips the zero columns of its
// trapezoidal A operand per row-tile.
static void trsm_right_rec(const gpmi_ctx *c, hipStream_t s, const double *L, size_t ldl, double *X, size_t ldx, int mrows,
                           const double *Fpack_all, int upper_tri, int c0, int c1)
{
    const int NB = GPMI_NB;
    const bool lower_out = upper_tri == 2;  // only the rows >= a column block's first column are wanted (and kept valid)
    if (c1 - c0 <= NB) {
        const int mr = (upper_tri == 1 && c1 < mrows) ? c1 : mrows;
        const int r0 = lower_out ? c0 : 0;
        if (r0 < mr)
            hipLaunchKernelGGL(k_trsm_panel, dim3((mr - r0 + 63) / 64), 256, 0, s, X + (size_t)c0 * ldx, ldx, r0, mr, c1 - c0,
                               Fpack_all + (size_t)(c0 / NB) * GPMI_FPACK);
        return;
    }
    const int npan = (c1 - c0 + NB - 1) / NB;
    const int cm = c0 + ((npan + 1) / 2) * NB;
    trsm_right_rec(c, s, L, ldl, X, ldx, mrows, Fpack_all, upper_tri, c0, cm);
    const int mr = (upper_tri == 1 && cm < mrows) ? cm : mrows;  // rows where X[:, c0:cm] is non-zero
    const int r0 = lower_out ? cm : 0;                           // rows of the right half that are wanted
    const double *A = X + (size_t)r0 + (size_t)c0 * ldx, *B = L + (size_t)cm + (size_t)c0 * ldl;
    double *C = X + (size_t)r0 + (size_t)cm * ldx;
    if (upper_tri == 1 && (c->tune.gemm_variant == 3 || c->tune.gemm_variant == 0)) {
        dim3 grid((c1 - cm + GT - 1) / GT, (mr + GT - 1) / GT);  // x: column tiles, y: row tiles (long K first)
        hipLaunchKernelGGL(k_gemm_nt<0>, grid, 256, 0, s, A, ldx, B, ldl, C, ldx, mr, c1 - cm, cm - c0, 0x100, 0,
                           FuseDiag{nullptr, nullptr, c0, 0, nullptr}, KSplit{});
    } else if (mr > r0) {
        launch_gemm_nt(c, s, A, ldx, B, ldl, C, ldx, mr - r0, c1 - cm, cm - c0, 1);
    }
    trsm_right_rec(c, s, L, ldl, X, ldx, mrows, Fpack_all, upper_tri, cm, c1);
}

int launch_trsm_right(gpmi_ctx *c, const double *L, size_t ldl, int n, double *X, size_t ldx,
                      int mrows, const double *Fpack_all, int upper_tri)
{
    // X <- X L^-T by block forward substitution over the 128-column panels.  upper_tri = 1: X is
    // upper triangular on entry (e.g. the identity) and stays so -- panel k then only has rows
    // [0, k + kb), which cuts the work to a third.  upper_tri = 2: only the LOWER triangle of the result is
    // wanted (rows >= the column's block start; the rest of X is left half-updated): the rows of a column block
    // need, from the blocks to their left, those same rows only -- also a third of the work.
    hipStream_t s = c->stream;
    const int NB = GPMI_NB;
    // many right-hand rows: products with large K.  For a triangular X the recursion pays from
    // N ~ 12k on (value + gradient at N = 16384: 88.9 -> 83.2 ms; N = 8192: 16.0 vs 16.5 ms, N = 4096:
    // 4.5 vs 5.2 ms -- unequal tile lengths and more launches)
    if (mrows >= 2 * GT && n > NB && (upper_tri != 1 || n >= 12288)) {
        trsm_right_rec(c, s, L, ldl, X, ldx, mrows, Fpack_all, upper_tri, 0, n);
    } else {  // a few rows (triangular solve of vectors): one pass over L, panel by panel
        for (int k = 0; k < n; k += NB) {
            const int kb = (n - k < NB) ? n - k : NB;
            const double *Fp = Fpack_all + (size_t)(k / NB) * GPMI_FPACK;
            const int mr = (upper_tri == 1 && k + kb < mrows) ? k + kb : mrows;  // (lower-only output: everything, it is small)
            hipLaunchKernelGGL(k_trsm_panel, dim3((mr + 63) / 64), 256, 0, s, X + (size_t)k * ldx, ldx, 0, mr, kb, Fp);
            const int r0 = k + kb;
            if (r0 < n)
                launch_gemm_nt(c, s, X + (size_t)k * ldx, ldx, L + (size_t)r0 + (size_t)k * ldl, ldl,
                               X + (size_t)r0 * ldx, ldx, mr, n - r0, kb, 1);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return gpmi_fail(GPMI_EHIP, "trsm launch failed: %s", hipGetErrorString(e));
    return 0;
}

// Dinv (ceil(n / 128) blocks of 128 x 128 doubles): inverses of L's diagonal blocks from its packed factors;
// tmp: the same size of scratch
void launch_diag_inverses(hipStream_t s, const double *Fpack_all, int n, double *Dinv, double *tmp)
{
    if (n <= 0) return;
    const int nblk = (n + TSV - 1) / TSV;
    hipLaunchKernelGGL(k_eye_blocks, dim3(nblk), 256, 0, s, tmp, nblk);
    for (int w = 0; w < nblk; ++w) {
        const int kb = (n - w * TSV < TSV) ? n - w * TSV : TSV;
        hipLaunchKernelGGL(k_trsm_panel, dim3(2), 256, 0, s, tmp + (size_t)w * TSV * TSV, (size_t)TSV, 0, TSV, kb,
                           Fpack_all + (size_t)w * GPMI_FPACK);
    }
    hipLaunchKernelGGL(k_transpose_blocks, dim3(nblk, 64), 256, 0, s, tmp, Dinv);
}

// t = L^-1 k (k, t: n contiguous doubles, different buffers) in one launch; Dinv from launch_diag_inverses
int launch_trsv_lower(hipStream_t s, const double *L, size_t ldl, int n, const double *k, double *t, const double *Dinv,
                      int *ticket /* zeroed device int owned by the calling context; left zero */)
{
    if (n <= 0) return 0;
    hipError_t e = hipMemsetAsync(t, 0xff, (size_t)n * sizeof(double), s);  // "not yet" in every element
    if (e != hipSuccess) return gpmi_fail(GPMI_EHIP, "memset failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(k_trsv_wave, dim3((n + TSV - 1) / TSV), 256, 0, s, L, ldl, n, k, t, Dinv, ticket);
    e = hipGetLastError();
    if (e != hipSuccess) return gpmi_fail(GPMI_EHIP, "trsv launch failed: %s", hipGetErrorString(e));
    return 0;
}

// ---- small-N evaluations: one workgroup each ------------------------------------------------
// workspace slice of one n-point problem: leading dimension and stride (doubles) between consecutive slices
void small_ws_layout(int n, size_t *ld, size_t *stride)
{
    *ld = (size_t)(((n + 1 + 15) / 16) * 16 + 16);
    *stride = *ld * (size_t)(n + 1) + 256;  // tile loads may over-read rows past the end of the last column
}

void launch_logml_small(hipStream_t s, const double *dX, int n, int ldx, const double *dy, const SeParams &p, double diag_add,
                        double *W, size_t ld, double *d_out3, int *d_info_out, int *d_info_work, double *stage, int *done, int seq)
{
    small_lds_attr();
    hipLaunchKernelGGL(k_logml_small, dim3(1), 256, SMALL_LDS_DOUBLES * sizeof(double), s, dX, n, ldx, dy, p, diag_add, W, ld, d_out3,
                       d_info_out, d_info_work, h_exp, stage, done, seq);
}

// G <= GPMI_SMALL_PTS points (alpha, rho, sigma) in ONE launch of G workgroups; Wall: G slices (small_ws_layout)
void launch_logml_small_batch(hipStream_t s, const double *dX, int n, int ldx, int D, const double *dy, const double *alpha,
                              const double *rho, const double *sigma, int G, double jitter, double *Wall, double *d_out3,
                              int *d_info_out, int *d_info_work)
{
    static_assert(SMALL_PTS == GPMI_SMALL_PTS, "batch size of the small-N grid launch");
    SmallBatch b;
    for (int g = 0; g < G; ++g) {
        b.a2[g] = alpha[g] * alpha[g];
        b.inv_rho[g] = 1.0 / rho[g];
        b.diag[g] = sigma[g] * sigma[g] + jitter;
    }
    size_t ld, stride;
    small_ws_layout(n, &ld, &stride);
    small_lds_attr();
    hipLaunchKernelGGL(k_logml_small_batch, dim3(G), 256, SMALL_LDS_DOUBLES * sizeof(double), s, dX, n, ldx, D, dy, b, Wall, stride, ld, d_out3, d_info_out,
                       d_info_work, h_exp);
}

// G <= GPMI_SMALL_PTS_ARD points with a length-scale per dimension: ell is G x D, point-major
void launch_logml_small_batch_ard(hipStream_t s, const double *dX, int n, int ldx, int D, const double *dy, const double *alpha,
                                  const double *ell, const double *sigma, int G, double jitter, double *Wall, double *d_out3,
                                  int *d_info_out, int *d_info_work)
{
    static_assert(SMALL_PTS_ARD == GPMI_SMALL_PTS_ARD, "batch size of the small-N ARD grid launch");
    SmallBatchArd b;
    for (int g = 0; g < G; ++g) {
        b.a2[g] = alpha[g] * alpha[g];
        b.diag[g] = sigma[g] * sigma[g] + jitter;
        for (int d = 0; d < GPMI_MAXD; ++d) b.inv_ell[g][d] = d < D ? 1.0 / ell[(size_t)g * D + d] : 0.0;
    }
    size_t ld, stride;
    small_ws_layout(n, &ld, &stride);
    small_lds_attr();
    hipLaunchKernelGGL(k_logml_small_batch_ard, dim3(G), 256, SMALL_LDS_DOUBLES * sizeof(double), s, dX, n, ldx, D, dy, b, Wall,
                       stride, ld, d_out3, d_info_out, d_info_work, h_exp);
}

// G points (any number) whose parameters are uploaded to d_par (G * (2 + GPMI_MAXD) doubles) in stream order; ell: one
// length-scale per point (n_ell == 1) or D per point (point-major); Wall: G slices (small_ws_layout)
void launch_logml_small_batch_dev(hipStream_t s, const double *dX, int n, int ldx, int D, const double *dy, const double *alpha,
                                  const double *ell, int n_ell, const double *sigma, int G, double jitter, double *d_par,
                                  double *Wall, double *d_out3, int *d_info_out, int *d_info_work)
{
    static_assert(PUT_MAX % SMALL_PAR == 0, "whole points per upload");
    PutArgs a;
    for (int g0 = 0; g0 < G; g0 += PUT_MAX / SMALL_PAR) {
        const int gc = (G - g0 < PUT_MAX / SMALL_PAR) ? G - g0 : PUT_MAX / SMALL_PAR;
        for (int g = 0; g < gc; ++g) {
            double *q = a.v + g * SMALL_PAR;
            q[0] = alpha[g0 + g] * alpha[g0 + g];
            q[1] = sigma[g0 + g] * sigma[g0 + g] + jitter;
            for (int d = 0; d < GPMI_MAXD; ++d)
                q[2 + d] = d < D ? 1.0 / (n_ell == 1 ? ell[g0 + g] : ell[(size_t)(g0 + g) * D + d]) : 0.0;
        }
        hipLaunchKernelGGL(k_put_doubles, dim3(1), 256, 0, s, a, d_par + (size_t)g0 * SMALL_PAR, gc * SMALL_PAR);
    }
    size_t ld, stride;
    small_ws_layout(n, &ld, &stride);
    small_lds_attr();
    hipLaunchKernelGGL(k_logml_small_batch_dev, dim3(G), 256, SMALL_LDS_DOUBLES * sizeof(double), s, dX, n, ldx, D, dy, d_par, Wall,
                       stride, ld, d_out3, d_info_out, d_info_work, h_exp);
}

// value + gradient sums by one workgroup per point: W holds, per point, two slices of small_ws_layout (W, then U)
void launch_logml_grad_small(hipStream_t s, const double *dX, int n, int ldx, const double *dy, const SeParams &p, double diag_add,
                             double *W, double *d_res, int *d_info_out, int *d_info_work, double *stage, int *done, int seq)
{
    size_t ld, stride;
    small_ws_layout(n, &ld, &stride);
    small_lds_attr();
    hipLaunchKernelGGL(k_logml_grad_small, dim3(1), 256, SMALL_GRAD_LDS_DOUBLES * sizeof(double), s, dX, n, ldx, dy, p, diag_add, W, ld,
                       W + stride, d_res, d_info_out, d_info_work, h_exp, stage, done, seq);
}

void launch_logml_grad_small_batch(hipStream_t s, const double *dX, int n, int ldx, int D, const double *dy, const double *alpha,
                                   const double *rho, const double *sigma, int G, double jitter, double *Wall, double *d_res,
                                   int *d_info_out, int *d_info_work, double *stage, int *done, int seq, int *arrive)
{
    SmallBatch b;
    for (int g = 0; g < G; ++g) {
        b.a2[g] = alpha[g] * alpha[g];
        b.inv_rho[g] = 1.0 / rho[g];
        b.diag[g] = sigma[g] * sigma[g] + jitter;
    }
    size_t ld, stride;
    small_ws_layout(n, &ld, &stride);
    small_lds_attr();
    hipLaunchKernelGGL(k_logml_grad_small_batch, dim3(G), 256, SMALL_GRAD_LDS_DOUBLES * sizeof(double), s, dX, n, ldx, D, dy, b, Wall,
                       2 * stride, stride, ld, d_res, d_info_out, d_info_work, h_exp, stage, done, seq, arrive);
}

// B draws by one workgroup each; Wall: B slices of small_ws_layout(n + m); dY: n x B, dZ, draws, mus: m x B (packed); d_par: 3 B
// doubles, d_info_work: 2 B ints
void launch_sample_derivs_small_batch(hipStream_t s, const double *dt, int n, const double *dts, int m, const double *dY,
                                      const double *params /* host: (l, a, sy) per draw */, int B, double jitter, const double *dZ,
                                      double *d_par, double *Wall, double *d_draws, double *d_mus, int *d_status, int *d_info_work)
{
    static_assert(PUT_MAX % 3 == 0, "whole draws per upload");
    PutArgs a;
    for (int g0 = 0; g0 < B; g0 += PUT_MAX / 3) {
        const int gc = (B - g0 < PUT_MAX / 3) ? B - g0 : PUT_MAX / 3;
        for (int q = 0; q < 3 * gc; ++q) a.v[q] = params[3 * (size_t)g0 + q];
        hipLaunchKernelGGL(k_put_doubles, dim3(1), 256, 0, s, a, d_par + 3 * (size_t)g0, 3 * gc);
    }
    size_t ld, stride;
    small_ws_layout(n + m, &ld, &stride);
    small_lds_attr();
    hipLaunchKernelGGL(k_sample_derivs_small_batch, dim3(B), 256, SMALL_LDS_DOUBLES * sizeof(double), s, dt, n, dts, m, dY, d_par, jitter, dZ,
                       Wall, stride, ld, d_draws, d_mus, d_status, d_info_work);
}

void launch_gp_condition_small(hipStream_t s, const double *t, int n, const double *ts, int m, const double *y, int kindK, int kindS,
                               int kindSS, int compat, double a2, double l2, double s2, double jitter, double *W, double *Kn, size_t ldo,
                               double *mn, int *info_out, int *d_info_work, double *stage, int *done, int seq)
{
    size_t ld, stride;
    small_ws_layout(n + m, &ld, &stride);
    small_lds_attr();
    CondArgs q{kindK, kindS, kindSS, compat, a2, l2, s2, jitter};
    hipLaunchKernelGGL(k_gp_condition_small, dim3(1), 256, SMALL_LDS_DOUBLES * sizeof(double), s, t, n, ts, m, y, q, W, ld, Kn, ldo, mn,
                       info_out, d_info_work, stage, done, seq);
}

void launch_exact_gp_small(hipStream_t s, const double *X, int n, int ldx, const double *z, const SeParams &p, double diag_add,
                           double *W, double *f, int *info_out, int *d_info_work, double *stage, int *done, int seq)
{
    size_t ld, stride;
    small_ws_layout(n, &ld, &stride);
    small_lds_attr();
    hipLaunchKernelGGL(k_exact_gp_small, dim3(1), 256, SMALL_LDS_DOUBLES * sizeof(double), s, X, n, ldx, z, p, diag_add, W, ld, f, info_out,
                       d_info_work, h_exp, stage, done, seq);
}

// P <= 64 length-scales, one workgroup each (n <= 128); Wall: 3 P slices of small_ws_layout(n)
void launch_rbf_cov_chol_small(hipStream_t s, const double *x, int n, const double *ls, int P, double *Wall, double *Lout, double *dLout,
                               size_t ostride, size_t ldo, int *info_out, int *d_info_work, double *stage)
{
    RbfBatch b;
    for (int p = 0; p < P; ++p) b.l[p] = ls[p];
    size_t ld, stride;
    small_ws_layout(n, &ld, &stride);
    small_lds_attr();
    hipLaunchKernelGGL(k_rbf_cov_chol_small, dim3(P), 256, SMALL_LDS_DOUBLES * sizeof(double), s, x, n, b, Wall, stride, ld, Lout, dLout,
                       ostride, ldo, info_out, d_info_work, h_exp, stage);
}

void launch_pack_factors(hipStream_t s, const double *L, size_t ldl, int n, double *Fpack_all)
{
    for (int k = 0; k < n; k += GPMI_NB) {
        const int kb = (n - k < GPMI_NB) ? n - k : GPMI_NB;
        hipLaunchKernelGGL(k_pack_factors, dim3(1), 512, 0, s, L + (size_t)k + (size_t)k * ldl, ldl, kb,
                           Fpack_all + (size_t)(k / GPMI_NB) * GPMI_FPACK);
    }
}

void launch_get_row(hipStream_t s, const double *W, size_t ld, int row, int col0, int m, double scale, double *out)
{
    if (m <= 0) return;
    hipLaunchKernelGGL(k_get_row, dim3((m + 255) / 256), 256, 0, s, W, ld, row, col0, m, scale, out);
}

// part: 2 * ceil(n / 256) doubles of scratch
void launch_logml_finalize(hipStream_t s, const double *W, size_t ld, int n, int zrow,
                           const int *d_info, double *d_out3, int *d_info_out, double *part)
{
    const int nslice = (n + FIN_SLICE - 1) / FIN_SLICE;
    hipLaunchKernelGGL(k_logml_partial, dim3(nslice), FIN_SLICE, 0, s, W, ld, n, zrow, part);
    hipLaunchKernelGGL(k_logml_finalize, dim3(1), 256, 0, s, part, nslice, n, d_info, d_out3, d_info_out);
}

int trmv_lower_chunks(int n) { return (n + TMV_COLS - 1) / TMV_COLS; }

// part: trmv_lower_chunks(n) * n doubles of scratch
void launch_trmv_lower(hipStream_t s, const double *L, size_t ldl, int n, const double *z, double *f, double *part)
{
    if (n <= 0) return;
    const int nchunk = trmv_lower_chunks(n);
    hipLaunchKernelGGL(k_trmv_lower_part, dim3((n + 255) / 256, nchunk), 256, 0, s, L, ldl, n, z, part);
    hipLaunchKernelGGL(k_trmv_lower_sum, dim3((n + 255) / 256), 256, 0, s, part, n, nchunk, f);
}

#ifdef GPMI_PROBES
int probe_fused_read(hipStream_t s, unsigned long long *out5)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out5, HIP_SYMBOL(g_fz), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_fz), z, sizeof z) != hipSuccess;
}

int probe_body_read(hipStream_t s, unsigned long long *out8)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_body), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
    return hipMemcpyToSymbol(HIP_SYMBOL(g_body), z, sizeof z) != hipSuccess;
}

int probe_clock_read(hipStream_t s, int reset, unsigned long long *out3)
{
    unsigned long long h[4] = {0, 0, 0, 0};
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk), sizeof h) != hipSuccess) return 1;
    for (int i = 0; i < 3; ++i) out3[i] = h[i];
    if (reset) {
        unsigned long long z[4] = {0, 0, 0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_clk), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}

void launch_probe_mfma(hipStream_t s, const double *A, const double *B, double *D)
{
    hipLaunchKernelGGL(k_probe_mfma, dim3(1), 64, 0, s, A, B, D);
}

void launch_probe_peak(hipStream_t s, double *sink, int iters, int *blocks, int *threads)
{
    *blocks = 256 * 4;
    *threads = 256;
    hipLaunchKernelGGL(k_probe_peak, dim3(*blocks), 256, 0, s, sink, iters);
}
#endif  // GPMI_PROBES
