// small_common.h -- shared device code of the single-CU ("small", n <= 16*NT <= 144) EKF and UKF kernels:
// HBM/LDS layout, the on-device callback front end (association, wait-list, growth), tile Cholesky and the
// f64-MFMA triangular solves.  The EKF design notes below apply to ekf_small.h; ukf_small.h has its own.
//
//
// One 768-thread workgroup (12 wave64) owns one filter ("trajectory") and runs whole callbacks of the
// reference node on the device:
//     cbSensorLandmark ekf.cpp:102-114 -> updateZandA ekf.cpp:137-213 (association, wait-list, growth
//     ekf.cpp:217-290) -> slam ekf.cpp:293-311
// for `nsteps` consecutive callbacks of a recorded trace (MODE_REPLAY), or just slam() for one callback
// whose association the host did (MODE_STEP).
//
// slam() is evaluated in measurement coordinates.  With H square and R = r*I (both true in the reference:
// ekf.cpp:61,65,276,278), Pt = H P H^T, S = Pt + r I = L L^T, Kt = Pt S^-1:
//     K = P H^T S^-1 = H^-1 Kt          ->  X += H^-1 (Kt Y)
//     (I - K H) P    = H^-1 (r Kt) H^-T
// which is algebraically identical to ekf.cpp:300-310 and needs at most a Cholesky factor and two triangular
// solves with n right-hand sides (2.33 n^3 flops instead of 18 n^3); H, H^-1, A are applied as the
// <=5-non-zeros-per-row operators they are.  With R = r I and P symmetric it is cheaper still:
//     r Kt = r I - r^2 S^-1,   Kt Y = Y - r S^-1 Y          (cholesky_inverse_tiles: ~n^3 flops, no backward solve)
//
// Data placement (EKF): P is symmetric and lives in LDS for the whole launch as its lower 16x16 tiles (read from HBM,
// row-major with row stride NP = 16*NT doubles and zero padding, once at launch start and written back at its end);
// H P H^T and H^-1 (.) H^-T are one in-place block pass each on those tiles (ekf_small.h); the same tiles are then
// factored in place (S = Pt + R -> L) next to the inverted diagonal blocks; the triangular solves keep a 16-row block of
// Pt^T per wave in MFMA accumulators (v_mfma_f64_16x16x4_f64), taken from the tiles before they are factored, with L tiles
// from LDS as the A operand and the freshly solved tile, untouched, as the B operand; the lower part of r*Kt returns to
// the tiles.  (The UKF keeps P in HBM/L2 and only stages S -> L in the tiles: its LDS is taken by D / DZ slabs.)
#pragma once

#include <type_traits>

#include "device_common.h"

namespace aslam
{
constexpr int SMALL_WG = 768; // 12 wave64: 3 per SIMD -> 168 VGPRs per lane for the MFMA chains
constexpr int SMALL_WAVES = SMALL_WG / 64;
constexpr int TILE_LD = 17;            // see TLD below
constexpr int SMALL_OBS_CAP = 128;  // LDS capacity for the stored sensor message
constexpr int SMALL_WAIT_CAP = 512; // LDS capacity for new_landmark_wait

enum
{
        MODE_REPLAY = 0,
        MODE_STEP = 1
};

/// Device view of a context (all pointers are HBM).
struct DevView
{
        int B, NP, dim_cap, max_obs, max_wait;
        double *X;          // [B][NP]
        double *Z;          // [B][NP]
        double *P;          // [B][NP][NP] row-major, zero padded
        double *A;          // [B][2]  A(0,0), A(1,0)
        int *n;             // [B] state dimension N
        int *flags;         // [B] FLAG_INIT_X | FLAG_INIT_Z
        uint32_t *status;   // [B] ASLAM_ST_* bits
        float *sens;        // [B][max_obs][2] stored sensor message (range, bearing)
        int *sens_n;        // [B]
        float *wait_rb;     // [B][max_wait][2]
        uint32_t *wait_cnt; // [B][max_wait]
        int *wait_n;        // [B]
        // bound trace
        int64_t T;
        const double *tr_pose;
        const float *tr_yaw;
        const double *tr_twist;
        const float *tr_dt;
        const uint8_t *tr_new;
        const int32_t *tr_nobs;
        const float *tr_obs;
        const float *step_in;    // [3][B] vx, az, dt of a batched step (aslam_*_step_batch)
        unsigned long long *dbg; // diagnostic builds only (-DASLAM_STAMPS): per-phase cycle sums of workgroup 0
};

#ifdef ASLAM_STAMPS
#define ASLAM_STAMP(i)                                                                                                 \
        do                                                                                                             \
        {                                                                                                              \
                __syncthreads();                                                                                       \
                if (tid == 0)                                                                                          \
                {                                                                                                      \
                        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                  \
                        stamp_acc[i] += now_ - stamp_last;                                                             \
                        stamp_last = now_;                                                                             \
                }                                                                                                      \
        } while (0)
#else
#define ASLAM_STAMP(i)
#endif

#ifdef ASLAM_FE_STAMPS
#define FE_STAMP(i)                                                                                                    \
        do                                                                                                             \
        {                                                                                                              \
                __syncthreads();                                                                                       \
                if (tid == 0 && blockIdx.x == 0)                                                                       \
                {                                                                                                      \
                        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                  \
                        d.dbg[40 + (i)] += now_ - fe_last;   /* slots 45..50: clear of the kernels' phase stamps (0..11) and wave counters (16..39) */                                                                    \
                        fe_last = now_;                                                                                \
                }                                                                                                      \
        } while (0)
#else
#define FE_STAMP(i)
#endif

struct StepArgs
{
        int traj;
        float vx, az, dt;
};

/// scalars of one filter, kept in LDS while the kernel runs
struct SmallShared
{
        int n, flags, sn, wn, nnew, grew, grow_from, any_miss, any_promote, obs_new, nobs, skip;
        uint32_t status;
        float vx, az, dt, yaw;
        double px, py, tvx, twz, a00, a10;
        double pp0, pp1, pp2; // the predicted pose of this callback (EKF replay: formed by an idle wave of the front end)
        // the NEXT callback's messages, fetched while this one computes (small_prefetch_intake): valid for callback staged_t
        long long staged_t;
        double nx_px, nx_py, nx_tvx, nx_twz;
        float nx_yaw, nx_dt;
        int nx_new, nx_nobs;
};

template <int NT> struct SmallLayout
{
        static constexpr int NP = 16 * NT;
        static constexpr int NTILES = NT * (NT + 1) / 2;
        // offsets in doubles
        static constexpr int oL = 0;
        static constexpr int oDinv = oL + NTILES * 16 * TILE_LD;
        static constexpr int oX = oDinv + NT * 16 * TILE_LD;
        static constexpr int oZ = oX + NP;
        static constexpr int oY = oZ + NP;
        static constexpr int oU = oY + NP;
        static constexpr int oH = oU + NP;           // [8][NP/2], coefficient-major: h00 h01 h10 h11 (rows of H) e00 e01 e10 e11 (of H^-1) per landmark (ekf_small.h lm_coef)
        static constexpr int oTv = oH + (NP / 2) * 8; // scratch vector of the EKF's inverse-based update
        static constexpr int oEnd = oTv + NP;
        // then floats / ints
        static constexpr size_t bytes_f64 = (size_t)oEnd * 8;
        static constexpr size_t oSr = bytes_f64;                       // float[OBS_CAP] range
        static constexpr size_t oSb = oSr + 4 * SMALL_OBS_CAP;         // float bearing
        static constexpr size_t oPx = oSb + 4 * SMALL_OBS_CAP;         // float world x of the observation
        static constexpr size_t oPy = oPx + 4 * SMALL_OBS_CAP;
        static constexpr size_t oMd = oPy + 4 * SMALL_OBS_CAP;         // float nearest distance
        static constexpr size_t oCid = oMd + 4 * SMALL_OBS_CAP;        // int   nearest landmark offset (corr_id)
        static constexpr size_t oWr = oCid + 4 * SMALL_OBS_CAP;        // wait-list range
        static constexpr size_t oWb = oWr + 4 * SMALL_WAIT_CAP;
        static constexpr size_t oWx = oWb + 4 * SMALL_WAIT_CAP;        // wait-list entry re-projected from the current pose
        static constexpr size_t oWy = oWx + 4 * SMALL_WAIT_CAP;
        static constexpr size_t oWc = oWy + 4 * SMALL_WAIT_CAP;        // uint count
        static constexpr size_t oNew = oWc + 4 * SMALL_WAIT_CAP;       // int[NP/2] wait entries promoted this callback
        static constexpr size_t oPd = oNew + 4 * (NP / 2);             // float[OBS_CAP][4] partial nearest distances
        static constexpr size_t oPi = oPd + 16 * SMALL_OBS_CAP;        // int  [OBS_CAP][4] partial nearest indices
        static constexpr size_t oLm = oPi + 16 * SMALL_OBS_CAP;        // float[NP/2][2] mapped landmarks narrowed to binary32
        static constexpr size_t oSm = (oLm + 8 * (NP / 2) + 15) & ~(size_t)15;
        static constexpr size_t total = oSm + sizeof(SmallShared);
};

/// Sum of x over the 64 lanes of a wave, valid in lane 63, by DPP moves only (quad permutes, row mirrors, row broadcasts: VALU instructions).
/// __shfl_xor is a ds_bpermute -- an LDS crossbar instruction: the five binary64 sums per row of large_x_update<.., SLIM> were 60 of them per
/// wave and made the kernel LDS-bound (393 us per 256 filters against 217 for the one sum of round 2; profiles/r03_experiments.md).
__device__ __forceinline__ double wave_sum_dpp(double x)
{
        auto step = [](double v, auto ctrl, auto rmask) {
                constexpr int C = decltype(ctrl)::value, RM = decltype(rmask)::value;
                const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
                const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, C, RM, 0xf, false);
                const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), C, RM, 0xf, false);
                const double o = __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
                return v + o; // lanes outside the row mask add the 0 of `old`
        };
        using std::integral_constant;
        x = step(x, integral_constant<int, 0xB1>{}, integral_constant<int, 0xf>{});  // quad_perm [1,0,3,2]
        x = step(x, integral_constant<int, 0x4E>{}, integral_constant<int, 0xf>{});  // quad_perm [2,3,0,1]
        x = step(x, integral_constant<int, 0x141>{}, integral_constant<int, 0xf>{}); // row_half_mirror
        x = step(x, integral_constant<int, 0x140>{}, integral_constant<int, 0xf>{}); // row_mirror: every lane holds its row's sum
        x = step(x, integral_constant<int, 0x142>{}, integral_constant<int, 0xa>{}); // row_bcast:15 into rows 1, 3
        x = step(x, integral_constant<int, 0x143>{}, integral_constant<int, 0xc>{}); // row_bcast:31 into rows 2, 3: lane 63 holds the wave's sum
        return x;
}

// 16x16 tiles live in LDS with a row stride of 17 doubles: with 16 the MFMA operand pattern [l&15][k] puts the 16
// lanes of a ds_read2_b64 lane group on ONE bank pair (row stride 128 B = 32 dwords): a 16-way conflict that made
// LDS, not the MFMA pipe, the bound of the factorisation (measured; profiles/r01_phase_stamps.txt)
constexpr int TLD = 17;
constexpr int TSZ = 16 * TLD;

__device__ __forceinline__ int tile_index(int ib, int jb)
{
        return ib * (ib + 1) / 2 + jb;
}

/// element (r, c), r >= c, of a symmetric matrix held as lower 16x16 tiles
__device__ __forceinline__ double *tile_elem(double *Lt, int r, int c)
{
        return Lt + tile_index(r >> 4, c >> 4) * TSZ + (r & 15) * TLD + (c & 15);
}
__device__ __forceinline__ double sym_get(double *Lt, int r, int c)
{
        return (r >= c) ? *tile_elem(Lt, r, c) : *tile_elem(Lt, c, r);
}

// ------------------------------------------------------------------------------------------------------
/// 1/sqrt(x): v_rsq_f64 seed + one third-order Newton step (1-2 ulp; the pivot chain is the serial spine of the
/// factorisation, and sqrt + divide cost about three times as much)
__device__ __forceinline__ double rsqrt_newton(double x)
{
        double y = __builtin_amdgcn_rsq(x); // ~2^-26 relative
        const double e = fma(-(x * y), y, 1.0);
        return fma(y * e, fma(e, 0.375, 0.5), y); // third-order step: error ~ (5/16) e^3, far below 2^-53
}

/// Cholesky factor of the 16x16 diagonal tile `T` (LDS, lower triangle valid) in place, and the inverse of that factor to `Ti`;
/// returns false on a non-positive pivot.  One wave; this is the serial spine of every fused factorisation of the single-CU kernels
/// (nine tiles per callback at n = 131).  Outer-product form with the inverse built INSIDE the pivot loop: lane i (& 15; the four rows of
/// sixteen lanes mirror each other) carries row i of the tile in a[] and column i of L^-1 as running sums in s[]; once column j of L is
/// final, x_j = s[j] / L(j,j), and a[c] -= L(i,j) L(c,j), s[c] -= L(c,j) x_j use the same multiplier L(c,j).
/// Round 4: that multiplier reaches the lanes through the DPP operand of the fmac itself (row_newbcast:c -- lane c of each row of sixteen,
/// the one DPP control gfx90a+ has for 64-bit operations; every row holds the same column, so the row-local broadcast is the right one):
/// ONE instruction per update instead of two v_readlane_b32, a wait state and the fma.  The function is issue-bound, not latency-bound --
/// measured on one wave (tools/ubench/fd_bench.hip, cycles per tile): 5 140 with v_readlane (4 340 in the loop; 2 070 with the pivot
/// chain alone; halving the fmas or shortening the chain from nine to six dependent operations changed nothing) against 4 025 with DPP.
/// The pivot chain is still hand-scheduled: hipcc sinks the updates into lazy dot-product chains in front of every pivot otherwise; column
/// j's updates are issued eagerly, the one the next pivot needs first, the rest between the steps of the next pivot's rsqrt chain.
#define ASLAM_DPP_FMAC(acc, bsrc, other, c)                                                                            \
        asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:" #c " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(other))
/// a[C] -= L(C, j) lij, s[C] -= L(C, j) xj for one column C (compile-time: the DPP control is an immediate)
template <int C> __device__ __forceinline__ void factor_diag_update(double (&a)[16], double (&s)[16], const double &lij, const double &xj)
{
        if constexpr (C < 16)
        {
#define ASLAM_CASE(cc)                                                                                                 \
        if constexpr (C == cc)                                                                                         \
        {                                                                                                              \
                ASLAM_DPP_FMAC(a[cc], lij, lij, cc);                                                                   \
                ASLAM_DPP_FMAC(s[cc], lij, xj, cc);                                                                    \
        }
                ASLAM_CASE(1) ASLAM_CASE(2) ASLAM_CASE(3) ASLAM_CASE(4) ASLAM_CASE(5) ASLAM_CASE(6) ASLAM_CASE(7) ASLAM_CASE(8) ASLAM_CASE(9)
                ASLAM_CASE(10) ASLAM_CASE(11) ASLAM_CASE(12) ASLAM_CASE(13) ASLAM_CASE(14) ASLAM_CASE(15)
#undef ASLAM_CASE
        }
}
template <int J> __device__ __forceinline__ void factor_diag_column(double (&a)[16], double (&s)[16], double &inv, bool &ok)
{
        double lij = a[J] * inv; // L(i,J) for i >= J
        double xj = s[J] * inv;  // (L^-1)(J, lane)
        a[J] = lij;
        s[J] = xj;
        asm volatile("s_nop 1" : "+v"(lij), "+v"(xj)); // a VALU result read through DPP: two wait states (inline assembly is not covered by the hazard recogniser)
        double d = 1.0, y = 1.0;
        factor_diag_update<J + 1>(a, s, lij, xj);
        if constexpr (J + 1 < 16)
        {
                d = readlane_f64(a[J + 1], J + 1);
                ok = ok && (d > 0.0);
                y = __builtin_amdgcn_rsq(d);
        }
        __builtin_amdgcn_sched_barrier(0);
        const double t = d * y;
        factor_diag_update<J + 2>(a, s, lij, xj);
        factor_diag_update<J + 3>(a, s, lij, xj);
        factor_diag_update<J + 4>(a, s, lij, xj);
        __builtin_amdgcn_sched_barrier(0);
        const double e = fma(-t, y, 1.0);
        factor_diag_update<J + 5>(a, s, lij, xj);
        factor_diag_update<J + 6>(a, s, lij, xj);
        factor_diag_update<J + 7>(a, s, lij, xj);
        __builtin_amdgcn_sched_barrier(0);
        const double ye = y * e, pp = fma(e, 0.375, 0.5); // third-order step of rsqrt_newton
        factor_diag_update<J + 8>(a, s, lij, xj);
        factor_diag_update<J + 9>(a, s, lij, xj);
        factor_diag_update<J + 10>(a, s, lij, xj);
        __builtin_amdgcn_sched_barrier(0);
        const double invn = fma(ye, pp, y);
        factor_diag_update<J + 11>(a, s, lij, xj);
        factor_diag_update<J + 12>(a, s, lij, xj);
        factor_diag_update<J + 13>(a, s, lij, xj);
        factor_diag_update<J + 14>(a, s, lij, xj);
        factor_diag_update<J + 15>(a, s, lij, xj);
        __builtin_amdgcn_sched_barrier(0);
        inv = readfirstlane_f64(invn);
        if constexpr (J + 1 < 16)
                factor_diag_column<J + 1>(a, s, inv, ok);
}
/// (all 64 lanes must be active: the DPP broadcasts are per row of sixteen lanes, and lanes 16-63 mirror rows/columns 0-15)
__device__ __forceinline__ bool factor_diag_tile_fast(double *T, double *Ti, int lane)
{
        double a[16], s[16];
        const int row = lane & 15;
#pragma unroll
        for (int c = 0; c < 16; ++c)
        {
                a[c] = T[row * TLD + c];
                s[c] = (row == c) ? 1.0 : 0.0;
        }
        const double d0 = readlane_f64(a[0], 0);
        bool ok = d0 > 0.0;
        double inv = readfirstlane_f64(rsqrt_newton(d0));
        factor_diag_column<0>(a, s, inv, ok);
        if (lane < 16)
        {
#pragma unroll
                for (int c = 0; c < 16; ++c)
                {
                        T[row * TLD + c] = (c <= row) ? a[c] : 0.0;
                        Ti[c * TLD + row] = s[c]; // Linv(c, row): zero above the diagonal by construction
                }
        }
        return ok;
}
#undef ASLAM_DPP_FMAC

/// Load the 16 rows [16 rb, 16 rb + 16) of Src (row-major, stride NP) into MFMA accumulator layout, transposed:
/// acc[cb][r] of lane l = Src[16 rb + (l&15)][16 cb + (l>>4) + 4 r].
template <int NT> __device__ __forceinline__ void load_row_block(d4 (&acc)[NT], const double *Src, int rb, int nt, int lane)
{
        constexpr int NP = 16 * NT;
        const double *rowp = Src + (size_t)(16 * rb + (lane & 15)) * NP + (lane >> 4);
#pragma unroll
        for (int cb = 0; cb < NT; ++cb)
        {
                if (cb < nt)
                {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                                acc[cb][r] = rowp[16 * cb + 4 * r];
                }
        }
}

/// Physical wave -> role index of the fused factorisations (round 4).  The twelve waves of the workgroup sit three to a SIMD (wave p on SIMD p & 3); the diagonal
/// wave (role DW) is the serial spine, and every v_mfma_f64 another wave of its SIMD issues holds that SIMD's FP64 pipe for 64 cycles.  Roles DW, DW + 1, DW + 2 -- the
/// diagonal wave and two helper waves, which only take panel / trailing shares and no forward-substitution step -- go to physical waves 3, 7, 11 (one SIMD); the
/// row-block roles and the remaining helpers fill the other nine in order.
__device__ __forceinline__ int role_of_wave(int p, int DW)
{
        if ((p & 3) == 3)
                return DW + (p >> 2);
        const int idx = p - (p >> 2);
        return idx < DW ? idx : idx + 3;
}

// ---- pieces of the fused factorisation, shared by the three wave roles below -------------------------------
/// this wave's share of panel kb: L(ib,kb) = S(ib,kb) * Linv(kb)^T for ib = kb+1+wave, +SMALL_WAVES, ...
__device__ __forceinline__ void chol_panel_share(double *Lt, const double *Dinv, int nt, int kb, int wave, int li, int lg)
{
        for (int ib = kb + 1 + wave; ib < nt; ib += SMALL_WAVES)
        {
                double *S = Lt + tile_index(ib, kb) * TSZ;
                const double *Di = Dinv + kb * TSZ;
                double sa[4], sb[4];
#pragma unroll
                for (int s = 0; s < 4; ++s)
                {
                        sa[s] = S[li * TLD + lg + 4 * s];
                        sb[s] = Di[li * TLD + lg + 4 * s];
                }
                d4 t = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int s = 0; s < 4; ++s)
                        t = mfma_f64(sa[s], sb[s], t);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                        S[(lg + 4 * r) * TLD + li] = t[r];
        }
}

/// one tile of the trailing update after panel kb: S(ib,jb) -= L(ib,kb) L(jb,kb)^T, q = i (i + 1) / 2 + j over the lower triangle kb < jb <= ib
__device__ __forceinline__ void chol_trailing_tile(double *Lt, int kb, int q, int li, int lg)
{
        {
                int i = (int)((sqrtf(8.0f * (float)q + 1.0f) - 1.0f) * 0.5f);
                while ((i + 1) * (i + 2) / 2 <= q)
                        ++i;
                while (i * (i + 1) / 2 > q)
                        --i;
                const int j = q - i * (i + 1) / 2;
                const int ib = kb + 1 + i, jb = kb + 1 + j;
                double *S = Lt + tile_index(ib, jb) * TSZ;
                const double *Li = Lt + tile_index(ib, kb) * TSZ;
                const double *Lj = Lt + tile_index(jb, kb) * TSZ;
                double la[4], lb[4];
                d4 t;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                {
                        la[s] = Li[li * TLD + lg + 4 * s];
                        lb[s] = Lj[li * TLD + lg + 4 * s];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r)
                        t[r] = S[(lg + 4 * r) * TLD + li];
#pragma unroll
                for (int s = 0; s < 4; ++s)
                        t = mfma_f64(-la[s], lb[s], t);
#pragma unroll
                for (int r = 0; r < 4; ++r)
                        S[(lg + 4 * r) * TLD + li] = t[r];
        }
}

/// this wave's share of the trailing update after panel kb, without tile (kb+1,kb+1), which belongs to the diagonal wave.  The eleven other waves do not
/// come to it equally loaded (round 4): a row-block role with a forward-substitution step in this block column has `fwd_load` = nt - kb tile products of
/// its own first (per-role stamps at n = 131: role 0 busy 48.7 k cycles in the loop, role 7 20.7 k, with the tiles dealt evenly).  So the first
/// nfree * fwd_load tiles go round the `nfree` waves WITHOUT such a step (`free_slot` in [0, nfree), -1 for the others), the rest round all `nall`
/// (`all_slot`).  (Round 4 also tried leaving the two helper roles that share the diagonal wave's SIMD idle -- role_of_wave -- so that nothing else
/// issues there: no gain, 112 k -> 115 k cycles.)
__device__ __forceinline__ void chol_trailing_share(double *Lt, int nt, int kb, int nfree, int free_slot, int nall, int all_slot, int fwd_load, int li,
                                                    int lg)
{
        const int m = nt - kb - 1;
        const int T = m * (m + 1) / 2 - 1; // tiles q = 1 .. T
        const int first = min(T, nfree * fwd_load);
        if (free_slot >= 0)
                for (int q = free_slot; q < first; q += nfree)
                        chol_trailing_tile(Lt, kb, 1 + q, li, lg);
        for (int q = first + all_slot; q < T; q += nall)
                chol_trailing_tile(Lt, kb, 1 + q, li, lg);
}

/// forward substitution step for block column kb on the row block in acc: acc[kb] <- Linv(kb) acc[kb], then
/// acc[c2] -= L(c2,kb) acc[kb] for c2 > kb, two independent accumulators in flight
template <int NT>
__device__ __forceinline__ void forward_step(d4 (&acc)[NT], const double *Lt, const double *Dinv, int nt, int kb, int li, int lg)
{
        const double *Di = Dinv + kb * TSZ;
        double da[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
                da[s] = Di[li * TLD + lg + 4 * s];
        d4 v = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int cb = 0; cb < NT; ++cb)
        {
                if (cb == kb)
                {
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                                v = mfma_f64(da[s], acc[cb][s], v);
                        acc[cb] = v;
                }
        }
        const d4 vn = -v;
#pragma unroll
        for (int c2 = 1; c2 < NT; c2 += 2)
        {
                const bool on0 = (c2 > kb && c2 < nt), on1 = (c2 + 1 > kb && c2 + 1 < nt && c2 + 1 < NT);
                double la[4], lb[4];
                if (on0)
                {
                        const double *Lc = Lt + tile_index(c2, kb) * TSZ;
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                                la[s] = Lc[li * TLD + lg + 4 * s];
                }
                if (on1)
                {
                        const double *Lc = Lt + tile_index(c2 + 1, kb) * TSZ;
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                                lb[s] = Lc[li * TLD + lg + 4 * s];
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
                {
                        if (on0)
                                acc[c2] = mfma_f64(la[s], vn[s], acc[c2]);
                        if (on1)
                                acc[(c2 + 1 < NT) ? c2 + 1 : c2] = mfma_f64(lb[s], vn[s], acc[(c2 + 1 < NT) ? c2 + 1 : c2]);
                }
        }
}

/// look-ahead of the diagonal wave after panel kb: update tile (kb+1,kb+1) with L(kb+1,kb) and factor it
__device__ __forceinline__ bool chol_lookahead(double *Lt, double *Dinv, int nt, int kb, int lane, int li, int lg)
{
        if (kb + 1 >= nt)
                return true;
        double *S = Lt + tile_index(kb + 1, kb + 1) * TSZ;
        const double *Li = Lt + tile_index(kb + 1, kb) * TSZ;
        double la[4];
        d4 t;
#pragma unroll
        for (int s = 0; s < 4; ++s)
                la[s] = Li[li * TLD + lg + 4 * s];
#pragma unroll
        for (int r = 0; r < 4; ++r)
                t[r] = S[(lg + 4 * r) * TLD + li];
#pragma unroll
        for (int s = 0; s < 4; ++s)
                t = mfma_f64(-la[s], la[s], t);
#pragma unroll
        for (int r = 0; r < 4; ++r)
                S[(lg + 4 * r) * TLD + li] = t[r];
        return factor_diag_tile_fast(S, Dinv + (kb + 1) * TSZ, lane);
}

/// Forward-only tail of the fused solve (UKF): the row block in acc is W = Src L^-T.  Store it, and let the wave that owns
/// row `qrow` publish that row (q^T) to LDS.
/// `scratch`: one tile (TSZ doubles) of LDS per wave.  In the accumulator layout a lane holds the columns lg + 4 r of row li: stored directly, an instruction writes 8 bytes
/// per lane into sixteen 32-byte row segments (36 such instructions per row block: 24 k cycles behind the loop, per-role stamps of round 4); through the scratch tile a lane
/// stores four consecutive columns of a row as 32 contiguous bytes and four lanes cover a 128-byte line.
template <int NT>
__device__ __forceinline__ void forward_store(const d4 (&acc)[NT], double *Dst, int rb, int nt, int qrow, double *Q, int lane, double *scratch)
{
        constexpr int NP = 16 * NT;
        typedef double dbl2 __attribute__((ext_vector_type(2)));
        const int li = lane & 15, lg = lane >> 4;
        const int orow = lane >> 2, oc = 4 * (lane & 3); // this lane's row and first column of the tile on the way out
        double *outp = Dst + (size_t)(16 * rb + orow) * NP + oc;
        const bool mine = (16 * rb + li == qrow);
#pragma unroll
        for (int cb = 0; cb < NT; ++cb)
        {
                if (cb < nt)
                {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                        {
                                scratch[li * TLD + lg + 4 * r] = acc[cb][r];
                                if (mine)
                                        Q[16 * cb + lg + 4 * r] = acc[cb][r];
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        const double *sr = scratch + orow * TLD + oc;
                        const dbl2 v0 = {sr[0], sr[1]}, v1 = {sr[2], sr[3]};
                        *reinterpret_cast<dbl2 *>(outp + 16 * cb) = v0;
                        *reinterpret_cast<dbl2 *>(outp + 16 * cb + 2) = v1;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier(); // (the next tile overwrites the scratch)
                }
        }
}

/// U[row] = W[row] . T and G[row] = W[row] . Q for the 16 rows of the block held in acc (T, Q: LDS vectors)
template <int NT>
__device__ __forceinline__ void row_dots(const d4 (&acc)[NT], int rb, int nt, const double *T, const double *Q, double *U, double *G, int lane)
{
        const int li = lane & 15, lg = lane >> 4;
        double pu = 0.0, pg = 0.0;
#pragma unroll
        for (int cb = 0; cb < NT; ++cb)
        {
                if (cb < nt)
                {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                        {
                                pu = fma(acc[cb][r], T[16 * cb + lg + 4 * r], pu);
                                pg = fma(acc[cb][r], Q[16 * cb + lg + 4 * r], pg);
                        }
                }
        }
        pu += __shfl_xor(pu, 16);
        pu += __shfl_xor(pu, 32);
        pg += __shfl_xor(pg, 16);
        pg += __shfl_xor(pg, 32);
        if (lg == 0)
        {
                U[16 * rb + li] = pu;
                G[16 * rb + li] = pg;
        }
}

/// T = L^-1 Y for a vector (LDS), tile by tile, by ONE wave: T_kb = Linv(kb) (Y_kb - sum_{j<kb} L(kb,j) T_j)
/// Block kb alone (blocks 0 .. kb-1 of T done): what a helper wave of cholesky_forward_rows does in block column kb of the fused loop (round 4) -- row kb of L and the
/// inverse of diagonal tile kb are final by then, and the wave has nothing else to do -- so that T is complete when the loop ends; the diagonal wave used to
/// run the whole chain behind the loop (17 k cycles at n = 131) while the row-block waves waited for T.
__device__ __forceinline__ void forward_vector_block(const double *Lt, const double *Dinv, int kb, const double *Y, double *T, int lane)
{
        const int li = lane & 15, lg = lane >> 4;
        double p = 0.0;
        for (int j = 0; j < kb; ++j)
        {
                const double *Lkj = Lt + tile_index(kb, j) * TSZ;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                        p = fma(Lkj[li * TLD + lg + 4 * s], T[16 * j + lg + 4 * s], p);
        }
        p += __shfl_xor(p, 16);
        p += __shfl_xor(p, 32);
        const double r = Y[16 * kb + li] - p; // every lane of row li holds it
        const double *Di = Dinv + kb * TSZ;
        double q = 0.0;
#pragma unroll
        for (int s = 0; s < 4; ++s)
                q = fma(Di[li * TLD + lg + 4 * s], __shfl(r, lg + 4 * s), q);
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
        if (lg == 0)
                T[16 * kb + li] = q;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier(); // the next block reads T through LDS
}

__device__ __forceinline__ void forward_vector(const double *Lt, const double *Dinv, int nt, const double *Y, double *T, int lane)
{
        const int li = lane & 15, lg = lane >> 4;
        for (int kb = 0; kb < nt; ++kb)
        {
                double p = 0.0;
                for (int j = 0; j < kb; ++j)
                {
                        const double *Lkj = Lt + tile_index(kb, j) * TSZ;
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                                p = fma(Lkj[li * TLD + lg + 4 * s], T[16 * j + lg + 4 * s], p);
                }
                p += __shfl_xor(p, 16);
                p += __shfl_xor(p, 32);
                const double r = Y[16 * kb + li] - p; // every lane of row li holds it
                const double *Di = Dinv + kb * TSZ;
                double q = 0.0;
#pragma unroll
                for (int s = 0; s < 4; ++s)
                        q = fma(Di[li * TLD + lg + 4 * s], __shfl(r, lg + 4 * s), q);
                q += __shfl_xor(q, 16);
                q += __shfl_xor(q, 32);
                if (lg == 0)
                        T[16 * kb + li] = q;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier(); // the next block reads T through LDS
        }
}

/// UKF solve (ukf.cpp:268-271 through K = Tc S^-1): S = L L^T in place (S = lower tiles in Lt, which leaves as its Cholesky
/// factor) and, riding along, the FORWARD substitution only:  Dst = W = Src L^-T for all rows (Src, Dst row-major HBM with
/// stride NP, may alias), Tv = L^-1 Y, Qv = row `qrow` of W, U = W Tv, G = W Qv (Tv, Qv, G: LDS vectors of 16*NT doubles) --
/// all a symmetric update P -= W W^T + rank one needs, so no backward substitution exists.  The whole workgroup takes
/// part, in three wave roles that run their own loops with the same barrier sequence (so that each role gets its own
/// register allocation):
///   * waves 0 .. nt-1 ("row-block waves"): 16 right-hand-side rows each in MFMA accumulators; the forward substitution
///     rides along with the factorisation (step kb as soon as panel kb exists);
///   * wave NT ("diagonal wave"): factors diagonal tile kb+1 (look-ahead) while the others do the trailing update
///     and the forward step of block column kb, so the serial 16x16 factorisations leave the critical path;
///   * the remaining waves only help with panel / trailing tiles.
/// Ends with a barrier.
template <int NT>
__device__ __forceinline__ void cholesky_forward_rows(const double *Src, double *Dst, double *Lt, double *Dinv, int nt,
                                                      const double *Y, double *U, int tid, uint32_t *status, double *Tv, double *Qv,
                                                      double *G, int qrow, unsigned long long *wave_busy = nullptr)
{
#ifdef ASLAM_STAMPS
        unsigned long long tb_[3] = {0, 0, 0}, tm_ = __builtin_amdgcn_s_memtime(), tn_;
#define WB(i) (tn_ = __builtin_amdgcn_s_memtime(), tb_[i] += tn_ - tm_, tm_ = tn_)
#else
#define WB(i)
#endif
        static_assert(NT < SMALL_WAVES, "one wave beyond the row-block waves is needed for the diagonal tiles");
        constexpr int DW = NT;
        const int wave = role_of_wave(__builtin_amdgcn_readfirstlane(tid >> 6), DW), lane = tid & 63; // scalar: the role branches are uniform
        const int li = lane & 15, lg = lane >> 4;
        if (wave < nt)
        {
                d4 acc[NT];
                load_row_block<NT>(acc, Src, wave, nt, lane);
                __syncthreads(); // (the diagonal wave factors tile 0)
                for (int kb = 0; kb < nt; ++kb)
                {
                        WB(1);
                        chol_panel_share(Lt, Dinv, nt, kb, wave, li, lg);
                        WB(0);
                        __syncthreads();
                        WB(1);
                        forward_step<NT>(acc, Lt, Dinv, nt, kb, li, lg);
                        chol_trailing_share(Lt, nt, kb, SMALL_WAVES - 2 - nt, -1, SMALL_WAVES - 1, wave, nt - kb, li, lg); // (every row block has a forward step in every block column)
                        WB(0);
                        __syncthreads();
                }
                WB(1);
                forward_store<NT>(acc, Dst, wave, nt, qrow, Qv, lane, Dinv + wave * TSZ); // (the inverted diagonal tiles are dead behind the loop: a scratch tile per wave)
                __syncthreads(); // Qv and Tv (diagonal wave) are in LDS
                row_dots<NT>(acc, wave, nt, Tv, Qv, U, G, lane);
                WB(2);
        }
        else if (wave == DW)
        {
                // the diagonal wave is the critical path and shares its SIMD with two MFMA-heavy waves: let it win issue arbitration
                __builtin_amdgcn_s_setprio(3);
                bool ok = factor_diag_tile_fast(Lt, Dinv, lane);
                __syncthreads();
                for (int kb = 0; kb < nt; ++kb)
                {
                        WB(1);
                        chol_panel_share(Lt, Dinv, nt, kb, wave, li, lg);
                        WB(0);
                        __syncthreads();
                        WB(1);
                        ok = chol_lookahead(Lt, Dinv, nt, kb, lane, li, lg) && ok;
                        WB(0);
                        __syncthreads();
                }
                WB(1);
                if (!ok && lane == 0)
                        *status |= 4u; // ASLAM_ST_NOT_PD
                // (Tv = L^-1 Y: helper role DW + 1, block by block inside the loop)
                __syncthreads();
                __builtin_amdgcn_s_setprio(0);
        }
        else
        {
                __syncthreads();
                for (int kb = 0; kb < nt; ++kb)
                {
                        WB(1);
                        chol_panel_share(Lt, Dinv, nt, kb, wave, li, lg);
                        WB(0);
                        __syncthreads();
                        WB(1);
                        if (wave == DW + 1)
                                forward_vector_block(Lt, Dinv, kb, Y, Tv, lane); // row kb of L and Linv(kb) are final: block kb of Tv = L^-1 Y
                        // (role DW + 1 has the vector's block: it takes tiles only with the loaded roles; the free ones are the roles nt .. DW - 1 and DW + 2 ..)
                        chol_trailing_share(Lt, nt, kb, SMALL_WAVES - 2 - nt, wave == DW + 1 ? -1 : wave - nt - (wave > DW ? 2 : 0), SMALL_WAVES - 1,
                                            wave < DW ? wave : wave - 1, nt - kb, li, lg);
                        WB(0);
                        __syncthreads();
                }
                WB(1);
                __syncthreads();
        }
#ifdef ASLAM_STAMPS
        if (wave_busy && lane == 0)
        {
                wave_busy[2 * wave] += tb_[0];      // busy inside the factorisation loop
                wave_busy[2 * wave + 1] += tb_[2];  // store + row dots
        }
#endif
#undef WB
        __syncthreads();
}

/// EKF update in measurement coordinates when the tiles hold the symmetric P~ and R = r I (ekf.cpp:65,278):
///     r Kt = r P~ S^-1 = r (S - r I) S^-1 = r I - r^2 S^-1,        Kt Y = Y - r S^-1 Y,        S = P~ + r I = L L^T
/// so only the symmetric inverse S^-1 = L^-T L^-1 is needed: Cholesky (n^3/3), the forward substitution of the IDENTITY
/// (its row blocks are triangular: n^3/3) and one triangular symmetric product (n^3/3) -- n^3 flops instead of the 2.33 n^3
/// of factor + forward + backward on n right-hand sides, and no backward substitution at all.  The cancellation in
/// r - r^2 (S^-1)_ii costs log10(r / P~_ii) digits (2-4 here), far inside the 1e-6 bar.
/// In: tiles = P~ (lower).  Out: tiles = lower part of r Kt (zero padding), U = Kt Y.  Same three wave roles and look-ahead
/// as cholesky_forward_rows; `Tv`: LDS scratch of 16*NT doubles.  Ends with a barrier.
template <int NT>
__device__ __forceinline__ void cholesky_inverse_tiles(double *Lt, double *Dinv, int nt, int n_true, const double *Y, double *U,
                                                       double *Tv, double r, int tid, uint32_t *status, unsigned long long *wave_busy = nullptr)
{
#ifdef ASLAM_STAMPS
        // diagnostic builds: per role, shader cycles busy between the barriers of the factorisation loop (slot 0) and, for the diagonal wave, inside the
        // look-ahead (update + factorisation of the next diagonal tile: slot 1)
        unsigned long long tb_[3] = {0, 0, 0}, tm_ = __builtin_amdgcn_s_memtime(), tn_;
#define WB(i) (tn_ = __builtin_amdgcn_s_memtime(), tb_[i] += tn_ - tm_, tm_ = tn_)
#else
#define WB(i)
#endif
        static_assert(NT + 1 < SMALL_WAVES, "row-block waves, the diagonal wave and at least one helper");
        constexpr int DW = NT;
        const int wave = role_of_wave(__builtin_amdgcn_readfirstlane(tid >> 6), DW), lane = tid & 63;
        const int li = lane & 15, lg = lane >> 4;
        if (tid < 16 * nt)
                *tile_elem(Lt, tid, tid) += (tid < n_true) ? r : 1.0; // S = P~ + R; padding decouples
        __syncthreads();
        if (wave < nt)
        {
                const int rb = wave;
                // right-hand side = rows 16 rb .. of the identity, in the transposed accumulator layout
                d4 acc[NT];
#pragma unroll
                for (int cb = 0; cb < NT; ++cb)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                                acc[cb][q] = (cb == rb && li == lg + 4 * q) ? 1.0 : 0.0;
                // Round 4: Y^T rides along as right-hand side row n (n is odd, so row n is a padding row of the last row block: its identity row is not
                // needed -- the padding decouples -- and is put back below): the forward substitution turns it into t^T = (L^-1 Y)^T, the first of the two
                // triangular mat-vecs of S^-1 Y = L^-T (L^-1 Y), which one helper wave used to do tile by tile behind the loop while eleven waves waited
                // (~ 25 k of the phase's 111 k cycles at n = 131); the second is a dot product of every row-block wave's own registers with t (below).
                const int nrow = n_true - 16 * (nt - 1); // row n inside the last row block
                const bool yrow = (rb == nt - 1) && (li == nrow);
                if (rb == nt - 1)
                {
#pragma unroll
                        for (int cb = 0; cb < NT; ++cb)
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                        if (yrow && cb < nt)
                                                acc[cb][q] = Y[16 * cb + lg + 4 * q]; // (zero from n on)
                }
                __syncthreads(); // (the diagonal wave factors tile 0)
                for (int kb = 0; kb < nt; ++kb)
                {
                        WB(1);
                        chol_panel_share(Lt, Dinv, nt, kb, wave, li, lg);
                        WB(0);
                        __syncthreads();
                        WB(1);
                        if (kb >= rb || rb == nt - 1) // block columns left of the diagonal block of this row block stay zero (the last one carries Y^T: all of them)
                                forward_step<NT>(acc, Lt, Dinv, nt, kb, li, lg);
                        chol_trailing_share(Lt, nt, kb, max(0, nt - 2 - kb) + SMALL_WAVES - 1 - nt, (kb >= rb || rb == nt - 1) ? -1 : rb - kb - 1, SMALL_WAVES - 1, wave, nt - kb, li, lg);
                        WB(0);
                        __syncthreads();
                }
                WB(1);
                // row n of the last row block is t^T = (L^-1 Y)^T: publish it, and give the row its identity values back
                if (rb == nt - 1)
                {
#pragma unroll
                        for (int cb = 0; cb < NT; ++cb)
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                        if (yrow && cb < nt)
                                        {
                                                Tv[16 * cb + lg + 4 * q] = acc[cb][q];
                                                acc[cb][q] = (16 * cb + lg + 4 * q == n_true) ? 1.0 : 0.0;
                                        }
                }
                // acc = rows of L^-T; L is dead: its tiles take L^-1 (tile (cb, rb) = transpose of block (rb, cb) of L^-T)
#pragma unroll
                for (int cb = 0; cb < NT; ++cb)
                {
                        if (cb >= rb && cb < nt)
                        {
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                        Lt[tile_index(cb, rb) * TSZ + (lg + 4 * q) * TLD + li] = acc[cb][q];
                        }
                }
                __syncthreads(); // [A] L^-1 complete, t published
                // U = Kt Y = Y - r S^-1 Y = Y - r L^-T t: this wave's rows of L^-T are its accumulators
                {
                        double pu = 0.0;
#pragma unroll
                        for (int cb = 0; cb < NT; ++cb)
                        {
                                if (cb >= rb && cb < nt)
                                {
#pragma unroll
                                        for (int q = 0; q < 4; ++q)
                                                pu = fma(acc[cb][q], Tv[16 * cb + lg + 4 * q], pu);
                                }
                        }
                        pu += __shfl_xor(pu, 16);
                        pu += __shfl_xor(pu, 32);
                        if (lg == 0)
                                U[16 * rb + li] = Y[16 * rb + li] - r * pu;
                }
                WB(2); // (row-block roles: everything behind the factorisation loop -- L^-1 to the tiles, the L^-T L^-1 product, r I - r^2 S^-1)
        }
        else if (wave == DW)
        {
                __builtin_amdgcn_s_setprio(3);
                bool ok = factor_diag_tile_fast(Lt, Dinv, lane);
                __syncthreads();
                for (int kb = 0; kb < nt; ++kb)
                {
                        WB(1);
                        chol_panel_share(Lt, Dinv, nt, kb, wave, li, lg);
                        WB(0);
                        __syncthreads();
                        WB(1);
                        ok = chol_lookahead(Lt, Dinv, nt, kb, lane, li, lg) && ok;
                        WB(2);
                        __syncthreads();
                }
                WB(1);
                if (!ok && lane == 0)
                        *status |= 4u; // ASLAM_ST_NOT_PD
                __builtin_amdgcn_s_setprio(0);
                __syncthreads(); // [A]
        }
        else
        {
                __syncthreads();
                for (int kb = 0; kb < nt; ++kb)
                {
                        chol_panel_share(Lt, Dinv, nt, kb, wave, li, lg);
                        __syncthreads();
                        chol_trailing_share(Lt, nt, kb, max(0, nt - 2 - kb) + SMALL_WAVES - 1 - nt, max(0, nt - 2 - kb) + wave - nt - (wave > DW ? 1 : 0), SMALL_WAVES - 1, wave < DW ? wave : wave - 1, nt - kb, li, lg);
                        __syncthreads();
                }
                __syncthreads(); // [A]
        }
#ifdef ASLAM_STAMPS
        unsigned long long tp0_ = __builtin_amdgcn_s_memtime(), tp1_ = 0;
#endif
        // ---- S^-1 = L^-T L^-1 on ALL twelve waves (round 4: the row-block waves did it alone, row block rb its rb + 1 tiles of (nt - rb) products each --
        // up to 25 tile products on one wave while three waves idled): (S^-1)(rb, jb) = sum_{k >= rb} Linv(k, rb)^T Linv(k, jb), jb <= rb; the 45 lower tiles
        // are dealt in snake order over the waves (costs fall with the tile index: <= 15 products per wave), two products in flight per tile
        {
                const int pw = __builtin_amdgcn_readfirstlane(tid >> 6); // (any bijection wave -> 0 .. 11 will do)
                const int ntl = nt * (nt + 1) / 2;
                constexpr int TPW = (NT * (NT + 1) / 2 + SMALL_WAVES - 1) / SMALL_WAVES;
                d4 out[TPW];
                int orb[TPW], ojb[TPW];
#pragma unroll
                for (int qq = 0; qq < TPW; ++qq)
                {
                        const int tl = SMALL_WAVES * qq + ((qq & 1) ? SMALL_WAVES - 1 - pw : pw);
                        int rb = 0, jb = 0;
                        d4 o0 = {0.0, 0.0, 0.0, 0.0}, o1 = {0.0, 0.0, 0.0, 0.0};
                        if (tl < ntl)
                        {
                                rb = (int)((sqrtf(8.0f * (float)tl + 1.0f) - 1.0f) * 0.5f);
                                while ((rb + 1) * (rb + 2) / 2 <= tl)
                                        ++rb;
                                while (rb * (rb + 1) / 2 > tl)
                                        --rb;
                                jb = tl - rb * (rb + 1) / 2;
                                for (int k = rb; k < nt; k += 2)
                                {
                                        const bool two = (k + 1 < nt);
                                        const double *A0 = Lt + tile_index(k, rb) * TSZ, *B0 = Lt + tile_index(k, jb) * TSZ;
                                        const double *A1 = Lt + tile_index(two ? k + 1 : k, rb) * TSZ, *B1 = Lt + tile_index(two ? k + 1 : k, jb) * TSZ;
                                        double a0[4], b0[4], a1[4], b1[4];
#pragma unroll
                                        for (int q = 0; q < 4; ++q)
                                        {
                                                a0[q] = A0[(lg + 4 * q) * TLD + li];
                                                b0[q] = B0[(lg + 4 * q) * TLD + li];
                                                a1[q] = two ? A1[(lg + 4 * q) * TLD + li] : 0.0;
                                                b1[q] = B1[(lg + 4 * q) * TLD + li];
                                        }
#pragma unroll
                                        for (int q = 0; q < 4; ++q)
                                        {
                                                o0 = mfma_f64(a0[q], b0[q], o0);
                                                o1 = mfma_f64(a1[q], b1[q], o1);
                                        }
                                }
                        }
                        out[qq] = o0 + o1; // C layout: element (row lg + 4 q, column li) of the tile
                        orb[qq] = (tl < ntl) ? rb : -1;
                        ojb[qq] = jb;
                }
#ifdef ASLAM_STAMPS
                tp1_ = __builtin_amdgcn_s_memtime();
#endif
                __syncthreads(); // [B] nobody reads L^-1 any more
                const double r2 = r * r;
#pragma unroll
                for (int qq = 0; qq < TPW; ++qq)
                {
                        if (orb[qq] >= 0)
                        {
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                {
                                        const int i = 16 * orb[qq] + lg + 4 * q, j = 16 * ojb[qq] + li;
                                        if (j <= i)
                                        {
                                                double v = 0.0;
                                                if (i < n_true) // (j <= i < n)
                                                        v = ((i == j) ? r : 0.0) - r2 * out[qq][q];
                                                Lt[tile_index(orb[qq], ojb[qq]) * TSZ + (lg + 4 * q) * TLD + li] = v;
                                        }
                                }
                        }
                }
        }
#ifdef ASLAM_STAMPS
        if (wave_busy && lane == 0)
        {
                wave_busy[2 * wave] += (wave == DW) ? tb_[1] : tb_[0]; // busy inside the factorisation loop (diagonal wave: its WAIT at the barriers instead)
                wave_busy[2 * wave + 1] += tb_[2]; // diagonal wave: update + factorisation of the next diagonal tile
                if (wave > DW)
                {
                        wave_busy[2 * wave] += tp1_ - tp0_;                               // helper roles: the S^-1 product section ...
                        wave_busy[2 * wave + 1] += __builtin_amdgcn_s_memtime() - tp1_; // ... and what follows it (barrier [B], r I - r^2 S^-1 to the tiles)
                }
        }
#endif
#undef WB
        __syncthreads();
}

/// Cholesky factorisation of the tile matrix alone (Lt -> L, Dinv -> inverted diagonal blocks), same look-ahead
/// scheme as cholesky_forward_rows: wave NT factors diagonal tile kb+1 while the others update the trailing tiles.
template <int NT> __device__ __forceinline__ void cholesky_lookahead(double *Lt, double *Dinv, int nt, int tid, uint32_t *status)
{
        static_assert(NT < SMALL_WAVES, "one wave is reserved for the diagonal tiles");
        constexpr int DW = NT;
        const int wave = role_of_wave(__builtin_amdgcn_readfirstlane(tid >> 6), DW), lane = tid & 63;
        const int li = lane & 15, lg = lane >> 4;
        if (wave == DW)
        {
                bool ok = factor_diag_tile_fast(Lt, Dinv, lane);
                __syncthreads();
                for (int kb = 0; kb < nt; ++kb)
                {
                        chol_panel_share(Lt, Dinv, nt, kb, wave, li, lg);
                        __syncthreads();
                        ok = chol_lookahead(Lt, Dinv, nt, kb, lane, li, lg) && ok;
                        __syncthreads();
                }
                if (!ok && lane == 0)
                        *status |= 4u; // ASLAM_ST_NOT_PD
        }
        else
        {
                __syncthreads();
                for (int kb = 0; kb < nt; ++kb)
                {
                        chol_panel_share(Lt, Dinv, nt, kb, wave, li, lg);
                        __syncthreads();
                        chol_trailing_share(Lt, nt, kb, 0, -1, SMALL_WAVES - 1, wave < DW ? wave : wave - 1, 0, li, lg);
                        __syncthreads();
                }
        }
        __syncthreads();
}

// ------------------------------------------------------------------------------------------------------
/// LDS pointers of one workgroup (carved by SmallLayout)
struct SmallLds
{
        double *Lt, *Dinv, *sX, *sZ, *sY, *sU, *sH, *sTv;
        float *sSr, *sSb, *sPx, *sPy, *sMd;
        int *sCid;
        float *sWr, *sWb, *sWx, *sWy;
        uint32_t *sWc;
        int *sNew;
        float *sPd;
        int *sPi;
        float *sLm;
        SmallShared *sm;
};

template <int NT> __device__ __forceinline__ SmallLds small_carve(unsigned char *smem)
{
        typedef SmallLayout<NT> LY;
        double *lds = reinterpret_cast<double *>(smem);
        SmallLds L;
        L.Lt = lds + LY::oL;
        L.Dinv = lds + LY::oDinv;
        L.sX = lds + LY::oX;
        L.sZ = lds + LY::oZ;
        L.sY = lds + LY::oY;
        L.sU = lds + LY::oU;
        L.sH = lds + LY::oH;
        L.sTv = lds + LY::oTv;
        L.sSr = reinterpret_cast<float *>(smem + LY::oSr);
        L.sSb = reinterpret_cast<float *>(smem + LY::oSb);
        L.sPx = reinterpret_cast<float *>(smem + LY::oPx);
        L.sPy = reinterpret_cast<float *>(smem + LY::oPy);
        L.sMd = reinterpret_cast<float *>(smem + LY::oMd);
        L.sCid = reinterpret_cast<int *>(smem + LY::oCid);
        L.sWr = reinterpret_cast<float *>(smem + LY::oWr);
        L.sWb = reinterpret_cast<float *>(smem + LY::oWb);
        L.sWx = reinterpret_cast<float *>(smem + LY::oWx);
        L.sWy = reinterpret_cast<float *>(smem + LY::oWy);
        L.sWc = reinterpret_cast<uint32_t *>(smem + LY::oWc);
        L.sNew = reinterpret_cast<int *>(smem + LY::oNew);
        L.sPd = reinterpret_cast<float *>(smem + LY::oPd);
        L.sPi = reinterpret_cast<int *>(smem + LY::oPi);
        L.sLm = reinterpret_cast<float *>(smem + LY::oLm);
        L.sm = reinterpret_cast<SmallShared *>(smem + LY::oSm);
        return L;
}

/// Bring filter `b` from HBM into LDS (vectors, scalars, stored sensor message, wait-list).
template <int MODE> __device__ __forceinline__ void small_load(const DevView &d, const SmallLds &L, int b, int tid, int NP)
{
        SmallShared &sm = *L.sm;
        for (int i = tid; i < NP; i += SMALL_WG)
        {
                L.sX[i] = d.X[(size_t)b * NP + i];
                L.sZ[i] = d.Z[(size_t)b * NP + i];
                L.sY[i] = 0.0;
                L.sU[i] = 0.0;
        }
        if (tid == 0)
        {
                sm.n = d.n[b];
                sm.staged_t = -1;
                sm.flags = d.flags[b];
                sm.status = d.status[b];
                sm.sn = d.sens_n[b];
                sm.wn = d.wait_n[b];
                sm.a00 = d.A[2 * b];
                sm.a10 = d.A[2 * b + 1];
        }
        __syncthreads();
        if (MODE == MODE_REPLAY)
        {
                for (int j = tid; j < sm.sn; j += SMALL_WG)
                {
                        L.sSr[j] = d.sens[((size_t)b * d.max_obs + j) * 2];
                        L.sSb[j] = d.sens[((size_t)b * d.max_obs + j) * 2 + 1];
                }
                for (int j = tid; j < sm.wn; j += SMALL_WG)
                {
                        L.sWr[j] = d.wait_rb[((size_t)b * d.max_wait + j) * 2];
                        L.sWb[j] = d.wait_rb[((size_t)b * d.max_wait + j) * 2 + 1];
                        L.sWc[j] = d.wait_cnt[(size_t)b * d.max_wait + j];
                }
        }
        __syncthreads();
}

/// Write filter `b` back to HBM.
template <int MODE> __device__ __forceinline__ void small_store(const DevView &d, const SmallLds &L, int b, int tid, int NP)
{
        SmallShared &sm = *L.sm;
        __syncthreads();
        for (int i = tid; i < NP; i += SMALL_WG)
        {
                d.X[(size_t)b * NP + i] = L.sX[i];
                d.Z[(size_t)b * NP + i] = L.sZ[i];
        }
        if (tid == 0)
        {
                d.n[b] = sm.n;
                d.flags[b] = sm.flags;
                d.status[b] = sm.status;
                d.A[2 * b] = sm.a00;
                d.A[2 * b + 1] = sm.a10;
        }
        if (MODE == MODE_REPLAY)
        {
                if (tid == 0)
                {
                        d.sens_n[b] = sm.sn;
                        d.wait_n[b] = sm.wn;
                }
                for (int j = tid; j < sm.sn; j += SMALL_WG)
                {
                        d.sens[((size_t)b * d.max_obs + j) * 2] = L.sSr[j];
                        d.sens[((size_t)b * d.max_obs + j) * 2 + 1] = L.sSb[j];
                }
                for (int j = tid; j < sm.wn; j += SMALL_WG)
                {
                        d.wait_rb[((size_t)b * d.max_wait + j) * 2] = L.sWr[j];
                        d.wait_rb[((size_t)b * d.max_wait + j) * 2 + 1] = L.sWb[j];
                        d.wait_cnt[(size_t)b * d.max_wait + j] = L.sWc[j];
                }
        }
}

/// The messages of callback t1 (= the next one of this launch) into LDS, by ONE wave with nothing else to do at the call site, while the others compute:
/// the front end of t1 then starts from LDS instead of two dependent global round trips (~ 2 k cycles per callback; round 4).  The staging area is the
/// association's partial-result scratch, unused since the scan combines in registers (entries 0 .. 15 stay the wait-list walk's).  Call between two
/// barriers of the callback, outside the front end; `lane` = 0 .. 63.
template <int OBS_CAP> __device__ __forceinline__ void small_prefetch_intake(const DevView &d, const SmallLds &L, int b, int64_t t1, int lane)
{
        SmallShared &sm = *L.sm;
        const int ocap = min(d.max_obs, OBS_CAP); // (OBS_CAP <= 4 OBS_CAP - 16: the staging area holds a whole message)
        float *const stg_r = L.sPd + 16, *const stg_b = reinterpret_cast<float *>(L.sPi) + 16;
        const float *src = d.tr_obs + ((size_t)b * d.T + t1) * d.max_obs * 2;
        for (int j0 = 0; j0 < ocap; j0 += 128)
        {
                const int ja = j0 + lane, jb = j0 + 64 + lane;
                float ra = 0.f, ba = 0.f, rb2 = 0.f, bb2 = 0.f;
                if (ja < ocap)
                        ra = src[2 * ja], ba = src[2 * ja + 1];
                if (jb < ocap)
                        rb2 = src[2 * jb], bb2 = src[2 * jb + 1];
                if (ja < ocap)
                        stg_r[ja] = ra, stg_b[ja] = ba;
                if (jb < ocap)
                        stg_r[jb] = rb2, stg_b[jb] = bb2;
        }
        if (lane == 0)
        {
                const size_t o = (size_t)b * d.T + t1;
                const double px = d.tr_pose[2 * o], py = d.tr_pose[2 * o + 1], tvx = d.tr_twist[2 * o], twz = d.tr_twist[2 * o + 1];
                const float yaw = d.tr_yaw[o], dt = d.tr_dt[o];
                const int nw = d.tr_new[o], k = d.tr_nobs[o];
                sm.nx_px = px, sm.nx_py = py, sm.nx_tvx = tvx, sm.nx_twz = twz, sm.nx_yaw = yaw, sm.nx_dt = dt, sm.nx_new = nw, sm.nx_nobs = k;
                sm.staged_t = t1;
        }
}

/// One callback's host-side work on the device: cbSensorLandmark (ekf.cpp:102-114 / ukf.cpp:98-110) when a
/// sensor message precedes this odom message, then updateZandA (ekf.cpp:137-213) / updateZ (ukf.cpp:113-180):
/// association, wait-list (ekf.cpp:217-253), promotion and growth (ekf.cpp:255-290), A (EKF), init_x.
/// Returns true when cbOdom returned early (no sensor message yet): the caller skips slam().
/// On return sm.vx / sm.az / sm.dt hold slam()'s binary32 arguments.
template <bool IS_EKF, int OBS_CAP, int WAIT_CAP, int NEW_CAP, typename TP>
__device__ __forceinline__ bool small_frontend(const DevView &d, const SmallLds &L, TP *Pg, int NP, int b, int64_t t, int s,
                                               int nsteps, double *poses_out, int32_t *dims_out, int tid, double *tilesP = nullptr)
{
        SmallShared &sm = *L.sm;
        double *const sX = L.sX, *const sZ = L.sZ;
        float *const sSr = L.sSr, *const sSb = L.sSb, *const sPx = L.sPx, *const sPy = L.sPy, *const sMd = L.sMd;
        int *const sCid = L.sCid, *const sNew = L.sNew;
        float *const sWr = L.sWr, *const sWb = L.sWb, *const sWx = L.sWx, *const sWy = L.sWy;
        uint32_t *const sWc = L.sWc;
#ifdef ASLAM_FE_STAMPS
        unsigned long long fe_last = __builtin_amdgcn_s_memtime();
#endif
        // ================= message intake
        const int ocap = min(d.max_obs, OBS_CAP);
        float *const stg_r = L.sPd + 16, *const stg_b = reinterpret_cast<float *>(L.sPi) + 16; // (the scratch of the wait-list walk is entries 0 .. 11)
        const bool staged = (sm.staged_t == t); // published by the barriers of the previous callback; uniform
        // Not staged (the first callback of a launch, the large-state front end): the observation row of this callback is fetched next to thread 0's
        // scalars (one global latency instead of two; the row has max_obs entries whatever tr_nobs says, and it is only kept when the message is new).
        const float *src = d.tr_obs + ((size_t)b * d.T + t) * d.max_obs * 2;
        float ob0r = 0.0f, ob0b = 0.0f;
        if (!staged && tid < ocap)
        {
                ob0r = src[2 * tid];
                ob0b = src[2 * tid + 1];
        }
        if (tid == 0)
        {
                const size_t o = (size_t)b * d.T + t;
                int k;
                if (staged)
                {
                        sm.px = sm.nx_px, sm.py = sm.nx_py, sm.yaw = sm.nx_yaw, sm.tvx = sm.nx_tvx, sm.twz = sm.nx_twz, sm.dt = sm.nx_dt;
                        sm.obs_new = sm.nx_new;
                        k = sm.nx_nobs;
                }
                else
                {
                        sm.px = d.tr_pose[2 * o];
                        sm.py = d.tr_pose[2 * o + 1];
                        sm.yaw = d.tr_yaw[o];
                        sm.tvx = d.tr_twist[2 * o];
                        sm.twz = d.tr_twist[2 * o + 1];
                        sm.dt = d.tr_dt[o];
                        sm.obs_new = d.tr_new[o];
                        k = d.tr_nobs[o];
                }
                if (k > d.max_obs || k > OBS_CAP)
                {
                        sm.status |= 8u; // ASLAM_ST_OBS_OVERFLOW
                        k = ocap;
                }
                sm.nobs = k;
                sm.any_miss = 0;
                sm.grew = 0;
                sm.nnew = 0;
        }
        __syncthreads();
        if (sm.obs_new)
        {
                // cbSensorLandmark, ekf.cpp:102-114
                if (staged)
                {
                        for (int j = tid; j < sm.nobs; j += SMALL_WG)
                        {
                                sSr[j] = stg_r[j];
                                sSb[j] = stg_b[j];
                        }
                }
                else
                {
                        if (tid < sm.nobs)
                        {
                                sSr[tid] = ob0r;
                                sSb[tid] = ob0b;
                        }
                        for (int j = tid + SMALL_WG; j < sm.nobs; j += SMALL_WG)
                        {
                                sSr[j] = src[2 * j];
                                sSb[j] = src[2 * j + 1];
                        }
                }
                if (tid == 0)
                {
                        sm.sn = sm.nobs;
                        sm.flags &= ~FLAG_INIT_Z;
                }
        }
        __syncthreads();
        if (sm.flags & FLAG_INIT_Z)
        {
                // cbOdom returns before anything happens, ekf.cpp:76-77
                if (tid < 3 && poses_out)
                        poses_out[((size_t)b * nsteps + s) * 3 + tid] = 0.0;
                if (tid == 0 && dims_out)
                        dims_out[(size_t)b * nsteps + s] = sm.n;
                return true;
        }
        FE_STAMP(5); // intake + sensor copy
        // ================= updateZandA, ekf.cpp:137-213
        if (tid == 0)
        {
                sZ[0] = sm.px;
                sZ[1] = sm.py;
                sZ[2] = (double)sm.yaw;
        }
        __syncthreads();
        const int n0 = sm.n;
        const int nl = (n0 - 3) / 2;
        for (int j = tid; j < sm.sn; j += SMALL_WG)
        {
                const float bb = normalizeAngle(sSb[j]);
                sSb[j] = bb;
                float a, c;
                toPoint(sSr[j], bb, sZ[0], sZ[1], sZ[2], a, c);
                sPx[j] = a;
                sPy[j] = c;
        }
        for (int k = tid; k < nl; k += SMALL_WG)
        {
                L.sLm[2 * k] = (float)sX[3 + 2 * k]; // Point(const float &, const float &), structures.h:50
                L.sLm[2 * k + 1] = (float)sX[4 + 2 * k];
                sNew[k] = -1; // last observation associated with landmark k
        }
        // Update A, ekf.cpp:206-212 (the UKF node has no A), and slam()'s binary32 arguments: they depend on the odom message alone, so the last two waves
        // form them here -- one the cosines, one the sines, the two angles on two lanes side by side -- instead of thread 0 one libm call after the other
        // behind the wait-list walk (round 4: ~ 3 k cycles of a callback's 150 k at n = 131)
        // (whole waves: scalar branches on the wave index -- no EXEC-masked region around the libm calls for the register allocator to spill into)
        const int fe_wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
        if (fe_wave >= SMALL_WG / 64 - 2)
        {
                const int l = tid & 63;
                const bool sines = fe_wave == SMALL_WG / 64 - 2; // (wave-uniform)
                if (IS_EKF && sm.tvx != 0.0 && sm.twz != 0.0)
                {
                        const float delta_theta = (float)(sm.twz * (double)sm.dt);
                        const float rr = (float)(sm.tvx / sm.twz);
                        const double ang = (l & 1) ? sZ[2] + (double)delta_theta : sZ[2];
                        const double v = sines ? sin(ang) : cos(ang);
                        const double v1 = __shfl_down(v, 1);
                        if (l == 0)
                                (sines ? sm.a10 : sm.a00) = (double)rr * (-v + v1);
                }
                if (l == 0 && !sines)
                {
                        sm.vx = (float)sm.tvx; // slam(const float &vx, ...), ekf.cpp:94,293
                        sm.az = (float)sm.twz;
                }
        }
        // The predicted pose (stateTransitionFunction, common.h:46-75, ekf.cpp:296) depends on the odom message and on the pose the last callback left
        // (param.X = param.Z on the first one, ekf.cpp:87-91): a lane of a third idle wave forms it here instead of thread 0 alone in front of the H
        // coefficients (round 4: ~ 1.5 k cycles and a barrier per callback of the single-CU EKF).
        if (IS_EKF && fe_wave == SMALL_WG / 64 - 3)
        {
                // stateTransition (device_common.h) with its two angles on two lanes: the same calls on the same arguments, side by side
                const int l = tid & 63;
                const bool ix = (sm.flags & FLAG_INIT_X) != 0;
                double p0 = ix ? sZ[0] : sX[0], p1 = ix ? sZ[1] : sX[1], p2 = ix ? sZ[2] : sX[2];
                const float vx = (float)sm.tvx, az = (float)sm.twz, dt = sm.dt;
                const bool arc = fabsf(az) > 0.001;
                const double th = p2, th2 = th + (double)(az * dt);
                const double ang = ((l & 1) && arc) ? th2 : th;
                const double sn = sin(ang), cs = cos(ang);
                const double sn2 = __shfl_down(sn, 1), cs2 = __shfl_down(cs, 1);
                if (l == 0)
                {
                        if (arc)
                        {
                                const float r = vx / az;
                                p0 += (double)r * (-sn + sn2);
                                p1 += (double)r * (cs - cs2);
                        }
                        else
                        {
                                const float vdt = vx * dt;
                                p0 += (double)vdt * cs;
                                p1 += (double)vdt * sn;
                        }
                        p2 += (double)(az * dt);
                        sm.pp0 = p0;
                        sm.pp1 = p1;
                        sm.pp2 = (double)normalizeAngle((float)p2);
                }
        }
        __syncthreads();
        FE_STAMP(6); // toPoint + landmark narrowing + A
        // nearest mapped landmark of every observation (ekf.cpp:159-173).  Eight lanes per observation scan the landmarks
        // k = c, c+8, c+16, ... each (adjacent LDS words across the eight: no bank conflicts), in index order with the
        // reference's strict `dist < mindist`.  sqrtf is correctly rounded, hence monotonic: a candidate whose squared
        // distance is not below the squared distance of the current best cannot have a smaller distance, so the root is only
        // taken for the few that pass that test -- the comparison that decides is still the reference's, on the rooted values.
        // The eight partial results are then combined by three exchanges inside the group (smallest distance, smallest index among
        // equals = first in index order; lane 0 of the group holds landmark 0's starting value and keeps it unless something is
        // strictly smaller -- also when it is an infinity or a NaN), so ties, infinities and NaNs resolve exactly as in the sequential scan.
        // (Round 4: four threads per observation and a pass through LDS before: 8.3 k cycles at n = 131.)
        constexpr int SCAN_LANES = (OBS_CAP > 128) ? 4 : 8; // (the 512-landmark front end has more observations than lanes either way: fewer, longer scans and two exchanges)
        for (int idx = tid; idx < SCAN_LANES * sm.sn; idx += SMALL_WG)
        {
                const int j = idx / SCAN_LANES, c = idx % SCAN_LANES;
                const float ox = sPx[j], oy = sPy[j];
                float bd = __builtin_inff(), bd2 = __builtin_inff();
                int bk = -1;
                const float2 *lm = reinterpret_cast<const float2 *>(L.sLm);
                // eulerDistance, tools.h:53-59.  The reference subtracts in double and narrows; for two binary32 inputs that is the
                // binary32 difference bit for bit (the double difference is exact unless the exponents are more than 28 apart, and
                // then both round to the larger operand), without six slow f64 instructions
                int k = c; // (every lane of a group takes the same number of chunks: landmark 0, lane 0's starting value, is skipped by the predicate below)
                if (c == 0 && nl > 0) // mindist starts as the distance to landmark 0, whatever it is
                {
                        const float2 p = lm[0];
                        const float dx = ox - p.x, dy = oy - p.y;
                        bd2 = dx * dx + dy * dy;
                        bd = sqrtf(bd2);
                        bk = 0;
                }
                constexpr int CH = 8; // candidates fetched together: the LDS reads of a chunk are in flight at once (one at a time: ~ 250 cycles per candidate)
                auto candidate = [&](const float2 p, int kk, bool live) {
                        const float dx = ox - p.x, dy = oy - p.y;
                        const float d2 = dx * dx + dy * dy;
                        if (live && d2 < bd2)
                        {
                                const float dd = sqrtf(d2);
                                if (dd < bd)
                                {
                                        bd = dd;
                                        bd2 = d2;
                                        bk = kk;
                                }
                        }
                };
                for (; k + (CH - 1) * SCAN_LANES < nl; k += SCAN_LANES * CH) // whole chunks
                {
                        float2 p[CH];
#pragma unroll
                        for (int u = 0; u < CH; ++u)
                                p[u] = lm[k + u * SCAN_LANES];
#pragma unroll
                        for (int u = 0; u < CH; ++u)
                                candidate(p[u], k + u * SCAN_LANES, k + u * SCAN_LANES > 0);
                }
                if (k < nl) // the last, partial one
                {
                        float2 p[CH];
#pragma unroll
                        for (int u = 0; u < CH; ++u)
                                p[u] = lm[min(k + u * SCAN_LANES, nl - 1)];
#pragma unroll
                        for (int u = 0; u < CH; ++u)
                                candidate(p[u], k + u * SCAN_LANES, k + u * SCAN_LANES < nl && k + u * SCAN_LANES > 0);
                }
#pragma unroll
                for (int off = 1; off < SCAN_LANES; off <<= 1)
                {
                        const float od = __shfl_xor(bd, off);
                        const int ok = __shfl_xor(bk, off);
                        if (ok >= 0 && (od < bd || (od == bd && ok < bk)))
                        {
                                bd = od;
                                bk = ok;
                        }
                }
                if (c == 0)
                {
                        if (nl == 0)
                        {
                                bd = __builtin_inff();
                                bk = 0;
                        }
                        sMd[j] = bd;
                        sCid[j] = 2 * bk;
                        if (n0 != 3 && bd < MIN_DIST_THRESH)
                                atomicMax(&sNew[bk], j); // observations are walked in order: the last one wins (ekf.cpp:175-181)
                        else
                                sm.any_miss = 1;
                }
        }
        FE_STAMP(7); // scan + combine
        __syncthreads();
        // associated observations: Z(3 + corr_id) = range, Z(4 + corr_id) = bearing
        for (int j = tid; j < sm.sn; j += SMALL_WG)
        {
                if (n0 != 3 && sMd[j] < MIN_DIST_THRESH && sNew[sCid[j] >> 1] == j)
                {
                        sZ[3 + sCid[j]] = (double)sSr[j];
                        sZ[4 + sCid[j]] = (double)sSb[j];
                }
        }
        FE_STAMP(8); // combine + Z
        if (sm.any_miss)
        {
                // wait-list entries re-projected from the current pose (ekf.cpp:229,233)
                for (int i = tid; i < sm.wn; i += SMALL_WG)
                {
                        float a, c;
                        toPoint(sWr[i], sWb[i], sZ[0], sZ[1], sZ[2], a, c);
                        sWx[i] = a;
                        sWy[i] = c;
                }
                __syncthreads();
        }
        // (without a miss -- the steady state of a mapped world -- nothing below reads what the loops above wrote before the barrier that ends the
        // front end: the two barriers of the wait-list walk are taken only with it; any_miss was published by the barrier behind the scan)
        if (sm.any_miss)
        {
                // Unassociated observations go through the wait-list in message order (they touch nothing the associated ones touch); counts only
                // change here, so without a miss there is nothing to promote either.  The observations are sequential -- a push changes the list
                // the next one searches -- but the nearest-neighbour scan of ONE observation (updateNewLandmarkWait, ekf.cpp:217-253: strict
                // `dist < mindist` over the entries in order, starting from entry 0) is spread over the workgroup: every thread scans its entries
                // in ascending order with the same strict compare, the partial results are combined by (distance, index), and entry 0 stays the
                // starting value (kept on ties and when its distance is a NaN), so the winner is the sequential scan's bit for bit.  One thread
                // did all of it before round 3: with a 2048-entry list a single miss cost ~30 k cycles, and the slowest filter of a launch set
                // the front end's 330 - 420 us (profiles/r03_experiments.md).
                const int wcap = min(d.max_wait, WAIT_CAP);
                const int lane = tid & 63, wv = tid >> 6;
                constexpr int NWAVE = SMALL_WG / 64;
                for (int j = 0; j < sm.sn; ++j) // (workgroup-uniform: every condition below reads LDS values published by a barrier)
                {
                        if (n0 != 3 && sMd[j] < MIN_DIST_THRESH)
                                continue;
                        const int wn = sm.wn;
                        if (wn > 0)
                        {
                                const float px = sPx[j], py = sPy[j];
                                float bd = __builtin_inff();
                                int bi = 0x7fffffff;
                                for (int i = 1 + tid; i < wn; i += SMALL_WG)
                                {
                                        const float dd = eulerDistance(px, py, sWx[i], sWy[i]);
                                        if (dd < bd)
                                        {
                                                bd = dd;
                                                bi = i;
                                        }
                                }
#pragma unroll
                                for (int off = 32; off >= 1; off >>= 1)
                                {
                                        const float od = __shfl_xor(bd, off);
                                        const int oi = __shfl_xor(bi, off);
                                        if (od < bd || (od == bd && oi < bi))
                                        {
                                                bd = od;
                                                bi = oi;
                                        }
                                }
                                if (lane == 0)
                                {
                                        L.sPd[wv] = bd; // (the association scratch is free here)
                                        L.sPi[wv] = bi;
                                }
                        }
                        __syncthreads();
                        if (tid == 0)
                        {
                                bool push = (wn == 0);
                                if (!push)
                                {
                                        int corr = 0;
                                        float mind = eulerDistance(sPx[j], sPy[j], sWx[0], sWy[0]);
                                        float bd = L.sPd[0];
                                        int bi = L.sPi[0];
                                        for (int w = 1; w < NWAVE; ++w)
                                        {
                                                const float od = L.sPd[w];
                                                const int oi = L.sPi[w];
                                                if (od < bd || (od == bd && oi < bi))
                                                {
                                                        bd = od;
                                                        bi = oi;
                                                }
                                        }
                                        if (bd < mind)
                                        {
                                                corr = bi;
                                                mind = bd;
                                        }
                                        if (mind < MIN_DIST_THRESH)
                                                sWc[corr]++;
                                        else
                                                push = true;
                                }
                                if (push)
                                {
                                        if (wn < wcap)
                                        {
                                                sWr[wn] = sSr[j];
                                                sWb[wn] = sSb[j];
                                                sWx[wn] = sPx[j];
                                                sWy[wn] = sPy[j];
                                                sWc[wn] = 1;
                                                sm.wn = wn + 1;
                                        }
                                        else
                                                sm.status |= 2u; // ASLAM_ST_WAIT_OVERFLOW
                                }
                        }
                        __syncthreads();
                }
                // promotion, ekf.cpp:187-195: rare -- every thread looks at its entries, one walks the list in order only if there is one
                if (tid == 0)
                        sm.any_promote = 0;
                __syncthreads();
                {
                        bool mine = false;
                        for (int i = tid; i < sm.wn; i += SMALL_WG)
                                mine = mine || (sWc[i] == MIN_LANDMARK_OCC);
                        if (mine)
                                sm.any_promote = 1;
                }
                __syncthreads();
                if (tid == 0 && sm.any_promote)
                {
                        const int wn = sm.wn;
                        int nnew = 0;
                        for (int i = 0; i < wn; ++i)
                        {
                                if (sWc[i] == MIN_LANDMARK_OCC)
                                {
                                        if (nnew < NEW_CAP)
                                                sNew[nnew] = i;
                                        ++nnew;
                                        sWc[i] += 1;
                                }
                        }
                        if (nnew)
                        {
                                // updateNewLandmark, ekf.cpp:255-290
                                const int nn = n0 + 2 * nnew;
                                if (nn >= d.dim_cap)
                                        sm.status |= 1u; // ASLAM_ST_GROWTH_REFUSED
                                else
                                {
                                        for (int k = 0; k < nnew; ++k)
                                        {
                                                const int e = sNew[k];
                                                const double zr = (double)sWr[e], zb = (double)sWb[e];
                                                sZ[n0 + 2 * k] = zr;
                                                sZ[n0 + 2 * k + 1] = zb;
                                                sX[n0 + 2 * k] = sZ[0] + zr * cos(sZ[2] + zb);
                                                sX[n0 + 2 * k + 1] = sZ[1] + zr * sin(sZ[2] + zb);
                                        }
                                        sm.grow_from = n0;
                                        sm.n = nn;
                                        sm.grew = 1;
                                }
                        }
                }
        }
        FE_STAMP(9); // wait-list walk
        if (sm.any_miss)
                __syncthreads(); // (grew, n, the new entries of X / Z: thread 0's)
        FE_STAMP(10);
        if (sm.grew)
        {
                // conservativeResizeLike(Identity * UKF_KP_LANDMARK_POSE), ekf.cpp:277
                const int g0 = sm.grow_from, n1 = sm.n;
                for (int idx = tid; idx < n1 * n1; idx += SMALL_WG)
                {
                        const int i = idx / n1, j = idx - i * n1;
                        if (tilesP)
                        {
                                // P lives in LDS as lower tiles (single-CU EKF): new rows of the lower triangle
                                if (i >= g0 && j <= i)
                                        *tile_elem(tilesP, i, j) = (i == j) ? (double)KP_LANDMARK_POSE : 0.0;
                        }
                        else if (i >= g0 || j >= g0)
                                Pg[(size_t)i * NP + j] = (i == j) ? (TP)KP_LANDMARK_POSE : (TP)0;
                }
        }
        if (sm.flags & FLAG_INIT_X)
        {
                // param.X = param.Z once, ekf.cpp:87-91
                __syncthreads(); // (Z is complete: the loops above may have run without a barrier behind them)
                for (int i = tid; i < sm.n; i += SMALL_WG)
                        sX[i] = sZ[i];
                __syncthreads();
                if (tid == 0)
                        sm.flags &= ~FLAG_INIT_X;
        }
        __syncthreads();
        return false;
}
} // namespace aslam
