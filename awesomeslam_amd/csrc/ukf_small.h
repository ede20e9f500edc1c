// ukf_small.h -- fused per-trajectory UKF-SLAM kernel for state dimensions that fit one CU (n <= 16*NT <= 144).
//
// One 768-thread workgroup owns one filter and runs whole callbacks of the reference's UKF node:
//     cbSensorLandmark ukf.cpp:98-110 -> updateZ ukf.cpp:113-180 (+ wait-list, growth ukf.cpp:184-257) -> slam ukf.cpp:260-392
// (the front end is shared with the EKF: small_common.h).
//
// slam() follows ukf.cpp line by line, with the 2N+5 sigma points kept implicit where they are affine:
//   * L = chol(P) by the LDS tile Cholesky (the two augmentation dimensions of Paug are diagonal, ukf.cpp:271-277);
//     a sigma point's landmark entries are X +- w L(:,c) (ukf.cpp:283-289) and pass through f unchanged
//     (common.h:49-50), so only the three pose entries of every sigma point are pushed through f (ukf.cpp:292-297);
//   * the difference matrices D (state, ukf.cpp:311-312) and DZ (measurement, ukf.cpp:346-351) are materialised once
//     in HBM/L2, row-major [n][2N+5 padded to 16];
//   * the three weighted outer-product sums P = sum w d d^T (ukf.cpp:307-319), S = sum w dz dz^T (ukf.cpp:342-357),
//     Tc = sum w d dz^T (ukf.cpp:360-375) are GEMMs A diag(w) B^T on the f64 MFMA, operand slabs staged through LDS
//     (double-buffered, register prefetch);
//   * K = Tc S^-1 (ukf.cpp:378): the central weight (1-N)/3 is negative, so S may be indefinite; its positive part
//     S+ goes through the tile Cholesky and the row-block MFMA solves shared with the EKF, the rank-1 rest through
//     Sherman-Morrison (exact); K S K^T is evaluated as Tc K^T (K S = Tc), one more GEMM (ukf.cpp:391).
// Every binary32 rounding point of the reference is kept: fp32 lambda / weights / sqrt(lambda+N+2) (ukf.h:73-81,
// ukf.cpp:284), fp32 noise variances (ukf.cpp:276-277), normalizeAngle on sigma-point headings, on the heading
// row of D and on the bearing rows of Zpred, DZ and the innovation.
#pragma once

#include "small_common.h"

namespace aslam
{
/// HBM scratch of the UKF kernels (per context)
struct UkfView
{
        int MP;     // padded sigma-point count (row stride of D / DZ), multiple of 16
        double *D;  // [B][NP][MP]  XsigPred - X (state differences)
        double *DZ; // [B][NP][MP]  Zsig, then Zsig - Zpred
        double *Tc; // [B][NP][NP]
        double *K;  // [B][NP][NP]
};

template <int NT> struct UkfLayout
{
        typedef SmallLayout<NT> LY;
        static constexpr int NP = LY::NP;
        static constexpr int MP = 2 * NP + 16; // >= 2 n + 5 rounded up to 16 for every n <= NP - 1
        static constexpr size_t oXP = (LY::total + 15) & ~(size_t)15; // double[3][MP] propagated sigma-point poses
        static constexpr size_t oW = oXP + 8 * 3 * MP;                 // double[MP + 1] weights + a spare slot for the threads beyond MP
        static constexpr size_t oXbar = oW + 8 * (MP + 2);             // double[NP] predicted mean (sW has one spare slot, padded to 16 bytes)
        static constexpr size_t oZpred = oXbar + 8 * NP;               // double[NP]
        // GEMM staging: 2 buffers x (A, B) slabs.  Where it fits it overlays the tile region (the Cholesky factor of
        // P is dead by then, and at NT = 9 there is no room for both); smaller NT get an area of their own.
        static constexpr int SLAB_LD = 17;                  // doubles per slab row (16 + 1 pad: conflict-free MFMA operand reads)
        static constexpr int SLAB = NP * SLAB_LD;           // doubles per slab
        static constexpr bool STAGE_OVERLAYS_TILES = (4 * SLAB <= LY::NTILES * TSZ);
        static constexpr size_t oZv = oZpred + 8 * NP;                 // double[NP] z = sqrt(-w0) dz_0 (rank-1 part of S)
        static constexpr size_t oVv = oZv + 8 * NP;                    // double[NP] v = S+^-1 z
        static constexpr size_t oGv = oVv + 8 * NP;                    // double[NP] g = K+ z
        static constexpr size_t oStage = oGv + 8 * NP;
        static constexpr size_t total = oStage + (STAGE_OVERLAYS_TILES ? 0 : 8 * 4 * SLAB);
};

enum
{
        GEMM_STORE = 0,    // C = A diag(w) B^T            (HBM, full matrix)
        GEMM_SUBTRACT = 1, // C -= A diag(w) B^T           (HBM, full matrix)
        GEMM_TILES = 2,    // lower tiles of A diag(w) B^T + diag_add on the true diagonal -> LDS tile storage
        GEMM_SUBTRACT_SYM = 3 // C -= A diag(w) A^T for a symmetric C: the lower tiles are computed (45 instead of 81 at NT = 9) and their new values stored
                              // into the upper triangle as well -- the (i, j) and (j, i) sums of the full product are the same MFMA chains, bit for bit
};

/// C (op)= A diag(w) B^T over k = 0 .. 16*nks-1 (SKIP_K0: without the k = 0 term).  A, B: row-major HBM with row stride ld (>= 16*nks), rows
/// [0, 16*nt).  w: LDS weights or nullptr.  All 1024 threads of the workgroup take part; `stage` is 4*SLAB doubles
/// of LDS.  Output tiles are dealt round-robin to the waves (<= TPW per wave).
template <int NT, int MODE, bool SKIP_K0 = false>
__device__ __forceinline__ void gemm_wabt(const double *A, const double *B, int ld, int nks, const double *w, int nt,
                                          double *Cg, double *Ct, double diag_add, int n_true, double *stage, int tid,
                                          const double *gvec = nullptr, double gscale = 0.0, const double *pdelta = nullptr, double pcll = 0.0,
                                          double pwsum = 0.0)
{
        typedef UkfLayout<NT> UL;
        constexpr int NP = UL::NP, LD = UL::SLAB_LD, SLAB = UL::SLAB;
        constexpr bool LOWER = (MODE == GEMM_TILES || MODE == GEMM_SUBTRACT_SYM);
        constexpr int TPW = LOWER ? (NT * (NT + 1) / 2 + SMALL_WAVES - 1) / SMALL_WAVES : (NT * NT + SMALL_WAVES - 1) / SMALL_WAVES;
        const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
        const bool same = (A == B);
        const int rows = 16 * nt;
        const int ntiles = LOWER ? nt * (nt + 1) / 2 : nt * nt;

        // my output tiles
        int tib[TPW], tjb[TPW];
        d4 acc[TPW];
#pragma unroll
        for (int q = 0; q < TPW; ++q)
        {
                const int tl = wave + q * SMALL_WAVES;
                int ib = 0, jb = 0;
                if (tl < ntiles)
                {
                        if (LOWER)
                        {
                                ib = (int)((sqrtf(8.0f * (float)tl + 1.0f) - 1.0f) * 0.5f);
                                while ((ib + 1) * (ib + 2) / 2 <= tl)
                                        ++ib;
                                while (ib * (ib + 1) / 2 > tl)
                                        --ib;
                                jb = tl - ib * (ib + 1) / 2;
                        }
                        else
                        {
                                ib = tl / nt;
                                jb = tl - ib * nt;
                        }
                }
                tib[q] = ib;
                tjb[q] = jb;
                acc[q] = (d4){0.0, 0.0, 0.0, 0.0};
        }
        // slab loader: element pairs (row, 2 consecutive k) -> one 16-byte load per pair
        const int npairs = rows * 8;
        double2 ra[2], rbv[2];
        auto fetch = [&](int ks) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
                {
                        const int pi = tid + u * SMALL_WG;
                        if (pi < npairs)
                        {
                                const int r = pi >> 3, c2 = (pi & 7) * 2;
                                ra[u] = *reinterpret_cast<const double2 *>(A + (size_t)r * ld + 16 * ks + c2);
                                if (!same)
                                        rbv[u] = *reinterpret_cast<const double2 *>(B + (size_t)r * ld + 16 * ks + c2);
                        }
                }
        };
        auto stash = [&](int buf) {
                double *sa_ = stage + (size_t)buf * 2 * SLAB, *sb_ = sa_ + SLAB;
#pragma unroll
                for (int u = 0; u < 2; ++u)
                {
                        const int pi = tid + u * SMALL_WG;
                        if (pi < npairs)
                        {
                                const int r = pi >> 3, c2 = (pi & 7) * 2;
                                sa_[r * LD + c2] = ra[u].x;
                                sa_[r * LD + c2 + 1] = ra[u].y;
                                if (!same)
                                {
                                        sb_[r * LD + c2] = rbv[u].x;
                                        sb_[r * LD + c2 + 1] = rbv[u].y;
                                }
                        }
                }
        };

        __syncthreads(); // whatever lived in the staging region before is dead from here on
        fetch(0);
        stash(0);
        __syncthreads();
        for (int ks = 0; ks < nks; ++ks)
        {
                const int buf = ks & 1;
                if (ks + 1 < nks)
                        fetch(ks + 1);
                const double *sa_ = stage + (size_t)buf * 2 * SLAB;
                const double *sb_ = same ? sa_ : sa_ + SLAB;
                double wv[4];
#pragma unroll
                for (int s = 0; s < 4; ++s)
                {
                        const int kk = 16 * ks + lg + 4 * s;
                        wv[s] = (SKIP_K0 && kk == 0) ? 0.0 : (w ? w[kk] : 1.0);
                }
#pragma unroll
                for (int q = 0; q < TPW; ++q)
                {
                        if (wave + q * SMALL_WAVES < ntiles)
                        {
                                const double *ar = sa_ + (16 * tib[q] + li) * LD + lg;
                                const double *br = sb_ + (16 * tjb[q] + li) * LD + lg;
#pragma unroll
                                for (int s = 0; s < 4; ++s)
                                        acc[q] = mfma_f64(wv[s] * ar[4 * s], br[4 * s], acc[q]); // (w d_a) d_b, ukf.cpp:315
                        }
                }
                if (ks + 1 < nks)
                        stash(buf ^ 1);
                __syncthreads();
        }

        // write out: lane l, register r of a tile = element (row (l>>4) + 4 r, column l&15)
#pragma unroll
        for (int q = 0; q < TPW; ++q)
        {
                if (wave + q * SMALL_WAVES < ntiles)
                {
                        const int ib = tib[q], jb = tjb[q];
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                        {
                                const int i = 16 * ib + lg + 4 * r, j = 16 * jb + li;
                                if (MODE == GEMM_TILES)
                                {
                                        double v = acc[q][r];
                                        if (i == j)
                                                v += (i < n_true) ? diag_add : 1.0;
                                        Ct[tile_index(ib, jb) * TSZ + (lg + 4 * r) * TLD + li] = v;
                                }
                                else if (MODE == GEMM_STORE)
                                        Cg[(size_t)i * NP + j] = acc[q][r];
                                else if (MODE == GEMM_SUBTRACT) // with an optional rank-one term gscale * g g^T (g: LDS, zero beyond the true dimension)
                                        Cg[(size_t)i * NP + j] -= gvec ? fma(gscale * gvec[i], gvec[j], acc[q][r]) : acc[q][r];
                                else // GEMM_SUBTRACT_SYM
                                {
                                        double old = Cg[(size_t)i * NP + j];
                                        // the UKF's predicted landmark block, folded into this pass (pdelta = X - Xbar): see ukf_small_kernel
                                        if (pdelta && i >= 3 && j >= 3 && i < n_true && j < n_true)
                                                old = fma(pcll, old, pwsum * pdelta[i] * pdelta[j]);
                                        const double nv = old - (gvec ? fma(gscale * gvec[i], gvec[j], acc[q][r]) : acc[q][r]);
                                        Cg[(size_t)i * NP + j] = nv;
                                        acc[q][r] = nv; // (the mirror image is stored below, tile by tile, through an LDS scratch tile)
                                }
                        }
                        if (MODE == GEMM_SUBTRACT_SYM && ib != jb)
                        {
                                // mirror image of the tile: element (i, j) -> (j, i).  Straight from the accumulators that is 8 bytes per lane into sixteen 32-byte
                                // row segments; through a scratch tile (the staging area is free behind the K loop) a lane stores four consecutive entries of a row
                                typedef double dbl2 __attribute__((ext_vector_type(2)));
                                double *scr = stage + wave * TSZ;
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                        scr[li * TLD + lg + 4 * r] = acc[q][r]; // scr[column of the tile][row of the tile]
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                                __builtin_amdgcn_wave_barrier();
                                const int orow = lane >> 2, oc = 4 * (lane & 3); // row of the MIRRORED tile (= column li of the tile), its first column
                                const double *sr = scr + orow * TLD + oc;
                                double *dst = Cg + (size_t)(16 * jb + orow) * NP + 16 * ib + oc;
                                *reinterpret_cast<dbl2 *>(dst) = (dbl2){sr[0], sr[1]};
                                *reinterpret_cast<dbl2 *>(dst + 2) = (dbl2){sr[2], sr[3]};
                                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                                __builtin_amdgcn_wave_barrier();
                        }
                }
        }
        __syncthreads();
}

/// Landmark rows of the cross covariance (round 4):  Tc(a, b) = sum_{c <= a} L(a, c) E(c, b) + delta_a s(b),  a >= 3  (rows 0 .. 2 are written by the
/// sigma-point phase and left alone).  A operand = the lower tiles of L = chol(P) where they lie in LDS (`Lt`; the diagonal tiles are masked to their
/// lower triangle), B operand = E^T (`ETg`: HBM / L2, row b = measurement, row stride NP), staged slab by slab (16 columns of E^T for all rows) in
/// `slab` -- the area of the inverted diagonal tiles, dead between the two factorisations; single-buffered: L keeps the tile region.  Both factors are
/// (block) triangular -- L(a, c) = 0 for c > a, E(c, b) = 0 for c > b + 2 -- so output tile (ib, jb) takes the slabs ks <= min(ib, jb + 1) only.
/// delta = X - Xbar, s = sum_i w_i dz_i (LDS vectors).  All threads take part; ends with a barrier.
template <int NT>
__device__ __forceinline__ void gemm_l_et(const double *Lt, const double *ETg, int nt, int n_true, const double *sX, const double *sXbar, const double *sS,
                                          double *Tcg, double *slab, int tid)
{
        typedef UkfLayout<NT> UL;
        constexpr int NP = UL::NP, LD = UL::SLAB_LD;
        constexpr int TPW = (NT * NT + SMALL_WAVES - 1) / SMALL_WAVES;
        static_assert(UL::SLAB <= NT * TSZ, "the slab of E^T takes the place of the inverted diagonal tiles");
        const int wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
        const int ntiles = nt * nt, rows = 16 * nt;
        int tib[TPW], tjb[TPW];
        d4 acc[TPW];
#pragma unroll
        for (int q = 0; q < TPW; ++q)
        {
                const int tl = wave + q * SMALL_WAVES;
                tib[q] = tl < ntiles ? tl / nt : 0;
                tjb[q] = tl < ntiles ? tl - tib[q] * nt : 0;
                acc[q] = (d4){0.0, 0.0, 0.0, 0.0};
        }
        const int npairs = rows * 8;
        // columns 16 ks .. 16 ks + 15 of E^T for all rows: element pairs, one 16-byte load per pair, fetched into registers one slab ahead (the slab
        // area is single: the loads of slab ks + 1 fly while slab ks is multiplied)
        double2 pre[2];
        auto fetch = [&](int ks) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
                {
                        const int pi = tid + u * SMALL_WG;
                        if (pi < npairs)
                                pre[u] = *reinterpret_cast<const double2 *>(ETg + (size_t)(pi >> 3) * NP + 16 * ks + (pi & 7) * 2);
                }
        };
        fetch(0);
        __syncthreads(); // whatever lived in the slab area before is dead from here on
        for (int ks = 0; ks < nt; ++ks)
        {
#pragma unroll
                for (int u = 0; u < 2; ++u)
                {
                        const int pi = tid + u * SMALL_WG;
                        if (pi < npairs)
                        {
                                const int r = pi >> 3, c2 = (pi & 7) * 2;
                                slab[r * LD + c2] = pre[u].x;
                                slab[r * LD + c2 + 1] = pre[u].y;
                        }
                }
                __syncthreads();
                if (ks + 1 < nt)
                        fetch(ks + 1);
#pragma unroll
                for (int q = 0; q < TPW; ++q)
                {
                        if (wave + q * SMALL_WAVES < ntiles && ks <= min(tib[q], tjb[q] + 1))
                        {
                                const double *ar = Lt + tile_index(tib[q], ks) * TSZ + li * TLD + lg; // L(16 ib + li, 16 ks + lg + 4 s)
                                const double *br = slab + (16 * tjb[q] + li) * LD + lg;             // E^T(16 jb + li, 16 ks + lg + 4 s)
                                const bool diag = (ks == tib[q]);
#pragma unroll
                                for (int s = 0; s < 4; ++s)
                                {
                                        const double a = (diag && lg + 4 * s > li) ? 0.0 : ar[4 * s];
                                        acc[q] = mfma_f64(a, br[4 * s], acc[q]);
                                }
                        }
                }
                __syncthreads();
        }
        // write out: lane l, register r of a tile = element (row (l >> 4) + 4 r, column l & 15)
#pragma unroll
        for (int q = 0; q < TPW; ++q)
        {
                if (wave + q * SMALL_WAVES < ntiles)
                {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                        {
                                const int i = 16 * tib[q] + lg + 4 * r, j = 16 * tjb[q] + li;
                                if (i >= 3)
                                        Tcg[(size_t)i * NP + j] = (i < n_true && j < n_true) ? fma(sX[i] - sXbar[i], sS[j], acc[q][r]) : 0.0;
                        }
                }
        }
        __syncthreads();
}

// ------------------------------------------------------------------------------------------------------
template <int NT, int MODE>
__global__ __launch_bounds__(SMALL_WG) void ukf_small_kernel(DevView d, UkfView uv, int64_t t0, int nsteps, double *poses_out,
                                                              int32_t *dims_out, StepArgs sa)
{
        typedef SmallLayout<NT> LY;
        typedef UkfLayout<NT> UL;
        constexpr int NP = LY::NP;
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
        const SmallLds L = small_carve<NT>(smem);
        double *const Lt = L.Lt, *const Dinv = L.Dinv, *const sX = L.sX, *const sZ = L.sZ, *const sY = L.sY, *const sU = L.sU;
        double *const sXP = reinterpret_cast<double *>(smem + UL::oXP);
        double *const sW = reinterpret_cast<double *>(smem + UL::oW);
        double *const sXbar = reinterpret_cast<double *>(smem + UL::oXbar);
        double *const sZpred = reinterpret_cast<double *>(smem + UL::oZpred);
        double *const sZv = reinterpret_cast<double *>(smem + UL::oZv);
        double *const sVv = reinterpret_cast<double *>(smem + UL::oVv);
        double *const sGv = reinterpret_cast<double *>(smem + UL::oGv);
        double *const stage = UL::STAGE_OVERLAYS_TILES ? Lt : reinterpret_cast<double *>(smem + UL::oStage);
        SmallShared &sm = *L.sm;

        const int tid_launch = threadIdx.x, tid = tid_launch;
        const int b = (MODE == MODE_STEP && sa.traj >= 0) ? sa.traj : (int)blockIdx.x; // sa.traj < 0: the batched step, one workgroup per filter
        const int MP = uv.MP;
        double *Pg = d.P + (size_t)b * NP * NP;
        double *Dg = uv.D + (size_t)b * NP * MP;
        double *DZg = uv.DZ + (size_t)b * NP * MP;
        double *Tcg = uv.Tc + (size_t)b * NP * NP;
        double *Kg = uv.K + (size_t)b * NP * NP;
        const double r_meas = (double)KR, q_proc = (double)KQ;

#ifdef ASLAM_STAMPS
        unsigned long long stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
#endif
        small_load<MODE>(d, L, b, tid, NP);

        for (int s = 0; s < nsteps; ++s)
        {
                // keep tid-derived addresses local to their phase instead of hoisted out of this loop and spilled (see ekf_small.h)
                int tid = tid_launch;
                asm volatile("" : "+v"(tid));
                const int64_t t = t0 + s;
                if (MODE == MODE_REPLAY)
                {
                        if (small_frontend<false, SMALL_OBS_CAP, SMALL_WAIT_CAP, NP / 2, double>(d, L, Pg, NP, b, t, s, nsteps, poses_out, dims_out, tid))
                                continue;
                }
                else
                {
                        if (tid == 0)
                        {
                                // one filter: the arguments of the call; batched step: this filter's entries of the per-call arrays
                                sm.vx = sa.traj >= 0 ? sa.vx : d.step_in[b];
                                sm.az = sa.traj >= 0 ? sa.az : d.step_in[d.B + b];
                                sm.dt = sa.traj >= 0 ? sa.dt : d.step_in[2 * d.B + b];
                        }
                        __syncthreads();
                }

                ASLAM_STAMP(0);
                // ================= slam(), ukf.cpp:260-392
                const int n = sm.n;
                const int nl = (n - 3) / 2;
                const int nt = (n + 15) >> 4;
                const int m = 2 * n + 5;
                const int mt = (m + 15) >> 4;
                const float vx = sm.vx, az = sm.az, dtf = sm.dt;
                // updateWeights, ukf.h:73-81 (binary32 lambda, weight) and w = sqrt(lambda + N + 2), ukf.cpp:284
                const float lambda_f = (float)(3.0 - (double)(n + 2));
                const float den_f = (lambda_f + (float)n) + 2.0f;
                const double w_i = (double)(float)(0.5 / (double)den_f);
                const double w_0 = (double)(lambda_f / den_f);
                const double wsp = (double)sqrtf(den_f);
                const double std_a = sqrt((double)(UKF_STD_A * UKF_STD_A)); // llt of the augmented diagonal, ukf.cpp:276,280
                {
                        // One store per thread, none of them masked: threads beyond the padded sigma-point count write a spare slot.
                        // (A thread-strided loop here leaves EXEC empty at its exit; hipcc has placed VGPR spill stores into exactly
                        // that exit block, ahead of the instruction that re-activates the lanes -- the stores then write nothing and
                        // the reloads read uninitialised scratch.  tools/check_spill_exec.py guards every build)
                        static_assert(UL::MP < SMALL_WG, "one sigma-point weight per thread");
                        const int i = min(tid, UL::MP);
                        sW[i] = (i < m) ? (i == 0 ? w_0 : w_i) : 0.0;
                }

                // ---- L = Paug.llt().matrixL(), ukf.cpp:280: lower tiles of P -> LDS -> tile Cholesky
                {
                        // a wave takes whole tiles (the row / column of a tile from its index once per tile, not per element)
                        const int ntl = nt * (nt + 1) / 2;
                        const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
                        for (int tl = wave; tl < ntl; tl += SMALL_WG / 64)
                        {
                                int ib = (int)((sqrtf(8.0f * (float)tl + 1.0f) - 1.0f) * 0.5f);
                                while ((ib + 1) * (ib + 2) / 2 <= tl)
                                        ++ib;
                                while (ib * (ib + 1) / 2 > tl)
                                        --ib;
                                const int jb = tl - ib * (ib + 1) / 2;
#pragma unroll
                                for (int q = 0; q < 4; ++q)
                                {
                                        const int e = lane + 64 * q;
                                        const int i = 16 * ib + (e >> 4), j = 16 * jb + (e & 15);
                                        double v = Pg[(size_t)i * NP + j];
                                        if (i == j && i >= n)
                                                v = 1.0;
                                        Lt[tl * TSZ + (e >> 4) * TLD + (e & 15)] = v;
                                }
                        }
                }
                __syncthreads();
                if (MODE == MODE_REPLAY && s + 1 < nsteps && __builtin_amdgcn_readfirstlane(tid_launch) >= SMALL_WG - 64) // (a scalar branch: one whole wave)
                        small_prefetch_intake<SMALL_OBS_CAP>(d, L, b, t + 1, tid_launch & 63); // (a helper role of the factorisation: the diagonal wave's first tile covers the wait)
                cholesky_lookahead<NT>(Lt, Dinv, nt, tid, &sm.status);

                ASLAM_STAMP(1);
                // L(k, c) for c <= k < n from the tile storage (0 above the diagonal)
                auto Lkc = [&](int k, int c) -> double {
                        return (c <= k) ? Lt[tile_index(k >> 4, c >> 4) * TSZ + (k & 15) * TLD + (c & 15)] : 0.0;
                };
                // sigma point i -> (column c of Laug, sign); i = 0 is the mean itself
                auto col_of = [&](int i, int &c, double &sg) {
                        if (i == 0)
                        {
                                c = -1;
                                sg = 0.0;
                        }
                        else if (i <= n + 2)
                        {
                                c = i - 1;
                                sg = 1.0;
                        }
                        else
                        {
                                c = i - n - 3;
                                sg = -1.0;
                        }
                };
                // entry k < n of augmented sigma point i: Xaug(k) +- w * L(k, c)  (ukf.cpp:287-288)
                auto xsig = [&](int k, int c, double sg) -> double {
                        if (c < 0)
                                return sX[k];
                        const double l = (c < n) ? Lkc(k, c) : 0.0;
                        return sg > 0.0 ? sX[k] + wsp * l : sX[k] - wsp * l;
                };

                // ---- propagate the pose of every sigma point, ukf.cpp:292-297
                for (int i = tid; i < 16 * mt; i += SMALL_WG)
                {
                        double p0 = 0.0, p1 = 0.0, p2 = 0.0;
                        if (i < m)
                        {
                                int c;
                                double sg;
                                col_of(i, c, sg);
                                p0 = xsig(0, c, sg);
                                p1 = xsig(1, c, sg);
                                p2 = xsig(2, c, sg);
                                // XsigAug(N, i): 0 +- w * Laug(N, N) on the acceleration-noise column, 0 elsewhere
                                double na = 0.0;
                                if (c == n)
                                        na = sg > 0.0 ? 0.0 + wsp * std_a : 0.0 - wsp * std_a;
                                stateTransition(p0, p1, p2, vx, az, dtf, true, na);
                                p2 = (double)normalizeAngle((float)p2);
                        }
                        sXP[i] = p0;
                        sXP[UL::MP + i] = p1;
                        sXP[2 * UL::MP + i] = p2;
                }
                __syncthreads();

                // ---- predicted mean, ukf.cpp:300-304.  Pose rows: the weighted sum over all sigma points, in the reference's
                // order.  Landmark rows are affine in the sigma points, the +- pairs cancel, and sum_i w_i x_i(k) = (sum_i w_i) X(k)
                // exactly (the sum of the binary32 weights is not 1: that factor is part of the reference's arithmetic).
                const double wsum = fma((double)(m - 1), w_i, w_0); // w_0 + (m - 1) w_i: the m - 1 equal weights summed in one step
                // (round 4: the three pose rows are summed by a wave each -- lane partial sums in index order, combined by DPP adds -- instead of 2 N + 5
                // sequential additions by one thread: the order of the additions differs from the reference's, the result by an ulp of the pose)
                if (tid < 3 * 64)
                {
                        const int k = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
                        double acc = 0.0;
                        for (int i = lane; i < m; i += 64)
                                acc += sW[i] * sXP[k * UL::MP + i];
                        acc = wave_sum_dpp(acc);
                        if (lane == 63)
                                sXbar[k] = acc;
                }
                for (int k = 3 + tid; k < NP; k += SMALL_WG)
                        sXbar[k] = (k < n) ? wsum * sX[k] : 0.0;
                __syncthreads();

                // ---- the few sigma points whose POSE differs from the centre's: L is lower triangular, so only the columns c < 3 of L move the pose, and the
                // acceleration-noise column c = n accelerates it (column n + 1, the yaw acceleration, does not enter f: common.h:64-73 adds az, not the noise);
                // every other sigma point carries the centre's pose through f bit for bit.  So for the pose entries d_i(k) = XsigPred_i(k) - Xbar(k):
                //   SWD(k)    = sum_i w_i d_i(k)                              (all 2 N + 5 points)
                //   EP(k, c)  = w_1 (d_{c+}(k) - d_{c-}(k)),  c < 3           (what the +- pair of column c leaves of the pose entry k)
                double *const sS = sVv, *const sSWD = sGv, *const sEP = sGv + 4; // (sVv, sGv: free until the solve)
                auto dpose = [&](int k, int i) -> double {
                        const double v = sXP[k * UL::MP + i] - sXbar[k];
                        return k == 2 ? (double)normalizeAngle((float)v) : v;
                };
                {
                        const int lane = tid & 63;
                        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
                        if (wave < 3)
                        {
                                double a = 0.0;
                                for (int i = lane; i < m; i += 64)
                                        a = fma(sW[i], dpose(wave, i), a);
                                a = wave_sum_dpp(a);
                                if (lane == 63)
                                        sSWD[wave] = a;
                        }
                        else if (wave == 3 && lane < 9)
                        {
                                const int k = lane / 3, c = lane - 3 * k;
                                sEP[lane] = w_i * (dpose(k, c + 1) - dpose(k, c + n + 3));
                        }
                }
                __syncthreads();

                ASLAM_STAMP(2);
                // ---- Zsig = h(XsigPred) (ukf.cpp:322-326, common.h:78-90), Zpred (ukf.cpp:329-339), DZ = Zsig - Zpred (ukf.cpp:346-351), Zdiff
                // (ukf.cpp:381-386), the pose rows of P (ukf.cpp:307-319) and what the cross covariance Tc (ukf.cpp:360-375) needs, one wave per measurement
                // row pair.  A sigma point moves landmark j (or the pose it is seen from) only if its column c of L is a pose column, the acceleration-
                // noise column (c = n) or c <= 4 + 2 j (L is lower triangular); every other sigma point reproduces the centre point's reading bit for
                // bit.  The wave evaluates h on the dense list of affected COLUMNS (a lane takes both signs of its column), keeps the readings in
                // registers, reduces Zpred across lanes and writes each DZ entry once.
                //
                // Round 4: the state differences D = XsigPred - Xbar are no longer materialised (350 KB per filter and callback written and read back).  For a
                // landmark entry a the sigma points are affine, d_i(a) = delta_a +- w L(a, c_i) with delta_a = X(a) - Xbar(a), so
                //     Tc(a, b)  = sum_i w_i d_i(a) dz_i(b)  =  sum_{c < n} L(a, c) E(c, b)  +  delta_a s(b)
                //     E(c, b)   = w_1 w (dz_{c+}(b) - dz_{c-}(b)),          s(b) = sum_i w_i dz_i(b)
                // -- a product over the n columns of L (in LDS, lower triangular) instead of 2 N + 5 sigma points, with E zero wherever column c does not
                // move measurement b (c > 5 + 2 j): the waves write E^T row by row next to DZ, and s; the three pose rows of Tc are
                //     Tc(k, b)  = d_0(k) s(b) + sum_{c in {0, 1, 2, n}, +-} w_1 (d_i(k) - d_0(k)) dz_i(b)
                // and the pose columns of P:  P(a, k) = w sum_{c < 3} L(a, c) EP(k, c) + delta_a SWD(k).
                {
                        const int lane = tid & 63;
                        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
                        constexpr int NWAVE = SMALL_WG / 64;
                        constexpr int DCH = (16 * NT + 2 + 63) / 64; // chunks of the dense list of columns (the centre, columns 0 .. cmax, column n)
                        double *ETg = Dg; // E^T [16 nt][NP] takes the place of D
                        for (int k = 16 * nt + tid; k < NP; k += SMALL_WG)
                        {
                                sY[k] = 0.0;
                                sZpred[k] = 0.0;
                        }
                        auto wave_sum = [](double v) -> double {
#pragma unroll
                                for (int o = 1; o < 64; o <<= 1)
                                        v += __shfl_xor(v, o);
                                return v;
                        };
                        const double d00 = dpose(0, 0), d01 = dpose(1, 0), d02 = dpose(2, 0); // the centre point's pose differences
                        const double ew = w_i * wsp;
                        for (int r = wave; r < 3 + nl + (16 * nt - n); r += NWAVE)
                        {
                                if (r < 3)
                                {
                                        // pose measurement row k: Zsig passes the pose through; 3 x 3 blocks of P and of Tc, s(k), row k of E^T
                                        const int k = r;
                                        const double zp = (k == 2) ? (double)normalizeAngle((float)sXbar[2]) : sXbar[k];
                                        double acc[3] = {0.0, 0.0, 0.0}, tcc[3] = {0.0, 0.0, 0.0}, sacc = 0.0;
                                        for (int i = lane; i < 16 * mt; i += 64)
                                        {
                                                double dv[3] = {0.0, 0.0, 0.0}, z = 0.0;
                                                if (i < m)
                                                {
                                                        dv[0] = dpose(0, i);
                                                        dv[1] = dpose(1, i);
                                                        dv[2] = dpose(2, i);
                                                        z = sXP[k * UL::MP + i] - zp;
                                                        if (k == 2)
                                                                z = (double)normalizeAngle((float)z);
                                                }
                                                const double dk = (k == 0) ? dv[0] : (k == 1) ? dv[1] : dv[2];
                                                DZg[(size_t)k * MP + i] = z;
                                                const double wd = sW[i] * dk;
                                                sacc = fma(sW[i], z, sacc);
#pragma unroll
                                                for (int a = 0; a < 3; ++a)
                                                {
                                                        acc[a] = fma(wd, dv[a], acc[a]);
                                                        tcc[a] = fma(sW[i] * dv[a], z, tcc[a]); // (w d_a) dz_k, ukf.cpp:374
                                                }
                                        }
#pragma unroll
                                        for (int a = 0; a < 3; ++a)
                                        {
                                                acc[a] = wave_sum_dpp(acc[a]);
                                                tcc[a] = wave_sum_dpp(tcc[a]);
                                        }
                                        sacc = wave_sum_dpp(sacc);
                                        // row k of E^T: only the pose columns (and column n, which no landmark row of L reaches) move a pose reading
                                        for (int c = lane; c < 16 * nt; c += 64)
                                        {
                                                double e = 0.0;
                                                if (c < n)
                                                {
                                                        double zp_ = sXP[k * UL::MP + c + 1] - zp, zm_ = sXP[k * UL::MP + c + n + 3] - zp;
                                                        if (k == 2)
                                                                zp_ = (double)normalizeAngle((float)zp_), zm_ = (double)normalizeAngle((float)zm_);
                                                        e = ew * (zp_ - zm_);
                                                }
                                                ETg[(size_t)k * NP + c] = e;
                                        }
                                        if (lane == 63)
                                        {
#pragma unroll
                                                for (int a = 0; a < 3; ++a)
                                                {
                                                        Pg[(size_t)k * NP + a] = acc[a] + ((a == k) ? q_proc : 0.0);
                                                        Tcg[(size_t)a * NP + k] = tcc[a];
                                                }
                                                double zd = sZ[k] - zp;
                                                if (k == 2)
                                                        zd = (double)normalizeAngle((float)zd);
                                                sY[k] = zd;
                                                sS[k] = sacc;
                                                sZpred[k] = zp;
                                        }
                                }
                                else if (r < 3 + nl)
                                {
                                        const int j = r - 3, ka = 3 + 2 * j, kb = 4 + 2 * j;
                                        // the pose columns of P for rows ka, kb (and their mirror image)
                                        if (lane < 6)
                                        {
                                                const int a = lane % 3, kk = (lane < 3) ? ka : kb;
                                                double v = 0.0;
#pragma unroll
                                                for (int c = 0; c < 3; ++c)
                                                        v = fma(Lkc(kk, c), sEP[3 * a + c], v);
                                                v = fma(wsp, v, (sX[kk] - sXbar[kk]) * sSWD[a]);
                                                Pg[(size_t)a * NP + kk] = v;
                                                Pg[(size_t)kk * NP + a] = v; // mirror (the reference's two roundings differ in the last bit only)
                                        }
                                        // dense list of affected columns: entry 0 is the centre point, entry 1 + e column c = e (e <= cmax) or n (e = cmax + 1), both signs
                                        const int cmax = min(4 + 2 * j, n - 1);
                                        const int npl = cmax + 2;
                                        const int nE = 1 + npl;        // list entries
                                        const int na = 1 + 2 * npl;    // affected sigma points
                                        double zrp[DCH], zbp[DCH], zrm[DCH], zbm[DCH];
                                        int cc[DCH];
                                        double sr = 0.0, sb = 0.0;
                                        auto hread = [&](int c, double sg, int i, double &zr_, double &zb_) {
                                                const double lx = xsig(ka, c, sg), ly = xsig(kb, c, sg);
                                                const double ddx = lx - sXP[i], ddy = ly - sXP[UL::MP + i];
                                                zr_ = sqrt(ddx * ddx + ddy * ddy);
                                                zb_ = atan2(ddy, ddx) - sXP[2 * UL::MP + i];
                                        };
#pragma unroll
                                        for (int q = 0; q < DCH; ++q)
                                        {
                                                const int a = lane + 64 * q;
                                                zrp[q] = zbp[q] = zrm[q] = zbm[q] = 0.0;
                                                cc[q] = -2; // no entry
                                                if (64 * q < nE && a < nE) // first test is wave-uniform: whole chunks are skipped
                                                {
                                                        if (a == 0)
                                                        {
                                                                cc[q] = -1;
                                                                hread(-1, 0.0, 0, zrp[q], zbp[q]);
                                                                sr = fma(sW[0], zrp[q], sr);
                                                                sb = fma(sW[0], zbp[q], sb);
                                                        }
                                                        else
                                                        {
                                                                const int e = a - 1, c = (e <= cmax) ? e : n;
                                                                cc[q] = c;
                                                                hread(c, 1.0, c + 1, zrp[q], zbp[q]);
                                                                hread(c, -1.0, c + n + 3, zrm[q], zbm[q]);
                                                                sr = fma(w_i, zrp[q], sr);
                                                                sb = fma(w_i, zbp[q], sb);
                                                                sr = fma(w_i, zrm[q], sr);
                                                                sb = fma(w_i, zbm[q], sb);
                                                        }
                                                }
                                        }
                                        const double z0r = readfirstlane_f64(zrp[0]), z0b = readfirstlane_f64(zbp[0]);
                                        const double wrest = w_i * (double)(m - na); // the unaffected points all carry w_i and the centre reading
                                        const double zpr = fma(wrest, z0r, wave_sum(sr));
                                        const double zpb = (double)normalizeAngle((float)fma(wrest, z0b, wave_sum(sb)));
                                        // unaffected sigma points and padding first (the affected ones are a disjoint set); E^T is zero beyond column cmax
                                        const double ur = z0r - zpr, ub = (double)normalizeAngle((float)(z0b - zpb));
                                        for (int i = lane; i < 16 * mt; i += 64)
                                        {
                                                int c;
                                                double sg;
                                                col_of(i, c, sg);
                                                const bool affected = (i == 0) || c <= cmax || c == n;
                                                if (!affected || i >= m)
                                                {
                                                        DZg[(size_t)ka * MP + i] = (i < m) ? ur : 0.0;
                                                        DZg[(size_t)kb * MP + i] = (i < m) ? ub : 0.0;
                                                }
                                        }
                                        for (int c = cmax + 1 + lane; c < 16 * nt; c += 64)
                                        {
                                                ETg[(size_t)ka * NP + c] = 0.0;
                                                ETg[(size_t)kb * NP + c] = 0.0;
                                        }
                                        // the affected points: DZ, E^T, s and the pose rows of Tc
                                        double s_r = 0.0, s_b = 0.0, tr[3] = {0.0, 0.0, 0.0}, tb[3] = {0.0, 0.0, 0.0};
#pragma unroll
                                        for (int q = 0; q < DCH; ++q)
                                        {
                                                if (cc[q] == -1)
                                                {
                                                        const double dr = zrp[q] - zpr, db = (double)normalizeAngle((float)(zbp[q] - zpb));
                                                        DZg[(size_t)ka * MP] = dr;
                                                        DZg[(size_t)kb * MP] = db;
                                                        s_r = fma(sW[0], dr, s_r);
                                                        s_b = fma(sW[0], db, s_b);
                                                }
                                                else if (cc[q] >= 0)
                                                {
                                                        const int c = cc[q], ip = c + 1, im = c + n + 3;
                                                        const double drp = zrp[q] - zpr, drm = zrm[q] - zpr;
                                                        const double dbp = (double)normalizeAngle((float)(zbp[q] - zpb)), dbm = (double)normalizeAngle((float)(zbm[q] - zpb));
                                                        DZg[(size_t)ka * MP + ip] = drp;
                                                        DZg[(size_t)ka * MP + im] = drm;
                                                        DZg[(size_t)kb * MP + ip] = dbp;
                                                        DZg[(size_t)kb * MP + im] = dbm;
                                                        if (c < n)
                                                        {
                                                                ETg[(size_t)ka * NP + c] = ew * (drp - drm);
                                                                ETg[(size_t)kb * NP + c] = ew * (dbp - dbm);
                                                        }
                                                        s_r = fma(w_i, drp, s_r);
                                                        s_r = fma(w_i, drm, s_r);
                                                        s_b = fma(w_i, dbp, s_b);
                                                        s_b = fma(w_i, dbm, s_b);
                                                        if (c < 3 || c == n) // the sigma points whose pose is not the centre's
                                                        {
                                                                const double dp[3] = {dpose(0, ip) - d00, dpose(1, ip) - d01, dpose(2, ip) - d02};
                                                                const double dm[3] = {dpose(0, im) - d00, dpose(1, im) - d01, dpose(2, im) - d02};
#pragma unroll
                                                                for (int k = 0; k < 3; ++k)
                                                                {
                                                                        tr[k] = fma(w_i * dp[k], drp, tr[k]);
                                                                        tr[k] = fma(w_i * dm[k], drm, tr[k]);
                                                                        tb[k] = fma(w_i * dp[k], dbp, tb[k]);
                                                                        tb[k] = fma(w_i * dm[k], dbm, tb[k]);
                                                                }
                                                        }
                                                }
                                        }
                                        s_r = wave_sum_dpp(s_r), s_b = wave_sum_dpp(s_b);
#pragma unroll
                                        for (int k = 0; k < 3; ++k)
                                                tr[k] = wave_sum_dpp(tr[k]), tb[k] = wave_sum_dpp(tb[k]);
                                        if (lane == 63)
                                        {
                                                // the m - na unaffected points read (ur, ub) and carry the centre's pose
                                                s_r = fma(wrest, ur, s_r);
                                                s_b = fma(wrest, ub, s_b);
                                                sS[ka] = s_r;
                                                sS[kb] = s_b;
                                                const double d0[3] = {d00, d01, d02};
#pragma unroll
                                                for (int k = 0; k < 3; ++k)
                                                {
                                                        Tcg[(size_t)k * NP + ka] = fma(d0[k], s_r, tr[k]);
                                                        Tcg[(size_t)k * NP + kb] = fma(d0[k], s_b, tb[k]);
                                                }
                                                sY[ka] = sZ[ka] - zpr;
                                                sY[kb] = (double)normalizeAngle((float)(sZ[kb] - zpb));
                                                sZpred[ka] = zpr;
                                                sZpred[kb] = zpb;
                                        }
                                }
                                else
                                {
                                        // padding rows n .. 16 nt - 1
                                        const int k = n + (r - 3 - nl);
                                        for (int i = lane; i < 16 * mt; i += 64)
                                                DZg[(size_t)k * MP + i] = 0.0;
                                        for (int c = lane; c < 16 * nt; c += 64)
                                                ETg[(size_t)k * NP + c] = 0.0;
                                        if (lane < 3)
                                                Tcg[(size_t)lane * NP + k] = 0.0; // the pose rows of Tc are written column by column by these waves: a padding column must not keep
                                                                                  // what an earlier, larger state left there (aslam_reset: tests/test_gpu_ukf.py, full-batch case)
                                        if (lane == 0)
                                        {
                                                sY[k] = 0.0;
                                                sZpred[k] = 0.0;
                                                sS[k] = 0.0;
                                        }
                                }
                        }
                }
                ASLAM_STAMP(3);
                // ---- landmark block of P.  For two landmark entries a, b the sigma points are affine,
                // d_i(a) = +-w L(a,c_i) + delta_a with delta_a = X(a) - Xbar(a) (not zero: the binary32 weights do not sum to 1),
                // so the +- pairs cancel and, exactly,  sum_i w_i d_i(a) d_i(b) = (sum_i w_i) delta_a delta_b + 2 w_1 w^2 (L L^T)(a,b),
                // with L L^T = P, the covariance that was just factored: a scale + rank-1 term instead of a GEMM -- applied (round 4) inside the pass
                // that subtracts K S K^T at the end of the callback (gemm_wabt<GEMM_SUBTRACT_SYM>, pdelta): nothing between here and there reads the
                // predicted landmark block, and a read-modify-write pass of its own over P cost 19 k cycles and 330 KB of L2 traffic per callback.
                const double cll = 2.0 * w_i * wsp * wsp;
                __syncthreads();
                ASLAM_STAMP(4);

                ASLAM_STAMP(5);
                // ---- the landmark rows of Tc = L E + delta s^T (ukf.cpp:360-375 in the form above) -> HBM; the pose rows are there already
                gemm_l_et<NT>(Lt, Dg, nt, n, sX, sXbar, sS, Tcg, Dinv, tid); // (E^T lies where D was)
                // ---- S = sum w dz dz^T + R (ukf.cpp:342-357).  The central weight w_0 = (1-N)/3 is negative, so S can be
                // indefinite (it is whenever the sigma-point headings straddle +-pi); the reference does not care because it
                // inverts S by LU (S.inverse(), ukf.cpp:378).  Here: S = S+ - z z^T with S+ = sum_{i>=1} w_i dz_i dz_i^T + R
                // (symmetric positive definite -> LDS tiles -> tile Cholesky) and z = sqrt(-w_0) dz_0, and
                //     Tc S^-1 = K+ + (K+ z) v^T / (1 - z^T v),   K+ = Tc S+^-1,  v = S+^-1 z      (Sherman-Morrison, exact).
                ASLAM_STAMP(6);
                const double zscale = sqrt(-w_0);
                for (int k = tid; k < NP; k += SMALL_WG)
                        sZv[k] = (k < n) ? zscale * DZg[(size_t)k * MP] : 0.0;
                gemm_wabt<NT, GEMM_TILES, true>(DZg, DZg, MP, mt, sW, nt, nullptr, Lt, r_meas, n, stage, tid);
                // (GEMM_TILES wrote the tiles over the staging area after its last barrier; it ends with a barrier)
                // z rides along as right-hand side row n (n is odd, so row n is always a padding row of the last tile)
                for (int j = tid; j < 16 * nt; j += SMALL_WG)
                        Tcg[(size_t)n * NP + j] = sZv[j];
                __syncthreads();
                ASLAM_STAMP(7);
                // ---- S+ = L L^T, W = Tc L^-T (and q^T = z^T L^-T in row n), t = L^-1 Zdiff, u = W t, g = W q: fused Cholesky +
                // forward substitution of the row blocks.  With S^-1 = S+^-1 + v v^T / (1 - z^T v), v = S+^-1 z = L^-T q (Sherman-Morrison):
                //   K Zdiff  = Tc S^-1 Zdiff  = W t + g (q.t) / (1 - q.q)
                //   K S K^T  = Tc S^-1 Tc^T   = W W^T + g g^T / (1 - q.q)
                // so neither K nor a backward substitution is needed (ukf.cpp:378-391 evaluated in this form).
                double *const sT = sZpred, *const sQ = sVv; // both dead by now
#ifdef ASLAM_STAMPS
                cholesky_forward_rows<NT>(Tcg, Kg, Lt, Dinv, nt, sY, sU, tid, &sm.status, sT, sQ, sGv, n, (blockIdx.x == 0 && d.dbg) ? d.dbg + 16 : nullptr);
#else
                cholesky_forward_rows<NT>(Tcg, Kg, Lt, Dinv, nt, sY, sU, tid, &sm.status, sT, sQ, sGv, n);
#endif
                ASLAM_STAMP(8);
                {
                        // q.q and q.t: the same sums in every wave
                        double qq = 0.0, qt = 0.0;
                        for (int j = (tid & 63); j < n; j += 64)
                        {
                                qq = fma(sQ[j], sQ[j], qq);
                                qt = fma(sQ[j], sT[j], qt);
                        }
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1)
                        {
                                qq += __shfl_xor(qq, o);
                                qt += __shfl_xor(qt, o);
                        }
                        const double inv_den = 1.0 / (1.0 - qq);
                        __syncthreads();
                        // the padding row goes back to zero before anything else reads Tc / W
                        for (int j = tid; j < 16 * nt; j += SMALL_WG)
                        {
                                Tcg[(size_t)n * NP + j] = 0.0;
                                Kg[(size_t)n * NP + j] = 0.0;
                        }
                        // ---- X = X + K Zdiff (ukf.cpp:389)
                        for (int k = tid; k < NP; k += SMALL_WG)
                        {
                                double dl = 0.0;
                                if (k < n)
                                {
                                        dl = sX[k] - sXbar[k]; // delta of the predicted landmark block (sZpred = sT is dead by now)
                                        sX[k] = sXbar[k] + sU[k] + sGv[k] * (qt * inv_den);
                                }
                                else
                                        sGv[k] = 0.0;
                                sZpred[k] = dl;
                        }
                        ASLAM_STAMP(9);
                        // ---- P = P - K S K^T (ukf.cpp:391) = P - W W^T - g g^T / (1 - q.q)
                        gemm_wabt<NT, GEMM_SUBTRACT_SYM>(Kg, Kg, NP, nt, nullptr, nt, Pg, nullptr, 0.0, n, stage, tid, sGv, inv_den, sZpred, cll, wsum);
                }

                ASLAM_STAMP(10);
                if (MODE == MODE_REPLAY)
                {
                        if (tid < 3 && poses_out)
                                poses_out[((size_t)b * nsteps + s) * 3 + tid] = sX[tid];
                        if (tid == 0 && dims_out)
                                dims_out[(size_t)b * nsteps + s] = n;
                }
                __syncthreads();
        }
#ifdef ASLAM_STAMPS
        if (tid == 0 && blockIdx.x == 0 && d.dbg)
                for (int i = 0; i < 12; ++i)
                        d.dbg[i] += stamp_acc[i];
#endif

        small_store<MODE>(d, L, b, tid, NP);
}
} // namespace aslam
