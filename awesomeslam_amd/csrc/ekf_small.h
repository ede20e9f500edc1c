// ekf_small.h -- fused per-trajectory EKF-SLAM kernel (see small_common.h for the design notes and shared pieces).
#pragma once

#include "small_common.h"

namespace aslam
{
/// The two rows of T_i applied to a column (m0, m1, m2, ma, mb) of values indexed by (p0, p1, p2, l_a, l_b):
///   FWD:  T_i = rows of H (updateH, ekf.cpp:117-134):   [h0 h1 0 -h0 -h1], [h2 h3 -1 -h2 -h3]       (c = hc[0..3])
///   else: T_i = rows of H^-1:                            [1 0 -g1 -g0 -g1], [0 1 -g3 -g2 -g3]         (c = hc[4..7])
struct LmCoef
{
        double c[4];
};
/// coefficients of landmark i: coefficient-major in LDS (sH[k][i], k = 0..3 the rows of H, 4..7 of H^-1), so that lanes running over landmarks read
/// consecutive words (round 4; landmark-major before: a 64-byte stride, sixteen lanes on a bank)
template <bool FWD> __device__ __forceinline__ LmCoef lm_coef(const double *sH, int nlm, int i)
{
        LmCoef r;
#pragma unroll
        for (int k = 0; k < 4; ++k)
                r.c[k] = sH[((FWD ? 0 : 4) + k) * nlm + i];
        return r;
}
template <bool FWD>
__device__ __forceinline__ void lm_rows(const LmCoef &cc, double m0, double m1, double m2, double ma, double mb, double &w0, double &w1)
{
        const double *c = cc.c;
        if (FWD)
        {
                const double d0 = m0 - ma, d1 = m1 - mb;
                w0 = fma(c[1], d1, c[0] * d0);
                w1 = fma(c[3], d1, c[2] * d0) - m2;
        }
        else
        {
                const double e = mb + m2;
                w0 = m0 - fma(c[1], e, c[0] * ma);
                w1 = m1 - fma(c[3], e, c[2] * ma);
        }
}

/// P <- T P T^T on the symmetric P held as lower tiles in LDS, T = H (FWD) or H^-1.  T is the identity on the pose and
/// couples landmark i only to the pose and to itself, so the 2x2 block (i, j) of the result needs the pose block, the pose
/// columns of landmarks i and j and its own old value: after a side copy of the pose columns (`pose`: column-major [3][ps], LDS)
/// every block is transformed in place, in one pass: W = T_i M (2 x 5, column by column), block = W T_j^T.
/// A thread walks up to CONG_SEG blocks j = j0 .. of ONE landmark row i (round 4): what depends on i alone -- the coefficients, the pose
/// columns of its two rows, the three pose columns of W, the row part of the tile addresses -- is formed once per thread, and the thread
/// of a row's first segment also stores the landmark-pose block (it is those three columns of W).  One block per thread before: ~ 300
/// instructions per block, half of them index arithmetic, three passes of the workgroup at 64 landmarks and 9 k cycles per call, bound by
/// instruction issue (a 4x4-entry ownership was no better: fewer instructions, fewer busy waves; LDS bank conflicts were not it either:
/// profiles/r04_experiments.md section 12).  Rows come in groups of CONG_SEG (group a: a + 1 segments per row), which makes the decode
/// of (i, j0) from the flat index one square root.  Every block is formed by the same expressions as before, bit for bit.
/// Ends with a barrier.
constexpr int CONG_SEG = 3;
template <bool FWD> __device__ __forceinline__ void congruence_tiles(double *Lt, double *pose, const double *sH, int nlm, int n, int nl, int tid)
{
        const int ps = 2 * nlm; // (= NP >= n)
        for (int idx = tid; idx < 3 * n; idx += SMALL_WG)
        {
                const int k = idx / n, r = idx - k * n;
                pose[k * ps + r] = sym_get(Lt, r, k);
        }
        __syncthreads();
        const double *pq0 = pose, *pq1 = pose + ps, *pq2 = pose + 2 * ps; // pqk[r] = P(r, k)
        constexpr int SEG = CONG_SEG;
        const int ng = (nl + SEG - 1) / SEG;
        const int nseg = SEG * ng * (ng + 1) / 2;
        for (int w = tid; w < nseg; w += SMALL_WG)
        {
                // group a: SEG a (a + 1) / 2 <= w
                int a = (int)((sqrtf(1.0f + (8.0f / (float)SEG) * (float)w) - 1.0f) * 0.5f);
                while (SEG * (a + 1) * (a + 2) / 2 <= w)
                        ++a;
                while (SEG * a * (a + 1) / 2 > w)
                        --a;
                const int rem = w - SEG * a * (a + 1) / 2; // < SEG (a + 1)
                int bq = 0;
#pragma unroll
                for (int u = 1; u < SEG; ++u)
                        bq += (rem >= u * (a + 1)) ? 1 : 0;
                const int sg = rem - bq * (a + 1);
                const int i = SEG * a + bq;
                if (i >= nl)
                        continue;
                const int j0 = SEG * sg;
                const int ra = 3 + 2 * i;
                const LmCoef ci = lm_coef<FWD>(sH, nlm, i);
                // columns of M = old P over rows (p0, p1, p2, a_i, b_i); columns (p0, p1, p2, a_j, b_j)
                double W0[5], W1[5];
                lm_rows<FWD>(ci, pq0[0], pq0[1], pq0[2], pq0[ra], pq0[ra + 1], W0[0], W1[0]);
                lm_rows<FWD>(ci, pq1[0], pq1[1], pq1[2], pq1[ra], pq1[ra + 1], W0[1], W1[1]);
                lm_rows<FWD>(ci, pq2[0], pq2[1], pq2[2], pq2[ra], pq2[ra + 1], W0[2], W1[2]);
                // tile_elem(r, c) = Lt + rowp(r) + colp(c) for c <= r
                double *const row_a = Lt + tile_index(ra >> 4, 0) * TSZ + (ra & 15) * TLD;
                double *const row_b = Lt + tile_index((ra + 1) >> 4, 0) * TSZ + ((ra + 1) & 15) * TLD;
                if (sg == 0)
                {
                        // landmark-pose block: rows 3+2i, 4+2i; columns 0..2 = the first three columns of W
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                        {
                                row_a[k] = W0[k];
                                row_b[k] = W1[k];
                        }
                }
#pragma unroll
                for (int u = 0; u < SEG; ++u)
                {
                        const int j = j0 + u;
                        if (j > i)
                                break;
                        const int ca = 3 + 2 * j;
                        const LmCoef cj = lm_coef<FWD>(sH, nlm, j);
                        const int col_a = (ca >> 4) * TSZ + (ca & 15), col_b = ((ca + 1) >> 4) * TSZ + ((ca + 1) & 15);
                        double *e_aa = row_a + col_a, *e_ba = row_b + col_a, *e_bb = row_b + col_b;
                        double *e_ab = (i == j) ? e_ba : row_a + col_b;
                        lm_rows<FWD>(ci, pq0[ca], pq1[ca], pq2[ca], *e_aa, *e_ba, W0[3], W1[3]);
                        lm_rows<FWD>(ci, pq0[ca + 1], pq1[ca + 1], pq2[ca + 1], *e_ab, *e_bb, W0[4], W1[4]);
                        // block = W T_j^T: the rows of T_j applied to the rows of W
                        double o00, o01, o10, o11;
                        lm_rows<FWD>(cj, W0[0], W0[1], W0[2], W0[3], W0[4], o00, o01);
                        lm_rows<FWD>(cj, W1[0], W1[1], W1[2], W1[3], W1[4], o10, o11);
                        *e_aa = o00;
                        *e_ba = o10;
                        *e_bb = o11;
                        if (i != j)
                                *e_ab = o01; // the diagonal block is symmetric: lower entries only
                }
        }
        __syncthreads();
}

template <int NT, int MODE>
__global__ __launch_bounds__(SMALL_WG) void ekf_small_kernel(DevView d, int64_t t0, int nsteps, double *poses_out,
                                                              int32_t *dims_out, StepArgs sa)
{
        typedef SmallLayout<NT> LY;
        constexpr int NP = LY::NP, NLM = NP / 2;
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
        const SmallLds L = small_carve<NT>(smem);
        double *const Lt = L.Lt, *const Dinv = L.Dinv, *const sX = L.sX, *const sZ = L.sZ, *const sY = L.sY, *const sU = L.sU,
                      *const sH = L.sH;
        SmallShared &sm = *L.sm;

        const int tid_launch = threadIdx.x, tid = tid_launch;
        const int b = (MODE == MODE_STEP && sa.traj >= 0) ? sa.traj : (int)blockIdx.x; // sa.traj < 0: the batched step, one workgroup per filter
        double *Pg = d.P + (size_t)b * NP * NP;
        const double r_meas = (double)KR, q_proc = (double)KQ;
#ifdef ASLAM_STAMPS
        unsigned long long stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
#endif

        small_load<MODE>(d, L, b, tid, NP);
        // P stays in LDS for the whole launch, as the lower 16x16 tiles of the symmetric matrix (the tile storage the solver
        // factors in place): HBM sees it once on the way in and once on the way out
        for (int idx = tid; idx < LY::NTILES * 256; idx += SMALL_WG)
        {
                const int tl = idx >> 8, e = idx & 255;
                int ib = (int)((sqrtf(8.0f * (float)tl + 1.0f) - 1.0f) * 0.5f);
                while ((ib + 1) * (ib + 2) / 2 <= tl)
                        ++ib;
                while (ib * (ib + 1) / 2 > tl)
                        --ib;
                const int jb = tl - ib * (ib + 1) / 2;
                Lt[tl * TSZ + (e >> 4) * TLD + (e & 15)] = Pg[(size_t)(16 * ib + (e >> 4)) * NP + 16 * jb + (e & 15)];
        }
        __syncthreads();

        for (int s = 0; s < nsteps; ++s)
        {
                // hipcc hoists every tid-derived address of the ~20 loops below out of this loop and then spills them (150+ VGPRs,
                // reloaded through scratch in every phase): an opaque re-definition per callback keeps them local to their phase
                int tid = tid_launch;
                asm volatile("" : "+v"(tid));
                const int64_t t = t0 + s;
                if (MODE == MODE_REPLAY)
                {
                        if (small_frontend<true, SMALL_OBS_CAP, SMALL_WAIT_CAP, NP / 2, double>(d, L, Pg, NP, b, t, s, nsteps, poses_out, dims_out, tid, Lt))
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
                // ================= slam(), ekf.cpp:293-311
                const int n = sm.n;
                const int nl = (n - 3) / 2;
                const int nt = (n + 15) >> 4;
                if (MODE == MODE_REPLAY && s + 1 < nsteps && __builtin_amdgcn_readfirstlane(tid_launch) >= SMALL_WG - 64) // (a scalar branch: one whole wave)
                        small_prefetch_intake<SMALL_OBS_CAP>(d, L, b, t + 1, tid_launch & 63); // (the last wave has no landmark of the loop below at n <= 143)
                double xp0, xp1, xp2; // the predicted pose
                if (MODE == MODE_REPLAY)
                {
                        // formed by an idle wave of the front end (small_frontend): nobody reads sX[0..2] before the barrier behind the H coefficients
                        xp0 = sm.pp0, xp1 = sm.pp1, xp2 = sm.pp2;
                        if (tid == 0)
                                sX[0] = xp0, sX[1] = xp1, sX[2] = xp2;
                }
                else
                {
                        if (tid == 0)
                        {
                                double p0 = sX[0], p1 = sX[1], p2 = sX[2];
                                stateTransition(p0, p1, p2, sm.vx, sm.az, sm.dt, false, 0.0);
                                sX[0] = p0;
                                sX[1] = p1;
                                sX[2] = (double)normalizeAngle((float)p2);
                        }
                        __syncthreads();
                        xp0 = sX[0], xp1 = sX[1], xp2 = sX[2];
                }
                // updateH (ekf.cpp:117-134) as per-landmark coefficients, their 2x2 inverse, and Y = Z - h(X)
                for (int i = tid; i < nl; i += SMALL_WG)
                {
                        const double x0 = xp0, x1 = xp1;
                        const double lx = sX[3 + 2 * i], ly = sX[4 + 2 * i];
                        const double ddx = lx - x0, ddy = ly - x1;
                        const float hyp = (float)(ddx * ddx + ddy * ddy);
                        const float dist = sqrtf(hyp);
                        const double h00 = (-lx + x0) / (double)dist;
                        const double h01 = (-ly + x1) / (double)dist;
                        const double h10 = -(-ly + x1) / (double)hyp;
                        const double h11 = (-lx + x0) / (double)hyp;
                        const double det = h00 * h11 - h01 * h10;
                        double *hc = sH + i; // coefficient-major: hc[k * NLM] (lm_coef)
                        hc[0] = h00;
                        hc[NLM] = h01;
                        hc[2 * NLM] = h10;
                        hc[3 * NLM] = h11;
                        hc[4 * NLM] = h11 / det;
                        hc[5 * NLM] = -h01 / det;
                        hc[6 * NLM] = -h10 / det;
                        hc[7 * NLM] = h00 / det;
                        // measurementFunction, common.h:78-90
                        const double hr = sqrt(ddx * ddx + ddy * ddy);
                        const double hb = atan2(ddy, ddx) - xp2;
                        sY[3 + 2 * i] = sZ[3 + 2 * i] - hr;
                        sY[4 + 2 * i] = (double)normalizeAngle((float)(sZ[4 + 2 * i] - hb));
                }
                if (tid == 0)
                {
                        sY[0] = sZ[0] - xp0;
                        sY[1] = sZ[1] - xp1;
                        sY[2] = (double)normalizeAngle((float)(sZ[2] - xp2));
                }
                for (int i = n + tid; i < NP; i += SMALL_WG)
                        sY[i] = 0.0;
                // P = A P A^T + Q (ekf.cpp:297) with A = I except A(0,0), A(1,0): on the symmetric storage that is columns 0, 1 of
                // the rows below, and the 2x2 corner
                __syncthreads(); // sY / sH are complete, nobody is still reading P
                {
                        const double a00 = sm.a00, a10 = sm.a10;
                        for (int r = 2 + tid; r < n; r += SMALL_WG)
                        {
                                const double p0 = *tile_elem(Lt, r, 0);
                                *tile_elem(Lt, r, 0) = a00 * p0;
                                *tile_elem(Lt, r, 1) = a10 * p0 + *tile_elem(Lt, r, 1);
                        }
                        if (tid == 0)
                        {
                                const double p00 = Lt[0], p10 = Lt[TLD], p11 = Lt[TLD + 1];
                                const double r10 = a10 * p00 + p10; // (A P)(1,0)
                                Lt[0] = a00 * (a00 * p00) + q_proc;
                                Lt[TLD] = a00 * r10;
                                Lt[TLD + 1] = a10 * r10 + (a10 * p10 + p11) + q_proc;
                                Lt[2 * TLD + 2] += q_proc;
                        }
                        __syncthreads();
                }
                ASLAM_STAMP(1);
                // Pt = H P H^T, in place on the tiles (the inverted-diagonal-tile area is free until the solve: side copy of the pose columns)
                congruence_tiles<true>(Lt, Dinv, sH, NLM, n, nl, tid);
                ASLAM_STAMP(2);
                ASLAM_STAMP(3);
                ASLAM_STAMP(4);
                // S = Pt + R = L L^T (ekf.cpp:300); r*Kt = r I - r^2 S^-1 -> the tiles (= (I - K H) P in measurement coordinates,
                // ekf.cpp:301,310); u = Kt Y = Y - r S^-1 Y
#ifdef ASLAM_STAMPS
                cholesky_inverse_tiles<NT>(Lt, Dinv, nt, n, sY, sU, L.sTv, r_meas, tid, &sm.status, (blockIdx.x == 0 && d.dbg) ? d.dbg + 16 : nullptr);
#else
                cholesky_inverse_tiles<NT>(Lt, Dinv, nt, n, sY, sU, L.sTv, r_meas, tid, &sm.status);
#endif
                ASLAM_STAMP(5);
                __syncthreads();
                ASLAM_STAMP(6);
                // X = X + K Y = X + H^-1 u (ekf.cpp:309)
                {
                        const double u0 = sU[0], u1 = sU[1], u2 = sU[2];
                        for (int i = tid; i < nl; i += SMALL_WG)
                        {
                                const double *hc = sH + i;
                                const double ua = sU[3 + 2 * i], ub = sU[4 + 2 * i] + u2;
                                sX[3 + 2 * i] += u0 - (hc[4 * NLM] * ua + hc[5 * NLM] * ub);
                                sX[4 + 2 * i] += u1 - (hc[6 * NLM] * ua + hc[7 * NLM] * ub);
                        }
                        __syncthreads();
                        if (tid < 3)
                                sX[tid] += sU[tid];
                }
                ASLAM_STAMP(7);
                // P = (I - K H) P = H^-1 (r Kt) H^-T (ekf.cpp:310), in place on the tiles
                congruence_tiles<false>(Lt, Dinv, sH, NLM, n, nl, tid);
                ASLAM_STAMP(8);
                ASLAM_STAMP(9);
                if (MODE == MODE_REPLAY)
                {
                        if (tid < 3 && poses_out)
                                poses_out[((size_t)b * nsteps + s) * 3 + tid] = sX[tid];
                        if (tid == 0 && dims_out)
                                dims_out[(size_t)b * nsteps + s] = n;
                }
        }

#ifdef ASLAM_STAMPS
        if (tid == 0 && blockIdx.x == 0 && d.dbg)
                for (int i = 0; i < 12; ++i)
                        d.dbg[i] += stamp_acc[i];
#endif
        // P back to HBM, both triangles
        __syncthreads();
        for (int idx = tid; idx < NP * NP; idx += SMALL_WG)
        {
                const int i = idx / NP, j = idx - i * NP;
                Pg[idx] = sym_get(Lt, i, j);
        }
        small_store<MODE>(d, L, b, tid, NP);
}
} // namespace aslam
