// ekf_small.h -- fused per-trajectory EKF-SLAM kernel (see small_common.h for the design notes and shared pieces).
#pragma once

#include "small_common.h"

namespace aslam
{
template <int NT, int MODE>
__global__ __launch_bounds__(SMALL_WG) void ekf_small_kernel(DevView d, int64_t t0, int nsteps, double *poses_out,
                                                              int32_t *dims_out, StepArgs sa)
{
        typedef SmallLayout<NT> LY;
        constexpr int NP = LY::NP;
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
        const SmallLds L = small_carve<NT>(smem);
        double *const Lt = L.Lt, *const Dinv = L.Dinv, *const sX = L.sX, *const sZ = L.sZ, *const sY = L.sY, *const sU = L.sU,
                      *const sH = L.sH;
        SmallShared &sm = *L.sm;

        const int tid = threadIdx.x;
        const int b = (MODE == MODE_STEP) ? sa.traj : (int)blockIdx.x;
        double *Pg = d.P + (size_t)b * NP * NP;
        const double r_meas = (double)KR, q_proc = (double)KQ;
#ifdef ASLAM_STAMPS
        unsigned long long stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
#endif

        small_load<MODE>(d, L, b, tid, NP);

        for (int s = 0; s < nsteps; ++s)
        {
                const int64_t t = t0 + s;
                if (MODE == MODE_REPLAY)
                {
                        if (small_frontend<true, SMALL_OBS_CAP, SMALL_WAIT_CAP, NP / 2, double>(d, L, Pg, NP, b, t, s, nsteps, poses_out, dims_out, tid))
                                continue;
                }
                else
                {
                        if (tid == 0)
                        {
                                sm.vx = sa.vx;
                                sm.az = sa.az;
                                sm.dt = sa.dt;
                        }
                        __syncthreads();
                }

                ASLAM_STAMP(0);
                // ================= slam(), ekf.cpp:293-311
                const int n = sm.n;
                const int nl = (n - 3) / 2;
                const int nt = (n + 15) >> 4;
                if (tid == 0)
                {
                        double p0 = sX[0], p1 = sX[1], p2 = sX[2];
                        stateTransition(p0, p1, p2, sm.vx, sm.az, sm.dt, false, 0.0);
                        sX[0] = p0;
                        sX[1] = p1;
                        sX[2] = (double)normalizeAngle((float)p2);
                }
                __syncthreads();
                // updateH (ekf.cpp:117-134) as per-landmark coefficients, their 2x2 inverse, and Y = Z - h(X)
                for (int i = tid; i < nl; i += SMALL_WG)
                {
                        const double x0 = sX[0], x1 = sX[1];
                        const double lx = sX[3 + 2 * i], ly = sX[4 + 2 * i];
                        const double ddx = lx - x0, ddy = ly - x1;
                        const float hyp = (float)(ddx * ddx + ddy * ddy);
                        const float dist = sqrtf(hyp);
                        const double h00 = (-lx + x0) / (double)dist;
                        const double h01 = (-ly + x1) / (double)dist;
                        const double h10 = -(-ly + x1) / (double)hyp;
                        const double h11 = (-lx + x0) / (double)hyp;
                        const double det = h00 * h11 - h01 * h10;
                        double *hc = sH + 8 * i;
                        hc[0] = h00;
                        hc[1] = h01;
                        hc[2] = h10;
                        hc[3] = h11;
                        hc[4] = h11 / det;
                        hc[5] = -h01 / det;
                        hc[6] = -h10 / det;
                        hc[7] = h00 / det;
                        // measurementFunction, common.h:78-90
                        const double hr = sqrt(ddx * ddx + ddy * ddy);
                        const double hb = atan2(ddy, ddx) - sX[2];
                        sY[3 + 2 * i] = sZ[3 + 2 * i] - hr;
                        sY[4 + 2 * i] = (double)normalizeAngle((float)(sZ[4 + 2 * i] - hb));
                }
                if (tid == 0)
                {
                        sY[0] = sZ[0] - sX[0];
                        sY[1] = sZ[1] - sX[1];
                        sY[2] = (double)normalizeAngle((float)(sZ[2] - sX[2]));
                }
                for (int i = n + tid; i < NP; i += SMALL_WG)
                        sY[i] = 0.0;
                // P = A P A^T + Q (ekf.cpp:297) with A = I except A(0,0), A(1,0): rows 0,1 then columns 0,1
                {
                        const double a00 = sm.a00, a10 = sm.a10;
                        if (tid < n)
                        {
                                const double r0 = Pg[tid];
                                Pg[tid] = a00 * r0;
                                Pg[NP + tid] = a10 * r0 + Pg[NP + tid];
                        }
                        __syncthreads();
                        if (tid < n)
                        {
                                double *row = Pg + (size_t)tid * NP;
                                const double c0 = row[0];
                                double v0 = a00 * c0;
                                double v1 = a10 * c0 + row[1];
                                if (tid == 0)
                                        v0 += q_proc;
                                if (tid == 1)
                                        v1 += q_proc;
                                row[0] = v0;
                                row[1] = v1;
                                if (tid == 2)
                                        row[2] += q_proc;
                        }
                        __syncthreads();
                }
                ASLAM_STAMP(1);
                // Pt = H P H^T in place: rows ...
                for (int idx = tid; idx < nl * n; idx += SMALL_WG)
                {
                        const int i = idx / n, c = idx - i * n;
                        const double *hc = sH + 8 * i;
                        const double p0 = Pg[c], p1 = Pg[NP + c], p2 = Pg[2 * NP + c];
                        double *ra = Pg + (size_t)(3 + 2 * i) * NP + c;
                        const double pa = ra[0], pb = ra[NP];
                        ra[0] = fma(-hc[1], pb, fma(-hc[0], pa, fma(hc[1], p1, hc[0] * p0)));
                        ra[NP] = fma(-hc[3], pb, fma(-hc[2], pa, fma(hc[3], p1, hc[2] * p0) - p2));
                }
                __syncthreads();
                ASLAM_STAMP(2);
                // ... then columns
                for (int idx = tid; idx < n * nl; idx += SMALL_WG)
                {
                        const int a = idx / nl, i = idx - a * nl;
                        const double *hc = sH + 8 * i;
                        double *row = Pg + (size_t)a * NP;
                        const double t0_ = row[0], t1_ = row[1], t2_ = row[2];
                        const double ta = row[3 + 2 * i], tb = row[4 + 2 * i];
                        row[3 + 2 * i] = fma(-hc[1], tb, fma(-hc[0], ta, fma(hc[1], t1_, hc[0] * t0_)));
                        row[4 + 2 * i] = fma(-hc[3], tb, fma(-hc[2], ta, fma(hc[3], t1_, hc[2] * t0_) - t2_));
                }
                __syncthreads();
                ASLAM_STAMP(3);
                // S = Pt + R (ekf.cpp:300) -> lower tiles in LDS; padding rows/columns decouple (unit diagonal)
                {
                        const int ntl = nt * (nt + 1) / 2;
                        for (int idx = tid; idx < ntl * 256; idx += SMALL_WG)
                        {
                                const int tl = idx >> 8, e = idx & 255;
                                int ib = (int)((sqrtf(8.0f * (float)tl + 1.0f) - 1.0f) * 0.5f);
                                while ((ib + 1) * (ib + 2) / 2 <= tl)
                                        ++ib;
                                while (ib * (ib + 1) / 2 > tl)
                                        --ib;
                                const int jb = tl - ib * (ib + 1) / 2;
                                const int i = 16 * ib + (e >> 4), j = 16 * jb + (e & 15);
                                double v = Pg[(size_t)i * NP + j];
                                if (i == j)
                                        v += (i < n) ? r_meas : 1.0;
                                Lt[tl * TSZ + (e >> 4) * TLD + (e & 15)] = v;
                        }
                }
                __syncthreads();
                ASLAM_STAMP(4);
                // Kt = Pt S^-1 (rows of Pt are independent right-hand sides), u = Kt Y; r*Kt written back in place
                #ifdef ASLAM_STAMPS
                cholesky_solve_rows<NT>(Pg, Pg, Lt, Dinv, nt, sY, sU, r_meas, tid, &sm.status,
                                        (blockIdx.x == 0 && d.dbg) ? d.dbg + 16 : nullptr);
#else
                cholesky_solve_rows<NT>(Pg, Pg, Lt, Dinv, nt, sY, sU, r_meas, tid, &sm.status);
#endif
                ASLAM_STAMP(5);
                __syncthreads();
                ASLAM_STAMP(6);
                // X = X + K Y = X + H^-1 u (ekf.cpp:309)
                {
                        const double u0 = sU[0], u1 = sU[1], u2 = sU[2];
                        for (int i = tid; i < nl; i += SMALL_WG)
                        {
                                const double *hc = sH + 8 * i;
                                const double ua = sU[3 + 2 * i], ub = sU[4 + 2 * i] + u2;
                                sX[3 + 2 * i] += u0 - (hc[4] * ua + hc[5] * ub);
                                sX[4 + 2 * i] += u1 - (hc[6] * ua + hc[7] * ub);
                        }
                        __syncthreads();
                        if (tid < 3)
                                sX[tid] += sU[tid];
                }
                ASLAM_STAMP(7);
                // P = (I - K H) P = H^-1 (r Kt) H^-T (ekf.cpp:310), in place: rows ...
                for (int idx = tid; idx < nl * n; idx += SMALL_WG)
                {
                        const int i = idx / n, c = idx - i * n;
                        const double *hc = sH + 8 * i;
                        const double m0 = Pg[c], m1 = Pg[NP + c], m2 = Pg[2 * NP + c];
                        double *ra = Pg + (size_t)(3 + 2 * i) * NP + c;
                        const double ma = ra[0], mb = ra[NP] + m2;
                        ra[0] = m0 - fma(hc[5], mb, hc[4] * ma);
                        ra[NP] = m1 - fma(hc[7], mb, hc[6] * ma);
                }
                __syncthreads();
                ASLAM_STAMP(8);
                // ... then columns
                for (int idx = tid; idx < n * nl; idx += SMALL_WG)
                {
                        const int a = idx / nl, i = idx - a * nl;
                        const double *hc = sH + 8 * i;
                        double *row = Pg + (size_t)a * NP;
                        const double w0 = row[0], w1 = row[1], w2 = row[2];
                        const double wa = row[3 + 2 * i], wb = row[4 + 2 * i] + w2;
                        row[3 + 2 * i] = w0 - fma(hc[5], wb, hc[4] * wa);
                        row[4 + 2 * i] = w1 - fma(hc[7], wb, hc[6] * wa);
                }
                __syncthreads();
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
        small_store<MODE>(d, L, b, tid, NP);
}
} // namespace aslam
