// ekf_small.h -- fused per-trajectory EKF-SLAM kernel for state dimensions that fit one CU (n <= 16*NT <= 144).
//
// One 1024-thread workgroup (16 wave64) owns one filter ("trajectory") and runs whole callbacks of the
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
// which is algebraically identical to ekf.cpp:300-310 (no symmetry of P is assumed) but needs only a
// Cholesky factor and two triangular solves with n right-hand sides (2.33 n^3 flops instead of 18 n^3),
// and H, H^-1, A are applied as the <=5-non-zeros-per-row operators they are.
//
// Data placement: P lives row-major in HBM/L2 (row stride NP = 16*NT doubles, zero padding) and is
// transformed in place; S -> L and the inverted diagonal blocks live in LDS as 16x16 tiles; the
// triangular solves keep a 16-row block of Pt^T per wave in MFMA accumulators (v_mfma_f64_16x16x4_f64),
// taking L tiles from LDS as the A operand and the freshly solved tile, untouched, as the B operand.
#pragma once

#include "device_common.h"

namespace aslam
{
constexpr int SMALL_WG = 1024;
constexpr int SMALL_WAVES = SMALL_WG / 64;
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
};

struct StepArgs
{
        int traj;
        float vx, az, dt;
};

/// scalars of one filter, kept in LDS while the kernel runs
struct SmallShared
{
        int n, flags, sn, wn, nnew, grew, grow_from, any_miss, obs_new, nobs, skip;
        uint32_t status;
        float vx, az, dt, yaw;
        double px, py, tvx, twz, a00, a10;
};

template <int NT> struct SmallLayout
{
        static constexpr int NP = 16 * NT;
        static constexpr int NTILES = NT * (NT + 1) / 2;
        // offsets in doubles
        static constexpr int oL = 0;
        static constexpr int oDinv = oL + NTILES * 256;
        static constexpr int oX = oDinv + NT * 256;
        static constexpr int oZ = oX + NP;
        static constexpr int oY = oZ + NP;
        static constexpr int oU = oY + NP;
        static constexpr int oH = oU + NP;           // per landmark: h00 h01 h10 h11 e00 e01 e10 e11
        static constexpr int oEnd = oH + (NP / 2) * 8;
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
        static constexpr size_t oSm = (oNew + 4 * (NP / 2) + 15) & ~(size_t)15;
        static constexpr size_t total = oSm + sizeof(SmallShared);
};

__device__ __forceinline__ int tile_index(int ib, int jb)
{
        return ib * (ib + 1) / 2 + jb;
}

// ------------------------------------------------------------------------------------------------------
/// Factor the 16x16 diagonal tile `T` (lower triangle valid, LDS, row-major) in place into its Cholesky
/// factor and write the inverse of that factor to `Ti`.  One wave; lane i < 16 owns row i in registers,
/// pivots and multipliers travel through v_readlane.  Returns false on a non-positive pivot.
__device__ __forceinline__ bool factor_diag_tile(double *T, double *Ti, int lane)
{
        double a[16];
        const int row = lane & 15;
#pragma unroll
        for (int c = 0; c < 16; ++c)
                a[c] = T[row * 16 + c];
        bool ok = true;
        double invd[16];
#pragma unroll
        for (int j = 0; j < 16; ++j)
        {
                const double djj = readlane_f64(a[j], j);
                ok = ok && (djj > 0.0);
                const double inv = readfirstlane_f64(1.0 / sqrt(djj)); // wave-uniform: keep it in SGPRs
                invd[j] = inv;
                const double lij = a[j] * inv; // L(i,j) for i >= j
                a[j] = lij;
#pragma unroll
                for (int c = j + 1; c < 16; ++c)
                {
                        const double lcj = readlane_f64(lij, c);
                        a[c] = fma(-lij, lcj, a[c]);
                }
        }
        // inverse: lane c computes column c of L^-1 by forward substitution, L(i,k) broadcast by readlane
        double x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i)
        {
                double s = (row == i) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < i; ++k)
                {
                        const double lik = readlane_f64(a[k], i);
                        s = fma(-lik, x[k], s);
                }
                x[i] = s * invd[i];
        }
        if (lane < 16)
        {
#pragma unroll
                for (int c = 0; c < 16; ++c)
                {
                        T[row * 16 + c] = (c <= row) ? a[c] : 0.0;
                        Ti[c * 16 + row] = x[c]; // Linv(c, row): zero above the diagonal by construction
                }
        }
        return ok;
}

// ------------------------------------------------------------------------------------------------------
/// Blocked Cholesky of the nt x nt tile matrix in LDS (lower block triangle), in place, plus the inverses
/// of the diagonal blocks.  All 16 waves take part; panel and trailing updates run on the f64 MFMA.
template <int NT>
__device__ __forceinline__ void cholesky_tiles(double *Lt, double *Dinv, int nt, int tid, uint32_t *status)
{
        const int wave = tid >> 6, lane = tid & 63;
        const int li = lane & 15, lg = lane >> 4;
        for (int kb = 0; kb < nt; ++kb)
        {
                if (wave == 0)
                {
                        const bool ok = factor_diag_tile(Lt + tile_index(kb, kb) * 256, Dinv + kb * 256, lane);
                        if (!ok && lane == 0)
                                *status |= 4u; // ASLAM_ST_NOT_PD
                }
                __syncthreads();
                // panel: L(ib,kb) = S(ib,kb) * Linv(kb)^T
                for (int ib = kb + 1 + wave; ib < nt; ib += SMALL_WAVES)
                {
                        double *S = Lt + tile_index(ib, kb) * 256;
                        const double *Di = Dinv + kb * 256;
                        d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                        {
                                const int k = lg + 4 * s;
                                acc = mfma_f64(S[li * 16 + k], Di[li * 16 + k], acc);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                                S[(lg + 4 * r) * 16 + li] = acc[r];
                }
                __syncthreads();
                // trailing update: S(ib,jb) -= L(ib,kb) L(jb,kb)^T for kb < jb <= ib
                const int m = nt - kb - 1;
                const int ntr = m * (m + 1) / 2;
                for (int q = wave; q < ntr; q += SMALL_WAVES)
                {
                        // q -> (i, j), 0 <= j <= i < m
                        int i = (int)((sqrtf(8.0f * (float)q + 1.0f) - 1.0f) * 0.5f);
                        while ((i + 1) * (i + 2) / 2 <= q)
                                ++i;
                        while (i * (i + 1) / 2 > q)
                                --i;
                        const int j = q - i * (i + 1) / 2;
                        const int ib = kb + 1 + i, jb = kb + 1 + j;
                        double *S = Lt + tile_index(ib, jb) * 256;
                        const double *Li = Lt + tile_index(ib, kb) * 256;
                        const double *Lj = Lt + tile_index(jb, kb) * 256;
                        d4 acc;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                                acc[r] = S[(lg + 4 * r) * 16 + li];
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                        {
                                const int k = lg + 4 * s;
                                acc = mfma_f64(-Li[li * 16 + k], Lj[li * 16 + k], acc);
                        }
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                                S[(lg + 4 * r) * 16 + li] = acc[r];
                }
                __syncthreads();
        }
}

// ------------------------------------------------------------------------------------------------------
/// Kt = Pt S^-1 for the 16 rows [16*rb, 16*rb+16) of Pt, in place in HBM, scaled by `scale` on the way
/// out; also u[row] = Kt[row,:] . Y (unscaled).  One wave.  The row block is held transposed in MFMA
/// accumulators: acc[cb][r] of lane l is Pt[16 rb + (l&15)][16 cb + (l>>4) + 4 r].
template <int NT>
__device__ __forceinline__ void solve_row_block(double *Pg, int rb, int nt, const double *Lt, const double *Dinv,
                                                const double *Y, double *U, double scale, int lane)
{
        constexpr int NP = 16 * NT;
        const int li = lane & 15, lg = lane >> 4;
        d4 acc[NT];
        double *rowp = Pg + (size_t)(16 * rb + li) * NP + lg;
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
        // forward: V^T = L^-1 Pt^T
#pragma unroll
        for (int cb = 0; cb < NT; ++cb)
        {
                if (cb < nt)
                {
                        const double *Di = Dinv + cb * 256;
                        d4 v = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                                v = mfma_f64(Di[li * 16 + lg + 4 * s], acc[cb][s], v);
                        acc[cb] = v;
#pragma unroll
                        for (int c2 = cb + 1; c2 < NT; ++c2)
                        {
                                if (c2 < nt)
                                {
                                        const double *Lc = Lt + tile_index(c2, cb) * 256;
#pragma unroll
                                        for (int s = 0; s < 4; ++s)
                                                acc[c2] = mfma_f64(-Lc[li * 16 + lg + 4 * s], v[s], acc[c2]);
                                }
                        }
                }
        }
        // backward: Kt^T = L^-T V^T
#pragma unroll
        for (int cb = NT - 1; cb >= 0; --cb)
        {
                if (cb < nt)
                {
                        const double *Di = Dinv + cb * 256;
                        d4 v = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int s = 0; s < 4; ++s)
                                v = mfma_f64(Di[(lg + 4 * s) * 16 + li], acc[cb][s], v);
                        acc[cb] = v;
#pragma unroll
                        for (int c2 = 0; c2 < cb; ++c2)
                        {
                                const double *Lc = Lt + tile_index(cb, c2) * 256;
#pragma unroll
                                for (int s = 0; s < 4; ++s)
                                        acc[c2] = mfma_f64(-Lc[(lg + 4 * s) * 16 + li], v[s], acc[c2]);
                        }
                }
        }
        // u = Kt Y, and write r*Kt back
        double part = 0.0;
#pragma unroll
        for (int cb = 0; cb < NT; ++cb)
        {
                if (cb < nt)
                {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                        {
                                part = fma(acc[cb][r], Y[16 * cb + lg + 4 * r], part);
                                rowp[16 * cb + 4 * r] = acc[cb][r] * scale;
                        }
                }
        }
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        if (lg == 0)
                U[16 * rb + li] = part;
}

// ------------------------------------------------------------------------------------------------------
template <int NT, int MODE>
__global__ __launch_bounds__(SMALL_WG) void ekf_small_kernel(DevView d, int64_t t0, int nsteps, double *poses_out,
                                                              int32_t *dims_out, StepArgs sa)
{
        typedef SmallLayout<NT> LY;
        constexpr int NP = LY::NP;
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
        double *lds = reinterpret_cast<double *>(smem);
        double *Lt = lds + LY::oL, *Dinv = lds + LY::oDinv;
        double *sX = lds + LY::oX, *sZ = lds + LY::oZ, *sY = lds + LY::oY, *sU = lds + LY::oU, *sH = lds + LY::oH;
        float *sSr = reinterpret_cast<float *>(smem + LY::oSr), *sSb = reinterpret_cast<float *>(smem + LY::oSb);
        float *sPx = reinterpret_cast<float *>(smem + LY::oPx), *sPy = reinterpret_cast<float *>(smem + LY::oPy);
        float *sMd = reinterpret_cast<float *>(smem + LY::oMd);
        int *sCid = reinterpret_cast<int *>(smem + LY::oCid);
        float *sWr = reinterpret_cast<float *>(smem + LY::oWr), *sWb = reinterpret_cast<float *>(smem + LY::oWb);
        float *sWx = reinterpret_cast<float *>(smem + LY::oWx), *sWy = reinterpret_cast<float *>(smem + LY::oWy);
        uint32_t *sWc = reinterpret_cast<uint32_t *>(smem + LY::oWc);
        int *sNew = reinterpret_cast<int *>(smem + LY::oNew);
        SmallShared &sm = *reinterpret_cast<SmallShared *>(smem + LY::oSm);

        const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
        const int b = (MODE == MODE_STEP) ? sa.traj : (int)blockIdx.x;
        double *Pg = d.P + (size_t)b * NP * NP;
        const double r_meas = (double)KR, q_proc = (double)KQ;

        // ---- load the filter into LDS
        for (int i = tid; i < NP; i += SMALL_WG)
        {
                sX[i] = d.X[(size_t)b * NP + i];
                sZ[i] = d.Z[(size_t)b * NP + i];
                sY[i] = 0.0;
                sU[i] = 0.0;
        }
        if (tid == 0)
        {
                sm.n = d.n[b];
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
                        sSr[j] = d.sens[((size_t)b * d.max_obs + j) * 2];
                        sSb[j] = d.sens[((size_t)b * d.max_obs + j) * 2 + 1];
                }
                for (int j = tid; j < sm.wn; j += SMALL_WG)
                {
                        sWr[j] = d.wait_rb[((size_t)b * d.max_wait + j) * 2];
                        sWb[j] = d.wait_rb[((size_t)b * d.max_wait + j) * 2 + 1];
                        sWc[j] = d.wait_cnt[(size_t)b * d.max_wait + j];
                }
        }
        __syncthreads();

        for (int s = 0; s < nsteps; ++s)
        {
                const int64_t t = t0 + s;
                if (MODE == MODE_REPLAY)
                {
                        // ================= message intake
                        if (tid == 0)
                        {
                                const size_t o = (size_t)b * d.T + t;
                                sm.px = d.tr_pose[2 * o];
                                sm.py = d.tr_pose[2 * o + 1];
                                sm.yaw = d.tr_yaw[o];
                                sm.tvx = d.tr_twist[2 * o];
                                sm.twz = d.tr_twist[2 * o + 1];
                                sm.dt = d.tr_dt[o];
                                sm.obs_new = d.tr_new[o];
                                int k = d.tr_nobs[o];
                                if (k > d.max_obs || k > SMALL_OBS_CAP)
                                {
                                        sm.status |= 8u; // ASLAM_ST_OBS_OVERFLOW
                                        k = min(d.max_obs, SMALL_OBS_CAP);
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
                                const float *src = d.tr_obs + ((size_t)b * d.T + t) * d.max_obs * 2;
                                for (int j = tid; j < sm.nobs; j += SMALL_WG)
                                {
                                        sSr[j] = src[2 * j];
                                        sSb[j] = src[2 * j + 1];
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
                                continue;
                        }
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
                        __syncthreads();
                        // nearest mapped landmark of every observation (ekf.cpp:159-173): one wave per observation,
                        // lanes over landmarks; ties and NaNs resolve as the sequential `dist < mindist` scan does
                        for (int j = wave; j < sm.sn; j += SMALL_WAVES)
                        {
                                float bd = __builtin_inff();
                                int bk = 0x7fffffff;
                                float d0 = 0.0f;
                                for (int k = lane; k < nl; k += 64)
                                {
                                        const float dd = eulerDistance(sPx[j], sPy[j], (float)sX[3 + 2 * k], (float)sX[4 + 2 * k]);
                                        if (k == 0)
                                                d0 = dd;
                                        if (dd < bd)
                                        {
                                                bd = dd;
                                                bk = k;
                                        }
                                }
#pragma unroll
                                for (int off = 32; off >= 1; off >>= 1)
                                {
                                        const float od = __shfl_xor(bd, off);
                                        const int ok = __shfl_xor(bk, off);
                                        if (od < bd || (od == bd && ok < bk))
                                        {
                                                bd = od;
                                                bk = ok;
                                        }
                                }
                                d0 = __shfl(d0, 0);
                                if (lane == 0)
                                {
                                        if (nl == 0)
                                        {
                                                sMd[j] = __builtin_inff();
                                                sCid[j] = 0;
                                                sm.any_miss = 1;
                                        }
                                        else
                                        {
                                                if (d0 != d0 || bk == 0x7fffffff)
                                                {
                                                        // mindist starts as dist(landmark 0); a NaN there is never replaced,
                                                        // and an all-inf scan keeps corr_id = 0
                                                        bd = d0;
                                                        bk = 0;
                                                }
                                                sMd[j] = bd;
                                                sCid[j] = 2 * bk;
                                                if (!(bd < MIN_DIST_THRESH))
                                                        sm.any_miss = 1;
                                        }
                                }
                        }
                        __syncthreads();
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
                        }
                        __syncthreads();
                        if (tid == 0)
                        {
                                int wn = sm.wn;
                                const int wcap = min(d.max_wait, SMALL_WAIT_CAP);
                                for (int j = 0; j < sm.sn; ++j)
                                {
                                        if (n0 != 3 && sMd[j] < MIN_DIST_THRESH)
                                        {
                                                sZ[3 + sCid[j]] = (double)sSr[j];
                                                sZ[4 + sCid[j]] = (double)sSb[j];
                                                continue;
                                        }
                                        // updateNewLandmarkWait, ekf.cpp:217-253
                                        bool push = (wn == 0);
                                        if (!push)
                                        {
                                                int corr = 0;
                                                float mind = eulerDistance(sPx[j], sPy[j], sWx[0], sWy[0]);
                                                for (int i = 1; i < wn; ++i)
                                                {
                                                        const float dd = eulerDistance(sPx[j], sPy[j], sWx[i], sWy[i]);
                                                        if (dd < mind)
                                                        {
                                                                corr = i;
                                                                mind = dd;
                                                        }
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
                                                        ++wn;
                                                }
                                                else
                                                        sm.status |= 2u; // ASLAM_ST_WAIT_OVERFLOW
                                        }
                                }
                                sm.wn = wn;
                                // promotion, ekf.cpp:187-195
                                int nnew = 0;
                                for (int i = 0; i < wn; ++i)
                                {
                                        if (sWc[i] == MIN_LANDMARK_OCC)
                                        {
                                                if (nnew < NP / 2)
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
                                // Update A, ekf.cpp:206-212
                                if (sm.tvx != 0.0 && sm.twz != 0.0)
                                {
                                        const float delta_theta = (float)(sm.twz * (double)sm.dt);
                                        const float rr = (float)(sm.tvx / sm.twz);
                                        sm.a00 = (double)rr * (-cos(sZ[2]) + cos(sZ[2] + (double)delta_theta));
                                        sm.a10 = (double)rr * (-sin(sZ[2]) + sin(sZ[2] + (double)delta_theta));
                                }
                                sm.vx = (float)sm.tvx; // slam(const float &vx, ...), ekf.cpp:94,293
                                sm.az = (float)sm.twz;
                        }
                        __syncthreads();
                        if (sm.grew)
                        {
                                // conservativeResizeLike(Identity * UKF_KP_LANDMARK_POSE), ekf.cpp:277
                                const int g0 = sm.grow_from, n1 = sm.n;
                                for (int idx = tid; idx < n1 * n1; idx += SMALL_WG)
                                {
                                        const int i = idx / n1, j = idx - i * n1;
                                        if (i >= g0 || j >= g0)
                                                Pg[(size_t)i * NP + j] = (i == j) ? (double)KP_LANDMARK_POSE : 0.0;
                                }
                        }
                        if (sm.flags & FLAG_INIT_X)
                        {
                                // param.X = param.Z once, ekf.cpp:87-91
                                for (int i = tid; i < sm.n; i += SMALL_WG)
                                        sX[i] = sZ[i];
                                __syncthreads();
                                if (tid == 0)
                                        sm.flags &= ~FLAG_INIT_X;
                        }
                        __syncthreads();
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
                                Lt[idx] = v;
                        }
                }
                __syncthreads();
                cholesky_tiles<NT>(Lt, Dinv, nt, tid, &sm.status);
                // Kt = Pt S^-1 (rows of Pt are independent right-hand sides), u = Kt Y
                if (wave < nt)
                        solve_row_block<NT>(Pg, wave, nt, Lt, Dinv, sY, sU, r_meas, lane);
                __syncthreads();
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
                if (MODE == MODE_REPLAY)
                {
                        if (tid < 3 && poses_out)
                                poses_out[((size_t)b * nsteps + s) * 3 + tid] = sX[tid];
                        if (tid == 0 && dims_out)
                                dims_out[(size_t)b * nsteps + s] = n;
                }
        }

        // ---- store the filter
        __syncthreads();
        for (int i = tid; i < NP; i += SMALL_WG)
        {
                d.X[(size_t)b * NP + i] = sX[i];
                d.Z[(size_t)b * NP + i] = sZ[i];
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
                        d.sens[((size_t)b * d.max_obs + j) * 2] = sSr[j];
                        d.sens[((size_t)b * d.max_obs + j) * 2 + 1] = sSb[j];
                }
                for (int j = tid; j < sm.wn; j += SMALL_WG)
                {
                        d.wait_rb[((size_t)b * d.max_wait + j) * 2] = sWr[j];
                        d.wait_rb[((size_t)b * d.max_wait + j) * 2 + 1] = sWb[j];
                        d.wait_cnt[(size_t)b * d.max_wait + j] = sWc[j];
                }
        }
}
} // namespace aslam
