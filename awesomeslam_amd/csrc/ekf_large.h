// ekf_large.h -- EKF-SLAM for state dimensions beyond one CU (144 < n <= 1087; BASELINE configs[3]/[4]: 512 landmarks,
// n = 1027), fp64 or fp32.  One callback = a short chain of launches that use the whole GPU for one filter (and
// `batch` filters side by side through blockIdx.y/z):
//
//   large_frontend   cbSensorLandmark + updateZandA (ekf.cpp:102-213, shared small_frontend code), predict
//                    (X <- f(X), P <- A P A^T + Q with A = I + 2 entries, ekf.cpp:295-297), updateH coefficients
//                    (ekf.cpp:117-134), Y = Z - h(X) (ekf.cpp:302-307)                               1 workgroup / filter
//   large_build_G    G = P H^T             (H has <= 5 non-zeros per row: a column pass, ekf.cpp:301)
//   large_build_S    S = H G + R           (row pass, ekf.cpp:300), Y^T appended to G as row n
//   17 x { gemm_nt, potrf_diag, panel_solve }   blocked left-looking Cholesky S = L L^T of the STACKED matrix
//                    [S; G; Y^T]: the same panel / trailing launches that factor S turn G into V = G L^-T and Y^T into
//                    (L^-1 Y)^T, so  K = P H^T S^-1 = V L^-1  never needs a separate triangular solve
//   gemm_nt          P <- P - V V^T        ( = (I - K H) P, ekf.cpp:310, since K H P = V V^T for symmetric P )
//   large_x_update   X <- X + V (L^-1 Y)   ( = X + K Y, ekf.cpp:309 )
//
// Unlike the single-CU kernel this path does not go through measurement coordinates: in fp32 the H / H^-1 change of
// basis would cost eps * cond(H)^2 per callback; G, S, V are formed directly (2.33 n^3 flops all the same).
// It does use the symmetry of P (P H^T = (H P)^T, K H P = V V^T); the reference's P is symmetric up to rounding.
// Matrices are row-major in HBM with row stride NP (multiple of 64, zero padding); GEMMs run on the 16x16x4 MFMA
// (f64 or f32 inputs), operands staged through LDS.
#pragma once

#include "small_common.h"

namespace aslam
{
constexpr int LB = 64;               // block size of the factorisation = GEMM tile edge
constexpr int LARGE_OBS_CAP = 1024;  // stored sensor message (LDS)
constexpr int LARGE_WAIT_CAP = 2048; // wait-list (LDS)
constexpr int LARGE_NP_MAX = 1088;   // 17 blocks of 64: n <= 1087

template <typename T> struct LargeView
{
        int NP;     // row stride, multiple of LB
        T *P;       // [B][NP][NP]
        T *G;       // [B][NP][NP]  P H^T, then V; row n carries Y^T, then (L^-1 Y)^T
        T *S;       // [B][NP][NP]  innovation covariance, then L (lower)
        double *Hc; // [B][NP/2][4] h00 h01 h10 h11 per landmark
        double *Y;  // [B][NP]
};

// ---- MFMA traits -------------------------------------------------------------------------------------------------
typedef float f4 __attribute__((ext_vector_type(4)));
template <typename T> struct Mfma;
template <> struct Mfma<double>
{
        typedef d4 acc_t;
        static __device__ __forceinline__ acc_t zero()
        {
                return (acc_t){0.0, 0.0, 0.0, 0.0};
        }
        static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c)
        {
                return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        }
        // C/D element (row, col) of register r in lane l
        static __device__ __forceinline__ int row(int lane, int r)
        {
                return (lane >> 4) + 4 * r;
        }
};
template <> struct Mfma<float>
{
        typedef f4 acc_t;
        static __device__ __forceinline__ acc_t zero()
        {
                return (acc_t){0.f, 0.f, 0.f, 0.f};
        }
        static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c)
        {
                return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
        }
        static __device__ __forceinline__ int row(int lane, int r)
        {
                return (lane >> 4) * 4 + r;
        }
};

/// number of active 64-blocks: the n state rows plus the Y^T row
__device__ __forceinline__ int large_blocks(int n)
{
        return (n + 1 + LB - 1) / LB;
}

// ---- LDS layout of the front-end workgroup (runtime NP) -----------------------------------------------------------
struct LargeLds
{
        static __host__ __device__ size_t bytes(int NP)
        {
                size_t o = 0;
                o += 8 * (size_t)NP * 2;                  // sX, sZ
                o += 4 * (size_t)LARGE_OBS_CAP * 6;       // sSr sSb sPx sPy sMd sCid
                o += 4 * (size_t)LARGE_WAIT_CAP * 5;      // sWr sWb sWx sWy sWc
                o += 4 * (size_t)(NP / 2);                // sNew
                o += 4 * (size_t)LARGE_OBS_CAP * 4 * 2;   // sPd sPi
                o += 4 * (size_t)NP;                      // sLm
                o = (o + 15) & ~(size_t)15;
                return o + sizeof(SmallShared);
        }
        static __device__ __forceinline__ SmallLds carve(unsigned char *smem, int NP)
        {
                SmallLds L;
                unsigned char *p = smem;
                L.Lt = nullptr;
                L.Dinv = nullptr;
                L.sY = nullptr;
                L.sU = nullptr;
                L.sH = nullptr;
                L.sX = reinterpret_cast<double *>(p);
                p += 8 * (size_t)NP;
                L.sZ = reinterpret_cast<double *>(p);
                p += 8 * (size_t)NP;
#define CARVE(field, type, count)                                                                                      \
        L.field = reinterpret_cast<type *>(p);                                                                         \
        p += sizeof(type) * (size_t)(count);
                CARVE(sSr, float, LARGE_OBS_CAP)
                CARVE(sSb, float, LARGE_OBS_CAP)
                CARVE(sPx, float, LARGE_OBS_CAP)
                CARVE(sPy, float, LARGE_OBS_CAP)
                CARVE(sMd, float, LARGE_OBS_CAP)
                CARVE(sCid, int, LARGE_OBS_CAP)
                CARVE(sWr, float, LARGE_WAIT_CAP)
                CARVE(sWb, float, LARGE_WAIT_CAP)
                CARVE(sWx, float, LARGE_WAIT_CAP)
                CARVE(sWy, float, LARGE_WAIT_CAP)
                CARVE(sWc, uint32_t, LARGE_WAIT_CAP)
                CARVE(sNew, int, NP / 2)
                CARVE(sPd, float, LARGE_OBS_CAP * 4)
                CARVE(sPi, int, LARGE_OBS_CAP * 4)
                CARVE(sLm, float, NP)
#undef CARVE
                const size_t off = ((size_t)(p - smem) + 15) & ~(size_t)15;
                L.sm = reinterpret_cast<SmallShared *>(smem + off);
                return L;
        }
};

// ------------------------------------------------------------------------------------------------------------------
/// Per-filter front end + predict (one workgroup of SMALL_WG threads per filter).  `skipped` [B] is set to 1 when the
/// callback returned early (no sensor message yet): the rest of the chain then leaves the filter alone.
template <typename T, int MODE>
__global__ __launch_bounds__(SMALL_WG) void large_frontend_kernel(DevView d, LargeView<T> lv, int64_t t, int s, int nsteps,
                                                                   double *poses_out, int32_t *dims_out, StepArgs sa, int *skipped)
{
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
        const int NP = lv.NP;
        const SmallLds L = LargeLds::carve(smem, NP);
        SmallShared &sm = *L.sm;
        double *const sX = L.sX, *const sZ = L.sZ;
        const int tid = threadIdx.x;
        const int b = (MODE == MODE_STEP) ? sa.traj : (int)blockIdx.x;
        T *Pg = lv.P + (size_t)b * NP * NP;
        T *Gg = lv.G + (size_t)b * NP * NP;
        double *Hc = lv.Hc + (size_t)b * (NP / 2) * 4;
        double *Yg = lv.Y + (size_t)b * NP;

        small_load<MODE>(d, L, b, tid, NP);
        if (MODE == MODE_REPLAY)
        {
                if (small_frontend<true, LARGE_OBS_CAP, LARGE_WAIT_CAP, LARGE_NP_MAX / 2, T>(d, L, Pg, NP, b, t, s, nsteps, poses_out,
                                                                                               dims_out, tid))
                {
                        if (tid == 0)
                                skipped[b] = 1;
                        small_store<MODE>(d, L, b, tid, NP);
                        return;
                }
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
        if (tid == 0)
                skipped[b] = 0;
        const int n = sm.n;
        const int nl = (n - 3) / 2;
        // X <- f(X), ekf.cpp:295-296
        if (tid == 0)
        {
                double p0 = sX[0], p1 = sX[1], p2 = sX[2];
                stateTransition(p0, p1, p2, sm.vx, sm.az, sm.dt, false, 0.0);
                sX[0] = p0;
                sX[1] = p1;
                sX[2] = (double)normalizeAngle((float)p2);
        }
        __syncthreads();
        // updateH coefficients (ekf.cpp:117-134) and Y = Z - h(X) (ekf.cpp:302-307)
        for (int i = tid; i < nl; i += SMALL_WG)
        {
                const double x0 = sX[0], x1 = sX[1];
                const double lx = sX[3 + 2 * i], ly = sX[4 + 2 * i];
                const double ddx = lx - x0, ddy = ly - x1;
                const float hyp = (float)(ddx * ddx + ddy * ddy);
                const float dist = sqrtf(hyp);
                Hc[4 * i + 0] = (-lx + x0) / (double)dist;
                Hc[4 * i + 1] = (-ly + x1) / (double)dist;
                Hc[4 * i + 2] = -(-ly + x1) / (double)hyp;
                Hc[4 * i + 3] = (-lx + x0) / (double)hyp;
                const double hr = sqrt(ddx * ddx + ddy * ddy);
                const double hb = atan2(ddy, ddx) - sX[2];
                Yg[3 + 2 * i] = sZ[3 + 2 * i] - hr;
                Yg[4 + 2 * i] = (double)normalizeAngle((float)(sZ[4 + 2 * i] - hb));
        }
        if (tid == 0)
        {
                Yg[0] = sZ[0] - sX[0];
                Yg[1] = sZ[1] - sX[1];
                Yg[2] = (double)normalizeAngle((float)(sZ[2] - sX[2]));
        }
        // P <- A P A^T + Q (ekf.cpp:297), A = I except A(0,0), A(1,0): rows 0,1 then columns 0,1
        {
                const T a00 = (T)sm.a00, a10 = (T)sm.a10, q = (T)(double)KQ;
                for (int c = tid; c < n; c += SMALL_WG)
                {
                        const T r0 = Pg[c];
                        Pg[c] = a00 * r0;
                        Pg[NP + c] = a10 * r0 + Pg[NP + c];
                }
                __syncthreads();
                for (int r = tid; r < n; r += SMALL_WG)
                {
                        T *row = Pg + (size_t)r * NP;
                        const T c0 = row[0];
                        T v0 = a00 * c0, v1 = a10 * c0 + row[1];
                        if (r == 0)
                                v0 += q;
                        if (r == 1)
                                v1 += q;
                        row[0] = v0;
                        row[1] = v1;
                        if (r == 2)
                                row[2] += q;
                }
        }
        (void)Gg;
        small_store<MODE>(d, L, b, tid, NP);
}

// ------------------------------------------------------------------------------------------------------------------
/// G = P H^T (one workgroup per matrix row; lanes over landmark pairs).  grid (NP, B).
template <typename T> __global__ __launch_bounds__(256) void large_build_G(DevView d, LargeView<T> lv, const int *skipped)
{
        const int b = blockIdx.y, a = blockIdx.x;
        if (skipped[b])
                return;
        const int n = d.n[b], NP = lv.NP;
        const int na = large_blocks(n) * LB;
        if (a >= na)
                return;
        const T *prow = lv.P + ((size_t)b * NP + a) * NP;
        T *grow = lv.G + ((size_t)b * NP + a) * NP;
        const double *Hc = lv.Hc + (size_t)b * (NP / 2) * 4;
        const int nl = (n - 3) / 2;
        if (a >= n)
        {
                // padding rows (row n is filled with Y^T by large_build_S)
                for (int c = threadIdx.x; c < na; c += 256)
                        grow[c] = (T)0;
                return;
        }
        const T t0 = prow[0], t1 = prow[1], t2 = prow[2];
        if (threadIdx.x < 3)
                grow[threadIdx.x] = prow[threadIdx.x];
        for (int i = threadIdx.x; i < nl; i += 256)
        {
                const T h00 = (T)Hc[4 * i], h01 = (T)Hc[4 * i + 1], h10 = (T)Hc[4 * i + 2], h11 = (T)Hc[4 * i + 3];
                const T ta = prow[3 + 2 * i], tb = prow[4 + 2 * i];
                grow[3 + 2 * i] = h00 * t0 + h01 * t1 - h00 * ta - h01 * tb;
                grow[4 + 2 * i] = h10 * t0 + h11 * t1 - t2 - h10 * ta - h11 * tb;
        }
        for (int c = n + threadIdx.x; c < na; c += 256)
                grow[c] = (T)0;
}

/// S = H G + R (one workgroup per landmark pair / per pose or padding row; lanes over columns).  Also copies Y^T into
/// row n of G.  grid (NP, B): blockIdx.x = output row.
template <typename T> __global__ __launch_bounds__(256) void large_build_S(DevView d, LargeView<T> lv, const int *skipped)
{
        const int b = blockIdx.y, r = blockIdx.x;
        if (skipped[b])
                return;
        const int n = d.n[b], NP = lv.NP;
        const int na = large_blocks(n) * LB;
        if (r >= na)
                return;
        const T *G = lv.G + (size_t)b * NP * NP;
        T *srow = lv.S + ((size_t)b * NP + r) * NP;
        const T rm = (T)(double)KR;
        if (r >= n)
        {
                for (int c = threadIdx.x; c < na; c += 256)
                        srow[c] = (c == r) ? (T)1 : (T)0;
                if (r == n)
                {
                        T *yrow = lv.G + ((size_t)b * NP + n) * NP;
                        const double *Y = lv.Y + (size_t)b * NP;
                        for (int c = threadIdx.x; c < na; c += 256)
                                yrow[c] = (c < n) ? (T)Y[c] : (T)0;
                }
                return;
        }
        if (r < 3)
        {
                for (int c = threadIdx.x; c < na; c += 256)
                        srow[c] = (c < n) ? G[(size_t)r * NP + c] + (c == r ? rm : (T)0) : (T)0;
                return;
        }
        const int i = (r - 3) >> 1, odd = (r - 3) & 1;
        const double *Hc = lv.Hc + ((size_t)b * (NP / 2) + i) * 4;
        const T ha = (T)Hc[2 * odd], hb = (T)Hc[2 * odd + 1];
        const T *g0 = G, *g1 = G + NP, *g2 = G + 2 * (size_t)NP;
        const T *ga = G + (size_t)(3 + 2 * i) * NP, *gb = ga + NP;
        for (int c = threadIdx.x; c < na; c += 256)
        {
                T v = (T)0;
                if (c < n)
                {
                        v = ha * g0[c] + hb * g1[c];
                        if (odd)
                                v -= g2[c];
                        v = v - ha * ga[c] - hb * gb[c];
                        if (c == r)
                                v += rm;
                }
                srow[c] = v;
        }
}

// ------------------------------------------------------------------------------------------------------------------
/// Cholesky of the 64x64 diagonal block k of S, in place (lower; the strict upper part is zeroed).  grid (B), 256 threads.
template <typename T> __global__ __launch_bounds__(256) void large_potrf_diag(DevView d, LargeView<T> lv, int k, const int *skipped)
{
        __shared__ T A[LB][LB + 1];
        const int b = blockIdx.x;
        if (skipped[b])
                return;
        const int n = d.n[b], NP = lv.NP;
        if (k >= large_blocks(n))
                return;
        T *S = lv.S + (size_t)b * NP * NP + (size_t)k * LB * NP + k * LB;
        const int tid = threadIdx.x;
        for (int idx = tid; idx < LB * LB; idx += 256)
                A[idx >> 6][idx & 63] = S[(size_t)(idx >> 6) * NP + (idx & 63)];
        __syncthreads();
        bool bad = false;
        for (int j = 0; j < LB; ++j)
        {
                const T djj = A[j][j];
                if (!(djj > (T)0))
                        bad = true;
                const T rj = (T)1 / sqrt(djj);
                // rank-1 update of the active sub-block (every thread scales the two column entries it needs itself: one
                // barrier per column); 16 x 16 thread grid striding over rows / columns j+1 .. 63
                {
                        const int ti = tid >> 4, tj = tid & 15;
                        for (int i = j + 1 + ti; i < LB; i += 16)
                        {
                                const T li = A[i][j] * rj;
                                for (int c = j + 1 + tj; c <= i; c += 16)
                                        A[i][c] -= li * (A[c][j] * rj);
                        }
                }
                __syncthreads();
                if (tid < LB && tid > j)
                        A[tid][j] *= rj;
                if (tid == 0)
                        A[j][j] = djj * rj;
                __syncthreads();
        }
        for (int idx = tid; idx < LB * LB; idx += 256)
        {
                const int i = idx >> 6, c = idx & 63;
                S[(size_t)i * NP + c] = (c <= i) ? A[i][c] : (T)0;
        }
        if (bad && tid == 0)
                atomicOr(&d.status[b], 4u); // ASLAM_ST_NOT_PD
}

/// virtual stacked matrix M = [S; G] (2*na rows): row pointer of virtual row vr
template <typename T> __device__ __forceinline__ T *stacked_row(const LargeView<T> &lv, int b, int na, int vr)
{
        const size_t NP = lv.NP;
        return (vr < na) ? lv.S + ((size_t)b * NP + vr) * NP : lv.G + ((size_t)b * NP + (vr - na)) * NP;
}

/// Panel k: every row below the diagonal block in S and every active row of G gets X <- X L_kk^-T on its 64 entries
/// of block column k (one thread per row, L_kk broadcast from LDS).  grid (ceil(2*NP/256), B), 256 threads.
template <typename T> __global__ __launch_bounds__(256) void large_panel_solve(DevView d, LargeView<T> lv, int k, const int *skipped)
{
        __shared__ T Lk[LB][LB + 1];
        __shared__ T inv[LB];
        const int b = blockIdx.y;
        if (skipped[b])
                return;
        const int n = d.n[b], NP = lv.NP;
        const int nb = large_blocks(n), na = nb * LB;
        if (k >= nb)
                return;
        const int first = (k + 1) * LB;       // first virtual row of the panel
        const int rows = 2 * na - first;      // S rows below the block, then all of G
        if ((int)blockIdx.x * 256 >= rows)
                return;
        const T *Sd = lv.S + (size_t)b * NP * NP + (size_t)k * LB * NP + k * LB;
        for (int idx = threadIdx.x; idx < LB * LB; idx += 256)
                Lk[idx >> 6][idx & 63] = Sd[(size_t)(idx >> 6) * NP + (idx & 63)];
        __syncthreads();
        if (threadIdx.x < LB)
                inv[threadIdx.x] = (T)1 / Lk[threadIdx.x][threadIdx.x];
        __syncthreads();
        const int lr = blockIdx.x * 256 + threadIdx.x;
        if (lr >= rows)
                return;
        T *x = stacked_row(lv, b, na, first + lr) + k * LB;
        T v[LB];
#pragma unroll
        for (int j = 0; j < LB; ++j)
                v[j] = x[j];
#pragma unroll
        for (int j = 0; j < LB; ++j)
        {
                // four independent partial sums: the dependent-FMA latency, not the issue rate, bounds a single chain
                T a0 = v[j], a1 = (T)0, a2 = (T)0, a3 = (T)0;
#pragma unroll
                for (int c = 0; c < j; c += 4)
                {
                        a0 -= v[c] * Lk[j][c];
                        if (c + 1 < j)
                                a1 -= v[c + 1] * Lk[j][c + 1];
                        if (c + 2 < j)
                                a2 -= v[c + 2] * Lk[j][c + 2];
                        if (c + 3 < j)
                                a3 -= v[c + 3] * Lk[j][c + 3];
                }
                v[j] = ((a0 + a1) + (a2 + a3)) * inv[j];
        }
#pragma unroll
        for (int j = 0; j < LB; ++j)
                x[j] = v[j];
}

// ------------------------------------------------------------------------------------------------------------------
/// C(tile r, tile j) -= A(tile r, kc..) B(tile j, kc..)^T over kblocks 64-wide column blocks starting at k0.
///   MODE 0 (left-looking update of block column k0 before it is factored): C(r, k0) -= sum_{k < k0} M(r, k) S(k0, k)^T for
///            every stacked row tile r >= k0 (S part from the diagonal block down, then all of G); K = 64 k0, so every
///            tile of the matrix is read-modify-written once per factorisation instead of once per earlier block column.
///   MODE 1 (P -= V V^T): A = B = G, C = P, all active tiles, kblocks = active blocks.
/// grid (row tiles, column tiles, B), 256 threads = 4 waves, each wave a 32x32 quadrant of the 64x64 tile as 2x2 MFMA
/// 16x16 tiles; operands staged through LDS 16 columns at a time.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void large_gemm_nt(DevView d, LargeView<T> lv, int k0, const int *skipped)
{
        typedef Mfma<T> MM;
        constexpr int KC = (sizeof(T) == 4) ? 64 : 32; // columns staged per round: a whole K = 64 slab in fp32
        __shared__ T As[LB][KC + 4]; // row stride = 4 mod 64 banks: the 16 rows x 4 k-values of an MFMA operand read hit 64 different banks
        __shared__ T Bs[LB][KC + 4];
        const int b = blockIdx.z;
        if (skipped[b])
                return;
        const int n = d.n[b], NP = lv.NP;
        const int nb = large_blocks(n), na = nb * LB;
        int rt, jt, kblocks;
        if (MODE == 0)
        {
                if (k0 >= nb)
                        return;
                jt = k0;
                rt = k0 + blockIdx.x; // virtual row tile: S rows from the diagonal block down, then all of G
                if (rt >= 2 * nb)
                        return;
                kblocks = k0;
                k0 = 0;
        }
        else
        {
                rt = blockIdx.x;
                jt = blockIdx.y;
                if (rt >= nb || jt > rt)
                        return; // V V^T is symmetric: lower tiles only, mirrored on store
                kblocks = nb;
                k0 = 0;
        }
        const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
        const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;
        // source row pointers
        const T *Arow0, *Brow0;
        T *Crow0;
        if (MODE == 0)
        {
                Arow0 = stacked_row(lv, b, na, rt * LB);
                Brow0 = lv.S + ((size_t)b * NP + (size_t)jt * LB) * NP;
                Crow0 = stacked_row(lv, b, na, rt * LB);
        }
        else
        {
                Arow0 = lv.G + ((size_t)b * NP + (size_t)rt * LB) * NP;
                Brow0 = lv.G + ((size_t)b * NP + (size_t)jt * LB) * NP;
                Crow0 = lv.P + ((size_t)b * NP + (size_t)rt * LB) * NP;
        }
        typename MM::acc_t acc[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v)
                        acc[u][v] = MM::zero();
        // stage: 64 rows x KC columns; a thread moves KC/4 consecutive columns of one row (all loads issued before use)
        const int lrow = tid >> 2, lc0 = (tid & 3) * (KC / 4);
        for (int kc = 0; kc < kblocks * LB; kc += KC)
        {
                const int col = k0 * LB + kc + lc0;
                T ta[KC / 4], tb[KC / 4];
#pragma unroll
                for (int q = 0; q < KC / 4; ++q)
                {
                        ta[q] = Arow0[(size_t)lrow * NP + col + q];
                        tb[q] = Brow0[(size_t)lrow * NP + col + q];
                }
#pragma unroll
                for (int q = 0; q < KC / 4; ++q)
                {
                        As[lrow][lc0 + q] = ta[q];
                        Bs[lrow][lc0 + q] = tb[q];
                }
                __syncthreads();
#pragma unroll
                for (int s = 0; s < KC / 4; ++s)
                {
                        T av[2], bv[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                        {
                                av[u] = As[wr + 16 * u + li][lg + 4 * s];
                                bv[u] = Bs[wc + 16 * u + li][lg + 4 * s];
                        }
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                                for (int v = 0; v < 2; ++v)
                                        acc[u][v] = MM::mma(av[u], bv[v], acc[u][v]);
                }
                __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 2; ++v)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                        {
                                const int row = wr + 16 * u + MM::row(lane, r), colc = jt * LB + wc + 16 * v + li;
                                if (MODE == 0)
                                        Crow0[(size_t)row * NP + colc] -= acc[u][v][r];
                                else if (rt * LB + row < n && colc < n) // keep P's padding clean (row n of G is Y^T, not V)
                                {
                                        Crow0[(size_t)row * NP + colc] -= acc[u][v][r];
                                        if (jt < rt) // mirror into the upper triangle
                                                lv.P[((size_t)b * NP + colc) * NP + rt * LB + row] -= acc[u][v][r];
                                }
                        }
}

/// P -= V V^T (ekf.cpp:311 in the A-form, V = G after the panel solves) on 128x128 tiles: the kernel is bound by the
/// operand traffic out of L2 (every row tile of V is read once per tile column), so the tile is as large as the
/// accumulators allow: 4 waves, each a 64x64 quadrant = 4x4 MFMA 16x16 tiles (64 accumulator registers in fp32).  Lower
/// tiles only, mirrored on store.  The next K slab is fetched into registers while the current one is multiplied.
/// grid (8 * lower tiles * ceil(B/8)), 256 threads.
template <typename T>
__global__ __launch_bounds__(256) void large_syrk(DevView d, LargeView<T> lv, int nfilters, const int *skipped)
{
        typedef Mfma<T> MM;
        constexpr int TB = 128;
        constexpr int KC = (sizeof(T) == 4) ? 32 : 16;
        constexpr int LDS_LD = (sizeof(T) == 4) ? KC + 4 : KC + 2; // fp32: stride = 4 mod 64 banks -> conflict-free operand reads
        __shared__ T As[TB][LDS_LD];
        __shared__ T Bs[TB][LDS_LD];
        // XCD-aware mapping: consecutive workgroup ids go round-robin to the 8 XCDs, each with its own L2.  All lower tiles of
        // one filter are given ids that are equal mod 8, so the ~45 workgroups that share a filter's V run on one XCD and its
        // row tiles are fetched into that L2 once instead of into all eight.
        const int ntile = (lv.NP + TB - 1) / TB, nlow = ntile * (ntile + 1) / 2;
        const int slot = blockIdx.x >> 3;
        const int b = (slot / nlow) * 8 + (blockIdx.x & 7);
        if (b >= nfilters || skipped[b])
                return;
        const int n = d.n[b], NP = lv.NP;
        const int na = large_blocks(n) * LB;
        const int tl = slot % nlow;
        int rt = (int)((sqrtf(8.0f * (float)tl + 1.0f) - 1.0f) * 0.5f);
        while ((rt + 1) * (rt + 2) / 2 <= tl)
                ++rt;
        while (rt * (rt + 1) / 2 > tl)
                --rt;
        const int jt = tl - rt * (rt + 1) / 2;
        if (rt * TB >= na)
                return;
        const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
        const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
        const T *G = lv.G + (size_t)b * NP * NP;
        T *P = lv.P + (size_t)b * NP * NP;
        // staging: 8 lanes cover one 128-byte row segment (KC elements) with 16-byte loads, so every wave-level load
        // instruction fetches 8 whole cache lines; 4 passes of 32 rows
        typedef T vec_t __attribute__((ext_vector_type(16 / sizeof(T))));
        constexpr int VW = 16 / sizeof(T);
        static_assert(KC / VW == 8, "8 lanes per staged row");
        const int lrow = tid >> 3, lc0 = (tid & 7) * VW;
        // the last tile row may hang over the allocation: such rows read row na-1 instead (their products are never stored)
        const T *Ap[4], *Bp[4];
#pragma unroll
        for (int q = 0; q < 4; ++q)
        {
                Ap[q] = G + (size_t)min(rt * TB + lrow + 32 * q, na - 1) * NP + lc0;
                Bp[q] = G + (size_t)min(jt * TB + lrow + 32 * q, na - 1) * NP + lc0;
        }
        typename MM::acc_t acc[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                        acc[u][v] = MM::zero();
        vec_t ta[4], tb[4];
        auto fetch = [&](int kc) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                {
                        ta[q] = *reinterpret_cast<const vec_t *>(Ap[q] + kc);
                        tb[q] = *reinterpret_cast<const vec_t *>(Bp[q] + kc);
                }
        };
        // 16-row / 16-column subtiles of this wave's quadrant that hold at least one of the n valid rows / columns (n = 3 mod 128
        // at full size: the last tile row is almost empty), and the upper quadrant of a diagonal tile is the mirror image of
        // its lower one: neither is multiplied
        const bool idle = (rt == jt && wc > wr);
        const int nu = idle ? 0 : __builtin_amdgcn_readfirstlane(max(0, min(4, (n - (rt * TB + wr) + 15) >> 4)));
        const int nv = idle ? 0 : __builtin_amdgcn_readfirstlane(max(0, min(4, (n - (jt * TB + wc) + 15) >> 4)));
        const bool full = (nu == 4 && nv == 4);
        fetch(0);
        for (int kc = 0; kc < na; kc += KC)
        {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                {
                        *reinterpret_cast<vec_t *>(&As[lrow + 32 * q][lc0]) = ta[q];
                        *reinterpret_cast<vec_t *>(&Bs[lrow + 32 * q][lc0]) = tb[q];
                }
                __syncthreads();
                if (kc + KC < na)
                        fetch(kc + KC);
                if (full)
                {
#pragma unroll
                        for (int s = 0; s < KC / 4; ++s)
                        {
                                T av[4], bv[4];
#pragma unroll
                                for (int u = 0; u < 4; ++u)
                                {
                                        av[u] = As[wr + 16 * u + li][lg + 4 * s];
                                        bv[u] = Bs[wc + 16 * u + li][lg + 4 * s];
                                }
#pragma unroll
                                for (int u = 0; u < 4; ++u)
#pragma unroll
                                        for (int v = 0; v < 4; ++v)
                                                acc[u][v] = MM::mma(av[u], bv[v], acc[u][v]);
                        }
                }
                else
                {
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                                if (u < nu)
#pragma unroll
                                        for (int v = 0; v < 4; ++v)
                                                if (v < nv)
#pragma unroll
                                                        for (int s = 0; s < KC / 4; ++s)
                                                                acc[u][v] = MM::mma(As[wr + 16 * u + li][lg + 4 * s], Bs[wc + 16 * v + li][lg + 4 * s],
                                                                                    acc[u][v]);
                }
                __syncthreads();
        }
        if (idle)
                return;
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                {
                        const int col = jt * TB + wc + 16 * v + li;
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                        {
                                const int row = rt * TB + wr + 16 * u + MM::row(lane, r);
                                if (row < n && col < n) // keep P's padding clean (row n of G is Y^T, not V)
                                        P[(size_t)row * NP + col] -= acc[u][v][r];
                        }
                        if ((jt < rt || wc < wr) && col < n)
                        {
                                // mirror into the upper triangle.  fp32: a lane's four registers are four consecutive rows of
                                // the tile = 16 contiguous bytes of the mirrored row: one 16-byte read-modify-write
                                const int row0 = rt * TB + wr + 16 * u + MM::row(lane, 0);
                                T *m = P + (size_t)col * NP + row0;
                                if (sizeof(T) == 4 && row0 + 3 < n)
                                {
                                        typedef T vec4_t __attribute__((ext_vector_type(4)));
                                        vec4_t x = *reinterpret_cast<vec4_t *>(m);
#pragma unroll
                                        for (int r = 0; r < 4; ++r)
                                                x[r] -= acc[u][v][r];
                                        *reinterpret_cast<vec4_t *>(m) = x;
                                }
                                else
                                {
#pragma unroll
                                        for (int r = 0; r < 4; ++r)
                                        {
                                                const int row = rt * TB + wr + 16 * u + MM::row(lane, r);
                                                if (row < n)
                                                        P[(size_t)col * NP + row] -= acc[u][v][r];
                                        }
                                }
                        }
                }
}

/// X <- X + V q with q = row n of G = (L^-1 Y)^T; one wave per state row.  grid (ceil(NP/4), B), 256 threads.  In replay
/// mode also writes the pose of this callback.
template <typename T, int MODE>
__global__ __launch_bounds__(256) void large_x_update(DevView d, LargeView<T> lv, int s, int nsteps, double *poses_out,
                                                      int32_t *dims_out, const int *skipped)
{
        const int b = blockIdx.y;
        if (skipped[b])
                return;
        const int n = d.n[b], NP = lv.NP;
        const int a = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (a >= n)
                return;
        const T *vrow = lv.G + ((size_t)b * NP + a) * NP;
        const T *q = lv.G + ((size_t)b * NP + n) * NP;
        double acc = 0.0;
        for (int j = lane; j < n; j += 64)
                acc += (double)vrow[j] * (double)q[j];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1)
                acc += __shfl_xor(acc, off);
        if (lane == 0)
        {
                const double xa = d.X[(size_t)b * NP + a] + acc;
                d.X[(size_t)b * NP + a] = xa;
                if (MODE == MODE_REPLAY)
                {
                        if (a < 3 && poses_out)
                                poses_out[((size_t)b * nsteps + s) * 3 + a] = xa;
                        if (a == 0 && dims_out)
                                dims_out[(size_t)b * nsteps + s] = n;
                }
        }
}
} // namespace aslam
