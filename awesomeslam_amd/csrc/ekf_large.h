// ekf_large.h -- EKF-SLAM for state dimensions beyond one CU (144 < n <= 1087; BASELINE configs[3]/[4]: 512 landmarks,
// n = 1027), fp64 or fp32.  One callback = a short chain of launches that use the whole GPU for one filter (and
// `batch` filters side by side through blockIdx.y/z):
//
//   large_frontend   cbSensorLandmark + updateZandA (ekf.cpp:102-213, shared small_frontend code), predict
//                    (X <- f(X), P <- A P A^T + Q with A = I + 2 entries, ekf.cpp:295-297), updateH coefficients
//                    (ekf.cpp:117-134), Y = Z - h(X) (ekf.cpp:302-307)                               1 workgroup / filter
//   large_build_GS   G = P H^T (H has <= 5 non-zeros per row, ekf.cpp:301) and S = H G + R (ekf.cpp:300) in one pass
//                    over P; Y^T appended to G as row n
//   17 x { potrf_inv, update_panel }   blocked Cholesky S = L L^T of the STACKED matrix [S; G; Y^T], 64-wide block
//                    columns: a one-wave in-register factorisation of the diagonal block that also yields its inverse,
//                    then ONE MFMA kernel per block column that brings the column up to date with all earlier ones
//                    (left-looking, K = 64 k), eliminates it (X <- X Linv^T, a 64-deep product instead of a triangular
//                    solve) and updates the diagonal blocks below (right-looking).  The same launches that factor S
//                    turn G into V = G L^-T and Y^T into (L^-1 Y)^T, so K = P H^T S^-1 = V L^-1 never needs a
//                    separate triangular solve
//   large_syrk       P <- P - V V^T        ( = (I - K H) P, ekf.cpp:310, since K H P = V V^T for symmetric P ),
//                    128x128 tiles, lower half mirrored, one filter's tiles on one XCD
//   large_x_update   X <- X + V (L^-1 Y)   ( = X + K Y, ekf.cpp:309 )
//
// Unlike the single-CU kernel this path does not go through measurement coordinates: in fp32 the H / H^-1 change of
// basis would cost eps * cond(H)^2 per callback; G, S, V are formed directly (2.33 n^3 flops all the same).
// It does use the symmetry of P (P H^T = (H P)^T, K H P = V V^T); the reference's P is symmetric up to rounding.
// Matrices are row-major in HBM with row stride NP (multiple of 64, zero padding); GEMMs run on the 16x16x4 MFMA
// (f64 or f32 inputs), operands staged through LDS.
#pragma once

#include <type_traits>

#include "small_common.h"

namespace aslam
{
constexpr int LB = 64;               // block size of the factorisation = GEMM tile edge
constexpr int LARGE_OBS_CAP = 1024;  // stored sensor message (LDS)
constexpr int LARGE_WAIT_CAP = 2048; // wait-list (LDS)
constexpr int LARGE_NP_MAX = 1088;   // 17 blocks of 64: n <= 1087
constexpr int LARGE_NB_MAX = LARGE_NP_MAX / LB;

template <typename T> struct LargeView
{
        int NP;     // row stride, multiple of LB
        double *P;  // [B][NP][NP]  always binary64 (fp32 mode: only G, S, L, V and the MFMA products are binary32)
        T *G;       // [B][NP][NP]  P H^T, then V; row n carries Y^T, then (L^-1 Y)^T
        T *S;       // [B][NP][NP]  innovation covariance, then L (lower)
        double *Hc; // [B][NP/2][4] h00 h01 h10 h11 per landmark
        double *Y;  // [B][NP]
        T *Linv;    // [B][LARGE_NB_MAX][LB][LB]  inverses of the diagonal blocks of L
        T *Vw;      // [B][NP][NP]  binary32 mode, few-filter chain (large_right_step): V is written HERE, not over G (every block of G is read by many workgroups of a launch); else nullptr
        unsigned short *Lpl; // binary32 mode with the bf16-pipe TRSM: LPlanes::base (bf16 planes of L and of the inverses, written by large_chol_resident), else nullptr
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

/// four floats -> their three bf16 planes, packed two to a register: x = h + m + l up to 2^-25 |x|.  Round to nearest at every level
/// (v_cvt_pk_bf16_f32): with truncated pieces, which all carry the sign of x, the dropped terms of a split product have the sign of the product
/// (large_syrk_bf16x3).
typedef unsigned u2x __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_bf16x3(const f4 &x, u2x &h, u2x &m, u2x &l)
{
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int e = 0; e < 2; ++e)
        {
                const f2 v = {x[2 * e], x[2 * e + 1]};
                const bf2 hh = __builtin_convertvector(v, bf2);
                const f2 r1 = v - __builtin_convertvector(hh, f2);
                const bf2 mm = __builtin_convertvector(r1, bf2);
                const f2 r2 = r1 - __builtin_convertvector(mm, f2);
                const bf2 ll = __builtin_convertvector(r2, bf2);
                h[e] = __builtin_bit_cast(unsigned, hh), m[e] = __builtin_bit_cast(unsigned, mm), l[e] = __builtin_bit_cast(unsigned, ll);
        }
}

/// Planes of L in HBM (binary32 mode; written by the resident Cholesky kernels, streamed by large_trsm_bf16 / large_chol_bf16 --
/// ekf_large_trsm16.h), per filter: Lq [3][NP][NP] bf16, row-major, columns permuted inside every 64-block.  Block (k, j), j < k, holds L(k, j);
/// the DIAGONAL block (k, k) holds the INVERSE of L(k, k) -- the sweeps multiply by it and never read L(k, k) itself -- so that every block of a
/// sweep's sequence  Linv_0; L(1,0), Linv_1; L(2,0), L(2,1), Linv_2; ...  has the same row / plane strides and its address is one multiply-add
/// (the round's first layout kept the inverses behind the planes: 40 scalar instructions and three branches per block to pick the layout).
struct LPlanes
{
        unsigned short *base;
        __host__ __device__ static size_t per_filter(int NP) { return (size_t)3 * NP * NP; }
        __host__ __device__ unsigned short *Lq(int b, int NP) const { return base + (size_t)b * per_filter(NP); }
        /// plane 0 of block (k, j) (row stride NP, plane stride NP * NP)
        __host__ __device__ unsigned short *block(int b, int NP, int k, int j) const { return Lq(b, NP) + (size_t)(64 * k) * NP + 64 * j; }
};
/// position of column c (0 .. 63) of a block inside a permuted plane row: 32 h + 8 g + 4 w + r holds column 32 h + 16 w + 4 g + r
__host__ __device__ __forceinline__ int lplane_pos(int c)
{
        return 32 * (c >> 5) + 8 * ((c >> 2) & 3) + 4 * ((c >> 4) & 1) + (c & 3);
}

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
                L.sTv = nullptr;
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
        const int b = (MODE == MODE_STEP && sa.traj >= 0) ? sa.traj : (int)blockIdx.x; // sa.traj < 0: the batched step, one workgroup per filter
        double *Pg = lv.P + (size_t)b * NP * NP;
        double *Hc = lv.Hc + (size_t)b * (NP / 2) * 4;
        double *Yg = lv.Y + (size_t)b * NP;

#ifdef ASLAM_STAMPS
        unsigned long long stamp_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
        const unsigned long long stamp_rt0 = __builtin_amdgcn_s_memrealtime(); // 100 MHz: wall time inside the kernel
#endif
        small_load<MODE>(d, L, b, tid, NP);
        ASLAM_STAMP(0);
        if (MODE == MODE_REPLAY)
        {
                if (small_frontend<true, LARGE_OBS_CAP, LARGE_WAIT_CAP, LARGE_NP_MAX / 2, double>(d, L, Pg, NP, b, t, s, nsteps, poses_out,
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
                        // one filter: the arguments of the call; batched step: this filter's entries of the per-call arrays
                        sm.vx = sa.traj >= 0 ? sa.vx : d.step_in[b];
                        sm.az = sa.traj >= 0 ? sa.az : d.step_in[d.B + b];
                        sm.dt = sa.traj >= 0 ? sa.dt : d.step_in[2 * d.B + b];
                }
                __syncthreads();
        }
        ASLAM_STAMP(1);
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
        ASLAM_STAMP(2);
        // P <- A P A^T + Q (ekf.cpp:297), A = I except A(0,0), A(1,0): rows 0,1 then columns 0,1
        {
                const double a00 = sm.a00, a10 = sm.a10, q = (double)KQ;
                for (int c = tid; c < n; c += SMALL_WG)
                {
                        const double r0 = Pg[c];
                        Pg[c] = a00 * r0;
                        Pg[NP + c] = a10 * r0 + Pg[NP + c];
                }
                __syncthreads();
                for (int r = tid; r < n; r += SMALL_WG)
                {
                        double *row = Pg + (size_t)r * NP;
                        const double c0 = row[0];
                        double v0 = a00 * c0, v1 = a10 * c0 + row[1];
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
        ASLAM_STAMP(3);
        small_store<MODE>(d, L, b, tid, NP);
        ASLAM_STAMP(4);
#ifdef ASLAM_STAMPS
        if (tid == 0 && blockIdx.x == 0)
        {
                for (int i = 0; i < 12; ++i)
                        d.dbg[i] += stamp_acc[i];
                d.dbg[56] += __builtin_amdgcn_s_memrealtime() - stamp_rt0;
                d.dbg[57] += 1;
        }
        if (tid == 0 && blockIdx.x < 1024)
                d.dbg[64 + blockIdx.x] = __builtin_amdgcn_s_memrealtime() - stamp_rt0; // every workgroup of the last launch (single stream group: the views are not shifted)
#endif
}

// ------------------------------------------------------------------------------------------------------------------
/// G = P H^T (ekf.cpp:301; H has <= 5 non-zeros per row) and S = H G + R (ekf.cpp:300) in one pass over P: a workgroup
/// owns two consecutive rows (a landmark pair, or two padding rows; workgroup 0 the three pose rows), forms their G entries
/// from the P rows it reads, re-forms the three pose rows of G it needs for H G (their P rows stay in L2), and writes G and S
/// without G ever being read back.  Also copies Y^T into row n of G.  Threads run over landmark column pairs.
/// P is binary64 in both modes and the products are formed in binary64; G and S are rounded to T on the way out (in fp32 mode
/// that is the one rounding of G).  Only the lower block triangle of S (with full diagonal blocks) is consumed downstream and written.  grid (1 + ceil((NP / 2) / GS_ROW_PAIRS), B), 256 threads: workgroup 0 the pose rows, the others GS_ROW_PAIRS landmark row pairs each.
constexpr int GS_ROW_PAIRS = 2; // 4: slower (29.1 k against 29.7 k filter-steps/s on one stream), 1: 28.5 k

template <typename T> __global__ __launch_bounds__(256) void large_build_GS(DevView d, LargeView<T> lv, const int *skipped)
{
        const int b = blockIdx.y;
        if (skipped[b])
                return;
        const int n = d.n[b], NP = lv.NP;
        const int na = large_blocks(n) * LB;
        const int nl = (n - 3) / 2;
        const double *P = lv.P + (size_t)b * NP * NP;
        T *G = lv.G + (size_t)b * NP * NP;
        T *S = lv.S + (size_t)b * NP * NP;
        const double *Hc = lv.Hc + (size_t)b * (NP / 2) * 4;
        const double rm = (double)KR;
        const int tid = threadIdx.x;
        const bool pose = (blockIdx.x == 0);
        constexpr int RP = GS_ROW_PAIRS; // landmark row pairs per workgroup: the pose rows of G and the H coefficients of a column pair are formed once for all of them
        const int r0 = pose ? 0 : 3 + 2 * RP * ((int)blockIdx.x - 1); // first row of this workgroup
        if (r0 >= na)
                return;
        int nlive = 0; // leading row pairs that are landmarks (rows < n; n is odd, so a pair never straddles n)
        if (!pose)
        {
                for (int rp = 0; rp < RP; ++rp)
                {
                        const int ra = r0 + 2 * rp;
                        if (ra >= na)
                                break;
                        if (ra < n)
                        {
                                nlive = rp + 1;
                                continue;
                        }
                        // padding rows ra, ra+1: G = 0 (row n: Y^T), S = identity
                        for (int rr = ra; rr < min(ra + 2, na); ++rr)
                        {
                                T *grow = G + (size_t)rr * NP, *srow = S + (size_t)rr * NP;
                                const double *Y = lv.Y + (size_t)b * NP;
                                for (int c = tid; c < na; c += 256)
                                {
                                        grow[c] = (rr == n && c < n) ? (T)Y[c] : (T)0;
                                        srow[c] = (c == rr) ? (T)1 : (T)0;
                                }
                        }
                }
                if (nlive == 0)
                        return;
        }
        // G(a, :) for one row a of P: columns 0..2 copy P, landmark column pair j mixes (t0, t1, t2, ta, tb) with H_j
        const double *p0 = P, *p1 = P + NP, *p2 = P + 2 * (size_t)NP;
        const double t00 = p0[0], t01 = p0[1], t02 = p0[2], t10 = p1[0], t11 = p1[1], t12 = p1[2], t20 = p2[0], t21 = p2[1], t22 = p2[2];
        const double *pa[RP], *pb[RP]; // landmark rows (unused by the pose workgroup)
        double ta0[RP], ta1[RP], ta2[RP], tb0[RP], tb1[RP], tb2[RP], ha0[RP], ha1[RP], hb0[RP], hb1[RP];
#pragma unroll
        for (int rp = 0; rp < RP; ++rp)
        {
                const int ra = min(r0 + 2 * rp, n - 2); // (pairs beyond nlive are never used: keep their addresses valid)
                pa[rp] = P + (size_t)ra * NP, pb[rp] = pa[rp] + NP;
                ta0[rp] = ta1[rp] = ta2[rp] = tb0[rp] = tb1[rp] = tb2[rp] = ha0[rp] = ha1[rp] = hb0[rp] = hb1[rp] = 0.0;
                if (!pose && rp < nlive)
                {
                        ta0[rp] = pa[rp][0], ta1[rp] = pa[rp][1], ta2[rp] = pa[rp][2];
                        tb0[rp] = pb[rp][0], tb1[rp] = pb[rp][1], tb2[rp] = pb[rp][2];
                        const double *hr = Hc + 4 * ((ra - 3) >> 1);
                        ha0[rp] = hr[0], ha1[rp] = hr[1], hb0[rp] = hr[2], hb1[rp] = hr[3]; // H rows ra (range) and ra+1 (bearing)
                }
        }
        auto grow = [](double h00, double h01, double h10, double h11, double t0, double t1, double t2, double ta, double tb, double &ge,
                       double &go) {
                ge = h00 * t0 + h01 * t1 - h00 * ta - h01 * tb;
                go = h10 * t0 + h11 * t1 - t2 - h10 * ta - h11 * tb;
        };
        // S(r, c) = (H G)(r, c) for landmark row r = ra (+1): ha g0 + hb g1 [- g2] - ha ga - hb gb
        auto srow = [](double ha, double hb, bool odd, double g0, double g1, double g2, double ga, double gb) -> double {
                double v = ha * g0 + hb * g1;
                if (odd)
                        v -= g2;
                return v - ha * ga - hb * gb;
        };
        // ---- pose columns 0..2 (G = P there)
        if (tid < 3)
        {
                const int c = tid;
                const double g0 = p0[c], g1 = p1[c], g2 = p2[c];
                if (pose)
                {
                        G[c] = (T)g0, G[NP + c] = (T)g1, G[2 * (size_t)NP + c] = (T)g2;
                        S[c] = (T)(g0 + (c == 0 ? rm : 0.0));
                        S[NP + c] = (T)(g1 + (c == 1 ? rm : 0.0));
                        S[2 * (size_t)NP + c] = (T)(g2 + (c == 2 ? rm : 0.0));
                }
                else
                {
#pragma unroll
                        for (int rp = 0; rp < RP; ++rp)
                                if (rp < nlive)
                                {
                                        const int ra = r0 + 2 * rp;
                                        const double ga = pa[rp][c], gb = pb[rp][c];
                                        G[(size_t)ra * NP + c] = (T)ga;
                                        G[(size_t)(ra + 1) * NP + c] = (T)gb;
                                        S[(size_t)ra * NP + c] = (T)srow(ha0[rp], ha1[rp], false, g0, g1, g2, ga, gb);
                                        S[(size_t)(ra + 1) * NP + c] = (T)srow(hb0[rp], hb1[rp], true, g0, g1, g2, ga, gb);
                                }
                }
        }
        // Only the lower block triangle of S and its 64x64 diagonal blocks are consumed downstream (the Cholesky kernels): a row of S
        // stops at the end of the diagonal block of its pair's last row -- 2.1 of the 4.7 MB of S per filter are never written.
        // ---- landmark column pairs
        for (int j = tid; j < nl; j += 256)
        {
                const int ce = 3 + 2 * j, co = ce + 1;
                const double h00 = Hc[4 * j], h01 = Hc[4 * j + 1], h10 = Hc[4 * j + 2], h11 = Hc[4 * j + 3];
                double g0e, g0o, g1e, g1o, g2e, g2o;
                grow(h00, h01, h10, h11, t00, t01, t02, p0[ce], p0[co], g0e, g0o);
                grow(h00, h01, h10, h11, t10, t11, t12, p1[ce], p1[co], g1e, g1o);
                grow(h00, h01, h10, h11, t20, t21, t22, p2[ce], p2[co], g2e, g2o);
                if (pose)
                {
                        G[ce] = (T)g0e, G[co] = (T)g0o;
                        G[NP + ce] = (T)g1e, G[NP + co] = (T)g1o;
                        G[2 * (size_t)NP + ce] = (T)g2e, G[2 * (size_t)NP + co] = (T)g2o;
                        if (ce < LB)
                        {
                                S[ce] = (T)g0e, S[co] = (T)g0o;
                                S[NP + ce] = (T)g1e, S[NP + co] = (T)g1o;
                                S[2 * (size_t)NP + ce] = (T)g2e, S[2 * (size_t)NP + co] = (T)g2o;
                        }
                }
                else
                {
#pragma unroll
                        for (int rp = 0; rp < RP; ++rp)
                                if (rp < nlive)
                                {
                                        const int ra = r0 + 2 * rp;
                                        double gae, gao, gbe, gbo;
                                        grow(h00, h01, h10, h11, ta0[rp], ta1[rp], ta2[rp], pa[rp][ce], pa[rp][co], gae, gao);
                                        grow(h00, h01, h10, h11, tb0[rp], tb1[rp], tb2[rp], pb[rp][ce], pb[rp][co], gbe, gbo);
                                        T *ga = G + (size_t)ra * NP, *gb = ga + NP, *sa = S + (size_t)ra * NP, *sb = sa + NP;
                                        ga[ce] = (T)gae, ga[co] = (T)gao;
                                        gb[ce] = (T)gbe, gb[co] = (T)gbo;
                                        if (ce < LB * ((ra + 1) / LB + 1))
                                        {
                                                double se = srow(ha0[rp], ha1[rp], false, g0e, g1e, g2e, gae, gbe),
                                                       so = srow(ha0[rp], ha1[rp], false, g0o, g1o, g2o, gao, gbo);
                                                double ue = srow(hb0[rp], hb1[rp], true, g0e, g1e, g2e, gae, gbe),
                                                       uo = srow(hb0[rp], hb1[rp], true, g0o, g1o, g2o, gao, gbo);
                                                if (ce == ra)
                                                        se += rm; // R on the diagonal
                                                if (co == ra + 1)
                                                        uo += rm;
                                                sa[ce] = (T)se, sa[co] = (T)so;
                                                sb[ce] = (T)ue, sb[co] = (T)uo;
                                        }
                                }
                }
        }
        // ---- zero padding columns n .. na-1
        const int nrows = pose ? 3 : 2 * nlive;
        for (int idx = tid; idx < nrows * (na - n); idx += 256)
        {
                const int rr = r0 + idx / (na - n), c = n + idx % (na - n);
                G[(size_t)rr * NP + c] = (T)0;
                S[(size_t)rr * NP + c] = (T)0;
        }
}

// ------------------------------------------------------------------------------------------------------------------
/// The same G and S, bit for bit, from the LOWER BLOCK TRIANGLE of P alone (round 4).  P is symmetric (large_syrk* store the mirror image of every
/// entry they compute, the predict writes rows and columns 0, 1 alike), so the 64x64 block (I, J), J <= I, serves three outputs:
///     G(I, J)  = P(I, :) H^T restricted to the columns of J                       (rows of the block)
///     G(J, I)  = P(J, :) H^T restricted to the columns of I, with P(J, I) = P(I, J)^T   (columns of the block, read transposed from LDS)
///     S(I, J)  = H(I, :) G(:, J) + R                                               (needs G(I +- 1 row, J) and G(0..2, J), kept in binary64 in LDS)
/// large_build_GS read all of P (8.4 MB per filter at n = 1027) to write G and the lower block triangle of S; this reads 4.4 MB: the kernel is
/// HBM-bound, 16.6 -> 12.4 MB per filter and callback.  H couples landmark column pair (c_odd, c_even) = (3 + 2 j, 4 + 2 j) only to the pose and to
/// itself (ekf.cpp:117-134), so an entry needs its own 2-column (2-row) pair -- pairs straddle the 64-blocks, hence the one-element halo around the
/// block -- and the pose columns of its row.  Every expression is evaluated in the order large_build_GS uses (no contraction): G and S come out
/// bit-identical, which tests/test_gpu_large.py::test_tiled_GS_is_bit_identical checks through the diagnostic read-back.
/// grid (NB (NB + 1) / 2, B) with NB = NP / 64, 256 threads; 40 KB of LDS (the block with its halo, 66 x 67 doubles, and the small vectors): four
/// workgroups per CU.  A thread forms, for its entry (r, c), G(r, c) AND the G entry of the other row of r's pair (the halo rows make that possible at
/// the block's edges), so S(r, c) needs nothing from other threads; the first version kept G(I +- 1, J) in LDS in binary64 (76 KB, two workgroups per CU)
/// and indexed by division: 2.0 ms per 256 filters against 0.81 - 0.94 ms for large_build_GS (profiles/r04_experiments.md section 3).
struct GsTilesLds
{
        static constexpr int HB = LB + 2, PLD = HB + 1; // block + halo; row stride 67 doubles: the transposed reads are conflict-free
        double Pt[HB][PLD];          // P(64 I - 1 + r, 64 J - 1 + c)
        double tI[HB][4], tJ[LB][4]; // pose columns P(a, 0..2) of the rows of I (with halo) and of J
        double cI[HB][2], cJ[LB][2]; // (u, v) of H for the rows of I (with halo: the partner rows) and the columns of J: (h00, h01) on odd, (h10, h11) on even indices
        double P3[3][HB], G3[3][LB], P33[3][4]; // P(0..2, 64 J - 1 + c), G(0..2, 64 J + c), P(0..2, 0..2)
};
template <typename T> __global__ __launch_bounds__(256) void large_build_GS_tiles(DevView d, LargeView<T> lv, const int *skipped)
{
        constexpr int HB = GsTilesLds::HB;
        __shared__ GsTilesLds sh;
        auto &Pt = sh.Pt;
        auto &tI = sh.tI;
        auto &tJ = sh.tJ;
        auto &cI = sh.cI;
        auto &cJ = sh.cJ;
        auto &P3 = sh.P3;
        auto &G3 = sh.G3;
        auto &P33 = sh.P33;
        const int b = blockIdx.y;
        if (skipped[b])
                return;
        const int n = d.n[b], NP = lv.NP;
        const int nb = large_blocks(n);
        const int tl = blockIdx.x;
        int I = (int)((sqrtf(8.0f * (float)tl + 1.0f) - 1.0f) * 0.5f);
        while ((I + 1) * (I + 2) / 2 <= tl)
                ++I;
        while (I * (I + 1) / 2 > tl)
                --I;
        const int J = tl - I * (I + 1) / 2;
        if (I >= nb)
                return;
        const double *P = lv.P + (size_t)b * NP * NP;
        T *G = lv.G + (size_t)b * NP * NP;
        T *S = lv.S + (size_t)b * NP * NP;
        const double *Hc = lv.Hc + (size_t)b * (NP / 2) * 4;
        const double *Y = lv.Y + (size_t)b * NP;
        const double rm = (double)KR;
        const int tid = threadIdx.x;
        const int r0 = LB * I - 1, c0 = LB * J - 1; // global index of halo row / column 0
        typedef double d2 __attribute__((ext_vector_type(2)));
        // ---- stage the block: 16-byte loads of the 64 aligned columns (32 lanes per row, 8 rows per pass), then the two halo columns
        {
                const int q = tid & 31, rr = tid >> 5;
#pragma unroll
                for (int ps = 0; ps < (HB + 7) / 8; ++ps)
                {
                        const int r = 8 * ps + rr, gr = r0 + r;
                        if (r < HB)
                        {
                                d2 v = {0.0, 0.0};
                                if (gr >= 0 && gr < NP)
                                        v = *reinterpret_cast<const d2 *>(P + (size_t)gr * NP + LB * J + 2 * q);
                                Pt[r][1 + 2 * q] = v[0], Pt[r][2 + 2 * q] = v[1];
                        }
                }
                if (tid < 2 * HB)
                {
                        const int r = tid >> 1, hc = (tid & 1) ? HB - 1 : 0;
                        const int gr = r0 + r, gc = c0 + hc;
                        Pt[r][hc] = (gr >= 0 && gr < NP && gc >= 0 && gc < NP) ? P[(size_t)gr * NP + gc] : 0.0;
                }
        }
        if (tid < HB)
        {
                const int gr = r0 + tid, gc = c0 + tid;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                {
                        tI[tid][k] = (gr >= 0 && gr < NP) ? P[(size_t)gr * NP + k] : 0.0;
                        P3[k][tid] = (gc >= 0 && gc < NP) ? P[(size_t)k * NP + gc] : 0.0;
                }
                // H coefficients of row gr (the rows of I and their partners)
                double u = 0.0, v = 0.0;
                if (gr >= 3 && gr < n)
                {
                        const double *h = Hc + 4 * ((gr - 3) >> 1);
                        u = (gr & 1) ? h[0] : h[2], v = (gr & 1) ? h[1] : h[3];
                }
                cI[tid][0] = u, cI[tid][1] = v;
        }
        else if (tid >= 128 && tid < 128 + LB)
        {
                const int c = tid - 128, gc = LB * J + c;
#pragma unroll
                for (int k = 0; k < 3; ++k)
                        tJ[c][k] = P[(size_t)gc * NP + k];
                double u = 0.0, v = 0.0;
                if (gc >= 3 && gc < n)
                {
                        const double *h = Hc + 4 * ((gc - 3) >> 1);
                        u = (gc & 1) ? h[0] : h[2], v = (gc & 1) ? h[1] : h[3];
                }
                cJ[c][0] = u, cJ[c][1] = v;
        }
        else if (tid >= 192 && tid < 201)
                P33[(tid - 192) / 3][(tid - 192) % 3] = P[(size_t)((tid - 192) / 3) * NP + (tid - 192) % 3];
        __syncthreads();
        // G(a, c) of ekf.cpp:301 for a landmark column c (3 <= c < n) from the pose entries t0, t1, t2 of row a and the row's entries at the pair of c
        auto gcol = [](bool even, double u, double v, double t0, double t1, double t2, double p_odd, double p_even) -> double {
                return even ? u * t0 + v * t1 - t2 - u * p_odd - v * p_even : u * t0 + v * t1 - u * p_odd - v * p_even;
        };
        const int c = tid & (LB - 1), rq = tid >> 6; // this thread's column of the block, its row inside a pass of four
        const int gc = LB * J + c;
        const bool ceven = !(gc & 1);
        const int co = ceven ? c : c + 1; // halo index of the odd member of column gc's pair (the even one: co + 1)
        const double uJ = cJ[c][0], vJ = cJ[c][1];
        // ---- the pose rows of G over the columns of J (binary64; for S)
        if (rq < 3)
        {
                const int m = rq;
                G3[m][c] = gc >= n ? 0.0 : gc < 3 ? P33[m][gc] : gcol(ceven, uJ, vJ, P33[m][0], P33[m][1], P33[m][2], P3[m][co], P3[m][co + 1]);
        }
        __syncthreads();
        // G of halo row index r (global row r0 + r) at this thread's column, binary64
        auto g_at = [&](int r) -> double {
                const int ga = r0 + r;
                if (ga < 0 || ga > n || gc >= n)
                        return 0.0; // padding (rows beyond n, columns from n on)
                if (ga == n)
                        return Y[gc]; // row n carries Y^T through the solve
                if (gc < 3)
                        return Pt[r][c + 1];
                return gcol(ceven, uJ, vJ, tI[r][0], tI[r][1], tI[r][2], Pt[r][co], Pt[r][co + 1]);
        };
        const double g30 = G3[0][c], g31 = G3[1][c], g32 = G3[2][c];
        // ---- G(I, J) and S(I, J) = H(I, :) G(:, J) + R: entry (r, c) with the G entry of the other row of r's pair
#pragma unroll 4
        for (int ps = 0; ps < LB / 4; ++ps)
        {
                const int r = 4 * ps + rq, gr = LB * I + r; // block row r = halo row r + 1
                const double g = g_at(r + 1);
                G[(size_t)gr * NP + gc] = (T)g;
                double sv;
                if (gr >= n)
                        sv = (gr == gc) ? 1.0 : 0.0; // padding rows: identity
                else if (gc >= n)
                        sv = 0.0;
                else if (gr < 3)
                        sv = g + (gr == gc ? rm : 0.0);
                else
                {
                        const bool even = !(gr & 1);                    // the bearing row of its landmark
                        const double gp = g_at(even ? r : r + 2);       // the other row of the pair: gr - 1 / gr + 1
                        const double go_ = even ? gp : g, ge_ = even ? g : gp;
                        const double ha = cI[r + 1][0], hb = cI[r + 1][1];
                        double v = ha * g30 + hb * g31;
                        if (even)
                                v -= g32;
                        sv = v - ha * go_ - hb * ge_;
                        if (gr == gc)
                                sv += rm;
                }
                S[(size_t)gr * NP + gc] = (T)sv;
        }
        // ---- G(J, I): row cr of J, column a of I, from the block read transposed
        if (I != J)
        {
                const int a = c, gcolx = LB * I + a; // this thread's output column (a landmark column: gcolx >= 64)
                const bool even = !(gcolx & 1);
                const int ao = even ? a : a + 1; // halo ROW index of the odd member of column gcolx's pair
                const double uI = cI[a + 1][0], vI = cI[a + 1][1];
#pragma unroll 4
                for (int ps = 0; ps < LB / 4; ++ps)
                {
                        const int cr = 4 * ps + rq, grow = LB * J + cr;
                        double g;
                        if (grow > n || gcolx >= n)
                                g = 0.0;
                        else if (grow == n)
                                g = Y[gcolx];
                        else
                                g = gcol(even, uI, vI, tJ[cr][0], tJ[cr][1], tJ[cr][2], Pt[ao][cr + 1], Pt[ao + 1][cr + 1]);
                        G[(size_t)grow * NP + gcolx] = (T)g;
                }
        }
}

// ------------------------------------------------------------------------------------------------------------------
// (the diagonal blocks: large_potrf_inv_tiles in ekf_large_chol.h)

/// virtual stacked matrix M = [S; G] (2*na rows): row pointer of virtual row vr
template <typename T> __device__ __forceinline__ T *stacked_row(const LargeView<T> &lv, int b, int na, int vr)
{
        const size_t NP = lv.NP;
        return (vr < na) ? lv.S + ((size_t)b * NP + vr) * NP : lv.G + ((size_t)b * NP + (vr - na)) * NP;
}

/// Block column k0 of the stacked matrix M = [S; G; Y^T], for every 64-row block rt below the diagonal block:
///   C  = M(rt, k0) - sum_{k < k0} M(rt, k) L(k0, k)^T      left-looking update, K = 64 k0, MFMA
///   X  = C L(k0,k0)^-T = C Linv^T                           the panel "triangular solve" as a 64-deep MFMA product
///   S(rt, rt) -= X X^T  (blocks of S only)                  the diagonal blocks are kept up to date right-looking, so
///                                                           large_potrf_inv_tiles finds block k0+1 ready when this kernel ends
/// One read-modify-write of every tile per factorisation.  A workgroup takes TWO consecutive 64-row blocks (a 128x64 tile:
/// the block row of L is staged once for both, 128 MFMAs per wave between barriers); the two halves are addressed
/// independently because the pair may straddle the S / G boundary.  4 waves, wave w = rows 32w..32w+31 of the tile as
/// 2x4 MFMA 16x16 tiles.  The next K slab is fetched into registers while the current one is multiplied.
/// grid (ceil((2 nb - k0 - 1) / 2), 1, B) -- or (ceil((nb - k0 - 1) / 2), 1, B) with s_only: the blocks of S alone --, 256 threads.
template <typename T>
__global__ __launch_bounds__(256) void large_update_panel(DevView d, LargeView<T> lv, int k0, int s_only, const int *skipped)
{
        typedef Mfma<T> MM;
        constexpr int KC = (sizeof(T) == 4) ? 64 : 32; // columns staged per round = 256 bytes of a row
        constexpr int LD = KC + (sizeof(T) == 4 ? 8 : 4); // fp32: 18 sixteen-byte slots per row -> conflict-free ds_read_b128 operand reads
        typedef T vec_t __attribute__((ext_vector_type(16 / sizeof(T))));
        constexpr int VW = 16 / sizeof(T);
        static_assert(KC / VW == 16, "16 lanes per staged row");
        __shared__ __attribute__((aligned(16))) T As[2 * LB][LD];
        __shared__ __attribute__((aligned(16))) T Bs[LB][LD];
        const int b = blockIdx.z;
        if (skipped[b])
                return;
        const int n = d.n[b], NP = lv.NP;
        const int nb = large_blocks(n), na = nb * LB;
        const int rt0 = k0 + 1 + 2 * blockIdx.x; // first virtual 64-row block: S rows below the diagonal block, then all of G
        const int vlim = s_only ? nb : 2 * nb;    // s_only: the rows of G are solved by large_trsm_pipe instead
        if (k0 >= nb || rt0 >= vlim)
                return;
        const bool two = rt0 + 1 < vlim; // the last workgroup may have a single block
        const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
        const int half = wave >> 1;            // which 64-row block this wave's rows belong to
        const int rt = rt0 + half;             // its virtual block index
        const bool live = (half == 0) || two;  // wave-uniform
        const int wr = 32 * wave;              // first tile row of the wave
        const T *Brow = lv.S + ((size_t)b * NP + (size_t)k0 * LB) * NP; // block row k0 of L
        typename MM::acc_t acc[2][4];
        auto clear = [&]() {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                                acc[u][v] = MM::zero();
        };
        clear();
        // acc += As(rows of this wave) * Bsrc(rows vb .. vb+63)^T over one staged slab
        auto multiply = [&](const T(*Bsrc)[LD], int vb) {
                if constexpr (sizeof(T) == 4)
                {
                        // one 16-byte LDS read per operand row feeds four MFMA steps: lane (li, lg) holds k = 16 c + 4 lg + r for
                        // step r -- a permuted walk over the contraction index, the same for both operands
#pragma unroll
                        for (int c = 0; c < KC / 16; ++c)
                        {
                                f4 av[2], bv[4];
#pragma unroll
                                for (int u = 0; u < 2; ++u)
                                        av[u] = *reinterpret_cast<const f4 *>(&As[wr + 16 * u + li][16 * c + 4 * lg]);
#pragma unroll
                                for (int v = 0; v < 4; ++v)
                                        bv[v] = *reinterpret_cast<const f4 *>(&Bsrc[vb + 16 * v + li][16 * c + 4 * lg]);
#pragma unroll
                                for (int r = 0; r < 4; ++r)
#pragma unroll
                                        for (int u = 0; u < 2; ++u)
#pragma unroll
                                                for (int v = 0; v < 4; ++v)
                                                        acc[u][v] = MM::mma(av[u][r], bv[v][r], acc[u][v]);
                        }
                        return;
                }
#pragma unroll
                for (int s = 0; s < KC / 4; ++s)
                {
                        T av[2], bv[4];
#pragma unroll
                        for (int u = 0; u < 2; ++u)
                                av[u] = As[wr + 16 * u + li][lg + 4 * s];
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                                bv[v] = Bsrc[vb + 16 * v + li][lg + 4 * s];
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                                for (int v = 0; v < 4; ++v)
                                        acc[u][v] = MM::mma(av[u], bv[v], acc[u][v]);
                }
        };
        // ---- left-looking update.  Staging: 16 lanes cover the 256-byte row segment of a slab with 16-byte loads.
        const int lrow = tid >> 4, lc0 = (tid & 15) * VW;
        const T *Ap[8], *Bp[4];
#pragma unroll
        for (int q = 0; q < 8; ++q)
        {
                const int r = lrow + 16 * q;                           // tile row 0..127
                const int vr = (rt0 + ((r >> 6) & (two ? 1 : 0))) * LB + (r & 63); // a missing second block re-reads the first
                Ap[q] = stacked_row(lv, b, na, vr) + lc0;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
                Bp[q] = Brow + (size_t)(lrow + 16 * q) * NP + lc0;
        vec_t ta[8], tb[4];
        auto fetch = [&](int kc) {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                        ta[q] = *reinterpret_cast<const vec_t *>(Ap[q] + kc);
#pragma unroll
                for (int q = 0; q < 4; ++q)
                        tb[q] = *reinterpret_cast<const vec_t *>(Bp[q] + kc);
        };
        const int Kend = k0 * LB;
        if (Kend > 0)
                fetch(0);
        for (int kc = 0; kc < Kend; kc += KC)
        {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                        *reinterpret_cast<vec_t *>(&As[lrow + 16 * q][lc0]) = ta[q];
#pragma unroll
                for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<vec_t *>(&Bs[lrow + 16 * q][lc0]) = tb[q];
                __syncthreads();
                if (kc + KC < Kend)
                        fetch(kc + KC);
                multiply(Bs, 0);
                __syncthreads();
        }
        // ---- C = M(rt, k0) - acc (kept in registers in the accumulator layout)
        T *Mrow = stacked_row(lv, b, na, (live ? rt : rt0) * LB) + (size_t)(wr & 63) * NP; // first row of this wave
        T *Ctile = Mrow + (size_t)k0 * LB;
        typename MM::acc_t cr[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                                cr[u][v][r] = Ctile[(size_t)(16 * u + MM::row(lane, r)) * NP + 16 * v + li] - acc[u][v][r];
        clear();
        // ---- X = C Linv^T, K = 64 in slabs of KC: stage C (from registers) and Linv (row c of Linv = column c of the product)
        const T *Li = lv.Linv + ((size_t)b * LARGE_NB_MAX + k0) * LB * LB;
        auto stage_regs = [&](const typename MM::acc_t(&src)[2][4], int kh) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v)
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                {
                                        const int col = 16 * v + li;
                                        if (col / KC == kh)
                                                As[wr + 16 * u + MM::row(lane, r)][col % KC] = src[u][v][r];
                                }
        };
        for (int kh = 0; kh < LB / KC; ++kh)
        {
                stage_regs(cr, kh);
                for (int idx = tid; idx < LB * KC; idx += 256)
                        Bs[idx / KC][idx % KC] = Li[(idx / KC) * LB + kh * KC + (idx % KC)];
                __syncthreads();
                multiply(Bs, 0);
                __syncthreads();
        }
        if (live)
        {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v)
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                        Ctile[(size_t)(16 * u + MM::row(lane, r)) * NP + 16 * v + li] = acc[u][v][r];
        }
        if (rt0 >= nb)
                return; // only blocks of G in this workgroup: done (workgroup-uniform)
        // ---- S(rt, rt) -= X X^T for the blocks of S: this wave's 32 rows of X against the 64 rows of its own block
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                        cr[u][v] = acc[u][v];
        clear();
        for (int kh = 0; kh < LB / KC; ++kh)
        {
                stage_regs(cr, kh);
                __syncthreads();
                multiply(As, 64 * half);
                __syncthreads();
        }
        if (live && rt < nb)
        {
                T *Dtile = Mrow + (size_t)rt * LB;
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v)
#pragma unroll
                                for (int r = 0; r < 4; ++r)
                                        Dtile[(size_t)(16 * u + MM::row(lane, r)) * NP + 16 * v + li] -= acc[u][v][r];
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
        static_assert(sizeof(T) == 8, "binary32 products go through large_syrk_bf16x3");
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
                                (void)row0;
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

/// P -= V V^T with the binary32 products formed on the BF16 matrix pipe ("bf16x3"): every float is the exact sum of three bf16 pieces of
/// eight mantissa bits (a = a1 + a2 + a3), and  a b ~= a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1):  six v_mfma_f32_16x16x32_bf16
/// (16x16x32 per instruction, fp32 accumulation, bf16 x bf16 products exact) do the work of eight v_mfma_f32_16x16x4_f32 at twice
/// the rate, and the dropped terms (2^-24 |a b| and below) are smaller than the rounding of a binary32 product chain
/// (tools/ubench/mfma_bf16x3.hip: 6.3e-8 of sum |a b| against 1.5e-7 for an fp32 FMA chain; 294 against 149 T fp32-equivalent FLOP/s).
/// P is stored in binary64 (the update V V^T is small against P in steady state, so rounding the UPDATE to binary32 and accumulating it into
/// an fp64 P costs eps32 |dP| per callback instead of eps32 |P|: tools/fp32_drift_model.py), V in binary32.  Same 128x128 tiling and tile ->
/// workgroup map as large_syrk (4 waves x 4x4 MFMA tiles, lower tiles only, a filter's tiles on one XCD); the K loop stops at the filter's
/// size n (columns n .. of V are zero); the lower tile is read-modify-written in fp64 (16 bytes per lane and access: the products are formed
/// as B x A^T so that a lane holds four consecutive columns of a row) and its NEW value is stored, transposed, into the upper triangle -- no
/// read of the upper triangle, and P stays exactly symmetric.  The slab of V is split into its three bf16 planes by the VALU on the way into LDS (8 bytes per thread, row and plane; unpadded 64-byte rows with XOR-swizzled k-groups),
/// the operand of an MFMA is ONE 16-byte read (row l & 15, k = 8 (l >> 4) .. + 7).
///
/// The X update (large_x_update_rows: X += V q, the diagonal and the pose columns of V V^T in binary64) follows this kernel on the same stream.  This kernel
/// never touches the entries of P the other one writes (pose columns, pose rows, diagonal), so the two COULD run side by side -- round 4 measured both
/// ways (the X update's workgroups inside this launch: 2436 us against 1997 + 324 per 256 filters; on a side stream: 31.7 k against 36.4 k filter-steps/s)
/// and both lose: at this kernel's register footprint the latency-bound X-update workgroups take slots while the matrix pipes idle, and next to it they
/// fight it for L2 (profiles/r04_experiments.md section 1).
#ifndef ASLAM_XU_ROWS
#define ASLAM_XU_ROWS 8
#endif
constexpr int XU_ROWS = ASLAM_XU_ROWS; // (rows per wave of large_x_update_rows; 4 and 16 measured in round 4: profiles/r04_experiments.md)
///
/// PL = 1 (round 4; the default chain): V arrives ALREADY SPLIT -- large_trsm_bf16 stores the three bf16 planes of every solved block next to the binary32 V
/// (`vpl`: [B][3][NP][NP], the columns of every 64-block permuted as in LPlanes; the permutation stays inside a 32-column half, and a slab's contraction
/// order is the same for both operands) -- and a slab goes global -> LDS by LDS-DMA (twelve 1-KiB pieces per wave: 16 rows x 64 bytes each, the k-group
/// swizzle applied on the source side): no staging registers, no VALU split (176 instructions per thread and slab: each element of V was split ~ 17 times,
/// once per tile that reads it), no ds_write.  tools/ubench/syrk_bench.hip: K loop 1.81 -> see profiles/r04_experiments.md.
template <int DIAG = 0, int PL = 0>
__global__ __launch_bounds__(256, 3) void large_syrk_bf16x3(DevView d, LargeView<float> lv, LPlanes vpl, int nfilters, const int *skipped)
{
        constexpr int TB = 128, KC = 32;
        constexpr int LDB = KC; // bf16 per LDS row: 64 bytes, unpadded; the four 16-byte k-groups of a row are XOR-swizzled with (row & 15) >> 2, which
                                // makes the 16-byte operand reads of 16 rows conflict-free (three workgroups per CU need the 48 KB this leaves)
        typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        __shared__ __attribute__((aligned(16))) unsigned short As[3][TB * LDB];
        __shared__ __attribute__((aligned(16))) unsigned short Bs[3][TB * LDB];
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
        if (rt * TB >= n)
                return;
        const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 15, lg = lane >> 4;
        const int wr = (wave >> 1) * 64, wc = (wave & 1) * 64;
        const float *G = lv.G + (size_t)b * NP * NP;
        double *P = lv.P + (size_t)b * NP * NP;
        // staging: 8 lanes cover the 32 columns of a slab row with 16-byte loads; 32 rows per pass
        constexpr int LPR = KC / 4, RPP = 256 / LPR, NPASS = TB / RPP;
        const int lrow = tid / LPR, lc0 = (tid % LPR) * 4;
        const float *Ap[NPASS], *Bp[NPASS];
#pragma unroll
        for (int q = 0; q < NPASS; ++q)
        {
                Ap[q] = G + (size_t)min(rt * TB + lrow + RPP * q, na - 1) * NP + lc0;
                Bp[q] = G + (size_t)min(jt * TB + lrow + RPP * q, na - 1) * NP + lc0;
        }
        f4 acc[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int v = 0; v < 4; ++v)
                        acc[u][v] = (f4){0.f, 0.f, 0.f, 0.f};
        f4 ta[NPASS], tb[NPASS];
        auto fetch = [&](int kc) {
                if constexpr (PL)
                        return;
#pragma unroll
                for (int q = 0; q < NPASS; ++q)
                {
                        ta[q] = *reinterpret_cast<const f4 *>(Ap[q] + kc);
                        tb[q] = *reinterpret_cast<const f4 *>(Bp[q] + kc);
                }
        };
        // PL: the LDS-DMA of a slab.  Piece = 16 rows x 64 bytes of one plane: lane l = row l >> 2, LDS chunk l & 3, which holds logical k-group
        // (l & 3) ^ ((row & 15) >> 2) = (l & 3) ^ (l >> 4) (the swizzle of the operand reads).  Wave w moves pieces 2 w, 2 w + 1 (rows 32 w .. 32 w + 31) of the
        // three planes of A and of B.  Rows past NP read the next plane / zeros (buffer bounds): finite, and their results are never stored.
        typedef __attribute__((address_space(3))) unsigned short lds_us;
        const __amdgpu_buffer_rsrc_t vrs = __builtin_amdgcn_make_buffer_rsrc(PL ? (void *)vpl.Lq(b, NP) : (void *)nullptr, 0, (int)(LPlanes::per_filter(NP) * 2), 0x00020000);
        const unsigned dma_vo = (unsigned)(((lane >> 2) * NP) * 2 + (((lane & 3) ^ (lane >> 4)) * 16));
        const unsigned ldsA = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)(lds_us *)&As[0][0] + (unsigned)wave * 2048u));
        const unsigned ldsB = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(uintptr_t)(lds_us *)&Bs[0][0] + (unsigned)wave * 2048u));
        auto dma_slab = [&](int kc) {
                const unsigned plane_b = (unsigned)(NP * NP * 2);
                const unsigned soA = (unsigned)__builtin_amdgcn_readfirstlane(((rt * TB + 32 * wave) * NP + kc) * 2);
                const unsigned soB = (unsigned)__builtin_amdgcn_readfirstlane(((jt * TB + 32 * wave) * NP + kc) * 2);
                const unsigned r16 = (unsigned)(16 * NP * 2);
#pragma unroll
                for (int p = 0; p < 3; ++p)
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj)
                        {
                                asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds"
                                             :
                                             : "s"(ldsA), "n"(p * TB * LDB * 2 + jj * 1024), "v"(dma_vo), "s"(vrs), "s"(soA + (unsigned)p * plane_b + (unsigned)jj * r16)
                                             : "m0", "scc", "memory");
                                asm volatile("s_add_u32 m0, %0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds"
                                             :
                                             : "s"(ldsB), "n"(p * TB * LDB * 2 + jj * 1024), "v"(dma_vo), "s"(vrs), "s"(soB + (unsigned)p * plane_b + (unsigned)jj * r16)
                                             : "m0", "scc", "memory");
                        }
        };
        // four floats -> their three bf16 planes: x = h + m + l up to 2^-25 |x|.  Round to nearest at every level (v_cvt_pk_bf16_f32): with
        // truncated pieces, which all carry the sign of x, the dropped terms a2 b3 + a3 b2 have the sign of the product, V V^T comes out
        // systematically small and the covariance error grew 1.5x faster than with binary32 MFMAs; rounded, it is 4x smaller than theirs.
        typedef float f2 __attribute__((ext_vector_type(2)));
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        auto stash = [&](unsigned short (&S)[3][TB * LDB], int off, const f4 &x) {
                unsigned pk[3][2];
#pragma unroll
                for (int e = 0; e < 2; ++e)
                {
                        const f2 v = {x[2 * e], x[2 * e + 1]};
                        const bf2 h = __builtin_convertvector(v, bf2);
                        const f2 r1 = v - __builtin_convertvector(h, f2);
                        const bf2 m = __builtin_convertvector(r1, bf2);
                        const f2 r2 = r1 - __builtin_convertvector(m, f2);
                        const bf2 l = __builtin_convertvector(r2, bf2);
                        pk[0][e] = __builtin_bit_cast(unsigned, h), pk[1][e] = __builtin_bit_cast(unsigned, m), pk[2][e] = __builtin_bit_cast(unsigned, l);
                }
#pragma unroll
                for (int p = 0; p < 3; ++p)
                        *reinterpret_cast<u2 *>(&S[p][off]) = (u2){pk[p][0], pk[p][1]};
        };
        const bool idle = (rt == jt && wc > wr); // upper quadrant of a diagonal tile: the mirror image of its lower one
        // A diagonal quadrant holds (I, J) and (J, I).  With binary32 MFMA products the two sums are bit-identical; here the six partial
        // products of the pair enter them in different orders, so only J <= I is computed and the mirror image is stored from it, element
        // by element inside the 16x16 tiles on the diagonal: P stays exactly symmetric.
        const bool diagq = (rt == jt && wc == wr);
        const int nu = idle ? 0 : __builtin_amdgcn_readfirstlane(max(0, min(4, (n - (rt * TB + wr) + 15) >> 4)));
        const int nv = idle ? 0 : __builtin_amdgcn_readfirstlane(max(0, min(4, (n - (jt * TB + wc) + 15) >> 4)));
        const bool full = !(DIAG & 2) && nu == 4 && nv == 4 && !diagq; // wave-uniform: an interior 64x64 quadrant
        const int kend = min(na, (n + KC - 1) / KC * KC); // columns n .. na-1 of V are zero (G = P H^T is zero there and L is the identity)
        const int a_off = (wr + li) * LDB + 8 * (lg ^ (li >> 2)), b_off = (wc + li) * LDB + 8 * (lg ^ (li >> 2)); // swizzled k-group
        const int s_off = lrow * LDB + 8 * ((lc0 >> 3) ^ ((lrow & 15) >> 2)) + (lc0 & 4); // this thread's 8 bytes of a staged row (rows lrow + 32 q: same row & 15)
        fetch(0);
        for (int kc = 0; kc < kend; kc += KC)
        {
                if constexpr (PL)
                {
                        // single-buffered: the slab is fetched behind the barrier that ended the previous slab's reads; the two other workgroups of the CU
                        // multiply while this one waits for its pieces
                        dma_slab(kc);
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                else if (!(DIAG & 4) || kc == 0) // (DIAG 4, tools/ubench/syrk_bench.hip: timing without the split and the LDS stash -- stale LDS, wrong results)
                {
#pragma unroll
                        for (int q = 0; q < NPASS; ++q)
                        {
                                stash(As, s_off + RPP * q * LDB, ta[q]);
                                stash(Bs, s_off + RPP * q * LDB, tb[q]);
                        }
                }
                else
                {
#pragma unroll
                        for (int q = 0; q < NPASS; ++q)
                                asm volatile("" ::"v"(ta[q]), "v"(tb[q])); // the loads stay
                }
                if (!(DIAG & 32)) // (DIAG 32: timing without the barriers -- racy)
                        __syncthreads();
                if (kc + KC < kend)
                        fetch(kc + KC);
                // two column tiles of the wave's 64x64 at a time: 24 operand registers instead of 48 (three workgroups per CU).
                // B x A^T (see the epilogue), small terms first.  The six products of a slab are summed in a ZERO-INITIALISED temporary and
                // added to the running sum by the VALU once per slab: an MFMA that adds small products to a large accumulator truncates them
                // (four sequential additions per instruction, each losing the addend's low bits: tools/ubench/mfma_rounding.hip), which over
                // 198 MFMAs per element left the DIAGONAL of V V^T (sums of squares) 5.8e-7 too small -- a bias, so the covariance error grew
                // linearly with the callbacks.  With the temporary the bias is 8e-10 and the mean error 9x smaller
                // (tools/ubench/syrk_accum.hip).  Interior tiles (the bulk) take the branch-free path, where the addition of a pair of
                // temporaries is issued behind the MFMAs of the NEXT pair.
#define ASLAM_MM(t, bb, aa, c) t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, bb), __builtin_bit_cast(bf8, aa), c, 0, 0, 0)
                if (DIAG & 8) // (timing without the MFMAs)
                {
                }
                else if (full)
                {
#pragma unroll
                        for (int vh = 0; vh < 4; vh += 2)
                        {
                                u4 bq[2][3];
#pragma unroll
                                for (int v = 0; v < 2; ++v)
#pragma unroll
                                        for (int p = 0; p < 3; ++p)
                                        {
                                                if (DIAG & 16) // (timing without the operand reads)
                                                        asm volatile("" : "=v"(bq[v][p]));
                                                else
                                                        bq[v][p] = *reinterpret_cast<const u4 *>(&Bs[p][b_off + 16 * (vh + v) * LDB]);
                                        }
                                f4 pa = {0.f, 0.f, 0.f, 0.f}, pb = {0.f, 0.f, 0.f, 0.f}; // the previous row tile's pair, not yet added
#pragma unroll
                                for (int u = 0; u < 4; ++u)
                                {
                                        u4 ap[3];
#pragma unroll
                                        for (int p = 0; p < 3; ++p)
                                        {
                                                if (DIAG & 16)
                                                        asm volatile("" : "=v"(ap[p]));
                                                else
                                                        ap[p] = *reinterpret_cast<const u4 *>(&As[p][a_off + 16 * u * LDB]);
                                        }
                                        f4 ta, tb;
                                        ASLAM_MM(ta, bq[0][0], ap[2], ((f4){0.f, 0.f, 0.f, 0.f}));
                                        ASLAM_MM(tb, bq[1][0], ap[2], ((f4){0.f, 0.f, 0.f, 0.f}));
                                        ASLAM_MM(ta, bq[0][1], ap[1], ta);
                                        ASLAM_MM(tb, bq[1][1], ap[1], tb);
                                        ASLAM_MM(ta, bq[0][2], ap[0], ta);
                                        ASLAM_MM(tb, bq[1][2], ap[0], tb);
                                        if (u > 0)
                                        {
                                                acc[u - 1][vh] += pa;
                                                acc[u - 1][vh + 1] += pb;
                                        }
                                        ASLAM_MM(ta, bq[0][0], ap[1], ta);
                                        ASLAM_MM(tb, bq[1][0], ap[1], tb);
                                        ASLAM_MM(ta, bq[0][1], ap[0], ta);
                                        ASLAM_MM(tb, bq[1][1], ap[0], tb);
                                        ASLAM_MM(ta, bq[0][0], ap[0], ta);
                                        ASLAM_MM(tb, bq[1][0], ap[0], tb);
                                        pa = ta, pb = tb;
                                        __builtin_amdgcn_sched_barrier(0); // (left alone hipcc hoists every operand read of the slab to its top and spills accumulators)
                                }
                                acc[3][vh] += pa;
                                acc[3][vh + 1] += pb;
                        }
                }
                else
                {
#pragma unroll
                        for (int vh = 0; vh < 4; vh += 2)
                        {
                                u4 bq[2][3];
#pragma unroll
                                for (int v = 0; v < 2; ++v)
#pragma unroll
                                        for (int p = 0; p < 3; ++p)
                                                bq[v][p] = *reinterpret_cast<const u4 *>(&Bs[p][b_off + 16 * (vh + v) * LDB]);
#pragma unroll
                                for (int u = 0; u < 4; ++u)
                                        if (u < nu)
                                        {
                                                u4 ap[3];
#pragma unroll
                                                for (int p = 0; p < 3; ++p)
                                                        ap[p] = *reinterpret_cast<const u4 *>(&As[p][a_off + 16 * u * LDB]);
#pragma unroll
                                                for (int v = 0; v < 2; ++v)
                                                        if (vh + v < nv && !(diagq && vh + v > u))
                                                        {
                                                                f4 tmp;
                                                                if constexpr (DIAG & 2) // diagnostic build only (trsm_bench): round 2's running accumulator
                                                                {
                                                                        ASLAM_MM(tmp, bq[v][0], ap[2], acc[u][vh + v]);
                                                                        ASLAM_MM(tmp, bq[v][1], ap[1], tmp);
                                                                        ASLAM_MM(tmp, bq[v][2], ap[0], tmp);
                                                                        ASLAM_MM(tmp, bq[v][0], ap[1], tmp);
                                                                        ASLAM_MM(tmp, bq[v][1], ap[0], tmp);
                                                                        ASLAM_MM(tmp, bq[v][0], ap[0], tmp);
                                                                        acc[u][vh + v] = tmp;
                                                                        continue;
                                                                }
                                                                ASLAM_MM(tmp, bq[v][0], ap[2], ((f4){0.f, 0.f, 0.f, 0.f}));
                                                                ASLAM_MM(tmp, bq[v][1], ap[1], tmp);
                                                                ASLAM_MM(tmp, bq[v][2], ap[0], tmp);
                                                                ASLAM_MM(tmp, bq[v][0], ap[1], tmp);
                                                                ASLAM_MM(tmp, bq[v][1], ap[0], tmp);
                                                                ASLAM_MM(tmp, bq[v][0], ap[0], tmp);
                                                                acc[u][vh + v] += tmp;
                                                        }
                                        }
                        }
                }
#undef ASLAM_MM
                if (!(DIAG & 32)) // (DIAG 32: timing without the barriers -- racy)
                        __syncthreads();
        }
        if (idle)
                return;
        if constexpr (DIAG & 1)
        {
                // diagnostic build only (tools/ubench/trsm_bench.hip): the K loop without the read-modify-write of P
                float sres = 0.f;
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                                sres += acc[u][v][0] + acc[u][v][1] + acc[u][v][2] + acc[u][v][3];
                if (sres == 12345.678f)
                        P[0] = sres;
                return;
        }
        const bool mirror = (jt < rt || wc < wr);
        // The DIAGONAL of V V^T and its three POSE columns (and their mirror image, the pose rows) are formed with binary64 accumulation by the X-update
        // kernel (large_x_update_rows) and are NOT TOUCHED here -- not even read and stored back (a store-back of an old value would tie the order of the two
        // kernels).  Pose columns 0..2 = registers 0..2 of lane group 0 in the first
        // column tile of tile column 0 (posecols below: only its column 3 is updated, by 8-byte accesses); the diagonal is skipped element-wise.
        const bool posecols_lane = (jt == 0 && wc == 0 && lg == 0);
        {
                // The K loop multiplies B x A^T (operands swapped), so a lane's four registers are four consecutive COLUMNS of one row of the
                // lower tile: 32 contiguous bytes, two 16-byte loads and stores per 16x16 tile; the mirror image is the strided side (four
                // 8-byte stores).  With A x B^T (four consecutive rows per lane: four 8-byte loads + stores, 16-byte mirror stores) the
                // epilogue cost 0.66 ms per 256 filters against 0.33 ms (trsm_bench).
                typedef double d2 __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int v = 0; v < 4; ++v)
                        {
                                const int row = rt * TB + wr + 16 * u + li;
                                const int col0 = jt * TB + wc + 16 * v + 4 * lg;
                                if (row >= n || col0 >= n || (diagq && v > u))
                                        continue;
                                const bool posecols = posecols_lane && v == 0; // this lane's four columns are 0 .. 3
                                if ((diagq && v == u) || posecols)
                                {
                                        // 16x16 tile on the diagonal: the elements J < I, each with its mirror image (J == I: the X update);
                                        // columns 0 .. 3: column 3 only
#pragma unroll
                                        for (int r = 0; r < 4; ++r)
                                                if (col0 + r < row && !(posecols && r < 3) && col0 + r < n)
                                                {
                                                        double *pe = P + (size_t)row * NP + col0 + r;
                                                        const double pn = *pe - (double)acc[u][v][r];
                                                        *pe = pn;
                                                        P[(size_t)(col0 + r) * NP + row] = pn;
                                                }
                                        continue;
                                }
                                double *pp = P + (size_t)row * NP + col0;
                                double nv4[4];
                                if (col0 + 3 < n)
                                {
                                        const d2 o0 = *reinterpret_cast<const d2 *>(pp), o1 = *reinterpret_cast<const d2 *>(pp + 2);
                                        nv4[0] = o0[0] - (double)acc[u][v][0], nv4[1] = o0[1] - (double)acc[u][v][1];
                                        nv4[2] = o1[0] - (double)acc[u][v][2], nv4[3] = o1[1] - (double)acc[u][v][3];
                                        *reinterpret_cast<d2 *>(pp) = (d2){nv4[0], nv4[1]};
                                        *reinterpret_cast<d2 *>(pp + 2) = (d2){nv4[2], nv4[3]};
                                }
                                else
                                {
#pragma unroll
                                        for (int r = 0; r < 4; ++r)
                                                if (col0 + r < n)
                                                {
                                                        nv4[r] = pp[r] - (double)acc[u][v][r];
                                                        pp[r] = nv4[r];
                                                }
                                }
                                if (mirror || diagq)
                                {
#pragma unroll
                                        for (int r = 0; r < 4; ++r)
                                                if (col0 + r < n)
                                                        P[(size_t)(col0 + r) * NP + row] = nv4[r];
                                }
                        }
        }
}

/// X <- X + V q with q = row n of G = (L^-1 Y)^T; one wave per state row.  grid (ceil(NP/4), B), 256 threads.  In replay
/// mode also writes the pose of this callback.
///
/// binary32 chain (SLIM): the same pass over row a of V also forms, with binary64 accumulation, the entries of V V^T that large_syrk_bf16x3
/// leaves out -- the DIAGONAL (a sum of squares: every rounding of an fp32 accumulator chain pulls it the same way, and the covariance
/// diagonal is where the error of the fp32 path was largest) and the three POSE columns (the pose block changes by as much as it holds in
/// every callback, Q against the update, so eps32 |dP| is eps32 |P| there) -- and subtracts them from P: P(a,a), P(a,0..2) and the mirror
/// image P(0..2,a).  Costs four more FMAs per element of a row the kernel reads anyway.
template <typename T, int MODE, bool SLIM = false>
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
        // 16 bytes per lane and load (rows are 256-byte aligned: NP is a multiple of 64); columns n .. of both rows are zero
        typedef T vec_t __attribute__((ext_vector_type(16 / sizeof(T))));
        constexpr int VW = 16 / sizeof(T);
        if constexpr (SLIM)
        {
                const T *p0 = lv.G + (size_t)b * NP * NP, *p1 = p0 + NP, *p2 = p1 + NP;
                double dd = 0.0, d0 = 0.0, d1 = 0.0, d2 = 0.0;
                for (int j = VW * lane; j < n; j += 64 * VW)
                {
                        const vec_t v = *reinterpret_cast<const vec_t *>(vrow + j), w = *reinterpret_cast<const vec_t *>(q + j);
                        const vec_t u0 = *reinterpret_cast<const vec_t *>(p0 + j), u1 = *reinterpret_cast<const vec_t *>(p1 + j),
                                    u2 = *reinterpret_cast<const vec_t *>(p2 + j);
#pragma unroll
                        for (int e = 0; e < VW; ++e)
                        {
                                const double ve = (double)v[e];
                                acc = fma(ve, (double)w[e], acc);
                                dd = fma(ve, ve, dd);
                                d0 = fma(ve, (double)u0[e], d0);
                                d1 = fma(ve, (double)u1[e], d1);
                                d2 = fma(ve, (double)u2[e], d2);
                        }
                }
                dd = wave_sum_dpp(dd), d0 = wave_sum_dpp(d0), d1 = wave_sum_dpp(d1), d2 = wave_sum_dpp(d2);
                if (lane == 63)
                {
                        double *P = lv.P + (size_t)b * NP * NP;
                        double *prow = P + (size_t)a * NP;
                        const double dp[3] = {d0, d1, d2};
                        // pose columns j < min(a, 3) with their mirror image; the diagonal entry itself (a < 3: dd == dp[a] bit for bit)
                        for (int j = 0; j < 3 && j < a; ++j)
                        {
                                const double pn = prow[j] - dp[j];
                                prow[j] = pn;
                                P[(size_t)j * NP + a] = pn;
                        }
                        prow[a] -= dd;
                }
        }
        else
        {
                for (int j = VW * lane; j < n; j += 64 * VW)
                {
                        const vec_t v = *reinterpret_cast<const vec_t *>(vrow + j), w = *reinterpret_cast<const vec_t *>(q + j);
#pragma unroll
                        for (int e = 0; e < VW; ++e)
                                acc = fma((double)v[e], (double)w[e], acc);
                }
        }
        acc = wave_sum_dpp(acc);
        if (lane == 63)
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
/// binary32 chain: X <- X + V q, and the diagonal + the three pose columns of V V^T in binary64 (see large_x_update<.., SLIM>), with the four rows
/// every row of V is multiplied with -- q and the pose rows of V -- staged ONCE per workgroup in LDS (17 KB) for the XU_ROWS x 4 rows its waves
/// walk.  With a wave per row and the shared rows read from memory every wave issued 25 loads of 16 bytes per lane for 4 KB of new data: 391 us
/// per 256 filters against 217 us for round 2's single product (profiles/r03_experiments.md).  grid (ceil(NP / (4 XU_ROWS)), B), 256 threads.
/// the body: workgroup `xb` of filter `b` (32 rows); sh = 4 x LARGE_NP_MAX floats of LDS
template <int MODE>
__device__ __forceinline__ void x_update_rows_body(float (*sh)[LARGE_NP_MAX], const DevView &d, const LargeView<float> &lv, int b, int xb, int n, int s, int nsteps,
                                                   double *poses_out, int32_t *dims_out)
{
        const int NP = lv.NP;
        const int tid = threadIdx.x, lane = tid & 63;
        if (xb * 4 * XU_ROWS >= n)
                return;
        const float *G = lv.G + (size_t)b * NP * NP;
        for (int j = 4 * tid; j < NP; j += 4 * 256)
        {
                *reinterpret_cast<f4 *>(&sh[0][j]) = *reinterpret_cast<const f4 *>(G + (size_t)n * NP + j);
                *reinterpret_cast<f4 *>(&sh[1][j]) = *reinterpret_cast<const f4 *>(G + j);
                *reinterpret_cast<f4 *>(&sh[2][j]) = *reinterpret_cast<const f4 *>(G + NP + j);
                *reinterpret_cast<f4 *>(&sh[3][j]) = *reinterpret_cast<const f4 *>(G + 2 * (size_t)NP + j);
        }
        __syncthreads();
        double *P = lv.P + (size_t)b * NP * NP;
        const int a0 = (xb * 4 + (tid >> 6)) * XU_ROWS;
        constexpr int NPASS = 5; // 5 x 64 lanes x 4 columns = 1280 >= LARGE_NP_MAX
        auto load_row = [&](f4 (&v)[NPASS], int a) {
                const float *vr = G + (size_t)min(a, n - 1) * NP;
#pragma unroll
                for (int i = 0; i < NPASS; ++i)
                {
                        const int j = 4 * lane + 256 * i;
                        v[i] = j < n ? *reinterpret_cast<const f4 *>(vr + j) : (f4){0.f, 0.f, 0.f, 0.f}; // (columns n .. of every row of V are zero)
                }
        };
        // the loads of row a + 1 are issued before row a is multiplied: the kernel lives on bytes in flight
        f4 vn[NPASS];
        load_row(vn, a0);
#pragma unroll 1
        for (int r = 0; r < XU_ROWS; ++r)
        {
                const int a = a0 + r;
                if (a >= n)
                        break; // wave-uniform
                f4 v[NPASS];
#pragma unroll
                for (int i = 0; i < NPASS; ++i)
                        v[i] = vn[i];
                if (r + 1 < XU_ROWS)
                        load_row(vn, a + 1);
                double acc = 0.0, dd = 0.0, d0 = 0.0, d1 = 0.0, d2 = 0.0;
#pragma unroll
                for (int i = 0; i < NPASS; ++i)
                {
                        const int j = 4 * lane + 256 * i;
                        if (j >= n)
                                continue; // (beyond NP nothing was staged)
                        const f4 wi = *reinterpret_cast<const f4 *>(&sh[0][j]), p0 = *reinterpret_cast<const f4 *>(&sh[1][j]),
                                 p1 = *reinterpret_cast<const f4 *>(&sh[2][j]), p2 = *reinterpret_cast<const f4 *>(&sh[3][j]);
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                        {
                                const double ve = (double)v[i][e];
                                acc = fma(ve, (double)wi[e], acc);
                                dd = fma(ve, ve, dd);
                                d0 = fma(ve, (double)p0[e], d0);
                                d1 = fma(ve, (double)p1[e], d1);
                                d2 = fma(ve, (double)p2[e], d2);
                        }
                }
                acc = wave_sum_dpp(acc), dd = wave_sum_dpp(dd), d0 = wave_sum_dpp(d0), d1 = wave_sum_dpp(d1), d2 = wave_sum_dpp(d2);
                if (lane == 63)
                {
                        double *prow = P + (size_t)a * NP;
                        const double dp[3] = {d0, d1, d2};
                        // pose columns j < min(a, 3) with their mirror image; the diagonal entry itself
                        for (int j = 0; j < 3 && j < a; ++j)
                        {
                                const double pn = prow[j] - dp[j];
                                prow[j] = pn;
                                P[(size_t)j * NP + a] = pn;
                        }
                        prow[a] -= dd;
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
}

/// grid (ceil(NP / (4 XU_ROWS)), B), 256 threads
template <int MODE>
__global__ __launch_bounds__(256) void large_x_update_rows(DevView d, LargeView<float> lv, int s, int nsteps, double *poses_out, int32_t *dims_out, const int *skipped)
{
        __shared__ __attribute__((aligned(16))) float sh[4][LARGE_NP_MAX]; // q, pose rows 0 .. 2 of V
        const int b = blockIdx.y;
        if (skipped[b])
                return;
        x_update_rows_body<MODE>(sh, d, lv, b, (int)blockIdx.x, d.n[b], s, nsteps, poses_out, dims_out);
}
} // namespace aslam

#include "ekf_large_trsm.h"
#include "ekf_large_chol.h"
#include "ekf_large_trsm16.h"
