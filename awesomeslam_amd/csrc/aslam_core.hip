// aslam_core.hip -- libaslam_core.so: context management + C ABI (include/aslam_core.h) over the gfx950 kernels.
//
// Build (see Makefile): hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC
// No CPU fallback exists: without a HIP device every compute entry point returns ASLAM_ERR_HIP.
#include "../../include/aslam_core.h"
#include "../../include/aslam_scan.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "ekf_small.h"
#include "ekf_large.h"
#include "aslam_large16.h"
#include "scan_front.h"
#if ASLAM_HAVE_UKF
#include "ukf_small.h"
#endif

using namespace aslam;

namespace
{
thread_local std::string g_err;

int fail(int code, const std::string &msg)
{
        g_err = msg;
        return code;
}

#define HIP_TRY(expr)                                                                                                  \
        do                                                                                                             \
        {                                                                                                              \
                hipError_t e_ = (expr);                                                                                \
                if (e_ != hipSuccess)                                                                                  \
                        return fail(ASLAM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));                 \
        } while (0)
} // namespace

struct aslam_ctx
{
        aslam_config cfg;
        int NT;   // 16-wide tiles per dimension
        int NP;   // padded dimension
        DevView dv;
        std::vector<void *> owned; // hipMalloc'ed blocks (state)
        std::vector<void *> trace_owned;
        hipStream_t last_stream;
        int64_t hbm_bytes;
        size_t lds_bytes;
        std::string kernel_name;
#if ASLAM_HAVE_UKF
        UkfView ukf = {};
#endif
        // large-state path (n > 143): typed covariance buffers, one launch chain per callback
        bool large = false;
        LargeView<double> lv64 = {};
        LargeView<float> lv32 = {};
        int *skipped = nullptr;
        float *step_in = nullptr; // [3][batch] vx, az, dt of a batched step
        double *largeP = nullptr; // = lv64.P or lv32.P: the covariance is binary64 in both modes
        // replay splits the batch into groups that run the launch chain side by side on separate streams: the latency-bound
        // launches of one group (one-wave diagonal factorisations, the front end, short-K panels) then overlap the GEMMs of
        // the others
        static constexpr int LARGE_GROUPS = 8; // capacity; the default below was chosen by measurement (profiles/)
        int large_groups = 3;                  // ASLAM_LARGE_GROUPS=1..8 overrides (1 = a single stream, for per-kernel profiling); 3: 88 / 88 / 80 of 256 filters --
                                               // measured best at the end of round 4 (tools/manual/sweep_groups.sh: 37.4 - 38.0 k filter-steps/s against 36.6 - 36.8 k with 4, 36.5 - 37.1 k with 2)
        // binary32 mode: Cholesky of S as ONE launch with a filter per workgroup (large_chol_resident) when the batch can fill the
        // chip that way, as 33 multi-workgroup launches (diagonal block + panel per block column) for few filters.
        // ASLAM_CHOL_RESIDENT=0/1 forces one form.
        int chol_resident = -1;
        int right_step = 1;   // binary32 mode below the resident batch: the right-looking one-launch-per-block-column chain (large_right_step); ASLAM_RIGHT_STEP=0: the left-looking chain of rounds 1 - 2 (potrf + panel launches, large_trsm_pipe)
        int bf16_pipe = 3;    // binary32 mode, resident Cholesky: Cholesky and TRSM on the bf16 matrix pipe (large_chol_bf16 + large_trsm_bf16, ekf_large_trsm16.h); ASLAM_BF16_PIPE=0: the fp32-MFMA kernels
        int keep_l32 = 0;     // ASLAM_KEEP_L32=1 (tests/manual/large_residuals.py reads L back): large_chol_bf16 also stores the off-diagonal blocks of L in binary32
        int gs_tiles = 0;     // G, S from the lower block triangle of P (large_build_GS_tiles); ASLAM_GS_TILES=0: the row-pair kernel that reads all of P (large_build_GS) -- bit-identical results
        int syrk_running = 0; // diagnostic (ASLAM_SYRK_RUNNING=1): round 2's accumulation order in large_syrk_bf16x3 (profiles/r03_experiments.md)
        static constexpr int CHOL_RESIDENT_MIN_BATCH = 32;
        hipStream_t aux[LARGE_GROUPS - 1] = {};
        hipEvent_t ev_fork = nullptr, ev_join[LARGE_GROUPS - 1] = {};
        // what the LAST launch of this context really did (aslam_get_launch_info: the tests assert on it, not on the configuration)
        int last_groups = 0;        // stream groups that received work (1 = the caller's stream alone; 0 = single-CU kernel / nothing launched yet)
        int last_resident = 0;      // 1 = the Cholesky of S ran as large_chol_resident
        int last_launches = 0;      // kernel launches per callback and stream group
};

namespace
{
template <typename T> int dev_alloc(aslam_ctx *c, T **p, size_t count, std::vector<void *> &pool)
{
        void *q = nullptr;
        HIP_TRY(hipMalloc(&q, count * sizeof(T)));
        HIP_TRY(hipMemset(q, 0, count * sizeof(T)));
        pool.push_back(q);
        c->hbm_bytes += (int64_t)(count * sizeof(T));
        *p = static_cast<T *>(q);
        return ASLAM_OK;
}

/// copy a host n x n double matrix into filter `traj`'s P (padded stride NP, converted to the context's dtype)
int upload_P(aslam_ctx *c, int traj, int n, const double *P)
{
        const size_t NP = (size_t)c->NP;
        if (!c->large)
        {
                std::vector<double> v(NP * NP, 0.0);
                for (int i = 0; i < n; ++i)
                        std::memcpy(&v[(size_t)i * NP], P + (size_t)i * n, sizeof(double) * n);
                HIP_TRY(hipMemcpy(c->dv.P + traj * NP * NP, v.data(), sizeof(double) * NP * NP, hipMemcpyHostToDevice));
        }
        else
        {
                // the large path keeps P in binary64 in both modes (fp32 mode: binary32 G, S, L, V and MFMA products)
                std::vector<double> v(NP * NP, 0.0);
                for (int i = 0; i < n; ++i)
                        std::memcpy(&v[(size_t)i * NP], P + (size_t)i * n, sizeof(double) * n);
                HIP_TRY(hipMemcpy(c->largeP + traj * NP * NP, v.data(), sizeof(double) * NP * NP, hipMemcpyHostToDevice));
        }
        return ASLAM_OK;
}

int download_P(aslam_ctx *c, int traj, int n, double *P)
{
        const size_t NP = (size_t)c->NP;
        const double *src = c->large ? c->largeP : c->dv.P;
        HIP_TRY(hipMemcpy2D(P, sizeof(double) * n, src + traj * NP * NP, sizeof(double) * NP, sizeof(double) * n, n, hipMemcpyDeviceToHost));
        return ASLAM_OK;
}

/// new rows of P on growth: zero, with KP_LANDMARK_POSE on the diagonal (rows n_old .. n_new-1, full padded length)
int grow_P_rows(aslam_ctx *c, int traj, int n_old, int n_new)
{
        const size_t NP = (size_t)c->NP;
        for (int i = n_old; i < n_new; ++i)
        {
                std::vector<double> row(NP, 0.0);
                row[i] = (double)KP_LANDMARK_POSE;
                double *dst = (c->large ? c->largeP : c->dv.P) + traj * NP * NP + (size_t)i * NP;
                HIP_TRY(hipMemcpy(dst, row.data(), sizeof(double) * NP, hipMemcpyHostToDevice));
        }
        return ASLAM_OK;
}

int check_traj(aslam_ctx *c, int traj)
{
        if (!c)
                return fail(ASLAM_ERR_ARG, "null context");
        if (traj < 0 || traj >= c->cfg.batch)
                return fail(ASLAM_ERR_ARG, "trajectory index out of range");
        return ASLAM_OK;
}

int sync_ctx(aslam_ctx *c)
{
        HIP_TRY(hipStreamSynchronize(c->last_stream));
        return ASLAM_OK;
}

template <typename T> int init_P_large(aslam_ctx *c, LargeView<T> &lv)
{
        const size_t B = c->cfg.batch, NP = c->NP;
        HIP_TRY(hipMemset(lv.P, 0, sizeof(double) * B * NP * NP));
        HIP_TRY(hipMemset(lv.G, 0, sizeof(T) * B * NP * NP));
        HIP_TRY(hipMemset(lv.S, 0, sizeof(T) * B * NP * NP));
        if (lv.Vw)
                HIP_TRY(hipMemset(lv.Vw, 0, sizeof(T) * B * NP * NP));
        std::vector<double> blk(3 * NP, 0.0);
        for (int i = 0; i < 3; ++i)
                blk[(size_t)i * NP + i] = (double)KP_ROBOT_POSE;
        for (size_t b = 0; b < B; ++b)
                HIP_TRY(hipMemcpy(lv.P + b * NP * NP, blk.data(), sizeof(double) * blk.size(), hipMemcpyHostToDevice));
        return ASLAM_OK;
}

int init_state_large(aslam_ctx *c)
{
        return c->cfg.dtype == ASLAM_F32 ? init_P_large(c, c->lv32) : init_P_large(c, c->lv64);
}

/// initialize() for the whole batch: ekf.cpp:49-71 / ukf.cpp:49-67
int init_state(aslam_ctx *c)
{
        const int B = c->cfg.batch, NP = c->NP;
        DevView &d = c->dv;
        HIP_TRY(hipMemset(d.X, 0, sizeof(double) * B * NP));
        HIP_TRY(hipMemset(d.Z, 0, sizeof(double) * B * NP));
        if (!c->large)
                HIP_TRY(hipMemset(d.P, 0, sizeof(double) * (size_t)B * NP * NP));
        HIP_TRY(hipMemset(d.status, 0, sizeof(uint32_t) * B));
        HIP_TRY(hipMemset(d.sens_n, 0, sizeof(int) * B));
        HIP_TRY(hipMemset(d.wait_n, 0, sizeof(int) * B));
        std::vector<int> n(B, 3), fl(B, FLAG_INIT_X | FLAG_INIT_Z);
        std::vector<double> A(2 * (size_t)B);
        for (int b = 0; b < B; ++b)
        {
                A[2 * b] = 1.0; // A = Identity
                A[2 * b + 1] = 0.0;
        }
        HIP_TRY(hipMemcpy(d.n, n.data(), sizeof(int) * B, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d.flags, fl.data(), sizeof(int) * B, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d.A, A.data(), sizeof(double) * 2 * B, hipMemcpyHostToDevice));
        if (c->large)
                return init_state_large(c);
#if ASLAM_HAVE_UKF
        if (c->ukf.D)
        { // a reset context is a fresh one: the scratch too (ukf_alloc zeroes it; the kernels rely on never-written padding staying zero)
                const size_t MP = (size_t)c->ukf.MP;
                HIP_TRY(hipMemset(c->ukf.D, 0, sizeof(double) * (size_t)B * NP * MP));
                HIP_TRY(hipMemset(c->ukf.DZ, 0, sizeof(double) * (size_t)B * NP * MP));
                HIP_TRY(hipMemset(c->ukf.Tc, 0, sizeof(double) * (size_t)B * NP * NP));
                HIP_TRY(hipMemset(c->ukf.K, 0, sizeof(double) * (size_t)B * NP * NP));
        }
#endif
        // P = Identity * KP_ROBOT_POSE on the 3 pose entries
        const double p0 = (double)KP_ROBOT_POSE;
        std::vector<double> blk((size_t)3 * NP, 0.0);
        for (int i = 0; i < 3; ++i)
                blk[(size_t)i * NP + i] = p0;
        for (int b = 0; b < B; ++b)
                HIP_TRY(hipMemcpy(d.P + (size_t)b * NP * NP, blk.data(), sizeof(double) * blk.size(), hipMemcpyHostToDevice));
        return ASLAM_OK;
}

template <int NT, int MODE>
int launch_ekf(aslam_ctx *c, int grid, int64_t t0, int nsteps, double *poses, int32_t *dims, StepArgs sa, hipStream_t st)
{
        auto kern = ekf_small_kernel<NT, MODE>;
        const size_t lds = SmallLayout<NT>::total;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(SMALL_WG), lds, st, c->dv, t0, nsteps, poses, dims, sa);
        HIP_TRY(hipGetLastError());
        return ASLAM_OK;
}

#if ASLAM_HAVE_UKF
/// HBM scratch of the UKF kernels: D, DZ ([NP][MP]) and Tc, K ([NP][NP]) per filter
int ukf_alloc(aslam_ctx *c)
{
        const size_t B = (size_t)c->cfg.batch, NP = (size_t)c->NP;
        const size_t MP = 2 * NP + 16;
        c->ukf.MP = (int)MP;
        int rc = dev_alloc(c, &c->ukf.D, B * NP * MP, c->owned);
        if (rc == ASLAM_OK)
                rc = dev_alloc(c, &c->ukf.DZ, B * NP * MP, c->owned);
        if (rc == ASLAM_OK)
                rc = dev_alloc(c, &c->ukf.Tc, B * NP * NP, c->owned);
        if (rc == ASLAM_OK)
                rc = dev_alloc(c, &c->ukf.K, B * NP * NP, c->owned);
        return rc;
}

template <int NT, int MODE>
int launch_ukf(aslam_ctx *c, int grid, int64_t t0, int nsteps, double *poses, int32_t *dims, StepArgs sa, hipStream_t st)
{
        auto kern = ukf_small_kernel<NT, MODE>;
        const size_t lds = UkfLayout<NT>::total;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(SMALL_WG), lds, st, c->dv, c->ukf, t0, nsteps, poses, dims, sa);
        HIP_TRY(hipGetLastError());
        return ASLAM_OK;
}
#endif

/// one callback of the large-state EKF: front end + predict, G, S, blocked factorisation of [S; G; Y^T], P -= V V^T, X += V q
template <typename T, int MODE>
int launch_large_T(aslam_ctx *c, LargeView<T> &lv, int grid, int64_t t0, int nsteps, double *poses, int32_t *dims, StepArgs sa,
                   hipStream_t st)
{
        const int NP = c->NP, NB = NP / LB;
        const size_t lds = LargeLds::bytes(NP);
        auto fk = large_frontend_kernel<T, MODE>;
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(fk), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int Bz = (MODE == MODE_STEP && sa.traj >= 0) ? 1 : c->cfg.batch; // sa.traj < 0: the batched step
        (void)grid;
        // the kernels index the filter through blockIdx: a group of filters starting at b0 gets views shifted to b0
        struct Group
        {
                DevView dv;
                LargeView<T> v;
                int *skip;
                double *poses;
                int32_t *dims;
                int nb;
                hipStream_t st;
        };
        auto make_group = [&](int b0, int nb, hipStream_t gst) {
                Group g = {c->dv, lv, c->skipped, poses, dims, nb, gst};
                const size_t b = (size_t)b0, np = (size_t)NP, T_ = (size_t)c->dv.T;
                g.dv.X += b * np;
                g.dv.Z += b * np;
                g.dv.A += 2 * b;
                g.dv.n += b;
                g.dv.flags += b;
                g.dv.status += b;
                g.dv.sens += b * (size_t)c->dv.max_obs * 2;
                g.dv.sens_n += b;
                g.dv.wait_rb += b * (size_t)c->dv.max_wait * 2;
                g.dv.wait_cnt += b * (size_t)c->dv.max_wait;
                g.dv.wait_n += b;
                g.dv.step_in += b; // [3][B]: the stride stays the whole batch
                if (MODE == MODE_REPLAY && b0 > 0)
                {
                        g.dv.tr_pose += 2 * b * T_;
                        g.dv.tr_yaw += b * T_;
                        g.dv.tr_twist += 2 * b * T_;
                        g.dv.tr_dt += b * T_;
                        g.dv.tr_new += b * T_;
                        g.dv.tr_nobs += b * T_;
                        g.dv.tr_obs += b * T_ * (size_t)c->dv.max_obs * 2;
                        if (g.poses)
                                g.poses += b * (size_t)nsteps * 3;
                        if (g.dims)
                                g.dims += b * (size_t)nsteps;
                }
                g.v.P += b * np * np;
                g.v.G += b * np * np;
                g.v.S += b * np * np;
                g.v.Hc += b * (np / 2) * 4;
                g.v.Linv += b * LARGE_NB_MAX * LB * LB;
                if constexpr (sizeof(T) == 4)
                {
                        if (g.v.Lpl)
                                g.v.Lpl += b * LPlanes::per_filter((int)np);
                        if (g.v.Vw)
                                g.v.Vw += b * np * np;
                }
                g.v.Y += b * np;
                g.skip += b;
                return g;
        };
        const bool resident = c->chol_resident >= 0 ? c->chol_resident != 0 : Bz >= aslam_ctx::CHOL_RESIDENT_MIN_BATCH;
        auto chain = [&](const Group &g, int s) {
                const int gb = g.nb;
                hipLaunchKernelGGL(fk, dim3(gb), dim3(SMALL_WG), lds, g.st, g.dv, g.v, t0 + s, s, nsteps, g.poses, g.dims, sa, g.skip);
                if (c->gs_tiles)
                        hipLaunchKernelGGL(large_build_GS_tiles<T>, dim3(NB * (NB + 1) / 2, gb), dim3(256), 0, g.st, g.dv, g.v, g.skip);
                else
                        hipLaunchKernelGGL(large_build_GS<T>, dim3(1 + (NP / 2 + GS_ROW_PAIRS - 1) / GS_ROW_PAIRS, gb), dim3(256), 0, g.st, g.dv, g.v, g.skip);
                const int ntile = (NP + 127) / 128;
                const dim3 syrk_grid(8 * (ntile * (ntile + 1) / 2) * ((gb + 7) / 8));
                LargeView<T> vv = g.v; // what the consumers of V read
                if constexpr (sizeof(T) == 4)
                {
                        // binary32: from 32 filters on the resident kernels (one launch each: Cholesky of S with a filter per workgroup, then V = G L^-T with
                        // the solved columns in registers); below that one right-looking launch per block column that factors S and solves the rows of G
                        // together (large_right_step; ASLAM_RIGHT_STEP=0: rounds 1 - 2's 17 x {diagonal block, panel of S} + large_trsm_pipe); then
                        // P -= V V^T into the fp64 covariance
                        const bool right = !resident && c->right_step && g.v.Vw != nullptr;
                        const bool pipe16 = resident && g.v.Lpl != nullptr && (c->bf16_pipe & 1), chol16 = resident && g.v.Lpl != nullptr && (c->bf16_pipe & 2);
                        if (chol16)
                                launch_chol_bf16(g.dv, g.v, gb, g.skip, g.st, !pipe16 || c->keep_l32); // (the bf16 TRSM reads L through its planes only)
                        else if (resident)
                                hipLaunchKernelGGL(large_chol_resident<LARGE_NB_MAX>, dim3(gb), dim3(256), 0, g.st, g.dv, g.v, g.skip);
                        else if (right)
                        {
                                hipLaunchKernelGGL(large_potrf_inv_tiles<T>, dim3(gb), dim3(256), 0, g.st, g.dv, g.v, 0, g.skip);
                                for (int k = 0; k < NB; ++k)
                                {
                                        const int M = NB - k - 1;
                                        hipLaunchKernelGGL(large_right_step<0>, dim3(M * (M + 1) / 2 + NB * M + NB, 1, gb), dim3(256), 0, g.st, g.dv, g.v, k, g.skip);
                                }
                                vv.G = g.v.Vw; // V is there, row n = q included
                        }
                        else
                                for (int k = 0; k < NB; ++k)
                                {
                                        hipLaunchKernelGGL(large_potrf_inv_tiles<T>, dim3(gb), dim3(256), 0, g.st, g.dv, g.v, k, g.skip);
                                        if (k + 1 < NB)
                                                hipLaunchKernelGGL(large_update_panel<T>, dim3((NB - k) / 2, 1, gb), dim3(256), 0, g.st, g.dv, g.v, k, 1, g.skip);
                                }
                        if (pipe16)
                                launch_trsm_bf16(g.dv, g.v, gb, g.skip, g.st);
                        else if (!right) // (large_right_step has solved the rows of G on its way)
                                hipLaunchKernelGGL(large_trsm_pipe<LARGE_NB_MAX>, dim3(8 * ((gb + 7) / 8) * NB), dim3(256), 0, g.st, g.dv, g.v, gb, g.skip);
                        if (c->syrk_running)
                                hipLaunchKernelGGL((large_syrk_bf16x3<2>), syrk_grid, dim3(256), 0, g.st, g.dv, vv, LPlanes{nullptr}, gb, g.skip);
                        else
                                hipLaunchKernelGGL((large_syrk_bf16x3<0>), syrk_grid, dim3(256), 0, g.st, g.dv, vv, LPlanes{nullptr}, gb, g.skip);
                        // X += V q (+ the diagonal and the pose columns of V V^T in binary64) BEHIND the syrk on the same stream.  Round 4 tried the two ways of
                        // running it next to the syrk -- its workgroups inside the syrk launch, and on a side stream of its own (the two write disjoint entries
                        // of P) -- and both were slower: 2436 us against 1997 + 324 per 256 filters, and 31.7 k against 36.4 k filter-steps/s
                        // (profiles/r04_experiments.md section 1)
                        hipLaunchKernelGGL((large_x_update_rows<MODE>), dim3((NP + 4 * XU_ROWS - 1) / (4 * XU_ROWS), gb), dim3(256), 0, g.st, g.dv, vv, s, nsteps, g.poses,
                                           g.dims, g.skip);
                }
                else
                {
                        for (int k = 0; k < NB; ++k)
                        {
                                hipLaunchKernelGGL(large_potrf_inv_tiles<T>, dim3(gb), dim3(256), 0, g.st, g.dv, g.v, k, g.skip);
                                hipLaunchKernelGGL(large_update_panel<T>, dim3((2 * NB - k) / 2, 1, gb), dim3(256), 0, g.st, g.dv, g.v, k, 0, g.skip);
                        }
                        hipLaunchKernelGGL(large_syrk<T>, syrk_grid, dim3(256), 0, g.st, g.dv, g.v, gb, g.skip);
                }
                if constexpr (sizeof(T) == 8)
                        hipLaunchKernelGGL((large_x_update<T, MODE, false>), dim3((NP + 3) / 4, gb), dim3(256), 0, g.st, g.dv, g.v, s, nsteps, g.poses, g.dims,
                                           g.skip);
        };
        const int NG = c->large_groups;
        c->last_resident = (sizeof(T) == 4 && resident) ? 1 : 0;
        c->last_launches = sizeof(T) == 4 ? (resident ? 6 : (c->right_step && c->lv32.Vw ? 5 + NB : 4 + 2 * NB)) : 4 + 2 * NB;
        c->last_groups = 1;
        if (MODE == MODE_STEP && sa.traj >= 0)
        {
                Group g = make_group(sa.traj, 1, st);
                sa.traj = 0;
                for (int s = 0; s < nsteps; ++s)
                        chain(g, s);
        }
        else if (Bz < std::max(32, 8 * NG)) // (one group below 32 filters, as with the former default of four groups)
        {
                Group g = make_group(0, Bz, st);
                for (int s = 0; s < nsteps; ++s)
                        chain(g, s);
        }
        else
        {
                Group g[aslam_ctx::LARGE_GROUPS];
                const int per = ((Bz + NG - 1) / NG + 7) & ~7; // multiples of 8: large_syrk deals filters to the 8 XCDs
                for (int q = 0; q < NG; ++q)
                {
                        const int b0 = min(q * per, Bz);
                        g[q] = make_group(b0, min(per, Bz - b0), q == 0 ? st : c->aux[q - 1]);
                }
                c->last_groups = 0;
                for (int q = 0; q < NG; ++q)
                        c->last_groups += g[q].nb > 0;
                HIP_TRY(hipEventRecord(c->ev_fork, st));
                for (int q = 1; q < NG; ++q)
                        HIP_TRY(hipStreamWaitEvent(c->aux[q - 1], c->ev_fork, 0));
                for (int s = 0; s < nsteps; ++s)
                        for (int q = 0; q < NG; ++q)
                                if (g[q].nb > 0)
                                        chain(g[q], s);
                for (int q = 1; q < NG; ++q)
                {
                        HIP_TRY(hipEventRecord(c->ev_join[q - 1], c->aux[q - 1]));
                        HIP_TRY(hipStreamWaitEvent(st, c->ev_join[q - 1], 0));
                }
        }
        HIP_TRY(hipGetLastError());
        return ASLAM_OK;
}

template <int MODE>
int launch(aslam_ctx *c, int grid, int64_t t0, int nsteps, double *poses, int32_t *dims, StepArgs sa, hipStream_t st)
{
        if (c->large)
                return c->cfg.dtype == ASLAM_F32 ? launch_large_T<float, MODE>(c, c->lv32, grid, t0, nsteps, poses, dims, sa, st)
                                                 : launch_large_T<double, MODE>(c, c->lv64, grid, t0, nsteps, poses, dims, sa, st);
        if (c->cfg.filter == ASLAM_EKF)
        {
                switch (c->NT)
                {
                case 2:
                        return launch_ekf<2, MODE>(c, grid, t0, nsteps, poses, dims, sa, st);
                case 5:
                        return launch_ekf<5, MODE>(c, grid, t0, nsteps, poses, dims, sa, st);
                case 9:
                        return launch_ekf<9, MODE>(c, grid, t0, nsteps, poses, dims, sa, st);
                }
        }
#if ASLAM_HAVE_UKF
        if (c->cfg.filter == ASLAM_UKF)
        {
                switch (c->NT)
                {
                case 2:
                        return launch_ukf<2, MODE>(c, grid, t0, nsteps, poses, dims, sa, st);
                case 5:
                        return launch_ukf<5, MODE>(c, grid, t0, nsteps, poses, dims, sa, st);
                case 9:
                        return launch_ukf<9, MODE>(c, grid, t0, nsteps, poses, dims, sa, st);
                }
        }
#endif
        return fail(ASLAM_ERR_UNSUPPORTED, "no kernel for this filter/size");
}
} // namespace

extern "C" {

int aslam_abi_version(void)
{
        return ASLAM_ABI_VERSION;
}

const char *aslam_last_error(void)
{
        return g_err.c_str();
}

int aslam_create(const aslam_config *cfg, aslam_ctx **out)
{
        if (!cfg || !out)
                return fail(ASLAM_ERR_ARG, "null argument");
        *out = nullptr;
        if (cfg->filter != ASLAM_EKF && cfg->filter != ASLAM_UKF)
                return fail(ASLAM_ERR_ARG, "filter must be ASLAM_EKF or ASLAM_UKF");
        if (cfg->batch < 1 || cfg->max_landmark_count < 4 || cfg->max_obs < 1 || cfg->max_wait < 1)
                return fail(ASLAM_ERR_ARG, "batch, max_landmark_count, max_obs, max_wait must be positive");
        if (cfg->dtype != ASLAM_F64 && cfg->dtype != ASLAM_F32)
                return fail(ASLAM_ERR_ARG, "dtype must be ASLAM_F64 or ASLAM_F32");
        const int n_max = cfg->max_landmark_count - 1; // growth is refused at N >= MAX_LANDMARK_COUNT
        const int need = (n_max + 15) / 16;
        int NT = 0;
        for (int cand : {2, 5, 9})
        {
                if (cand >= need)
                {
                        NT = cand;
                        break;
                }
        }
        const bool large = (NT == 0) || cfg->dtype == ASLAM_F32; // fp32 exists on the multi-workgroup path only
        if (large)
        {
                if (cfg->filter != ASLAM_EKF)
                        return fail(ASLAM_ERR_UNSUPPORTED, "the UKF is limited to state dimensions up to 143 (single-CU kernels) and fp64");
                if (n_max + 1 > LARGE_NP_MAX)
                        return fail(ASLAM_ERR_UNSUPPORTED, "state dimension above 1087 is not supported");
                if (cfg->max_obs > LARGE_OBS_CAP || cfg->max_wait > LARGE_WAIT_CAP)
                        return fail(ASLAM_ERR_UNSUPPORTED, "max_obs above 1024 / max_wait above 2048 are not supported");
        }
        else
        {
                if (cfg->max_obs > SMALL_OBS_CAP)
                        return fail(ASLAM_ERR_UNSUPPORTED, "max_obs above 128 is not supported by the single-CU kernels");
                if (cfg->max_wait > SMALL_WAIT_CAP)
                        return fail(ASLAM_ERR_UNSUPPORTED, "max_wait above 512 is not supported by the single-CU kernels");
        }
        HIP_TRY(hipSetDevice(cfg->device));

        aslam_ctx *c = new aslam_ctx();
        c->cfg = *cfg;
        c->large = large;
        c->NT = large ? 0 : NT;
        c->NP = large ? ((n_max + 1 + LB - 1) / LB) * LB : 16 * NT; // the large path needs one spare row (Y^T rides in G)
        c->last_stream = nullptr;
        c->hbm_bytes = 0;
        std::memset(&c->dv, 0, sizeof(c->dv));
        DevView &d = c->dv;
        d.B = cfg->batch;
        d.NP = c->NP;
        d.dim_cap = cfg->max_landmark_count;
        d.max_obs = cfg->max_obs;
        d.max_wait = cfg->max_wait;
        const size_t B = (size_t)cfg->batch, NP = (size_t)c->NP;
        int rc = ASLAM_OK;
        auto A_ = [&](int r) {
                if (rc == ASLAM_OK)
                        rc = r;
        };
        A_(dev_alloc(c, &d.X, B * NP, c->owned));
        A_(dev_alloc(c, &d.Z, B * NP, c->owned));
        if (!large)
                A_(dev_alloc(c, &d.P, B * NP * NP, c->owned));
        else
        {
                A_(dev_alloc(c, &c->skipped, B, c->owned));
                if (const char *e = std::getenv("ASLAM_LARGE_GROUPS"))
                        c->large_groups = std::max(1, std::min((int)aslam_ctx::LARGE_GROUPS, std::atoi(e)));
                if (const char *e = std::getenv("ASLAM_CHOL_RESIDENT"))
                        c->chol_resident = std::atoi(e) != 0;
                if (const char *e = std::getenv("ASLAM_KEEP_L32"))
                        c->keep_l32 = std::atoi(e) != 0;
                if (const char *e = std::getenv("ASLAM_GS_TILES"))
                        c->gs_tiles = std::atoi(e) != 0;
                if (const char *e = std::getenv("ASLAM_SYRK_RUNNING"))
                        c->syrk_running = std::atoi(e) != 0;
                if (const char *e = std::getenv("ASLAM_RIGHT_STEP"))
                        c->right_step = std::atoi(e) != 0;
                if (const char *e = std::getenv("ASLAM_BF16_PIPE"))
                        c->bf16_pipe = std::atoi(e) & 3; // bit 0: the TRSM, bit 1: the Cholesky (diagnostics: 1 = large_chol_resident writes the planes, 2 = large_trsm_pipe solves)
                for (hipStream_t &q : c->aux)
                        if (hipStreamCreateWithFlags(&q, hipStreamNonBlocking) != hipSuccess)
                                rc = ASLAM_ERR_HIP;
                if (hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess)
                        rc = ASLAM_ERR_HIP;
                for (hipEvent_t &e : c->ev_join)
                        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
                                rc = ASLAM_ERR_HIP;
                if (cfg->dtype == ASLAM_F32)
                {
                        c->lv32.NP = c->NP;
                        A_(dev_alloc(c, &c->lv32.P, B * NP * NP, c->owned));
                        c->largeP = c->lv32.P;
                        A_(dev_alloc(c, &c->lv32.G, B * NP * NP, c->owned));
                        A_(dev_alloc(c, &c->lv32.S, B * NP * NP, c->owned));
                        A_(dev_alloc(c, &c->lv32.Hc, B * (NP / 2) * 4, c->owned));
                        A_(dev_alloc(c, &c->lv32.Y, B * NP, c->owned));
                        A_(dev_alloc(c, &c->lv32.Linv, B * LARGE_NB_MAX * LB * LB, c->owned));
                        if (c->bf16_pipe)
                                A_(dev_alloc(c, &c->lv32.Lpl, B * LPlanes::per_filter((int)NP), c->owned));
                        const bool may_be_resident = c->chol_resident >= 0 ? c->chol_resident != 0 : cfg->batch >= aslam_ctx::CHOL_RESIDENT_MIN_BATCH;
                        if (c->right_step && !may_be_resident)
                                A_(dev_alloc(c, &c->lv32.Vw, B * NP * NP, c->owned));
                }
                else
                {
                        c->lv64.NP = c->NP;
                        A_(dev_alloc(c, &c->lv64.P, B * NP * NP, c->owned));
                        c->largeP = c->lv64.P;
                        A_(dev_alloc(c, &c->lv64.G, B * NP * NP, c->owned));
                        A_(dev_alloc(c, &c->lv64.S, B * NP * NP, c->owned));
                        A_(dev_alloc(c, &c->lv64.Hc, B * (NP / 2) * 4, c->owned));
                        A_(dev_alloc(c, &c->lv64.Y, B * NP, c->owned));
                        A_(dev_alloc(c, &c->lv64.Linv, B * LARGE_NB_MAX * LB * LB, c->owned));
                }
        }
        A_(dev_alloc(c, &d.A, B * 2, c->owned));
        A_(dev_alloc(c, &c->step_in, B * 3, c->owned));
        d.step_in = c->step_in;
        A_(dev_alloc(c, &d.n, B, c->owned));
        A_(dev_alloc(c, &d.flags, B, c->owned));
        A_(dev_alloc(c, &d.status, B, c->owned));
        A_(dev_alloc(c, &d.sens, B * cfg->max_obs * 2, c->owned));
        A_(dev_alloc(c, &d.sens_n, B, c->owned));
        A_(dev_alloc(c, &d.wait_rb, B * cfg->max_wait * 2, c->owned));
        A_(dev_alloc(c, &d.wait_cnt, B * cfg->max_wait, c->owned));
        A_(dev_alloc(c, &d.wait_n, B, c->owned));
#ifdef ASLAM_STAMPS
        A_(dev_alloc(c, &d.dbg, 64 + 1024, c->owned)); // [64 ..): diagnostic builds, last front-end launch, 100 MHz ticks per workgroup
#endif
#if ASLAM_HAVE_UKF
        if (rc == ASLAM_OK && cfg->filter == ASLAM_UKF)
                rc = ukf_alloc(c);
#else
        if (rc == ASLAM_OK && cfg->filter == ASLAM_UKF)
                rc = fail(ASLAM_ERR_UNSUPPORTED, "library built without the UKF kernels");
#endif
        if (rc == ASLAM_OK)
                rc = init_state(c);
        if (rc != ASLAM_OK)
        {
                std::string keep = g_err;
                aslam_destroy(c);
                g_err = keep;
                return rc;
        }
        *out = c;
        return ASLAM_OK;
}

int aslam_destroy(aslam_ctx *c)
{
        if (!c)
                return ASLAM_OK;
        for (void *p : c->owned)
                (void)hipFree(p);
        for (void *p : c->trace_owned)
                (void)hipFree(p);
        for (hipStream_t q : c->aux)
                if (q)
                        (void)hipStreamDestroy(q);
        if (c->ev_fork)
                (void)hipEventDestroy(c->ev_fork);
        for (hipEvent_t e : c->ev_join)
                if (e)
                        (void)hipEventDestroy(e);
        delete c;
        return ASLAM_OK;
}

int aslam_reset(aslam_ctx *c)
{
        if (!c)
                return fail(ASLAM_ERR_ARG, "null context");
        int rc = sync_ctx(c);
        if (rc != ASLAM_OK)
                return rc;
        return init_state(c);
}

int aslam_set_state(aslam_ctx *c, int traj, int n, const double *X, const double *Z, const double *P)
{
        int rc = check_traj(c, traj);
        if (rc != ASLAM_OK)
                return rc;
        if (n < 3 || n >= c->cfg.max_landmark_count || ((n - 3) & 1))
                return fail(ASLAM_ERR_ARG, "n must be 3 + 2k and below max_landmark_count");
        rc = sync_ctx(c);
        if (rc != ASLAM_OK)
                return rc;
        const size_t NP = (size_t)c->NP;
        DevView &d = c->dv;
        if (X)
        {
                std::vector<double> v(NP, 0.0);
                std::memcpy(v.data(), X, sizeof(double) * n);
                HIP_TRY(hipMemcpy(d.X + traj * NP, v.data(), sizeof(double) * NP, hipMemcpyHostToDevice));
        }
        if (Z)
        {
                std::vector<double> v(NP, 0.0);
                std::memcpy(v.data(), Z, sizeof(double) * n);
                HIP_TRY(hipMemcpy(d.Z + traj * NP, v.data(), sizeof(double) * NP, hipMemcpyHostToDevice));
        }
        if (P)
        {
                rc = upload_P(c, traj, n, P);
                if (rc != ASLAM_OK)
                        return rc;
        }
        const int fl = 0; // a filter whose state was handed over is past both init flags
        HIP_TRY(hipMemcpy(d.n + traj, &n, sizeof(int), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d.flags + traj, &fl, sizeof(int), hipMemcpyHostToDevice));
        return ASLAM_OK;
}

int aslam_grow(aslam_ctx *c, int traj, int n_new, const double *x_seed, const double *z_seed)
{
        int rc = check_traj(c, traj);
        if (rc != ASLAM_OK)
                return rc;
        rc = sync_ctx(c);
        if (rc != ASLAM_OK)
                return rc;
        DevView &d = c->dv;
        int n_old = 0;
        HIP_TRY(hipMemcpy(&n_old, d.n + traj, sizeof(int), hipMemcpyDeviceToHost));
        if (n_new <= n_old || ((n_new - n_old) & 1) || !x_seed || !z_seed)
                return fail(ASLAM_ERR_ARG, "n_new must exceed the current dimension by an even amount; seeds required");
        if (n_new >= c->cfg.max_landmark_count)
        {
                // ekf.cpp:263-268: warn, keep N, drop the landmarks
                uint32_t st = 0;
                HIP_TRY(hipMemcpy(&st, d.status + traj, sizeof(st), hipMemcpyDeviceToHost));
                st |= ASLAM_ST_GROWTH_REFUSED;
                HIP_TRY(hipMemcpy(d.status + traj, &st, sizeof(st), hipMemcpyHostToDevice));
                return ASLAM_OK;
        }
        const size_t NP = (size_t)c->NP;
        const int k = n_new - n_old;
        HIP_TRY(hipMemcpy(d.X + traj * NP + n_old, x_seed, sizeof(double) * k, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(d.Z + traj * NP + n_old, z_seed, sizeof(double) * k, hipMemcpyHostToDevice));
        // conservativeResizeLike(Identity * UKF_KP_LANDMARK_POSE): new rows (new columns of old rows are zero padding already)
        rc = grow_P_rows(c, traj, n_old, n_new);
        if (rc != ASLAM_OK)
                return rc;
        HIP_TRY(hipMemcpy(d.n + traj, &n_new, sizeof(int), hipMemcpyHostToDevice));
        return ASLAM_OK;
}

int aslam_ekf_step(aslam_ctx *c, int traj, float vx, float az, float dt, const double *Z, double a00, double a10,
                   double *X_out, void *stream)
{
        int rc = check_traj(c, traj);
        if (rc != ASLAM_OK)
                return rc;
        if (c->cfg.filter != ASLAM_EKF)
                return fail(ASLAM_ERR_STATE, "context was created for the UKF");
        if (!Z)
                return fail(ASLAM_ERR_ARG, "Z is required");
        hipStream_t st = static_cast<hipStream_t>(stream);
        c->last_stream = st;
        DevView &d = c->dv;
        int n = 0;
        HIP_TRY(hipMemcpyAsync(&n, d.n + traj, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const size_t NP = (size_t)c->NP;
        const double A[2] = {a00, a10};
        HIP_TRY(hipMemcpyAsync(d.Z + traj * NP, Z, sizeof(double) * n, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(d.A + 2 * traj, A, sizeof(A), hipMemcpyHostToDevice, st));
        StepArgs sa{traj, vx, az, dt};
        rc = launch<MODE_STEP>(c, 1, 0, 1, nullptr, nullptr, sa, st);
        if (rc != ASLAM_OK)
                return rc;
        if (X_out)
        {
                HIP_TRY(hipMemcpyAsync(X_out, d.X + traj * NP, sizeof(double) * n, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
        }
        return ASLAM_OK;
}

int aslam_ukf_step(aslam_ctx *c, int traj, float vx, float az, float dt, const double *Z, double *X_out, void *stream)
{
        int rc = check_traj(c, traj);
        if (rc != ASLAM_OK)
                return rc;
        if (c->cfg.filter != ASLAM_UKF)
                return fail(ASLAM_ERR_STATE, "context was created for the EKF");
        if (!Z)
                return fail(ASLAM_ERR_ARG, "Z is required");
        hipStream_t st = static_cast<hipStream_t>(stream);
        c->last_stream = st;
        DevView &d = c->dv;
        int n = 0;
        HIP_TRY(hipMemcpyAsync(&n, d.n + traj, sizeof(int), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        const size_t NP = (size_t)c->NP;
        HIP_TRY(hipMemcpyAsync(d.Z + traj * NP, Z, sizeof(double) * n, hipMemcpyHostToDevice, st));
        StepArgs sa{traj, vx, az, dt};
        rc = launch<MODE_STEP>(c, 1, 0, 1, nullptr, nullptr, sa, st);
        if (rc != ASLAM_OK)
                return rc;
        if (X_out)
        {
                HIP_TRY(hipMemcpyAsync(X_out, d.X + traj * NP, sizeof(double) * n, hipMemcpyDeviceToHost, st));
                HIP_TRY(hipStreamSynchronize(st));
        }
        return ASLAM_OK;
}

namespace
{
/// the batched per-callback seam: inputs of all filters to the device (asynchronously, straight from the caller's arrays), one launch
/// chain for the whole batch, optional read-back of X; no synchronisation
int step_batch(aslam_ctx *c, int filter, const float *vx, const float *az, const float *dt, const double *Z, int ldz, const double *a00,
               const double *a10, double *X_out, int ldx, void *stream)
{
        if (!c)
                return fail(ASLAM_ERR_ARG, "null context");
        if (c->cfg.filter != filter)
                return fail(ASLAM_ERR_STATE, "context was created for the other filter");
        if (!vx || !az || !dt || !Z || (filter == ASLAM_EKF && (!a00 || !a10)))
                return fail(ASLAM_ERR_ARG, "vx, az, dt, Z (and a00, a10 for the EKF) are required");
        const int B = c->cfg.batch, NP = c->NP;
        if (ldz < 3 || (X_out && ldx < 3))
                return fail(ASLAM_ERR_ARG, "row strides must cover the state");
        hipStream_t st = static_cast<hipStream_t>(stream);
        c->last_stream = st;
        DevView &d = c->dv;
        HIP_TRY(hipMemcpyAsync(c->step_in, vx, sizeof(float) * B, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(c->step_in + B, az, sizeof(float) * B, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(c->step_in + 2 * (size_t)B, dt, sizeof(float) * B, hipMemcpyHostToDevice, st));
        const size_t wz = sizeof(double) * (size_t)std::min(ldz, NP);
        HIP_TRY(hipMemcpy2DAsync(d.Z, sizeof(double) * NP, Z, sizeof(double) * ldz, wz, B, hipMemcpyHostToDevice, st));
        if (filter == ASLAM_EKF)
        {
                HIP_TRY(hipMemcpy2DAsync(d.A, 2 * sizeof(double), a00, sizeof(double), sizeof(double), B, hipMemcpyHostToDevice, st));
                HIP_TRY(hipMemcpy2DAsync(d.A + 1, 2 * sizeof(double), a10, sizeof(double), sizeof(double), B, hipMemcpyHostToDevice, st));
        }
        StepArgs sa{-1, 0.f, 0.f, 0.f};
        int rc = launch<MODE_STEP>(c, B, 0, 1, nullptr, nullptr, sa, st);
        if (rc != ASLAM_OK)
                return rc;
        if (X_out)
                HIP_TRY(hipMemcpy2DAsync(X_out, sizeof(double) * ldx, d.X, sizeof(double) * NP, sizeof(double) * (size_t)std::min(ldx, NP), B,
                                         hipMemcpyDeviceToHost, st));
        return ASLAM_OK;
}
} // namespace

int aslam_ekf_step_batch(aslam_ctx *c, const float *vx, const float *az, const float *dt, const double *Z, int ldz, const double *a00,
                         const double *a10, double *X_out, int ldx, void *stream)
{
        return step_batch(c, ASLAM_EKF, vx, az, dt, Z, ldz, a00, a10, X_out, ldx, stream);
}

int aslam_ukf_step_batch(aslam_ctx *c, const float *vx, const float *az, const float *dt, const double *Z, int ldz, double *X_out, int ldx,
                         void *stream)
{
        return step_batch(c, ASLAM_UKF, vx, az, dt, Z, ldz, nullptr, nullptr, X_out, ldx, stream);
}

int aslam_set_trace(aslam_ctx *c, const aslam_trace *tr)
{
        if (!c || !tr)
                return fail(ASLAM_ERR_ARG, "null argument");
        if (tr->T < 1 || tr->max_obs != c->cfg.max_obs)
                return fail(ASLAM_ERR_ARG, "trace must have T >= 1 and the context's max_obs");
        if (!tr->pose || !tr->yaw || !tr->twist || !tr->dt || !tr->obs_new || !tr->n_obs || !tr->obs)
                return fail(ASLAM_ERR_ARG, "trace arrays must all be given");
        int rc = sync_ctx(c);
        if (rc != ASLAM_OK)
                return rc;
        for (void *p : c->trace_owned)
                (void)hipFree(p);
        c->trace_owned.clear();
        DevView &d = c->dv;
        d.T = tr->T;
        const size_t BT = (size_t)c->cfg.batch * (size_t)tr->T;
        if (tr->is_device)
        {
                d.tr_pose = tr->pose;
                d.tr_yaw = tr->yaw;
                d.tr_twist = tr->twist;
                d.tr_dt = tr->dt;
                d.tr_new = tr->obs_new;
                d.tr_nobs = tr->n_obs;
                d.tr_obs = tr->obs;
                return ASLAM_OK;
        }
        auto up = [&](const void *src, size_t bytes, const void **dst) -> int {
                void *q = nullptr;
                HIP_TRY(hipMalloc(&q, bytes));
                c->trace_owned.push_back(q);
                HIP_TRY(hipMemcpy(q, src, bytes, hipMemcpyHostToDevice));
                *dst = q;
                return ASLAM_OK;
        };
        const void *p = nullptr;
#define UP(field, src, bytes)                                                                                          \
        rc = up(src, bytes, &p);                                                                                       \
        if (rc != ASLAM_OK)                                                                                            \
                return rc;                                                                                             \
        d.field = static_cast<decltype(d.field)>(p);
        UP(tr_pose, tr->pose, BT * 2 * sizeof(double));
        UP(tr_yaw, tr->yaw, BT * sizeof(float));
        UP(tr_twist, tr->twist, BT * 2 * sizeof(double));
        UP(tr_dt, tr->dt, BT * sizeof(float));
        UP(tr_new, tr->obs_new, BT * sizeof(uint8_t));
        UP(tr_nobs, tr->n_obs, BT * sizeof(int32_t));
        UP(tr_obs, tr->obs, BT * (size_t)tr->max_obs * 2 * sizeof(float));
#undef UP
        return ASLAM_OK;
}

int aslam_replay(aslam_ctx *c, int64_t t0, int64_t nsteps, double *poses_out, int32_t *dims_out, void *stream)
{
        if (!c)
                return fail(ASLAM_ERR_ARG, "null context");
        if (!c->dv.tr_pose)
                return fail(ASLAM_ERR_STATE, "no trace bound (aslam_set_trace)");
        if (t0 < 0 || nsteps < 1 || t0 + nsteps > c->dv.T || nsteps > 0x7fffffff)
                return fail(ASLAM_ERR_ARG, "step range outside the bound trace");
        hipStream_t st = static_cast<hipStream_t>(stream);
        c->last_stream = st;
        StepArgs sa{0, 0.f, 0.f, 0.f};
        return launch<MODE_REPLAY>(c, c->cfg.batch, t0, (int)nsteps, poses_out, dims_out, sa, st);
}

int aslam_get_dim(aslam_ctx *c, int traj, int *n)
{
        int rc = check_traj(c, traj);
        if (rc != ASLAM_OK)
                return rc;
        if (!n)
                return fail(ASLAM_ERR_ARG, "null output");
        rc = sync_ctx(c);
        if (rc != ASLAM_OK)
                return rc;
        HIP_TRY(hipMemcpy(n, c->dv.n + traj, sizeof(int), hipMemcpyDeviceToHost));
        return ASLAM_OK;
}

int aslam_get_state(aslam_ctx *c, int traj, double *X, double *Z, double *P)
{
        int n = 0;
        int rc = aslam_get_dim(c, traj, &n);
        if (rc != ASLAM_OK)
                return rc;
        const size_t NP = (size_t)c->NP;
        DevView &d = c->dv;
        if (X)
                HIP_TRY(hipMemcpy(X, d.X + traj * NP, sizeof(double) * n, hipMemcpyDeviceToHost));
        if (Z)
                HIP_TRY(hipMemcpy(Z, d.Z + traj * NP, sizeof(double) * n, hipMemcpyDeviceToHost));
        if (P)
                return download_P(c, traj, n, P);
        return ASLAM_OK;
}

int aslam_get_A(aslam_ctx *c, int traj, double *a00, double *a10)
{
        int rc = check_traj(c, traj);
        if (rc != ASLAM_OK)
                return rc;
        rc = sync_ctx(c);
        if (rc != ASLAM_OK)
                return rc;
        double A[2];
        HIP_TRY(hipMemcpy(A, c->dv.A + 2 * traj, sizeof(A), hipMemcpyDeviceToHost));
        if (a00)
                *a00 = A[0];
        if (a10)
                *a10 = A[1];
        return ASLAM_OK;
}

int aslam_get_landmarks(aslam_ctx *c, int traj, double *x, double *y, int *n_landmarks)
{
        int n = 0;
        int rc = aslam_get_dim(c, traj, &n);
        if (rc != ASLAM_OK)
                return rc;
        std::vector<double> X(n);
        HIP_TRY(hipMemcpy(X.data(), c->dv.X + (size_t)traj * c->NP, sizeof(double) * n, hipMemcpyDeviceToHost));
        const int L = (n - 3) / 2;
        for (int i = 0; i < L; ++i)
        {
                if (x)
                        x[i] = X[3 + 2 * i];
                if (y)
                        y[i] = X[4 + 2 * i];
        }
        if (n_landmarks)
                *n_landmarks = L;
        return ASLAM_OK;
}

int aslam_get_wait(aslam_ctx *c, int traj, float *range, float *bearing, uint32_t *count, int cap, int *size)
{
        int rc = check_traj(c, traj);
        if (rc != ASLAM_OK)
                return rc;
        rc = sync_ctx(c);
        if (rc != ASLAM_OK)
                return rc;
        DevView &d = c->dv;
        int wn = 0;
        HIP_TRY(hipMemcpy(&wn, d.wait_n + traj, sizeof(int), hipMemcpyDeviceToHost));
        if (size)
                *size = wn;
        const int k = wn < cap ? wn : cap;
        if (k > 0)
        {
                std::vector<float> rb(2 * (size_t)k);
                std::vector<uint32_t> cn(k);
                HIP_TRY(hipMemcpy(rb.data(), d.wait_rb + (size_t)traj * d.max_wait * 2, sizeof(float) * 2 * k, hipMemcpyDeviceToHost));
                HIP_TRY(hipMemcpy(cn.data(), d.wait_cnt + (size_t)traj * d.max_wait, sizeof(uint32_t) * k, hipMemcpyDeviceToHost));
                for (int i = 0; i < k; ++i)
                {
                        if (range)
                                range[i] = rb[2 * i];
                        if (bearing)
                                bearing[i] = rb[2 * i + 1];
                        if (count)
                                count[i] = cn[i];
                }
        }
        return ASLAM_OK;
}

int aslam_get_status(aslam_ctx *c, int traj, uint32_t *status_bits)
{
        int rc = check_traj(c, traj);
        if (rc != ASLAM_OK)
                return rc;
        if (!status_bits)
                return fail(ASLAM_ERR_ARG, "null output");
        rc = sync_ctx(c);
        if (rc != ASLAM_OK)
                return rc;
        HIP_TRY(hipMemcpy(status_bits, c->dv.status + traj, sizeof(uint32_t), hipMemcpyDeviceToHost));
        return ASLAM_OK;
}

int aslam_get_layout(aslam_ctx *c, int *padded_dim, int64_t *hbm_bytes)
{
        if (!c)
                return fail(ASLAM_ERR_ARG, "null context");
        if (padded_dim)
                *padded_dim = c->NP;
        if (hbm_bytes)
                *hbm_bytes = c->hbm_bytes;
        return ASLAM_OK;
}

/* diagnostic (tests/manual/large_residuals.py): the work matrices of the large path after the last callback, widened to double.
   which: 0 = G (-> V in place) [NP][NP], 1 = S (-> L, lower block triangle) [NP][NP], 2 = Linv [17][64][64], 3 = Y [NP] */
int aslam_debug_large(aslam_ctx *c, int traj, int which, double *out, int64_t cap)
{
        if (check_traj(c, traj) != ASLAM_OK)
                return ASLAM_ERR_ARG;
        if (!c->large || which < 0 || which > 3)
                return fail(ASLAM_ERR_UNSUPPORTED, "aslam_debug_large: large-state contexts only");
        if (sync_ctx(c) != ASLAM_OK)
                return ASLAM_ERR_HIP;
        const size_t NP = c->NP;
        const size_t cnt = which < 2 ? NP * NP : which == 2 ? (size_t)LARGE_NB_MAX * LB * LB : NP;
        if ((int64_t)cnt > cap)
                return fail(ASLAM_ERR_ARG, "aslam_debug_large: buffer too small");
        if (which == 3)
        {
                const double *src = (c->cfg.dtype == ASLAM_F32 ? c->lv32.Y : c->lv64.Y) + traj * NP;
                HIP_TRY(hipMemcpy(out, src, cnt * sizeof(double), hipMemcpyDeviceToHost));
                return ASLAM_OK;
        }
        if (c->cfg.dtype == ASLAM_F32)
        {
                const float *src = which == 0 ? c->lv32.G + traj * NP * NP : which == 1 ? c->lv32.S + traj * NP * NP
                                                                                      : c->lv32.Linv + traj * cnt;
                std::vector<float> tmp(cnt);
                HIP_TRY(hipMemcpy(tmp.data(), src, cnt * sizeof(float), hipMemcpyDeviceToHost));
                for (size_t i = 0; i < cnt; ++i)
                        out[i] = (double)tmp[i];
        }
        else
        {
                const double *src = which == 0 ? c->lv64.G + traj * NP * NP : which == 1 ? c->lv64.S + traj * NP * NP
                                                                                       : c->lv64.Linv + traj * cnt;
                HIP_TRY(hipMemcpy(out, src, cnt * sizeof(double), hipMemcpyDeviceToHost));
        }
        return ASLAM_OK;
}

#if defined(ASLAM_STAMPS) && ASLAM_HAVE_UKF
/* diagnostic builds only: copy a UKF scratch matrix of filter `traj` to the host. which: 0 D, 1 DZ ([NP][MP]), 2 Tc, 3 K ([NP][NP]) */
int aslam_debug_ukf(aslam_ctx *c, int traj, int which, double *out, int *rows, int *cols)
{
        if (sync_ctx(c) != ASLAM_OK)
                return ASLAM_ERR_HIP;
        const size_t NP = c->NP, MP = c->ukf.MP;
        const double *src = which == 0 ? c->ukf.D + traj * NP * MP : which == 1 ? c->ukf.DZ + traj * NP * MP
                            : which == 2 ? c->ukf.Tc + traj * NP * NP : c->ukf.K + traj * NP * NP;
        const size_t cnt = which < 2 ? NP * MP : NP * NP;
        *rows = (int)NP;
        *cols = which < 2 ? (int)MP : (int)NP;
        HIP_TRY(hipMemcpy(out, src, cnt * sizeof(double), hipMemcpyDeviceToHost));
        return ASLAM_OK;
}
#endif

#ifdef ASLAM_STAMPS
/* diagnostic builds only: per-phase shader-cycle sums of workgroup 0 since the context was created */
int aslam_debug_stamps(aslam_ctx *c, unsigned long long *out12)
{
        if (sync_ctx(c) != ASLAM_OK)
                return ASLAM_ERR_HIP;
        HIP_TRY(hipMemcpy(out12, c->dv.dbg, 12 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        return ASLAM_OK;
}

/* diagnostic builds only: 100 MHz ticks spent inside large_frontend_kernel (workgroup 0) and the number of launches */
int aslam_debug_fe_realtime(aslam_ctx *c, unsigned long long *out2)
{
        if (sync_ctx(c) != ASLAM_OK)
                return ASLAM_ERR_HIP;
        HIP_TRY(hipMemcpy(out2, c->dv.dbg + 56, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        return ASLAM_OK;
}

/* diagnostic builds only: 100 MHz ticks every workgroup (filter) spent inside the last large_frontend_kernel launch */
int aslam_debug_fe_per_filter(aslam_ctx *c, unsigned long long *out, int count)
{
        if (sync_ctx(c) != ASLAM_OK)
                return ASLAM_ERR_HIP;
        HIP_TRY(hipMemcpy(out, c->dv.dbg + 64, (size_t)std::min(count, 1024) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        return ASLAM_OK;
}

/* diagnostic builds only (-DASLAM_FE_STAMPS): front-end phase cycles, 6 values */
int aslam_debug_fe_stamps(aslam_ctx *c, unsigned long long *out6)
{
        if (sync_ctx(c) != ASLAM_OK)
                return ASLAM_ERR_HIP;
        HIP_TRY(hipMemcpy(out6, c->dv.dbg + 45, 6 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        return ASLAM_OK;
}

/* diagnostic builds only: per-wave busy cycles inside cholesky_forward: [wave][panel, trailing/forward/factor] */
int aslam_debug_wave_busy(aslam_ctx *c, unsigned long long *out24)
{
        if (sync_ctx(c) != ASLAM_OK)
                return ASLAM_ERR_HIP;
        HIP_TRY(hipMemcpy(out24, c->dv.dbg + 16, 24 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        return ASLAM_OK;
}
#endif

/* ---- include/aslam_scan.h ------------------------------------------------------------------------------------------ */
namespace
{
constexpr int MAX_SCAN_DEVICES = 16;
static float *g_scan_tables[MAX_SCAN_DEVICES] = {}; // [2][360]: cos_map, sin_map of LandMarks::initialize (sensor_landmark.cpp:49-56)

static int scan_tables(int device, float **out)
{
        if (device < 0 || device >= MAX_SCAN_DEVICES)
                return fail(ASLAM_ERR_ARG, "device index out of range");
        static std::mutex mu;
        std::lock_guard<std::mutex> lock(mu);
        if (!g_scan_tables[device])
        {
                // the reference fills them with the host libm's binary32 sin / cos of DEG2RAD * float(theta): so does this
                const float DEG2RAD = 0.01745329251f; // config.h:41
                float tab[2 * SCAN_BEAMS];
                for (int t = 0; t < SCAN_BEAMS; ++t)
                {
                        tab[t] = std::cos(DEG2RAD * static_cast<float>(t));
                        tab[SCAN_BEAMS + t] = std::sin(DEG2RAD * static_cast<float>(t));
                }
                float *p = nullptr;
                HIP_TRY(hipMalloc(reinterpret_cast<void **>(&p), sizeof(tab)));
                HIP_TRY(hipMemcpy(p, tab, sizeof(tab), hipMemcpyHostToDevice));
                g_scan_tables[device] = p;
        }
        *out = g_scan_tables[device];
        return ASLAM_OK;
}
} // namespace

int aslam_scan_landmarks(const float *ranges, int64_t count, int is_device, int max_out, float *range_out, float *bearing_out,
                         int32_t *n_out, uint32_t *status_out, int device, void *stream)
{
        if (!ranges || !range_out || !bearing_out || !n_out || !status_out || count < 0 || max_out <= 0)
                return fail(ASLAM_ERR_ARG, "aslam_scan_landmarks: bad argument");
        if (count == 0)
                return ASLAM_OK;
        HIP_TRY(hipSetDevice(device));
        float *tab = nullptr;
        int rc = scan_tables(device, &tab);
        if (rc != ASLAM_OK)
                return rc;
        hipStream_t st = static_cast<hipStream_t>(stream);
        const size_t n = (size_t)count;
        if (is_device)
        {
                hipLaunchKernelGGL(scan_landmarks_kernel, dim3((unsigned)count), dim3(64), 0, st, ranges, tab, tab + SCAN_BEAMS, count, max_out,
                                   range_out, bearing_out, n_out, status_out);
                HIP_TRY(hipGetLastError());
                return ASLAM_OK;
        }
        float *d_in = nullptr, *d_r = nullptr, *d_b = nullptr;
        int32_t *d_n = nullptr;
        uint32_t *d_s = nullptr;
        auto cleanup = [&]() {
                (void)hipFree(d_in), (void)hipFree(d_r), (void)hipFree(d_b), (void)hipFree(d_n), (void)hipFree(d_s);
        };
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_in), n * SCAN_BEAMS * sizeof(float));
        if (e == hipSuccess)
                e = hipMalloc(reinterpret_cast<void **>(&d_r), n * max_out * sizeof(float));
        if (e == hipSuccess)
                e = hipMalloc(reinterpret_cast<void **>(&d_b), n * max_out * sizeof(float));
        if (e == hipSuccess)
                e = hipMalloc(reinterpret_cast<void **>(&d_n), n * sizeof(int32_t));
        if (e == hipSuccess)
                e = hipMalloc(reinterpret_cast<void **>(&d_s), n * sizeof(uint32_t));
        if (e == hipSuccess)
                e = hipMemcpyAsync(d_in, ranges, n * SCAN_BEAMS * sizeof(float), hipMemcpyHostToDevice, st);
        if (e == hipSuccess)
                e = hipMemsetAsync(d_r, 0, n * max_out * sizeof(float), st);
        if (e == hipSuccess)
                e = hipMemsetAsync(d_b, 0, n * max_out * sizeof(float), st);
        if (e == hipSuccess)
        {
                hipLaunchKernelGGL(scan_landmarks_kernel, dim3((unsigned)count), dim3(64), 0, st, d_in, tab, tab + SCAN_BEAMS, count, max_out, d_r,
                                   d_b, d_n, d_s);
                e = hipGetLastError();
        }
        if (e == hipSuccess)
                e = hipMemcpyAsync(range_out, d_r, n * max_out * sizeof(float), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
                e = hipMemcpyAsync(bearing_out, d_b, n * max_out * sizeof(float), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
                e = hipMemcpyAsync(n_out, d_n, n * sizeof(int32_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
                e = hipMemcpyAsync(status_out, d_s, n * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess)
                e = hipStreamSynchronize(st);
        cleanup();
        if (e != hipSuccess)
                return fail(ASLAM_ERR_HIP, std::string("aslam_scan_landmarks: ") + hipGetErrorString(e));
        return ASLAM_OK;
}

int aslam_get_launch_info(aslam_ctx *c, int *stream_groups, int *chol_resident, int *launches_per_callback)
{
        if (!c)
                return fail(ASLAM_ERR_ARG, "null context");
        if (stream_groups)
                *stream_groups = c->large ? c->last_groups : 0;
        if (chol_resident)
                *chol_resident = c->large ? c->last_resident : 0;
        if (launches_per_callback)
                *launches_per_callback = c->large ? c->last_launches : 1;
        return ASLAM_OK;
}

int aslam_kernel_info(aslam_ctx *c, char *name, int name_cap, int *grid, int *block, int *lds_bytes)
{
        if (!c)
                return fail(ASLAM_ERR_ARG, "null context");
        char buf[192];
        size_t lds = 0;
        if (c->large)
        {
                const bool resident = c->chol_resident >= 0 ? c->chol_resident != 0 : c->cfg.batch >= aslam_ctx::CHOL_RESIDENT_MIN_BATCH;
                if (c->cfg.dtype == ASLAM_F32 && resident)
                {
                        // the names say which pipe each kernel's products run on (bench.py prices them from this string): bit 1 of bf16_pipe = the
                        // Cholesky, bit 0 = the TRSM on the bf16 matrix pipe; the X update rides in the syrk launch
                        const bool chol16 = c->lv32.Lpl && (c->bf16_pipe & 2), trsm16 = c->lv32.Lpl && (c->bf16_pipe & 1);
                        std::snprintf(buf, sizeof(buf), "%s + %s + large_syrk_bf16x3 (6-launch chain per callback, %d stream groups)",
                                      chol16 ? "large_chol_bf16" : "large_chol_resident", trsm16 ? "large_trsm_bf16" : "large_trsm_pipe<17>", c->large_groups);
                }
                else if (c->cfg.dtype == ASLAM_F32 && c->right_step && c->lv32.Vw)
                        std::snprintf(buf, sizeof(buf), "large_right_step + large_syrk_bf16x3 (%d-launch chain per callback: one right-looking launch per block column)", 5 + c->NP / LB);
                else if (c->cfg.dtype == ASLAM_F32)
                        std::snprintf(buf, sizeof(buf), "large_trsm_pipe<%d> + large_syrk_bf16x3 (%d-launch chain per callback: multi-workgroup Cholesky)",
                                      (int)LARGE_NB_MAX, 4 + 2 * (c->NP / LB));
                else
                        std::snprintf(buf, sizeof(buf), "large_update_panel<double> (%d-launch chain per callback, %d stream groups)",
                                      4 + 2 * (c->NP / LB), c->large_groups);
                lds = LargeLds::bytes(c->NP);
        }
        else if (c->cfg.filter == ASLAM_EKF)
        {
                std::snprintf(buf, sizeof(buf), "ekf_small_kernel<%d,0>", c->NT);
                lds = c->NT == 2 ? SmallLayout<2>::total : c->NT == 5 ? SmallLayout<5>::total : SmallLayout<9>::total;
        }
#if ASLAM_HAVE_UKF
        else
        {
                std::snprintf(buf, sizeof(buf), "ukf_small_kernel<%d,0>", c->NT);
                lds = c->NT == 2 ? UkfLayout<2>::total : c->NT == 5 ? UkfLayout<5>::total : UkfLayout<9>::total;
        }
#endif
        if (name && name_cap > 0)
        {
                std::strncpy(name, buf, name_cap - 1);
                name[name_cap - 1] = 0;
        }
        if (grid)
                *grid = c->cfg.batch;
        if (block)
                *block = SMALL_WG;
        if (lds_bytes)
                *lds_bytes = (int)lds;
        return ASLAM_OK;
}

} // extern "C"
