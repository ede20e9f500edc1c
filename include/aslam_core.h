/*
 * aslam_core.h -- C ABI of the MI355X-native EKF/UKF-SLAM predict/update core (libaslam_core.so).
 *
 * The reference (iamarkaj/AwesomeSLAM) has no plugin/FFI API: its filter step is the private method
 *     aslam::EKFSlam::slam(vx, az, dt)   awesome_slam/src/ekf/ekf.h:130-131, ekf.cpp:293-311
 *     aslam::UKFSlam::slam(vx, az, dt)   awesome_slam/src/ukf/ukf.h:142-143, ukf.cpp:260-392
 * operating on `Parameters` (ekf.h:56-67, ukf.h:56-82).  This header is the seam a maintainer would cut
 * there: the entry points below replace, one for one, the members of EKFSlam/UKFSlam named beside them.
 * INTEGRATION.md shows the reference-side binding.
 *
 * Conventions: plain C types, caller-owned buffers, `int` status returns (0 = ASLAM_OK, negative =
 * error, text via aslam_last_error()); no exceptions cross the seam.  A context owns all filter state
 * for `batch` independent filters ("trajectories") in HBM; matrices are row-major.  One HIP stream per
 * call (the `stream` argument is a hipStream_t passed as void*, NULL = the default stream); a context
 * is not re-entrant.  Everything is fp64 unless the context was created with ASLAM_F32.
 */
#ifndef ASLAM_CORE_H
#define ASLAM_CORE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASLAM_ABI_VERSION 1

enum
{
        ASLAM_OK = 0,
        ASLAM_ERR_ARG = -1,      /* bad argument */
        ASLAM_ERR_HIP = -2,      /* a HIP runtime call failed (no device, out of memory, launch failure) */
        ASLAM_ERR_UNSUPPORTED = -3, /* configuration outside what the kernels cover */
        ASLAM_ERR_STATE = -4     /* call sequence error (e.g. replay without a trace) */
};

enum
{
        ASLAM_EKF = 0, /* rosrun awesome_slam ekf */
        ASLAM_UKF = 1  /* rosrun awesome_slam ukf */
};

enum
{
        ASLAM_F64 = 0,
        ASLAM_F32 = 1
};

/* per-trajectory status bits, sticky until aslam_reset (aslam_get_status) */
enum
{
        ASLAM_ST_GROWTH_REFUSED = 1, /* updateNewLandmark hit N >= MAX_LANDMARK_COUNT (ekf.cpp:263-268): landmarks dropped */
        ASLAM_ST_WAIT_OVERFLOW = 2,  /* wait-list capacity (max_wait) exceeded: an entry was dropped (deviation!) */
        ASLAM_ST_NOT_PD = 4,         /* a Cholesky pivot was <= 0 (the reference would go on with garbage, ukf.cpp:280) */
        ASLAM_ST_OBS_OVERFLOW = 8,   /* a sensor message had more entries than max_obs */
        ASLAM_ST_INTERNAL = 16       /* an on-chip synchronisation of the large-state kernels timed out (bounded spin instead of a hung GPU):
                                        this filter's state is invalid from that callback on */
};

typedef struct aslam_ctx aslam_ctx;

typedef struct
{
        int32_t filter;             /* ASLAM_EKF | ASLAM_UKF */
        int32_t dtype;              /* ASLAM_F64 | ASLAM_F32 */
        int32_t max_landmark_count; /* config.h:45 MAX_LANDMARK_COUNT (30): growth is refused when the state
                                       DIMENSION N would reach it (ekf.cpp:263).  A run-time field here. */
        int32_t batch;              /* independent filters held by this context */
        int32_t max_obs;            /* capacity of the stored sensor message (sensor_landmark, ekf.h:102) */
        int32_t max_wait;           /* capacity of new_landmark_wait (ekf.h:105); the reference's is unbounded */
        int32_t device;             /* HIP device ordinal */
        int32_t reserved;
} aslam_config;

/* A recorded input stream for `batch` filters x T callbacks, already narrowed the way the node's
 * callbacks narrow their messages (cbOdom/updateZandA ekf.cpp:137-142, cbSensorLandmark ekf.cpp:102-114):
 *   pose   [batch][T][2]  f64  msg->pose.pose.position.{x,y}
 *   yaw    [batch][T]     f32  quat2euler(orientation)  (tools.h:62-66; done by the host mirror)
 *   twist  [batch][T][2]  f64  msg->twist.twist.linear.x, angular.z
 *   dt     [batch][T]     f32  delta_time (ekf.cpp:80)
 *   obs_new[batch][T]     u8   1 = a sensor message precedes this odom message
 *   n_obs  [batch][T]     i32
 *   obs    [batch][T][max_obs][2] f32  range, bearing (LaserData, structures.h:85-101)
 * Pointers may be host or device memory (is_device). */
typedef struct
{
        int64_t T;
        int32_t max_obs;
        int32_t is_device;
        const double *pose;
        const float *yaw;
        const double *twist;
        const float *dt;
        const uint8_t *obs_new;
        const int32_t *n_obs;
        const float *obs;
} aslam_trace;

/* ---- life cycle ---------------------------------------------------------------------------------- */
/* EKFSlam()/UKFSlam() + initialize(): ekf.cpp:39-71, ukf.cpp:39-67 (for every filter of the batch) */
int aslam_create(const aslam_config *cfg, aslam_ctx **out);
int aslam_destroy(aslam_ctx *ctx);
/* initialize() again: N = 3, P/Q/R defaults, empty wait-list, init_x = init_z = true */
int aslam_reset(aslam_ctx *ctx);
const char *aslam_last_error(void);
int aslam_abi_version(void);

/* ---- the per-callback seam (host keeps association/growth: ekf.cpp:137-290 stay on the host) ------ */
/* Set the device state of one filter: dimension n, X[n], Z[n], P[n*n] row-major (any of them NULL = keep).
 * Used for state hand-over and kernel-level tests. */
int aslam_set_state(aslam_ctx *ctx, int traj, int n, const double *X, const double *Z, const double *P);
/* the matrix part of updateNewLandmark (ekf.cpp:271-278 / ukf.cpp:238-245): grow filter `traj` from its
 * current dimension to n_new; new P diagonal = UKF_KP_LANDMARK_POSE, new X/Z entries from the seeds
 * (x_seed, z_seed hold n_new - n_old values). */
int aslam_grow(aslam_ctx *ctx, int traj, int n_new, const double *x_seed, const double *z_seed);
/* EKFSlam::slam (ekf.cpp:293-311) for one filter.  Z[n] is param.Z after updateZandA, a00/a10 are
 * param.A(0,0)/A(1,0) (ekf.cpp:210-211).  X_out (n doubles, may be NULL) receives param.X. */
int aslam_ekf_step(aslam_ctx *ctx, int traj, float vx, float az, float dt, const double *Z, double a00,
                   double a10, double *X_out, void *stream);
/* UKFSlam::slam (ukf.cpp:260-392) for one filter. */
int aslam_ukf_step(aslam_ctx *ctx, int traj, float vx, float az, float dt, const double *Z, double *X_out,
                   void *stream);

/* The same seam for ALL `batch` filters of the context at once (B live robots: one launch chain instead of B calls; replaces B x
 * `slam(...)` at ekf.cpp:94 / ukf.cpp:90).  Host arrays: vx, az, dt [batch]; Z [batch][ldz], row b = param.Z of filter b (its first
 * N_b entries are used); a00, a10 [batch] (EKF: param.A(0,0), param.A(1,0)); X_out [batch][ldx] or NULL.
 * ASYNCHRONOUS on `stream`: nothing inside synchronises.  The input arrays must stay untouched and X_out must not be read until
 * the caller has synchronised the stream (use pinned host memory for copies that really overlap). */
int aslam_ekf_step_batch(aslam_ctx *ctx, const float *vx, const float *az, const float *dt, const double *Z, int ldz,
                         const double *a00, const double *a10, double *X_out, int ldx, void *stream);
int aslam_ukf_step_batch(aslam_ctx *ctx, const float *vx, const float *az, const float *dt, const double *Z, int ldz,
                         double *X_out, int ldx, void *stream);

/* ---- the replay seam (the whole callback, association and growth included, runs on the device) ----- */
/* Bind a trace.  Host pointers are copied to HBM; device pointers are used in place and must stay valid. */
int aslam_set_trace(aslam_ctx *ctx, const aslam_trace *trace);
/* Run callbacks t0 .. t0+nsteps-1 of the bound trace for every filter of the batch: cbSensorLandmark (if
 * obs_new) + cbOdom (updateZ[andA], growth, slam).  Asynchronous on `stream`.  poses_out (device memory,
 * [batch][nsteps][3] f64, may be NULL) receives X(0..2) after each callback; dims_out (device memory,
 * [batch][nsteps] i32, may be NULL) the state dimension N. */
int aslam_replay(aslam_ctx *ctx, int64_t t0, int64_t nsteps, double *poses_out, int32_t *dims_out, void *stream);

/* ---- read-back (synchronises the context's last stream) ------------------------------------------- */
int aslam_get_dim(aslam_ctx *ctx, int traj, int *n);
/* X[n], Z[n], P[n*n] row-major; any may be NULL */
int aslam_get_state(aslam_ctx *ctx, int traj, double *X, double *Z, double *P);
int aslam_get_A(aslam_ctx *ctx, int traj, double *a00, double *a10);
/* convertToLandmarkMsg (common.h:93-108): x[i] = X(3+2i), y[i] = X(4+2i); returns the count in *n_landmarks */
int aslam_get_landmarks(aslam_ctx *ctx, int traj, double *x, double *y, int *n_landmarks);
/* wait-list: up to `cap` entries of (range, bearing, count); *size = entries held */
int aslam_get_wait(aslam_ctx *ctx, int traj, float *range, float *bearing, uint32_t *count, int cap, int *size);
int aslam_get_status(aslam_ctx *ctx, int traj, uint32_t *status_bits);
/* padded row length of the device layout (multiple of 16) and bytes of HBM held, for reporting */
int aslam_get_layout(aslam_ctx *ctx, int *padded_dim, int64_t *hbm_bytes);
/* name + launch geometry of the kernel aslam_replay uses for this context (for profiles / bench reports) */
int aslam_kernel_info(aslam_ctx *ctx, char *name, int name_cap, int *grid, int *block, int *lds_bytes);
/* what the LAST aslam_replay / aslam_*_step[_batch] of this context really launched (large-state path; the single-CU kernels report
 * 0, 0, 1): stream groups the batch was split into (1 = the caller's stream alone), whether the Cholesky of S ran as the one-launch
 * resident kernel, kernel launches per callback and group.  Lets a test assert that it exercised the launch shape it means to. */
int aslam_get_launch_info(aslam_ctx *ctx, int *stream_groups, int *chol_resident, int *launches_per_callback);

#ifdef __cplusplus
}
#endif
#endif /* ASLAM_CORE_H */
