/*
 * aslam_oracle.h -- C interface of the CPU oracle.
 *
 * TEST INFRASTRUCTURE ONLY.  This is an Eigen-free, ROS-free CPU restatement of the
 * reference's EKF/UKF-SLAM filter nodes (iamarkaj/AwesomeSLAM):
 *     awesome_slam/src/ekf/ekf.cpp:49-311, awesome_slam/src/ukf/ukf.cpp:49-392,
 *     awesome_slam/src/ukf/ukf.h:56-82, awesome_slam/include/awesome_slam/{common,tools,structures,config}.h
 * It exists so that tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg can
 * check / time the HIP product path against the reference arithmetic.  Nothing under
 * awesomeslam_amd/ may include, link or call it.
 *
 * PARITY STATUS: "parity unpinned" by the reference -- the reference ships no tests,
 * fixtures or golden vectors and cannot be compiled here (Eigen and ROS
 * are absent).  The pins are this repo's own: an independent NumPy
 * restatement (oracle/np_oracle.py) must agree with this file, and committed fixtures under
 * tests/golden/ freeze both.
 *
 * The one sanctioned deviation from the reference: MAX_LANDMARK_COUNT (config.h:45) is a
 * run-time field (default 30) so that the 64- and 512-landmark configurations exist at all
 * (SURVEY.md F4).
 */
#ifndef ASLAM_ORACLE_H
#define ASLAM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct orc_filter orc_filter;

enum { ORC_EKF = 0, ORC_UKF = 1 };

/* constructor + initialize(): ekf.cpp:39-71 / ukf.cpp:39-67.  max_landmark_count = config.h:45 (30). */
orc_filter *orc_create(int kind, int max_landmark_count);
void orc_destroy(orc_filter *f);

/* cbSensorLandmark: ekf.cpp:102-114 / ukf.cpp:98-110.  x = range, y = bearing (float64 as on the wire). */
void orc_sensor(orc_filter *f, int n, const double *x, const double *y);

/* cbOdom: ekf.cpp:74-99 / ukf.cpp:70-95 with delta_time supplied by the caller (the reference
 * derives it from ros::Time).  Returns 0 when the callback returned early (init_z), else 1. */
int orc_odom(orc_filter *f, double px, double py, double qw, double qx, double qy, double qz, double vx,
             double wz, float delta_time);

/* the core only: slam(), ekf.cpp:293-311 / ukf.cpp:260-392 */
void orc_slam(orc_filter *f, float vx, float az, float delta_time);

/* state access (row-major matrices, N x N) */
int orc_dim(const orc_filter *f);
void orc_get(const orc_filter *f, double *X, double *Z, double *P);
void orc_get_A(const orc_filter *f, double *a00, double *a10);
/* overwrite the filter state with a synthetic one of dimension N (kernel-level tests): X, Z, P are
 * taken as given; Q, R, A, H, I and the UKF weights are what N/2-1 calls of updateNewLandmark would
 * have left (ekf.cpp:271-278 / ukf.cpp:238-245). */
void orc_set(orc_filter *f, int N, const double *X, const double *Z, const double *P, double a00, double a10);

/* landmark bookkeeping (the bit-exact gate of SURVEY.md a19) */
int orc_wait_size(const orc_filter *f);
void orc_get_wait(const orc_filter *f, float *range, float *bearing, uint32_t *count);
int orc_sensor_size(const orc_filter *f);
void orc_get_sensor(const orc_filter *f, float *range, float *bearing);
void orc_get_weights(const orc_filter *f, double *w, float *lambda); /* UKF only: 2N+5 weights */

/* Replay T steps of a message-level trace through cbSensorLandmark/cbOdom.
 *   odom    [T][8] : px, py, qw, qx, qy, qz, vx, wz
 *   dt      [T]
 *   obs_new [T]    : 1 = a sensor message (n_obs[t] entries of obs[t]) arrives before this odom message
 *   obs     [T][max_obs][2] : range, bearing
 * poses_out [T][3] (X(0..2) after each step, zeros for dropped callbacks), dims_out [T] (N after each step).
 * Returns the number of callbacks that ran slam(). */
int64_t orc_replay(orc_filter *f, int64_t T, const double *odom, const float *dt, const uint8_t *obs_new,
                   const int32_t *n_obs, const double *obs, int max_obs, double *poses_out, int32_t *dims_out);

/* small numeric helpers exposed for known-answer tests (tools.h:44-66) */
float orc_normalize_angle(float theta);
float orc_quat2euler(float w, float x, float y, float z);

#ifdef __cplusplus
}
#endif
#endif
