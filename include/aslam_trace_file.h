/* aslam_trace_file.h -- on-disk format of a recorded input stream and its reader / writer.
 *
 * A trace holds, for `batch` independent robots, the messages the reference nodes consume, at MESSAGE level -- before the
 * callbacks narrow them -- in the order the node's 1 Hz spin loop (ekf.cpp:313-326, queue size 1: ekf.cpp:41-42) hands
 * them to cbSensorLandmark (ekf.cpp:102-114) and cbOdom (ekf.cpp:74-100):
 *
 *   odom    [batch][T][8]  f64  nav_msgs/Odometry: pose.position.{x,y}, orientation.{w,x,y,z}, twist.linear.x, twist.angular.z
 *   dt      [batch][T]     f32  delta_time of the callback (ekf.cpp:80: min(now - last_time, 1.0) as float)
 *   obs_new [batch][T]     u8   1 = an awesome_slam_msgs/Landmarks message is delivered before this odometry message
 *   n_obs   [batch][T]     i32  its length
 *   obs     [batch][T][max_obs][2] f32  (range, bearing) after LaserData::assign's double -> float (structures.h:85-101)
 *
 * File layout (little endian): a 64-byte header, then the five arrays in the order above, each starting at a multiple of
 * 64 bytes.  Optional ground truth (synthetic traces) follows: landmarks [batch][L][2] f64, truth [batch][T][3] f64.
 *
 *   offset  0  char     magic[8] = "ASLTRC01"
 *           8  int64    batch
 *          16  int64    T
 *          24  int32    max_obs
 *          28  int32    L          (0 = no ground truth stored)
 *          32  int32    warmup     (callbacks of the survey lap; informational)
 *          36  int32    reserved[7]
 *
 * The reader keeps the file in host memory and produces the NARROWED view aslam_set_trace() takes (aslam_core.h:
 * pose / yaw / twist split, yaw = quat2euler in binary32 as updateZandA does, tools.h:62-66).
 */
#ifndef ASLAM_TRACE_FILE_H
#define ASLAM_TRACE_FILE_H

#include "aslam_core.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct aslam_trace_file aslam_trace_file;

/* Read `path`.  ASLAM_OK, or ASLAM_ERR_ARG (cannot open / not a trace file / truncated); message in aslam_trace_file_error(). */
int aslam_trace_file_open(const char *path, aslam_trace_file **out);
void aslam_trace_file_close(aslam_trace_file *f);
const char *aslam_trace_file_error(void);
int aslam_trace_file_dims(const aslam_trace_file *f, int64_t *batch, int64_t *T, int32_t *max_obs, int32_t *landmarks);
/* Host-memory view for aslam_set_trace (valid until aslam_trace_file_close). */
int aslam_trace_file_view(aslam_trace_file *f, aslam_trace *view);
/* Message-level arrays as stored (any pointer may be NULL). */
int aslam_trace_file_raw(const aslam_trace_file *f, const double **odom, const float **dt, const uint8_t **obs_new,
                         const int32_t **n_obs, const float **obs);
/* Write a trace (ground truth may be NULL with landmarks = 0). */
int aslam_trace_file_write(const char *path, int64_t batch, int64_t T, int32_t max_obs, int32_t warmup, const double *odom,
                           const float *dt, const uint8_t *obs_new, const int32_t *n_obs, const float *obs, int32_t landmarks,
                           const double *landmark_xy, const double *truth);

#ifdef __cplusplus
}
#endif
#endif
