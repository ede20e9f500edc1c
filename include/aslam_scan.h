/* aslam_scan.h -- the step before the filter: a 360-beam laser scan -> the (range, bearing) list the
 * filter nodes subscribe to.  Replaces aslam::LandMarks::callback (src/sensor_landmark/sensor_landmark.cpp:59-132) with its
 * helpers bearing2pose (:135-142), circleClassification (:147-186, Xavier et al. ICRA 2005) and circleFitting (:192-298,
 * Al-Sharadqah & Chernov 2009 "hyper" fit), batched: one wavefront per scan, any number of scans per call.
 *
 *   ranges      [count][360] f32  sensor_msgs/LaserScan::ranges (inf = no return)
 *   range_out   [count][max_out] f32, bearing_out likewise: awesome_slam_msgs/Landmarks x[], y[] of each scan, in beam order
 *   n_out       [count] i32  number of landmarks of each scan (<= max_out)
 *   status_out  [count] u32  ASLAM_SCAN_* bits
 *
 * All five arrays live in host memory (is_device = 0: copied in and out, synchronous) or all in device memory
 * (is_device = 1: asynchronous on `stream`).  Implemented in libaslam_core.so.
 */
#ifndef ASLAM_SCAN_H
#define ASLAM_SCAN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ASLAM_SCAN_BEAMS 360
/* beams 0 and 359 are closer than MIN_DIST_THRESH: the reference walks theta off the scan and fails assert(theta < 359)
 * (sensor_landmark.cpp:69-81); no landmarks are produced for such a scan */
#define ASLAM_SCAN_REF_ABORT 1u
/* more than max_out landmarks: the list is truncated */
#define ASLAM_SCAN_OVERFLOW 2u

int aslam_scan_landmarks(const float *ranges, int64_t count, int is_device, int max_out, float *range_out, float *bearing_out,
                         int32_t *n_out, uint32_t *status_out, int device, void *stream);

#ifdef __cplusplus
}
#endif
#endif
