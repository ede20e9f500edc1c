// aslam_node.h -- host-side mirror of the reference's filter nodes, ROS-free.
//
// aslam::EKFSlam / aslam::UKFSlam keep the reference's class and member names
// (awesome_slam/src/ekf/ekf.h:71-131, awesome_slam/src/ukf/ukf.h:84-143): the callbacks, the data
// association and the landmark bookkeeping (ekf.cpp:74-290, ukf.cpp:70-257) run on the host exactly as
// in the reference, while `param.P`, `param.X` and slam() live on the MI355X behind the C ABI of
// include/aslam_core.h (per-callback seam: aslam_grow + aslam_ekf_step / aslam_ukf_step).
// Messages are plain structs with the fields the nodes read; ros/ekf_node.cpp and ros/ukf_node.cpp
// (compile-gated on catkin) wrap them into the real ROS callbacks.
//
// This is product code: it never touches oracle/.
#pragma once

#include <cstdint>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/aslam_core.h"

namespace aslam
{
/// the fields of nav_msgs/Odometry the nodes read (ekf.cpp:139-142,94)
struct Odometry
{
        double px, py;         // pose.pose.position.{x,y}
        double qw, qx, qy, qz; // pose.pose.orientation
        double vx, wz;         // twist.twist.linear.x, twist.twist.angular.z
};

/// awesome_slam_msgs/Landmarks (msg/Landmarks.msg:1-2)
struct Landmarks
{
        std::vector<double> x, y;
};

/// structures.h:85-112
struct LaserData
{
        float range;
        float bearing;
};

/// tools.h:44-50 / 62-66 (binary32)
float normalizeAngle(float theta);
float quat2euler(float w, float x, float y, float z);

class FilterNode
{
      public:
        /// `now_init` stands for the ros::Time::now().toSec() of initialize() (ekf.cpp:54 / ukf.cpp:54): it seeds the binary32 last_time
        FilterNode(int filter, int max_landmark_count, int device, double now_init = 0.0);
        virtual ~FilterNode();
        FilterNode(const FilterNode &) = delete;
        FilterNode &operator=(const FilterNode &) = delete;

        /// ekf.cpp:102-114 / ukf.cpp:98-110
        void cbSensorLandmark(const Landmarks &msg);
        /// ekf.cpp:74-99 / ukf.cpp:70-95; `now` stands for ros::Time::now().toSec().  Returns false when the
        /// callback returned early (no sensor message yet).  Throws std::runtime_error if the core fails.
        bool cbOdom(const Odometry &msg, double now);
        /// same with delta_time given instead of derived from the clock
        bool cbOdomDt(const Odometry &msg, float delta_time);
        /// convertToLandmarkMsg(N, param.X), common.h:93-108: what the node publishes on out/landmarks/kalman
        Landmarks landmarks() const;

        uint32_t dim() const
        {
                return N;
        }
        const std::vector<double> &X() const
        {
                return param_X;
        }
        const std::vector<double> &Z() const
        {
                return param_Z;
        }
        double A00() const
        {
                return a00;
        }
        double A10() const
        {
                return a10;
        }
        const std::vector<std::pair<LaserData, uint32_t>> &waitList() const
        {
                return new_landmark_wait;
        }
        aslam_ctx *core() const
        {
                return ctx;
        }
        bool growthRefused() const
        {
                return growth_refused;
        }

      private:
        int filter;
        int MAX_LANDMARK_COUNT;
        aslam_ctx *ctx;

        uint32_t N;
        bool init_z;
        bool init_x;
        float last_time;
        bool growth_refused;
        std::vector<LaserData> sensor_landmark;
        std::vector<std::pair<LaserData, uint32_t>> new_landmark_wait;
        std::vector<double> param_X, param_Z; // host copies; P and the authoritative X are device-resident
        double a00, a10;                      // param.A(0,0), param.A(1,0) (EKF)

        void updateZ(const Odometry &msg, float delta_time);
        void updateNewLandmarkWait(const LaserData &data);
        void updateNewLandmark(const std::vector<LaserData> &new_landmark);
        void slam(float vx, float az, float delta_time);
};

class EKFSlam : public FilterNode
{
      public:
        explicit EKFSlam(int max_landmark_count = 30, int device = 0, double now_init = 0.0) : FilterNode(ASLAM_EKF, max_landmark_count, device, now_init)
        {
        }
};

class UKFSlam : public FilterNode
{
      public:
        explicit UKFSlam(int max_landmark_count = 30, int device = 0, double now_init = 0.0) : FilterNode(ASLAM_UKF, max_landmark_count, device, now_init)
        {
        }
};
} // namespace aslam

// ---- C shim (ctypes / other FFI) ------------------------------------------------------------------------
extern "C" {
typedef struct aslam_node aslam_node;
/* filter: ASLAM_EKF | ASLAM_UKF.  NULL on failure (aslam_node_error()). */
aslam_node *aslam_node_create(int filter, int max_landmark_count, int device);
/* the same with the construction time of the node (seconds): initialize() stores ros::Time::now().toSec() in the binary32 last_time
 * (ekf.cpp:54), which the first delta_time of aslam_node_odom_now() is measured from.  aslam_node_create = ..._at(..., 0.0). */
aslam_node *aslam_node_create_at(int filter, int max_landmark_count, int device, double now_init);
void aslam_node_destroy(aslam_node *n);
const char *aslam_node_error(void);
int aslam_node_sensor(aslam_node *n, int count, const double *x, const double *y);
/* returns 1 if slam() ran, 0 if the callback returned early, negative on error */
int aslam_node_odom(aslam_node *n, const double msg[8], float delta_time);
int aslam_node_odom_now(aslam_node *n, const double msg[8], double now);
int aslam_node_dim(const aslam_node *n);
int aslam_node_get(const aslam_node *n, double *X, double *Z, double *a00, double *a10);
int aslam_node_wait(const aslam_node *n, float *range, float *bearing, uint32_t *count, int cap);
aslam_ctx *aslam_node_core(const aslam_node *n);
/* Narrow `count` recorded odometry messages ([count][8]: px,py,qw,qx,qy,qz,vx,wz) the way cbOdom/updateZandA do
 * (ekf.cpp:139-142): pose[count][2], yaw[count] = quat2euler(...) as binary32, twist[count][2]. */
void aslam_host_narrow_odom(int64_t count, const double *odom, double *pose, float *yaw, double *twist);
}
