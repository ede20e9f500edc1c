// ros/ukf_node.cpp -- `rosrun awesome_slam ukf`: the reference's node surface (awesome_slam/src/ukf/ukf.cpp:39-46,394-407:
// node name aslam_ukf, /odom and /out/landmarks/sensor with queue size 10, out/landmarks/kalman, 1 Hz spin loop) on top of
// the host mirror aslam::UKFSlam, whose covariance and slam() live on the MI355X behind include/aslam_core.h.
//
// Compile-gated: built only where catkin/roscpp and awesome_slam_msgs exist (ros/CMakeLists.txt).  The build image of
// this repository has no ROS, so this file has never been compiled here; the classes it wraps are tested through the
// aslam_node_* C shim (tests/test_gpu_ukf.py::test_per_callback_seam_host_mirror).
#include <awesome_slam_msgs/Landmarks.h>
#include <nav_msgs/Odometry.h>
#include <ros/ros.h>

#include <iostream>

#include "../host/aslam_node.h"

namespace
{
aslam::UKFSlam *filter = nullptr;
ros::Publisher pub_landmark;

void cbOdom(const nav_msgs::Odometry::ConstPtr &msg)
{
        const aslam::Odometry o{msg->pose.pose.position.x,    msg->pose.pose.position.y,    msg->pose.pose.orientation.w,
                                msg->pose.pose.orientation.x, msg->pose.pose.orientation.y, msg->pose.pose.orientation.z,
                                msg->twist.twist.linear.x,    msg->twist.twist.angular.z};
        if (!filter->cbOdom(o, ros::Time::now().toSec()))
                return; // no sensor message yet (ukf.cpp:72-73)
        const aslam::Landmarks l = filter->landmarks(); // convertToLandmarkMsg, common.h:93-108
        awesome_slam_msgs::Landmarks out;
        out.x = l.x;
        out.y = l.y;
        pub_landmark.publish(out);
}

void cbSensorLandmark(const awesome_slam_msgs::Landmarks::ConstPtr &msg)
{
        filter->cbSensorLandmark(aslam::Landmarks{msg->x, msg->y});
}
} // namespace

int main(int argc, char **argv)
{
        ros::init(argc, argv, "aslam_ukf");
        ros::Time::init();
        ros::NodeHandle nh;
        int max_landmark_count = 30; // config.h:45; a private parameter here instead of a recompile
        ros::param::param("~max_landmark_count", max_landmark_count, 30);
        aslam::UKFSlam node(max_landmark_count);
        filter = &node;
        ros::Subscriber sub_odom = nh.subscribe("/odom", 10, cbOdom);
        ros::Subscriber sub_sensor_landmark = nh.subscribe("/out/landmarks/sensor", 10, cbSensorLandmark);
        pub_landmark = nh.advertise<awesome_slam_msgs::Landmarks>("out/landmarks/kalman", 1);
        ros::Rate rate(1); // FREQ, config.h:42
        std::cerr << "[UKF] Node started!\n";
        while (ros::ok())
        {
                ros::spinOnce();
                rate.sleep();
        }
        return 0;
}
