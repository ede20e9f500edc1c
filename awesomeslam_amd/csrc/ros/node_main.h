// ros/node_main.h -- the reference's node surface (awesome_slam/src/ekf/ekf.cpp:39-46,74-114,313-326; src/ukf/ukf.cpp:39-46,70-110,394-407)
// on top of the host mirror aslam::EKFSlam / aslam::UKFSlam (../host/aslam_node.h), whose covariance and slam() live on the MI355X behind
// include/aslam_core.h.  One template for both executables: they differ in the filter class, the node name, the subscriber queue size
// (1 for the EKF, 10 for the UKF) and the start-up message, exactly as the reference's two main() do.
//
// Compile-gated: built only where catkin/roscpp and awesome_slam_msgs exist (ros/CMakeLists.txt).  This repository's image has no ROS;
// tests/test_ros_wrappers.py compiles and links both executables against the minimal interface stubs in tests/ros_stub/ (syntax, types and
// the aslam_node.h calls -- nothing about ROS behaviour).
#pragma once

#include <awesome_slam_msgs/Landmarks.h>
#include <nav_msgs/Odometry.h>
#include <ros/ros.h>

#include <iostream>

#include "../host/aslam_node.h"

namespace aslam_ros
{
template <class Filter> struct Node
{
        Filter filter;
        ros::NodeHandle nh;
        ros::Subscriber sub_odom, sub_sensor_landmark;
        ros::Publisher pub_landmark;

        /// the reference's constructor (ekf.cpp:39-46): subscribe, advertise, initialize() -- whose `last_time = ros::Time::now().toSec()`
        /// (ekf.cpp:54) is the `now_init` of the mirror
        Node(int max_landmark_count, uint32_t queue_size) : filter(max_landmark_count, 0, ros::Time::now().toSec())
        {
                sub_odom = nh.subscribe("/odom", queue_size, &Node::cbOdom, this);
                sub_sensor_landmark = nh.subscribe("/out/landmarks/sensor", queue_size, &Node::cbSensorLandmark, this);
                pub_landmark = nh.template advertise<awesome_slam_msgs::Landmarks>("out/landmarks/kalman", 1);
        }

        /// ekf.cpp:74-99: delta_time from the clock, updateZandA, slam(), publish the landmark estimates
        void cbOdom(const nav_msgs::Odometry::ConstPtr &msg)
        {
                const aslam::Odometry o{msg->pose.pose.position.x,    msg->pose.pose.position.y,    msg->pose.pose.orientation.w,
                                        msg->pose.pose.orientation.x, msg->pose.pose.orientation.y, msg->pose.pose.orientation.z,
                                        msg->twist.twist.linear.x,    msg->twist.twist.angular.z};
                if (!filter.cbOdom(o, ros::Time::now().toSec()))
                        return; // no sensor message yet (ekf.cpp:76-77)
                const aslam::Landmarks l = filter.landmarks(); // convertToLandmarkMsg, common.h:93-108
                awesome_slam_msgs::Landmarks out;
                out.x = l.x;
                out.y = l.y;
                pub_landmark.publish(out);
        }

        /// ekf.cpp:102-114
        void cbSensorLandmark(const awesome_slam_msgs::Landmarks::ConstPtr &msg)
        {
                filter.cbSensorLandmark(aslam::Landmarks{msg->x, msg->y});
        }
};

/// the reference's main() (ekf.cpp:313-326): init, Rate(FREQ), the node object, the 1 Hz spin loop
template <class Filter> int node_main(int argc, char **argv, const char *node_name, uint32_t queue_size, const char *banner)
{
        ros::init(argc, argv, node_name);
        ros::Time::init();
        ros::Rate rate(1); // FREQ, config.h:42
        int max_landmark_count = 30; // config.h:45; a private parameter here instead of a recompile
        ros::param::param("~max_landmark_count", max_landmark_count, 30);
        Node<Filter> a(max_landmark_count, queue_size);
        std::cerr << banner;
        while (ros::ok())
        {
                ros::spinOnce();
                rate.sleep();
        }
        return 0;
}
} // namespace aslam_ros
