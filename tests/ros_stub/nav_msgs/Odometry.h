// tests/ros_stub/nav_msgs/Odometry.h -- NOT ROS: the fields of nav_msgs/Odometry the wrappers read (pose.pose.position.{x,y},
// pose.pose.orientation.{x,y,z,w}, twist.twist.linear.x, twist.twist.angular.z), with the generated header's nesting and ConstPtr typedef.
#pragma once
#include <boost_stub/shared_ptr.h>
namespace geometry_msgs
{
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Pose { Point position; Quaternion orientation; };
struct Twist { Vector3 linear, angular; };
struct PoseWithCovariance { Pose pose; double covariance[36] = {}; };
struct TwistWithCovariance { Twist twist; double covariance[36] = {}; };
} // namespace geometry_msgs
namespace nav_msgs
{
struct Odometry
{
        geometry_msgs::PoseWithCovariance pose;
        geometry_msgs::TwistWithCovariance twist;
        typedef boost::shared_ptr<Odometry> Ptr;
        typedef boost::shared_ptr<Odometry const> ConstPtr;
};
} // namespace nav_msgs
