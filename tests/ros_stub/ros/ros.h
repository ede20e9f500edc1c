// tests/ros_stub/ros/ros.h -- NOT ROS.  A minimal stand-in for the handful of roscpp declarations csrc/ros/*.cpp use, so that the wrappers can be
// parsed, type-checked and linked in an image without ROS (tests/test_ros_wrappers.py).  Signatures follow roscpp (Noetic) for exactly these
// members: ros::init, ros::ok, ros::spinOnce, ros::Time::{init, now, toSec}, ros::Rate::{Rate(double), sleep}, ros::param::param<T>,
// ros::NodeHandle::{subscribe(topic, queue, member function, object), advertise<M>(topic, queue)}, ros::Publisher::publish<M>,
// ros::Subscriber.  Nothing here delivers a message: ros::ok() is false, the spin loop of a wrapper linked against this never runs.
// It is test scaffolding for a build check, never part of the product and never a substitute for the reference's dependencies.
#pragma once
#include <cstdint>
#include <string>

#include <boost_stub/shared_ptr.h>

namespace ros
{
inline void init(int &, char **, const std::string &) {}
inline bool ok() { return false; }
inline void spinOnce() {}

class Time
{
      public:
        static void init() {}
        static Time now() { return Time(); }
        double toSec() const { return 0.0; }
};

class Rate
{
      public:
        explicit Rate(double) {}
        bool sleep() { return true; }
};

namespace param
{
template <typename T> bool param(const std::string &, T &value, const T &fallback)
{
        value = fallback;
        return false;
}
} // namespace param

class Subscriber
{
};

class Publisher
{
      public:
        template <typename M> void publish(const M &) const {}
};

class NodeHandle
{
      public:
        template <class M, class T> Subscriber subscribe(const std::string &, uint32_t, void (T::*)(const boost::shared_ptr<M const> &), T *)
        {
                return Subscriber();
        }
        template <class M> Subscriber subscribe(const std::string &, uint32_t, void (*)(const boost::shared_ptr<M const> &)) { return Subscriber(); }
        template <class M> Publisher advertise(const std::string &, uint32_t, bool = false) { return Publisher(); }
};
} // namespace ros
