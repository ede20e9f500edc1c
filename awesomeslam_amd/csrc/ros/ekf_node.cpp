// ros/ekf_node.cpp -- `rosrun awesome_slam ekf` (awesome_slam/src/ekf/ekf.cpp:313-326): node aslam_ekf, subscriber queues of size 1.
#include "node_main.h"

int main(int argc, char **argv)
{
        return aslam_ros::node_main<aslam::EKFSlam>(argc, argv, "aslam_ekf", 1, "[EKF] Node started!\n");
}
