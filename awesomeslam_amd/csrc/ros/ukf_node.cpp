// ros/ukf_node.cpp -- `rosrun awesome_slam ukf` (awesome_slam/src/ukf/ukf.cpp:394-407): node aslam_ukf, subscriber queues of size 10.
#include "node_main.h"

int main(int argc, char **argv)
{
        return aslam_ros::node_main<aslam::UKFSlam>(argc, argv, "aslam_ukf", 10, "[UKF] Node started!\n");
}
