"""the ROS node wrappers (csrc/ros/ekf_node.cpp, ukf_node.cpp, node_main.h) meet a compiler.

This image has no ROS, and stand-ins for it may not be used to claim anything about ROS behaviour.  What this test DOES prove: both wrapper
sources parse and type-check under -Wall -Wextra against headers that declare the roscpp / nav_msgs / awesome_slam_msgs members they use with
roscpp's signatures (tests/ros_stub/: ros::init, NodeHandle::subscribe / advertise, Publisher::publish, Rate, Time::now().toSec(),
param::param, nav_msgs::Odometry's fields, awesome_slam_msgs::Landmarks = float64[] x, y), that every call into csrc/host/aslam_node.h is
well-formed (constructor with the construction-time clock, cbOdom(msg, now), cbSensorLandmark, landmarks()), and that the two executables
LINK against libaslam_node.so / libaslam_core.so (no unresolved aslam symbol).  What it does NOT prove: anything about topics, queues,
callbacks being delivered or the spin loop -- the stubs deliver nothing (ros::ok() is false).  Row N2 stays "untestable here" for behaviour
(reference: ekf.cpp:39-46,74-114,313-326; ukf.cpp:39-46,394-407)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "awesomeslam_amd", "csrc")
STUB = os.path.join(ROOT, "tests", "ros_stub")


@pytest.mark.parametrize("node", ["ekf", "ukf"])
def test_wrapper_compiles_and_links_against_interface_stubs(node, built, tmp_path):
    src = os.path.join(CSRC, "ros", f"{node}_node.cpp")
    obj, exe = str(tmp_path / f"{node}_node.o"), str(tmp_path / f"{node}_node")
    r = subprocess.run(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", f"-I{STUB}", f"-I{os.path.join(ROOT, 'include')}", "-c", src, "-o", obj],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    # the HIP runtime behind libaslam_core.so is resolved at load time (PyTorch's copy, core.py): allow it to stay undefined at link time
    r = subprocess.run(["g++", "-o", exe, obj, f"-L{CSRC}", "-laslam_node", "-laslam_core", f"-Wl,-rpath,{CSRC}", "-Wl,--allow-shlib-undefined"],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    syms = subprocess.run(["nm", "-u", "-C", exe], stdout=subprocess.PIPE, text=True).stdout
    want = "aslam::FilterNode::cbOdom(aslam::Odometry const&, double)"
    assert want in syms and "aslam::FilterNode::cbSensorLandmark" in syms and "aslam::FilterNode::FilterNode(int, int, int, double)" in syms, syms


def test_stubs_are_not_shipped():
    """the stand-in headers live under tests/ only: nothing in the package or in include/ may reach them"""
    for base in (os.path.join(ROOT, "awesomeslam_amd"), os.path.join(ROOT, "include")):
        for d, _, files in os.walk(base):
            for f in files:
                if f.endswith((".h", ".cpp", ".hip", ".txt", ".py", "Makefile")):
                    assert "ros_stub" not in open(os.path.join(d, f), errors="replace").read().replace("tests/ros_stub/", "").replace("tests/test_ros_wrappers.py", ""), os.path.join(d, f)
