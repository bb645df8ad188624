// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <std_msgs/Header.h>
#include <geometry_msgs/Pose.h>
#include <geometry_msgs/Twist.h>
#include <boost/shared_ptr.hpp>
namespace nav_msgs { struct Odometry { std_msgs::Header header; std::string child_frame_id; struct { geometry_msgs::Pose pose; } pose; struct { geometry_msgs::Twist twist; } twist; typedef boost::shared_ptr<Odometry const> ConstPtr; }; typedef boost::shared_ptr<Odometry const> OdometryConstPtr; }
