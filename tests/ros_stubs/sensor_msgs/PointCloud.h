// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <vector>
#include <std_msgs/Header.h>
#include <geometry_msgs/Point32.h>
#include <boost/shared_ptr.hpp>
namespace sensor_msgs { struct ChannelFloat32 { std::string name; std::vector<float> values; }; struct PointCloud { std_msgs::Header header; std::vector<geometry_msgs::Point32> points; std::vector<ChannelFloat32> channels; }; typedef boost::shared_ptr<PointCloud const> PointCloudConstPtr; }
