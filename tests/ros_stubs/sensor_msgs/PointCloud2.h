// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <vector>
#include <std_msgs/Header.h>
#include <boost/shared_ptr.hpp>
namespace sensor_msgs { struct PointCloud2 { std_msgs::Header header; uint32_t height, width; std::vector<uint8_t> data; }; typedef boost::shared_ptr<PointCloud2 const> PointCloud2ConstPtr; }
