// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <vector>
#include <std_msgs/Header.h>
#include <boost/shared_ptr.hpp>
namespace sensor_msgs { struct LaserScan { std_msgs::Header header; float angle_min, angle_max, angle_increment, time_increment, scan_time, range_min, range_max; std::vector<float> ranges, intensities; }; typedef boost::shared_ptr<LaserScan const> LaserScanConstPtr; }
