// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <vector>
#include <std_msgs/Header.h>
#include <boost/shared_ptr.hpp>
namespace map_msgs { struct OccupancyGridUpdate { std_msgs::Header header; int32_t x, y; uint32_t width, height; std::vector<int8_t> data; }; typedef boost::shared_ptr<OccupancyGridUpdate const> OccupancyGridUpdateConstPtr; }
