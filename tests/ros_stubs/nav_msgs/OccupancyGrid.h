// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <vector>
#include <std_msgs/Header.h>
#include <geometry_msgs/Pose.h>
#include <boost/shared_ptr.hpp>
namespace nav_msgs { struct MapMetaData { ros::Time map_load_time; float resolution; uint32_t width, height; geometry_msgs::Pose origin; }; struct OccupancyGrid { std_msgs::Header header; MapMetaData info; std::vector<int8_t> data; typedef boost::shared_ptr<OccupancyGrid const> ConstPtr; }; typedef boost::shared_ptr<OccupancyGrid const> OccupancyGridConstPtr; }
