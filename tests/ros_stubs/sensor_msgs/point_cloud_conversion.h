// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <sensor_msgs/PointCloud.h>
#include <sensor_msgs/PointCloud2.h>
namespace sensor_msgs { inline bool convertPointCloudToPointCloud2(const PointCloud&, PointCloud2&) { return true; } inline bool convertPointCloud2ToPointCloud(const PointCloud2&, PointCloud&) { return true; } }
