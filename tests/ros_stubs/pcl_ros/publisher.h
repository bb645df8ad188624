// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <ros/ros.h>
namespace pcl_ros { template <class T> class Publisher { public: template <class C> void publish(const C&) const {} void advertise(ros::NodeHandle&, const std::string&, uint32_t) {} }; }
