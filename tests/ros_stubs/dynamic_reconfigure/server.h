// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <ros/ros.h>
#include <boost/thread.hpp>
namespace dynamic_reconfigure { template <class C> class Server { public: typedef boost::function<void(C&, uint32_t)> CallbackType; Server(const ros::NodeHandle& = ros::NodeHandle()) {} Server(boost::recursive_mutex&, const ros::NodeHandle& = ros::NodeHandle()) {} void setCallback(const CallbackType& cb) { C c; cb(c, 0); } void updateConfig(const C&) {} void getConfigDefault(C&) {} void clearCallback() {} }; }
