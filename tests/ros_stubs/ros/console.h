// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <cstdio>
#define ROS_DEBUG(...) ((void)0)
#define ROS_INFO(...) ((void)0)
#define ROS_WARN(...) ((void)0)
#define ROS_ERROR(...) ((void)0)
#define ROS_FATAL(...) ((void)0)
#define ROS_DEBUG_NAMED(...) ((void)0)
#define ROS_INFO_NAMED(...) ((void)0)
#define ROS_WARN_NAMED(...) ((void)0)
#define ROS_ERROR_NAMED(...) ((void)0)
#define ROS_DEBUG_THROTTLE(...) ((void)0)
#define ROS_INFO_THROTTLE(...) ((void)0)
#define ROS_WARN_THROTTLE(...) ((void)0)
#define ROS_ERROR_THROTTLE(...) ((void)0)
#define ROS_INFO_ONCE(...) ((void)0)
#define ROS_WARN_ONCE(...) ((void)0)
#define ROS_INFO_COND(...) ((void)0)
#define ROS_DEBUG_COND(...) ((void)0)
