// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
namespace costmap_2d { struct ObstaclePluginConfig { bool enabled, footprint_clearing_enabled; double max_obstacle_height; int combination_method; ObstaclePluginConfig() : enabled(true), footprint_clearing_enabled(true), max_obstacle_height(2), combination_method(1) {} }; }
