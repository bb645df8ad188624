// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <string>
namespace costmap_2d { struct Costmap2DConfig { double transform_tolerance, update_frequency, publish_frequency, resolution, origin_x, origin_y, robot_radius, footprint_padding; int width, height; std::string footprint; Costmap2DConfig() {} }; }
