// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <cmath>
namespace angles { inline double normalize_angle(double a) { return a; } inline double shortest_angular_distance(double a, double b) { return b - a; } inline double from_degrees(double d) { return d; } inline double to_degrees(double r) { return r; } }
