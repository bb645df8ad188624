// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
namespace costmap_2d { struct InflationPluginConfig { bool enabled; double cost_scaling_factor, inflation_radius; bool inflate_unknown; InflationPluginConfig() : enabled(true), cost_scaling_factor(10), inflation_radius(0.55), inflate_unknown(false) {} }; }
