// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <memory>
namespace boost { using std::shared_ptr; using std::make_shared; using std::dynamic_pointer_cast; using std::static_pointer_cast; using std::const_pointer_cast; }
