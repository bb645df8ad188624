// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <string>
#include <vector>
#include <boost/shared_ptr.hpp>
namespace pluginlib { template <class T> class ClassLoader { public: ClassLoader(const std::string&, const std::string&) {} boost::shared_ptr<T> createInstance(const std::string&) { return boost::shared_ptr<T>(); } std::vector<std::string> getDeclaredClasses() { return std::vector<std::string>(); } std::string getName(const std::string& s) { return s; } bool isClassAvailable(const std::string&) { return true; } }; struct PluginlibException : public std::runtime_error { PluginlibException(const std::string& s) : std::runtime_error(s) {} }; }
