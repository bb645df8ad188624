// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <vector>
#include <stdint.h>
#include <boost/shared_ptr.hpp>
#include <std_msgs/Header.h>
namespace pcl { struct PCLHeader { uint32_t seq; uint64_t stamp; std::string frame_id; }; template <class T> struct PointCloud { PCLHeader header; std::vector<T> points; uint32_t width, height; bool is_dense; typedef boost::shared_ptr<PointCloud<T> > Ptr; typedef boost::shared_ptr<const PointCloud<T> > ConstPtr; size_t size() const { return points.size(); } void push_back(const T& p) { points.push_back(p); } void clear() { points.clear(); } T& operator[](size_t i) { return points[i]; } const T& operator[](size_t i) const { return points[i]; } }; }
