// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <tf/tf.h>
#include <sensor_msgs/PointCloud.h>
namespace tf { class Transformer { public: virtual ~Transformer() {} }; class TransformListener : public Transformer { public:
  TransformListener() {} TransformListener(ros::Duration) {}
  void transformPose(const std::string&, const Stamped<Pose>&, Stamped<Pose>&) const {}
  void transformPose(const std::string&, const geometry_msgs::PoseStamped&, geometry_msgs::PoseStamped&) const {}
  template <class A, class B> void transformPoint(const std::string&, const A&, B&) const {}
  void transformPointCloud(const std::string&, const sensor_msgs::PointCloud&, sensor_msgs::PointCloud&) const {}
  void lookupTransform(const std::string&, const std::string&, const ros::Time&, StampedTransform&) const {}
  void lookupTransform(const std::string&, const ros::Time&, const std::string&, const ros::Time&, const std::string&, StampedTransform&) const {}
  bool waitForTransform(const std::string&, const std::string&, const ros::Time&, const ros::Duration&, const ros::Duration& = ros::Duration(), std::string* = 0) const { return true; }
  bool canTransform(const std::string&, const std::string&, const ros::Time&, std::string* = 0) const { return true; }
  std::string resolve(const std::string& s) const { return s; } }; }
