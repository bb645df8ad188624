// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <cmath>
#include <stdexcept>
#include <string>
#include <ros/time.h>
#include <geometry_msgs/PoseStamped.h>
#include <geometry_msgs/Quaternion.h>
namespace tf {
struct Vector3 { double v[3]; Vector3(double x = 0, double y = 0, double z = 0) { v[0] = x; v[1] = y; v[2] = z; } double getX() const { return v[0]; } double getY() const { return v[1]; } double getZ() const { return v[2]; } double x() const { return v[0]; } double y() const { return v[1]; } double z() const { return v[2]; }
  void setX(double a) { v[0] = a; } void setY(double a) { v[1] = a; } void setZ(double a) { v[2] = a; } double length() const { return 0; } Vector3 operator-(const Vector3&) const { return Vector3(); } };
typedef Vector3 Point;
struct Quaternion { double q[4]; Quaternion(double x = 0, double y = 0, double z = 0, double w = 1) { q[0] = x; q[1] = y; q[2] = z; q[3] = w; } void setRPY(double, double, double) {} double getAngle() const { return 0; } };
struct Matrix3x3 { Matrix3x3() {} void setRotation(const Quaternion&) {} Matrix3x3(const Quaternion&) {} void getRPY(double& r, double& p, double& y) const { r = p = y = 0; } void getEulerYPR(double& y, double& p, double& r) const { r = p = y = 0; } };
struct Transform { Vector3 o; Quaternion r; Transform() {} Transform(const Quaternion& q, const Vector3& v = Vector3()) : o(v), r(q) {} const Vector3& getOrigin() const { return o; } Vector3& getOrigin() { return o; } Quaternion getRotation() const { return r; } Matrix3x3 getBasis() const { return Matrix3x3(r); }
  void setOrigin(const Vector3& v) { o = v; } void setRotation(const Quaternion& q) { r = q; } void setIdentity() {} Transform inverse() const { return *this; } Transform operator*(const Transform&) const { return *this; } Vector3 operator*(const Vector3& v) const { return v; } void setBasis(const Matrix3x3&) {} };
typedef Transform Pose;
template <class T> struct Stamped : public T { ros::Time stamp_; std::string frame_id_; Stamped() {} Stamped(const T& t, const ros::Time& s, const std::string& f) : T(t), stamp_(s), frame_id_(f) {} void setData(const T& t) { *static_cast<T*>(this) = t; } };
struct StampedTransform : public Transform { ros::Time stamp_; std::string frame_id_, child_frame_id_; };
inline double getYaw(const Quaternion&) { return 0; }
inline double getYaw(const geometry_msgs::Quaternion&) { return 0; }
inline Quaternion createQuaternionFromYaw(double) { return Quaternion(); }
inline Quaternion createQuaternionFromRPY(double, double, double) { return Quaternion(); }
inline Quaternion createIdentityQuaternion() { return Quaternion(); }
inline geometry_msgs::Quaternion createQuaternionMsgFromYaw(double) { return geometry_msgs::Quaternion(); }
inline void poseStampedMsgToTF(const geometry_msgs::PoseStamped&, Stamped<Pose>&) {}
inline void poseStampedTFToMsg(const Stamped<Pose>&, geometry_msgs::PoseStamped&) {}
inline void poseMsgToTF(const geometry_msgs::Pose&, Pose&) {}
inline void poseTFToMsg(const Pose&, geometry_msgs::Pose&) {}
inline void quaternionMsgToTF(const geometry_msgs::Quaternion&, Quaternion&) {}
struct TransformException : public std::runtime_error { TransformException(const std::string& s) : std::runtime_error(s) {} };
typedef TransformException LookupException; typedef TransformException ConnectivityException; typedef TransformException ExtrapolationException;
}
