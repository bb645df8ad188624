// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <stdint.h>
namespace ros {
struct Duration { double s; Duration(double d = 0) : s(d) {} double toSec() const { return s; } bool sleep() const { return true; } };
struct Time { double s; Time(double d = 0) : s(d) {} static Time now() { return Time(); } double toSec() const { return s; }
  Duration operator-(const Time& o) const { return Duration(s - o.s); } Time operator+(const Duration& d) const { return Time(s + d.s); }
  bool operator<(const Time& o) const { return s < o.s; } bool operator>(const Time& o) const { return s > o.s; } bool operator==(const Time& o) const { return s == o.s; } };
struct Rate { Rate(double) {} bool sleep() { return true; } Duration cycleTime() const { return Duration(); } };
struct WallTime { static WallTime now() { return WallTime(); } double toSec() const { return 0; } };
}
