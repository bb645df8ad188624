// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <cmath>
#include <cstring>
#include <climits>
#include <cfloat>
#include <limits>
#include <map>
#include <string>
#include <vector>
#include <stdexcept>
#include <stdint.h>
#include <boost/shared_ptr.hpp>
#include <boost/function.hpp>
#include <boost/bind.hpp>
#include <ros/console.h>
#include <ros/assert.h>
#include <ros/time.h>
namespace XmlRpc { class XmlRpcValue { public: enum Type { TypeInvalid, TypeArray, TypeString, TypeDouble, TypeInt, TypeStruct }; typedef std::map<std::string, XmlRpcValue> ValueStruct; typedef std::vector<XmlRpcValue> ValueArray; Type _type; union { ValueStruct* asStruct; ValueArray* asArray; } _value; Type getType() const { return TypeInvalid; } int size() const { return 0; } XmlRpcValue& operator[](int) { return *this; } XmlRpcValue& operator[](const char*) { return *this; } operator double() const { return 0; } operator int() const { return 0; } operator std::string() const { return std::string(); } bool hasMember(const std::string&) const { return false; } }; }
namespace ros {
struct Publisher { template <class M> void publish(const M&) const {} uint32_t getNumSubscribers() const { return 0; } void shutdown() {} operator bool() const { return true; } };
struct Subscriber { void shutdown() {} std::string getTopic() const { return std::string(); } };
struct Timer { void stop() {} void start() {} };
struct TimerEvent {};
struct SingleSubscriberPublisher {};
class NodeHandle {
 public:
  NodeHandle() {}
  NodeHandle(const std::string&) {}
  NodeHandle(const NodeHandle&, const std::string&) {}
  template <class T> bool param(const std::string&, T& v, const T& d) const { v = d; return false; }
  template <class T> bool getParam(const std::string&, T&) const { return false; }
  template <class T> void setParam(const std::string&, const T&) const {}
  bool hasParam(const std::string&) const { return false; }
  bool searchParam(const std::string&, std::string&) const { return false; }
  bool deleteParam(const std::string&) const { return false; }
  std::string getNamespace() const { return std::string(); }
  std::string resolveName(const std::string& n) const { return n; }
  template <class M> Publisher advertise(const std::string&, uint32_t, bool = false) { return Publisher(); }
  template <class M, class C> Publisher advertise(const std::string&, uint32_t, const C&) { return Publisher(); }
  template <class M, class T> Subscriber subscribe(const std::string&, uint32_t, void (T::*)(const boost::shared_ptr<M const>&), T*) { return Subscriber(); }
  template <class M> Subscriber subscribe(const std::string&, uint32_t, const boost::function<void(const boost::shared_ptr<M const>&)>&) { return Subscriber(); }
  template <class T> Timer createTimer(Duration, void (T::*)(const TimerEvent&), T*, bool = false) { return Timer(); }
  bool ok() const { return true; }
};
inline bool ok() { return true; }
inline void spinOnce() {}
namespace this_node { inline std::string getName() { return std::string(); } }
}
