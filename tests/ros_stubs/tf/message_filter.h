// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <tf/transform_listener.h>
namespace tf { class MessageFilterBase { public: virtual ~MessageFilterBase() {} virtual void clear() {} virtual void setTargetFrame(const std::string&) {} virtual void setTolerance(const ros::Duration&) {} }; template <class M> class MessageFilter : public MessageFilterBase { public: template <class F> MessageFilter(F&, Transformer&, const std::string&, uint32_t) {} template <class C> void registerCallback(const C&) {} }; }
