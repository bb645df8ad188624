// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <mutex>
#include <thread>
#include <condition_variable>
#include <boost/shared_ptr.hpp>
#include <boost/function.hpp>
#include <boost/bind.hpp>
namespace boost {
using std::mutex; using std::recursive_mutex; using std::thread; using std::condition_variable; using std::condition_variable_any;
template <class M> using unique_lock = std::unique_lock<M>;
template <class M> using lock_guard = std::lock_guard<M>;
typedef std::recursive_mutex shared_mutex;
template <class M> using shared_lock = std::unique_lock<M>;
namespace this_thread { inline void yield() {} }
}
