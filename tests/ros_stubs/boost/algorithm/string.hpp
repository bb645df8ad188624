// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <string>
#include <vector>
namespace boost {
struct is_any_of_t { std::string s; };
inline is_any_of_t is_any_of(const std::string& s) { is_any_of_t t; t.s = s; return t; }
enum token_compress_mode_type { token_compress_on, token_compress_off };
template <class V> void split(V& out, const std::string& in, const is_any_of_t& sep, token_compress_mode_type = token_compress_off) {
  out.clear(); std::string cur; for (char c : in) { if (sep.s.find(c) != std::string::npos) { out.push_back(cur); cur.clear(); } else cur += c; } out.push_back(cur); }
inline void trim(std::string&) {}
}
