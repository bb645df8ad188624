// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
#include <string>
namespace base_local_planner { struct BaseLocalPlannerConfig { double acc_lim_x, acc_lim_y, acc_lim_theta, max_vel_x, min_vel_x, max_vel_theta, min_vel_theta, min_in_place_vel_theta, sim_time, sim_granularity, angular_sim_granularity, pdist_scale, gdist_scale, occdist_scale, oscillation_reset_dist, escape_reset_dist, escape_reset_theta, heading_lookahead, escape_vel, stop_time_buffer; int vx_samples, vtheta_samples, heading_scoring_timestep; bool holonomic_robot, heading_scoring, simple_attractor, dwa, meter_scoring, restore_defaults; std::string y_vels; BaseLocalPlannerConfig() {} }; }
