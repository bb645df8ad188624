// SYNTAX-CHECK STAND-IN for a ROS / Boost / PCL / Eigen header that this image lacks.  Test infrastructure only
// (tests/test_plugin_syntax.py): it lets g++ -fsyntax-only parse navigation_amd/plugin/*.cpp against the REFERENCE'S OWN
// headers.  No reference code is built with it, nothing is linked, nothing here is part of the product.
#pragma once
namespace dwa_local_planner { struct DWAPlannerConfig { double max_trans_vel, min_trans_vel, max_vel_x, min_vel_x, max_vel_y, min_vel_y, max_rot_vel, min_rot_vel, acc_lim_x, acc_lim_y, acc_lim_theta, acc_limit_trans, xy_goal_tolerance, yaw_goal_tolerance, trans_stopped_vel, rot_stopped_vel, sim_time, sim_granularity, angular_sim_granularity, path_distance_bias, goal_distance_bias, occdist_scale, stop_time_buffer, oscillation_reset_dist, oscillation_reset_angle, forward_point_distance, scaling_speed, max_scaling_factor; int vx_samples, vy_samples, vth_samples; bool prune_plan, use_dwa, restore_defaults; DWAPlannerConfig() {} }; }
