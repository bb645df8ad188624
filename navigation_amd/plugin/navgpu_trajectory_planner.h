// navgpu::TrajectoryPlanner — stand-in for base_local_planner::TrajectoryPlanner (the `tc_` member of
// TrajectoryPlannerROS, base_local_planner/include/base_local_planner/trajectory_planner.h:63-231) over the
// navgpu C-ABI: same method names and argument meaning, so TrajectoryPlannerROS (trajectory_planner_ros.cpp)
// keeps its own goal-tolerance / stop / rotate logic and only its `tc_->...` calls change target:
//   tc_->reconfigure(cfg)                      -> navgpu_tp_configure
//   tc_->updatePlan(plan, compute_dists)       -> navgpu_tp_update_plan
//   tc_->findBestPath(pose, vel, drive_cmds)   -> navgpu_tp_find_best_path + navgpu_tp_trajectory
//   tc_->checkTrajectory / scoreTrajectory     -> navgpu_tp_score_trajectory
// Source-only in this repository (needs the ROS headers).
#ifndef NAVGPU_TRAJECTORY_PLANNER_H_
#define NAVGPU_TRAJECTORY_PLANNER_H_

#include <base_local_planner/BaseLocalPlannerConfig.h>
#include <base_local_planner/trajectory.h>
#include <costmap_2d/costmap_2d.h>
#include <geometry_msgs/Point.h>
#include <geometry_msgs/PoseStamped.h>
#include <tf/transform_datatypes.h>

#include <navgpu.h>

#include <vector>

namespace navgpu {

class TrajectoryPlanner {
 public:
  // the costmap is borrowed (as the reference borrows it); its bytes are uploaded before every findBestPath
  TrajectoryPlanner(const costmap_2d::Costmap2D& costmap, const std::vector<geometry_msgs::Point>& footprint_spec,
                    bool meter_scoring, double sim_period);
  ~TrajectoryPlanner();

  void reconfigure(base_local_planner::BaseLocalPlannerConfig& cfg);                       // trajectory_planner.cpp:58-141
  void updatePlan(const std::vector<geometry_msgs::PoseStamped>& new_plan, bool compute_dists = false);  // :474-500
  base_local_planner::Trajectory findBestPath(tf::Stamped<tf::Pose> global_pose, tf::Stamped<tf::Pose> global_vel,
                                              tf::Stamped<tf::Pose>& drive_velocities);    // :908-984
  bool checkTrajectory(double x, double y, double theta, double vx, double vy, double vtheta, double vx_samp, double vy_samp,
                       double vtheta_samp);                                                // :502-516
  double scoreTrajectory(double x, double y, double theta, double vx, double vy, double vtheta, double vx_samp, double vy_samp,
                         double vtheta_samp);                                              // :518-531
  void setFootprint(const std::vector<geometry_msgs::Point>& footprint);                  // trajectory_planner.h:214-216

 private:
  bool uploadCostmap();
  const costmap_2d::Costmap2D& costmap_;
  navgpu_fleet* fleet_;
  navgpu_tp_config cfg_;
  bool meter_scoring_;
  double sim_period_;
};

}  // namespace navgpu
#endif
