// navgpu::DWAPlannerROS — nav_core::BaseLocalPlanner adapter over the navgpu C-ABI.
// Drop-in for dwa_local_planner::DWAPlannerROS (dwa_local_planner/src/dwa_planner_ros.cpp): same
// plugin base class, same parameters (read through the same dynamic_reconfigure type), same
// call sequence; only DWAPlanner::findBestPath / checkTrajectory run on the GPU.
// Source-only in this repository: ROS (roscpp, tf, pluginlib, costmap_2d, base_local_planner
// headers) is not installed in the build image, so this file is compiled in a catkin workspace
// (see INTEGRATION.md), not here.
#ifndef NAVGPU_DWA_PLANNER_ROS_H_
#define NAVGPU_DWA_PLANNER_ROS_H_

#include <base_local_planner/latched_stop_rotate_controller.h>
#include <base_local_planner/local_planner_util.h>
#include <base_local_planner/odometry_helper_ros.h>
#include <costmap_2d/costmap_2d_ros.h>
#include <dwa_local_planner/DWAPlannerConfig.h>
#include <dynamic_reconfigure/server.h>
#include <nav_core/base_local_planner.h>
#include <tf/transform_listener.h>

#include <boost/thread.hpp>

#include <navgpu.h>

namespace navgpu {

class DWAPlannerROS : public nav_core::BaseLocalPlanner {
 public:
  DWAPlannerROS();
  ~DWAPlannerROS();
  // nav_core/include/nav_core/base_local_planner.h:50-87
  void initialize(std::string name, tf::TransformListener* tf, costmap_2d::Costmap2DROS* costmap_ros);
  bool computeVelocityCommands(geometry_msgs::Twist& cmd_vel);
  bool setPlan(const std::vector<geometry_msgs::PoseStamped>& orig_global_plan);
  bool isGoalReached();

 private:
  void reconfigureCB(dwa_local_planner::DWAPlannerConfig& config, uint32_t level);
  bool uploadCostmap();
  bool gpuStage(const tf::Stamped<tf::Pose>& pose, const tf::Stamped<tf::Pose>& vel,
                const std::vector<geometry_msgs::PoseStamped>& local_plan);
  bool gpuFindBestPath(navgpu_plan_result* out);
  bool gpuCheckTrajectory(Eigen::Vector3f pos, Eigen::Vector3f vel, Eigen::Vector3f vel_samples);

  tf::TransformListener* tf_;
  costmap_2d::Costmap2DROS* costmap_ros_;
  base_local_planner::LocalPlannerUtil planner_util_;          // f-1: kept from the reference
  base_local_planner::LatchedStopRotateController latchedStopRotateController_;
  base_local_planner::OdometryHelperRos odom_helper_;
  dynamic_reconfigure::Server<dwa_local_planner::DWAPlannerConfig>* dsrv_;
  dwa_local_planner::DWAPlannerConfig default_config_;
  bool setup_, initialized_;
  // DWAPlanner::configuration_mutex_ (dwa_planner.h:163): reconfigure() takes it (dwa_planner.cpp:55) and so does
  // findBestPath (:301); here it also covers the staging and checkTrajectory, because cfg_ and the fleet's tables are
  // shared between the dynamic_reconfigure (spinner) thread and move_base's control thread.  libnavgpu serialises the
  // calls on a fleet itself (see include/navgpu.h, "Threading"); this mutex keeps a reconfigure from landing BETWEEN the
  // stage and the cycle of one computeVelocityCommands.
  boost::mutex configuration_mutex_;
  navgpu_fleet* fleet_;
  navgpu_dwa_config cfg_;
  navgpu_robot_state staged_;  // pose / velocity of the last stage
  double sim_period_;
  tf::Stamped<tf::Pose> current_pose_;
  std::vector<geometry_msgs::PoseStamped> last_local_plan_;
};

}  // namespace navgpu
#endif
