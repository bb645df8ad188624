// See navgpu_dwa_planner_ros.h.  Mirrors dwa_local_planner/src/dwa_planner_ros.cpp call for call;
// comments cite the reference lines each block stands in for.
#include "navgpu_dwa_planner_ros.h"

#include <base_local_planner/goal_functions.h>
#include <pluginlib/class_list_macros.h>

PLUGINLIB_EXPORT_CLASS(navgpu::DWAPlannerROS, nav_core::BaseLocalPlanner)  // dwa_planner_ros.cpp:50

namespace navgpu {

DWAPlannerROS::DWAPlannerROS() : tf_(NULL), costmap_ros_(NULL), odom_helper_("odom"), dsrv_(NULL), setup_(false),
                                 initialized_(false), fleet_(NULL), sim_period_(0.05) {}
DWAPlannerROS::~DWAPlannerROS() {
  delete dsrv_;
  if (fleet_) navgpu_fleet_destroy(fleet_);
}

void DWAPlannerROS::initialize(std::string name, tf::TransformListener* tf, costmap_2d::Costmap2DROS* costmap_ros) {
  if (initialized_) return;  // dwa_planner_ros.cpp:98-128
  ros::NodeHandle private_nh("~/" + name);
  tf_ = tf;
  costmap_ros_ = costmap_ros;
  costmap_ros_->getRobotPose(current_pose_);
  costmap_2d::Costmap2D* costmap = costmap_ros_->getCostmap();
  planner_util_.initialize(tf, costmap, costmap_ros_->getGlobalFrameID());

  // sim_period from controller_frequency (dwa_planner.cpp:134-150)
  std::string freq_name;
  if (private_nh.searchParam("controller_frequency", freq_name)) {
    double f = 20.0;
    private_nh.param(freq_name, f, 20.0);
    sim_period_ = f > 0 ? 1.0 / f : 0.05;
  }
  navgpu_fleet_desc d = {};
  d.n_instances = 1;
  d.size_x = costmap->getSizeInCellsX();
  d.size_y = costmap->getSizeInCellsY();
  d.resolution = costmap->getResolution();
  d.layers = NAVGPU_LAYER_OBSTACLE;  // planner only: the master grid is uploaded each cycle
  d.max_plan = 4096;
  d.max_footprint = 32;
  d.max_sim_steps = 256;
  if (navgpu_fleet_create(&d, &fleet_) != NAVGPU_OK)
    throw std::runtime_error(std::string("navgpu: ") + navgpu_last_error());  // init failures throw (obstacle_layer.cpp:113-114)

  if (private_nh.hasParam("odom_topic")) {
    std::string odom_topic;
    private_nh.getParam("odom_topic", odom_topic);
    odom_helper_.setOdomTopic(odom_topic);
  }
  initialized_ = true;
  dsrv_ = new dynamic_reconfigure::Server<dwa_local_planner::DWAPlannerConfig>(private_nh);
  dsrv_->setCallback(boost::bind(&DWAPlannerROS::reconfigureCB, this, _1, _2));
}

void DWAPlannerROS::reconfigureCB(dwa_local_planner::DWAPlannerConfig& config, uint32_t) {
  boost::unique_lock<boost::mutex> l(configuration_mutex_);  // DWAPlanner::reconfigure (dwa_planner.cpp:55)
  if (setup_ && config.restore_defaults) {  // dwa_planner_ros.cpp:54-62
    config = default_config_;
    config.restore_defaults = false;
  }
  if (!setup_) {
    default_config_ = config;
    setup_ = true;
  }
  base_local_planner::LocalPlannerLimits limits;  // :64-83
  limits.max_trans_vel = config.max_trans_vel;
  limits.min_trans_vel = config.min_trans_vel;
  limits.max_vel_x = config.max_vel_x;
  limits.min_vel_x = config.min_vel_x;
  limits.max_vel_y = config.max_vel_y;
  limits.min_vel_y = config.min_vel_y;
  limits.max_rot_vel = config.max_rot_vel;
  limits.min_rot_vel = config.min_rot_vel;
  limits.acc_lim_x = config.acc_lim_x;
  limits.acc_lim_y = config.acc_lim_y;
  limits.acc_lim_theta = config.acc_lim_theta;
  limits.acc_limit_trans = config.acc_limit_trans;
  limits.xy_goal_tolerance = config.xy_goal_tolerance;
  limits.yaw_goal_tolerance = config.yaw_goal_tolerance;
  limits.prune_plan = config.prune_plan;
  limits.trans_stopped_vel = config.trans_stopped_vel;
  limits.rot_stopped_vel = config.rot_stopped_vel;
  planner_util_.reconfigureCB(limits, config.restore_defaults);

  navgpu_dwa_config& c = cfg_;  // DWAPlanner::reconfigure (dwa_planner.cpp:52-116)
  c.max_trans_vel = config.max_trans_vel;  c.min_trans_vel = config.min_trans_vel;
  c.max_vel_x = config.max_vel_x;          c.min_vel_x = config.min_vel_x;
  c.max_vel_y = config.max_vel_y;          c.min_vel_y = config.min_vel_y;
  c.max_rot_vel = config.max_rot_vel;      c.min_rot_vel = config.min_rot_vel;
  c.acc_lim_x = config.acc_lim_x;  c.acc_lim_y = config.acc_lim_y;  c.acc_lim_theta = config.acc_lim_theta;
  c.sim_time = config.sim_time;  c.sim_granularity = config.sim_granularity;
  c.angular_sim_granularity = config.angular_sim_granularity;  c.sim_period = sim_period_;
  c.path_distance_bias = config.path_distance_bias;  c.goal_distance_bias = config.goal_distance_bias;
  c.occdist_scale = config.occdist_scale;  c.forward_point_distance = config.forward_point_distance;
  c.oscillation_reset_dist = config.oscillation_reset_dist;  c.oscillation_reset_angle = config.oscillation_reset_angle;
  c.vx_samples = config.vx_samples;  c.vy_samples = config.vy_samples;  c.vth_samples = config.vth_samples;
  c.use_dwa = config.use_dwa;
  c.discretize_by_time = 0;  // DWAPlanner::findBestPath passes the default false (dwa_planner.cpp:310-314)
  ros::NodeHandle nh("~");
  bool sum_scores = false;  double cheat = 1.0;  bool allow_unknown = true;
  nh.param("sum_scores", sum_scores, false);        // dwa_planner.cpp:155-157
  nh.param("cheat_factor", cheat, 1.0);             // :181
  nh.param("navgpu_allow_unknown", allow_unknown, true);  // explicit here; uninitialised in the reference
  int rollout_trig = 0;
  nh.param("navgpu_rollout_trig", rollout_trig, 0);       // 0: cos(pos[2]) is ::cos(double) in the reference build being replaced, 1: the float overload
  c.sum_scores = sum_scores;  c.cheat_factor = cheat;  c.allow_unknown = allow_unknown;  c.rollout_trig = rollout_trig;
  if (navgpu_planner_configure(fleet_, &c) != NAVGPU_OK) ROS_ERROR("navgpu_planner_configure: %s", navgpu_last_error());
}

bool DWAPlannerROS::setPlan(const std::vector<geometry_msgs::PoseStamped>& orig_global_plan) {
  if (!initialized_) return false;  // dwa_planner_ros.cpp:131-142
  boost::unique_lock<boost::mutex> l(configuration_mutex_);
  latchedStopRotateController_.resetLatching();
  navgpu_planner_set_plan(fleet_, 0, 1);  // DWAPlanner::setPlan: resetOscillationFlags
  return planner_util_.setPlan(orig_global_plan);
}

bool DWAPlannerROS::isGoalReached() {  // dwa_planner_ros.cpp:144-160
  if (!initialized_) return false;
  if (!costmap_ros_->getRobotPose(current_pose_)) return false;
  return latchedStopRotateController_.isGoalReached(&planner_util_, odom_helper_, current_pose_);
}

bool DWAPlannerROS::uploadCostmap() {
  // the planner borrows the Costmap2D for life (dwa_planner_ros.cpp:104-110); move_base holds its
  // mutex around computeVelocityCommands (move_base.cpp:947), so the bytes are stable here.
  costmap_2d::Costmap2D* cm = costmap_ros_->getCostmap();
  double origin[2] = {cm->getOriginX(), cm->getOriginY()};
  return navgpu_fleet_set_origin(fleet_, 0, 1, origin) == NAVGPU_OK &&
         navgpu_grid_upload(fleet_, NAVGPU_GRID_MASTER, 0, 1, cm->getCharMap()) == NAVGPU_OK;
}

bool DWAPlannerROS::gpuStage(const tf::Stamped<tf::Pose>& pose, const tf::Stamped<tf::Pose>& vel,
                             const std::vector<geometry_msgs::PoseStamped>& plan) {
  // what every cycle does before it branches (dwa_planner_ros.cpp:268-274): the costmap the planner borrows, the
  // footprint, and DWAPlanner::updatePlanAndLocalCosts (target poses, nose goal, alignment switch).  No wavefront runs here.
  std::vector<geometry_msgs::Point> fp = costmap_ros_->getRobotFootprint();
  std::vector<double> fxy;
  for (size_t i = 0; i < fp.size(); ++i) { fxy.push_back(fp[i].x); fxy.push_back(fp[i].y); }
  std::vector<double> pxy;
  for (size_t i = 0; i < plan.size(); ++i) { pxy.push_back(plan[i].pose.position.x); pxy.push_back(plan[i].pose.position.y); }
  navgpu_robot_state st;  // Eigen::Vector3f narrowing of dwa_planner.cpp:303-304
  st.pos[0] = pose.getOrigin().getX(); st.pos[1] = pose.getOrigin().getY(); st.pos[2] = tf::getYaw(pose.getRotation());
  st.vel[0] = vel.getOrigin().getX();  st.vel[1] = vel.getOrigin().getY();  st.vel[2] = tf::getYaw(vel.getRotation());
  st.plan_first = 0; st.plan_count = plan.size();
  staged_ = st;
  return uploadCostmap() && navgpu_set_footprint(fleet_, 0, 1, fxy.empty() ? NULL : &fxy[0], fp.size()) == NAVGPU_OK &&
         navgpu_planner_stage(fleet_, 0, 1, &st, &pxy[0], plan.size()) == NAVGPU_OK;
}

bool DWAPlannerROS::gpuFindBestPath(navgpu_plan_result* out) {  // DWAPlanner::findBestPath (dwa_planner.cpp:292-371)
  return navgpu_planner_cycle(fleet_, 0, 1) == NAVGPU_OK && navgpu_planner_results(fleet_, 0, 1, out) == NAVGPU_OK;
}

bool DWAPlannerROS::gpuCheckTrajectory(Eigen::Vector3f pos, Eigen::Vector3f vel, Eigen::Vector3f vs) {
  // LatchedStopRotateController's collision oracle (dwa_planner_ros.cpp:281-285 -> DWAPlanner::checkTrajectory,
  // dwa_planner.cpp:213-237): scores ONE sample from (pos, vel) against the MapGrids of the last findBestPath - the
  // reference does not refresh them here either.  The controller passes the current pose and odometry velocity
  // (latched_stop_rotate_controller.cpp:126-131, 167-172); should they differ from what this cycle staged, they are re-staged.
  bool same = true;
  for (int k = 0; k < 3; ++k) same = same && staged_.pos[k] == pos[k] && staged_.vel[k] == vel[k];
  if (!same) {
    float p[3] = {pos[0], pos[1], pos[2]}, v0[3] = {vel[0], vel[1], vel[2]};
    if (navgpu_planner_stage_poses(fleet_, 0, 1, p, v0) != NAVGPU_OK) return false;
    for (int k = 0; k < 3; ++k) { staged_.pos[k] = pos[k]; staged_.vel[k] = vel[k]; }
  }
  float v[3] = {vs[0], vs[1], vs[2]};
  int32_t ok = 0;
  return navgpu_planner_check_trajectory(fleet_, 0, v, &ok) == NAVGPU_OK && ok;
}

bool DWAPlannerROS::computeVelocityCommands(geometry_msgs::Twist& cmd_vel) {  // dwa_planner_ros.cpp:252-300
  if (!costmap_ros_->getRobotPose(current_pose_)) return false;
  std::vector<geometry_msgs::PoseStamped> transformed_plan;
  if (!planner_util_.getLocalPlan(current_pose_, transformed_plan)) return false;
  if (transformed_plan.empty()) return false;
  tf::Stamped<tf::Pose> robot_vel;
  odom_helper_.getRobotVel(robot_vel);
  // one lock over stage + (stop-rotate | findBestPath): the reference takes configuration_mutex_ inside findBestPath
  // (dwa_planner.cpp:301); its updatePlanAndLocalCosts reads no reconfigurable state that the GPU tables depend on,
  // ours stages the nose goal and the wavefront boxes from cfg_, so the lock starts one call earlier
  boost::unique_lock<boost::mutex> l(configuration_mutex_);
  if (!gpuStage(current_pose_, robot_vel, transformed_plan)) return false;  // dp_->updatePlanAndLocalCosts (:274)

  if (latchedStopRotateController_.isPositionReached(&planner_util_, current_pose_)) {  // :276-288
    base_local_planner::LocalPlannerLimits limits = planner_util_.getCurrentLimits();
    return latchedStopRotateController_.computeVelocityCommandsStopRotate(
        cmd_vel, limits.getAccLimits(), sim_period_, &planner_util_, odom_helper_, current_pose_,
        boost::bind(&DWAPlannerROS::gpuCheckTrajectory, this, _1, _2, _3));
  }
  navgpu_plan_result r;
  if (!gpuFindBestPath(&r)) return false;
  cmd_vel.linear.x = r.drive[0];   // dwa_planner_ros.cpp:210-212
  cmd_vel.linear.y = r.drive[1];
  cmd_vel.angular.z = r.drive[2];  // yaw of createQuaternionFromYaw(thetav) == thetav
  return r.cost >= 0;              // :215-222 (the empty local-plan publish is visualisation, out of scope)
}

}  // namespace navgpu
