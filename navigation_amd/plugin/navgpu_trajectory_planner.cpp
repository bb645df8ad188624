// See navgpu_trajectory_planner.h.  Comments cite the reference lines each block stands in for.
#include "navgpu_trajectory_planner.h"

#include <boost/algorithm/string.hpp>

#include <sstream>
#include <stdexcept>

namespace navgpu {

TrajectoryPlanner::TrajectoryPlanner(const costmap_2d::Costmap2D& costmap, const std::vector<geometry_msgs::Point>& footprint_spec,
                                     bool meter_scoring, double sim_period)
    : costmap_(costmap), fleet_(NULL), meter_scoring_(meter_scoring), sim_period_(sim_period) {
  navgpu_fleet_desc d = {};
  d.n_instances = 1;
  d.size_x = costmap.getSizeInCellsX();
  d.size_y = costmap.getSizeInCellsY();
  d.resolution = costmap.getResolution();
  d.layers = NAVGPU_LAYER_OBSTACLE;  // planner only: the master grid is uploaded each cycle
  d.max_plan = 4096;
  d.max_footprint = 32;
  d.max_sim_steps = 512;
  if (navgpu_fleet_create(&d, &fleet_) != NAVGPU_OK) throw std::runtime_error(std::string("navgpu: ") + navgpu_last_error());
  setFootprint(footprint_spec);
}

TrajectoryPlanner::~TrajectoryPlanner() {
  if (fleet_) navgpu_fleet_destroy(fleet_);
}

void TrajectoryPlanner::setFootprint(const std::vector<geometry_msgs::Point>& footprint) {
  std::vector<double> xy;
  for (size_t i = 0; i < footprint.size(); ++i) {
    xy.push_back(footprint[i].x);
    xy.push_back(footprint[i].y);
  }
  navgpu_set_footprint(fleet_, 0, 1, xy.empty() ? NULL : &xy[0], (uint32_t)footprint.size());
}

void TrajectoryPlanner::reconfigure(base_local_planner::BaseLocalPlannerConfig& config) {
  navgpu_tp_config c = {};
  c.acc_lim_x = config.acc_lim_x;
  c.acc_lim_y = config.acc_lim_y;
  c.acc_lim_theta = config.acc_lim_theta;
  c.max_vel_x = config.max_vel_x;
  c.min_vel_x = config.min_vel_x;
  c.max_vel_th = config.max_vel_theta;
  c.min_vel_th = config.min_vel_theta;
  c.min_in_place_vel_th = config.min_in_place_vel_theta;
  c.sim_time = config.sim_time;
  c.sim_granularity = config.sim_granularity;
  c.angular_sim_granularity = config.angular_sim_granularity;
  c.pdist_scale = config.pdist_scale;
  c.gdist_scale = config.gdist_scale;
  c.occdist_scale = config.occdist_scale;
  if (meter_scoring_) {  // :81-87
    const double resolution = costmap_.getResolution();
    c.gdist_scale *= resolution;
    c.pdist_scale *= resolution;
    c.occdist_scale *= resolution;
  }
  c.oscillation_reset_dist = config.oscillation_reset_dist;
  c.escape_reset_dist = config.escape_reset_dist;
  c.escape_reset_theta = config.escape_reset_theta;
  c.vx_samples = config.vx_samples;          // <= 0 becomes 1 inside navgpu_tp_configure, as :98-107
  c.vtheta_samples = config.vtheta_samples;
  c.heading_lookahead = config.heading_lookahead;
  c.holonomic_robot = config.holonomic_robot;
  c.backup_vel = config.escape_vel;
  c.dwa = config.dwa;
  c.heading_scoring = config.heading_scoring;
  c.simple_attractor = config.simple_attractor;
  c.heading_scoring_timestep = config.heading_scoring_timestep;  // :119
  c.sim_period = sim_period_;
  c.allow_unknown = costmap_2d::Costmap2D(costmap_).getDefaultValue() == 0 ? 0 : 1;  // trajectory_planner.cpp:196
  std::vector<std::string> y_strs;  // :124-139
  std::string y_string = config.y_vels;
  boost::split(y_strs, y_string, boost::is_any_of(", "), boost::token_compress_on);
  for (size_t i = 0; i < y_strs.size() && c.n_y_vels < 8; ++i) {
    std::istringstream iss(y_strs[i]);
    double v;
    iss >> v;
    c.y_vels[c.n_y_vels++] = v;
  }
  if (navgpu_tp_configure(fleet_, &c) != NAVGPU_OK) throw std::runtime_error(std::string("navgpu: ") + navgpu_last_error());
  cfg_ = c;
}

bool TrajectoryPlanner::uploadCostmap() {
  double origin[2] = {costmap_.getOriginX(), costmap_.getOriginY()};
  return navgpu_fleet_set_origin(fleet_, 0, 1, origin) == NAVGPU_OK &&
         navgpu_grid_upload(fleet_, NAVGPU_GRID_MASTER, 0, 1, costmap_.getCharMap()) == NAVGPU_OK;
}

void TrajectoryPlanner::updatePlan(const std::vector<geometry_msgs::PoseStamped>& new_plan, bool compute_dists) {
  std::vector<double> xy;
  for (size_t i = 0; i < new_plan.size(); ++i) {
    xy.push_back(new_plan[i].pose.position.x);
    xy.push_back(new_plan[i].pose.position.y);
  }
  if (compute_dists) uploadCostmap();
  navgpu_tp_update_plan(fleet_, 0, xy.empty() ? NULL : &xy[0], (uint32_t)new_plan.size(), compute_dists ? 1 : 0);
}

base_local_planner::Trajectory TrajectoryPlanner::findBestPath(tf::Stamped<tf::Pose> global_pose, tf::Stamped<tf::Pose> global_vel,
                                                               tf::Stamped<tf::Pose>& drive_velocities) {
  base_local_planner::Trajectory best;
  best.cost_ = -1.0;
  if (!uploadCostmap()) {
    drive_velocities.setIdentity();
    return best;
  }
  navgpu_robot_state st = {};  // Eigen::Vector3f pos / vel (:911-912)
  st.pos[0] = global_pose.getOrigin().getX();
  st.pos[1] = global_pose.getOrigin().getY();
  st.pos[2] = tf::getYaw(global_pose.getRotation());
  st.vel[0] = global_vel.getOrigin().getX();
  st.vel[1] = global_vel.getOrigin().getY();
  st.vel[2] = tf::getYaw(global_vel.getRotation());
  navgpu_tp_result r;
  if (navgpu_tp_find_best_path(fleet_, 0, 1, &st, &r) != NAVGPU_OK) {
    drive_velocities.setIdentity();
    return best;
  }
  best.xv_ = r.xv;
  best.yv_ = r.yv;
  best.thetav_ = r.thetav;
  best.cost_ = r.cost;
  std::vector<double> pts(3 * (size_t)std::max(r.n_points, 1));
  const int n = navgpu_tp_trajectory(fleet_, 0, &pts[0], (uint32_t)(pts.size() / 3));
  for (int i = 0; i < n; ++i) best.addPoint(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]);
  if (best.cost_ < 0) {  // :969-978
    drive_velocities.setIdentity();
  } else {
    tf::Vector3 start(best.xv_, best.yv_, 0);
    drive_velocities.setOrigin(start);
    tf::Matrix3x3 matrix;
    matrix.setRotation(tf::createQuaternionFromYaw(best.thetav_));
    drive_velocities.setBasis(matrix);
  }
  return best;
}

double TrajectoryPlanner::scoreTrajectory(double x, double y, double theta, double vx, double vy, double vtheta, double vx_samp,
                                          double vy_samp, double vtheta_samp) {
  const double pose[3] = {x, y, theta}, vel[3] = {vx, vy, vtheta}, vs[3] = {vx_samp, vy_samp, vtheta_samp};
  double cost = -1.0;
  if (navgpu_tp_score_trajectory(fleet_, 0, pose, vel, vs, &cost) != NAVGPU_OK) return -1.0;
  return cost;
}

bool TrajectoryPlanner::checkTrajectory(double x, double y, double theta, double vx, double vy, double vtheta, double vx_samp,
                                        double vy_samp, double vtheta_samp) {
  return scoreTrajectory(x, y, theta, vx, vy, vtheta, vx_samp, vy_samp, vtheta_samp) >= 0;  // :502-516
}

}  // namespace navgpu
