// See navgpu_layers.h.  Mode (A) adapters; every block cites the reference lines it stands in for.
#include "navgpu_layers.h"

#include <costmap_2d/footprint.h>
#include <pluginlib/class_list_macros.h>

PLUGINLIB_EXPORT_CLASS(navgpu::InflationLayer, costmap_2d::Layer)  // inflation_layer.cpp:45
PLUGINLIB_EXPORT_CLASS(navgpu::ObstacleLayer, costmap_2d::Layer)   // obstacle_layer.cpp:43

namespace navgpu {

static navgpu_fleet* makeFleet(costmap_2d::Costmap2D* m, int layers, bool track_unknown) {
  navgpu_fleet_desc d = {};
  d.n_instances = 1;
  d.size_x = m->getSizeInCellsX();
  d.size_y = m->getSizeInCellsY();
  d.resolution = m->getResolution();
  d.layers = layers;
  d.track_unknown = track_unknown;
  d.max_points = 1 << 16;
  d.max_observations = 16;
  d.max_footprint = 32;
  navgpu_fleet* f = NULL;
  if (navgpu_fleet_create(&d, &f) != NAVGPU_OK) throw std::runtime_error(std::string("navgpu: ") + navgpu_last_error());
  double origin[2] = {m->getOriginX(), m->getOriginY()};
  navgpu_fleet_set_origin(f, 0, 1, origin);
  return f;
}

// ----------------------------------------------------------------------------- InflationLayer
InflationLayer::InflationLayer() : fleet_(NULL), need_reinflation_(false), last_min_x_(-FLT_MAX), last_min_y_(-FLT_MAX),
                                   last_max_x_(FLT_MAX), last_max_y_(FLT_MAX), dsrv_(NULL) {
  p_.enabled = 1; p_.reserved = 0; p_.inflation_radius = 0; p_.cost_scaling_factor = 0; p_.inscribed_radius = 0;
}
InflationLayer::~InflationLayer() {
  delete dsrv_;
  if (fleet_) navgpu_fleet_destroy(fleet_);
}
void InflationLayer::onInitialize() {  // inflation_layer.cpp:70-99
  ros::NodeHandle nh("~/" + name_);
  current_ = true;
  need_reinflation_ = false;
  if (!dsrv_) dsrv_ = new dynamic_reconfigure::Server<costmap_2d::InflationPluginConfig>(nh);
  dsrv_->setCallback(boost::bind(&InflationLayer::reconfigureCB, this, _1, _2));
  matchSize();
}
void InflationLayer::reconfigureCB(costmap_2d::InflationPluginConfig& c, uint32_t) {  // :101-109, :362-376
  if (p_.cost_scaling_factor != c.cost_scaling_factor || p_.inflation_radius != c.inflation_radius) need_reinflation_ = true;
  if (enabled_ != c.enabled) need_reinflation_ = true;
  enabled_ = c.enabled;
  p_.enabled = c.enabled; p_.inflation_radius = c.inflation_radius; p_.cost_scaling_factor = c.cost_scaling_factor;
  pushParams();
}
void InflationLayer::pushParams() {
  if (fleet_ && navgpu_inflation_configure(fleet_, &p_) != NAVGPU_OK) ROS_ERROR("navgpu_inflation_configure: %s", navgpu_last_error());
}
void InflationLayer::matchSize() {  // :110-123 — the fleet is tied to the master's geometry
  if (fleet_) navgpu_fleet_destroy(fleet_);
  fleet_ = makeFleet(layered_costmap_->getCostmap(), NAVGPU_LAYER_INFLATION, layered_costmap_->isTrackingUnknown());
  pushParams();
}
void InflationLayer::onFootprintChanged() {  // :160-170
  p_.inscribed_radius = layered_costmap_->getInscribedRadius();
  need_reinflation_ = true;
  pushParams();
}
void InflationLayer::updateBounds(double, double, double, double* min_x, double* min_y, double* max_x, double* max_y) {
  // inflation_layer.cpp:125-158, verbatim semantics (four doubles of host state)
  if (need_reinflation_) {
    last_min_x_ = *min_x; last_min_y_ = *min_y; last_max_x_ = *max_x; last_max_y_ = *max_y;
    *min_x = -std::numeric_limits<float>::max(); *min_y = -std::numeric_limits<float>::max();
    *max_x = std::numeric_limits<float>::max();  *max_y = std::numeric_limits<float>::max();
    need_reinflation_ = false;
  } else {
    double tx0 = last_min_x_, ty0 = last_min_y_, tx1 = last_max_x_, ty1 = last_max_y_;
    last_min_x_ = *min_x; last_min_y_ = *min_y; last_max_x_ = *max_x; last_max_y_ = *max_y;
    *min_x = std::min(tx0, *min_x) - p_.inflation_radius; *min_y = std::min(ty0, *min_y) - p_.inflation_radius;
    *max_x = std::max(tx1, *max_x) + p_.inflation_radius; *max_y = std::max(ty1, *max_y) + p_.inflation_radius;
  }
}
void InflationLayer::updateCosts(costmap_2d::Costmap2D& master, int min_i, int min_j, int max_i, int max_j) {
  if (!enabled_) return;  // :172-266 on the GPU: upload, inflate the box, download
  int32_t box[4] = {min_i, min_j, max_i, max_j};
  if (navgpu_grid_upload(fleet_, NAVGPU_GRID_MASTER, 0, 1, master.getCharMap()) != NAVGPU_OK ||
      navgpu_inflate(fleet_, 0, 1, box) != NAVGPU_OK ||
      navgpu_grid_download(fleet_, NAVGPU_GRID_MASTER, 0, 1, master.getCharMap()) != NAVGPU_OK) {
    ROS_ERROR_THROTTLE(1.0, "navgpu inflation failed: %s", navgpu_last_error());
    current_ = false;  // the reference's failure channel: isCurrent() false stops the robot
  }
}

// ----------------------------------------------------------------------------- ObstacleLayer
ObstacleLayer::~ObstacleLayer() { if (fleet_) navgpu_fleet_destroy(fleet_); }
void ObstacleLayer::onInitialize() {
  costmap_2d::ObstacleLayer::onInitialize();  // topics, buffers, parameters: unchanged reference code
  matchSize();
}
void ObstacleLayer::matchSize() {
  costmap_2d::ObstacleLayer::matchSize();
  if (fleet_) navgpu_fleet_destroy(fleet_);
  fleet_ = makeFleet(layered_costmap_->getCostmap(), NAVGPU_LAYER_OBSTACLE, default_value_ == costmap_2d::NO_INFORMATION);
  navgpu_obstacle_params p = {};
  p.enabled = enabled_; p.footprint_clearing_enabled = footprint_clearing_enabled_; p.combination_method = combination_method_;
  p.max_obstacle_height = max_obstacle_height_;
  navgpu_obstacle_configure(fleet_, &p);
}
void ObstacleLayer::updateBounds(double rx, double ry, double ryaw, double* min_x, double* min_y, double* max_x, double* max_y) {
  if (rolling_window_) updateOrigin(rx - getSizeInMetersX() / 2, ry - getSizeInMetersY() / 2);  // f-2: host shift, then re-upload
  if (!enabled_) return;
  useExtraBounds(min_x, min_y, max_x, max_y);
  std::vector<costmap_2d::Observation> marking, clearing;  // obstacle_layer.cpp:349-359
  bool current = getMarkingObservations(marking);
  current = getClearingObservations(clearing) && current;
  current_ = current;
  std::vector<navgpu_observation> obs;
  std::vector<float> pts;
  for (int pass = 0; pass < 2; ++pass) {
    const std::vector<costmap_2d::Observation>& v = pass ? marking : clearing;
    for (size_t k = 0; k < v.size(); ++k) {
      navgpu_observation o = {};
      o.instance = 0; o.first_point = pts.size() / 3; o.n_points = v[k].cloud_->points.size();
      o.flags = pass ? NAVGPU_OBS_MARKING : NAVGPU_OBS_CLEARING;
      o.origin_x = v[k].origin_.x; o.origin_y = v[k].origin_.y; o.origin_z = v[k].origin_.z;
      o.obstacle_range = v[k].obstacle_range_; o.raytrace_range = v[k].raytrace_range_;
      for (size_t i = 0; i < v[k].cloud_->points.size(); ++i) {
        pts.push_back(v[k].cloud_->points[i].x); pts.push_back(v[k].cloud_->points[i].y); pts.push_back(v[k].cloud_->points[i].z);
      }
      obs.push_back(o);
    }
  }
  std::vector<geometry_msgs::Point> fp = getFootprint();
  std::vector<double> fxy;
  for (size_t i = 0; i < fp.size(); ++i) { fxy.push_back(fp[i].x); fxy.push_back(fp[i].y); }
  double pose[3] = {rx, ry, ryaw}, b[4] = {*min_x, *min_y, *max_x, *max_y};
  navgpu_set_footprint(fleet_, 0, 1, fxy.data(), fp.size());
  if (navgpu_costmap_stage(fleet_, 0, 1, pose, obs.data(), obs.size(), pts.data(), pts.size() / 3) != NAVGPU_OK ||
      navgpu_obstacle_update_bounds(fleet_, 0, 1, b) != NAVGPU_OK) {
    ROS_ERROR_THROTTLE(1.0, "navgpu obstacle update failed: %s", navgpu_last_error());
    current_ = false;
    return;
  }
  *min_x = b[0]; *min_y = b[1]; *max_x = b[2]; *max_y = b[3];
  costmap_2d::transformFootprint(rx, ry, ryaw, fp, transformed_footprint_);
}
void ObstacleLayer::updateCosts(costmap_2d::Costmap2D& master, int min_i, int min_j, int max_i, int max_j) {
  if (!enabled_) return;  // obstacle_layer.cpp:427-448 on the GPU
  int32_t box[4] = {min_i, min_j, max_i, max_j};
  if (navgpu_grid_upload(fleet_, NAVGPU_GRID_MASTER, 0, 1, master.getCharMap()) != NAVGPU_OK ||
      navgpu_obstacle_update_costs(fleet_, 0, 1, box) != NAVGPU_OK ||
      navgpu_grid_download(fleet_, NAVGPU_GRID_MASTER, 0, 1, master.getCharMap()) != NAVGPU_OK ||
      navgpu_grid_download(fleet_, NAVGPU_GRID_OBSTACLE, 0, 1, costmap_) != NAVGPU_OK) {  // keep the layer's own grid in step
    ROS_ERROR_THROTTLE(1.0, "navgpu obstacle merge failed: %s", navgpu_last_error());
    current_ = false;
  }
}

}  // namespace navgpu
