// See navgpu_layers.h.  Every block cites the reference lines it stands in for.
#include "navgpu_layers.h"

#include <costmap_2d/footprint.h>
#include <pluginlib/class_list_macros.h>

PLUGINLIB_EXPORT_CLASS(navgpu::InflationLayer, costmap_2d::Layer)  // inflation_layer.cpp:45
PLUGINLIB_EXPORT_CLASS(navgpu::ObstacleLayer, costmap_2d::Layer)   // obstacle_layer.cpp:43
PLUGINLIB_EXPORT_CLASS(navgpu::VoxelLayer, costmap_2d::Layer)      // voxel_layer.cpp:43
PLUGINLIB_EXPORT_CLASS(navgpu::GpuLayers, costmap_2d::Layer)

namespace navgpu {

// ----------------------------------------------------------------------------- LayerBridge
LayerBridge::~LayerBridge() {
  if (fleet_) navgpu_fleet_destroy(fleet_);
}

void LayerBridge::create(const costmap_2d::Costmap2D& m, int layers, bool track_unknown, bool rolling) {
  if (fleet_) navgpu_fleet_destroy(fleet_);
  fleet_ = NULL;
  navgpu_fleet_desc d = {};
  d.n_instances = 1;
  d.size_x = m.getSizeInCellsX();
  d.size_y = m.getSizeInCellsY();
  d.resolution = m.getResolution();
  d.layers = layers;
  d.track_unknown = track_unknown;
  d.max_points = 1 << 16;
  d.max_observations = 16;
  d.max_footprint = 32;
  d.rolling_window = rolling;  // the device then applies Costmap2D::updateOrigin to its resident grids itself
  if (navgpu_fleet_create(&d, &fleet_) != NAVGPU_OK) throw std::runtime_error(std::string("navgpu: ") + navgpu_last_error());
  layers_ = layers;
  rolling_ = rolling;
  double origin[2] = {m.getOriginX(), m.getOriginY()};
  navgpu_fleet_set_origin(fleet_, 0, 1, origin);
}

bool LayerBridge::updateBounds(double rx, double ry, double ryaw, const std::vector<costmap_2d::Observation>& marking,
                               const std::vector<costmap_2d::Observation>& clearing,
                               const std::vector<geometry_msgs::Point>& footprint_spec, double* b, double* origin_xy) {
  obs_.clear();
  pts_.clear();
  for (int pass = 0; pass < 2; ++pass) {  // clearing observations first, then marking (obstacle_layer.cpp:361-410)
    const std::vector<costmap_2d::Observation>& v = pass ? marking : clearing;
    for (size_t k = 0; k < v.size(); ++k) {
      navgpu_observation o = {};
      o.instance = 0;
      o.first_point = pts_.size() / 3;
      o.n_points = v[k].cloud_->points.size();
      o.flags = pass ? NAVGPU_OBS_MARKING : NAVGPU_OBS_CLEARING;
      o.origin_x = v[k].origin_.x;
      o.origin_y = v[k].origin_.y;
      o.origin_z = v[k].origin_.z;
      o.obstacle_range = v[k].obstacle_range_;
      o.raytrace_range = v[k].raytrace_range_;
      for (size_t i = 0; i < v[k].cloud_->points.size(); ++i) {
        pts_.push_back(v[k].cloud_->points[i].x);
        pts_.push_back(v[k].cloud_->points[i].y);
        pts_.push_back(v[k].cloud_->points[i].z);
      }
      obs_.push_back(o);
    }
  }
  std::vector<double> fxy;
  for (size_t i = 0; i < footprint_spec.size(); ++i) {
    fxy.push_back(footprint_spec[i].x);
    fxy.push_back(footprint_spec[i].y);
  }
  const double pose[3] = {rx, ry, ryaw};
  // navgpu_costmap_stage evaluates the rolling window's new origin exactly as ObstacleLayer::updateBounds :344-345 /
  // Costmap2D::updateOrigin :264-276 do; navgpu_obstacle_update_bounds shifts the resident grids before it clears and marks
  if (navgpu_set_footprint(fleet_, 0, 1, fxy.empty() ? NULL : &fxy[0], footprint_spec.size()) != NAVGPU_OK ||
      navgpu_costmap_stage(fleet_, 0, 1, pose, obs_.empty() ? NULL : &obs_[0], obs_.size(), pts_.empty() ? NULL : &pts_[0],
                           pts_.size() / 3) != NAVGPU_OK ||
      navgpu_obstacle_update_bounds(fleet_, 0, 1, b) != NAVGPU_OK)
    return false;
  return navgpu_fleet_get_origin(fleet_, 0, 1, origin_xy) == NAVGPU_OK;
}

bool LayerBridge::updateCosts(costmap_2d::Costmap2D& master, int min_i, int min_j, int max_i, int max_j, bool merge, bool inflate,
                              unsigned char* layer_grid_out) {
  int32_t box[4] = {min_i, min_j, max_i, max_j};
  // the master grid as the layers before this one left it goes up, the result comes back: one round trip
  if (navgpu_grid_upload(fleet_, NAVGPU_GRID_MASTER, 0, 1, master.getCharMap()) != NAVGPU_OK) return false;
  if (merge && navgpu_obstacle_update_costs(fleet_, 0, 1, box) != NAVGPU_OK) return false;
  if (inflate && navgpu_inflate(fleet_, 0, 1, box) != NAVGPU_OK) return false;
  if (navgpu_grid_download(fleet_, NAVGPU_GRID_MASTER, 0, 1, master.getCharMap()) != NAVGPU_OK) return false;
  if (layer_grid_out && navgpu_grid_download(fleet_, NAVGPU_GRID_OBSTACLE, 0, 1, layer_grid_out) != NAVGPU_OK) return false;
  return true;
}

// ----------------------------------------------------------------------------- InflationBounds
InflationBounds::InflationBounds() : last_min_x(-std::numeric_limits<float>::max()), last_min_y(-std::numeric_limits<float>::max()),
                                     last_max_x(std::numeric_limits<float>::max()), last_max_y(std::numeric_limits<float>::max()) {}
void InflationBounds::update(bool* need_reinflation, double inflation_radius, double* min_x, double* min_y, double* max_x, double* max_y) {
  // inflation_layer.cpp:125-158
  if (*need_reinflation) {
    last_min_x = *min_x;
    last_min_y = *min_y;
    last_max_x = *max_x;
    last_max_y = *max_y;
    // "For some reason when I make these -<double>::max() it does not work with Costmap2D::worldToMapEnforceBounds()"
    *min_x = -std::numeric_limits<float>::max();
    *min_y = -std::numeric_limits<float>::max();
    *max_x = std::numeric_limits<float>::max();
    *max_y = std::numeric_limits<float>::max();
    *need_reinflation = false;
  } else {
    const double tx0 = last_min_x, ty0 = last_min_y, tx1 = last_max_x, ty1 = last_max_y;
    last_min_x = *min_x;
    last_min_y = *min_y;
    last_max_x = *max_x;
    last_max_y = *max_y;
    *min_x = std::min(tx0, *min_x) - inflation_radius;
    *min_y = std::min(ty0, *min_y) - inflation_radius;
    *max_x = std::max(tx1, *max_x) + inflation_radius;
    *max_y = std::max(ty1, *max_y) + inflation_radius;
  }
}

// ----------------------------------------------------------------------------- InflationLayer
InflationLayer::InflationLayer() : need_reinflation_(false), dsrv_(NULL) {
  p_.enabled = 1;
  p_.priority_queue_order = 0;
  p_.inflation_radius = 0;
  p_.cost_scaling_factor = 0;
  p_.inscribed_radius = 0;
}
InflationLayer::~InflationLayer() { delete dsrv_; }
void InflationLayer::onInitialize() {  // inflation_layer.cpp:70-99
  ros::NodeHandle nh("~/" + name_);
  current_ = true;
  need_reinflation_ = false;
  bool pq = false;  // byte-for-byte the reference's priority-queue walk instead of the exact transform (slow)
  nh.param("navgpu_priority_queue_order", pq, false);
  p_.priority_queue_order = pq;
  if (!dsrv_) dsrv_ = new dynamic_reconfigure::Server<costmap_2d::InflationPluginConfig>(nh);
  dsrv_->setCallback(boost::bind(&InflationLayer::reconfigureCB, this, _1, _2));
  matchSize();
}
void InflationLayer::reconfigureCB(costmap_2d::InflationPluginConfig& c, uint32_t) {  // :101-109, :362-376
  boost::unique_lock<boost::recursive_mutex> lock(inflation_access_);  // setInflationParameters :366
  if (p_.cost_scaling_factor != c.cost_scaling_factor || p_.inflation_radius != c.inflation_radius) need_reinflation_ = true;
  if (enabled_ != c.enabled) need_reinflation_ = true;
  enabled_ = c.enabled;
  p_.enabled = c.enabled;
  p_.inflation_radius = c.inflation_radius;
  p_.cost_scaling_factor = c.cost_scaling_factor;
  pushParams();
}
void InflationLayer::pushParams() {
  if (gpu_.fleet() && navgpu_inflation_configure(gpu_.fleet(), &p_) != NAVGPU_OK)
    ROS_ERROR("navgpu_inflation_configure: %s", navgpu_last_error());
}
void InflationLayer::matchSize() {  // :110-123 — the fleet is tied to the master's geometry
  boost::unique_lock<boost::recursive_mutex> lock(inflation_access_);  // :112
  gpu_.create(*layered_costmap_->getCostmap(), NAVGPU_LAYER_INFLATION, layered_costmap_->isTrackingUnknown(), false);
  pushParams();
}
void InflationLayer::onFootprintChanged() {  // :160-170
  boost::unique_lock<boost::recursive_mutex> lock(inflation_access_);
  p_.inscribed_radius = layered_costmap_->getInscribedRadius();
  need_reinflation_ = true;
  pushParams();
}
void InflationLayer::updateBounds(double, double, double, double* min_x, double* min_y, double* max_x, double* max_y) {
  boost::unique_lock<boost::recursive_mutex> lock(inflation_access_);  // (need_reinflation_ and the radius are the reconfigure thread's too)
  box_.update(&need_reinflation_, p_.inflation_radius, min_x, min_y, max_x, max_y);
}
void InflationLayer::updateCosts(costmap_2d::Costmap2D& master, int min_i, int min_j, int max_i, int max_j) {
  boost::unique_lock<boost::recursive_mutex> lock(inflation_access_);  // :175
  if (!enabled_) return;  // :172-266 on the GPU: upload, inflate the box, download
  if (!gpu_.updateCosts(master, min_i, min_j, max_i, max_j, false, true, NULL)) {
    ROS_ERROR_THROTTLE(1.0, "navgpu inflation failed: %s", navgpu_last_error());
    current_ = false;  // the reference's failure channel: isCurrent() false stops the robot
  }
}

// ----------------------------------------------------------------------------- ObstacleLayer
void ObstacleLayer::onInitialize() {
  costmap_2d::ObstacleLayer::onInitialize();  // topics, buffers, parameters: unchanged reference code
  matchSize();  // (the reference calls its own matchSize by qualified name, :118, so the virtual one has not run yet)
}
void ObstacleLayer::pushObstacleParams() {
  navgpu_obstacle_params p = {};
  p.enabled = enabled_;
  p.footprint_clearing_enabled = footprint_clearing_enabled_;
  p.combination_method = combination_method_;
  p.max_obstacle_height = max_obstacle_height_;
  if (gpu_.fleet() && navgpu_obstacle_configure(gpu_.fleet(), &p) != NAVGPU_OK)
    ROS_ERROR("navgpu_obstacle_configure: %s", navgpu_last_error());
}
void ObstacleLayer::matchSize() {  // obstacle_layer.cpp via CostmapLayer::matchSize (costmap_layer.cpp:17-22)
  boost::unique_lock<boost::recursive_mutex> lock(gpu_access_);
  costmap_2d::ObstacleLayer::matchSize();
  gpu_.create(*layered_costmap_->getCostmap(), gpuLayers(), default_value_ == costmap_2d::NO_INFORMATION, rolling_window_);
  pushObstacleParams();
  if (gpuLayers() & NAVGPU_LAYER_INFLATION) onFootprintChanged();
}
void ObstacleLayer::reset() {  // obstacle_layer.cpp:589-596: deactivate, resetMaps, current_ = true, activate
  boost::unique_lock<boost::recursive_mutex> lock(gpu_access_);
  costmap_2d::ObstacleLayer::reset();
  if (gpu_.fleet()) navgpu_grid_reset(gpu_.fleet(), NAVGPU_GRID_OBSTACLE, 0, 1);  // + the voxel columns of a voxel fleet
}
void ObstacleLayer::resetMap(unsigned int x0, unsigned int y0, unsigned int xn, unsigned int yn) {
  boost::unique_lock<boost::recursive_mutex> lock(gpu_access_);
  // Costmap2D::resetMap is what CostmapLayer::resetBoundingBox (costmap_layer.cpp:30-43; Costmap2DROS::resetBoundingBox's
  // per-layer call) clears the layer with: the device-resident layer grid follows; the extra bounds it adds stay on the
  // host and enter gpuUpdateBounds through useExtraBounds
  costmap_2d::ObstacleLayer::resetMap(x0, y0, xn, yn);
  if (gpu_.fleet() && navgpu_grid_reset_window(gpu_.fleet(), NAVGPU_GRID_OBSTACLE, 0, 1, x0, y0, xn, yn) != NAVGPU_OK)
    ROS_ERROR("navgpu_grid_reset_window: %s", navgpu_last_error());
}
bool ObstacleLayer::gpuUpdateBounds(double rx, double ry, double ryaw, double* min_x, double* min_y, double* max_x, double* max_y) {
  boost::unique_lock<boost::recursive_mutex> lock(gpu_access_);
  // obstacle_layer.cpp:340-413.  The reconfigurable parameters are re-pushed every cycle (ObstaclePluginConfig's
  // callback is private to the reference class; four scalars)
  pushObstacleParams();
  if (!enabled_) {
    // the reference rolls the window BEFORE it looks at enabled_ (obstacle_layer.cpp:343-346): a disabled layer's origin
    // and grid keep following the robot.  With `enabled` pushed as 0 the device call below shifts the resident grids,
    // leaves cells and bounds alone, and hands back the new origin.
    if (rolling_window_) {
      const std::vector<costmap_2d::Observation> none;
      double b[4] = {*min_x, *min_y, *max_x, *max_y}, origin[2];
      if (!gpu_.updateBounds(rx, ry, ryaw, none, none, getFootprint(), b, origin)) {
        ROS_ERROR_THROTTLE(1.0, "navgpu obstacle window roll failed: %s", navgpu_last_error());
        return false;
      }
      origin_x_ = origin[0];
      origin_y_ = origin[1];
    }
    return true;
  }
  useExtraBounds(min_x, min_y, max_x, max_y);
  std::vector<costmap_2d::Observation> marking, clearing;  // :349-359
  bool current = getMarkingObservations(marking);
  current = getClearingObservations(clearing) && current;
  current_ = current;
  double b[4] = {*min_x, *min_y, *max_x, *max_y}, origin[2];
  if (!gpu_.updateBounds(rx, ry, ryaw, marking, clearing, getFootprint(), b, origin)) {
    ROS_ERROR_THROTTLE(1.0, "navgpu obstacle update failed: %s", navgpu_last_error());
    current_ = false;
    return false;
  }
  // rolling window: the layer's own Costmap2D follows the origin the device moved to (its bytes are refreshed from
  // the device in updateCosts); the reference would memcpy-shift the host grid here (costmap_2d.cpp:264-313)
  origin_x_ = origin[0];
  origin_y_ = origin[1];
  *min_x = b[0];
  *min_y = b[1];
  *max_x = b[2];
  *max_y = b[3];
  costmap_2d::transformFootprint(rx, ry, ryaw, getFootprint(), transformed_footprint_);  // updateFootprint :415-425
  return true;
}
void ObstacleLayer::updateBounds(double rx, double ry, double ryaw, double* min_x, double* min_y, double* max_x, double* max_y) {
  gpuUpdateBounds(rx, ry, ryaw, min_x, min_y, max_x, max_y);
}
void ObstacleLayer::updateCosts(costmap_2d::Costmap2D& master, int min_i, int min_j, int max_i, int max_j) {
  boost::unique_lock<boost::recursive_mutex> lock(gpu_access_);
  if (!enabled_) return;  // obstacle_layer.cpp:427-448 on the GPU
  if (!gpu_.updateCosts(master, min_i, min_j, max_i, max_j, true, false, costmap_)) {  // costmap_: keep the layer's own grid in step
    ROS_ERROR_THROTTLE(1.0, "navgpu obstacle merge failed: %s", navgpu_last_error());
    current_ = false;
  }
}

// ----------------------------------------------------------------------------- VoxelLayer
VoxelLayer::~VoxelLayer() { delete voxel_dsrv_; }
void VoxelLayer::setupDynamicReconfigure(ros::NodeHandle& nh) {  // voxel_layer.cpp:63-69
  voxel_dsrv_ = new dynamic_reconfigure::Server<costmap_2d::VoxelPluginConfig>(nh);
  voxel_dsrv_->setCallback(boost::bind(&VoxelLayer::reconfigureCB, this, _1, _2));
}
void VoxelLayer::reconfigureCB(costmap_2d::VoxelPluginConfig& config, uint32_t) {  // voxel_layer.cpp:77-91
  boost::unique_lock<boost::recursive_mutex> lock(gpu_access_);
  enabled_ = config.enabled;
  footprint_clearing_enabled_ = config.footprint_clearing_enabled;
  max_obstacle_height_ = config.max_obstacle_height;
  z_voxels_ = config.z_voxels;
  origin_z_ = config.origin_z;
  z_resolution_ = config.z_resolution;
  unknown_threshold_ = config.unknown_threshold;  // the + (VOXEL_BITS - size_z) of :87 is applied inside the library
  mark_threshold_ = config.mark_threshold;
  combination_method_ = config.combination_method;
  matchSize();
}
void VoxelLayer::pushObstacleParams() {
  navgpu_obstacle_params p = {};
  p.enabled = enabled_;
  p.footprint_clearing_enabled = footprint_clearing_enabled_;
  p.combination_method = combination_method_;
  p.max_obstacle_height = max_obstacle_height_;
  p.z_voxels = z_voxels_;
  p.origin_z = origin_z_;
  p.z_resolution = z_resolution_;
  p.unknown_threshold = unknown_threshold_;
  p.mark_threshold = mark_threshold_;
  if (gpu_.fleet() && navgpu_obstacle_configure(gpu_.fleet(), &p) != NAVGPU_OK)
    ROS_ERROR("navgpu_obstacle_configure (voxel): %s", navgpu_last_error());
}

// ----------------------------------------------------------------------------- GpuLayers
GpuLayers::GpuLayers() : voxel_(false), need_reinflation_(false) {
  memset(&vp_, 0, sizeof(vp_));
  memset(&ip_, 0, sizeof(ip_));
  ip_.enabled = 1;
}
void GpuLayers::onInitialize() {
  ros::NodeHandle nh("~/" + name_);
  nh.param("voxel", voxel_, false);
  int z_voxels = 10, unknown_threshold = 15, mark_threshold = 0;
  nh.param("z_voxels", z_voxels, 10);  // VoxelPlugin.cfg:10-15
  nh.param("origin_z", vp_.origin_z, 0.0);
  nh.param("z_resolution", vp_.z_resolution, 0.2);
  nh.param("unknown_threshold", unknown_threshold, 15);
  nh.param("mark_threshold", mark_threshold, 0);
  vp_.z_voxels = z_voxels;
  vp_.unknown_threshold = unknown_threshold;
  vp_.mark_threshold = mark_threshold;
  nh.param("inflation_radius", ip_.inflation_radius, 0.55);  // InflationPlugin.cfg:8-9
  nh.param("cost_scaling_factor", ip_.cost_scaling_factor, 10.0);
  bool pq = false;
  nh.param("navgpu_priority_queue_order", pq, false);
  ip_.priority_queue_order = pq;
  need_reinflation_ = true;
  ObstacleLayer::onInitialize();
}
void GpuLayers::pushObstacleParams() {
  vp_.enabled = enabled_;
  vp_.footprint_clearing_enabled = footprint_clearing_enabled_;
  vp_.combination_method = combination_method_;
  vp_.max_obstacle_height = max_obstacle_height_;
  if (gpu_.fleet() && navgpu_obstacle_configure(gpu_.fleet(), &vp_) != NAVGPU_OK)
    ROS_ERROR("navgpu_obstacle_configure: %s", navgpu_last_error());
}
void GpuLayers::matchSize() {
  ObstacleLayer::matchSize();  // creates the fleet with gpuLayers() and calls onFootprintChanged
}
void GpuLayers::onFootprintChanged() {  // inflation_layer.cpp:160-170
  boost::unique_lock<boost::recursive_mutex> lock(gpu_access_);
  ip_.inscribed_radius = layered_costmap_->getInscribedRadius();
  need_reinflation_ = true;
  if (gpu_.fleet() && navgpu_inflation_configure(gpu_.fleet(), &ip_) != NAVGPU_OK)
    ROS_ERROR("navgpu_inflation_configure: %s", navgpu_last_error());
}
void GpuLayers::updateBounds(double rx, double ry, double ryaw, double* min_x, double* min_y, double* max_x, double* max_y) {
  boost::unique_lock<boost::recursive_mutex> lock(gpu_access_);
  // the obstacle layer's updateBounds followed by the inflation layer's, as LayeredCostmap::updateMap :96-115 runs them
  gpuUpdateBounds(rx, ry, ryaw, min_x, min_y, max_x, max_y);
  box_.update(&need_reinflation_, ip_.inflation_radius, min_x, min_y, max_x, max_y);
}
void GpuLayers::updateCosts(costmap_2d::Costmap2D& master, int min_i, int min_j, int max_i, int max_j) {
  boost::unique_lock<boost::recursive_mutex> lock(gpu_access_);
  // ObstacleLayer::updateCosts then InflationLayer::updateCosts (layered_costmap.cpp:138-142) on the resident grids
  if (!gpu_.updateCosts(master, min_i, min_j, max_i, max_j, enabled_, true, costmap_)) {
    ROS_ERROR_THROTTLE(1.0, "navgpu layered update failed: %s", navgpu_last_error());
    current_ = false;
  }
}

}  // namespace navgpu
