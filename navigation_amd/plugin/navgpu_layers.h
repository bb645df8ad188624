// costmap_2d::Layer adapters over the navgpu C-ABI (costmap_2d/include/costmap_2d/layer.h:50-130).
// Source-only here (ROS is absent from the build image: tests/test_plugin_syntax.py checks that these files parse
// against the reference's own headers); compiled in a catkin workspace, see INTEGRATION.md.
//
//  (A) per-layer drop-ins, one class per reference plugin (costmap_plugins.xml:1-17):
//        navgpu::ObstacleLayer   for costmap_2d::ObstacleLayer   (plugins/obstacle_layer.cpp)
//        navgpu::VoxelLayer      for costmap_2d::VoxelLayer      (plugins/voxel_layer.cpp)
//        navgpu::InflationLayer  for costmap_2d::InflationLayer  (plugins/inflation_layer.cpp)
//      Each keeps its layer grid resident on the GPU across cycles (rolling windows are shifted on the device) and
//      exchanges the master grid with the host once per updateCosts.
//  (B) navgpu::GpuLayers: ONE plugin in place of the obstacle|voxel + inflation pair.  The master grid crosses PCIe once
//      per cycle (up with what the layers before it wrote, down with marking, merging and inflation applied);
//      this is the LayeredCostmap::updateMap sequence bench.py measures, behind the Layer contract.
//          plugins: [{name: static_layer, type: "costmap_2d::StaticLayer"}, {name: gpu_layers, type: "navgpu::GpuLayers"}]
#ifndef NAVGPU_LAYERS_H_
#define NAVGPU_LAYERS_H_

#include <costmap_2d/InflationPluginConfig.h>
#include <costmap_2d/VoxelPluginConfig.h>
#include <costmap_2d/costmap_layer.h>
#include <costmap_2d/layer.h>
#include <costmap_2d/layered_costmap.h>
#include <costmap_2d/obstacle_layer.h>
#include <dynamic_reconfigure/server.h>

#include <boost/thread.hpp>

#include <navgpu.h>

namespace navgpu {

// One robot's fleet-of-1 and the calls every obstacle-type adapter makes with it.  Not a plugin.
class LayerBridge {
 public:
  LayerBridge() : fleet_(NULL), layers_(0), rolling_(false) {}
  ~LayerBridge();
  // (re)create the fleet for the master grid's geometry (Layer::matchSize)
  void create(const costmap_2d::Costmap2D& master, int layers, bool track_unknown, bool rolling);
  navgpu_fleet* fleet() const { return fleet_; }
  bool rolling() const { return rolling_; }
  // ObstacleLayer::updateBounds / VoxelLayer::updateBounds on the device: stage pose, footprint and observations,
  // shift a rolling window, clear, mark, grow bounds = {min_x, min_y, max_x, max_y}.  origin_xy: the layer's origin
  // after the call (moves with a rolling window).
  bool updateBounds(double rx, double ry, double ryaw, const std::vector<costmap_2d::Observation>& marking,
                    const std::vector<costmap_2d::Observation>& clearing, const std::vector<geometry_msgs::Point>& footprint_spec,
                    double* bounds, double* origin_xy);
  // updateCosts of the obstacle-type layer and / or the inflation layer on the master grid handed in
  bool updateCosts(costmap_2d::Costmap2D& master, int min_i, int min_j, int max_i, int max_j, bool merge, bool inflate,
                   unsigned char* layer_grid_out);

 private:
  navgpu_fleet* fleet_;
  int layers_;
  bool rolling_;
  std::vector<navgpu_observation> obs_;
  std::vector<float> pts_;
};

// InflationLayer::updateBounds's box bookkeeping (inflation_layer.cpp:125-158): four doubles of host state
struct InflationBounds {
  InflationBounds();
  void update(bool* need_reinflation, double inflation_radius, double* min_x, double* min_y, double* max_x, double* max_y);
  double last_min_x, last_min_y, last_max_x, last_max_y;
};

// ----------------------------------------------------------------------------- (A) per-layer drop-ins
class InflationLayer : public costmap_2d::Layer {
 public:
  InflationLayer();
  virtual ~InflationLayer();
  virtual void onInitialize();
  virtual void updateBounds(double robot_x, double robot_y, double robot_yaw, double* min_x, double* min_y, double* max_x,
                            double* max_y);
  virtual void updateCosts(costmap_2d::Costmap2D& master_grid, int min_i, int min_j, int max_i, int max_j);
  virtual void matchSize();
  virtual bool isDiscretized() { return true; }
  virtual void reset() { onInitialize(); }

 protected:
  virtual void onFootprintChanged();

 private:
  void reconfigureCB(costmap_2d::InflationPluginConfig& config, uint32_t level);
  void pushParams();
  LayerBridge gpu_;
  // InflationLayer::inflation_access_ (inflation_layer.h:158): reconfigureCB / matchSize / updateBounds / updateCosts
  // exclude one another (inflation_layer.cpp:68,112,175,366) - here that also keeps matchSize from destroying the fleet
  // under a running updateCosts
  boost::recursive_mutex inflation_access_;
  navgpu_inflation_params p_;
  bool need_reinflation_;
  InflationBounds box_;
  dynamic_reconfigure::Server<costmap_2d::InflationPluginConfig>* dsrv_;
};

// Reuses the reference class for topics / observation buffers / parameters (ROS I/O, out of scope) and replaces the
// hot virtuals.
class ObstacleLayer : public costmap_2d::ObstacleLayer {
 public:
  ObstacleLayer() {}
  virtual ~ObstacleLayer() {}
  virtual void onInitialize();
  virtual void matchSize();
  virtual void reset();
  virtual void resetMap(unsigned int x0, unsigned int y0, unsigned int xn, unsigned int yn);  // CostmapLayer::resetBoundingBox's clear
  virtual void updateBounds(double robot_x, double robot_y, double robot_yaw, double* min_x, double* min_y, double* max_x,
                            double* max_y);
  virtual void updateCosts(costmap_2d::Costmap2D& master_grid, int min_i, int min_j, int max_i, int max_j);

 protected:
  virtual int gpuLayers() const { return NAVGPU_LAYER_OBSTACLE; }
  virtual void pushObstacleParams();
  bool gpuUpdateBounds(double rx, double ry, double ryaw, double* min_x, double* min_y, double* max_x, double* max_y);
  LayerBridge gpu_;
  // The reference's obstacle / voxel layers take no lock in their reconfigure callbacks (a VoxelLayer reconfigure runs
  // matchSize from the spinner thread, voxel_layer.cpp:77-91).  Here matchSize destroys and recreates the fleet handle,
  // so every method that touches gpu_ holds this; GpuLayers uses it as its inflation_access_ as well.
  boost::recursive_mutex gpu_access_;
};

// costmap_2d::VoxelLayer's members are private, so this adapter derives from ObstacleLayer like VoxelLayer itself does
// and re-reads VoxelPluginConfig with its own dynamic_reconfigure server (voxel_layer.cpp:63-91).  The voxel_grid /
// clearing_endpoints debug topics of the reference are not published (visualisation, out of scope).
class VoxelLayer : public ObstacleLayer {
 public:
  VoxelLayer() : voxel_dsrv_(NULL), z_voxels_(10), unknown_threshold_(15), mark_threshold_(0), origin_z_(0.0), z_resolution_(0.2) {}
  virtual ~VoxelLayer();
  virtual bool isDiscretized() { return true; }

 protected:
  virtual void setupDynamicReconfigure(ros::NodeHandle& nh);
  virtual int gpuLayers() const { return NAVGPU_LAYER_VOXEL; }
  virtual void pushObstacleParams();

 private:
  void reconfigureCB(costmap_2d::VoxelPluginConfig& config, uint32_t level);
  dynamic_reconfigure::Server<costmap_2d::VoxelPluginConfig>* voxel_dsrv_;
  int z_voxels_, unknown_threshold_, mark_threshold_;
  double origin_z_, z_resolution_;
};

// ----------------------------------------------------------------------------- (B) whole update in one plugin
// Parameters: those of the obstacle layer (ObstaclePlugin.cfg, observation sources) in this plugin's namespace, plus
// `voxel` (bool, default false: z_voxels / origin_z / z_resolution / unknown_threshold / mark_threshold then apply) and
// the inflation layer's `inflation_radius`, `cost_scaling_factor` (InflationPlugin.cfg:8-9).
class GpuLayers : public ObstacleLayer {
 public:
  GpuLayers();
  virtual ~GpuLayers() {}
  virtual void onInitialize();
  virtual void matchSize();
  virtual void updateBounds(double robot_x, double robot_y, double robot_yaw, double* min_x, double* min_y, double* max_x,
                            double* max_y);
  virtual void updateCosts(costmap_2d::Costmap2D& master_grid, int min_i, int min_j, int max_i, int max_j);

 protected:
  virtual void onFootprintChanged();
  virtual int gpuLayers() const { return (voxel_ ? NAVGPU_LAYER_VOXEL : NAVGPU_LAYER_OBSTACLE) | NAVGPU_LAYER_INFLATION; }
  virtual void pushObstacleParams();

 private:
  bool voxel_;
  navgpu_obstacle_params vp_;
  navgpu_inflation_params ip_;
  bool need_reinflation_;
  InflationBounds box_;
};

}  // namespace navgpu
#endif
