// navgpu::InflationLayer / navgpu::ObstacleLayer — costmap_2d::Layer adapters over the navgpu
// C-ABI (costmap_2d/include/costmap_2d/layer.h:50-130).  Source-only here (ROS absent from the
// build image); compiled in a catkin workspace, see INTEGRATION.md.
//
// Two integration modes:
//  (A) per-layer drop-in (this file): each layer keeps the reference's updateBounds arithmetic on
//      the host (a handful of doubles) and runs updateCosts on the GPU, moving only the update
//      window's rows of the master grid across PCIe.
//  (B) whole-costmap residency: a single navgpu::GpuLayers plugin replaces the obstacle + inflation
//      pair and calls navgpu_costmap_stage / navgpu_costmap_update once per cycle, downloading the
//      master grid at the end of updateCosts; this is the path bench.py measures.
#ifndef NAVGPU_LAYERS_H_
#define NAVGPU_LAYERS_H_

#include <costmap_2d/InflationPluginConfig.h>
#include <costmap_2d/costmap_layer.h>
#include <costmap_2d/layer.h>
#include <costmap_2d/layered_costmap.h>
#include <costmap_2d/obstacle_layer.h>
#include <dynamic_reconfigure/server.h>

#include <navgpu.h>

namespace navgpu {

// Drop-in for costmap_2d::InflationLayer (plugins/inflation_layer.cpp)
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
  navgpu_fleet* fleet_;
  navgpu_inflation_params p_;
  bool need_reinflation_;
  double last_min_x_, last_min_y_, last_max_x_, last_max_y_;
  dynamic_reconfigure::Server<costmap_2d::InflationPluginConfig>* dsrv_;
};

// Drop-in for costmap_2d::ObstacleLayer: reuses the reference class for topics / observation
// buffers (ROS I/O, out of scope) and replaces the two hot virtuals.
class ObstacleLayer : public costmap_2d::ObstacleLayer {
 public:
  ObstacleLayer() : fleet_(NULL) {}
  virtual ~ObstacleLayer();
  virtual void onInitialize();
  virtual void matchSize();
  virtual void updateBounds(double robot_x, double robot_y, double robot_yaw, double* min_x, double* min_y, double* max_x,
                            double* max_y);
  virtual void updateCosts(costmap_2d::Costmap2D& master_grid, int min_i, int min_j, int max_i, int max_j);

 private:
  navgpu_fleet* fleet_;
};

}  // namespace navgpu
#endif
