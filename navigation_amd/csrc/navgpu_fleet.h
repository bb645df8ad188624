// Internal to libnavgpu.so: the fleet object behind the opaque navgpu_fleet handle and the helpers the host
// translation units share (navgpu_host.cpp: lifetime / costmap layers / DWA planner / measurement,
// navgpu_local_planner.cpp: DWAPlannerROS control cycle, navgpu_tp.cpp: legacy TrajectoryPlanner).
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <mutex>
#include <string>
#include <vector>

#include "navgpu_device.h"

using namespace navgpu;

namespace navgpu {
extern thread_local std::string g_last_error;  // defined in navgpu_host.cpp

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      g_last_error = std::string(#expr) + ": " + hipGetErrorString(e_);                            \
      return NAVGPU_ERR_HIP;                                                                       \
    }                                                                                              \
  } while (0)

// hipStreamSynchronize also flushes the runtime's batched launches; a hipStreamQuery polling loop does
// not (measured: the queue then only drains when the poll gives up), so no polling here.
inline hipError_t waitStream(hipStream_t s) { return hipStreamSynchronize(s); }

struct EventPair {
  int kernel;
  hipEvent_t a, b;
};
int ensureCompleteGrids(navgpu_fleet* f, uint32_t first, uint32_t count);
}  // namespace navgpu

struct navgpu_fleet {
  navgpu_fleet_desc desc{};
  // Every entry point that takes the fleet holds this for its whole body (FleetGuard): the calls on one fleet are
  // serialised INSIDE the library, so a reconfigure from another host thread (dynamic_reconfigure's spinner thread:
  // DWAPlanner::configuration_mutex_, dwa_planner.cpp:55,301; InflationLayer::inflation_access_,
  // inflation_layer.cpp:68,112,175) can never run beside a stage / update / cycle of the same fleet.  Recursive because
  // the control-cycle mirror (navgpu_local_planner_*) calls other entry points.
  std::recursive_mutex mu;
  hipStream_t stream = nullptr;
  CostmapDev cm{};
  PlannerDev pl{};
  bool planner_configured = false, inflation_configured = false, planner_staged = false;
  bool shift_pending = false;                   // rolling window: staged origins not yet applied to the grids
  uint32_t shift_first = 0, shift_count = 0;
  std::vector<void*> allocs;
  // host mirrors
  std::vector<double> h_origin;                 // [n][2]
  std::vector<double> h_fp_spec;                // [n][kMaxFootprint][2]
  std::vector<uint32_t> h_fp_n;                 // [n]
  navgpu_inflation_params infl{};
  navgpu_obstacle_params obsp{};
  double fp_radius = 0.0;                       // largest vertex distance over all instances
  // pinned host mirrors of the per-cycle staging arrays (full fleet size): H2D copies are truly
  // asynchronous and no per-call allocation happens on the staging path
  std::vector<void*> pinned;
  ObsCsr* hp_obs = nullptr;
  uint32_t *hp_cnt = nullptr, *hp_used = nullptr, *hp_plan_cnt = nullptr;
  float* hp_pts = nullptr;
  double *hp_fpw = nullptr, *hp_pose = nullptr, *hp_plan = nullptr, *hp_front = nullptr;
  int32_t *hp_shift = nullptr, *hp_align = nullptr;
  uint32_t *hp_reach = nullptr, *hp_fpn = nullptr;
  // the two staging blocks (device / pinned host, same layout): the pointers above and cm.pose ... / pl.state ... point into them
  uint8_t *cm_stage_dev = nullptr, *cm_stage_host = nullptr, *pl_stage_dev = nullptr, *pl_stage_host = nullptr;
  size_t cm_stage_bytes = 0, pl_stage_bytes = 0;
  // bounded MapGrid wavefronts (navgpu_planner_set_bounded_map_grids): which robots hold grids that are exact inside
  // their box only, that box, and whether the inputs of the cycle that made them are still the staged ones
  bool bounded_grids = true;
  std::vector<uint8_t> grid_partial;            // [n]
  std::vector<int32_t> h_box;                   // [n][4] x0, x1, y0, y1 of the last bounded cycle
  std::vector<uint64_t> inputs_gen, cycle_gen;  // [n] bumped by whatever a wavefront reads / value at the last cycle
  void touchInputs(uint32_t first, uint32_t count) {
    for (uint32_t i = first; i < first + count && i < inputs_gen.size(); ++i) ++inputs_gen[i];
  }
  navgpu_robot_state* hp_state = nullptr;
  navgpu_plan_result* hp_result = nullptr;      // [2][n]: the slot a cycle writes alternates while two cycles are in flight
  bool hp_dma_pending = false;  // a full stage's copies out of hp_state / hp_front / hp_align / hp_reach may still be queued
  // Two control cycles in flight on the stream (navgpu_planner_set_cycles_in_flight(f, 2)): cycle k + 1 is handed over and
  // queued while cycle k still runs, so the stream never idles between the end of a cycle and the host's next hand-over.
  // What has to be known for that is finer than "the stream has drained": that the copies out of the pinned mirrors have
  // run (ev_cm_h2d / ev_pl_h2d, recorded behind them) and that ONE cycle's results are in place (ev_cycle[slot], recorded
  // behind its k_select).
  int cycles_in_flight = 1;
  int res_slot = 0;                             // slot of hp_result the latest queued cycle writes
  hipEvent_t ev_cycle[2] = {nullptr, nullptr}, ev_cm_h2d = nullptr, ev_pl_h2d = nullptr;
  bool ev_cycle_set[2] = {false, false}, ev_cm_set = false, ev_pl_set = false;
  // the pinned mirrors are free again: the whole stream with one cycle in flight, the marker behind their copies with two
  hipError_t waitMirrors(hipEvent_t ev, bool set) {
    if (cycles_in_flight < 2) return navgpu::waitStream(stream);
    return set ? hipEventSynchronize(ev) : hipSuccess;
  }
  // DWAPlannerROS mirror (navgpu_local_planner_*): per-instance controller state, host only
  struct LocalPlannerState {
    std::vector<double> plan;      // stored global plan, (x, y, yaw) triples in the plan's frame (prunePlan shrinks it)
    double T[3] = {0, 0, 0};       // planar plan -> global transform
    bool has_T = false, have_plan = false;
    bool xy_tolerance_latch = false, rotating_to_goal = false;  // LatchedStopRotateController members
  };
  std::vector<LocalPlannerState> lp;
  navgpu_local_limits lp_limits{};
  bool lp_configured = false;
  // legacy TrajectoryPlanner (navgpu_tp_*)
  struct TpHost {
    std::vector<double> plan;  // global_plan_ as x, y pairs
    double final_goal_x = 0, final_goal_y = 0;
    bool final_goal_position_valid = false;
    navgpu_tp_state st{};
    std::vector<navgpu_tp_sample> made;  // the generateTrajectory calls of the last cycle, in call order
    int n_points = 0;                    // of the winner
  };
  TpDev tp{};
  bool tp_configured = false;
  std::vector<TpHost> tph;
  std::vector<double> tp_h_samples, tp_h_start;
  std::vector<TpOut> tp_h_out;
  std::vector<uint32_t> tp_h_nsamples, tp_h_within, tp_h_within_count;
  std::vector<int32_t> tp_h_winner;
  // scratch device buffers
  double* d_bounds_tmp = nullptr;               // [n][4]
  int32_t* d_boxes_tmp = nullptr;               // [n][4]
  float* d_explicit = nullptr;                  // [3]
  int8_t* d_occ = nullptr;
  float4* d_cell_costs = nullptr;                // [cells] navgpu_planner_cost_cloud
  size_t alloc_limit = 0;                       // fault injection (navgpu_fleet_set_alloc_limit); 0 = none
  // profiling
  bool profiling = false;
  uint32_t prof_mask = ~0u;                     // kernels bracketed by events while profiling (navgpu_profile_select)
  std::vector<EventPair> events;
  std::vector<EventPair> free_events;
  double prof_ms[NAVGPU_K_COUNT] = {0};
  uint64_t prof_n[NAVGPU_K_COUNT] = {0};

  template <class T>
  int alloc(T** p, size_t count) {
    void* q = nullptr;
    size_t bytes = std::max<size_t>(count * sizeof(T), 16);
    if (alloc_limit && bytes > alloc_limit) {  // navgpu_fleet_set_alloc_limit: tests make a large allocation fail on purpose
      g_last_error = "hipMalloc: refused by navgpu_fleet_set_alloc_limit";
      return NAVGPU_ERR_HIP;
    }
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess) {
      g_last_error = std::string("hipMalloc: ") + hipGetErrorString(e);
      return NAVGPU_ERR_HIP;
    }
    e = hipMemsetAsync(q, 0, bytes, stream);
    if (e != hipSuccess) {
      g_last_error = std::string("hipMemsetAsync: ") + hipGetErrorString(e);
      return NAVGPU_ERR_HIP;
    }
    allocs.push_back(q);
    *p = static_cast<T*>(q);
    return NAVGPU_OK;
  }
  template <class T>
  int allocPinned(T** p, size_t count) {
    void* q = nullptr;
    hipError_t e = hipHostMalloc(&q, std::max<size_t>(count * sizeof(T), 16), hipHostMallocDefault);
    if (e != hipSuccess) {
      g_last_error = std::string("hipHostMalloc: ") + hipGetErrorString(e);
      return NAVGPU_ERR_HIP;
    }
    memset(q, 0, std::max<size_t>(count * sizeof(T), 16));
    pinned.push_back(q);
    *p = static_cast<T*>(q);
    return NAVGPU_OK;
  }
  void release(void* q) {
    if (!q) return;
    auto it = std::find(allocs.begin(), allocs.end(), q);
    if (it != allocs.end()) allocs.erase(it);
    hipFree(q);
  }
  bool rangeOk(uint32_t first, uint32_t count) const { return count > 0 && first < desc.n_instances && count <= desc.n_instances - first; }

  int beginKernel(int k, EventPair* ep) {
    if (!profiling || !((prof_mask >> k) & 1u)) return NAVGPU_OK;
    if (free_events.empty()) {
      EventPair n{};
      HIP_TRY(hipEventCreate(&n.a));
      HIP_TRY(hipEventCreate(&n.b));
      free_events.push_back(n);
    }
    *ep = free_events.back();
    free_events.pop_back();
    ep->kernel = k;
    HIP_TRY(hipEventRecord(ep->a, stream));
    return NAVGPU_OK;
  }
  int endKernel(EventPair* ep) {
    if (!profiling || !ep->a) return NAVGPU_OK;  // (not one of the selected kernels)
    HIP_TRY(hipEventRecord(ep->b, stream));
    events.push_back(*ep);
    if (events.size() > 8192) return foldEvents();
    return NAVGPU_OK;
  }
  int foldEvents() {
    if (events.empty()) return NAVGPU_OK;
    HIP_TRY(waitStream(stream));
    for (auto& e : events) {
      float ms = 0;
      HIP_TRY(hipEventElapsedTime(&ms, e.a, e.b));
      prof_ms[e.kernel] += ms;
      prof_n[e.kernel] += 1;
      free_events.push_back(e);
    }
    events.clear();
    return NAVGPU_OK;
  }
};

// lock + "this thread talks to the fleet's GPU" (hipSetDevice is per host thread: a caller's second thread starts on
// device 0 whatever device the fleet lives on)
#ifndef NAVGPU_TEST_NO_FLEET_LOCK
struct FleetGuard {
  std::lock_guard<std::recursive_mutex> lk;
  explicit FleetGuard(navgpu_fleet* f) : lk(f->mu) { (void)hipSetDevice(f->desc.device); }
};
#else  // negative control of tests/test_fleet_threads_tsan.py only: ThreadSanitizer must object to this build
struct FleetGuard {
  explicit FleetGuard(navgpu_fleet* f) { (void)hipSetDevice(f->desc.device); }
};
#endif

#define PROFILED(fleet, kid, launch_expr)            \
  do {                                               \
    EventPair ep_{};                                 \
    int rc_ = (fleet)->beginKernel((kid), &ep_);     \
    if (rc_) return rc_;                             \
    launch_expr;                                     \
    rc_ = (fleet)->endKernel(&ep_);                  \
    if (rc_) return rc_;                             \
  } while (0)

inline int checkLaunch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_last_error = std::string("kernel launch: ") + hipGetErrorString(e);
    return NAVGPU_ERR_HIP;
  }
  return NAVGPU_OK;
}

