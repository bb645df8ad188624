// TEST INFRASTRUCTURE ONLY (tests/test_fleet_threads_tsan.py): a host-memory stand-in for the handful of HIP runtime
// calls and kernel launchers that libnavgpu's HOST translation units reference, so that those units - the fleet
// bookkeeping, staging mirrors, configure / stage / cycle sequencing and their locking - can be built with
// g++ -fsanitize=thread and hammered from several threads on the CPU box.  Nothing here computes anything: "device"
// memory is calloc'ed host memory, copies are memcpy, streams are synchronous, kernels are no-ops (k_select's result
// store is imitated so that navgpu_planner_results has something to read).  It is never linked into libnavgpu.so.
#include <cstdlib>
#include <cstring>

#include "../../navigation_amd/csrc/navgpu_device.h"

extern "C" {
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned int) { *p = calloc(1, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < h; ++r) memcpy((char*)d + r * dp, (const char*)s + r * sp, w);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned int) { *s = (hipStream_t)calloc(1, 8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)calloc(1, 8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)calloc(1, 8); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.001f; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stub"; }
}

namespace navgpu {
void launch_tp_within(const PlannerDev&, const TpDev&, uint32_t, uint32_t, hipStream_t) {}
void launch_tp_rollout(const PlannerDev&, const TpDev&, uint32_t, uint32_t, int, hipStream_t) {}
void launch_obstacle(const CostmapDev&, uint32_t, uint32_t, const double*, int, hipStream_t) {}
void launch_reset_window(uint8_t*, size_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, uint8_t, hipStream_t) {}
void launch_reset_bounding_box(const CostmapDev&, uint32_t, uint32_t, const double*, hipStream_t) {}
void launch_merge(const CostmapDev&, uint32_t, uint32_t, const int32_t*, hipStream_t, bool) {}
void launch_inflate(const CostmapDev&, uint32_t, uint32_t, const int32_t*, hipStream_t) {}
void launch_static_interpret(uint8_t*, const int8_t*, uint32_t, uint32_t, uint32_t, int, int, int, int, hipStream_t) {}
void launch_export_window(const uint8_t*, uint32_t, uint32_t, uint32_t, uint32_t, uint32_t, int8_t*, hipStream_t) {}
void launch_fill_u8(uint8_t* d, uint8_t v, size_t n, hipStream_t) { memset(d, v, n); }
void launch_fill_u32(uint32_t* d, uint32_t v, size_t n, hipStream_t) { for (size_t i = 0; i < n; ++i) d[i] = v; }
void launch_shift_u8(const uint8_t*, uint8_t*, const CostmapDev&, uint32_t, uint32_t, uint8_t, hipStream_t) {}
void launch_shift_u32(const uint32_t*, uint32_t*, const CostmapDev&, uint32_t, uint32_t, uint32_t, hipStream_t) {}
void launch_cell_costs(const PlannerDev&, uint32_t, float4*, hipStream_t) {}
void launch_samples(const PlannerDev& pl, uint32_t first, uint32_t count, hipStream_t) {
  for (uint32_t i = first; i < first + count; ++i) pl.axis_samples[(size_t)i * 3 * pl.max_axis] = 1.0f;  // touches the table a reconfigure re-allocates
}
void launch_bfs(const PlannerDev&, uint32_t, uint32_t, hipStream_t, const uint32_t*, bool) {}
uint32_t launch_score(const PlannerDev& pl, uint32_t first, uint32_t count, const float*, hipStream_t) {
  for (uint32_t i = first; i < first + count; ++i) pl.part_cost[(size_t)i * pl.score_blocks + pl.score_blocks - 1] = 1.0;
  return 1;
}
void launch_select(const PlannerDev& pl, uint32_t first, uint32_t count, uint32_t, hipStream_t) {
  for (uint32_t i = first; i < first + count; ++i) {
    navgpu_plan_result r;
    memset(&r, 0, sizeof(r));
    r.best_index = (int32_t)pl.max_samples;  // lets the harness see WHICH configuration a cycle ran under
    r.n_scored = pl.cfg.vx_samples;
    pl.result[i] = r;
  }
}
void launch_stage_poses(const PoseChunk& c, hipStream_t) {
  memcpy(c.state + c.first, c.st, sizeof(navgpu_robot_state) * c.count);
  memcpy(c.bfs_reach + c.first, c.reach, sizeof(uint32_t) * c.count);
}
void launch_sincos(const double*, uint32_t, double*, double*, hipStream_t) {}
size_t score_table_bytes(const PlannerDev&) { return 1024; }
size_t score_window_bytes(uint32_t win) { return (size_t)win * win; }
size_t score_prep_bytes(const PlannerDev& pl) { return (size_t)pl.win * pl.win + 4096; }
size_t score_prep_slot_bytes(const PlannerDev& pl) { return (size_t)pl.win * pl.win + 4096 + 64 + 256 * 256; }
size_t bfs_scratch_words(uint32_t, uint32_t) { return 0; }
uint32_t score_table_rows(const PlannerDev&, uint32_t) { return 4; }
void launch_navfn_costmap(const NavfnDev&, uint32_t, uint32_t, const uint8_t*, size_t, int, int, hipStream_t) {}
void launch_gp_plan(const NavfnDev&, uint32_t, uint32_t, const navgpu_global_planner_params&, const double*, const double*, const int32_t*, void*, hipStream_t) {}
void launch_navfn_plan(const NavfnDev&, uint32_t, uint32_t, const int32_t*, const int32_t*, int, int, hipStream_t) {}
void launch_navfn_wf_init(const NavfnDev&, uint32_t, uint32_t, const NavfnWfRule&, const int32_t*, const float*, hipStream_t) {}
void launch_navfn_wf_round(const NavfnDev& nv, uint32_t first, uint32_t count, const NavfnWfRule&, const int32_t*, int, int, hipStream_t) {
  for (uint32_t i = first; i < first + count; ++i) nv.wf_status[i].done = 1;  // a search that settles at once
}
void launch_navfn_wf_path(const NavfnDev&, uint32_t, uint32_t, const int32_t*, const int32_t*, hipStream_t) {}
void launch_gp_wf_finish(const NavfnDev&, uint32_t, uint32_t, const navgpu_global_planner_params&, const double*, const double*, const int32_t*, hipStream_t) {}
}  // namespace navgpu
