// MapGrid wavefronts (gfx950), register-resident row sweeps: k_bfs_rows (one row per lane, maps up to 640 x 800) and
// k_bfs_rows2 (two rows per lane, up to 1024 x 1344).  MapGridCostFunction::prepare (map_grid_cost_function.cpp:59-68) =
//   MapGrid::resetPathDist + adjustPlanResolution (map_grid.cpp:135-171) + setTargetCells (:174-213) | setLocalGoal
//   (:216-258) + computeTargetDistance (:262-310) with updatePathCell (:103-122).
#include "planner_common.h"

namespace navgpu {

#ifdef NAVGPU_BFS_STATS  // experiment builds only (make EXTRA=-DNAVGPU_BFS_STATS, tools/probe_bfs_stats.py): where a level's time goes
__device__ unsigned long long g_bfs_stats[16];  // shader clocks per wave: [0] poll [1] halo + words [2] stores [3] publish [4] group end; [5] wave-levels [6] spins [7] active wave-levels [8] active groups
#define BFS_STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define BFS_ACC(i, x) bst[i] += (x)
#else
#define BFS_STAMP(v)
#define BFS_ACC(i, x)
#endif
// An exchange record (a row's frontier and blocked words) is 2 x Wp words + 4: with records a multiple of 32 words apart, the uint4
// accesses of the seven lanes that publish / take in neighbouring rows fell into the same four LDS banks (SQ_LDS_BANK_CONFLICT 0.50)
#ifndef NAVGPU_ROWS_RECORD_PAD
#define NAVGPU_ROWS_RECORD_PAD 4
#endif
constexpr uint32_t kRowsRecordPad = NAVGPU_ROWS_RECORD_PAD;
template <int W>
__host__ __device__ inline size_t bfs_rows_lds_words(uint32_t nx, uint32_t ny) {
  constexpr uint32_t Wp = (W + 3) & ~3u;
  const uint32_t nw = bfs_rows_waves(ny);
  return (((size_t)nw * kRowsPerWave * ((nx + 31) >> 5) + 3) & ~(size_t)3) + (size_t)kCareRows * kCareWords + (size_t)(nw + 2) * 2 * kRowsHalo * (2 * Wp + kRowsRecordPad);
}
// One group of four words (A B C D, left neighbour word L, right neighbour word R) of one level, skipped as a whole when
// bit g of the wave's active mask is clear.  Per word:
//   x = (f << 1 | left >> 31) | (f >> 1 | right << 31) | up | down;   cand = x & ~blocked;   blocked |= x
// (six vector instructions: two v_alignbit, two v_or with the DPP row shift folded in, v_bitop3, v_or3)
// cand* leave in h* (the words are written back by rowsCommit4 once every group has read the old frontier).
// nz: bit g set when any lane has new cells in the group; lo / hi: when its first / last word has (the neighbouring
// group borders them next level).
__device__ __forceinline__ void rowsGroup4(const int g, const uint32_t aw, uint32_t& nz, uint32_t& lo, uint32_t& hi,
                                           uint32_t& bA, uint32_t& bB, uint32_t& bC, uint32_t& bD, const uint32_t fL, const uint32_t fA,
                                           const uint32_t fB, const uint32_t fC, const uint32_t fD, const uint32_t fR, uint32_t& hA, uint32_t& hB,
                                           uint32_t& hC, uint32_t& hD) {
  uint32_t tA, tB, tC, tD, uA, uB, uC, uD, st;
  asm volatile(
      "s_bitcmp1_b32 %[aw], %[g]\n\t"
      "s_cbranch_scc0 1f\n\t"
      "v_alignbit_b32 %[uA], %[fA], %[fL], 31\n\t"
      "v_alignbit_b32 %[uB], %[fB], %[fA], 31\n\t"
      "v_alignbit_b32 %[uC], %[fC], %[fB], 31\n\t"
      "v_alignbit_b32 %[uD], %[fD], %[fC], 31\n\t"
      "v_alignbit_b32 %[hA], %[fB], %[fA], 1\n\t"
      "v_alignbit_b32 %[hB], %[fC], %[fB], 1\n\t"
      "v_alignbit_b32 %[hC], %[fD], %[fC], 1\n\t"
      "v_alignbit_b32 %[hD], %[fR], %[fD], 1\n\t"
      "v_or_b32_dpp %[tA], %[fA], %[uA] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %[tB], %[fB], %[uB] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %[tC], %[fC], %[uC] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %[tD], %[fD], %[uD] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %[uA], %[fA], %[hA] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %[uB], %[fB], %[hB] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %[uC], %[fC], %[hC] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_or_b32_dpp %[uD], %[fD], %[hD] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
      "v_bitop3_b32 %[hA], %[tA], %[bA], %[uA] bitop3:0x32\n\t"
      "v_bitop3_b32 %[hB], %[tB], %[bB], %[uB] bitop3:0x32\n\t"
      "v_bitop3_b32 %[hC], %[tC], %[bC], %[uC] bitop3:0x32\n\t"
      "v_bitop3_b32 %[hD], %[tD], %[bD], %[uD] bitop3:0x32\n\t"
      "v_or3_b32 %[bA], %[bA], %[tA], %[uA]\n\t"
      "v_or3_b32 %[bB], %[bB], %[tB], %[uB]\n\t"
      "v_or3_b32 %[bC], %[bC], %[tC], %[uC]\n\t"
      "v_or3_b32 %[bD], %[bD], %[tD], %[uD]\n\t"
      "v_or3_b32 %[tA], %[hA], %[hB], %[hC]\n\t"
      "v_or_b32_e32 %[tA], %[tA], %[hD]\n\t"
      "v_cmp_ne_u32_e32 vcc, 0, %[tA]\n\t"
      "s_cbranch_vccz 1f\n\t"
      "s_bitset1_b32 %[nz], %[g]\n\t"
      "v_cmp_ne_u32_e32 vcc, 0, %[hA]\n\t"
      "s_cbranch_vccz 3f\n\t"
      "s_bitset1_b32 %[lo], %[g]\n\t"
      "3:\n\t"
      "v_cmp_ne_u32_e32 vcc, 0, %[hD]\n\t"
      "s_cbranch_vccz 1f\n\t"
      "s_bitset1_b32 %[hi], %[g]\n\t"
      "1:\n\t"
      : [bA] "+v"(bA), [bB] "+v"(bB), [bC] "+v"(bC), [bD] "+v"(bD), [hA] "=&v"(hA), [hB] "=&v"(hB), [hC] "=&v"(hC), [hD] "=&v"(hD), [nz] "+s"(nz),
        [lo] "+s"(lo), [hi] "+s"(hi), [tA] "=&v"(tA), [tB] "=&v"(tB), [tC] "=&v"(tC), [tD] "=&v"(tD),
        [uA] "=&v"(uA), [uB] "=&v"(uB), [uC] "=&v"(uC), [uD] "=&v"(uD), [st] "=&s"(st)
      : [aw] "s"(aw), [g] "n"(g), [fL] "v"(fL), [fA] "v"(fA), [fB] "v"(fB), [fC] "v"(fC), [fD] "v"(fD), [fR] "v"(fR)
      : "vcc", "scc");
}
__device__ __forceinline__ void rowsCommit4(const int g, const uint32_t aw, uint32_t& fA, uint32_t& fB, uint32_t& fC, uint32_t& fD, const uint32_t hA,
                                            const uint32_t hB, const uint32_t hC, const uint32_t hD) {
  asm volatile(
      "s_bitcmp1_b32 %[aw], %[g]\n\t"
      "s_cbranch_scc0 2f\n\t"
      "v_mov_b32 %[fA], %[hA]\n\t"
      "v_mov_b32 %[fB], %[hB]\n\t"
      "v_mov_b32 %[fC], %[hC]\n\t"
      "v_mov_b32 %[fD], %[hD]\n\t"
      "2:\n\t"
      : [fA] "+v"(fA), [fB] "+v"(fB), [fC] "+v"(fC), [fD] "+v"(fD)
      : [aw] "s"(aw), [g] "n"(g), [hA] "v"(hA), [hB] "v"(hB), [hC] "v"(hC), [hD] "v"(hD)
      : "scc");
}
// The seed cells of wavefront `which` of robot `inst`, from its plan (as bfsWaveGrid; map_grid.cpp:160-187, 190-233): every
// lane of the workgroup takes a slice of the plan, `set(mx, my)` is called once per seed cell.
template <typename Set>
__device__ __forceinline__ void rowsPlanSeeds(const PlannerDev& pl, const uint32_t inst, const int which, const Geom& g, const uint8_t* master,
                                              const uint32_t nx, const uint32_t tid, uint32_t* s_wave, Set&& set) {
  const uint32_t n = pl.plan_count[inst];
  const double* P = pl.plan + (size_t)inst * pl.max_plan * 2;
  const bool ovr = which == 2;
  const double lx = pl.front_last[2 * inst], ly = pl.front_last[2 * inst + 1];
  const uint32_t chunk = (n + blockDim.x - 1) / blockDim.x;
  const uint32_t i0 = min(n, tid * chunk), i1 = min(n, i0 + chunk);
  uint32_t mine = 0;
  for (uint32_t i = i0; i < i1; ++i) mine += adjustedPoints(P, i, lx, ly, ovr, n, g.res, true, [](uint32_t, double, double) {});
  uint32_t total;
  const uint32_t base = blockExclusiveScan1024(mine, s_wave, &total);
  auto valid = [&](double x, double y, uint32_t& cell) {
    uint32_t mx, my;
    if (!worldToMap(g, x, y, mx, my)) return false;
    cell = my * nx + mx;
    return master[cell] != kNoInfo;
  };
  uint32_t fmin_ = 0xFFFFFFFFu, b = base;
  for (uint32_t i = i0; i < i1; ++i)
    b += adjustedPoints(P, i, lx, ly, ovr, n, g.res, false, [&](uint32_t k, double x, double y) {
      uint32_t cell;
      if (valid(x, y, cell)) fmin_ = min(fmin_, b + k);
    });
  const uint32_t f = blockMin1024(fmin_, s_wave);
  if (f == 0xFFFFFFFFu) return;  // (uniform over the workgroup)
  uint32_t emin = total;
  b = base;
  for (uint32_t i = i0; i < i1; ++i)
    b += adjustedPoints(P, i, lx, ly, ovr, n, g.res, false, [&](uint32_t k, double x, double y) {
      uint32_t cell;
      if (b + k > f && !valid(x, y, cell)) emin = min(emin, b + k);
    });
  const uint32_t e = blockMin1024(emin, s_wave);
  b = base;
  for (uint32_t i = i0; i < i1; ++i)
    b += adjustedPoints(P, i, lx, ly, ovr, n, g.res, false, [&](uint32_t k, double x, double y) {
      const uint32_t idx = b + k;
      const bool seed = (which == 0) ? (idx >= f && idx < e) : (idx == e - 1);
      if (!seed) return;
      uint32_t cell;
      if (!valid(x, y, cell)) return;
      const uint32_t my = cell / nx;
      set(cell - my * nx, my);
    });
}
template <int W>
__device__ __forceinline__ void bfsRowsGrid(const PlannerDev& pl, const uint32_t inst, const int which, const uint32_t item) {
  constexpr int NG = (W + 3) / 4;         // groups of four words
  constexpr int WP = NG * 4;              // words kept per row: W rounded up (the extra ones are blocked everywhere)
  constexpr int D = kRowsHalo;
  int bx0 = 0, bx1 = -1, by0 = 0, by1 = -1, care_ok = 0;  // the robot's region (box + 2 cells) and whether its pockets are known
  if (pl.bfs_bounded) {
    const int4 bb = reinterpret_cast<const int4*>(pl.bfs_box)[2 * inst];
    bx0 = __builtin_amdgcn_readfirstlane(bb.x);
    bx1 = __builtin_amdgcn_readfirstlane(bb.y);
    by0 = __builtin_amdgcn_readfirstlane(bb.z);
    by1 = __builtin_amdgcn_readfirstlane(bb.w);
    care_ok = __builtin_amdgcn_readfirstlane(pl.bfs_box[8 * inst + 4]);
  }
  if (!(bx1 >= bx0 && by1 >= by0)) {  // (uniform over the workgroup) a whole-grid search: the region is the map, nothing is ever "settled"
    bx0 = 0;
    by0 = 0;
    bx1 = (int)pl.nx - 1;
    by1 = (int)pl.ny - 1;
    care_ok = 0;
  }
  extern __shared__ __align__(16) uint32_t sm[];
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_flag[3];  // rotating by exchange: something new was reached since the last one
  __shared__ uint32_t s_open[3];  //                       something of the robot's box is still open
  uint32_t tid_ = threadIdx.x, nx_ = pl.nx, ny_ = pl.ny;
  asm volatile("" : "+v"(tid_), "+s"(nx_), "+s"(ny_));  // opaque per item, as in bfsWaveGrid
  const uint32_t tid = tid_;
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8] = wall_clock64();
  const Geom g = geomOf(pl, inst);
  const uint32_t nx = nx_, ny = ny_, Wr = (nx + 31) >> 5;  // Wr <= W words really exist
  const uint32_t nw = blockDim.x >> 6;
  const uint32_t lane = tid & 63u, wave_id = tid >> 6;
  // lanes D .. 63 - D own rows wave * 50 .. wave * 50 + 49; the D lanes on either side copy the neighbouring waves' rows
  const int row_i = (int)(wave_id * kRowsPerWave + lane) - D;
  const bool real = row_i >= 0 && row_i < (int)ny;           // the lane's row exists (own or halo)
  const bool owner = real && lane >= (uint32_t)D && lane < 64u - D;
  const uint32_t row = real ? (uint32_t)row_i : 0u;
  const uint32_t rows_p = nw * kRowsPerWave;
  const uint32_t seed_words = (rows_p * Wr + 3u) & ~3u;
  uint32_t* seedm = sm;                                   // [rows_p][Wr], padded to whole 16 bytes
  uint32_t* care_l = sm + seed_words;                     // [kCareRows][kCareWords]
  uint32_t* edge = care_l + kCareRows * kCareWords;       // [nw + 2][top | bottom][D rows][frontier WP | blocked WP]; slot = wave + 1
  constexpr uint32_t RS = 2 * WP + kRowsRecordPad;  // record stride (words)
  const uint32_t edge_words = (nw + 2) * 2 * D * RS;
  const uint8_t* master = pl.master + (size_t)inst * pl.cells_padded;
  const uint32_t* freew = bfsFreeBitmap(pl, which, inst, ny * Wr);
  uint32_t* dist = (which == 0 ? pl.path : (which == 1 ? pl.goal : pl.goal_front)) + (size_t)inst * pl.cells;
  const uint32_t N_obst = pl.cells, N_unreach = pl.cells + 1;
  const uint32_t last_mask = (nx & 31) ? ((1u << (nx & 31)) - 1u) : 0xFFFFFFFFu;
  const bool aligned4 = (nx & 3) == 0;

  for (uint32_t i = tid; i < seed_words + kCareRows * kCareWords + edge_words; i += blockDim.x) sm[i] = 0;
  if (tid < 3) s_flag[tid] = s_open[tid] = 0;
  __syncthreads();
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 4] = wall_clock64();
  if (care_ok) {  // the pocket mask of the robot's box: by region row, four words from the region's first (k_samples)
    const uint32_t* care = pl.bfs_care + (size_t)inst * kCareRows * kCareWords;
    for (uint32_t i = tid; i < (uint32_t)(kCareRows * kCareWords); i += blockDim.x) care_l[i] = care[i];
  }
  // --- seeds from the plan
  rowsPlanSeeds(pl, inst, which, g, master, nx, tid, s_wave, [&](uint32_t mx, uint32_t my) {
    atomicOr(&seedm[my * Wr + (mx >> 5)], 1u << (mx & 31));  // a few hundred seeds, once
  });
  __syncthreads();
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 5] = wall_clock64();

  uint32_t blocked[WP], fr[WP];
#pragma unroll
  for (int j = 0; j < WP; ++j) {
    blocked[j] = 0xFFFFFFFFu;  // rows beyond the grid and words beyond the row never produce cells
    fr[j] = 0;
    if (real && (uint32_t)j < Wr) {  // (halo lanes too: exact copies of the neighbours' rows)
      fr[j] = seedm[row * Wr + j];  // seeds expand whatever their cost (map_grid.cpp:160-187)
      blocked[j] = ~(freew[row * Wr + j] & ((uint32_t)j + 1 == Wr ? last_mask : 0xFFFFFFFFu)) | fr[j];
    }
  }
  // the region in this lane's terms
  const bool wave_in_box = (int)(wave_id * kRowsPerWave) <= by1 && (int)((wave_id + 1) * kRowsPerWave) > by0;  // wave-uniform
  const bool row_in_box = owner && row_i >= by0 && row_i <= by1;
  const int w0 = bx0 >> 5, w1 = bx1 >> 5;
  uint32_t region_groups = 0;  // groups that hold words of the region
  for (int jr = w0; jr <= w1; ++jr) region_groups |= 1u << (jr >> 2);
  uint32_t* drow = dist + (size_t)row * nx;
  // distances of the cells `cells` of word j of this lane's row.  Two plain bit loops (every lane runs the longest one, so
  // their bodies are kept to a find-first-bit, an address and a store): whole aligned groups of four first - fronts that
  // run along a row reach 32 cells of a word at once - then what is left, cell by cell
  auto storeCells = [&](int j, uint32_t cells, uint32_t value) {
    uint32_t* dw = drow + j * 32;
    if (aligned4) {
      uint32_t full = cells & (cells >> 1) & (cells >> 2) & (cells >> 3) & 0x11111111u;
      cells &= ~(full * 15u);
      const uint4 v4 = make_uint4(value, value, value, value);
      while (full) {
        const uint32_t bpos = (uint32_t)__ffs(full) - 1u;
        *reinterpret_cast<uint4*>(dw + bpos) = v4;
        full &= full - 1;
      }
    }
    while (cells) {
      const uint32_t bpos = (uint32_t)__ffs(cells) - 1u;
      dw[bpos] = value;
      cells &= cells - 1;
    }
  };
  // LDS offsets (words) of the 2 * WP-word row record this lane publishes / takes in at an exchange:
  //   own rows 0 .. D-1 (lanes D .. 2D-1) -> this wave's TOP record, read by the wave above into its lanes 64-D .. 63;
  //   own rows 50-D .. 49 (lanes 64-2D .. 63-D) -> BOTTOM record, read by the wave below into its lanes 0 .. D-1
  const bool pub_top = lane >= (uint32_t)D && lane < 2u * D, pub_bot = lane >= 64u - 2 * D && lane < 64u - D;
  const uint32_t pub_wr = (((wave_id + 1) * 2 + (pub_top ? 0u : 1u)) * D + (pub_top ? lane - D : lane - (64u - 2 * D))) * RS;
  const bool halo_top = lane < (uint32_t)D, halo_bot = lane >= 64u - D;
  const uint32_t halo_rd = ((halo_top ? (wave_id * 2 + 1) : ((wave_id + 2) * 2)) * D + (halo_top ? lane : lane - (64u - D))) * RS;
  constexpr uint32_t gmask = (1u << NG) - 1u;

  // which groups hold or border a frontier cell of this wave's 64 rows
  auto activity = [&]() -> uint32_t {
    uint32_t nz = 0, lo = 0, hi = 0;
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const uint32_t t = fr[4 * q] | fr[4 * q + 1] | fr[4 * q + 2] | fr[4 * q + 3];
      if (__builtin_amdgcn_ballot_w64(t != 0) != 0) {
        nz |= 1u << q;
        if (__builtin_amdgcn_ballot_w64(fr[4 * q] != 0) != 0) lo |= 1u << q;
        if (__builtin_amdgcn_ballot_w64(fr[4 * q + 3] != 0) != 0) hi |= 1u << q;
      }
    }
    return (nz | (lo >> 1) | (hi << 1)) & gmask;
  };
  if (wave_in_box) {
#pragma unroll
    for (int j = 0; j < W; ++j)
      if (j >= w0 && j <= w1 && row_in_box) storeCells(j, fr[j], 0u);  // the seeds: distance 0
  }
  uint32_t a_own = activity();
  uint32_t level = 0, xch = 0, any_blk = 0;
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 2] = wall_clock64();
  bool done = false;
#ifdef NAVGPU_BFS_STATS
  unsigned long long bst[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  while (!done) {
    // ---- D levels on the wave's own: registers and DPP only
    for (int k = 0; k < D; ++k) {
      BFS_STAMP(ts1);
      BFS_ACC(5, 1);
      const uint32_t aw = __builtin_amdgcn_readfirstlane(a_own);  // (provably uniform, but the "s" operands below need the compiler to know it)
      BFS_ACC(7, aw != 0 ? 1 : 0);
      BFS_ACC(8, __builtin_popcount(aw));
      if (aw != 0) {
        uint32_t nz = 0, lo = 0, hi = 0;
        // One group = one asm statement that carries its own wave-uniform skip, so the compiler sees straight-line code
        // with in-place (tied) updates of `blocked` and `fr`.  (Written as C++ branches the same loop made it rename both
        // arrays per word: register copies in the path of every SKIPPED word and a dozen more at the loop's back edge.)
        // A group's new frontier waits in h[] until the NEXT group has read the old words (its left neighbour), then goes back.
        uint32_t h[2][4];
        const uint32_t zero = 0;
#pragma unroll
        for (int q = 0; q < NG; ++q) {
          rowsGroup4(q, aw, nz, lo, hi, blocked[4 * q], blocked[4 * q + 1], blocked[4 * q + 2], blocked[4 * q + 3],
                     q > 0 ? fr[q > 0 ? 4 * q - 1 : 0] : zero, fr[4 * q], fr[4 * q + 1], fr[4 * q + 2], fr[4 * q + 3],
                     q + 1 < NG ? fr[q + 1 < NG ? 4 * q + 4 : 0] : zero, h[q & 1][0], h[q & 1][1], h[q & 1][2], h[q & 1][3]);
          if (q > 0) {
            const int p = q > 0 ? q - 1 : 0;
            rowsCommit4(p, aw, fr[4 * p], fr[4 * p + 1], fr[4 * p + 2], fr[4 * p + 3], h[p & 1][0], h[p & 1][1], h[p & 1][2], h[p & 1][3]);
          }
        }
        rowsCommit4(NG - 1, aw, fr[4 * (NG - 1)], fr[4 * (NG - 1) + 1], fr[4 * (NG - 1) + 2], fr[4 * (NG - 1) + 3], h[(NG - 1) & 1][0], h[(NG - 1) & 1][1],
                    h[(NG - 1) & 1][2], h[(NG - 1) & 1][3]);
        BFS_STAMP(ts2);
        BFS_ACC(1, ts2 - ts1);
        // the new cells of the robot's region get their distance now, from the lane that owns the row
        if (wave_in_box && (nz & region_groups) != 0) {
#pragma unroll
          for (int q = 0; q < NG; ++q) {
            if (((nz & region_groups) >> q) & 1u) {  // wave-uniform
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const int j = 4 * q + c;
                if (j < W && j >= w0 && j <= w1) {
#ifndef NAVGPU_BFS_X_NOSTORE  // (timing experiment, results garbage: the levels without their distance stores - 126 M -> 102 M wave-instructions, 0.33 -> 0.27 ms)
                  if (row_in_box && fr[j < W ? j : 0] != 0) storeCells(j, fr[j < W ? j : 0], level + 1);
#endif
                }
              }
            }
          }
        }
        BFS_STAMP(ts3);
        BFS_ACC(2, ts3 - ts2);
        any_blk |= nz;
        a_own = (nz | (lo >> 1) | (hi << 1)) & gmask;
      }
      ++level;
    }
    // ---- exchange: the outer D own rows go to the neighbours, theirs come into the halo lanes; stop flags
    BFS_STAMP(ts4);
    const uint32_t slot = xch % 3u;
    if (pub_top || pub_bot) {
#pragma unroll
      for (int q = 0; q < WP; q += 4) {
        *reinterpret_cast<uint4*>(edge + pub_wr + q) = make_uint4(fr[q], fr[q + 1], fr[q + 2], fr[q + 3]);
        *reinterpret_cast<uint4*>(edge + pub_wr + WP + q) = make_uint4(blocked[q], blocked[q + 1], blocked[q + 2], blocked[q + 3]);
      }
    }
    if (any_blk) s_flag[slot] = 1;
    if (wave_in_box) {  // wave-uniform: is anything of the robot's box still open, or a frontier cell inside the region?
      uint32_t open_any = 0;
      const uint32_t rr = (uint32_t)(row_i - by0);
#pragma unroll
      for (int j = 0; j < W; ++j) {
        if (j >= w0 && j <= w1) {  // wave-uniform
          const uint32_t cw_i = (uint32_t)(j - w0);
          if (row_in_box) {
            const uint32_t care = (care_ok != 0 && cw_i < (uint32_t)kCareWords && rr < (uint32_t)kCareRows) ? care_l[rr * kCareWords + cw_i] : (care_ok ? 0u : 0xFFFFFFFFu);
            const int c_lo = max(bx0 - j * 32, 0), c_hi = min(bx1 - j * 32, 31);
            const uint32_t open = (~blocked[j] & care) | fr[j];
            if (c_hi >= c_lo) open_any |= open & (0xFFFFFFFFu >> (31 - c_hi)) & (0xFFFFFFFFu << c_lo);
          }
        }
      }
      if (open_any != 0) s_open[slot] = 1;
    }
    if (tid == 0) {
      s_flag[(xch + 1) % 3u] = 0;
      s_open[(xch + 1) % 3u] = 0;
    }
    BFS_STAMP(ts5);
    BFS_ACC(3, ts5 - ts4);
    __syncthreads();
    done = !s_flag[slot] || !s_open[slot];  // nothing new in D levels, or nothing open in the box: the search is over
    if (!done) {
      if ((halo_top || halo_bot) && real) {
#pragma unroll
        for (int q = 0; q < WP; q += 4) {
          const uint4 v = *reinterpret_cast<const uint4*>(edge + halo_rd + q);
          const uint4 b = *reinterpret_cast<const uint4*>(edge + halo_rd + WP + q);
          fr[q] = v.x;
          fr[q + 1] = v.y;
          fr[q + 2] = v.z;
          fr[q + 3] = v.w;
          blocked[q] = b.x;
          blocked[q + 1] = b.y;
          blocked[q + 2] = b.z;
          blocked[q + 3] = b.w;
        }
      }
      a_own = activity();
      __syncthreads();  // the records are free for the next exchange
    }
    ++xch;
    any_blk = 0;
    BFS_STAMP(ts6);
    BFS_ACC(4, ts6 - ts5);
  }
#ifdef NAVGPU_BFS_STATS
  if (lane == 0)
    for (int k = 0; k < 9; ++k) atomicAdd(&g_bfs_stats[k], bst[k]);
#endif
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 3] = wall_clock64();

  // --- the rest of the region: obstacle cells an expanded cell touched -> obstacleCosts(), everything else that was not
  // reached -> unreachableCellCosts().  Expanded = reached free cells + seeds.  (The halo lanes next to the own rows are
  // exact copies as of the last exchange; the own rows have moved on since, so their neighbours are exchanged once more.)
  __syncthreads();
  {
    uint32_t ex[WP], fb[WP];
#pragma unroll
    for (int j = 0; j < WP; ++j) {
      const bool in = owner && (uint32_t)j < Wr;
      fb[j] = in ? (freew[row * Wr + j] & ((uint32_t)j + 1 == Wr ? last_mask : 0xFFFFFFFFu)) : 0u;
      ex[j] = in ? ((blocked[j] & fb[j]) | seedm[row * Wr + j]) : 0u;
    }
    if (pub_top || pub_bot) {
#pragma unroll
      for (int j = 0; j < WP; ++j) edge[pub_wr + j] = ex[j];
    }
    __syncthreads();
    if ((halo_top || halo_bot) && real) {
#pragma unroll
      for (int j = 0; j < WP; ++j) ex[j] = edge[halo_rd + j];
    }
    if (wave_in_box) {
#pragma unroll
      for (int j = 0; j < W; ++j) {
        if (j >= w0 && j <= w1) {  // wave-uniform
          const uint32_t fc = ex[j];
          const uint32_t lw = j > 0 ? ex[j > 0 ? j - 1 : 0] : 0u, rw = j + 1 < WP ? ex[j + 1 < WP ? j + 1 : 0] : 0u;
          const uint32_t nbr = __builtin_amdgcn_alignbit(fc, lw, 31) | __builtin_amdgcn_alignbit(rw, fc, 1) | fromLaneBelow(fc) | fromLaneAbove(fc);
          const uint32_t cm = (uint32_t)j + 1 == Wr ? last_mask : ((uint32_t)j < Wr ? 0xFFFFFFFFu : 0u);
          const uint32_t touched = nbr & ~fb[j] & ~fc & cm;
          if (row_in_box) {
            storeCells(j, touched, N_obst);
            storeCells(j, ~fc & ~touched & cm, N_unreach);
          }
        }
      }
    }
  }
  if (pl.bfs_trace && tid == 0) {
    pl.bfs_trace[(size_t)item * 8 + 6] = wall_clock64();
    pl.bfs_trace[(size_t)item * 8 + 1] = wall_clock64() | ((unsigned long long)level << 48);
  }
  if (tid == 0 && pl.bfs_grids == 3) pl.bfs_levels[(size_t)inst * 3 + which] = level;  // next cycle's dispatch order
}
template <int W>
__global__ __launch_bounds__(1024, 6) void k_bfs_rows(PlannerDev pl, uint32_t first, uint32_t count, uint32_t* next_item, const uint32_t* order) {
  __shared__ uint32_t s_item;
  const uint32_t total = count * pl.bfs_grids;  // 3 (DWA: path, goal, goal_front) or 2 (legacy TrajectoryPlanner)
  for (;;) {
    if (threadIdx.x == 0) s_item = atomicAdd(next_item, 1u);
    __syncthreads();
    const uint32_t slot = s_item;
    if (slot >= total) break;  // (every workgroup gets here: the counter only grows)
    const uint32_t item = order ? order[slot] : slot;  // longest searches first
    const uint32_t g = item / count;
    bfsRowsGrid<W>(pl, first + (item - g * count), (int)pl.bfs_grids - 1 - (int)g, item);
    __syncthreads();
  }
}
// ------------------------------------------------------------------------------------------------
// k_bfs_rows2: the row sweep of k_bfs_rows for maps up to 1024 cells wide and 1344 rows (configs[4]'s 1000 x 1000), TWO rows
// per lane.  A lane keeps rows (A, B) = (2k, 2k + 1) of its wave's block: row A's upper neighbour is row B of the lane
// above (DPP), its lower one the lane's own row B (a register); row B's upper neighbour is the lane's own row A, its lower
// one row A of the lane below.  A wave owns 112 rows (56 lanes) and copies 8 rows (4 lanes) of either neighbour, so the
// waves meet every 8 levels; 1000 rows are 9 waves.  2 x 2 x 32 words of state per lane: 168 registers, three waves per
// SIMD, one search per CU.  The seed bitmap does not fit LDS next to the exchange records and lives in global scratch
// (one per workgroup), read with plain loads behind an agent-scope acquire.  Everything else is k_bfs_rows.
// ------------------------------------------------------------------------------------------------
constexpr int kRows2HaloLanes = 4;                                // lanes a wave copies from either neighbour
constexpr int kRows2Levels = 2 * kRows2HaloLanes;                 // = rows copied = levels between two exchanges
constexpr int kRows2PerWave = 2 * (64 - 2 * kRows2HaloLanes);     // rows a wave owns
constexpr int kRows2Words = 32;
__host__ __device__ inline uint32_t bfs_rows2_waves(uint32_t ny) { return (ny + kRows2PerWave - 1) / kRows2PerWave; }
__host__ __device__ inline size_t bfs_rows2_lds_words(uint32_t ny) {
  return (size_t)kCareRows * kCareWords + (size_t)(bfs_rows2_waves(ny) + 2) * 2 * kRows2HaloLanes * 4 * kRows2Words;
}
// One group of four words of BOTH rows of a lane, one level, in place; skipped as a whole when bit g of the wave's active mask
// is clear.  Per word (rowsGroup4's arithmetic with the vertical neighbours of a row pair):
//   row A: x = left | right | (row B of the lane above) | own row B;   row B: x = left | right | own row A | (row A of the lane below)
//   new frontier = x & ~blocked;   blocked |= x
// The words are updated where they stand, so the old last word of the group is kept in pA / pB for the next group's left
// neighbour.  They are valid only if this group was live; a group that is not holds no frontier cell (it either never had one
// or wrote its empty result back the level it went quiet), so its neighbour shifts in zeros instead (the G > 0 prologue).
// rA / rB: the first word of the next group (still old).  nz / lo / hi as rowsGroup4.
template <int G>
__device__ __forceinline__ void rows2Group(const uint32_t aw, uint32_t& nz, uint32_t& lo, uint32_t& hi, uint32_t* __restrict__ bA, uint32_t* __restrict__ bB,
                                           uint32_t* __restrict__ fA, uint32_t* __restrict__ fB, uint32_t& pA, uint32_t& pB, const uint32_t rA,
                                           const uint32_t rB) {
  uint32_t u0, u1, u2, u3, v0, v1, v2, v3, h0, h1, h2, h3, st;
#define NAVGPU_ROWS2_HEAD_FIRST                \
  "s_bitcmp1_b32 %[aw], %[g]\n\t"              \
  "s_cbranch_scc0 1f\n\t"                      \
  "v_lshlrev_b32 %[u0], 1, %[fA0]\n\t"         \
  "v_lshlrev_b32 %[v0], 1, %[fB0]\n\t"
#define NAVGPU_ROWS2_HEAD_NEXT                 \
  "s_bitcmp1_b32 %[aw], %[g]\n\t"              \
  "s_cbranch_scc0 1f\n\t"                      \
  "s_bitcmp1_b32 %[aw], %[gp]\n\t"             \
  "s_cbranch_scc1 3f\n\t"                      \
  "v_lshlrev_b32 %[u0], 1, %[fA0]\n\t"         \
  "v_lshlrev_b32 %[v0], 1, %[fB0]\n\t"         \
  "s_branch 4f\n\t"                            \
  "3:\n\t"                                     \
  "v_alignbit_b32 %[u0], %[fA0], %[pA], 31\n\t" \
  "v_alignbit_b32 %[v0], %[fB0], %[pB], 31\n\t" \
  "4:\n\t"
#define NAVGPU_ROWS2_BODY(HEAD)                                                                        \
  asm volatile(                                                                                        \
      HEAD                                                                                             \
      "v_alignbit_b32 %[u1], %[fA1], %[fA0], 31\n\t"                                                   \
      "v_alignbit_b32 %[u2], %[fA2], %[fA1], 31\n\t"                                                   \
      "v_alignbit_b32 %[u3], %[fA3], %[fA2], 31\n\t"                                                   \
      "v_alignbit_b32 %[v1], %[fB1], %[fB0], 31\n\t"                                                   \
      "v_alignbit_b32 %[v2], %[fB2], %[fB1], 31\n\t"                                                   \
      "v_alignbit_b32 %[v3], %[fB3], %[fB2], 31\n\t"                                                   \
      "v_mov_b32 %[pA], %[fA3]\n\t"                                                                    \
      "v_mov_b32 %[pB], %[fB3]\n\t"                                                                    \
      "v_alignbit_b32 %[h0], %[fA1], %[fA0], 1\n\t"                                                    \
      "v_alignbit_b32 %[h1], %[fA2], %[fA1], 1\n\t"                                                    \
      "v_alignbit_b32 %[h2], %[fA3], %[fA2], 1\n\t"                                                    \
      "v_alignbit_b32 %[h3], %[rA], %[fA3], 1\n\t"                                                     \
      "v_or_b32_dpp %[u0], %[fB0], %[u0] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
      "v_or_b32_dpp %[u1], %[fB1], %[u1] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
      "v_or_b32_dpp %[u2], %[fB2], %[u2] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
      "v_or_b32_dpp %[u3], %[fB3], %[u3] wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
      "v_or3_b32 %[u0], %[u0], %[h0], %[fB0]\n\t"                                                      \
      "v_or3_b32 %[u1], %[u1], %[h1], %[fB1]\n\t"                                                      \
      "v_or3_b32 %[u2], %[u2], %[h2], %[fB2]\n\t"                                                      \
      "v_or3_b32 %[u3], %[u3], %[h3], %[fB3]\n\t"                                                      \
      "v_alignbit_b32 %[h0], %[fB1], %[fB0], 1\n\t"                                                    \
      "v_alignbit_b32 %[h1], %[fB2], %[fB1], 1\n\t"                                                    \
      "v_alignbit_b32 %[h2], %[fB3], %[fB2], 1\n\t"                                                    \
      "v_alignbit_b32 %[h3], %[rB], %[fB3], 1\n\t"                                                     \
      "v_or_b32_dpp %[v0], %[fA0], %[v0] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
      "v_or_b32_dpp %[v1], %[fA1], %[v1] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
      "v_or_b32_dpp %[v2], %[fA2], %[v2] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
      "v_or_b32_dpp %[v3], %[fA3], %[v3] wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"       \
      "v_or3_b32 %[v0], %[v0], %[h0], %[fA0]\n\t"                                                      \
      "v_or3_b32 %[v1], %[v1], %[h1], %[fA1]\n\t"                                                      \
      "v_or3_b32 %[v2], %[v2], %[h2], %[fA2]\n\t"                                                      \
      "v_or3_b32 %[v3], %[v3], %[h3], %[fA3]\n\t"                                                      \
      "v_bitop3_b32 %[fA0], %[u0], %[bA0], %[u0] bitop3:0x30\n\t"                                      \
      "v_bitop3_b32 %[fA1], %[u1], %[bA1], %[u1] bitop3:0x30\n\t"                                      \
      "v_bitop3_b32 %[fA2], %[u2], %[bA2], %[u2] bitop3:0x30\n\t"                                      \
      "v_bitop3_b32 %[fA3], %[u3], %[bA3], %[u3] bitop3:0x30\n\t"                                      \
      "v_bitop3_b32 %[fB0], %[v0], %[bB0], %[v0] bitop3:0x30\n\t"                                      \
      "v_bitop3_b32 %[fB1], %[v1], %[bB1], %[v1] bitop3:0x30\n\t"                                      \
      "v_bitop3_b32 %[fB2], %[v2], %[bB2], %[v2] bitop3:0x30\n\t"                                      \
      "v_bitop3_b32 %[fB3], %[v3], %[bB3], %[v3] bitop3:0x30\n\t"                                      \
      "v_or_b32_e32 %[bA0], %[bA0], %[u0]\n\t"                                                         \
      "v_or_b32_e32 %[bA1], %[bA1], %[u1]\n\t"                                                         \
      "v_or_b32_e32 %[bA2], %[bA2], %[u2]\n\t"                                                         \
      "v_or_b32_e32 %[bA3], %[bA3], %[u3]\n\t"                                                         \
      "v_or_b32_e32 %[bB0], %[bB0], %[v0]\n\t"                                                         \
      "v_or_b32_e32 %[bB1], %[bB1], %[v1]\n\t"                                                         \
      "v_or_b32_e32 %[bB2], %[bB2], %[v2]\n\t"                                                         \
      "v_or_b32_e32 %[bB3], %[bB3], %[v3]\n\t"                                                         \
      "v_or_b32_e32 %[h0], %[fA0], %[fB0]\n\t"                                                         \
      "v_or_b32_e32 %[h3], %[fA3], %[fB3]\n\t"                                                         \
      "v_or3_b32 %[h1], %[fA1], %[fA2], %[fB1]\n\t"                                                    \
      "v_or3_b32 %[h2], %[h0], %[h3], %[fB2]\n\t"                                                      \
      "v_or_b32_e32 %[h1], %[h1], %[h2]\n\t"                                                           \
      "v_cmp_ne_u32_e32 vcc, 0, %[h1]\n\t"                                                             \
      "s_cbranch_vccz 1f\n\t"                                                                          \
      "s_bitset1_b32 %[nz], %[g]\n\t"                                                                  \
      "v_cmp_ne_u32_e32 vcc, 0, %[h0]\n\t"                                                             \
      "s_cbranch_vccz 3f\n\t"                                                                          \
      "s_bitset1_b32 %[lo], %[g]\n\t"                                                                  \
      "3:\n\t"                                                                                         \
      "v_cmp_ne_u32_e32 vcc, 0, %[h3]\n\t"                                                             \
      "s_cbranch_vccz 1f\n\t"                                                                          \
      "s_bitset1_b32 %[hi], %[g]\n\t"                                                                  \
      "1:\n\t"                                                                                         \
      : [bA0] "+v"(bA[0]), [bA1] "+v"(bA[1]), [bA2] "+v"(bA[2]), [bA3] "+v"(bA[3]), [bB0] "+v"(bB[0]), [bB1] "+v"(bB[1]), [bB2] "+v"(bB[2]),  \
        [bB3] "+v"(bB[3]), [fA0] "+v"(fA[0]), [fA1] "+v"(fA[1]), [fA2] "+v"(fA[2]), [fA3] "+v"(fA[3]), [fB0] "+v"(fB[0]), [fB1] "+v"(fB[1]),   \
        [fB2] "+v"(fB[2]), [fB3] "+v"(fB[3]), [pA] "+v"(pA), [pB] "+v"(pB), [nz] "+s"(nz), [lo] "+s"(lo), [hi] "+s"(hi), [u0] "=&v"(u0),       \
        [u1] "=&v"(u1), [u2] "=&v"(u2), [u3] "=&v"(u3), [v0] "=&v"(v0), [v1] "=&v"(v1), [v2] "=&v"(v2), [v3] "=&v"(v3), [h0] "=&v"(h0),        \
        [h1] "=&v"(h1), [h2] "=&v"(h2), [h3] "=&v"(h3), [st] "=&s"(st)                                                                         \
      : [aw] "s"(aw), [g] "n"(G), [gp] "n"(G > 0 ? G - 1 : 0), [rA] "v"(rA), [rB] "v"(rB)                                                      \
      : "vcc", "scc")
  if constexpr (G == 0) NAVGPU_ROWS2_BODY(NAVGPU_ROWS2_HEAD_FIRST);
  else NAVGPU_ROWS2_BODY(NAVGPU_ROWS2_HEAD_NEXT);
#undef NAVGPU_ROWS2_BODY
#undef NAVGPU_ROWS2_HEAD_FIRST
#undef NAVGPU_ROWS2_HEAD_NEXT
}
// the NG groups of a level, first to last (a compile-time recursion: the group number is an immediate of the asm block)
template <int G>
__device__ __forceinline__ void rows2Level(const uint32_t aw, uint32_t& nz, uint32_t& lo, uint32_t& hi, uint32_t* __restrict__ blA, uint32_t* __restrict__ blB,
                                           uint32_t* __restrict__ frA, uint32_t* __restrict__ frB, uint32_t& pA, uint32_t& pB, const uint32_t zero) {
  constexpr int NG = kRows2Words / 4;
  if constexpr (G < NG) {
    rows2Group<G>(aw, nz, lo, hi, blA + 4 * G, blB + 4 * G, frA + 4 * G, frB + 4 * G, pA, pB, G + 1 < NG ? frA[G + 1 < NG ? 4 * G + 4 : 0] : zero,
                  G + 1 < NG ? frB[G + 1 < NG ? 4 * G + 4 : 0] : zero);
    rows2Level<G + 1>(aw, nz, lo, hi, blA, blB, frA, frB, pA, pB, zero);
  }
}
__device__ __forceinline__ void bfsRows2Grid(const PlannerDev& pl, const uint32_t inst, const int which, const uint32_t item, uint32_t* seedw) {
  constexpr int W = kRows2Words, NG = W / 4, D = kRows2Levels, HL = kRows2HaloLanes;
  int bx0 = 0, bx1 = -1, by0 = 0, by1 = -1, care_ok = 0;  // the robot's region (box + 2 cells) and whether its pockets are known
  if (pl.bfs_bounded) {
    const int4 bb = reinterpret_cast<const int4*>(pl.bfs_box)[2 * inst];
    bx0 = __builtin_amdgcn_readfirstlane(bb.x);
    bx1 = __builtin_amdgcn_readfirstlane(bb.y);
    by0 = __builtin_amdgcn_readfirstlane(bb.z);
    by1 = __builtin_amdgcn_readfirstlane(bb.w);
    care_ok = __builtin_amdgcn_readfirstlane(pl.bfs_box[8 * inst + 4]);
  }
  if (!(bx1 >= bx0 && by1 >= by0)) {  // (uniform over the workgroup) a whole-grid search: the region is the map
    bx0 = 0;
    by0 = 0;
    bx1 = (int)pl.nx - 1;
    by1 = (int)pl.ny - 1;
    care_ok = 0;
  }
  extern __shared__ __align__(16) uint32_t sm[];
  __shared__ uint32_t s_wave[16];
  __shared__ uint32_t s_flag[3];
  __shared__ uint32_t s_open[3];
  uint32_t tid_ = threadIdx.x, nx_ = pl.nx, ny_ = pl.ny;
  asm volatile("" : "+v"(tid_), "+s"(nx_), "+s"(ny_));  // opaque per item, as in bfsWaveGrid
  const uint32_t tid = tid_;
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8] = wall_clock64();
  const Geom g = geomOf(pl, inst);
  const uint32_t nx = nx_, ny = ny_, Wr = (nx + 31) >> 5;  // Wr <= 32 words really exist
  const uint32_t nw = blockDim.x >> 6;
  const uint32_t lane = tid & 63u, wave_id = tid >> 6;
  // lanes HL .. 63 - HL own rows wave * 112 + 2 (lane - HL) and the one after; the HL lanes on either side copy the neighbours'
  const int rowA_i = (int)(wave_id * kRows2PerWave) + 2 * ((int)lane - HL), rowB_i = rowA_i + 1;
  const bool own_lane = lane >= (uint32_t)HL && lane < 64u - HL;
  const bool realA = rowA_i >= 0 && rowA_i < (int)ny, realB = rowB_i >= 0 && rowB_i < (int)ny;
  const bool ownerA = realA && own_lane, ownerB = realB && own_lane;
  const uint32_t rowA = realA ? (uint32_t)rowA_i : 0u, rowB = realB ? (uint32_t)rowB_i : 0u;
  uint32_t* care_l = sm;                             // [kCareRows][kCareWords]
  uint32_t* edge = care_l + kCareRows * kCareWords;  // [nw + 2][top | bottom][HL lanes][A frontier | A blocked | B frontier | B blocked][W]; slot = wave + 1
  const uint32_t edge_words = (nw + 2) * 2 * HL * 4 * W;
  const uint8_t* master = pl.master + (size_t)inst * pl.cells_padded;
  const uint32_t* freew = bfsFreeBitmap(pl, which, inst, ny * Wr);
  uint32_t* dist = (which == 0 ? pl.path : (which == 1 ? pl.goal : pl.goal_front)) + (size_t)inst * pl.cells;
  const uint32_t N_obst = pl.cells, N_unreach = pl.cells + 1;
  const uint32_t last_mask = (nx & 31) ? ((1u << (nx & 31)) - 1u) : 0xFFFFFFFFu;
  const bool aligned4 = (nx & 3) == 0;

  for (uint32_t i = tid; i < kCareRows * kCareWords + edge_words; i += blockDim.x) sm[i] = 0;
  for (uint32_t i = tid; i < ny * Wr; i += blockDim.x) seedw[i] = 0;
  if (tid < 3) s_flag[tid] = s_open[tid] = 0;
  __syncthreads();
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 4] = wall_clock64();
  if (care_ok) {  // the pocket mask of the robot's box: by region row, four words from the region's first (k_samples)
    const uint32_t* care = pl.bfs_care + (size_t)inst * kCareRows * kCareWords;
    for (uint32_t i = tid; i < (uint32_t)(kCareRows * kCareWords); i += blockDim.x) care_l[i] = care[i];
  }
  rowsPlanSeeds(pl, inst, which, g, master, nx, tid, s_wave, [&](uint32_t mx, uint32_t my) {
    atomicOr(&seedw[my * Wr + (mx >> 5)], 1u << (mx & 31));
  });
  __syncthreads();
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 5] = wall_clock64();
  // The seed words were zeroed and set by other lanes of this workgroup and read by an earlier item: drop this CU's stale L1
  // lines, then plain loads see what the atomics left in L2.
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  auto colMask = [&](int j) { return (uint32_t)j + 1 == Wr ? last_mask : 0xFFFFFFFFu; };
  // words 4q .. 4q + 3 of a row of a [ny][Wr] bitmap: one 16-byte load where rows are whole 16-byte units (1000 cells: 32
  // words).  A lane's two rows are 256 contiguous bytes; word by word a wave's load touched 64 cache lines for 256 bytes
  // and the 128 loads of a lane took 150 us per search (tools/trace_bfs_configs4.py)
  const bool rows16 = (Wr & 3u) == 0;
  auto rowWords = [&](const uint32_t* base, uint32_t row, int q) -> uint4 {
    uint4 v = make_uint4(0, 0, 0, 0);
    if (rows16) {
      if ((uint32_t)(4 * q) < Wr) v = *reinterpret_cast<const uint4*>(base + (size_t)row * Wr + 4 * q);
    } else {
      const uint32_t* r = base + (size_t)row * Wr;
      if ((uint32_t)(4 * q) < Wr) v.x = r[4 * q];
      if ((uint32_t)(4 * q + 1) < Wr) v.y = r[4 * q + 1];
      if ((uint32_t)(4 * q + 2) < Wr) v.z = r[4 * q + 2];
      if ((uint32_t)(4 * q + 3) < Wr) v.w = r[4 * q + 3];
    }
    return v;
  };

  uint32_t blA[W], frA[W], blB[W], frB[W];
#pragma unroll
  for (int q = 0; q < NG; ++q) {
    // rows beyond the grid and words beyond the row never produce cells; halo lanes hold exact copies of the neighbours' rows;
    // seeds expand whatever their cost (map_grid.cpp:160-187)
    const uint4 sA = realA ? rowWords(seedw, rowA, q) : make_uint4(0, 0, 0, 0), sB = realB ? rowWords(seedw, rowB, q) : make_uint4(0, 0, 0, 0);
    const uint4 fA = realA ? rowWords(freew, rowA, q) : make_uint4(0, 0, 0, 0), fB = realB ? rowWords(freew, rowB, q) : make_uint4(0, 0, 0, 0);
    const uint32_t sa[4] = {sA.x, sA.y, sA.z, sA.w}, sb[4] = {sB.x, sB.y, sB.z, sB.w}, fa[4] = {fA.x, fA.y, fA.z, fA.w}, fb[4] = {fB.x, fB.y, fB.z, fB.w};
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int j = 4 * q + c;
      const bool in = (uint32_t)j < Wr;
      // (a real move: as plain copies the frontier words stay tied to the 4-register tuples of the loads for the whole sweep,
      // and the allocator spills whole tuples inside the level loop)
      asm volatile("v_mov_b32 %0, %1" : "=v"(frA[j]) : "v"(sa[c]));
      asm volatile("v_mov_b32 %0, %1" : "=v"(frB[j]) : "v"(sb[c]));
      blA[j] = (in && realA) ? (~(fa[c] & colMask(j)) | sa[c]) : 0xFFFFFFFFu;
      blB[j] = (in && realB) ? (~(fb[c] & colMask(j)) | sb[c]) : 0xFFFFFFFFu;
    }
  }
  const bool wave_in_box = (int)(wave_id * kRows2PerWave) <= by1 && (int)((wave_id + 1) * kRows2PerWave) > by0;  // wave-uniform
  const bool inA = ownerA && rowA_i >= by0 && rowA_i <= by1, inB = ownerB && rowB_i >= by0 && rowB_i <= by1;
  const int w0 = bx0 >> 5, w1 = bx1 >> 5;
  uint32_t region_groups = 0;
  for (int jr = w0; jr <= w1; ++jr) region_groups |= 1u << (jr >> 2);
  auto storeCells = [&](const uint32_t row, int j, uint32_t cells, uint32_t value) {  // as bfsRowsGrid's (the row's address is worked out here: registers)
    uint32_t* dw = dist + (size_t)row * nx + j * 32;
    if (aligned4) {
      uint32_t full = cells & (cells >> 1) & (cells >> 2) & (cells >> 3) & 0x11111111u;
      cells &= ~(full * 15u);
      const uint4 v4 = make_uint4(value, value, value, value);
      while (full) {
        const uint32_t bpos = (uint32_t)__ffs(full) - 1u;
        *reinterpret_cast<uint4*>(dw + bpos) = v4;
        full &= full - 1;
      }
    }
    while (cells) {
      const uint32_t bpos = (uint32_t)__ffs(cells) - 1u;
      dw[bpos] = value;
      cells &= cells - 1;
    }
  };
  // LDS offsets (words) of the 4 * W-word record (both rows) this lane publishes / takes in at an exchange:
  //   lanes HL .. 2 HL - 1 -> this wave's TOP record, read by the wave above into its lanes 64 - HL .. 63;
  //   lanes 64 - 2 HL .. 63 - HL -> BOTTOM record, read by the wave below into its lanes 0 .. HL - 1
  const bool pub_top = lane >= (uint32_t)HL && lane < 2u * HL, pub_bot = lane >= 64u - 2 * HL && lane < 64u - HL;
  const uint32_t pub_wr = (((wave_id + 1) * 2 + (pub_top ? 0u : 1u)) * HL + (pub_top ? lane - HL : lane - (64u - 2 * HL))) * 4 * W;
  const bool halo_top = lane < (uint32_t)HL, halo_bot = lane >= 64u - HL;
  const uint32_t halo_rd = ((halo_top ? (wave_id * 2 + 1) : ((wave_id + 2) * 2)) * HL + (halo_top ? lane : lane - (64u - HL))) * 4 * W;
  constexpr uint32_t gmask = (1u << NG) - 1u;

  auto activity = [&]() -> uint32_t {
    uint32_t nz = 0, lo = 0, hi = 0;
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const uint32_t t = frA[4 * q] | frA[4 * q + 1] | frA[4 * q + 2] | frA[4 * q + 3] | frB[4 * q] | frB[4 * q + 1] | frB[4 * q + 2] | frB[4 * q + 3];
      if (__builtin_amdgcn_ballot_w64(t != 0) != 0) {
        nz |= 1u << q;
        if (__builtin_amdgcn_ballot_w64((frA[4 * q] | frB[4 * q]) != 0) != 0) lo |= 1u << q;
        if (__builtin_amdgcn_ballot_w64((frA[4 * q + 3] | frB[4 * q + 3]) != 0) != 0) hi |= 1u << q;
      }
    }
    return (nz | (lo >> 1) | (hi << 1)) & gmask;
  };
  if (wave_in_box) {
#pragma unroll
    for (int j = 0; j < W; ++j)
      if (j >= w0 && j <= w1) {
        if (inA) storeCells(rowA, j, frA[j], 0u);  // the seeds: distance 0
        if (inB) storeCells(rowB, j, frB[j], 0u);
      }
  }
  uint32_t a_own = activity();
  uint32_t level = 0, xch = 0, any_blk = 0;
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 2] = wall_clock64();
  bool done = false;
#ifdef NAVGPU_BFS_STATS
  unsigned long long bst[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  while (!done) {
    // ---- D levels on the wave's own: registers and DPP only
    for (int k = 0; k < D; ++k) {
      BFS_STAMP(ts1);
      BFS_ACC(5, 1);
      const uint32_t aw = __builtin_amdgcn_readfirstlane(a_own);
      BFS_ACC(7, aw != 0 ? 1 : 0);
      BFS_ACC(8, __builtin_popcount(aw));
      if (aw != 0) {
        uint32_t nz = 0, lo = 0, hi = 0;
        uint32_t pA = 0, pB = 0;  // the old last words of the previous live group
        const uint32_t zero = 0;
        rows2Level<0>(aw, nz, lo, hi, blA, blB, frA, frB, pA, pB, zero);
        BFS_STAMP(ts2);
        BFS_ACC(1, ts2 - ts1);
        // the new cells of the robot's region get their distance now, from the lane that owns the row
        if (wave_in_box && (nz & region_groups) != 0) {
#pragma unroll
          for (int q = 0; q < NG; ++q) {
            if (((nz & region_groups) >> q) & 1u) {  // wave-uniform
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const int j = 4 * q + c;
                if (j >= w0 && j <= w1) {
                  if (inA && frA[j] != 0) storeCells(rowA, j, frA[j], level + 1);
                  if (inB && frB[j] != 0) storeCells(rowB, j, frB[j], level + 1);
                }
              }
            }
          }
        }
        BFS_STAMP(ts3);
        BFS_ACC(2, ts3 - ts2);
        any_blk |= nz;
        a_own = (nz | (lo >> 1) | (hi << 1)) & gmask;
      }
      ++level;
    }
    // ---- exchange: the outer D own rows go to the neighbours, theirs come into the halo lanes; stop flags
    BFS_STAMP(ts4);
    const uint32_t slot = xch % 3u;
    // (the words pass through real moves on their way to and from the 16-byte LDS accesses: tied to those 4-register tuples
    // the allocator keeps the state in tuples for the whole sweep and spills them inside the level loop)
    auto quad = [&](const uint32_t* w4) {
      uint32_t a, b, c, d;
      asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7" : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d) : "v"(w4[0]), "v"(w4[1]), "v"(w4[2]), "v"(w4[3]));
      return make_uint4(a, b, c, d);
    };
    auto unquad = [&](const uint4 v, uint32_t* w4) {
      asm volatile("v_mov_b32 %0, %4\n\tv_mov_b32 %1, %5\n\tv_mov_b32 %2, %6\n\tv_mov_b32 %3, %7" : "=&v"(w4[0]), "=&v"(w4[1]), "=&v"(w4[2]), "=&v"(w4[3]) : "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
    };
    if (pub_top || pub_bot) {
#pragma unroll
      for (int q = 0; q < W; q += 4) {
        *reinterpret_cast<uint4*>(edge + pub_wr + q) = quad(frA + q);
        *reinterpret_cast<uint4*>(edge + pub_wr + W + q) = quad(blA + q);
        *reinterpret_cast<uint4*>(edge + pub_wr + 2 * W + q) = quad(frB + q);
        *reinterpret_cast<uint4*>(edge + pub_wr + 3 * W + q) = quad(blB + q);
      }
    }
    if (any_blk) s_flag[slot] = 1;
    if (wave_in_box) {  // wave-uniform: is anything of the robot's box still open, or a frontier cell inside the region?
      uint32_t open_any = 0;
      const uint32_t rrA = (uint32_t)(rowA_i - by0), rrB = (uint32_t)(rowB_i - by0);
#pragma unroll
      for (int j = 0; j < W; ++j) {
        if (j >= w0 && j <= w1) {  // wave-uniform
          const uint32_t cw_i = (uint32_t)(j - w0);
          const int c_lo = max(bx0 - j * 32, 0), c_hi = min(bx1 - j * 32, 31);
          const uint32_t cm = c_hi >= c_lo ? ((0xFFFFFFFFu >> (31 - c_hi)) & (0xFFFFFFFFu << c_lo)) : 0u;
          if (inA) {
            const uint32_t care = (care_ok != 0 && cw_i < (uint32_t)kCareWords && rrA < (uint32_t)kCareRows) ? care_l[rrA * kCareWords + cw_i] : (care_ok ? 0u : 0xFFFFFFFFu);
            open_any |= ((~blA[j] & care) | frA[j]) & cm;
          }
          if (inB) {
            const uint32_t care = (care_ok != 0 && cw_i < (uint32_t)kCareWords && rrB < (uint32_t)kCareRows) ? care_l[rrB * kCareWords + cw_i] : (care_ok ? 0u : 0xFFFFFFFFu);
            open_any |= ((~blB[j] & care) | frB[j]) & cm;
          }
        }
      }
      if (open_any != 0) s_open[slot] = 1;
    }
    if (tid == 0) {
      s_flag[(xch + 1) % 3u] = 0;
      s_open[(xch + 1) % 3u] = 0;
    }
    BFS_STAMP(ts5);
    BFS_ACC(3, ts5 - ts4);
    __syncthreads();
    done = !s_flag[slot] || !s_open[slot];  // nothing new in D levels, or nothing open in the box: the search is over
    if (!done) {
      if (halo_top || halo_bot) {
#pragma unroll
        for (int q = 0; q < W; q += 4) {
          if (realA) {
            unquad(*reinterpret_cast<const uint4*>(edge + halo_rd + q), frA + q);
            unquad(*reinterpret_cast<const uint4*>(edge + halo_rd + W + q), blA + q);
          }
          if (realB) {
            unquad(*reinterpret_cast<const uint4*>(edge + halo_rd + 2 * W + q), frB + q);
            unquad(*reinterpret_cast<const uint4*>(edge + halo_rd + 3 * W + q), blB + q);
          }
        }
      }
      a_own = activity();
      __syncthreads();  // the records are free for the next exchange
    }
    ++xch;
    any_blk = 0;
    BFS_STAMP(ts6);
    BFS_ACC(4, ts6 - ts5);
  }
#ifdef NAVGPU_BFS_STATS
  if (lane == 0)
    for (int k = 0; k < 9; ++k) atomicAdd(&g_bfs_stats[k], bst[k]);
#endif
  if (pl.bfs_trace && tid == 0) pl.bfs_trace[(size_t)item * 8 + 3] = wall_clock64();

  // --- the rest of the region: obstacle cells an expanded cell touched -> obstacleCosts(), everything else that was not
  // reached -> unreachableCellCosts().  Expanded = reached free cells + seeds.  (The own rows have moved on since the last
  // exchange, so the rows next to a wave's block are exchanged once more.)
  __syncthreads();
  {
    // (in place: blocked -> expanded)
#pragma unroll
    for (int q = 0; q < NG; ++q) {
      const uint4 sA = ownerA ? rowWords(seedw, rowA, q) : make_uint4(0, 0, 0, 0), sB = ownerB ? rowWords(seedw, rowB, q) : make_uint4(0, 0, 0, 0);
      const uint4 fA = ownerA ? rowWords(freew, rowA, q) : make_uint4(0, 0, 0, 0), fB = ownerB ? rowWords(freew, rowB, q) : make_uint4(0, 0, 0, 0);
      const uint32_t sa[4] = {sA.x, sA.y, sA.z, sA.w}, sb[4] = {sB.x, sB.y, sB.z, sB.w}, fa[4] = {fA.x, fA.y, fA.z, fA.w}, fb[4] = {fB.x, fB.y, fB.z, fB.w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int j = 4 * q + c;
        const bool in = (uint32_t)j < Wr;
        blA[j] = (in && ownerA) ? ((blA[j] & fa[c] & colMask(j)) | sa[c]) : 0u;
        blB[j] = (in && ownerB) ? ((blB[j] & fb[c] & colMask(j)) | sb[c]) : 0u;
      }
    }
    // only the row next to the neighbouring wave's block is needed: row B of its last own lane / row A of its first
    if (lane == (uint32_t)HL || lane == 63u - HL) {
      const bool top = lane == (uint32_t)HL;
      uint32_t* rec = edge + ((wave_id + 1) * 2 + (top ? 0u : 1u)) * HL * 4 * W;
#pragma unroll
      for (int j = 0; j < W; ++j) rec[j] = top ? blA[j] : blB[j];
    }
    __syncthreads();
    if (lane == (uint32_t)HL - 1u && realB) {  // the last halo lane above: its row B is the row over this wave's first
      const uint32_t* rec = edge + (wave_id * 2 + 1) * HL * 4 * W;
#pragma unroll
      for (int j = 0; j < W; ++j) blB[j] = rec[j];
    }
    if (lane == 64u - HL && realA) {  // the first halo lane below: its row A is the row under this wave's last
      const uint32_t* rec = edge + ((wave_id + 2) * 2) * HL * 4 * W;
#pragma unroll
      for (int j = 0; j < W; ++j) blA[j] = rec[j];
    }
    if (wave_in_box) {
#pragma unroll
      for (int j = 0; j < W; ++j) {
        if (j >= w0 && j <= w1) {  // wave-uniform
          const uint32_t cm = (uint32_t)j + 1 == Wr ? last_mask : ((uint32_t)j < Wr ? 0xFFFFFFFFu : 0u);
          const uint32_t eA = blA[j], eB = blB[j];
          const uint32_t lA = j > 0 ? blA[j > 0 ? j - 1 : 0] : 0u, rA = j + 1 < W ? blA[j + 1 < W ? j + 1 : 0] : 0u;
          const uint32_t lB = j > 0 ? blB[j > 0 ? j - 1 : 0] : 0u, rB = j + 1 < W ? blB[j + 1 < W ? j + 1 : 0] : 0u;
          const uint32_t upA = fromLaneBelow(eB), dnB = fromLaneAbove(eA);
          if (inA) {
            const uint32_t nbr = __builtin_amdgcn_alignbit(eA, lA, 31) | __builtin_amdgcn_alignbit(rA, eA, 1) | upA | eB;
            const uint32_t fb = freew[rowA * Wr + j] & cm;
            const uint32_t touched = nbr & ~fb & ~eA & cm;
            storeCells(rowA, j, touched, N_obst);
            storeCells(rowA, j, ~eA & ~touched & cm, N_unreach);
          }
          if (inB) {
            const uint32_t nbr = __builtin_amdgcn_alignbit(eB, lB, 31) | __builtin_amdgcn_alignbit(rB, eB, 1) | eA | dnB;
            const uint32_t fb = freew[rowB * Wr + j] & cm;
            const uint32_t touched = nbr & ~fb & ~eB & cm;
            storeCells(rowB, j, touched, N_obst);
            storeCells(rowB, j, ~eB & ~touched & cm, N_unreach);
          }
        }
      }
    }
  }
  if (pl.bfs_trace && tid == 0) {
    pl.bfs_trace[(size_t)item * 8 + 6] = wall_clock64();
    pl.bfs_trace[(size_t)item * 8 + 1] = wall_clock64() | ((unsigned long long)level << 48);
  }
  if (tid == 0 && pl.bfs_grids == 3) pl.bfs_levels[(size_t)inst * 3 + which] = level;  // next cycle's dispatch order
}
__global__ __launch_bounds__(768, 3) void k_bfs_rows2(PlannerDev pl, uint32_t first, uint32_t count, uint32_t* next_item, const uint32_t* order, uint32_t* scratch) {
  __shared__ uint32_t s_item;
  const uint32_t total = count * pl.bfs_grids;  // 3 (DWA: path, goal, goal_front) or 2 (legacy TrajectoryPlanner)
  uint32_t* seedw = scratch + (size_t)blockIdx.x * pl.ny * ((pl.nx + 31) >> 5);  // this workgroup's seed bitmap
  for (;;) {
    if (threadIdx.x == 0) s_item = atomicAdd(next_item, 1u);
    __syncthreads();
    const uint32_t slot = s_item;
    if (slot >= total) break;  // (every workgroup gets here: the counter only grows)
    const uint32_t item = order ? order[slot] : slot;  // longest searches first
    const uint32_t g = item / count;
    bfsRows2Grid(pl, first + (item - g * count), (int)pl.bfs_grids - 1 - (int)g, item, seedw);
    __syncthreads();
  }
}
// words per row the one-row-per-lane sweep is instantiated for; 0 = not this map's kernel
static int bfs_rows_words(uint32_t nx, uint32_t ny) {
  if (NAVGPU_DEBUG_ENV("NAVGPU_DEBUG_BFS_NO_ROWS")) return 0;  // A/B timing (tool builds only)
  const uint32_t Wr = (nx + 31) / 32;
  if (bfs_rows_waves(ny) > 16) return 0;
  return Wr <= 7 ? 7 : (Wr <= 13 ? 13 : (Wr <= 20 ? 20 : 0));
}
bool bfs_rows_fits(uint32_t nx, uint32_t ny) { return bfs_rows_words(nx, ny) != 0; }
// searches resident per CU: as many as the wave slots and LDS hold, at most four
template <int W>
static void launch_bfs_rows_w(const PlannerDev& pl, uint32_t first, uint32_t count, hipStream_t s, const uint32_t* order) {
  const uint32_t nw = bfs_rows_waves(pl.ny);
  const size_t lds = bfs_rows_lds_words<W>(pl.nx, pl.ny) * 4;
  if (lds > 48 * 1024) hipFuncSetAttribute((const void*)k_bfs_rows<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const uint32_t per_cu = std::max<uint32_t>(1u, std::min<uint32_t>(std::min<uint32_t>(24u / nw, (uint32_t)((156u * 1024u) / (lds + 1024))), 4u));
  hipLaunchKernelGGL(k_bfs_rows<W>, dim3(std::min(count * pl.bfs_grids, per_cu * bfs_cu_count())), dim3(nw * 64), lds, s, pl, first, count, pl.bfs_next_item + 1, order);
}
bool launch_bfs_rows(const PlannerDev& pl, uint32_t first, uint32_t count, hipStream_t s, const uint32_t* order) {
  switch (bfs_rows_words(pl.nx, pl.ny)) {
    case 7: launch_bfs_rows_w<7>(pl, first, count, s, order); return true;
    case 13: launch_bfs_rows_w<13>(pl, first, count, s, order); return true;
    case 20: launch_bfs_rows_w<20>(pl, first, count, s, order); return true;
    default: return false;
  }
}
bool launch_bfs_rows2(const PlannerDev& pl, uint32_t first, uint32_t count, hipStream_t s, const uint32_t* order) {
  if (NAVGPU_DEBUG_ENV("NAVGPU_DEBUG_BFS_NO_ROWS")) return false;  // A/B timing (tool builds only): k_bfs_global
  if ((pl.nx + 31) / 32 > (uint32_t)kRows2Words || bfs_rows2_waves(pl.ny) > 12) return false;  // 12 waves of 168 registers
  // (the scratch holds 12 bitmaps per robot of the fleet; a workgroup uses one)
  const size_t lds2 = bfs_rows2_lds_words(pl.ny) * 4;
  if (lds2 > 48 * 1024) hipFuncSetAttribute((const void*)k_bfs_rows2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);
  hipLaunchKernelGGL(k_bfs_rows2, dim3(std::min(count * pl.bfs_grids, bfs_cu_count())), dim3(bfs_rows2_waves(pl.ny) * 64), lds2, s, pl, first, count, pl.bfs_next_item, order, pl.bfs_scratch);
  return true;
}

#ifdef NAVGPU_BFS_STATS
extern "C" int navgpu_debug_bfs_stats(unsigned long long* out16, int reset) {
  if (out16) hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_bfs_stats), sizeof(unsigned long long) * 16);
  if (reset) {
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(g_bfs_stats), z, sizeof(z));
  }
  return 0;
}
#endif

}  // namespace navgpu
