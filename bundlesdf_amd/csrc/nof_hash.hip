// Multiresolution hash-grid encode / backward for gfx950.
//
// What it computes is kernel_grid / kernel_grid_backward / kernel_input_backward of the reference
// (mycuda/torch_ngp_grid_encoder/gridencoder.cu:107-365); how it is laid out is MI355X-first:
//   * one lane = one (point, level); the block->level map is `level = blockIdx % L`, so with the
//     dispatcher's block b -> XCD b%8 placement every XCD touches only L/8 (or 1 of L<8) levels and
//     keeps those levels' table rows in ITS 4 MiB L2 (placement is a speed assumption only);
//   * features are stored level-major [L,B,2] so each XCD streams its own contiguous slab (8 B/lane);
//   * table rows are one aligned 8-byte gather (C == 2); 8 gathers are issued back to back per lane;
//   * backward recomputes indices/weights (no dy_dx tensor).  gfx950 executes fp32 atomics memory-side (~12 ns per op
//     on one line), and with one (sample, level) per lane the coarse levels funnel ~10^5 atomics into each of a few
//     dozen lines (measured: 13.6 ms per step at cfg2).  The scatter is therefore three launches:
//       k_hash_dx      input gradients per sample from gathers only (no atomics);
//       k_hash_bwd_lds levels whose whole slice fits in LDS accumulate there (ds_add_f32) and are flushed once per
//                      workgroup, skipping untouched entries;
//       k_hash_bwd_agg all other levels: lanes of a wave are consecutive samples of ONE ray, so lanes falling into the
//                      same cell form contiguous runs; each run is summed out of an LDS stage and emitted once by 16
//                      adjacent lanes (cfg2: 3..15 samples per run at the hashed levels).
#include <type_traits>
#include "nof_hash_dev.h"
#include "nof_reduce_dev.h"
#include "nof_pose_dev.h"
#ifndef NOF_AGG_PRIO
#define NOF_AGG_PRIO 1                                    // s_setprio by phase in the table scatter (A/B: profiles/r05_v_*): 1 = emission high, 2 = loads high
#endif
#pragma clang fp contract(off)

struct LevelList {
  int32_t n;
  int32_t level[NOF_MAX_LEVELS];
};

// One workgroup = 256 samples x the levels of ONE slot group: block b works on slots b % G, b % G + G, ... (G = min(8, L)), so
// with the dispatcher's block b -> XCD b % 8 placement every XCD keeps its L/8 levels' rows in its own L2 (xcd_level_slots deals a
// large hashed and a small dense level to each).  A lane reads its point ONCE and encodes those levels one after the other: at
// L = 16 that is 3 + 2 x 8 (or 4, see level_pairs) + 2 vector-memory instructions per lane instead of 2 x (3 + 8 + 1) -- the
// kernel is bound by the vector-memory INSTRUCTION rate, not by bytes.
__global__ __launch_bounds__(256) void k_hash_fwd(NofHashGrid g, LevelList slots, int G, const float* __restrict__ pts_w,
                                                   const float2* __restrict__ table, float2* __restrict__ feat,
                                                   int64_t B) {
  const int64_t b = (int64_t)(blockIdx.x / G) * 256 + threadIdx.x;
  if (b >= B) return;
  const float p[3] = {pts_w[b * 3], pts_w[b * 3 + 1], pts_w[b * 3 + 2]};
  for (int s = blockIdx.x % G; s < g.L; s += G) {                    // (unrolled by two: the same 80 us)
    const int level = slots.level[s];                                  // slot -> level: see xcd_level_slots()
    const HashLevel lv = load_level(g, level);
    const CellPos c = locate3(p, lv.scale);
    float2 acc;
    if (level_pairs(lv)) acc = encode_level<true>(lv, table, c);       // (uniform branch: the level is the workgroup's)
    else acc = encode_level<false>(lv, table, c);
    feat[(int64_t)level * B + b] = acc;
  }
}

// ---- backward -----------------------------------------------------------------------------------------
struct Scatter {
  uint32_t idx[8];
  float vx[8], vy[8];
  uint32_t key;                                                    // cell id (10 bits per axis) or ~0 when out of range
};

// Eikonal option (nerf_runner.py:734-738 with the normal of run_network_density, :1342-1345): the normal n = d sdf / d x is
// 0.5 * sum_levels sum_c g[l,c] * dy_dx[l,d,c] with g = d sdf / d feature (`geik`, level-major like dfeat) and dy_dx the finite
// differences of kernel_grid (gridencoder.cu:202-245), which are LINEAR in the table: corner k of level l receives
// g[l,c] * 0.5 * scale * sum_d dE/dn_d * (+-1 by bit d of k) * w'_{k,d} on top of the ordinary w_k * dfeat.  `dedn` [B,3] is
// dE/dn per sample (already carrying the loss weight); both pointers NULL = no eikonal term.
__device__ __forceinline__ Scatter make_scatter(const HashLevel& lv, const float* __restrict__ pts_w,
                                                const float2* __restrict__ dfeat, int level, int64_t b, int64_t B,
                                                const float2* __restrict__ geik = nullptr, const float* __restrict__ dedn = nullptr) {
  // straight-line code: a lane past the end or out of range (gridencoder.cu:276-281) reads sample B-1 / keeps its cell but
  // contributes exact zeros (gradient and fractions zeroed, so nothing non-finite can leak into the wave-level sums)
  Scatter sc;
  const int64_t bb = b < B ? b : B - 1;
  CellPos c = locate(pts_w, bb, lv.scale);
  const bool ok = b < B && !c.oob;
  float2 gr = dfeat[(int64_t)level * B + bb];
  if (!ok) { gr = make_float2(0.f, 0.f); c.f[0] = c.f[1] = c.f[2] = 0.f; }
  sc.key = ok ? (c.g[0] | (c.g[1] << 10) | (c.g[2] << 20)) : 0xFFFFFFFFu;
  // weights: ((1 * a_x) * a_y) * a_z like the reference's loop over d (gridencoder.cu:302-312), shared partial products
  const float ax[2] = {1.0f - c.f[0], c.f[0]}, ay[2] = {1.0f - c.f[1], c.f[1]}, az[2] = {1.0f - c.f[2], c.f[2]};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const float wk = (ax[k & 1] * ay[(k >> 1) & 1]) * az[k >> 2];
    sc.vx[k] = wk * gr.x;
    sc.vy[k] = wk * gr.y;
  }
  // rows: the same values as grid_index() per corner, with the shared terms computed once and ONE wrap test per lane
  if (lv.hashed) {
    const uint32_t hy0 = c.g[1] * 2654435761u, hy1 = hy0 + 2654435761u, hz0 = c.g[2] * 805459861u, hz1 = hz0 + 805459861u;
    const uint32_t yz[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
#pragma unroll
    for (int k = 0; k < 8; ++k) sc.idx[k] = (c.g[0] + (k & 1)) ^ yz[k >> 1];
    if ((lv.size & (lv.size - 1u)) == 0u) {
#pragma unroll
      for (int k = 0; k < 8; ++k) sc.idx[k] &= lv.size - 1u;
    } else {
      asm volatile("" ::: "memory");                                   // a real branch: do not compute eight divisions to discard them
#pragma unroll
      for (int k = 0; k < 8; ++k) sc.idx[k] %= lv.size;
    }
  } else {
    const uint32_t r1 = lv.res + 1u, r2 = r1 * r1;
    const uint32_t base = c.g[0] + c.g[1] * r1 + c.g[2] * r2;
#pragma unroll
    for (int k = 0; k < 8; ++k) sc.idx[k] = base + (k & 1) + ((k >> 1) & 1) * r1 + (k >> 2) * r2;
    if (sc.idx[7] >= lv.size) {                                        // the float32 resolution quirk of exact-power levels only
      asm volatile("" ::: "memory");
#pragma unroll
      for (int k = 0; k < 8; ++k) sc.idx[k] %= lv.size;
    }
  }
  if (geik != nullptr) {
    float2 ge = geik[(int64_t)level * B + bb];
    if (!ok) ge = make_float2(0.f, 0.f);
    const float dn[3] = {dedn[bb * 3], dedn[bb * 3 + 1], dedn[bb * 3 + 2]};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float ce = 0.0f;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        float wp = 1.0f;                                               // w'_{k,d}: the weights of the other two dimensions
#pragma unroll
        for (int e = 0; e < 3; ++e)
          if (e != d) wp *= (k & (1 << e)) ? c.f[e] : 1.0f - c.f[e];
        ce += ((k & (1 << d)) ? dn[d] : -dn[d]) * wp;
      }
      ce *= 0.5f * lv.scale;
      sc.vx[k] += ce * ge.x;
      sc.vy[k] += ce * ge.y;
    }
  }
  return sc;
}

// Segmented suffix-sum over RUNS of adjacent lanes with equal keys (lanes are consecutive samples; along one ray a cell's
// samples are adjacent, but nothing is assumed: equal keys that are not adjacent simply form separate runs).  After the call
// the FIRST lane of every run holds the run's total; returns whether this lane is such a leader.
__device__ __forceinline__ bool wave_merge_runs(Scatter& sc) {
  const int lane = threadIdx.x & 63;
  const uint32_t next = __shfl_down(sc.key, 1, 64);
  const uint32_t prev = __shfl_up(sc.key, 1, 64);
  int tail = (lane == 63 || next != sc.key) ? 1 : 0;                 // a run ends inside the range this lane has summed so far
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t_other = __shfl_down(tail, off, 64);
    const bool take = !tail && (lane + off < 64);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float tx = __shfl_down(sc.vx[k], off, 64);
      const float ty = __shfl_down(sc.vy[k], off, 64);
      if (take) { sc.vx[k] += tx; sc.vy[k] += ty; }
    }
    if (take) tail |= t_other;
  }
  return (lane == 0 || prev != sc.key) && sc.key != 0xFFFFFFFFu;
}


// Merge + emission for the levels that do not fit LDS.  Measured on MI355X (tools/atomic_probe.py): fp32 atomics retire at
// ~20.8 G line-requests/s chip-wide, lanes of ONE instruction that fall into the same 64-byte line merge into one request
// (x/y pair in adjacent lanes: 2x; consecutive entries: 8x), lanes that hit the SAME word do not merge, and a hot line
// serialises (3.8 G/s).  The kernel therefore cuts the number of requests on the CU before anything leaves it:
//   1. lanes of a wave are consecutive samples of ONE ray, so lanes in the same cell form contiguous RUNS (cfg2: 3..15
//      samples per run at the hashed levels).  The 16 products of a lane are summed over its run by a segmented inclusive
//      scan in REGISTERS: DPP row shifts (1, 2, 4, 8) inside the 16-lane rows, then row_bcast:15 / row_bcast:31 across rows,
//      one v_fmac per value and step with a 0/1 lane mask (`distance to the run's first lane >= reach of the step`).  The
//      last lane of a run ends up with the run's totals; the summation tree depends on lane positions only (deterministic).
//      Rounds 1-2 parked all 64 lanes' products in LDS and summed them there: that, not the atomics, was what the kernel
//      waited for;
//   2. only those last lanes write to LDS: 16 totals, 8 table rows and the cell id per run;
//   3. a run is emitted by 16 ADJACENT lanes -- [corner k][channel], k's bit 0 = the x neighbour, whose row is idx+1 for
//      dense levels and for even x of hashed levels (prime 1) -- i.e. 4..5 line requests per cell instead of 16;
//   4. consecutive cells along a ray share a face (4 of their 8 grid vertices), and same-word lanes do not merge, so a
//      vertex is emitted once per CHAIN of consecutive runs whose cells contain it: corner (m, k) hands its total on when the
//      vertex also belongs to the cell of run m+1, otherwise it collects runs m-1, m-2, ... while their cells contain the
//      vertex, and emits.  The test is geometric (cell ids), not a comparison of table rows: any partition of the
//      contributions into atomics is exact, so hash collisions need no special case -- they are simply separate atomics.
// Every LDS region of the stage is private to one wave and a wave's LDS instructions execute in order, so the phases are
// separated by compiler-only fences (no workgroup barrier), and the kernel is PERSISTENT per wave: a wave that has issued
// the atomics of one (64 samples, level) tile goes straight on to the next tile while the memory side retires them.
#define AGG_RUNS 32                                                    // runs per emission window (a tile has up to 64: two windows)
#define AGG_STRIDE 20                                                  // floats per run slot: 16-byte aligned, conflict-free b128
#define AGG_NONE 0xFFFFFFFFu
struct AggStage {
  float val[4][AGG_RUNS * AGG_STRIDE];                                // [wave][run * 20 + corner * 2 + channel]
  uint32_t row[4][AGG_RUNS * 8];                                      // [wave][run * 8 + corner]
  uint2 link[4][AGG_RUNS];                                            // [wave][run]: how the run's corners chain to its neighbours
  uint32_t key[4][AGG_RUNS];                                          // [wave][run]: cell id (only read by chains of >= 4 cells)
};

// One step of the scan: v += take * v[lane the DPP control points at], for the 16 values of a lane.  v_fmac_f32 takes its
// first factor through DPP; lanes without a source lane (row edge, rows outside row_mask) are not written, i.e. add nothing.
// Written as assembly because the compiler does not fold update_dpp into v_fmac (it emits v_mov_dpp + v_fmac, twice the
// instructions).  s_nop 4 covers the VALU-write -> DPP-read (2 wait states) and EXEC-write -> DPP (5) hazards the
// assembler cannot see across the statement boundary; inside the block every instruction reads a register written >= 16
// instructions earlier.
#define AGG_F(i, ctrl) "v_fmac_f32_dpp %" #i ", %" #i ", %16 " ctrl "\n\t"
#define AGG_SCAN_STEP(sc, take, ctrl)                                                                                    \
  asm volatile("s_nop 4\n\t" AGG_F(0, ctrl) AGG_F(1, ctrl) AGG_F(2, ctrl) AGG_F(3, ctrl) AGG_F(4, ctrl) AGG_F(5, ctrl)     \
               AGG_F(6, ctrl) AGG_F(7, ctrl) AGG_F(8, ctrl) AGG_F(9, ctrl) AGG_F(10, ctrl) AGG_F(11, ctrl) AGG_F(12, ctrl) \
               AGG_F(13, ctrl) AGG_F(14, ctrl) AGG_F(15, ctrl)                                                            \
               : "+v"(sc.vx[0]), "+v"(sc.vy[0]), "+v"(sc.vx[1]), "+v"(sc.vy[1]), "+v"(sc.vx[2]), "+v"(sc.vy[2]),          \
                 "+v"(sc.vx[3]), "+v"(sc.vy[3]), "+v"(sc.vx[4]), "+v"(sc.vy[4]), "+v"(sc.vx[5]), "+v"(sc.vy[5]),          \
                 "+v"(sc.vx[6]), "+v"(sc.vy[6]), "+v"(sc.vx[7]), "+v"(sc.vy[7])                                            \
               : "v"(take))

// Which of the 8 corners of cell `key` are also vertices of cell `other`, and by how much their corner number shifts there
// (corner k of `key` is corner k - shift of `other`).  Per axis the cells differ by delta: 0 -> every corner, +1 -> the
// corners with that axis bit set, -1 -> those with it clear, anything else -> none.
__device__ __forceinline__ uint32_t shared_corners(uint32_t key, uint32_t other, int& shift) {
  const uint32_t dx = (other & 1023u) - (key & 1023u) + 1u, dy = ((other >> 10) & 1023u) - ((key >> 10) & 1023u) + 1u,
                 dz = ((other >> 20) & 1023u) - ((key >> 20) & 1023u) + 1u;   // delta + 1 in 0..2 when the cells touch
  shift = (int)(dx + 2u * dy + 4u * dz) - 7;
  const uint32_t m = (0xAAFF55u >> (8u * dx)) & (0xCCFF33u >> (8u * dy)) & (0xF0FF0Fu >> (8u * dz)) & 0xFFu;
  return (other != AGG_NONE && max(dx, max(dy, dz)) < 3u) ? m : 0u;
}

// does the cell `key` contain grid vertex (vx, vy, vz)?  kk = the vertex's corner number in that cell
__device__ __forceinline__ bool cell_has(uint32_t key, uint32_t vx, uint32_t vy, uint32_t vz, int& kk) {
  const uint32_t dx = vx - (key & 1023u), dy = vy - ((key >> 10) & 1023u), dz = vz - ((key >> 20) & 1023u);
  kk = (int)(dx | (dy << 1) | (dz << 2));
  return key != AGG_NONE && (dx | dy | dz) < 2u;
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// (the body of k_hash_bwd_agg as a device function of (workgroup index, workgroup count): the stand-alone kernel passes blockIdx /
//  gridDim, the merged launch k_hash_bwd_agg_dx its own numbering)
template <bool EIK>
__device__ __forceinline__ void hash_bwd_agg_block(const NofHashGrid& g, const LevelList& ll, const float* __restrict__ pts_w,
                                                   const float2* __restrict__ dfeat, float* __restrict__ grad_table,
                                                   int64_t B, const float2* __restrict__ geik, const float* __restrict__ dedn,
                                                   const uint32_t* __restrict__ tile_list, int trim, const uint32_t block_id,
                                                   const uint32_t grid_n) {
  __shared__ __attribute__((aligned(16))) AggStage st;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* val = st.val[w];
  uint32_t* row = st.row[w];
  uint2* link = st.link[w];
  uint32_t* keys = st.key[w];
  const int e = lane & 15, q = lane >> 4, k = e >> 1, ch = e & 1;
  const unsigned long long le = (2ull << lane) - 1ull;                // lanes 0..lane
  // (64 samples, level) tiles, level fastest; tile = block * 4 + wave, then + gridDim * 4 per round -- kept as (sample block,
  // level slot) with a carry instead of dividing every round
  // With a work list (NofTileList) a "sample block" is a PAIR of listed 32-sample tiles, lanes 0-31 / 32-63: the run / chain logic
  // below only ever compares cell ids of neighbouring lanes, so two tiles from different rays in one wave are just a place where
  // a run ends (or, by coincidence, continues -- same cell, same vertices, same sum).  An odd list ends in a tile past the batch.
  const uint32_t n_items = tile_list ? ((uint32_t)__builtin_amdgcn_readfirstlane((int)tile_list[0]) + 1u) / 2u : (uint32_t)((B + 63) / 64);
  // How many of the launch's persistent workgroups take part is decided HERE, from the list's length (known on the device only):
  // with most tiles listed the scatter wants 4 workgroups per CU in flight, with half of them or fewer -- a training batch: 37 % of
  // the tiles once the field has settled, 50 % in the first steps -- 3 (A/B on one box, whole 501-step round 0.425-0.434 vs
  // 0.436-0.441 ms/step, settled 0.398-0.406 vs 0.410-0.414; every tile listed 0.662 vs 0.616: profiles/r04_s_scatter_wgs.txt).
  // The host launches the larger grid (default workgroup count only, `trim` != 0); the last quarter leaves at once when the list is short.
  uint32_t n_blocks = grid_n;
  if (trim != 0 && tile_list != nullptr && 4u * (uint32_t)__builtin_amdgcn_readfirstlane((int)tile_list[0]) < 3u * (uint32_t)__builtin_amdgcn_readfirstlane((int)tile_list[1]))
    n_blocks = (grid_n * 3u) / 4u;
  if (block_id >= n_blocks) return;
  const uint32_t n_lv = (uint32_t)ll.n, n_sb = n_items, n_waves = n_blocks * 4u;
  const uint32_t t0 = block_id * 4u + (uint32_t)w, dq = n_waves / n_lv, dr = n_waves % n_lv;
  uint32_t sb = t0 / n_lv, slot_l = t0 % n_lv;
  for (; sb < n_sb; sb += dq, slot_l += dr, sb += slot_l >= n_lv ? 1u : 0u, slot_l -= slot_l >= n_lv ? n_lv : 0u) {
    const int level = ll.level[slot_l];
    const int64_t b = tile_list ? (int64_t)tile_list[4 + 2 * sb + (lane >> 5)] * 32 + (lane & 31) : (int64_t)sb * 64 + lane;
    const HashLevel lv = load_level(g, level);
    // Ray-samples whose loss gradient is EXACTLY zero (background rays, free-space samples whose loss has saturated: two
    // thirds of a cfg2 batch once the field has settled, tools/zero_grad_probe.py) add nothing to the table: a tile of 64 such
    // samples is skipped as a whole, and a vertex total of exactly 0 is not emitted -- the same sums, fewer atomics.
    // (Without a list only: a listed tile has a non-zero gradient by construction.)
    if constexpr (!EIK) {
      if (tile_list == nullptr) {
        const float2 g0 = dfeat[(int64_t)level * B + (b < B ? b : B - 1)];
        if (__ballot(b < B && (g0.x != 0.0f || g0.y != 0.0f)) == 0ull) continue;
      }
    }
#if NOF_AGG_PRIO == 2
    __builtin_amdgcn_s_setprio(3);                                    // (A/B: the loads of the next item first)
#elif NOF_AGG_PRIO == 1
    __builtin_amdgcn_s_setprio(0);
#endif
    Scatter sc = make_scatter(lv, pts_w, dfeat, level, b, B, EIK ? geik : nullptr, EIK ? dedn : nullptr);
    const bool valid = sc.key != AGG_NONE;
    // neighbours' cells through ds_bpermute (the DPP wavefront shifts do not cross the 16-lane rows on gfx950)
    const uint32_t kpv = (uint32_t)__builtin_amdgcn_ds_bpermute((lane - 1) << 2, (int)sc.key);
    const uint32_t knx = (uint32_t)__builtin_amdgcn_ds_bpermute((lane + 1) << 2, (int)sc.key);
    const uint32_t kprev = lane > 0 ? kpv : AGG_NONE, knext = lane < 63 ? knx : AGG_NONE;
    const bool head = valid && kprev != sc.key;                       // lanes are consecutive samples of one ray
    const bool tail = valid && knext != sc.key;
    const unsigned long long heads = __ballot(head);
    const int nl = __popcll(heads);
    if (nl == 0) continue;
    // 1. run totals in registers
    const int dist = lane - (63 - __builtin_clzll((heads & le) | 1ull));   // distance to the run's first lane (valid lanes)
    const float t1 = dist >= 1 ? 1.0f : 0.0f, t2 = dist >= 2 ? 1.0f : 0.0f, t4 = dist >= 4 ? 1.0f : 0.0f,
                t8 = dist >= 8 ? 1.0f : 0.0f, t15 = dist > (lane & 15) ? 1.0f : 0.0f, t31 = dist >= lane - 31 ? 1.0f : 0.0f;
#ifdef NOF_AGG_PERTURB
    // Test build only (libnof_hash_perturb.so, tests/test_gpu_ops.py): two dozen extra values stay live across the hand-written DPP
    // block below, which moves every register the block uses.  The block's hazard padding (the s_nop, the >= 16-instruction
    // spacing) must not depend on where the allocator put things: round 2 saw inline-assembly VALU code in the MLP kernels
    // miscompute after an unrelated change moved the allocation.
    float pert[24];
#pragma unroll
    for (int i = 0; i < 24; ++i) pert[i] = sc.vx[i & 7] * (float)(i + 1) + (float)lane;
#define AGG_PIN_PERT()                                                                                                  \
  _Pragma("unroll") for (int i = 0; i < 24; ++i) asm volatile("" : "+v"(pert[i]))
#else
#define AGG_PIN_PERT()
#endif
    AGG_SCAN_STEP(sc, t1, "row_shr:1 row_mask:0xf bank_mask:0xf");
    AGG_PIN_PERT();
    AGG_SCAN_STEP(sc, t2, "row_shr:2 row_mask:0xf bank_mask:0xf");
    AGG_PIN_PERT();
    if (__ballot(valid && dist >= 4)) {
      AGG_SCAN_STEP(sc, t4, "row_shr:4 row_mask:0xf bank_mask:0xf");
      if (__ballot(valid && dist >= 8)) AGG_SCAN_STEP(sc, t8, "row_shr:8 row_mask:0xf bank_mask:0xf");
    }
    AGG_SCAN_STEP(sc, t15, "row_bcast:15 row_mask:0xa bank_mask:0xf");   // lane 15 / 47 into rows 1 / 3
    AGG_SCAN_STEP(sc, t31, "row_bcast:31 row_mask:0xc bank_mask:0xf");   // lane 31 into rows 2, 3
    AGG_PIN_PERT();
#ifdef NOF_AGG_PERTURB
    {
      float ps = 0.0f;
#pragma unroll
      for (int i = 0; i < 24; ++i) ps += pert[i];
      if (ps == 1.2345678e30f) sc.vx[0] += ps;                        // keeps the values alive; never true
    }
#endif
    // 2. how a run chains to the runs next to it (lane-adjacent runs only: an out-of-range sample breaks the chain).  The
    //    cell before the run's first lane and the one before that come through ds_bpermute.
    const int hd = lane - dist;                                       // first lane of this lane's run
    const uint32_t pk = (uint32_t)__builtin_amdgcn_ds_bpermute((hd - 1) << 2, (int)(sc.key));
    const int pd = __builtin_amdgcn_ds_bpermute((hd - 1) << 2, dist);
    const uint32_t kp1 = hd > 0 ? pk : AGG_NONE;
    const int hd1 = hd - 1 - pd;                                      // first lane of the previous run
    const uint32_t ppk = (uint32_t)__builtin_amdgcn_ds_bpermute((hd1 - 1) << 2, (int)(sc.key));
    const uint32_t kp2 = (kp1 != AGG_NONE && hd1 > 0) ? ppk : AGG_NONE;
    int sh0, sh1, sh2;
    uint32_t hand_on = shared_corners(sc.key, knext, sh0);            // corners the next run's chain takes
    uint32_t take1 = shared_corners(sc.key, kp1, sh1);                // corners that collect the previous run ...
    uint32_t take2 = take1 & shared_corners(sc.key, kp2, sh2);        // ... and the one before it
    const int slot = __popcll(heads & le) - 1;
    float* __restrict__ gt = grad_table + 2 * (size_t)lv.offset;
#if NOF_AGG_PRIO == 1
    __builtin_amdgcn_s_setprio(3);                                    // (A/B: the emission -- LDS stage, atomics -- first)
#elif NOF_AGG_PRIO == 2
    __builtin_amdgcn_s_setprio(0);
#endif
    for (int base = 0; base < nl; base += AGG_RUNS) {
      const int cnt = nl - base < AGG_RUNS ? nl - base : AGG_RUNS;
      // 3. the runs of this window (a window edge breaks the chains, too)
      if (tail && slot >= base && slot < base + AGG_RUNS) {
        const int s = slot - base;
        float4* v4 = reinterpret_cast<float4*>(&val[s * AGG_STRIDE]);
#pragma unroll
        for (int c = 0; c < 4; ++c) v4[c] = make_float4(sc.vx[2 * c], sc.vy[2 * c], sc.vx[2 * c + 1], sc.vy[2 * c + 1]);
        uint4* r4 = reinterpret_cast<uint4*>(&row[s * 8]);
        r4[0] = make_uint4(sc.idx[0], sc.idx[1], sc.idx[2], sc.idx[3]);
        r4[1] = make_uint4(sc.idx[4], sc.idx[5], sc.idx[6], sc.idx[7]);
        const uint32_t ho = s == cnt - 1 ? 0u : hand_on, c1 = s == 0 ? 0u : take1, c2 = s <= 1 ? 0u : take2;
        link[s] = make_uint2(ho | (c1 << 8) | (c2 << 16), (uint32_t)(sh1 + 16) | ((uint32_t)(sh2 + 16) << 8));
        keys[s] = sc.key;
      }
      wave_lds_sync();
      // 4. lane (q, e) emits element e = (corner, channel) of runs q, q+4, ...; 16 adjacent lanes = one run
      for (int m = q; m < cnt; m += 4) {
        const uint2 lk = link[m];
        const uint32_t r = row[m * 8 + k];
        float acc = val[m * AGG_STRIDE + e];
        if ((lk.x >> k) & 1u) continue;                               // the next run's chain takes it
        if ((lk.x >> (8 + k)) & 1u) {
          acc += val[(m - 1) * AGG_STRIDE + 2 * (k - (int)(lk.y & 255u) + 16) + ch];
          if ((lk.x >> (16 + k)) & 1u) {
            acc += val[(m - 2) * AGG_STRIDE + 2 * (k - (int)(lk.y >> 8) + 16) + ch];
            const uint32_t km = keys[m];                              // four or more cells around one vertex: rare
            const uint32_t vx = (km & 1023u) + (k & 1), vy = ((km >> 10) & 1023u) + ((k >> 1) & 1), vz = ((km >> 20) & 1023u) + (k >> 2);
            for (int j = m - 3; j >= 0; --j) {
              int kk;
              if (!cell_has(keys[j], vx, vy, vz, kk)) break;
              if (!((link[j].x >> kk) & 1u)) break;                   // (j, kk) did not hand on: it emitted by itself
              acc += val[j * AGG_STRIDE + 2 * kk + ch];
            }
          }
        }
        if (acc != 0.0f) atomicAdd(&gt[2 * (size_t)r + ch], acc);    // gridencoder.cu:317-333 (fp32 atomics)
      }
      wave_lds_sync();                                                // the next window / tile overwrites the stage
    }
  }
}

template <bool EIK>
__global__ __launch_bounds__(256) void k_hash_bwd_agg(NofHashGrid g, LevelList ll, const float* __restrict__ pts_w,
                                                       const float2* __restrict__ dfeat, float* __restrict__ grad_table,
                                                       int64_t B, const float2* __restrict__ geik, const float* __restrict__ dedn,
                                                       const uint32_t* __restrict__ tile_list, int trim) {
  hash_bwd_agg_block<EIK>(g, ll, pts_w, dfeat, grad_table, B, geik, dedn, tile_list, trim, blockIdx.x, gridDim.x);
}

// The MLP backward's row reduction as a passenger of this launch (nof_hash_encode_bwd_parts_reduce): `partials` != NULL puts
// ncb * RED_RSPLIT workgroups of reduce_partials_block in FRONT of the launch's own.  Both are 1024-thread workgroups; the reduction
// needs the MLP backward only, which is long done here, and the step's tail loses a launch and the gap in front of it.
struct RedArgs {
  const float* partials;
  float* out;
  int32_t* flags;
  int n_rows, n_cols, ncb;
};
// ... and the per-ray pose / frame-feature gradient rows (nof_pose_grad_accum) as a second passenger (round 6: it needs dL/dx, which
// the step's previous launch -- the merged scatter + dL/dx -- has finished; behind the reduction's workgroups, 16 rays per workgroup)
struct PoseArgs {
  NofPoseAccum a;                                                      // a.batch == NULL: no passenger
  int nwg;
};

// levels whose slice fits LDS: accumulate privately, flush once
__global__ __launch_bounds__(1024) void k_hash_bwd_lds(NofHashGrid g, LevelList ll, int chunks, const float* __restrict__ pts_w,
                                                        const float2* __restrict__ dfeat, float* __restrict__ grad_table,
                                                        int64_t B, const float2* __restrict__ geik, const float* __restrict__ dedn,
                                                        const uint32_t* __restrict__ tile_list, RedArgs red, PoseArgs pose) {
  extern __shared__ __attribute__((aligned(16))) float acc[];
  int bid = (int)blockIdx.x;
  if (red.partials != nullptr) {                                       // (workgroup-uniform)
    const int nred = red.ncb * RED_RSPLIT;
    if (bid < nred) {
      reduce_partials_block(red.partials, red.n_rows, red.n_cols, red.out, red.flags, bid % red.ncb, bid / red.ncb);
      return;
    }
    bid -= nred;
  }
  if (pose.a.batch != nullptr) {                                       // (workgroup-uniform)
    if (bid < pose.nwg) {
      const int64_t r = (int64_t)bid * 16 + (threadIdx.x >> 6);
      if (r < pose.a.R)
        pose_grad_accum_ray(pose.a.dpts, pose.a.dview, pose.a.batch, pose.a.z_vals, pose.a.c2w, pose.a.tf, pose.a.ff, pose.a.sh_degree,
                            pose.a.R, pose.a.S, pose.a.g_ray, pose.a.frame_slots, r, (int)(threadIdx.x & 63));
      return;
    }
    bid -= pose.nwg;
  }
  const int level = ll.level[bid % ll.n];
  const int chunk = bid / ll.n;
  const HashLevel lv = load_level(g, level);
  const int n2 = 2 * (int)lv.size;
  for (int e = threadIdx.x; e < n2; e += blockDim.x) acc[e] = 0.0f;
  __syncthreads();
  // wave items = 64 samples: 64 consecutive samples of the batch, or a pair of listed tiles (see k_hash_bwd_agg); a chunk is a
  // contiguous range of items, its 16 waves stride over it
  const int64_t n_items = tile_list ? ((int64_t)tile_list[0] + 1) / 2 : (B + 63) / 64;
  const int64_t per = (n_items + chunks - 1) / chunks;
  const int64_t lo = (int64_t)chunk * per;
  const int64_t hi = lo + per < n_items ? lo + per : n_items;
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  for (int64_t it = lo + w; it < hi; it += nw) {
    const int64_t b = tile_list ? (int64_t)tile_list[4 + 2 * it + (lane >> 5)] * 32 + (lane & 31) : it * 64 + lane;
    if (geik == nullptr && tile_list == nullptr) {                     // all 64 gradients exactly zero: nothing to add
      const float2 g0 = dfeat[(int64_t)level * B + (b < B ? b : B - 1)];
      if (__ballot(b < B && (g0.x != 0.0f || g0.y != 0.0f)) == 0ull) continue;
    }
    Scatter sc = make_scatter(lv, pts_w, dfeat, level, b < B ? b : B, B, geik, dedn);
    if (wave_merge_runs(sc)) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (sc.vx[k] != 0.0f) atomicAdd(&acc[2 * sc.idx[k]], sc.vx[k]);
        if (sc.vy[k] != 0.0f) atomicAdd(&acc[2 * sc.idx[k] + 1], sc.vy[k]);
      }
    }
  }
  __syncthreads();
  float* __restrict__ gt = grad_table + 2 * (size_t)lv.offset;
  for (int e = threadIdx.x; e < n2; e += blockDim.x) {
    const float v = acc[e];
    if (v != 0.0f) atomicAdd(&gt[e], v);
  }
}

// dL/dpts_w per sample: gathers only (kernel_input_backward + the dy_dx part of kernel_grid, gridencoder.cu:202-245,340-365).
// One level's contribution: dy/dx01[gd] = scale * sum over the 4 corner pairs along gd of w' * (f_right - f_left), dotted with the
// level's feature gradient; PAIRS as in encode_level (x neighbour in the same 16-byte load).
template <bool PAIRS, bool EIK>
__device__ __forceinline__ void dx_level(const HashLevel& lv, const float2* __restrict__ table, const CellPos& c, float2 gr,
                                         float2 ge, const float (&dn)[3], float (&dx)[3], float (&dxe)[3]) {
  const float2* __restrict__ tl = table + lv.offset;
  uint32_t idx[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) idx[k] = grid_index(lv, c.g[0] + (k & 1), c.g[1] + ((k >> 1) & 1), c.g[2] + ((k >> 2) & 1));
  float2 v[8];
  if constexpr (PAIRS) {
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      const RowPair t = *reinterpret_cast<const RowPair*>(tl + idx[k]);
      v[k] = make_float2(t.x, t.y);
      v[k + 1] = make_float2(t.z, t.w);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = tl[idx[k]];
  }
#pragma unroll
  for (int gd = 0; gd < 3; ++gd) {
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      if (k & (1 << gd)) continue;
      float wk = lv.scale;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (d == gd) continue;
        wk *= (k & (1 << d)) ? c.f[d] : (1.0f - c.f[d]);
      }
      const float2 l = v[k], r = v[k | (1 << gd)];
      s += wk * ((r.x - l.x) * gr.x + (r.y - l.y) * gr.y);
    }
    dx[gd] += s;
  }
  if constexpr (EIK) {
    // eikonal option: dE/dx through the normal's own dependence on x -- the mixed second derivatives of the trilinear blend
    // (d n_d / d x_e = 0.25 * scale^2 * sum_t w_t (F[d1,e1,t] - F[d1,e0,t] - F[d0,e1,t] + F[d0,e0,t]), F = g . corner features)
    float Fk[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) Fk[k] = ge.x * v[k].x + ge.y * v[k].y;
#pragma unroll
    for (int e = 0; e < 3; ++e)
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (d == e) continue;
        const int t = 3 - d - e;                                       // the third dimension
        float m = 0.0f;
#pragma unroll
        for (int bt = 0; bt < 2; ++bt) {
          const int base = bt << t;
          const float wt = bt ? c.f[t] : 1.0f - c.f[t];
          m += wt * (((Fk[base | (1 << d) | (1 << e)] - Fk[base | (1 << d)]) - Fk[base | (1 << e)]) + Fk[base]);
        }
        dxe[e] += dn[d] * 0.25f * lv.scale * lv.scale * m;
      }
  }
}

// lane = sample, all levels.  The levels come as two lists (pair-loadable dense levels, the others) so that each loop is free of
// branches and the compiler can keep the gathers of two levels in flight (the former single loop tested `gradient == 0` and the
// pair condition per level: one level's latency at a time).  A zero gradient or an out-of-range point contributes exact zeros
// through its factors instead of through a branch.
template <bool EIK>
__device__ __forceinline__ void hash_dx_block(const NofHashGrid& g, const LevelList& pairs, const LevelList& singles,
                                              const float* __restrict__ pts_w, const float2* __restrict__ table,
                                              const float2* __restrict__ dfeat, float* __restrict__ dpts, int64_t B,
                                              const float2* __restrict__ geik, const float* __restrict__ dedn,
                                              const uint8_t* __restrict__ tile_flags, const uint32_t block_id) {
  const int64_t b = (int64_t)block_id * 256 + threadIdx.x;
  if (b >= B) return;
  // work list given: the samples of an unflagged tile have dL/dx = 0 (and their dfeat was not written: do not read it)
  if (tile_flags != nullptr && tile_flags[b >> 5] == 0) {
    dpts[b * 3] = 0.0f; dpts[b * 3 + 1] = 0.0f; dpts[b * 3 + 2] = 0.0f;
    return;
  }
  const float p[3] = {pts_w[b * 3], pts_w[b * 3 + 1], pts_w[b * 3 + 2]};
  float dx[3] = {0.f, 0.f, 0.f}, dxe[3] = {0.f, 0.f, 0.f}, dn[3] = {0.f, 0.f, 0.f};
  if constexpr (EIK) { dn[0] = dedn[b * 3]; dn[1] = dedn[b * 3 + 1]; dn[2] = dedn[b * 3 + 2]; }
  auto one = [&](int level, auto PAIRS) {
    const HashLevel lv = load_level(g, level);
    CellPos c = locate3(p, lv.scale);
    float2 gr = dfeat[(int64_t)level * B + b];
    float2 ge = make_float2(0.f, 0.f);
    if constexpr (EIK) ge = geik[(int64_t)level * B + b];
    if (c.oob) {                                                       // out of range for every level (gridencoder.cu:131): zero factors,
      gr = ge = make_float2(0.f, 0.f);                                 // an in-range cell for the (discarded) gathers
      c.g[0] = c.g[1] = c.g[2] = 0u;
      c.f[0] = c.f[1] = c.f[2] = 0.0f;
    }
    dx_level<decltype(PAIRS)::value, EIK>(lv, table, c, gr, ge, dn, dx, dxe);
  };
#pragma unroll 2
  for (int i = 0; i < pairs.n; ++i) one(pairs.level[i], std::true_type());
#pragma unroll 2
  for (int i = 0; i < singles.n; ++i) one(singles.level[i], std::false_type());
#pragma unroll
  for (int gd = 0; gd < 3; ++gd) dpts[b * 3 + gd] = dx[gd] * 0.5f + dxe[gd];    // d x01 / d x = 1/2 (grid.py:160)
}

template <bool EIK>
__global__ __launch_bounds__(256) void k_hash_dx(NofHashGrid g, LevelList pairs, LevelList singles, const float* __restrict__ pts_w,
                                                  const float2* __restrict__ table, const float2* __restrict__ dfeat,
                                                  float* __restrict__ dpts, int64_t B, const float2* __restrict__ geik,
                                                  const float* __restrict__ dedn, const uint8_t* __restrict__ tile_flags) {
  hash_dx_block<EIK>(g, pairs, singles, pts_w, table, dfeat, dpts, B, geik, dedn, tile_flags, blockIdx.x);
}

// The table scatter of the large levels and dL/dx as two ROLES of one launch (round 6; NOF_HASH_BWD_MERGE_INPUT): the training step
// ran them beside each other on two streams, paying a fork and a join (12-15 us each on this runtime) for it.  Every `period`-th
// workgroup of the launch is one of the scatter's persistent workgroups -- so that they are resident from the start --, the others
// take 256 samples of dL/dx each.  Same device functions as the two stand-alone kernels: the same results.
template <bool EIK>
__global__ __launch_bounds__(256) void k_hash_bwd_agg_dx(NofHashGrid g, LevelList ll, LevelList pairs, LevelList singles,
                                                          const float* __restrict__ pts_w, const float2* __restrict__ table,
                                                          const float2* __restrict__ dfeat, float* __restrict__ grad_table,
                                                          float* __restrict__ dpts, int64_t B, const float2* __restrict__ geik,
                                                          const float* __restrict__ dedn, const uint32_t* __restrict__ tile_list,
                                                          const uint8_t* __restrict__ tile_flags, int trim, uint32_t n_agg,
                                                          uint32_t period, int32_t* __restrict__ batch_flags) {
  // (NOF_HASH_BWD_NEW_BATCH: the overflow mark of the PREVIOUS step -- bit 2, read by that step's optimiser launch, long done --
  //  becomes the sticky bit 3 here instead of in the batch's ray marcher; this step's own mark is raised by the launch after this one)
  if (batch_flags != nullptr && blockIdx.x == gridDim.x - 1u && threadIdx.x == 0) {
    if (atomicAnd(&batch_flags[0], ~4) & 4) atomicOr(&batch_flags[0], 8);
  }
  const uint32_t q = blockIdx.x / period, r = blockIdx.x - q * period;
  if (r == 0u && q < n_agg) {                                          // (workgroup-uniform)
    hash_bwd_agg_block<EIK>(g, ll, pts_w, dfeat, grad_table, B, geik, dedn, tile_list, trim, q, n_agg);
    return;
  }
  const uint32_t before = q + 1u < n_agg ? q + 1u : n_agg;            // scatter workgroups among the launch's workgroups 0 .. blockIdx
  hash_dx_block<EIK>(g, pairs, singles, pts_w, table, dfeat, dpts, B, geik, dedn, tile_flags, blockIdx.x - before);
}

__global__ __launch_bounds__(256) void k_hash_indices(NofHashGrid g, const float* __restrict__ pts_w,
                                                       int32_t* __restrict__ out, int64_t B) {
  const int level = blockIdx.x % g.L;
  const int64_t b = (int64_t)(blockIdx.x / g.L) * 256 + threadIdx.x;
  if (b >= B) return;
  const HashLevel lv = load_level(g, level);
  const CellPos c = locate(pts_w, b, lv.scale);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    uint32_t p[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) p[d] = c.g[d] + ((k >> d) & 1);
    out[(b * g.L + level) * 8 + k] = c.oob ? -1 : (int32_t)(grid_index(lv, p[0], p[1], p[2]) + lv.offset);
  }
}

static int check_grid(const NofHashGrid* g) {
  if (g == nullptr) return nof_set_error(-1, "hash grid descriptor is NULL");
  if (g->C != 2) return nof_set_error(-1, "hash grid: only C == 2 features per level is supported (got %d)", g->C);
  if (g->L < 1 || g->L > NOF_MAX_LEVELS) return nof_set_error(-1, "hash grid: L=%d out of range", g->L);
  return 0;
}

// Block b runs on XCD b % 8 and handles the level in slot b % L, so slots with equal (slot % 8) share one XCD's 4 MiB L2.
// Levels are dealt to the slots largest first in snake order (0..7, 7..0, ...): every XCD gets one large hashed level and
// one small dense level instead of levels l and l + 8 (at cfg2 that paired the two 4 MB levels 7/15 ... on one L2).
// Placement is a speed assumption only.
static LevelList xcd_level_slots(const NofHashGrid* g) {
  int order[NOF_MAX_LEVELS];
  for (int l = 0; l < g->L; ++l) order[l] = l;
  for (int i = 1; i < g->L; ++i)                                       // insertion sort by size, descending (stable)
    for (int j = i; j > 0 && g->size[order[j]] > g->size[order[j - 1]]; --j) { const int t = order[j]; order[j] = order[j - 1]; order[j - 1] = t; }
  LevelList s;
  s.n = g->L;
  for (int i = 0; i < g->L; ++i) {
    const int grp = i / 8, r = (grp & 1) ? 7 - (i % 8) : (i % 8);
    int slot = grp * 8 + r;
    if (slot >= g->L) slot = i;                                        // ragged last group: keep it simple
    s.level[slot] = order[i];
  }
  if (g->L % 8 != 0)                                                   // ragged: fall back to the identity permutation
    for (int i = 0; i < g->L; ++i) s.level[i] = i;
  return s;
}

extern "C" int nof_hash_encode_fwd(const NofHashGrid* g, const float* pts_w, const float* table, float* feat,
                                    int64_t B, void* stream) {
  if (int e = check_grid(g)) return e;
  NOF_ARG(pts_w && table && feat && B >= 0);
  if (B == 0) return 0;
  const int G = g->L < 8 ? g->L : 8;
  const int64_t blocks = nof_div_up(B, 256) * G;
  NOF_ARG(blocks < (1ll << 31));
  hipLaunchKernelGGL(k_hash_fwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *g, xcd_level_slots(g), G, pts_w,
                     (const float2*)table, (float2*)feat, B);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_hash_encode_bwd(const NofHashGrid* g, const float* pts_w, const float* table, const float* dfeat,
                                    float* grad_table, float* dpts, int64_t B, void* stream) {
  if (int e = check_grid(g)) return e;
  return nof_hash_encode_bwd_levels(g, pts_w, table, dfeat, grad_table, dpts, 0, g->L, B, stream);
}

// The table gradient of levels [level_lo, level_hi) only (+ the input gradient over ALL levels when dpts is given).  The
// data-parallel step calls it twice, fine levels first, so that the all-reduce of their (large) slice of the gradient
// buffer runs beside the scatter of the coarse levels and the pose kernels.
extern "C" int nof_hash_encode_bwd_levels(const NofHashGrid* g, const float* pts_w, const float* table, const float* dfeat,
                                           float* grad_table, float* dpts, int32_t level_lo, int32_t level_hi, int64_t B,
                                           void* stream) {
  return nof_hash_encode_bwd_eik(g, pts_w, table, dfeat, nullptr, nullptr, grad_table, dpts, level_lo, level_hi, B, stream);
}

// The same with the eikonal term's contributions (geik [L,B,2] = d sdf / d feature, dedn [B,3] = dE/dn, both from nof_eikonal;
// both NULL = plain backward): table gradient through the finite differences, input gradient through the mixed second
// derivatives.  One scatter pass serves both terms.
extern "C" int nof_hash_encode_bwd_eik(const NofHashGrid* g, const float* pts_w, const float* table, const float* dfeat,
                                        const float* geik_, const float* dedn, float* grad_table, float* dpts, int32_t level_lo,
                                        int32_t level_hi, int64_t B, void* stream) {
  return nof_hash_encode_bwd_parts(g, pts_w, table, dfeat, geik_, dedn, grad_table, dpts, level_lo, level_hi, nullptr,
                                   NOF_HASH_BWD_ALL, 0, B, stream);
}

// The full-featured entry point.  The backward is three independent kernels (they share read-only inputs and write disjoint
// outputs / disjoint levels of grad_table); `parts` selects which of them THIS call launches, all on `stream`, one after the
// other.  Running them beside each other is the caller's business -- it owns the streams: the training step puts
// NOF_HASH_BWD_TABLE_BIG on its main stream and the other two on its side stream (field.py); the library creates no stream, no
// event, reads no environment.
//   NOF_HASH_BWD_TABLE_BIG    levels whose slice exceeds 48 KiB: run-merged global atomics (k_hash_bwd_agg), persistent waves,
//                             `wgs_per_cu` workgroups per CU (0 = default: 4)
//   NOF_HASH_BWD_TABLE_SMALL  the others: accumulated in LDS, flushed once per workgroup (k_hash_bwd_lds)
//   NOF_HASH_BWD_INPUT        dL/dpts over ALL levels (k_hash_dx; needs dpts)
// tile_list (NofTileList or NULL): only the listed tiles are read and scattered; dpts of unlisted tiles is written as 0.
static __global__ __launch_bounds__(64) void k_pose_accum_alone(NofPoseAccum a) {
  pose_grad_accum_ray(a.dpts, a.dview, a.batch, a.z_vals, a.c2w, a.tf, a.ff, a.sh_degree, a.R, a.S, a.g_ray, a.frame_slots,
                      (int64_t)blockIdx.x, (int)threadIdx.x);
}

static int hash_bwd_parts(const NofHashGrid* g, const float* pts_w, const float* table, const float* dfeat,
                          const float* geik_, const float* dedn, float* grad_table, float* dpts, int32_t level_lo,
                          int32_t level_hi, const void* tile_list, int32_t parts, int32_t wgs_per_cu, int64_t B,
                          RedArgs red, void* stream, const NofPoseAccum* pose_accum = nullptr) {
  if (int e = check_grid(g)) return e;
  NOF_ARG(pts_w && table && dfeat && grad_table && B >= 0 && level_lo >= 0 && level_lo <= level_hi && level_hi <= g->L);
  NOF_ARG((geik_ == nullptr) == (dedn == nullptr));
  NOF_ARG(parts >= 0 && parts <= (NOF_HASH_BWD_ALL | NOF_HASH_BWD_MERGE_INPUT | NOF_HASH_BWD_NEW_BATCH) && wgs_per_cu >= 0 && wgs_per_cu <= 16);
  int32_t* batch_flags = (parts & NOF_HASH_BWD_NEW_BATCH) ? red.flags : nullptr;
  NOF_ARG(!(parts & NOF_HASH_BWD_NEW_BATCH) || (batch_flags != nullptr && (parts & NOF_HASH_BWD_MERGE_INPUT)));
  parts &= ~NOF_HASH_BWD_NEW_BATCH;
  NOF_ARG(tile_list == nullptr || geik_ == nullptr);                   // the eikonal term has a gradient at every sample
  const float2* geik = (const float2*)geik_;
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  // split the levels: slices of <= 48 KiB are accumulated in LDS (their few hundred rows would be hot lines for global
  // atomics), the others go through wave-merged global atomics.  Measured at cfg2: a 128 KiB cap (levels 0-2 in LDS) makes
  // the LDS kernel's 1024-thread / 125 KiB blocks wait for empty CUs beside k_hash_bwd_agg (501 us overlapped vs 85 us
  // alone) and the whole call 75 us slower than with level 0 alone in LDS; no LDS level at all is 230 us slower.
  const size_t lds_cap = 48 * 1024;
  LevelList small, big;
  small.n = big.n = 0;
  size_t lds_need = 0;
  for (int l = level_lo; l < level_hi; ++l) {
    const size_t bytes = (size_t)g->size[l] * 8;
    if (bytes <= lds_cap) { small.level[small.n++] = l; if (bytes > lds_need) lds_need = bytes; }
    else big.level[big.n++] = l;
  }
  const uint32_t* tl = (const uint32_t*)tile_list;
  LevelList pairs, singles;                                            // dL/dx: the same test as level_pairs() on the device
  pairs.n = singles.n = 0;
  for (int l = 0; l < g->L; ++l) {
    const uint64_t r1 = (uint64_t)g->resolution[l] + 1;
    if (!g->hashed[l] && r1 <= 1024 && r1 * r1 * r1 <= g->size[l]) pairs.level[pairs.n++] = l;
    else singles.level[singles.n++] = l;
  }
  // NOF_HASH_BWD_MERGE_INPUT: the large levels' scatter and dL/dx as two roles of ONE launch (k_hash_bwd_agg_dx)
  if ((parts & NOF_HASH_BWD_MERGE_INPUT) && (parts & NOF_HASH_BWD_TABLE_BIG) && (parts & NOF_HASH_BWD_INPUT) && dpts && big.n > 0) {
    int64_t n_agg = (int64_t)(wgs_per_cu > 0 ? wgs_per_cu : 4) * nof_cu_count();
    if (tl == nullptr) {
      const int64_t need = nof_div_up(nof_div_up(B, 64) * big.n, 4);
      if (n_agg > need) n_agg = need;
    }
    const int64_t n_dx = nof_div_up(B, 256), total = n_agg + n_dx;
    // the scatter's persistent workgroups FIRST (consecutive workgroups go round the eight XCDs: resident everywhere from the start),
    // dL/dx's behind them.  (Measured: every 4th workgroup a scatter workgroup put all of them on two XCDs -- the call 236 instead
    // of 121 us; every 3rd: 140, the late ones start late.)
    const int64_t period = 1;
    const int trim = wgs_per_cu == 0 ? 1 : 0;
    if (geik != nullptr)
      hipLaunchKernelGGL(k_hash_bwd_agg_dx<true>, dim3((unsigned)total), dim3(256), 0, st, *g, big, pairs, singles, pts_w,
                         (const float2*)table, (const float2*)dfeat, grad_table, dpts, B, geik, dedn, tl, nof_tile_flags(tile_list, B), trim,
                         (uint32_t)n_agg, (uint32_t)period, batch_flags);
    else
      hipLaunchKernelGGL(k_hash_bwd_agg_dx<false>, dim3((unsigned)total), dim3(256), 0, st, *g, big, pairs, singles, pts_w,
                         (const float2*)table, (const float2*)dfeat, grad_table, dpts, B, geik, dedn, tl, nof_tile_flags(tile_list, B), trim,
                         (uint32_t)n_agg, (uint32_t)period, batch_flags);
    NOF_LAUNCH_OK();
    batch_flags = nullptr;
    parts &= ~(NOF_HASH_BWD_TABLE_BIG | NOF_HASH_BWD_INPUT);
  }
  NOF_ARG(batch_flags == nullptr);                                      // (NEW_BATCH rides in the merged launch only)
  if ((parts & NOF_HASH_BWD_TABLE_BIG) && big.n > 0) {
    // persistent waves.  With every sample contributing (6.9 M line requests at cfg2) the kernel is bound by the atomic rate of
    // the memory side: two workgroups per CU saturate it and more only take L2 bandwidth from the kernels beside it
    // (whole call 1 -> 646 us, 2 -> 439, 3 -> 474, 4 -> 525, 8 -> 537).  With the zero-gradient tiles gone (two thirds of a settled
    // cfg2 batch: 1.3 M requests) it wants more waves in flight: 4 are launched by default, of which the kernel
    // itself keeps 3 when fewer than three quarters of the tiles are listed (see there).
    int64_t blocks = (int64_t)(wgs_per_cu > 0 ? wgs_per_cu : 4) * nof_cu_count();
    if (tl == nullptr) {                                                // (the size of a list is only known on the device)
      const int64_t need = nof_div_up(nof_div_up(B, 64) * big.n, 4);
      if (blocks > need) blocks = need;
    }
    const int trim = wgs_per_cu == 0 ? 1 : 0;                          // (an explicit workgroup count is taken literally)
    if (geik != nullptr)
      hipLaunchKernelGGL(k_hash_bwd_agg<true>, dim3((unsigned)blocks), dim3(256), 0, st, *g, big, pts_w, (const float2*)dfeat,
                         grad_table, B, geik, dedn, tl, trim);
    else
      hipLaunchKernelGGL(k_hash_bwd_agg<false>, dim3((unsigned)blocks), dim3(256), 0, st, *g, big, pts_w, (const float2*)dfeat,
                         grad_table, B, geik, dedn, tl, trim);
    NOF_LAUNCH_OK();
  }
  if ((parts & NOF_HASH_BWD_INPUT) && dpts) {
    if (geik != nullptr)
      hipLaunchKernelGGL(k_hash_dx<true>, dim3((unsigned)nof_div_up(B, 256)), dim3(256), 0, st, *g, pairs, singles, pts_w,
                         (const float2*)table, (const float2*)dfeat, dpts, B, geik, dedn, nof_tile_flags(tile_list, B));
    else
      hipLaunchKernelGGL(k_hash_dx<false>, dim3((unsigned)nof_div_up(B, 256)), dim3(256), 0, st, *g, pairs, singles, pts_w,
                         (const float2*)table, (const float2*)dfeat, dpts, B, geik, dedn, nof_tile_flags(tile_list, B));
    NOF_LAUNCH_OK();
  }
  if ((parts & NOF_HASH_BWD_TABLE_SMALL) && small.n > 0) {
    // Two workgroups per CU (78 of the 160 KB of LDS): over the work list a workgroup's 16 waves then have about one item each
    // and the kernel is its fixed parts (zero 39 KB, one item's latency chain, flush).  64 workgroups: 33 us and the step 15 us
    // longer (dense backward: 60 us); 1024 / 2048 smaller workgroups: the same as 512 (A/B on one box)
    const int chunks = 2 * nof_cu_count();
    if (lds_need > 64 * 1024)
      NOF_HIP(hipFuncSetAttribute((const void*)k_hash_bwd_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_need));
    const unsigned nred = red.partials != nullptr ? (unsigned)(red.ncb * RED_RSPLIT) : 0u;
    PoseArgs pa;
    pa.a = NofPoseAccum{};
    pa.nwg = 0;
    if (pose_accum != nullptr && pose_accum->R > 0) { pa.a = *pose_accum; pa.nwg = (int)nof_div_up(pose_accum->R, 16); }
    hipLaunchKernelGGL(k_hash_bwd_lds, dim3((unsigned)(chunks * small.n) + nred + (unsigned)pa.nwg), dim3(1024), lds_need, st, *g, small,
                       chunks, pts_w, (const float2*)dfeat, grad_table, B, geik, dedn, tl, red, pa);
    NOF_LAUNCH_OK();
    red.partials = nullptr;                                            // done
    pose_accum = nullptr;
  }
  if (pose_accum != nullptr && pose_accum->R > 0) {                    // no LDS-level launch to ride in: on its own (one wave per ray)
    PoseArgs pa;
    pa.a = *pose_accum;
    hipLaunchKernelGGL(k_pose_accum_alone, dim3((unsigned)pa.a.R), dim3(64), 0, st, pa.a);
    NOF_LAUNCH_OK();
  }
  if (red.partials != nullptr)                                         // no LDS-level launch to ride in: on its own
    return reduce_partials_launch(red.partials, red.n_rows, red.n_cols, red.out, red.flags, stream);
  return 0;
}

extern "C" int nof_hash_encode_bwd_parts(const NofHashGrid* g, const float* pts_w, const float* table, const float* dfeat,
                                          const float* geik_, const float* dedn, float* grad_table, float* dpts, int32_t level_lo,
                                          int32_t level_hi, const void* tile_list, int32_t parts, int32_t wgs_per_cu, int64_t B,
                                          void* stream) {
  return hash_bwd_parts(g, pts_w, table, dfeat, geik_, dedn, grad_table, dpts, level_lo, level_hi, tile_list, parts, wgs_per_cu, B,
                        RedArgs{nullptr, nullptr, nullptr, 0, 0, 0}, stream);
}

// nof_hash_encode_bwd_parts followed by nof_reduce_partials(partials, n_rows, n_cols, grad_mlp, flags) -- same results -- with the
// reduction inside the launch of the LDS-accumulated levels when `parts` has one (else as its own launch).
extern "C" int nof_hash_encode_bwd_parts_reduce(const NofHashGrid* g, const float* pts_w, const float* table, const float* dfeat,
                                                 const float* geik_, const float* dedn, float* grad_table, float* dpts,
                                                 int32_t level_lo, int32_t level_hi, const void* tile_list, int32_t parts,
                                                 int32_t wgs_per_cu, int64_t B, const float* partials, int32_t n_rows, int32_t n_cols,
                                                 float* grad_mlp, int32_t* flags, void* stream) {
  NOF_ARG(partials && grad_mlp && n_rows >= 0 && n_cols >= 0);
  RedArgs red{partials, grad_mlp, flags, n_rows, n_cols, (int)nof_div_up(n_cols, 32)};
  if (n_rows == 0 || n_cols == 0) red.partials = nullptr;
  if (B == 0 && red.partials != nullptr) return reduce_partials_launch(partials, n_rows, n_cols, grad_mlp, flags, stream);
  return hash_bwd_parts(g, pts_w, table, dfeat, geik_, dedn, grad_table, dpts, level_lo, level_hi, tile_list, parts, wgs_per_cu, B, red,
                        stream);
}

/* The hash side of the training step's backward in TWO launches (round 6): nof_hash_encode_bwd_parts_reduce with
 * NOF_HASH_BWD_MERGE_INPUT (the large levels' scatter + dL/dpts as roles of one launch), and -- `pose` != NULL -- the per-ray pose /
 * frame-feature gradient rows of nof_pose_grad_accum (same arguments, as a struct) as a passenger of the LDS levels' launch beside
 * the MLP backward's row reduction.  Same results as the separate calls. */
extern "C" int nof_hash_encode_bwd_step(const NofHashGrid* g, const float* pts_w, const float* table, const float* dfeat,
                                         const float* geik_, const float* dedn, float* grad_table, float* dpts, int32_t level_lo,
                                         int32_t level_hi, const void* tile_list, int32_t parts, int32_t wgs_per_cu, int64_t B,
                                         const float* partials, int32_t n_rows, int32_t n_cols, float* grad_mlp, int32_t* flags,
                                         const NofPoseAccum* pose, void* stream) {
  RedArgs red{partials, grad_mlp, flags, n_rows, n_cols, (int)nof_div_up(n_cols, 32)};
  if (partials == nullptr || n_rows <= 0 || n_cols <= 0) red.partials = nullptr;
  if (pose != nullptr) {
    NOF_ARG(pose->batch && pose->z_vals && pose->c2w && pose->tf && pose->g_ray && pose->R >= 0 && pose->S >= 1 && pose->ff >= 0 &&
            pose->ff <= NOF_VIEW_COLS && pose->sh_degree >= 1 && pose->sh_degree <= 3 && (pose->frame_slots == nullptr || pose->dview != nullptr));
  }
  if (B == 0) {
    if (red.partials != nullptr) return reduce_partials_launch(partials, n_rows, n_cols, grad_mlp, flags, stream);
    return 0;
  }
  return hash_bwd_parts(g, pts_w, table, dfeat, geik_, dedn, grad_table, dpts, level_lo, level_hi, tile_list, parts, wgs_per_cu, B, red,
                        stream, pose);
}

extern "C" int nof_hash_corner_indices(const NofHashGrid* g, const float* pts_w, int32_t* idx, int64_t B, void* stream) {
  if (int e = check_grid(g)) return e;
  NOF_ARG(pts_w && idx && B >= 0);
  if (B == 0) return 0;
  const int64_t blocks = nof_div_up(B, 256) * g->L;
  hipLaunchKernelGGL(k_hash_indices, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *g, pts_w, idx, B);
  NOF_LAUNCH_OK();
  return 0;
}
