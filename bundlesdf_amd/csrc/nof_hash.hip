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
#include "nof_hash_dev.h"
#pragma clang fp contract(off)

struct LevelList {
  int32_t n;
  int32_t level[NOF_MAX_LEVELS];
};

__global__ __launch_bounds__(256) void k_hash_fwd(NofHashGrid g, LevelList slots, const float* __restrict__ pts_w,
                                                   const float2* __restrict__ table, float2* __restrict__ feat,
                                                   int64_t B) {
  const int level = slots.level[blockIdx.x % g.L];                   // slot -> level: see xcd_level_slots()
  const int64_t b = (int64_t)(blockIdx.x / g.L) * 256 + threadIdx.x;
  if (b >= B) return;
  const HashLevel lv = load_level(g, level);
  const CellPos c = locate(pts_w, b, lv.scale);
  const float2 acc = encode_level(lv, table, c);
  feat[(int64_t)level * B + b] = acc;
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
  Scatter sc;
  sc.key = 0xFFFFFFFFu;
#pragma unroll
  for (int k = 0; k < 8; ++k) { sc.idx[k] = 0; sc.vx[k] = 0.f; sc.vy[k] = 0.f; }
  if (b >= B) return sc;
  const CellPos c = locate(pts_w, b, lv.scale);
  if (c.oob) return sc;                                              // gridencoder.cu:276-281
  const float2 gr = dfeat[(int64_t)level * B + b];
  sc.key = c.g[0] | (c.g[1] << 10) | (c.g[2] << 20);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float wk = 1.0f;
    uint32_t p[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (k & (1 << d)) { wk *= c.f[d]; p[d] = c.g[d] + 1u; }
      else              { wk *= 1.0f - c.f[d]; p[d] = c.g[d]; }
    }
    sc.idx[k] = grid_index(lv, p[0], p[1], p[2]);
    sc.vx[k] = wk * gr.x;
    sc.vy[k] = wk * gr.y;
  }
  if (geik != nullptr) {
    const float2 ge = geik[(int64_t)level * B + b];
    const float dn[3] = {dedn[b * 3], dedn[b * 3 + 1], dedn[b * 3 + 2]};
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
// (x/y pair in adjacent lanes: 2x; consecutive entries: 8x) and a hot line serialises (3.8 G/s).  Two consequences:
//   * runs of adjacent lanes in the same cell are merged before anything leaves the CU.  Two merges were measured and
//     dropped: a segmented shuffle scan (16 values x 6 steps of ds_bpermute + select + add: ~45% of the kernel's
//     instructions and all of its LDS round-trip latency) and ds_add_f32 into one LDS slot per run (LDS float atomics
//     retire ~1 lane per 4.5 cycles on gfx950, 92 us per level regardless of the run structure).  Instead every lane parks
//     its 16 products in LDS with plain 16-byte stores and the run is summed by the lanes that emit it;
//   * a run is emitted by 16 ADJACENT lanes -- [corner k][channel] with k's bit 0 = the x neighbour, whose row is
//     idx+1 for dense levels and for even x of hashed levels (prime 1) -- i.e. 4..8 line requests per cell instead of 16
//     (different words of one line merge into one request from anywhere in the instruction, variants 7/8 of the probe).
//     Lane (q, e) of a wave walks the runs q, q+4, ... and adds up element e of their lanes in lane order (deterministic).
#define AGG_STRIDE 20                                                  // floats per lane slot: 16-byte aligned, conflict-free b128
#define AGG_ROWS 12                                                    // words per run in the row list (8 used), same reason
struct AggStage {
  float val[4][64 * AGG_STRIDE];                                      // [wave][lane * 20 + corner * 2 + channel]
  uint32_t row[4][64 * AGG_ROWS];                                     // [wave][run * 12 + corner]
  uint32_t span[4][64];                                               // [wave][run] = first lane | (length << 8)
};

struct Rows8 {
  uint32_t r[8];
};
__device__ __forceinline__ Rows8 load_rows(const uint32_t* p) {
  const uint4 a = *reinterpret_cast<const uint4*>(p), b = *reinterpret_cast<const uint4*>(p + 4);
  Rows8 o;
  o.r[0] = a.x; o.r[1] = a.y; o.r[2] = a.z; o.r[3] = a.w; o.r[4] = b.x; o.r[5] = b.y; o.r[6] = b.z; o.r[7] = b.w;
  return o;
}
__device__ __forceinline__ uint32_t match_mask(const Rows8& rs, uint32_t row) {
  uint32_t m = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) m |= (rs.r[k] == row ? 1u : 0u) << k;
  return m;
}

// A third measured fact shapes the last step: lanes of one instruction that hit the SAME ADDRESS are not merged (4 lanes on
// one word cost 4 requests, tools/atomic_probe.py variant 6), and consecutive cells along a ray share a face, i.e. 4 of
// their 8 table rows.  So rows are de-duplicated across the runs of a wave before emission: the first run of a chain of
// consecutive runs containing a row owns it and adds up the chain's contributions (exact for any collision pattern: a
// run that also lists the row earlier in itself, or whose predecessor lists it, is not an owner).
// Every LDS region of the stage is private to one wave and a wave's LDS instructions execute in order, so the phases are
// separated by compiler-only fences (no workgroup barrier), and the kernel is PERSISTENT per wave: a wave that has issued
// the atomics of one (64 samples, level) tile goes straight on to the next tile while the memory side retires them.  With
// one tile per workgroup the waves piled up in the emission phase holding their LDS (4 workgroups/CU), compute and atomics
// ran back to back (575 us = 256 us with plain stores + ~330 us of atomic time) instead of overlapped.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__global__ __launch_bounds__(256) void k_hash_bwd_agg(NofHashGrid g, LevelList ll, const float* __restrict__ pts_w,
                                                       const float2* __restrict__ dfeat, float* __restrict__ grad_table,
                                                       int64_t B, const float2* __restrict__ geik, const float* __restrict__ dedn) {
  __shared__ __attribute__((aligned(16))) AggStage st;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t n_tiles = ((B + 63) / 64) * ll.n;                     // (64 samples, level) tiles, level fastest
  const int64_t n_waves = (int64_t)gridDim.x * 4;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + w; tile < n_tiles; tile += n_waves) {
  const int level = ll.level[tile % ll.n];
  const int64_t b = (tile / ll.n) * 64 + lane;
  const HashLevel lv = load_level(g, level);
  const Scatter sc = make_scatter(lv, pts_w, dfeat, level, b, B, geik, dedn);
  const bool valid = sc.key != 0xFFFFFFFFu;
  const uint32_t prev = __shfl_up(sc.key, 1, 64);
  const bool head = valid && (lane == 0 || prev != sc.key);           // lanes are consecutive samples of one ray
  const unsigned long long heads = __ballot(head);
  const unsigned long long breaks = heads | ~__ballot(valid);         // a run ends before the next head or invalid lane
  const int nl = __popcll(heads);
  float* val = st.val[w];
  uint32_t* row = st.row[w];
  float4* v4 = reinterpret_cast<float4*>(&val[lane * AGG_STRIDE]);
#pragma unroll
  for (int k = 0; k < 4; ++k) v4[k] = make_float4(sc.vx[2 * k], sc.vy[2 * k], sc.vx[2 * k + 1], sc.vy[2 * k + 1]);
  if (head) {
    const int slot = __popcll(heads & ((1ull << lane) - 1ull));
    const unsigned long long after = lane == 63 ? 0ull : (breaks >> (lane + 1));
    const int len = after ? __builtin_ctzll(after) + 1 : 64 - lane;
    st.span[w][slot] = (uint32_t)lane | ((uint32_t)len << 8);
    uint4* r4 = reinterpret_cast<uint4*>(&row[slot * AGG_ROWS]);
    r4[0] = make_uint4(sc.idx[0], sc.idx[1], sc.idx[2], sc.idx[3]);
    r4[1] = make_uint4(sc.idx[4], sc.idx[5], sc.idx[6], sc.idx[7]);
  }
  wave_lds_sync();
  // phase 1: lane (q, e) sums element e over the lanes of runs q, q+4, ... (lane order); the total replaces the head's slot
  const int e = lane & 15, q = lane >> 4;
  for (int m = q; m < nl; m += 4) {
    const uint32_t sp = st.span[w][m];
    float* src = &val[(sp & 0xFF) * AGG_STRIDE + e];
    const int len = (int)(sp >> 8);
    float acc = src[0];
    for (int i = 1; i < len; ++i) acc += src[i * AGG_STRIDE];
    if (len > 1) src[0] = acc;
  }
  wave_lds_sync();
  // phase 2: row owners collect their chain and emit
  float* __restrict__ gt = grad_table + 2 * (size_t)lv.offset;
  const int k = e >> 1, ch = e & 1;
  for (int m = q; m < nl; m += 4) {
    const Rows8 mine = load_rows(&row[m * AGG_ROWS]);
    const uint32_t r = row[m * AGG_ROWS + k];
    uint32_t mm = match_mask(mine, r);
    bool owner = (mm & ((1u << k) - 1u)) == 0u;                       // no earlier corner of this run has the same row
    if (m > 0 && match_mask(load_rows(&row[(m - 1) * AGG_ROWS]), r) != 0u) owner = false;
    if (!owner) continue;
    float acc = 0.0f;
    int j = m;
    while (true) {
      const float* sv = &val[(st.span[w][j] & 0xFF) * AGG_STRIDE + ch];
      while (mm) {
        const int kk = __builtin_ctz(mm);
        mm &= mm - 1u;
        acc += sv[2 * kk];
      }
      if (++j >= nl) break;
      mm = match_mask(load_rows(&row[j * AGG_ROWS]), r);
      if (mm == 0u) break;
    }
    atomicAdd(&gt[2 * (size_t)r + ch], acc);                          // gridencoder.cu:317-333 (fp32 atomics)
  }
  wave_lds_sync();                                                    // the next tile overwrites the stage
  }
}

// levels whose slice fits LDS: accumulate privately, flush once
__global__ __launch_bounds__(1024) void k_hash_bwd_lds(NofHashGrid g, LevelList ll, int chunks, const float* __restrict__ pts_w,
                                                        const float2* __restrict__ dfeat, float* __restrict__ grad_table,
                                                        int64_t B, const float2* __restrict__ geik, const float* __restrict__ dedn) {
  extern __shared__ __attribute__((aligned(16))) float acc[];
  const int level = ll.level[blockIdx.x % ll.n];
  const int chunk = blockIdx.x / ll.n;
  const HashLevel lv = load_level(g, level);
  const int n2 = 2 * (int)lv.size;
  for (int e = threadIdx.x; e < n2; e += blockDim.x) acc[e] = 0.0f;
  __syncthreads();
  const int64_t per = ((B + chunks - 1) / chunks + 63) / 64 * 64;     // whole waves per chunk
  const int64_t lo = (int64_t)chunk * per;
  const int64_t hi = lo + per < B ? lo + per : B;
  for (int64_t base = lo; base < hi; base += blockDim.x) {
    const int64_t b = base + threadIdx.x;
    Scatter sc = make_scatter(lv, pts_w, dfeat, level, b < hi ? b : B, B, geik, dedn);
    if (wave_merge_runs(sc)) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        atomicAdd(&acc[2 * sc.idx[k]], sc.vx[k]);
        atomicAdd(&acc[2 * sc.idx[k] + 1], sc.vy[k]);
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

// dL/dpts_w per sample: gathers only (kernel_input_backward + the dy_dx part of kernel_grid, gridencoder.cu:202-245,340-365)
__global__ __launch_bounds__(256) void k_hash_dx(NofHashGrid g, const float* __restrict__ pts_w,
                                                  const float2* __restrict__ table, const float2* __restrict__ dfeat,
                                                  float* __restrict__ dpts, int64_t B, const float2* __restrict__ geik,
                                                  const float* __restrict__ dedn) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  float dx[3] = {0.f, 0.f, 0.f};
  // eikonal option: dE/dx through the normal's own dependence on x -- the mixed second derivatives of the trilinear blend
  // (d n_d / d x_e = 0.25 * scale^2 * sum_t w_t (F[d1,e1,t] - F[d1,e0,t] - F[d0,e1,t] + F[d0,e0,t]), F = g . corner features)
  float dxe[3] = {0.f, 0.f, 0.f};
  float dn[3] = {0.f, 0.f, 0.f};
  if (geik != nullptr) { dn[0] = dedn[b * 3]; dn[1] = dedn[b * 3 + 1]; dn[2] = dedn[b * 3 + 2]; }
  for (int level = 0; level < g.L; ++level) {
    const HashLevel lv = load_level(g, level);
    const CellPos c = locate(pts_w, b, lv.scale);
    if (c.oob) break;                                                  // the point is out of range for every level
    const float2 gr = dfeat[(int64_t)level * B + b];
    const float2* __restrict__ tl = table + lv.offset;
    float2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t p0 = c.g[0] + (k & 1), p1 = c.g[1] + ((k >> 1) & 1), p2 = c.g[2] + ((k >> 2) & 1);
      v[k] = tl[grid_index(lv, p0, p1, p2)];
    }
    // dy/dx01[gd] = scale * sum_{other two dims} w' * (f_right - f_left)   (gridencoder.cu:202-245)
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
    if (geik != nullptr) {
      const float2 ge = geik[(int64_t)level * B + b];
      float Fk[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) Fk[k] = ge.x * v[k].x + ge.y * v[k].y;
#pragma unroll
      for (int e = 0; e < 3; ++e)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
          if (d == e) continue;
          const int t = 3 - d - e;                                     // the third dimension
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
#pragma unroll
  for (int gd = 0; gd < 3; ++gd) dpts[b * 3 + gd] = dx[gd] * 0.5f + dxe[gd];    // d x01 / d x = 1/2 (grid.py:160)
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
  const int64_t blocks = nof_div_up(B, 256) * g->L;
  NOF_ARG(blocks < (1ll << 31));
  hipLaunchKernelGGL(k_hash_fwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *g, xcd_level_slots(g), pts_w,
                     (const float2*)table, (float2*)feat, B);
  NOF_LAUNCH_OK();
  return 0;
}

// Fork/join helper: the three backward kernels are independent of each other (they only share read-only inputs and write
// disjoint outputs / disjoint levels of grad_table).  k_hash_bwd_agg is bound by memory-side atomic throughput and leaves the
// CUs mostly idle, so the gather-bound k_hash_dx and the LDS-bound k_hash_bwd_lds run beside it on an internal stream.
// Event record/wait pairs are legal during stream capture (they become graph edges).
struct SideStream {
  hipStream_t stream = nullptr;
  hipEvent_t fork = nullptr, join = nullptr;
  int device = -1;
};
static thread_local SideStream g_side;

static int side_stream(SideStream** out) {
  int dev = 0;
  NOF_HIP(hipGetDevice(&dev));
  if (g_side.stream == nullptr || g_side.device != dev) {
    if (g_side.stream != nullptr) {                                    // device changed: the old stream and events belong to the other device
      (void)hipEventDestroy(g_side.fork);
      (void)hipEventDestroy(g_side.join);
      (void)hipStreamDestroy(g_side.stream);
      g_side = SideStream();
    }
    NOF_HIP(hipStreamCreateWithFlags(&g_side.stream, hipStreamNonBlocking));   // (stream priority made no measurable difference)
    NOF_HIP(hipEventCreateWithFlags(&g_side.fork, hipEventDisableTiming));
    NOF_HIP(hipEventCreateWithFlags(&g_side.join, hipEventDisableTiming));
    g_side.device = dev;
  }
  *out = &g_side;
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
  if (int e = check_grid(g)) return e;
  NOF_ARG(pts_w && table && dfeat && grad_table && B >= 0 && level_lo >= 0 && level_lo <= level_hi && level_hi <= g->L);
  NOF_ARG((geik_ == nullptr) == (dedn == nullptr));
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
  SideStream* side = nullptr;
  const bool fork = big.n > 0 && (dpts != nullptr || small.n > 0);
  if (big.n == 0 && small.n == 0 && dpts == nullptr) return 0;
  hipStream_t s2 = st;
  if (fork) {
    if (int e = side_stream(&side)) return e;
    s2 = side->stream;
    NOF_HIP(hipEventRecord(side->fork, st));
    NOF_HIP(hipStreamWaitEvent(s2, side->fork, 0));
  }
  if (big.n > 0) {
    int64_t blocks = nof_div_up(nof_div_up(B, 64) * big.n, 4);        // persistent: at most 4 workgroups per CU (LDS)
    const int64_t cap = 4ll * nof_cu_count();
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_hash_bwd_agg, dim3((unsigned)blocks), dim3(256), 0, st, *g, big, pts_w, (const float2*)dfeat,
                       grad_table, B, geik, dedn);
    NOF_LAUNCH_OK();
  }
  if (dpts) {
    hipLaunchKernelGGL(k_hash_dx, dim3((unsigned)nof_div_up(B, 256)), dim3(256), 0, s2, *g, pts_w, (const float2*)table,
                       (const float2*)dfeat, dpts, B, geik, dedn);
    NOF_LAUNCH_OK();
  }
  if (small.n > 0) {
    const int chunks = 64;
    if (lds_need > 64 * 1024)
      NOF_HIP(hipFuncSetAttribute((const void*)k_hash_bwd_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_need));
    hipLaunchKernelGGL(k_hash_bwd_lds, dim3((unsigned)(chunks * small.n)), dim3(1024), lds_need, s2, *g, small, chunks, pts_w,
                       (const float2*)dfeat, grad_table, B, geik, dedn);
    NOF_LAUNCH_OK();
  }
  if (fork) {
    NOF_HIP(hipEventRecord(side->join, s2));
    NOF_HIP(hipStreamWaitEvent(st, side->join, 0));
  }
  return 0;
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
