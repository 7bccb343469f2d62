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
//                      same cell form contiguous runs; a segmented shuffle-sum merges each run and only its first lane
//                      issues the 16 atomics (16x fewer at the coarsest hashed level, 1x at the finest).
#include "nof_common.h"
#pragma clang fp contract(off)

struct HashLevel {
  float scale;
  uint32_t res, offset, size, hashed;
};

__device__ __forceinline__ uint32_t grid_index(const HashLevel& lv, uint32_t x, uint32_t y, uint32_t z) {
  uint32_t index;
  if (lv.hashed) {
    index = (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u);      // fast_hash, gridencoder.cu:47-62
  } else {
    const uint32_t r1 = lv.res + 1u;                                // align_corners == false
    index = x + y * r1 + z * r1 * r1;                               // gridencoder.cu:70-74
  }
  return index % lv.size;
}

struct CellPos {
  uint32_t g[3];
  float f[3];
  bool oob;
};

__device__ __forceinline__ CellPos locate(const float* __restrict__ pts_w, int64_t b, float scale) {
  CellPos c;
  c.oob = false;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float x01 = (pts_w[b * 3 + d] + 1.0f) * 0.5f;            // grid.py:160
    if (x01 < 0.0f || x01 > 1.0f) c.oob = true;                     // gridencoder.cu:131
    const float pos = x01 * scale + 0.5f;                           // gridencoder.cu:164
    const float fl = floorf(pos);
    c.g[d] = (uint32_t)fl;
    c.f[d] = pos - fl;
  }
  return c;
}

__device__ __forceinline__ HashLevel load_level(const NofHashGrid& g, int l) {
  HashLevel lv;
  lv.scale = g.scale[l]; lv.res = g.resolution[l]; lv.offset = g.offset[l]; lv.size = g.size[l]; lv.hashed = g.hashed[l];
  return lv;
}

__global__ __launch_bounds__(256) void k_hash_fwd(NofHashGrid g, const float* __restrict__ pts_w,
                                                   const float2* __restrict__ table, float2* __restrict__ feat,
                                                   int64_t B) {
  const int level = blockIdx.x % g.L;
  const int64_t b = (int64_t)(blockIdx.x / g.L) * 256 + threadIdx.x;
  if (b >= B) return;
  const HashLevel lv = load_level(g, level);
  const CellPos c = locate(pts_w, b, lv.scale);
  float2 acc = make_float2(0.f, 0.f);
  if (!c.oob) {
    const float2* __restrict__ tl = table + lv.offset;
    uint32_t idx[8];
    float w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float wk = 1.0f;
      uint32_t p[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (k & (1 << d)) { wk *= c.f[d]; p[d] = c.g[d] + 1u; }
        else              { wk *= 1.0f - c.f[d]; p[d] = c.g[d]; }
      }
      w[k] = wk;
      idx[k] = grid_index(lv, p[0], p[1], p[2]);
    }
    float2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = tl[idx[k]];                  // 8 independent 8-byte gathers in flight
#pragma unroll
    for (int k = 0; k < 8; ++k) { acc.x += w[k] * v[k].x; acc.y += w[k] * v[k].y; }
  }
  feat[(int64_t)level * B + b] = acc;
}

// ---- backward -----------------------------------------------------------------------------------------
struct Scatter {
  uint32_t idx[8];
  float vx[8], vy[8];
  uint32_t key;                                                    // cell id (10 bits per axis) or ~0 when out of range
};

__device__ __forceinline__ Scatter make_scatter(const HashLevel& lv, const float* __restrict__ pts_w,
                                                const float2* __restrict__ dfeat, int level, int64_t b, int64_t B) {
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

struct LevelList {
  int32_t n;
  int32_t level[NOF_MAX_LEVELS];
};

// Emission of the merged runs.  Measured on MI355X (tools/atomic_probe.py): fp32 atomics retire at ~20.8 G line-requests/s
// chip-wide, lanes of ONE instruction that fall into the same 64-byte line merge into one request (x/y pair in adjacent
// lanes: 2x; consecutive entries: 8x) and a hot line serialises (3.8 G/s).  So a run leader does not issue its 16 atomics
// itself (16 instructions, one line each): the leaders' (row, value) lists are compacted through a small LDS stage and
// re-read so that 16 ADJACENT lanes carry one cell -- [corner k][channel] with k's bit 0 = the x neighbour, whose row is
// idx+1 for dense levels and for even x of hashed levels (prime 1) -- i.e. 4..8 line requests per cell instead of 16.
struct EmitStage {                                                    // odd row strides: conflict-free LDS writes by rank
  float val[4][64][17];
  uint32_t row[4][64][9];
};

__device__ __forceinline__ void emit_packed(EmitStage& st, const Scatter& sc, bool lead, float* __restrict__ gt) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned long long mask = __ballot(lead);
  const int nl = __popcll(mask);
  const int rank = __popcll(mask & ((1ull << lane) - 1ull));
  if (lead) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      st.val[w][rank][2 * k] = sc.vx[k];
      st.val[w][rank][2 * k + 1] = sc.vy[k];
      st.row[w][rank][k] = sc.idx[k];
    }
  }
  __syncthreads();                                                    // block-uniform call site (all four waves arrive)
  const int e = lane & 15;
  for (int m = lane >> 4; m < nl; m += 4) {
    const uint32_t r = st.row[w][m][e >> 1];
    atomicAdd(&gt[2 * (size_t)r + (e & 1)], st.val[w][m][e]);     // gridencoder.cu:317-333 (fp32 atomics)
  }
}

// levels that do not fit LDS: wave-merged, lane-packed global atomics
__global__ __launch_bounds__(256) void k_hash_bwd_agg(NofHashGrid g, LevelList ll, const float* __restrict__ pts_w,
                                                       const float2* __restrict__ dfeat, float* __restrict__ grad_table,
                                                       int64_t B, uint32_t merge_max_res) {
  __shared__ EmitStage stage;
  const int level = ll.level[blockIdx.x % ll.n];
  const int64_t b = (int64_t)(blockIdx.x / ll.n) * 256 + threadIdx.x;
  const HashLevel lv = load_level(g, level);
  Scatter sc = make_scatter(lv, pts_w, dfeat, level, b, B);
  bool lead = sc.key != 0xFFFFFFFFu;
  if (lv.res <= merge_max_res) lead = wave_merge_runs(sc);          // block-uniform branch
  emit_packed(stage, sc, lead, grad_table + 2 * (size_t)lv.offset);
}

// levels whose slice fits LDS: accumulate privately, flush once
__global__ __launch_bounds__(1024) void k_hash_bwd_lds(NofHashGrid g, LevelList ll, int chunks, const float* __restrict__ pts_w,
                                                        const float2* __restrict__ dfeat, float* __restrict__ grad_table,
                                                        int64_t B) {
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
    Scatter sc = make_scatter(lv, pts_w, dfeat, level, b < hi ? b : B, B);
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
                                                  float* __restrict__ dpts, int64_t B) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (b >= B) return;
  float dx[3] = {0.f, 0.f, 0.f};
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
  }
#pragma unroll
  for (int gd = 0; gd < 3; ++gd) dpts[b * 3 + gd] = dx[gd] * 0.5f;    // d x01 / d x = 1/2 (grid.py:160)
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

extern "C" int nof_hash_encode_fwd(const NofHashGrid* g, const float* pts_w, const float* table, float* feat,
                                    int64_t B, void* stream) {
  if (int e = check_grid(g)) return e;
  NOF_ARG(pts_w && table && feat && B >= 0);
  if (B == 0) return 0;
  const int64_t blocks = nof_div_up(B, 256) * g->L;
  NOF_ARG(blocks < (1ll << 31));
  hipLaunchKernelGGL(k_hash_fwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *g, pts_w,
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
    NOF_HIP(hipStreamCreateWithFlags(&g_side.stream, hipStreamNonBlocking));
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
  NOF_ARG(pts_w && table && dfeat && grad_table && B >= 0);
  if (B == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  // split the levels: slices of <= 128 KiB are accumulated in LDS, the others go through wave-merged global atomics
  const size_t lds_cap = 128 * 1024;
  LevelList small, big;
  small.n = big.n = 0;
  size_t lds_need = 0;
  for (int l = 0; l < g->L; ++l) {
    const size_t bytes = (size_t)g->size[l] * 8;
    if (bytes <= lds_cap) { small.level[small.n++] = l; if (bytes > lds_need) lds_need = bytes; }
    else big.level[big.n++] = l;
  }
  SideStream* side = nullptr;
  const bool fork = big.n > 0 && (dpts != nullptr || small.n > 0);
  hipStream_t s2 = st;
  if (fork) {
    if (int e = side_stream(&side)) return e;
    s2 = side->stream;
    NOF_HIP(hipEventRecord(side->fork, st));
    NOF_HIP(hipStreamWaitEvent(s2, side->fork, 0));
  }
  if (big.n > 0) {
    const int64_t blocks = nof_div_up(B, 256) * big.n;
    NOF_ARG(blocks < (1ll << 31));
    hipLaunchKernelGGL(k_hash_bwd_agg, dim3((unsigned)blocks), dim3(256), 0, st, *g, big, pts_w, (const float2*)dfeat,
                       grad_table, B, 1023u);
    NOF_LAUNCH_OK();
  }
  if (dpts) {
    hipLaunchKernelGGL(k_hash_dx, dim3((unsigned)nof_div_up(B, 256)), dim3(256), 0, s2, *g, pts_w, (const float2*)table,
                       (const float2*)dfeat, dpts, B);
    NOF_LAUNCH_OK();
  }
  if (small.n > 0) {
    const int chunks = 64;
    if (lds_need > 64 * 1024)
      NOF_HIP(hipFuncSetAttribute((const void*)k_hash_bwd_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_need));
    hipLaunchKernelGGL(k_hash_bwd_lds, dim3((unsigned)(chunks * small.n)), dim3(1024), lds_need, s2, *g, small, chunks, pts_w,
                       (const float2*)dfeat, grad_table, B);
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
