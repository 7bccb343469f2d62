// Multiresolution hash-grid encode / backward for gfx950.
//
// What it computes is kernel_grid / kernel_grid_backward / kernel_input_backward of the reference
// (mycuda/torch_ngp_grid_encoder/gridencoder.cu:107-365); how it is laid out is MI355X-first:
//   * one lane = one (point, level); the block->level map is `level = blockIdx % L`, so with the
//     dispatcher's block b -> XCD b%8 placement every XCD touches only L/8 (or 1 of L<8) levels and
//     keeps those levels' table rows in ITS 4 MiB L2 (placement is a speed assumption only);
//   * features are stored level-major [L,B,2] so each XCD streams its own contiguous slab (8 B/lane);
//   * table rows are one aligned 8-byte gather (C == 2); 8 gathers are issued back to back per lane;
//   * backward recomputes indices/weights (no dy_dx tensor) and uses hardware fp32 atomics.
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

__global__ __launch_bounds__(256) void k_hash_bwd(NofHashGrid g, const float* __restrict__ pts_w,
                                                   const float2* __restrict__ table, const float2* __restrict__ dfeat,
                                                   float* __restrict__ grad_table, float* __restrict__ dpts, int64_t B) {
  const int level = blockIdx.x % g.L;
  const int64_t b = (int64_t)(blockIdx.x / g.L) * 256 + threadIdx.x;
  if (b >= B) return;
  const HashLevel lv = load_level(g, level);
  const CellPos c = locate(pts_w, b, lv.scale);
  if (c.oob) return;                                                // gridencoder.cu:276-281
  const float2 gr = dfeat[(int64_t)level * B + b];
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
  if (dpts != nullptr) {
    const float2* __restrict__ tl = table + lv.offset;
    float2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = tl[idx[k]];
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
      atomicAdd(&dpts[b * 3 + gd], s * 0.5f);                       // d x01 / d x = 1/2 (grid.py:160)
    }
  }
  float* __restrict__ gt = grad_table + 2 * (size_t)lv.offset;
#pragma unroll
  for (int k = 0; k < 8; ++k) {                                     // gridencoder.cu:317-333 (fp32 atomics)
    atomicAdd(&gt[2 * (size_t)idx[k]], w[k] * gr.x);
    atomicAdd(&gt[2 * (size_t)idx[k] + 1], w[k] * gr.y);
  }
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

extern "C" int nof_hash_encode_bwd(const NofHashGrid* g, const float* pts_w, const float* table, const float* dfeat,
                                    float* grad_table, float* dpts, int64_t B, void* stream) {
  if (int e = check_grid(g)) return e;
  NOF_ARG(pts_w && table && dfeat && grad_table && B >= 0);
  if (B == 0) return 0;
  if (dpts) NOF_HIP(hipMemsetAsync(dpts, 0, sizeof(float) * 3 * (size_t)B, (hipStream_t)stream));
  const int64_t blocks = nof_div_up(B, 256) * g->L;
  NOF_ARG(blocks < (1ll << 31));
  hipLaunchKernelGGL(k_hash_bwd, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *g, pts_w,
                     (const float2*)table, (const float2*)dfeat, grad_table, dpts, B);
  NOF_LAUNCH_OK();
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
