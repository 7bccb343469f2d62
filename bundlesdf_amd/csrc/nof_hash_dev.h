// Device helpers of the multiresolution hash grid shared by nof_hash.hip (encode / scatter) and nof_mlp.hip (fused dense
// SDF-grid query): level constants, the index rule and the trilinear cell of a point.
// Reference: mycuda/torch_ngp_grid_encoder/gridencoder.cu:47-83 (index), :131-172 (cell position / weights), grid.py:160.
#pragma once
#include "nof_common.h"

struct HashLevel {
  float scale;
  uint32_t res, offset, size, hashed;
};

// `index % size` without the division sequence on the hot paths: a hashed level has size 2^T (mask), a dense level's linear
// index is below its size except for the float32 resolution quirk of exact-power levels (SURVEY 8a notes), where the
// reference's modulo wrap must be kept -- so the general `%` stays as the rare branch.  Same values as gridencoder.cu:82.
__device__ __forceinline__ uint32_t grid_index(const HashLevel& lv, uint32_t x, uint32_t y, uint32_t z) {
  uint32_t index;
  if (lv.hashed) {
    index = (x * 1u) ^ (y * 2654435761u) ^ (z * 805459861u);      // fast_hash, gridencoder.cu:47-62
    if ((lv.size & (lv.size - 1u)) == 0u) return index & (lv.size - 1u);
    return index % lv.size;
  }
  const uint32_t r1 = lv.res + 1u;                                  // align_corners == false
  index = x + y * r1 + z * r1 * r1;                                 // gridencoder.cu:70-74
  if (index >= lv.size) index %= lv.size;
  return index;
}

typedef float RowPair __attribute__((ext_vector_type(4), aligned(8)));   // two consecutive table rows: ONE 16-byte load, 8-byte aligned

struct CellPos {
  uint32_t g[3];
  float f[3];
  bool oob;
};

__device__ __forceinline__ CellPos locate3(const float (&p)[3], float scale) {
  CellPos c;
  c.oob = false;
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float x01 = (p[d] + 1.0f) * 0.5f;                        // grid.py:160
    if (x01 < 0.0f || x01 > 1.0f) c.oob = true;                     // gridencoder.cu:131
    const float pos = x01 * scale + 0.5f;                           // gridencoder.cu:164
    const float fl = floorf(pos);
    c.g[d] = (uint32_t)fl;
    c.f[d] = pos - fl;
  }
  return c;
}

__device__ __forceinline__ CellPos locate(const float* __restrict__ pts_w, int64_t b, float scale) {
  const float p[3] = {pts_w[b * 3], pts_w[b * 3 + 1], pts_w[b * 3 + 2]};
  return locate3(p, scale);
}

// The 8 corner rows of a cell.  A gather costs the vector memory pipe per INSTRUCTION (~37 clocks per CU for 64 lanes, whatever
// the number of active lanes or distinct rows: tools/atomic_probe.py variants 30-32), so the x neighbour comes in the same
// 16-byte load where it is the next row for the whole wave: dense levels (row = x + y r + z r^2) unless the modulo wrap of an
// exact-power level hits.  `lv.hashed` is wave-uniform in the caller; the row test is made uniform with a ballot.  Same values
// as eight separate loads.  Used by k_hash_dx (lane = sample, loop over levels: 247 -> 193 us beside the scatter, 110 -> 91 alone);
// in k_hash_fwd (lane = (sample, level), 8x the waves) it was 5 % slower and is not used.
__device__ __forceinline__ void gather_corners(const HashLevel& lv, const float2* __restrict__ tl, const uint32_t (&idx)[8],
                                               float2 (&v)[8]) {
  bool pairs = !lv.hashed;
#pragma unroll
  for (int k = 0; k < 8; k += 2) pairs = pairs && (idx[k + 1] == idx[k] + 1u);
  if (__builtin_amdgcn_ballot_w64(!pairs) == 0ull) {
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
}

// features of one level at one point: 8 independent 8-byte gathers in flight, then the trilinear blend (gridencoder.cu:174-200).
// PAIRS (a compile-time variant the caller selects with a WAVE-UNIFORM test, see level_pairs()): the x neighbour's row is the next
// row, so the two come in one 16-byte load -- four gather instructions instead of eight; same values, same blend order.
template <bool PAIRS = false>
__device__ __forceinline__ float2 encode_level(const HashLevel& lv, const float2* __restrict__ table, const CellPos& c) {
  float2 acc = make_float2(0.f, 0.f);
  if (c.oob) return acc;
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
  for (int k = 0; k < 8; ++k) { acc.x += w[k] * v[k].x; acc.y += w[k] * v[k].y; }
  return acc;
}

// Is row(x+1, y, z) == row(x, y, z) + 1 for EVERY cell of the level?  True for a dense level whose linear index never reaches
// the modulo wrap: (res+1)^3 <= size (false on the exact-power levels where the float32 resolution is one above the allocated
// one, SURVEY 8a notes).  Depends on the level only, i.e. uniform for a workgroup that works on one level.
__device__ __forceinline__ bool level_pairs(const HashLevel& lv) {
  const uint32_t r1 = lv.res + 1u;
  return !lv.hashed && r1 <= 1024u && r1 * r1 * r1 <= lv.size;
}

__device__ __forceinline__ HashLevel load_level(const NofHashGrid& g, int l) {
  HashLevel lv;
  lv.scale = g.scale[l]; lv.res = g.resolution[l]; lv.offset = g.offset[l]; lv.size = g.size[l]; lv.hashed = g.hashed[l];
  return lv;
}


// kaolin quantize_points + occupancy bit of the level-`n` grid (Utils.py:393-398): floor(clamp(n*(x+1)/2, 0, n-1))
__device__ __forceinline__ bool occ_point_test(const uint32_t* __restrict__ bits, int n, float px, float py, float pz) {
  const float p[3] = {px, py, pz};
  int c[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float q = (float)n * (p[d] + 1.0f) / 2.0f;
    q = fminf(fmaxf(q, 0.0f), (float)n - 1.0f);
    c[d] = (int)floorf(q);
  }
  const uint32_t id = ((uint32_t)c[0] * n + c[1]) * n + c[2];
  return (bits[id >> 5] >> (id & 31)) & 1u;
}
