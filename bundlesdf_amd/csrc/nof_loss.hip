// Depth-guided compositing, loss terms and dL/draw in one pass; fused Adam; small reductions.
//
//   raw2outputs                 nerf_runner.py:1132-1169   (weights come from (depth - z)/trunc only)
//   train_loop loss assembly    nerf_runner.py:680-732
//   get_masks / get_sdf_loss    nerf_helpers.py:367-399
//   torch.optim.Adam step       nerf_runner.py:502,756-761
// One wave64 per ray: lanes stride over the S samples, three short passes, wave reductions by DPP shuffles;
// raw[R,S,4] is read as one float4 per sample and dL/draw written the same way.
#include "nof_common.h"
#include "nof_reduce_dev.h"
#include "nof_adam_dev.h"
#pragma clang fp contract(off)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float depth_weight(const NofLossCfg& c, float depth, float z, bool invalid) {
  const float sfd = (depth - z) / c.trunc;
  float w = sigmoidf_(sfd * c.sdf_lambda) * sigmoidf_(-sfd * c.sdf_lambda);
  const bool mask = (z - depth <= c.trunc * c.neg_trunc_ratio) && (z - depth >= -c.trunc);
  return invalid ? 0.0f : (mask ? w : 0.0f);
}

__global__ __launch_bounds__(64) void k_composite_loss(NofLossCfg c, const float4* __restrict__ raw,
                                                        const float* __restrict__ z_vals, const uint8_t* __restrict__ valid,
                                                        const float* __restrict__ batch, int64_t R, int S,
                                                        float* __restrict__ rgb_map, float* __restrict__ weights,
                                                        float4* __restrict__ draw, float* __restrict__ loss_rows,
                                                        uint8_t* __restrict__ tile_flags) {
  const int64_t r = blockIdx.x;
  const int lane = threadIdx.x;
  const float* row = batch + r * NOF_RAY_COLS;
  const float gt[3] = {row[3], row[4], row[5]};
  const float depth = row[6];
  const bool first = row[8] == 0.0f;
  const bool type1 = row[9] == 1.0f;
  const bool type0 = row[9] == 0.0f;
  const bool invalid = depth > c.far_sc;
  const int64_t base = r * S;

  float wsum = 0.0f, nvalid = 0.0f;
  for (int s = lane; s < S; s += 64) {
    wsum += depth_weight(c, depth, z_vals[base + s], invalid);
    nvalid += valid[base + s] ? 1.0f : 0.0f;
  }
  wsum = wave_sum(wsum);
  nvalid = wave_sum(nvalid);
  const float denom = wsum + 1e-10f;

  float m0 = 0.f, m1 = 0.f, m2 = 0.f;
  for (int s = lane; s < S; s += 64) {
    float w = depth_weight(c, depth, z_vals[base + s], invalid) / denom;
    if (!valid[base + s]) w = 0.0f;                                   // nerf_runner.py:1166
    if (weights) weights[base + s] = w;
    const float4 q = raw[base + s];
    m0 += w * sigmoidf_(q.x);
    m1 += w * sigmoidf_(q.y);
    m2 += w * sigmoidf_(q.z);
  }
  m0 = wave_sum(m0); m1 = wave_sum(m1); m2 = wave_sum(m2);
  const bool valid_ray = (nvalid > 0.0f) && type0;                     // nerf_runner.py:693
  const float ray_w = valid_ray ? (first ? c.first_frame_weight : 1.0f) : 0.0f;
  if (lane == 0) { rgb_map[r * 3] = m0; rgb_map[r * 3 + 1] = m1; rgb_map[r * 3 + 2] = m2; }
  const float e0 = m0 - gt[0], e1 = m1 - gt[1], e2 = m2 - gt[2];
  const float inv3R = 1.0f / (3.0f * (float)R);
  const float invRS = 1.0f / ((float)R * (float)S);
  const float k_rgb = 2.0f * ray_w * c.rgb_weight * inv3R;
  const float dm0 = k_rgb * e0, dm1 = k_rgb * e1, dm2 = k_rgb * e2;

  float l_fs = 0.f, l_empty = 0.f, l_sdf = 0.f, l_fsrgb = 0.f;
  const bool valid_depth = (depth >= c.near_sc) && (depth <= c.far_sc);
  for (int s = lane; s < S; s += 64) {
    const float z = z_vals[base + s];
    const bool v = valid[base + s] != 0;
    float w = depth_weight(c, depth, z, invalid) / denom;
    if (!v) w = 0.0f;
    const float4 q = raw[base + s];
    const float sdf = q.w;
    const float sw = (v && !type1) ? ray_w : 0.0f;                      // nerf_runner.py:699,723
    const float c0 = sigmoidf_(q.x), c1 = sigmoidf_(q.y), c2 = sigmoidf_(q.z);
    float4 g;
    g.x = dm0 * w * c0 * (1.0f - c0);
    g.y = dm1 * w * c1 * (1.0f - c1);
    g.z = dm2 * w * c2 * (1.0f - c2);
    float gs = 0.0f;
    const bool front = z < depth - c.trunc;
    const bool back = z > depth + c.trunc * c.neg_trunc_ratio;
    if (invalid && sdf < c.fs_sdf) {                                     // nerf_helpers.py:387-389
      const float d = sdf - c.fs_sdf;
      l_fs += d * d * sw;
      gs += 2.0f * d * sw * 0.5f * c.fs_weight * invRS;
    }
    if (front && !invalid && sdf < 1.0f) {                               // nerf_helpers.py:391-392
      l_empty += fabsf(sdf - 1.0f) * sw;
      gs += -sw * c.empty_weight * c.fs_weight * invRS;
    }
    if (!front && !back && valid_depth) {                                // nerf_helpers.py:372,395
      const float d = (z + sdf * c.trunc) - depth;
      l_sdf += d * d * sw;
      gs += 2.0f * d * c.trunc * sw * 0.5f * c.trunc_weight * invRS;
    }
    if (c.fs_rgb_weight > 0.0f && front) {                               // nerf_runner.py:730-732
      const float k = 2.0f * sw * c.fs_rgb_weight * invRS / 3.0f;
      l_fsrgb += ((c0 - 1.0f) * (c0 - 1.0f) + (c1 - 1.0f) * (c1 - 1.0f) + (c2 - 1.0f) * (c2 - 1.0f)) * sw;
      g.x += k * (c0 - 1.0f) * c0 * (1.0f - c0);
      g.y += k * (c1 - 1.0f) * c1 * (1.0f - c1);
      g.z += k * (c2 - 1.0f) * c2 * (1.0f - c2);
    }
    g.w = gs;
    g.x *= c.grad_scale; g.y *= c.grad_scale; g.z *= c.grad_scale; g.w *= c.grad_scale;
    draw[base + s] = g;
    if (tile_flags != nullptr) {
      // work list of the backward (S % 32 == 0: this wave's 64 samples are exactly two 32-sample tiles of the batch): a tile is
      // flagged when any of its rows of dL/draw is non-zero.  Everything the backward derives from a sample is linear in its row.
      const unsigned long long nz = __ballot(g.x != 0.0f || g.y != 0.0f || g.z != 0.0f || g.w != 0.0f);
      const int64_t t0 = (base + (s - lane)) >> 5;
      if (lane == 0) tile_flags[t0] = (uint32_t)nz != 0u ? 1 : 0;
      if (lane == 32) tile_flags[t0 + 1] = (uint32_t)(nz >> 32) != 0u ? 1 : 0;     // (lane 32 inactive = that tile does not exist)
    }
  }
  l_fs = wave_sum(l_fs); l_empty = wave_sum(l_empty); l_sdf = wave_sum(l_sdf); l_fsrgb = wave_sum(l_fsrgb);
  if (lane == 0 && loss_rows) {
    // per-ray terms go to their own row (plain stores); k_loss_reduce sums the rows.  (4096 rays x 7 atomics on ONE line
    // cost 0.35 ms: gfx950 atomics execute memory-side and serialise per line.)
    const float rgb_loss = c.rgb_weight * (e0 * e0 + e1 * e1 + e2 * e2) * ray_w * inv3R;
    const float fs_loss = c.fs_weight * (0.5f * l_fs + c.empty_weight * l_empty) * invRS;
    const float sdf_loss = c.trunc_weight * 0.5f * l_sdf * invRS;
    const float fsrgb = c.fs_rgb_weight * l_fsrgb * invRS / 3.0f;
    float4* row = (float4*)(loss_rows + r * 8);
    row[0] = make_float4(rgb_loss + fs_loss + sdf_loss + fsrgb, rgb_loss, fs_loss, sdf_loss);
    row[1] = make_float4(fsrgb, nvalid, valid_ray ? 1.0f : 0.0f, 0.0f);
  }
}

// The same with the ray's samples held in registers (S <= 64 * NS, NS <= 4: the reference's 128 + 64 samples are three per lane):
// ONE round trip to memory instead of three dependent passes over z / valid / raw -- the launch is latency-bound (4096 waves, a
// few hundred bytes each) -- and four rays per workgroup.  Element arithmetic and summation order are those of the loop form above
// (per lane s = lane, lane + 64, ...; then the wave reduction), so both give the same bits.
template <int NS>
__global__ __launch_bounds__(256) void k_composite_loss_reg(NofLossCfg c, const float4* __restrict__ raw,
                                                             const float* __restrict__ z_vals, const uint8_t* __restrict__ valid,
                                                             const float* __restrict__ batch, int64_t R, int S,
                                                             float* __restrict__ rgb_map, float* __restrict__ weights,
                                                             float4* __restrict__ draw, float* __restrict__ loss_rows,
                                                             uint8_t* __restrict__ tile_flags) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;                                                    // (wave-uniform)
  const int64_t base = r * S;
  float z[NS], w[NS];
  bool in[NS], v[NS];
  float4 q[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const int s = lane + 64 * k;
    in[k] = s < S;
    z[k] = in[k] ? z_vals[base + s] : 0.0f;
    v[k] = in[k] && valid[base + s] != 0;
    q[k] = in[k] ? raw[base + s] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float* row = batch + r * NOF_RAY_COLS;
  const float gt[3] = {row[3], row[4], row[5]};
  const float depth = row[6];
  const bool first = row[8] == 0.0f;
  const bool type1 = row[9] == 1.0f;
  const bool type0 = row[9] == 0.0f;
  const bool invalid = depth > c.far_sc;

  float wsum = 0.0f, nvalid = 0.0f;
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    w[k] = in[k] ? depth_weight(c, depth, z[k], invalid) : 0.0f;
    if (in[k]) {
      wsum += w[k];
      nvalid += v[k] ? 1.0f : 0.0f;
    }
  }
  wsum = wave_sum(wsum);
  nvalid = wave_sum(nvalid);
  const float denom = wsum + 1e-10f;

  float m0 = 0.f, m1 = 0.f, m2 = 0.f;
  float c0[NS], c1[NS], c2[NS];
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    w[k] = v[k] ? w[k] / denom : 0.0f;                                  // nerf_runner.py:1166
    c0[k] = sigmoidf_(q[k].x); c1[k] = sigmoidf_(q[k].y); c2[k] = sigmoidf_(q[k].z);
    if (in[k]) {
      if (weights) weights[base + lane + 64 * k] = w[k];
      m0 += w[k] * c0[k];
      m1 += w[k] * c1[k];
      m2 += w[k] * c2[k];
    }
  }
  m0 = wave_sum(m0); m1 = wave_sum(m1); m2 = wave_sum(m2);
  const bool valid_ray = (nvalid > 0.0f) && type0;                     // nerf_runner.py:693
  const float ray_w = valid_ray ? (first ? c.first_frame_weight : 1.0f) : 0.0f;
  if (lane == 0) { rgb_map[r * 3] = m0; rgb_map[r * 3 + 1] = m1; rgb_map[r * 3 + 2] = m2; }
  const float e0 = m0 - gt[0], e1 = m1 - gt[1], e2 = m2 - gt[2];
  const float inv3R = 1.0f / (3.0f * (float)R);
  const float invRS = 1.0f / ((float)R * (float)S);
  const float k_rgb = 2.0f * ray_w * c.rgb_weight * inv3R;
  const float dm0 = k_rgb * e0, dm1 = k_rgb * e1, dm2 = k_rgb * e2;

  float l_fs = 0.f, l_empty = 0.f, l_sdf = 0.f, l_fsrgb = 0.f;
  const bool valid_depth = (depth >= c.near_sc) && (depth <= c.far_sc);
#pragma unroll
  for (int k = 0; k < NS; ++k) {
    const float sdf = q[k].w;
    const float sw = (v[k] && !type1) ? ray_w : 0.0f;                   // nerf_runner.py:699,723
    float4 g;
    g.x = dm0 * w[k] * c0[k] * (1.0f - c0[k]);
    g.y = dm1 * w[k] * c1[k] * (1.0f - c1[k]);
    g.z = dm2 * w[k] * c2[k] * (1.0f - c2[k]);
    float gs = 0.0f;
    const bool front = z[k] < depth - c.trunc;
    const bool back = z[k] > depth + c.trunc * c.neg_trunc_ratio;
    if (in[k] && invalid && sdf < c.fs_sdf) {                            // nerf_helpers.py:387-389
      const float d = sdf - c.fs_sdf;
      l_fs += d * d * sw;
      gs += 2.0f * d * sw * 0.5f * c.fs_weight * invRS;
    }
    if (in[k] && front && !invalid && sdf < 1.0f) {                      // nerf_helpers.py:391-392
      l_empty += fabsf(sdf - 1.0f) * sw;
      gs += -sw * c.empty_weight * c.fs_weight * invRS;
    }
    if (in[k] && !front && !back && valid_depth) {                       // nerf_helpers.py:372,395
      const float d = (z[k] + sdf * c.trunc) - depth;
      l_sdf += d * d * sw;
      gs += 2.0f * d * c.trunc * sw * 0.5f * c.trunc_weight * invRS;
    }
    if (in[k] && c.fs_rgb_weight > 0.0f && front) {                      // nerf_runner.py:730-732
      const float kk = 2.0f * sw * c.fs_rgb_weight * invRS / 3.0f;
      l_fsrgb += ((c0[k] - 1.0f) * (c0[k] - 1.0f) + (c1[k] - 1.0f) * (c1[k] - 1.0f) + (c2[k] - 1.0f) * (c2[k] - 1.0f)) * sw;
      g.x += kk * (c0[k] - 1.0f) * c0[k] * (1.0f - c0[k]);
      g.y += kk * (c1[k] - 1.0f) * c1[k] * (1.0f - c1[k]);
      g.z += kk * (c2[k] - 1.0f) * c2[k] * (1.0f - c2[k]);
    }
    g.w = gs;
    g.x *= c.grad_scale; g.y *= c.grad_scale; g.z *= c.grad_scale; g.w *= c.grad_scale;
    if (in[k]) draw[base + lane + 64 * k] = g;
    if (tile_flags != nullptr) {                                         // (S % 32 == 0; see the loop form)
      const unsigned long long nz = __ballot(in[k] && (g.x != 0.0f || g.y != 0.0f || g.z != 0.0f || g.w != 0.0f));
      const int64_t t0 = (base + 64 * k) >> 5;
      if (lane == 0 && in[k]) tile_flags[t0] = (uint32_t)nz != 0u ? 1 : 0;
      if (lane == 32 && in[k]) tile_flags[t0 + 1] = (uint32_t)(nz >> 32) != 0u ? 1 : 0;
    }
  }
  l_fs = wave_sum(l_fs); l_empty = wave_sum(l_empty); l_sdf = wave_sum(l_sdf); l_fsrgb = wave_sum(l_fsrgb);
  if (lane == 0 && loss_rows) {
    const float rgb_loss = c.rgb_weight * (e0 * e0 + e1 * e1 + e2 * e2) * ray_w * inv3R;
    const float fs_loss = c.fs_weight * (0.5f * l_fs + c.empty_weight * l_empty) * invRS;
    const float sdf_loss = c.trunc_weight * 0.5f * l_sdf * invRS;
    const float fsrgb = c.fs_rgb_weight * l_fsrgb * invRS / 3.0f;
    float4* out = (float4*)(loss_rows + r * 8);
    out[0] = make_float4(rgb_loss + fs_loss + sdf_loss + fsrgb, rgb_loss, fs_loss, sdf_loss);
    out[1] = make_float4(fsrgb, nvalid, valid_ray ? 1.0f : 0.0f, 0.0f);
  }
}

// ---- work list of the backward (include/nof_hip.h: NofTileList) --------------------------------------------------------------
// flags of a batch whose S is not a multiple of 32 (tiles straddle rays): one wave per 64 samples = two tiles, straight from draw
__global__ __launch_bounds__(256) void k_tile_flags(const float4* __restrict__ draw, int64_t B, uint8_t* __restrict__ tile_flags) {
  const int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  if (b < B) g = draw[b];
  const unsigned long long nz = __ballot(g.x != 0.0f || g.y != 0.0f || g.z != 0.0f || g.w != 0.0f);
  const int64_t t0 = (b - lane) >> 5;
  if (lane == 0 && b < B) tile_flags[t0] = (uint32_t)nz != 0u ? 1 : 0;
  if (lane == 32 && b < B) tile_flags[t0 + 1] = (uint32_t)(nz >> 32) != 0u ? 1 : 0;
}

// flags -> ascending list of the flagged tiles + their number.  Thread t owns a contiguous chunk, counts, the counts are scanned
// through LDS, every WAVE writes its 64 chunks' tiles.  Deterministic (no atomics): the order of the list fixes the order in which
// the backward kernels sum their partial results.
// Round 6: `nchunks` workgroups share the flags (one took 36 us for the 98 304 tiles of a cfg5 batch and 10 for cfg2's 24 576,
// with the step waiting for it): workgroup `chunk` owns the words [chunk * W, (chunk + 1) * W), counts the listed tiles IN FRONT of
// its range by itself (at most ~100 KB of flags out of L2, 16 bytes per load: no hand-over between workgroups, no ordering among
// them) and places its own behind that count; the last one writes the head.  The same list as one workgroup's.
__device__ void tile_scan(const uint8_t* __restrict__ flags, uint32_t ntiles, uint32_t* __restrict__ head,
                          uint32_t* __restrict__ tiles, int all, uint32_t chunk = 0u, uint32_t nchunks = 1u) {
  __shared__ uint32_t wsum[16], psum[16];
  const uint32_t nt = blockDim.x, t = threadIdx.x;
  // four flags per 32-bit word (the flag array is 16-byte aligned and padded to 16 bytes); a thread owns `wper` consecutive words
  // and requests them eight at a time, so that their latencies overlap (a byte-by-byte loop waited ~0.5 us per flag: 15 us)
  const uint32_t* __restrict__ fw = reinterpret_cast<const uint32_t*>(flags);
  const uint32_t nwords_all = (ntiles + 3) / 4;
  const uint32_t W = ((nwords_all + nchunks - 1) / nchunks + 3u) & ~3u;            // words per workgroup (whole 16-byte groups)
  const uint32_t w0 = chunk * W < nwords_all ? chunk * W : nwords_all;
  const uint32_t nwords = w0 + W < nwords_all ? w0 + W : nwords_all;               // this workgroup's words: [w0, nwords)
  const uint32_t wper = (nwords - w0 + nt - 1) / nt;
  const uint32_t lo = w0 + t * wper < nwords ? w0 + t * wper : nwords, hi = lo + wper < nwords ? lo + wper : nwords;
  auto word = [&](uint32_t wi) -> uint32_t {                           // 0x01 in every byte whose tile is listed
    const uint32_t left = ntiles - 4u * wi;                            // flags that exist in this word (the last one may be partial)
    const uint32_t m = left >= 4u ? 0x01010101u : ((1u << (8u * left)) - 1u) & 0x01010101u;
    return (all ? 0x01010101u : fw[wi]) & m;
  };
  // listed tiles in front of this workgroup's range (whole words, all of them complete: w0 < nwords_all is a multiple of 4 words)
  uint32_t pre = 0;
  if (all) {
    pre = t == 0 ? (4u * w0 < ntiles ? 4u * w0 : ntiles) : 0u;
  } else {
    const uint4* __restrict__ f4 = reinterpret_cast<const uint4*>(flags);
    for (uint32_t q = t; q < w0 / 4u; q += nt) {
      const uint4 v = f4[q];
      pre += __popc(v.x & 0x01010101u) + __popc(v.y & 0x01010101u) + __popc(v.z & 0x01010101u) + __popc(v.w & 0x01010101u);
    }
    for (uint32_t wi = (w0 & ~3u) + t; wi < w0; wi += nt) pre += __popc(word(wi));       // (an empty last range behind a partial group)
  }
  uint32_t cnt = 0;
  for (uint32_t base = lo; base < hi; base += 8) {
    uint32_t w[8];
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) w[k] = base + k < hi ? word(base + k) : 0u;
#pragma unroll
    for (uint32_t k = 0; k < 8; ++k) cnt += __popc(w[k]);
  }
  // inclusive scan inside the wave, then across the waves
  uint32_t inc = cnt;
  const int lane = t & 63;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = __shfl_up(inc, o, 64);
    if (lane >= o) inc += v;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) pre += __shfl_xor(pre, o, 64);
  if (lane == 63) wsum[t >> 6] = inc;
  if (lane == 0) psum[t >> 6] = pre;
  __syncthreads();
  uint32_t before = 0, total = 0;
  for (uint32_t w = 0; w < (nt + 63) / 64; ++w) {
    if (w < (t >> 6)) before += wsum[w];
    total += wsum[w];
    before += psum[w];
    total += psum[w];
  }
  // emission, wave-cooperative (round 6): the wave's 64 threads own one contiguous range of flags; it walks that range 64 flags at a
  // time, one flag per lane, and a ballot's prefix count places the listed ones -- one coalesced store instruction per 64 tiles.
  // (Rounds 3-5: every thread wrote its own chunk's tiles one store after the other, ~55 per thread at cfg5: 51 us for 98 304 tiles.)
  {
    const uint32_t wv = t >> 6;
    const uint32_t wlo = w0 + wv * 64u * wper < nwords ? w0 + wv * 64u * wper : nwords;       // the wave's words [wlo, whi)
    const uint32_t whi = wlo + 64u * wper < nwords ? wlo + 64u * wper : nwords;
    // tiles listed before this wave's range = `before`; (inc - cnt of lane 0 is 0)
    uint32_t at = before;
    const uint32_t t_lo = 4u * wlo, t_hi = 4u * whi < ntiles ? 4u * whi : ntiles;
#pragma unroll 4
    for (uint32_t base = t_lo; base < t_hi; base += 64u) {
      const uint32_t id = base + (uint32_t)lane;
      const bool on = id < t_hi && (all || (flags[id] & 1u) != 0u);    // (bit 0, like the counting pass)
      const unsigned long long m = __ballot(on);
      if (on) tiles[at + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = id;
      at += (uint32_t)__popcll(m);
    }
  }
  if (chunk + 1u == nchunks) {                                         // (the last workgroup's `total` is the list's)
    if (t == 0) { head[0] = total; head[1] = ntiles; head[2] = 0; head[3] = 0; }
    if (t == 0 && (total & 1u)) tiles[total] = ntiles;                 // an odd list ends in a tile that does not exist (pairs of tiles per wave)
  }
}

__global__ __launch_bounds__(1024) void k_tile_scan(uint8_t* __restrict__ flags, uint32_t ntiles, uint32_t* __restrict__ head,
                                                     uint32_t* __restrict__ tiles, int all) {
  if (all)
    for (uint32_t i = threadIdx.x; i < ntiles; i += blockDim.x) flags[i] = 1;
  tile_scan(flags, ntiles, head, tiles, all);
}

extern "C" int64_t nof_tile_list_bytes(int64_t B) {
  if (B < 0 || B >= (1ll << 36)) return -1;
  const uint32_t nt = nof_tile_count(B);
  return (int64_t)(nof_tile_list_words(nt) * 4 + (((size_t)nt + 15) & ~(size_t)15));
}

// workgroup 0: the per-ray loss rows -> loss_out; the others (when a work list is asked for): the tile scan, a share of the flags each
__global__ __launch_bounds__(1024) void k_loss_reduce(const float* __restrict__ rows, int64_t R, float* __restrict__ loss_out,
                                                       const uint8_t* __restrict__ tile_flags, uint32_t ntiles,
                                                       uint32_t* __restrict__ head, uint32_t* __restrict__ tiles, int overwrite,
                                                       uint32_t nchunks) {
  if (blockIdx.x >= 1 || loss_out == nullptr) {
    tile_scan(tile_flags, ntiles, head, tiles, 0, blockIdx.x - (loss_out != nullptr ? 1u : 0u), nchunks);
    return;
  }
  __shared__ float sm[16][8];
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t r0 = threadIdx.x; r0 < R; r0 += 4096) {                 // four rows per thread and round, all eight loads in flight
    float4 v[4][2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t r = r0 + 1024 * u;
      v[u][0] = v[u][1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < R) { v[u][0] = ((const float4*)rows)[2 * r]; v[u][1] = ((const float4*)rows)[2 * r + 1]; }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      acc[0] += v[u][0].x; acc[1] += v[u][0].y; acc[2] += v[u][0].z; acc[3] += v[u][0].w;
      acc[4] += v[u][1].x; acc[5] += v[u][1].y; acc[6] += v[u][1].z; acc[7] += v[u][1].w;
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = wave_sum(acc[k]);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0)
#pragma unroll
    for (int k = 0; k < 8; ++k) sm[wave][k] = acc[k];
  __syncthreads();
  if (threadIdx.x < 8) {
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < 16; ++w) s += sm[w][threadIdx.x];
    loss_out[threadIdx.x] = overwrite ? s : loss_out[threadIdx.x] + s;
  }
}

// forward compositing + losses + dL/draw in one launch, and -- with `tile_list` -- the work list of the backward: the 32-sample
// tiles that hold at least one non-zero row of dL/draw (NofTileList, include/nof_hip.h).  S % 32 == 0 (the reference's default
// 128 + 64): the flags come out of the loss kernel itself and the scan rides in the second workgroup of the loss reduction, so the
// list costs no launch; otherwise one extra pass over draw.
static int composite_loss(const NofLossCfg* cfg, const float* raw, const float* z_vals, const uint8_t* valid,
                          const float* batch, int64_t R, int32_t S, float* rgb_map, float* weights,
                          float* draw, float* loss_rows, float* loss_out, void* tile_list, int overwrite, void* stream) {
  NOF_ARG(cfg && raw && z_vals && valid && batch && rgb_map && draw && R >= 0 && S >= 1);
  NOF_ARG(loss_out == nullptr || loss_rows != nullptr);
  NOF_ARG((int64_t)R * S < (1ll << 36));
  if (R == 0) return 0;
  const int64_t B = R * (int64_t)S;
  const uint32_t nt = nof_tile_count(B);
  uint32_t* head = (uint32_t*)tile_list;
  uint32_t* tiles = head ? head + 4 : nullptr;
  uint8_t* flags = head ? (uint8_t*)(head + nof_tile_list_words(nt)) : nullptr;
  const bool fused_flags = flags != nullptr && S % 32 == 0;
#define NOF_COMPOSITE_REG(NS)                                                                                                   \
  hipLaunchKernelGGL(k_composite_loss_reg<NS>, dim3((unsigned)nof_div_up(R, 4)), dim3(256), 0, (hipStream_t)stream, *cfg,        \
                     (const float4*)raw, z_vals, valid, batch, R, S, rgb_map, weights, (float4*)draw,                            \
                     loss_out ? loss_rows : nullptr, fused_flags ? flags : nullptr)
  if (S <= 64) NOF_COMPOSITE_REG(1);
  else if (S <= 128) NOF_COMPOSITE_REG(2);
  else if (S <= 192) NOF_COMPOSITE_REG(3);
  else if (S <= 256) NOF_COMPOSITE_REG(4);
  else
    hipLaunchKernelGGL(k_composite_loss, dim3((unsigned)R), dim3(64), 0, (hipStream_t)stream, *cfg, (const float4*)raw,
                       z_vals, valid, batch, R, S, rgb_map, weights, (float4*)draw, loss_out ? loss_rows : nullptr,
                       fused_flags ? flags : nullptr);
#undef NOF_COMPOSITE_REG
  NOF_LAUNCH_OK();
  if (flags != nullptr && !fused_flags) {
    hipLaunchKernelGGL(k_tile_flags, dim3((unsigned)nof_div_up(B, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)draw, B, flags);
    NOF_LAUNCH_OK();
  }
  if (loss_out || flags) {
    // (the scan: one workgroup per ~6000 tiles -- 4 at cfg2, 16 at cfg5)
    const uint32_t nchunks = flags ? (nt / 6144u < 1u ? 1u : (nt / 6144u > 32u ? 32u : nt / 6144u)) : 0u;
    hipLaunchKernelGGL(k_loss_reduce, dim3((loss_out ? 1u : 0u) + nchunks), dim3(1024), 0, (hipStream_t)stream, loss_rows, R, loss_out,
                       flags, nt, head, tiles, overwrite, nchunks);
    NOF_LAUNCH_OK();
  }
  return 0;
}

extern "C" int nof_composite_loss_fwd_bwd(const NofLossCfg* cfg, const float* raw, const float* z_vals, const uint8_t* valid,
                                           const float* batch, int64_t R, int32_t S, float* rgb_map, float* weights,
                                           float* draw, float* loss_rows, float* loss_out, void* tile_list, void* stream) {
  return composite_loss(cfg, raw, z_vals, valid, batch, R, S, rgb_map, weights, draw, loss_rows, loss_out, tile_list, 1, stream);
}

extern "C" int nof_composite_loss(const NofLossCfg* cfg, const float* raw, const float* z_vals, const uint8_t* valid,
                                   const float* batch, int64_t R, int32_t S, float* rgb_map, float* weights, float* draw,
                                   float* loss_rows, float* loss_out, void* stream) {
  return composite_loss(cfg, raw, z_vals, valid, batch, R, S, rgb_map, weights, draw, loss_rows, loss_out, nullptr, 0, stream);
}

// render_images' depth (nerf_runner.py:604-612): z at the first sample pair whose SDFs differ in sign, `far` for a ray whose pairs
// all have a strictly positive product, z_vals[:,0] for the rest (torch.argmax of an all-false mask is 0).  One wave per ray.
__global__ __launch_bounds__(64) void k_render_depth(const float4* __restrict__ raw, const float* __restrict__ z_vals, int64_t R, int S,
                                                      float far, float* __restrict__ depth) {
  const int64_t r = blockIdx.x;
  const int lane = threadIdx.x;
  const int64_t base = r * S;
  bool all_pos = true;
  int first = -1;
  for (int s0 = 0; s0 < S - 1 && first < 0; s0 += 64) {
    const int s = s0 + lane;
    float prod = 1.0f;
    if (s < S - 1) prod = raw[base + s + 1].w * raw[base + s].w;       // signs = sdf[:, 1:] * sdf[:, :-1]
    const unsigned long long neg = __builtin_amdgcn_ballot_w64(prod < 0.0f);
    const unsigned long long pos = __builtin_amdgcn_ballot_w64(s >= S - 1 || prod > 0.0f);
    all_pos = all_pos && (pos == ~0ull);
    if (neg != 0ull) first = s0 + __builtin_ctzll(neg);
  }
  if (lane == 0) depth[r] = all_pos ? far : z_vals[base + (first < 0 ? 0 : first)];
}

extern "C" int nof_render_depth(const float* raw, const float* z_vals, int64_t R, int32_t S, float far, float* depth, void* stream) {
  NOF_ARG(raw && z_vals && depth && R >= 0 && S >= 2);
  if (R == 0) return 0;
  hipLaunchKernelGGL(k_render_depth, dim3((unsigned)R), dim3(64), 0, (hipStream_t)stream, (const float4*)raw, z_vals, R, (int)S, far, depth);
  NOF_LAUNCH_OK();
  return 0;
}

// The work list on its own: from an existing dL/draw [B,4] (`all` == 0), or every tile of the batch (`all` != 0; draw may be NULL):
// the list that makes the backward kernels do the whole batch without looking for zeros (the dense-backward measurement of bench.py).
extern "C" int nof_tile_list_build(const float* draw, int64_t B, int32_t all, void* tile_list, void* stream) {
  NOF_ARG(tile_list && B >= 0 && B < (1ll << 36) && (all || draw));
  const uint32_t nt = nof_tile_count(B);
  uint32_t* head = (uint32_t*)tile_list;
  uint8_t* flags = (uint8_t*)(head + nof_tile_list_words(nt));
  if (!all && B > 0) {
    hipLaunchKernelGGL(k_tile_flags, dim3((unsigned)nof_div_up(B, 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)draw, B, flags);
    NOF_LAUNCH_OK();
  }
  hipLaunchKernelGGL(k_tile_scan, dim3(1), dim3(1024), 0, (hipStream_t)stream, flags, nt, head, head + 4, (int)(all != 0));
  NOF_LAUNCH_OK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// (AdamK, adam_one, adam_range: nof_adam_dev.h)
__global__ __launch_bounds__(256) void k_adam(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, int64_t n, int64_t n_basic, float step_basic,
                                               float step_pose, float b1, float b2, float eps, float inv_sqrt_bc2,
                                               const int32_t* __restrict__ skip_flags) {
  adam_range(p, g, m, v, n, n_basic, AdamK{step_basic, step_pose, b1, b2, eps, inv_sqrt_bc2}, skip_flags, blockIdx.x, gridDim.x);
}

// ---- the same with the per-step scalars in device memory (replayable captured step) ----------------------------------
__global__ void k_step_advance(NofStepState* st, float lrate, float lrate_pose, float decay_rate, int n_iters, float b1, float b2,
                               int set_step) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  step_state_advance(st, lrate, lrate_pose, decay_rate, n_iters, b1, b2, set_step);
}

__global__ __launch_bounds__(256) void k_adam_dyn(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, int64_t n, int64_t n_basic,
                                                   const NofStepState* __restrict__ st, float b1, float b2, float eps,
                                                   const int32_t* __restrict__ skip_flags) {
  adam_range(p, g, m, v, n, n_basic, AdamK{st->step_basic, st->step_pose, b1, b2, eps, st->inv_sqrt_bc2}, skip_flags, blockIdx.x, gridDim.x);
}

extern "C" int nof_step_state_advance(NofStepState* d_state, float lrate, float lrate_pose, float decay_rate, int32_t n_iters,
                                       float beta1, float beta2, int32_t set_step, void* stream) {
  NOF_ARG(d_state && n_iters > 0);
  hipLaunchKernelGGL(k_step_advance, dim3(1), dim3(64), 0, (hipStream_t)stream, d_state, lrate, lrate_pose, decay_rate, (int)n_iters,
                     beta1, beta2, (int)set_step);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_adam_step_dyn(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t n_basic,
                                  const NofStepState* d_state, float beta1, float beta2, float eps, const int32_t* skip_flags,
                                  void* stream) {
  NOF_ARG(params && grads && exp_avg && exp_avg_sq && d_state && n >= 0 && n_basic >= 0 && n_basic <= n);
  if (n == 0) return 0;
  const int64_t blocks = nof_div_up(n, 1024) < 4096 ? nof_div_up(n, 1024) : 4096;
  hipLaunchKernelGGL(k_adam_dyn, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, n,
                     n_basic, d_state, beta1, beta2, eps, skip_flags);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t n_basic,
                              float lr, float lr_pose, float beta1, float beta2, float eps, int32_t step,
                              const int32_t* skip_flags, void* stream) {
  NOF_ARG(params && grads && exp_avg && exp_avg_sq && n >= 0 && n_basic >= 0 && n_basic <= n && step >= 1);
  if (n == 0) return 0;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const float inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  const int64_t blocks = nof_div_up(n, 1024) < 4096 ? nof_div_up(n, 1024) : 4096;
  hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, params, grads, exp_avg,
                     exp_avg_sq, n, n_basic, (float)(lr / bc1), (float)(lr_pose / bc1), beta1, beta2, eps, inv_sqrt_bc2, skip_flags);
  NOF_LAUNCH_OK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// (kernel and launch: nof_reduce_dev.h)
extern "C" int nof_reduce_partials(const float* partials, int32_t n_rows, int32_t n_cols, float* out, int32_t* flags, void* stream) {
  return reduce_partials_launch(partials, n_rows, n_cols, out, flags, stream);
}

// flags[0] |= 4 when any of grad[0, n) is not finite: the check of nof_reduce_partials for a gradient that was summed over the
// data-parallel ranks afterwards (a rank whose own partial sums were finite receives the other rank's inf with the all-reduce and
// must skip the same step, or the replicas part).
__global__ __launch_bounds__(256) void k_grad_check(const float* __restrict__ grad, int64_t n, int32_t* __restrict__ flags) {
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    bad = bad || !(fabsf(grad[i]) <= 3.0e38f);
  if (__builtin_amdgcn_ballot_w64(bad) != 0ull && (threadIdx.x & 63) == 0) atomicOr(&flags[0], 4);
}

extern "C" int nof_grad_check(const float* grad, int64_t n, int32_t* flags, void* stream) {
  NOF_ARG(grad && flags && n >= 0);
  if (n == 0) return 0;
  const int64_t blocks = nof_div_up(n, 256) < 256 ? nof_div_up(n, 256) : 256;
  hipLaunchKernelGGL(k_grad_check, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, grad, n, flags);
  NOF_LAUNCH_OK();
  return 0;
}

__global__ void k_feature_reg(const float* __restrict__ data, float* __restrict__ grad, int n, float k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) grad[i] += k * data[i];
}

// pose_reg = w * || pose_array.data[1:] ||_2  (nerf_runner.py:749-752; frame 0 is the anchor and is excluded):
// grad += scale * w * x / ||x||  (0 when the norm is 0, like torch), loss_out[0] += w * ||x||
__global__ __launch_bounds__(256) void k_pose_reg(const float* __restrict__ pose, float* __restrict__ grad, int n, float w,
                                                   float scale, float* __restrict__ loss_out) {
  __shared__ float sm[256];
  float s = 0.0f;
  for (int i = 6 + threadIdx.x; i < n; i += 256) s += pose[i] * pose[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int k = 128; k > 0; k >>= 1) {
    if (threadIdx.x < k) sm[threadIdx.x] += sm[threadIdx.x + k];
    __syncthreads();
  }
  const float nrm = sqrtf(sm[0]);
  if (nrm > 0.0f)
    for (int i = 6 + threadIdx.x; i < n; i += 256) grad[i] += scale * w * (pose[i] / nrm);
  if (threadIdx.x == 0 && loss_out) loss_out[0] += w * nrm;
}

extern "C" int nof_pose_reg(const float* pose_data, float* grad_pose, int32_t F, float pose_reg_weight, float grad_scale,
                             float* loss_out, void* stream) {
  if (F <= 1 || pose_reg_weight == 0.0f) return 0;
  NOF_ARG(pose_data && grad_pose);
  hipLaunchKernelGGL(k_pose_reg, dim3(1), dim3(256), 0, (hipStream_t)stream, pose_data, grad_pose, (int)F * 6, pose_reg_weight,
                     grad_scale, loss_out);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_small_regs(const float* feat_data, float* grad_feat, int32_t n_feat, float feature_reg_weight,
                               float grad_scale, void* stream) {
  if (n_feat <= 0 || feature_reg_weight == 0.0f) return 0;
  NOF_ARG(feat_data && grad_feat);
  const float k = 2.0f * feature_reg_weight / (float)n_feat * grad_scale;   // d/dx w*mean(x^2)  (nerf_runner.py:746)
  hipLaunchKernelGGL(k_feature_reg, dim3((unsigned)nof_div_up(n_feat, 256)), dim3(256), 0, (hipStream_t)stream, feat_data,
                     grad_feat, n_feat, k);
  NOF_LAUNCH_OK();
  return 0;
}
