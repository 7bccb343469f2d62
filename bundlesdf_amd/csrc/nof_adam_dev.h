// torch.optim.Adam's element update and the flat-buffer loop around it (nerf_runner.py:502,756-761): shared by nof_loss.hip
// (nof_adam_step / nof_adam_step_dyn) and nof_mlp.hip (nof_adam_step_tail: the same update with the MLP operand image and the pose
// table kept current by the launch itself).
#pragma once
#include "nof_common.h"

// ------------------------------------------------------------------------------------------------
// torch.optim.Adam, single tensor form: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2;
// p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps).  HBM streaming: 16 B read + 16 B written per parameter.
struct AdamK { float step_basic, step_pose, b1, b2, eps, inv_sqrt_bc2; };

__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, float ss, const AdamK& k) {
  const float mi = k.b1 * m + (1.0f - k.b1) * g;
  const float vi = k.b2 * v + (1.0f - k.b2) * g * g;
  const float denom = sqrtf(vi) * k.inv_sqrt_bc2 + k.eps;
  p = p - ss * (mi / denom);
  m = mi;
  v = vi;
  g = 0.0f;                                                            // optimizer.zero_grad() for the next step
}

// Entries [0, n) of the four flat buffers.  When they share their offset from a 16-byte boundary (they do: same index range of
// four allocations) the body moves 16 bytes per lane and array -- 4x the bytes in flight of the scalar form, which is what a
// 59 M-parameter table (cfg5: nothing of it stays in the MALL) needs to approach the HBM rate; element arithmetic is unchanged.
// `skip_flags` (may be NULL): bit 2 of skip_flags[0] = this step's weight gradient is not finite (raised by nof_reduce_partials /
// nof_grad_check before this launch).  Then the step is SKIPPED the way torch's GradScaler.step skips it (nerf_runner.py:756-761):
// parameters and moments stay as they are, the gradient is zeroed for the next step.
__device__ __forceinline__ void adam_range(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                           float* __restrict__ v, int64_t n, int64_t n_basic, const AdamK& k,
                                           const int32_t* __restrict__ skip_flags, const uint32_t block_id, const uint32_t n_blocks) {
  const int64_t tid = (int64_t)block_id * blockDim.x + threadIdx.x, stride = (int64_t)n_blocks * blockDim.x;
  if (skip_flags != nullptr && (skip_flags[0] & 4)) {        // workgroup-uniform
    for (int64_t i = tid; i < n; i += stride) g[i] = 0.0f;
    return;
  }
  const unsigned mis = (unsigned)((uintptr_t)p >> 2) & 3u;
  const bool same = (((uintptr_t)g >> 2) & 3u) == mis && (((uintptr_t)m >> 2) & 3u) == mis && (((uintptr_t)v >> 2) & 3u) == mis;
  int64_t head = same ? (int64_t)((4u - mis) & 3u) : n;
  if (head > n) head = n;
  const int64_t nvec = (n - head) >> 2, tail = head + (nvec << 2);
  for (int64_t i = tid; i < head; i += stride) adam_one(p[i], g[i], m[i], v[i], i < n_basic ? k.step_basic : k.step_pose, k);
  float4* p4 = (float4*)(p + head); float4* g4 = (float4*)(g + head); float4* m4 = (float4*)(m + head); float4* v4 = (float4*)(v + head);
  for (int64_t q = tid; q < nvec; q += stride) {
    float4 pp = p4[q], gg = g4[q], mm = m4[q], vv = v4[q];
    // Stores that would write back the bits already there are left out (the loads are not: they decide).  A gradient that is
    // all-zero bits needs no zeroing -- six of seven table rows at cfg2 in any one step -- and an entry whose gradient and moments
    // are all-zero bits is a fixed point of the update (m = v = +0, p - step * (0 / eps) = p): hash rows no sample has reached yet,
    // most of the table in the first steps of a run.  Bit patterns, not values: -0 takes the arithmetic path.
    const bool g_zero = (__float_as_uint(gg.x) | __float_as_uint(gg.y) | __float_as_uint(gg.z) | __float_as_uint(gg.w)) == 0u;
    const bool idle = g_zero && (__float_as_uint(mm.x) | __float_as_uint(mm.y) | __float_as_uint(mm.z) | __float_as_uint(mm.w) |
                                 __float_as_uint(vv.x) | __float_as_uint(vv.y) | __float_as_uint(vv.z) | __float_as_uint(vv.w)) == 0u;
    if (idle) continue;
    const int64_t i = head + (q << 2);
    adam_one(pp.x, gg.x, mm.x, vv.x, i < n_basic ? k.step_basic : k.step_pose, k);
    adam_one(pp.y, gg.y, mm.y, vv.y, i + 1 < n_basic ? k.step_basic : k.step_pose, k);
    adam_one(pp.z, gg.z, mm.z, vv.z, i + 2 < n_basic ? k.step_basic : k.step_pose, k);
    adam_one(pp.w, gg.w, mm.w, vv.w, i + 3 < n_basic ? k.step_basic : k.step_pose, k);
    p4[q] = pp; m4[q] = mm; v4[q] = vv;
    if (!g_zero) g4[q] = gg;
  }
  for (int64_t i = tail + tid; i < n; i += stride) adam_one(p[i], g[i], m[i], v[i], i < n_basic ? k.step_basic : k.step_pose, k);
}

// NofStepState of the NEXT optimiser step (one thread): set_step < 0: step += 1, else step = set_step
__device__ __forceinline__ void step_state_advance(NofStepState* st, float lrate, float lrate_pose, float decay_rate, int n_iters,
                                                   float b1, float b2, int set_step) {
  const uint32_t s = set_step < 0 ? st->step + 1u : (uint32_t)set_step;
  st->step = s;
  // the optimiser step with index s uses the rate set at the last g <= s - 1 with g % 10 == 0, g > 0 (nerf_runner.py:762-763)
  const uint32_t g = s <= 10u ? 0u : ((s - 1u) / 10u) * 10u;
  const double k = g == 0u ? 1.0 : pow((double)decay_rate, (double)g / (double)n_iters);
  const double t = (double)s + 1.0;
  const double bc1 = 1.0 - pow((double)b1, t), bc2 = 1.0 - pow((double)b2, t);
  st->step_basic = (float)((double)lrate * k / bc1);
  st->step_pose = (float)((double)lrate_pose * k / bc1);
  st->inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
}
