// Per-frame SE(3) pose corrections: forward (tanh -> se3 exp -> Delta @ c2w), analytic backward, and the
// per-ray accumulation of dL/dDelta from point and view-direction gradients.
//
//   PoseArray.get_matrices            nerf_helpers.py:143-154
//   pytorch3d se3_exp_map             (third-party; algorithm restated in oracle/nof_oracle.py:se3_exp)
//   tf = get_matrices(frame_ids) @ c2w_array[frame_ids]      nerf_runner.py:1051-1053
//   input_dirs = tf[:3,:3] @ viewdirs -> SHEncoder           nerf_runner.py:1282-1283, nerf_helpers.py:67-105
// The reference leaves these gradients to autograd over ~20 eager kernels; here they are three tiny launches.
#include "nof_common.h"
#include "nof_pose_dev.h"
#pragma clang fp contract(off)

__global__ void k_pose_fwd(const float* __restrict__ pose, const float* __restrict__ c2w, float max_trans, float max_rot,
                           float* __restrict__ tf, int F) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F) return;
  pose_fwd_frame(f, pose, c2w, max_trans, max_rot, tf);
}

// (se3_backward: nof_pose_dev.h)
__global__ void k_pose_bwd(const float* __restrict__ pose, const float* __restrict__ g_delta, float max_trans,
                           float max_rot, float* __restrict__ grad_pose, int F) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F || f == 0) return;
  float gp[6];
  se3_backward(pose + (size_t)f * 6, g_delta + (size_t)f * 12, max_trans, max_rot, gp);
#pragma unroll
  for (int k = 0; k < 6; ++k) grad_pose[(size_t)f * 6 + k] += gp[k];
}

// one wave per ray (the body lives in nof_pose_dev.h: the training step's LDS-level launch carries it as a rider too)
__global__ __launch_bounds__(64) void k_pose_grad_accum(const float* __restrict__ dpts, float* __restrict__ dview,
                                                         const float* __restrict__ batch, const float* __restrict__ z_vals,
                                                         const float* __restrict__ c2w, const float* __restrict__ tf, int ff,
                                                         int sh_degree, int64_t R, int S, float* __restrict__ g_ray,
                                                         float* __restrict__ slots) {
  pose_grad_accum_ray(dpts, dview, batch, z_vals, c2w, tf, ff, sh_degree, R, S, g_ray, slots, (int64_t)blockIdx.x, (int)threadIdx.x);
}

// a frame's summed row (in LDS, visible to the workgroup) -> its gradients
__device__ __forceinline__ void pose_frame_epilogue(const float* sm0, int f, int ff, const float* __restrict__ pose, float max_trans,
                                                    float max_rot, float* __restrict__ grad_pose, float* __restrict__ grad_feat,
                                                    float* __restrict__ g_delta) {
  if (threadIdx.x < ff && grad_feat) grad_feat[(size_t)f * ff + threadIdx.x] += sm0[12 + threadIdx.x];
  if (threadIdx.x < 12 && g_delta) g_delta[(size_t)f * 12 + threadIdx.x] = sm0[threadIdx.x];
  if (threadIdx.x == 0 && pose && grad_pose && f != 0) {
    float G[12], gp[6];
#pragma unroll
    for (int k = 0; k < 12; ++k) G[k] = sm0[k];
    se3_backward(pose + (size_t)f * 6, G, max_trans, max_rot, gp);
#pragma unroll
    for (int k = 0; k < 6; ++k) grad_pose[(size_t)f * 6 + k] += gp[k];
  }
}

// One workgroup per frame: sums the per-ray rows of its frame (no atomics: gfx950 atomics serialise per 64-byte line and
// all frames' 12-vectors share a handful of lines), then lane 0 runs the SE(3) backward; frame-feature gradients likewise.
__global__ __launch_bounds__(256) void k_pose_reduce_bwd(const float* __restrict__ pose, const float* __restrict__ g_ray,
                                                          float* __restrict__ dview, const float* __restrict__ batch,
                                                          int64_t R, int ff, float max_trans, float max_rot,
                                                          float* __restrict__ grad_pose, float* __restrict__ grad_feat,
                                                          float* __restrict__ g_delta, int zero_dview,
                                                          float* __restrict__ slots) {
  __shared__ float sm[4][32];
  const int f = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (slots != nullptr) {
    // slot mode: k_pose_grad_accum left NOF_POSE_SLOTS partial sums per frame; add them in slot order, hand them back zeroed
    if (threadIdx.x < NOF_POSE_SLOT_W) {
      float* sl = slots + (size_t)f * NOF_POSE_SLOTS * NOF_POSE_SLOT_W + threadIdx.x;
      float v[NOF_POSE_SLOTS];
#pragma unroll
      for (int k = 0; k < NOF_POSE_SLOTS; ++k) v[k] = sl[k * NOF_POSE_SLOT_W];
      float t = 0.0f;
#pragma unroll
      for (int k = 0; k < NOF_POSE_SLOTS; ++k) {
        t += v[k];
        if (v[k] != 0.0f) sl[k * NOF_POSE_SLOT_W] = 0.0f;
      }
      sm[0][threadIdx.x] = t;
    }
    __syncthreads();
    pose_frame_epilogue(sm[0], f, ff, pose, max_trans, max_rot, grad_pose, grad_feat, g_delta);
    return;
  }
  float acc[12 + NOF_VIEW_COLS];
#pragma unroll
  for (int k = 0; k < 12 + NOF_VIEW_COLS; ++k) acc[k] = 0.0f;
  // the frame ids of 16 rays per thread are requested together (one ray at a time the loop waited for each id: 25 us at 4096 rays);
  // only the ~R/F rays of this frame then read their rows
  for (int64_t r0 = threadIdx.x; r0 < R; r0 += 16 * 256) {
    int fr[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int64_t r = r0 + 256 * u;
      fr[u] = r < R ? (int)batch[r * NOF_RAY_COLS + 8] : -1;
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (fr[u] != f) continue;
      const int64_t r = r0 + 256 * u;
      if (g_ray) {
        const float4* g4 = reinterpret_cast<const float4*>(g_ray + r * 12);
        const float4 a = g4[0], b = g4[1], c = g4[2];
        acc[0] += a.x; acc[1] += a.y; acc[2] += a.z; acc[3] += a.w;
        acc[4] += b.x; acc[5] += b.y; acc[6] += b.z; acc[7] += b.w;
        acc[8] += c.x; acc[9] += c.y; acc[10] += c.z; acc[11] += c.w;
      }
      for (int k = 0; k < ff; ++k) acc[12 + k] += dview[r * NOF_VIEW_COLS + k];
      if (zero_dview) {                                                 // this kernel is the last reader of the row: ready for the next
        float4* d4 = reinterpret_cast<float4*>(dview + r * NOF_VIEW_COLS);      // step's atomics (every ray belongs to one frame)
        d4[0] = d4[1] = d4[2] = d4[3] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 12 + NOF_VIEW_COLS; ++k) {
    const float t = wave_sum_p(acc[k]);
    if (lane == 0) sm[wave][k] = t;
  }
  __syncthreads();
  if (threadIdx.x < 12 + NOF_VIEW_COLS) {
    const int k = threadIdx.x;
    sm[0][k] = (sm[0][k] + sm[1][k]) + (sm[2][k] + sm[3][k]);
  }
  __syncthreads();
  pose_frame_epilogue(sm[0], f, ff, pose, max_trans, max_rot, grad_pose, grad_feat, g_delta);
}

extern "C" int nof_pose_fwd(const float* pose_data, const float* c2w, float max_trans, float max_rot_rad, float* tf,
                             int32_t F, void* stream) {
  NOF_ARG(c2w && tf && F >= 0);
  if (F == 0) return 0;
  hipLaunchKernelGGL(k_pose_fwd, dim3((unsigned)nof_div_up(F, 64)), dim3(64), 0, (hipStream_t)stream, pose_data, c2w,
                     max_trans, max_rot_rad, tf, F);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_pose_bwd(const float* pose_data, const float* g_delta, float max_trans, float max_rot_rad,
                             float* grad_pose, int32_t F, void* stream) {
  NOF_ARG(pose_data && g_delta && grad_pose && F >= 0);
  if (F == 0) return 0;
  hipLaunchKernelGGL(k_pose_bwd, dim3((unsigned)nof_div_up(F, 64)), dim3(64), 0, (hipStream_t)stream, pose_data, g_delta,
                     max_trans, max_rot_rad, grad_pose, F);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_pose_grad_accum(const float* dpts, float* dview, const float* batch, const float* z_vals,
                                    const float* c2w, const float* tf, int32_t ff, int32_t sh_degree, int64_t R, int32_t S,
                                    float* g_ray, float* frame_slots, void* stream) {
  NOF_ARG(batch && z_vals && c2w && tf && g_ray && R >= 0 && S >= 1 && ff >= 0 && ff <= NOF_VIEW_COLS);
  NOF_ARG(sh_degree >= 1 && sh_degree <= 3 && (frame_slots == nullptr || dview != nullptr));
  static_assert(NOF_POSE_SLOT_W == 12 + NOF_VIEW_COLS && (NOF_POSE_SLOTS & (NOF_POSE_SLOTS - 1)) == 0, "slot layout");
  if (R == 0) return 0;
  hipLaunchKernelGGL(k_pose_grad_accum, dim3((unsigned)R), dim3(64), 0, (hipStream_t)stream, dpts, dview, batch, z_vals,
                     c2w, tf, ff, sh_degree, R, S, g_ray, frame_slots);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_pose_reduce_bwd(const float* pose_data, const float* g_ray, float* dview, const float* batch,
                                    int64_t R, int32_t ff, float max_trans, float max_rot_rad, float* grad_pose,
                                    float* grad_feat, float* g_delta, int32_t F, int32_t zero_dview, float* frame_slots,
                                    void* stream) {
  NOF_ARG(R >= 0 && F >= 0 && ff >= 0 && ff <= NOF_VIEW_COLS);
  if (frame_slots == nullptr) {
    NOF_ARG(batch && (!zero_dview || dview));
    NOF_ARG((ff == 0 || grad_feat == nullptr || dview != nullptr) && (grad_pose == nullptr || (pose_data && g_ray)));
  } else {
    NOF_ARG(grad_pose == nullptr || pose_data);
  }
  if (F == 0) return 0;
  hipLaunchKernelGGL(k_pose_reduce_bwd, dim3((unsigned)F), dim3(256), 0, (hipStream_t)stream, pose_data, g_ray, dview,
                     batch, R, ff, max_trans, max_rot_rad, grad_pose, grad_feat, g_delta, (int)zero_dview, frame_slots);
  NOF_LAUNCH_OK();
  return 0;
}
