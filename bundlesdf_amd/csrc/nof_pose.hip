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

// dL/dDelta[:3,:4] (row-major 12) -> dL/dxi (6) for one frame
__device__ void se3_backward(const float* xi, const float* G, float max_trans, float max_rot, float* gp) {
  Se3 s;
  se3_forward(xi, max_trans, max_rot, s);
  float GR[9], Gt[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
#pragma unroll
    for (int j = 0; j < 3; ++j) GR[i * 3 + j] = G[i * 4 + j];
    Gt[i] = G[i * 4 + 3];
  }
  float gu[3], GV[9];
#pragma unroll
  for (int j = 0; j < 3; ++j) gu[j] = (s.V[j] * Gt[0] + s.V[3 + j] * Gt[1]) + s.V[6 + j] * Gt[2];   // V^T Gt
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) GV[i * 3 + j] = Gt[i] * s.u[j];
  float gA = 0.f, gB = 0.f, gC = 0.f;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    gA += GR[k] * s.K[k];
    gB += GR[k] * s.K2[k] + GV[k] * s.K[k];
    gC += GV[k] * s.K2[k];
  }
  float GK2[9], GK[9], Kt[9], t1[9], t2[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) GK2[k] = s.Bc * GR[k] + s.Cc * GV[k];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) Kt[i * 3 + j] = s.K[j * 3 + i];
  mat3mul(GK2, Kt, t1);                                               // d(K K)/dK : G Kt + Kt G
  mat3mul(Kt, GK2, t2);
#pragma unroll
  for (int k = 0; k < 9; ++k) GK[k] = (s.A * GR[k] + s.Bc * GV[k]) + (t1[k] + t2[k]);
  float gw[3] = {GK[7] - GK[5], GK[2] - GK[6], GK[3] - GK[1]};
  if (!s.clamped) {
    const float th = s.th, sn = sinf(th), cs = cosf(th);
    const float dA = (th * cs - sn) / (th * th);
    const float dB = (th * sn - 2.0f * (1.0f - cs)) / (th * th * th);
    const float dC = ((1.0f - cs) * th - 3.0f * (th - sn)) / (th * th * th * th);
    const float gth = (gA * dA + gB * dB) + gC * dC;
#pragma unroll
    for (int k = 0; k < 3; ++k) gw[k] += gth * s.w[k] / th;
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    gp[k] = gu[k] * max_trans * (1.0f - s.tanhv[k] * s.tanhv[k]);
    gp[3 + k] = gw[k] * max_rot * (1.0f - s.tanhv[3 + k] * s.tanhv[3 + k]);
  }
}

__global__ void k_pose_bwd(const float* __restrict__ pose, const float* __restrict__ g_delta, float max_trans,
                           float max_rot, float* __restrict__ grad_pose, int F) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= F || f == 0) return;
  float gp[6];
  se3_backward(pose + (size_t)f * 6, g_delta + (size_t)f * 12, max_trans, max_rot, gp);
#pragma unroll
  for (int k = 0; k < 6; ++k) grad_pose[(size_t)f * 6 + k] += gp[k];
}

__device__ __forceinline__ float wave_sum_p(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one wave per ray: G += g (x) [q,1] over the ray's samples, q = c2w (rays_d z) (the point BEFORE the correction),
// plus the view-direction path through the SH Jacobian.
__global__ __launch_bounds__(64) void k_pose_grad_accum(const float* __restrict__ dpts, float* __restrict__ dview,
                                                         const float* __restrict__ batch, const float* __restrict__ z_vals,
                                                         const float* __restrict__ c2w, const float* __restrict__ tf, int ff,
                                                         int sh_degree, int64_t R, int S, float* __restrict__ g_ray,
                                                         float* __restrict__ slots) {
  const int64_t r = blockIdx.x;
  const int lane = threadIdx.x;
  const float* row = batch + r * NOF_RAY_COLS;
  const int f = (int)row[8];
  // the frame-feature gradient of the ray (slot mode): lanes 12 .. 12 + ff - 1
  float fpart = 0.0f;
  if (slots != nullptr && lane >= 12 && lane < 12 + ff) fpart = dview[r * NOF_VIEW_COLS + (lane - 12)];
  float mine = 0.0f;                                                    // lane k keeps component k: one coalesced 48-byte store
  if (f != 0) {                                                         // frame 0 carries no correction: its rows stay 0
    const float* M = c2w + (size_t)f * 16;
    const float dx = row[0], dy = row[1], dz = row[2];
    float G[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) G[k] = 0.0f;
    if (dpts != nullptr) {
      for (int s = lane; s < S; s += 64) {
        const int64_t b = r * S + s;
        const float z = z_vals[b];
        const float px = dx * z, py = dy * z, pz = dz * z;
        float q[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) q[k] = ((M[k * 4] * px + M[k * 4 + 1] * py) + M[k * 4 + 2] * pz) + M[k * 4 + 3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const float g = dpts[b * 3 + i];
          G[i * 4 + 0] += g * q[0];
          G[i * 4 + 1] += g * q[1];
          G[i * 4 + 2] += g * q[2];
          G[i * 4 + 3] += g;
        }
      }
    }
    if (lane == 0 && dview != nullptr && sh_degree > 1) {
      // world view dir d = tf_R v ; dL/dd through SH (nerf_helpers.py:72-85), then dL/dDelta_R += g (x) (c2w_R v)
      const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
      const float v[3] = {dx / nrm, dy / nrm, dz / nrm};
      const float* T = tf + (size_t)f * 12;
      float d[3], cv[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        d[k] = (T[k * 4] * v[0] + T[k * 4 + 1] * v[1]) + T[k * 4 + 2] * v[2];
        cv[k] = (M[k * 4] * v[0] + M[k * 4 + 1] * v[1]) + M[k * 4 + 2] * v[2];
      }
      const float* gs = dview + r * NOF_VIEW_COLS + ff;
      const float x = d[0], y = d[1], z = d[2];
      const float C1 = 0.4886025119029199f;
      float gx = -C1 * gs[3], gy = -C1 * gs[1], gz = C1 * gs[2];
      if (sh_degree > 2) {
        const float a0 = 1.0925484305920792f, a1 = -1.0925484305920792f, a2 = 0.31539156525252005f,
                    a3 = -1.0925484305920792f, a4 = 0.5462742152960396f;
        gx += gs[4] * a0 * y + gs[6] * a2 * (-2.0f * x) + gs[7] * a3 * z + gs[8] * a4 * (2.0f * x);
        gy += gs[4] * a0 * x + gs[5] * a1 * z + gs[6] * a2 * (-2.0f * y) + gs[8] * a4 * (-2.0f * y);
        gz += gs[5] * a1 * y + gs[6] * a2 * (4.0f * z) + gs[7] * a3 * x;
      }
      const float gd[3] = {gx, gy, gz};
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        G[i * 4 + 0] += gd[i] * cv[0];
        G[i * 4 + 1] += gd[i] * cv[1];
        G[i * 4 + 2] += gd[i] * cv[2];
      }
    }
#pragma unroll
    for (int k = 0; k < 12; ++k) {
      const float t = wave_sum_p(G[k]);
      if (lane == k) mine = t;
    }
  }
  if (lane < 12) g_ray[r * 12 + lane] = mine;
  if (slots != nullptr) {
    // one atomic instruction per ray into slot (frame, ray % 16): 28 consecutive floats = two 64-byte lines; a slot collects
    // R / (16 F) rays (4 at 4096 rays and 64 frames), so the same-line serialisation of the memory-side atomics stays short
    const float v = lane < 12 ? mine : fpart;
    float* slot = slots + ((size_t)f * NOF_POSE_SLOTS + (size_t)(r & (NOF_POSE_SLOTS - 1))) * NOF_POSE_SLOT_W;
    if (lane < 12 + ff && v != 0.0f) atomicAdd(slot + lane, v);
    // every value of the row has been consumed above (`mine` / `fpart` depend on the loads): ready for the next step's atomics
    if (lane < NOF_VIEW_COLS) dview[r * NOF_VIEW_COLS + lane] = 0.0f;
  }
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
