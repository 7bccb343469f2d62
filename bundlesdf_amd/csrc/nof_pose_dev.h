// Per-frame SE(3) pose correction, forward (PoseArray.get_matrices, nerf_helpers.py:143-154, + the product with c2w,
// nerf_runner.py:1051-1053): shared by nof_pose.hip (nof_pose_fwd and the backward kernels) and nof_mlp.hip (the step's
// prologue launch packs the MLP fragment image and updates the pose table in ONE launch: nof_mlp_pack_pose).
#pragma once
#include "nof_common.h"

struct Se3 {
  float R[9], V[9], K[9], K2[9];
  float th, A, Bc, Cc;
  float u[3], w[3], tanhv[6];
  bool clamped;
};

__device__ __forceinline__ void mat3mul(const float* a, const float* b, float* c) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) c[i * 3 + j] = (a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j]) + a[i * 3 + 2] * b[6 + j];
}

__device__ __forceinline__ void se3_forward(const float* xi, float max_trans, float max_rot, Se3& s) {
#pragma unroll
  for (int k = 0; k < 6; ++k) s.tanhv[k] = tanhf(xi[k]);
#pragma unroll
  for (int k = 0; k < 3; ++k) { s.u[k] = s.tanhv[k] * max_trans; s.w[k] = s.tanhv[3 + k] * max_rot; }
  const float nrm = (s.w[0] * s.w[0] + s.w[1] * s.w[1]) + s.w[2] * s.w[2];
  const float eps = 1e-4f;
  s.clamped = nrm < eps;
  s.th = sqrtf(fmaxf(nrm, eps));
  const float K[9] = {0.f, -s.w[2], s.w[1], s.w[2], 0.f, -s.w[0], -s.w[1], s.w[0], 0.f};
#pragma unroll
  for (int k = 0; k < 9; ++k) s.K[k] = K[k];
  mat3mul(s.K, s.K, s.K2);
  const float th = s.th, sn = sinf(th), cs = cosf(th);
  s.A = sn / th;
  s.Bc = (1.0f - cs) / (th * th);
  s.Cc = (th - sn) / (th * th * th);
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const float e = (k == 0 || k == 4 || k == 8) ? 1.0f : 0.0f;
    s.R[k] = (s.A * s.K[k] + s.Bc * s.K2[k]) + e;
    s.V[k] = (e + s.Bc * s.K[k]) + s.Cc * s.K2[k];
  }
}

// tf[f] = Delta(pose[f]) @ c2w[f], rows 0..2 (frame 0 is the anchor: Delta = identity, nerf_helpers.py:151-153)
__device__ __forceinline__ void pose_fwd_frame(int f, const float* __restrict__ pose, const float* __restrict__ c2w, float max_trans,
                                               float max_rot, float* __restrict__ tf) {
  const float* M = c2w + (size_t)f * 16;
  float D[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  if (pose != nullptr && f != 0) {
    Se3 s;
    se3_forward(pose + (size_t)f * 6, max_trans, max_rot, s);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) D[i * 4 + j] = s.R[i * 3 + j];
      D[i * 4 + 3] = (s.V[i * 3] * s.u[0] + s.V[i * 3 + 1] * s.u[1]) + s.V[i * 3 + 2] * s.u[2];
    }
  }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      tf[(size_t)f * 12 + i * 4 + j] =
          ((D[i * 4] * M[j] + D[i * 4 + 1] * M[4 + j]) + D[i * 4 + 2] * M[8 + j]) + D[i * 4 + 3] * M[12 + j];
}
