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

// tf[f] = Delta(xi) @ c2w[f], rows 0..2 (frame 0 is the anchor: Delta = identity, nerf_helpers.py:151-153); xi == NULL: identity
__device__ __forceinline__ void pose_fwd_frame_xi(int f, const float* xi, const float* __restrict__ c2w, float max_trans,
                                                  float max_rot, float* __restrict__ tf) {
  const float* M = c2w + (size_t)f * 16;
  float D[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  if (xi != nullptr && f != 0) {
    Se3 s;
    se3_forward(xi, max_trans, max_rot, s);
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

__device__ __forceinline__ void pose_fwd_frame(int f, const float* __restrict__ pose, const float* __restrict__ c2w, float max_trans,
                                               float max_rot, float* __restrict__ tf) {
  pose_fwd_frame_xi(f, pose != nullptr ? pose + (size_t)f * 6 : nullptr, c2w, max_trans, max_rot, tf);
}

// dL/dDelta[:3,:4] (row-major 12) -> dL/dxi (6) for one frame
__device__ __forceinline__ void se3_backward(const float* xi, const float* G, float max_trans, float max_rot, float* gp) {
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


// ---- per-ray pose / frame-feature gradient rows (k_pose_grad_accum of nof_pose.hip; also a rider of k_hash_bwd_lds) ----
__device__ __forceinline__ float wave_sum_p(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one wave per ray: G += g (x) [q,1] over the ray's samples, q = c2w (rays_d z) (the point BEFORE the correction),
// plus the view-direction path through the SH Jacobian.
__device__ __forceinline__ void pose_grad_accum_ray(const float* __restrict__ dpts, float* __restrict__ dview,
                                                    const float* __restrict__ batch, const float* __restrict__ z_vals,
                                                    const float* __restrict__ c2w, const float* __restrict__ tf, int ff,
                                                    int sh_degree, int64_t R, int S, float* __restrict__ g_ray,
                                                    float* __restrict__ slots, const int64_t r, const int lane) {
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

