// Device side of nof_adam_step_tail (nof_adam_tail.hip) and nof_adam_step_tail_march (nof_trace.hip): the roles of the optimiser
// launch of a single-GPU training step.  NOT for translation units compiled with -fno-honor-nans (nof_mlp.hip): the skip path must
// see the NaNs of an overflowed backward to zero them.
#pragma once
#include "nof_mlp_dev.h"
#include "nof_adam_dev.h"
#include "nof_pose_dev.h"

// =====================================================================================================
// nof_adam_step_tail (round 6): the optimiser launch of a training step that also does what the step's neighbours did in two
// launches of their own -- the per-frame pose gradient sums in front of it (nof_pose_reduce_bwd, slot mode) and the NEXT step's
// prologue behind it (nof_mlp_pack_pose: MFMA operand image + pose table).  Nothing in those depends on more than ONE entry's
// update, so they are roles of the Adam launch and need no ordering inside it:
//   * workgroup f < F: adds frame f's partial sums (the slots nof_pose_grad_accum left), runs the SE(3) backward, updates the
//     frame's six pose entries and writes its row of the pose table tf from the updated values;
//   * the next `mlp_blocks` workgroups walk the FORWARD operand image element by element (the image is a permutation of the weight
//     matrices + structural zeros: every weight is met exactly once), update the weight an element holds and store its new value at
//     the element itself, at its rounding residual (split precisions) and at its place in the BACKWARD image -- the inverse of
//     k_mlp_pack's second mapping, both lane maps being nloc(); then the biases.  The structural zeros are never touched (the image
//     must have been packed once by nof_mlp_pack);
//   * the rest: the table entries [0, mlp_off) as in nof_adam_step.
// Same bits as nof_pose_reduce_bwd + nof_adam_step + nof_mlp_pack_pose (tests/test_gpu_step.py).
// =====================================================================================================
struct TailArgs {                                                     // (by value: the neighbours' share of the kernel arguments)
  char* image;
  int with_lo, mlp_blocks, F;
  int64_t mlp_off, pose_off;
  const float* c2w;
  float max_trans, max_rot;
  float* tf;
  float* slots;
};

template <class P>
__device__ __forceinline__ void adam_tail_roles(const NofMlpDesc& d, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                float* __restrict__ v, const AdamK& k, const int32_t* __restrict__ skip_flags,
                                                const TailArgs& a, const uint32_t bx, const uint32_t nb) {
  // (bx of nb: the workgroup's number among the launch's workgroups that play these roles -- blockIdx.x of gridDim.x in the plain
  //  launches, less the ray marcher's workgroups in nof_adam_step_tail_march)
  char* __restrict__ image = a.image;
  const int with_lo = a.with_lo, mlp_blocks = a.mlp_blocks, F = a.F;
  const int64_t mlp_off = a.mlp_off, pose_off = a.pose_off;
  const float* __restrict__ c2w = a.c2w;
  const float max_trans = a.max_trans, max_rot = a.max_rot;
  float* __restrict__ tf = a.tf;
  float* __restrict__ slots = a.slots;
  const bool skip = skip_flags != nullptr && (skip_flags[0] & 4);     // (uniform) this step's gradient is not finite: no update
  if ((int)bx < F) {                                           // ---- one frame's pose entries (workgroup-uniform) ----
    __shared__ float sm[32];
    const int f = (int)bx;
    if (threadIdx.x < NOF_POSE_SLOT_W) {                                // (k_pose_reduce_bwd's slot mode: slot order, handed back zeroed)
      float* sl = slots + (size_t)f * NOF_POSE_SLOTS * NOF_POSE_SLOT_W + threadIdx.x;
      float q[NOF_POSE_SLOTS];
#pragma unroll
      for (int u = 0; u < NOF_POSE_SLOTS; ++u) q[u] = sl[u * NOF_POSE_SLOT_W];
      float t = 0.0f;
#pragma unroll
      for (int u = 0; u < NOF_POSE_SLOTS; ++u) {
        t += q[u];
        if (q[u] != 0.0f) sl[u * NOF_POSE_SLOT_W] = 0.0f;
      }
      sm[threadIdx.x] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      const size_t at = (size_t)pose_off + (size_t)f * 6;
      float xi[6], gr[6];
#pragma unroll
      for (int u = 0; u < 6; ++u) { xi[u] = p[at + u]; gr[u] = g[at + u]; }
      if (f != 0) {
        float G[12], gp[6];
#pragma unroll
        for (int u = 0; u < 12; ++u) G[u] = sm[u];
        se3_backward(xi, G, max_trans, max_rot, gp);
#pragma unroll
        for (int u = 0; u < 6; ++u) gr[u] += gp[u];
      }
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        if (!skip) {
          float mm = m[at + u], vv = v[at + u];
          adam_one(xi[u], gr[u], mm, vv, k.step_pose, k);
          p[at + u] = xi[u]; m[at + u] = mm; v[at + u] = vv;
        }
        g[at + u] = 0.0f;
      }
      pose_fwd_frame_xi(f, xi, c2w, max_trans, max_rot, tf);
    }
    return;
  }
  const int bid = (int)bx - F;
  if (bid < mlp_blocks) {                                              // ---- the MLP entries and their operand image ----
    constexpr int KR = P::KR;
    typedef typename P::elem elem;
    float* pm = p + mlp_off; float* gm = g + mlp_off; float* mm_ = m + mlp_off; float* vm = v + mlp_off;
    const int n_layers = d.n_sigma + d.n_color;
    const int npair = pair_base(d, n_layers);
    elem* fw = (elem*)image;
    elem* bw = fw + (size_t)npair * 16 * 64;
    float* bias = (float*)(bw + (size_t)npair * 16 * 64);
    elem* fw_lo = (elem*)(bias + (size_t)oblk_base(d, n_layers) * 32);
    const int total = npair * 16 * 64;
    for (int e = bid * blockDim.x + threadIdx.x; e < total; e += mlp_blocks * blockDim.x) {
      const int lane = e & 63, r = (e >> 6) & 15;
      int pair = e >> 10, l = 0;
      for (;; ++l) {
        const int cnt = lay_pn(d, l) * lay_qn(d, l);
        if (pair < cnt) break;
        pair -= cnt;
      }
      const int qn = lay_qn(d, l), pn = lay_pn(d, l), base = pair_base(d, l);
      const int hi = lane >> 5, i = lane & 31;
      const int pb = pair / qn, q = pair % qn;
      const int row = 32 * pb + i, col = inmap(d, l, q, hi, r);
      if (!(row < d.out_dim[l] && col >= 0)) continue;                  // a structural zero of the image
      const size_t idx = (size_t)d.w_off[l] + (size_t)row * d.in_dim[l] + col;
      float gv = gm[idx];
      if (skip) {
        if (__float_as_uint(gv) != 0u) gm[idx] = 0.0f;               // (bit pattern: a NaN gradient is what gets here)
        continue;
      }
      float pv = pm[idx], mv = mm_[idx], vv = vm[idx];
      adam_one(pv, gv, mv, vv, k.step_basic, k);
      pm[idx] = pv; mm_[idx] = mv; vm[idx] = vv; gm[idx] = 0.0f;
      const size_t at = (((size_t)(base + pb * qn + q) * (16 / KR) + r / KR) * 64 + lane) * KR + r % KR;
      fw[at] = (elem)pv;
      if (with_lo) fw_lo[at] = (elem)(pv - (float)(elem)pv);
      // the backward image holds the same weight at pair (q, pb), lane (hi', i') with nloc(hi', r') = i (the row inside its block)
      // and i' = nloc(hi, r) (the input slot): k_mlp_pack's second mapping, inverted
      const int hb = (i >> 2) & 1, rb = (i & 3) + 4 * (i >> 3), lb = 32 * hb + nloc(hi, r);
      bw[(((size_t)(base + q * pn + pb) * (16 / KR) + rb / KR) * 64 + lb) * KR + rb % KR] = (elem)pv;
    }
    const int nob = oblk_base(d, n_layers);
    for (int e = bid * blockDim.x + threadIdx.x; e < nob * 32; e += mlp_blocks * blockDim.x) {
      int ob = e >> 5, l = 0;
      for (;; ++l) {
        const int pn = lay_pn(d, l);
        if (ob < pn) break;
        ob -= pn;
      }
      const int row = 32 * ob + (e & 31);
      if (row >= d.out_dim[l]) continue;
      const size_t idx = (size_t)d.b_off[l] + row;
      float gv = gm[idx];
      if (skip) {
        if (__float_as_uint(gv) != 0u) gm[idx] = 0.0f;               // (bit pattern: a NaN gradient is what gets here)
        continue;
      }
      float pv = pm[idx], mv = mm_[idx], vv = vm[idx];
      adam_one(pv, gv, mv, vv, k.step_basic, k);
      pm[idx] = pv; mm_[idx] = mv; vm[idx] = vv; gm[idx] = 0.0f;
      bias[e] = pv;
    }
    return;
  }
  adam_range(p, g, m, v, mlp_off, mlp_off, k, skip_flags, (uint32_t)(bid - mlp_blocks), nb - (uint32_t)F - (uint32_t)mlp_blocks);
}

static int adam_tail_check(const float* params, const float* grads, const float* exp_avg, const float* exp_avg_sq, int64_t n,
                           int64_t n_basic, const NofAdamTail* t, TailArgs* a, dim3* grid) {
  NOF_ARG(params && grads && exp_avg && exp_avg_sq && n >= 0 && n_basic >= 0 && n_basic <= n && t);
  if (int e = check_desc(t->desc)) return e;
  const NofMlpDesc& d = *t->desc;
  // the flat layout this launch understands: [table | MLP | 6 F pose entries], the learning-rate boundary in front of the poses
  NOF_ARG(t->packed && t->c2w && t->tf && t->frame_slots && t->F >= 1 && t->mlp_off >= 0 && t->n_mlp > 0);
  NOF_ARG(t->mlp_off + t->n_mlp == t->pose_off && t->pose_off == n_basic && t->pose_off + 6 * (int64_t)t->F == n);
  const int nl = d.n_sigma + d.n_color;
  NOF_ARG((int64_t)d.b_off[nl - 1] + d.out_dim[nl - 1] <= t->n_mlp);
  const int mlp_blocks = (int)nof_div_up((int64_t)n_pairs(d, nl) * 1024, 256);
  const int64_t tb = t->mlp_off > 0 ? (nof_div_up(t->mlp_off, 1024) < 4096 ? nof_div_up(t->mlp_off, 1024) : 4096) : 0;
  *grid = dim3((unsigned)(t->F + mlp_blocks + tb));
  *a = TailArgs{(char*)t->packed, is_split(d.precision) ? 1 : 0, mlp_blocks, (int)t->F, t->mlp_off, t->pose_off, t->c2w,
                t->max_trans, t->max_rot, t->tf, t->frame_slots};
  return 0;
}

