// The optimiser launch of a single-GPU training step (torch.optim.Adam, nerf_runner.py:502,756-761) with the step's two small
// neighbours inside: the per-frame pose gradient sums (PoseArray's autograd, nerf_helpers.py:143-154) and the next step's MFMA
// operand image + pose table (what nof_mlp_pack_pose builds).  Its own translation unit: nof_mlp.hip is compiled with
// -fno-honor-nans (bundlesdf_amd/build.py), and this kernel must SEE the NaNs of an overflowed 16-bit backward to zero them.
#include "nof_adam_tail_dev.h"

template <class P>
__global__ __launch_bounds__(256) void k_adam_tail(NofMlpDesc d, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, AdamK k, const int32_t* __restrict__ skip_flags, TailArgs a) {
  adam_tail_roles<P>(d, p, g, m, v, k, skip_flags, a, blockIdx.x, gridDim.x);
}

// The same with the step's scalars in device memory (a captured, replayable step): every workgroup reads the NofStepState when it
// starts; the LAST one to finish -- counters in `done`, left at zero -- advances the state to the next optimiser step, which was
// the launch nof_step_state_advance behind nof_adam_step_dyn.
template <class P>
__global__ __launch_bounds__(256) void k_adam_tail_dyn(NofMlpDesc d, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                        float* __restrict__ v, NofStepState* st, float b1, float b2, float eps,
                                                        const int32_t* __restrict__ skip_flags, TailArgs a, float lrate,
                                                        float lrate_pose, float decay_rate, int n_iters, uint32_t* __restrict__ done) {
  const AdamK k{st->step_basic, st->step_pose, b1, b2, eps, st->inv_sqrt_bc2};
  adam_tail_roles<P>(d, p, g, m, v, k, skip_flags, a, blockIdx.x, gridDim.x);
  // No fence: a workgroup's reads of *st have returned before anything that depends on them was stored, and that is all the
  // advance has to wait for (an agent-scope release here writes the XCD's L2 back once per workgroup: 242 instead of 37 us).
  // Two levels of counters, each on its own 64-byte line: 4000 atomics on ONE line would serialise at 12 ns each.
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t G = gridDim.x, grp = blockIdx.x & 63u;
    const uint32_t members = (G - grp + 63u) / 64u, groups = G < 64u ? G : 64u;
    uint32_t* c = done + (1u + grp) * 16u;
    if (atomicAdd(c, 1u) == members - 1u) {
      *c = 0u;
      if (atomicAdd(done, 1u) == groups - 1u) {                        // every other workgroup has read the state and left
        *done = 0u;
        step_state_advance(st, lrate, lrate_pose, decay_rate, n_iters, b1, b2, -1);
      }
    }
  }
}

extern "C" int nof_adam_step_tail(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t n_basic,
                                   float lr, float lr_pose, float beta1, float beta2, float eps, int32_t step,
                                   const int32_t* skip_flags, const NofAdamTail* t, void* stream) {
  TailArgs a;
  dim3 grid;
  NOF_ARG(step >= 1);
  if (int e = adam_tail_check(params, grads, exp_avg, exp_avg_sq, n, n_basic, t, &a, &grid)) return e;
  const NofMlpDesc& d = *t->desc;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const AdamK k{(float)(lr / bc1), (float)(lr_pose / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2))};
#define NOF_TAIL(P) hipLaunchKernelGGL(k_adam_tail<P>, grid, dim3(256), 0, (hipStream_t)stream, d, params, grads, exp_avg, exp_avg_sq, k, skip_flags, a)
  if (d.precision == 0) NOF_TAIL(PrecF32);
  else if (is_bf16(d.precision)) NOF_TAIL(PrecBF16);
  else NOF_TAIL(PrecF16);
#undef NOF_TAIL
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_adam_step_tail_dyn(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t n_basic,
                                       NofStepState* d_state, float lrate, float lrate_pose, float decay_rate, int32_t n_iters,
                                       float beta1, float beta2, float eps, const int32_t* skip_flags, const NofAdamTail* t,
                                       uint32_t* d_done, void* stream) {
  TailArgs a;
  dim3 grid;
  NOF_ARG(d_state && d_done && n_iters > 0);
  if (int e = adam_tail_check(params, grads, exp_avg, exp_avg_sq, n, n_basic, t, &a, &grid)) return e;
  const NofMlpDesc& d = *t->desc;
#define NOF_TAIL(P)                                                                                                              \
  hipLaunchKernelGGL(k_adam_tail_dyn<P>, grid, dim3(256), 0, (hipStream_t)stream, d, params, grads, exp_avg, exp_avg_sq, d_state, \
                     beta1, beta2, eps, skip_flags, a, lrate, lrate_pose, decay_rate, (int)n_iters, d_done)
  if (d.precision == 0) NOF_TAIL(PrecF32);
  else if (is_bf16(d.precision)) NOF_TAIL(PrecBF16);
  else NOF_TAIL(PrecF16);
#undef NOF_TAIL
  NOF_LAUNCH_OK();
  return 0;
}
