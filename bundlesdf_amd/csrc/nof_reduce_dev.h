// The MLP backward's per-workgroup weight-gradient rows -> the flat gradient (nof_reduce_partials).  A device header because two
// launches carry it: k_reduce_partials on its own (nof_loss.hip), and the LDS-accumulated hash levels' launch of the training step,
// where it rides as extra workgroups (nof_hash.hip: nof_hash_encode_bwd_parts_reduce).
#pragma once
#include "nof_common.h"

// out[col] += sum over rows of partials [n_rows, n_cols] (n_rows = one per workgroup of the MLP backward = 2 x CUs = 512 on
// MI355X: 19 MB at cfg2).  A workgroup (1024 threads) owns 32 columns (one 128-byte line per row) and one of RED_RSPLIT row ranges;
// 32 row groups x 8 independent loads in flight per lane; one fp32 atomic per (column, row range) at the end.
#define RED_RSPLIT 2
__device__ __forceinline__ void reduce_partials_block(const float* __restrict__ partials, int n_rows, int n_cols,
                                                      float* __restrict__ out, int32_t* __restrict__ flags, int bx, int by) {
  __shared__ float sm[32][33];
  const int c = threadIdx.x & 31, grp = threadIdx.x >> 5;
  const int col = bx * 32 + c;
  const int per = (n_rows + RED_RSPLIT - 1) / RED_RSPLIT;
  const int r0 = by * per, r1 = r0 + per < n_rows ? r0 + per : n_rows;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col < n_cols) {
    int i = r0 + grp;
    for (; i + 32 * 7 < r1; i += 32 * 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += partials[(size_t)(i + 32 * u) * n_cols + col];
    }
    for (; i < r1; i += 32) acc[0] += partials[(size_t)i * n_cols + col];
  }
  sm[grp][c] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (grp == 0 && col < n_cols) {
    float s = 0.0f;
#pragma unroll
    for (int g = 0; g < 32; ++g) s += sm[g][c];
    atomicAdd(&out[col], s);
    // a non-finite weight gradient = an overflow inside the 16-bit backward (the reference's GradScaler would skip the step and
    // back its scale off, nerf_runner.py:756-761): raised as bit 2 of flags[0] for the host to act on, no host sync here
    if (flags != nullptr && !(fabsf(s) <= 3.0e38f)) atomicOr(&flags[0], 4);
  }
}

// The stand-alone launch of the reduction.  `static` so that every translation unit that needs it (nof_loss.hip: the exported
// nof_reduce_partials; nof_hash.hip: the fall-back of nof_hash_encode_bwd_parts_reduce when no LDS-level launch is there to ride in)
// carries its own copy: no library of the build has an undefined reference to another's entry point.
static __global__ __launch_bounds__(1024) void k_reduce_partials(const float* __restrict__ partials, int n_rows, int n_cols,
                                                                  float* __restrict__ out, int32_t* __restrict__ flags) {
  reduce_partials_block(partials, n_rows, n_cols, out, flags, (int)blockIdx.x, (int)blockIdx.y);
}

static inline int reduce_partials_launch(const float* partials, int32_t n_rows, int32_t n_cols, float* out, int32_t* flags, void* stream) {
  NOF_ARG(partials && out && n_rows >= 0 && n_cols >= 0);
  if (n_cols == 0 || n_rows == 0) return 0;
  hipLaunchKernelGGL(k_reduce_partials, dim3((unsigned)nof_div_up(n_cols, 32), RED_RSPLIT), dim3(1024), 0, (hipStream_t)stream,
                     partials, n_rows, n_cols, out, flags);
  NOF_LAUNCH_OK();
  return 0;
}
