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
