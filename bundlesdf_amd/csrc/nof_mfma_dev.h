// The matrix-core operand types of the MLP kernels: one struct per arithmetic type with the fragment type, the packing of eight
// (or one) float values into a fragment and the 32x32 MFMA, and the accumulator's row map.  A header because the test-only probe
// library (csrc/test/nof_probe.hip: one raw tile through exactly these definitions) shares it with nof_mlp.hip.
#pragma once
#include "nof_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

struct PrecF32 {
  static constexpr int KR = 1;
  typedef float elem;
  typedef float frag;
  static __device__ __forceinline__ frag pack(const float* v) { return v[0]; }
  static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
};
struct PrecBF16 {
  static constexpr int KR = 8;
  typedef __bf16 elem;
  typedef bf16x8 frag;
  static __device__ __forceinline__ frag pack(const float* v) {
    frag f;
#pragma unroll
    for (int t = 0; t < 8; ++t) f[t] = (__bf16)v[t];
    return f;
  }
  static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
};
struct PrecF16 {
  static constexpr int KR = 8;
  typedef _Float16 elem;
  typedef f16x8 frag;
  static __device__ __forceinline__ frag pack(const float* v) {
    frag f;
#pragma unroll
    for (int t = 0; t < 8; ++t) f[t] = (_Float16)v[t];
    return f;
  }
  static __device__ __forceinline__ f32x16 mma(frag a, frag b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

__host__ __device__ __forceinline__ constexpr int nloc(int hi, int r) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
