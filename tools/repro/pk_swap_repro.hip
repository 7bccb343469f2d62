// Stand-alone reproducer of the quarter-wave fault behind DESIGN 2.10 (gfx950 / MI355X, ROCm 7.2).  A packed fp32 VALU instruction
// whose SOURCE 1 is read through op_sel = 1 (its HIGH register feeds the LOW result), e.g.
//     v_pk_mul_f32 v[a:a+1], v[a:a+1], v[b:b+1] op_sel:[0,1]
// returns -- now and then, in lanes 48-63 only, and only while other waves of the SIMD issue MFMAs -- a result computed from the wrong
// half.  clang's SLP vectoriser emits such forms (that is how the fused forward met it); this file issues them as inline assembly.
// Every wave of a 12-wave workgroup alternates 16 executions of FORM on constant operands (counting, per lane, the executions whose
// result differs from the scalar arithmetic the instruction is defined as) with a chain of `mfma_iters` MFMAs, the three waves of a
// SIMD staggered (mfma_iters = 0: no MFMA is issued at all).
//     hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -shared -fPIC pk_swap_repro.hip -o libpk_swap.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define PK(text) asm volatile(text : "+v"(t) : "v"(b), "v"(c))

template <int FORM>
__global__ __launch_bounds__(768) void k_pk(int32_t* __restrict__ counts, float* __restrict__ sink, int iters, int mfma_iters, int pad,
                                             int32_t* __restrict__ by_index) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f16x8 ma, mb;
#pragma unroll
  for (int r = 0; r < 8; ++r) { ma[r] = (_Float16)(0.01f * (float)(lane + r)); mb[r] = (_Float16)(0.02f * (float)(lane - r)); }
  f32x16 acc = {0};
  const v2f a = {1.5f + 0.001f * (float)lane, 2.5f - 0.0007f * (float)lane};
  // source 1 differs from wave to wave: a result computed from ANOTHER wave's register is recognisable (see the driver)
  const v2f b = {3.25f + 0.01f * (float)(lane & 7), 0.75f - 0.01f * (float)(lane & 3) + 0.03125f * (float)(wave + 1)};
  const v2f c = {0.125f, 8.0f};
  v2f want;
  switch (FORM) {
    case 0: case 6: want = {a.x * b.y, a.y * b.y}; break;             // mul  op_sel:[0,1]
    case 1: want = {a.x * b.y, a.y * b.x}; break;                     // mul  op_sel:[0,1] op_sel_hi:[1,0]
    case 2: want = {a.x + b.y, a.y + b.y}; break;                     // add  op_sel:[0,1]
    case 3: want = {fmaf(a.x, b.y, c.x), fmaf(a.y, b.y, c.y)}; break; // fma  op_sel:[0,1,0]
    case 4: want = {a.y * b.x, a.y * b.y}; break;                     // mul  op_sel:[1,0]  (source 0)
    case 7: case 8: break;                                            // v_fma_mixlo_f16 / v_fma_mixhi_f16 (below)
    default: want = {a.x * b.x, a.y * b.y}; break;                    // mul, no op_sel
  }
  // forms 7 / 8 (round 6): v_fma_mixlo_f16 / v_fma_mixhi_f16 with an f16 source 0 -- what the operand split of the MLP forward
  // compiles to for its residual: lo = f16(float(hi) * (-1) + x), written to the low / high half of the destination
  const _Float16 hs = (_Float16)(1.5f + 0.001f * (float)lane + 0.03125f * (float)(wave + 1));
  const uint32_t hreg = (uint32_t)__builtin_bit_cast(uint16_t, hs);
  float negone = -1.0f;
  asm volatile("" : "+v"(negone));
  if (FORM == 7 || FORM == 8) want = {(float)(_Float16)fmaf((float)hs, -1.0f, b.x), a.y};
  int bad = 0, bad_lo = 0, bad_hi = 0;
  uint32_t which = 0;                                                  // bit i: execution i after an MFMA chain was wrong at least once
  float first_lo = 0.0f;
  // every wave alternates a phase of packed instructions and a phase of MFMAs; the waves of a SIMD (w, w + 4, w + 8) start staggered
  for (int m = 0; m < (wave >> 2) * (mfma_iters / 3); ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ma, mb, acc, 0, 0, 0);
  for (int outer = 0; outer < iters; ++outer) {
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
      v2f t = a;
      asm volatile("" : "+v"(t));
      if (FORM == 0) PK("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1]");
      if (FORM == 1) PK("v_pk_mul_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]");
      if (FORM == 2) PK("v_pk_add_f32 %0, %0, %1 op_sel:[0,1]");
      if (FORM == 3) PK("v_pk_fma_f32 %0, %0, %1, %2 op_sel:[0,1,0]");
      if (FORM == 4) PK("v_pk_mul_f32 %0, %0, %1 op_sel:[1,0]");
      if (FORM == 5) PK("v_pk_mul_f32 %0, %0, %1");
      if (FORM == 6) { v2f d; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=&v"(d) : "v"(t), "v"(b)); t = d; }
      if (FORM == 7) { uint32_t d = 0u; asm volatile("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hreg), "v"(negone), "v"(b.x));
                       t.x = (float)__builtin_bit_cast(_Float16, (uint16_t)(d & 0xffffu)); }
      if (FORM == 8) { uint32_t d = 0u; asm volatile("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "+v"(d) : "v"(hreg), "v"(negone), "v"(b.x));
                       t.x = (float)__builtin_bit_cast(_Float16, (uint16_t)(d >> 16)); }
      const bool wl = t.x != want.x, wh = t.y != want.y;
      if (wl && bad_lo == 0) first_lo = t.x;
      bad += wl || wh; bad_lo += wl; bad_hi += wh;
      which |= (wl || wh) ? (1u << i) : 0u;
    }
    for (int m = 0; m < mfma_iters; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ma, mb, acc, 0, 0, 0);
    for (int k = 0; k < pad; ++k) asm volatile("s_nop 15");             // idle cycles between the chain and the next packed instruction
  }
  if (acc[3] == 123.456f) sink[threadIdx.x] = acc[0];                 // (keeps the chains alive)
  int32_t* o = counts + (((int64_t)blockIdx.x * 12 + wave) * 64 + lane) * 4;
  o[0] = bad; o[1] = bad_lo; o[2] = bad_hi; o[3] = __float_as_int(first_lo);
  by_index[((int64_t)blockIdx.x * 12 + wave) * 64 + lane] = (int32_t)which;
}

extern "C" int pk_run(int form, int blocks, int32_t* counts, float* sink, int iters, int mfma_iters, int pad, int32_t* by_index, void* stream) {
#define GO(F) case F: hipLaunchKernelGGL(k_pk<F>, dim3(blocks), dim3(768), 0, (hipStream_t)stream, counts, sink, iters, mfma_iters, pad, by_index); break;
  switch (form) { GO(0) GO(1) GO(2) GO(3) GO(4) GO(5) GO(6) GO(7) GO(8) default: return -1; }
  return (int)hipGetLastError();
}
