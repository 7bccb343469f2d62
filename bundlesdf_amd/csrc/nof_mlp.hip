// SDF + colour tiny-MLPs (NeRFSmall, nerf_helpers.py:243-321) forward / backward on gfx950 matrix cores.
//
// MI355X-first design (nothing here mirrors the reference's cuBLAS-per-layer structure):
//   * one wave64 owns a tile of 32 samples; every layer is a chain of v_mfma_f32_32x32x{16 bf16|16 f16|2 f32};
//   * "swapped" orientation  D[neuron][sample] = W[neuron][k] X[k][sample]  puts one SAMPLE per lane with its 16
//     neurons of a 32-block in registers, and that accumulator layout IS the B-operand layout of the next layer
//     once the weight matrix' K columns are permuted to match -> layers chain in registers, no LDS round trip,
//     no cross-lane traffic.  The permuted weight fragments are packed once per workgroup into LDS;
//   * the backward needs sample-contracted GEMMs (dW = dY X^T).  Instead of transposing through LDS, every layer is
//     ALSO evaluated in the other orientation D[sample][neuron] (operands swapped, same fragments): that leaves one
//     NEURON per lane with 16 samples in registers = exactly the A/B operand layout of the dW MFMA.  The matrix cores
//     are ~idle in this network (53 kFLOP/sample), so spending 2x MFMAs to delete all transposes is the cheap side;
//   * dW / db accumulate in registers (AGPRs) across a persistent loop, are reduced per workgroup in LDS and written
//     as per-workgroup partial sums (deterministic, no global atomics on 9k hot addresses);
//   * forward activations are recomputed in the backward (features are re-read, 128 B/sample) - nothing [B,64]
//     ever goes to HBM.
// Accumulator layout of v_mfma_f32_32x32x*: lane l = (hi = l>>5, j = l&31) holds column j, rows (r&3)+8(r>>2)+4hi.
#include <type_traits>
#include "nof_common.h"
#include "nof_pose_dev.h"
#include "nof_hash_dev.h"

#include "nof_mfma_dev.h"

// ---- layer bookkeeping (runtime, from the descriptor) -------------------------------------------
// input blocks: hash features fit one block; colour layer 0 reads [sigma-out block | view block]; every other layer reads the
// hidden/32 blocks of the previous hidden layer.  output blocks: ceil(out/32).  (hidden = 64: qn = 2, pn in {1,2} as before.)
__host__ __device__ __forceinline__ int lay_qn(const NofMlpDesc& d, int l) { return l == 0 ? 1 : (l == d.n_sigma ? 2 : d.hidden / 32); }
__host__ __device__ __forceinline__ int lay_pn(const NofMlpDesc& d, int l) { return (d.out_dim[l] + 31) / 32; }
__host__ __device__ inline int pair_base(const NofMlpDesc& d, int l) {
  int s = 0;
  for (int k = 0; k < l; ++k) s += lay_pn(d, k) * lay_qn(d, k);
  return s;
}
__host__ __device__ inline int oblk_base(const NofMlpDesc& d, int l) {
  int s = 0;
  for (int k = 0; k < l; ++k) s += lay_pn(d, k);
  return s;
}

// weight-matrix column that input slot (q, hi, r) of layer l reads; -1 = structural zero
__device__ __forceinline__ int inmap(const NofMlpDesc& d, int l, int q, int hi, int r) {
  if (l == 0) {
    const int c = 16 * hi + r;                                       // hash features, natural order 2*level + ch
    return c < d.in_feat ? c : -1;
  }
  if (l == d.n_sigma) {                                               // colour layer 0: [views | geo_feat] (nerf_helpers.py:316)
    if (q == 0) {
      const int o = nloc(hi, r);                                      // sigma output o: 0 = sdf, 1..geo = geo_feat
      return (o >= 1 && o <= d.geo) ? d.n_view + o - 1 : -1;
    }
    // view block: column u sits in slot (hi, r) = (u >> 3, u & 7), registers r >= 8 unused.  Both lane halves then hold eight
    // columns in their first eight registers, and in the TRANSPOSED weight-gradient block of this layer (dW^T[slot][neuron], see
    // dw_block) the slot rows nloc(hi, r), r < 8, are the only non-zero ones: 8 accumulator registers per block instead of 16,
    // like the sigma-out block beside it (o = nloc(hi, r) <= 15 <=> r < 8).
    const int u = 8 * hi + r;
    return (r < 8 && u < d.n_view) ? u : -1;
  }
  const int c = 32 * q + nloc(hi, r);
  return c < d.in_dim[l] ? c : -1;
}
// Packs the fp32 PyTorch-layout weights into the MFMA fragment image the kernels keep in LDS (once per optimiser step,
// by one small launch; every workgroup of the fwd/bwd kernels then just streams the image into LDS with 16-byte copies
// -- packing inside each workgroup cost ~60 us of dependent global loads per workgroup and dominated the forward).
//   fw[(pair_base(l) + p*QN + q)][step][lane][t] = W_l[32p + i][inmap(l,q,hi,KR*step+t)]                (lane = hi*32+i)
//   bw[(pair_base(l) + q*PN + p)][step][lane][t] = W_l[32p + nloc(hi,KR*step+t)][inmap(l,q,hi(i),r(i))]
//   image = [ fw : npair*1024 elems | bw : npair*1024 elems | bias : nob*32 floats | (fw_lo : npair*1024 elems) ]
// fw_lo (split-forward precisions only) holds the rounding residual of fw: fw_lo = round(W - float(fw)), so that fw + fw_lo
// carries twice the operand's mantissa (fp16: 22 bits, bf16: 16 bits).
// `pack_blocks` workgroups pack; when F > 0 the launch carries ONE more, which updates the pose table tf [F,12] from the pose
// corrections (pose_fwd_frame, nof_pose_dev.h) -- the two things a step needs before its ray marcher, in one launch instead of two
// 6-microsecond ones (nof_mlp_pack_pose).
template <class P>
__global__ __launch_bounds__(256) void k_mlp_pack(NofMlpDesc d, const float* __restrict__ params, char* __restrict__ image,
                                                  int with_lo, int pack_blocks, const float* __restrict__ pose,
                                                  const float* __restrict__ c2w, float max_trans, float max_rot,
                                                  float* __restrict__ tf, int F) {
  if ((int)blockIdx.x >= pack_blocks) {                                  // (workgroup-uniform)
    for (int f = threadIdx.x; f < F; f += blockDim.x) pose_fwd_frame(f, pose, c2w, max_trans, max_rot, tf);
    return;
  }
  constexpr int KR = P::KR;
  typedef typename P::elem elem;
  const int n_layers = d.n_sigma + d.n_color;
  const int npair = pair_base(d, n_layers);
  elem* fw = (elem*)image;
  elem* bw = fw + (size_t)npair * 16 * 64;
  float* bias = (float*)(bw + (size_t)npair * 16 * 64);
  elem* fw_lo = (elem*)(bias + (size_t)oblk_base(d, n_layers) * 32);
  const int total = npair * 16 * 64;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += pack_blocks * blockDim.x) {
    const int lane = e & 63, r = (e >> 6) & 15;
    int pair = e >> 10, l = 0;
    for (;; ++l) {
      const int cnt = lay_pn(d, l) * lay_qn(d, l);
      if (pair < cnt) break;
      pair -= cnt;
    }
    const int qn = lay_qn(d, l), pn = lay_pn(d, l), base = pair_base(d, l);
    const int hi = lane >> 5, i = lane & 31;
    const float* W = params + d.w_off[l];
    const int in_dim = d.in_dim[l], out_dim = d.out_dim[l];
    {
      const int p = pair / qn, q = pair % qn;
      const int row = 32 * p + i, col = inmap(d, l, q, hi, r);
      const float v = (row < out_dim && col >= 0) ? W[row * in_dim + col] : 0.0f;
      const size_t at = (((size_t)(base + p * qn + q) * (16 / KR) + r / KR) * 64 + lane) * KR + r % KR;
      fw[at] = (elem)v;
      if (with_lo) fw_lo[at] = (elem)(v - (float)(elem)v);
    }
    {
      const int q = pair / pn, p = pair % pn;
      const int hi_i = (i >> 2) & 1, r_i = (i & 3) + 4 * (i >> 3);
      const int row = 32 * p + nloc(hi, r), col = inmap(d, l, q, hi_i, r_i);
      const float v = (row < out_dim && col >= 0) ? W[row * in_dim + col] : 0.0f;
      bw[(((size_t)(base + q * pn + p) * (16 / KR) + r / KR) * 64 + lane) * KR + r % KR] = (elem)v;
    }
  }
  const int nob = oblk_base(d, n_layers);
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < nob * 32; e += pack_blocks * blockDim.x) {
    int ob = e >> 5, l = 0;
    for (;; ++l) {
      const int pn = lay_pn(d, l);
      if (ob < pn) break;
      ob -= pn;
    }
    const int row = 32 * ob + (e & 31);
    bias[e] = row < d.out_dim[l] ? params[d.b_off[l] + row] : 0.0f;
  }
}

__device__ __forceinline__ void copy16(char* __restrict__ dst, const char* __restrict__ src, size_t bytes) {
  const uint4* s4 = (const uint4*)src;
  uint4* d4 = (uint4*)dst;
  for (size_t e = threadIdx.x; e < bytes / 16; e += blockDim.x) d4[e] = s4[e];
}

template <int NS, int NC>
struct Shp {                                       // compile-time layer table (32-neuron blocks)
  static constexpr int NL = NS + NC;
  static constexpr __host__ __device__ int pn(int l) { return (l == NS - 1 || l == NL - 1) ? 1 : 2; }
  static constexpr __host__ __device__ int qn(int l) { return l == 0 ? 1 : 2; }
  // rows of a dW accumulator that can be non-zero: sigma head has 16 outputs (regs 0..7), colour head 3 (regs 0..3)
  // colour layer 0 (l == NS) accumulates the TRANSPOSED block dW^T[input slot][neuron]: both its input blocks use slots r < 8 only
  static constexpr __host__ __device__ int nacc(int l) { return (l == NS - 1 || l == NS) ? 8 : (l == NL - 1 ? 4 : 16); }
  static constexpr __host__ __device__ bool tr(int l) { return l == NS; }
  static constexpr __host__ __device__ int pair_base(int l) { int s = 0; for (int k = 0; k < l; ++k) s += pn(k) * qn(k); return s; }
  static constexpr __host__ __device__ int oblk_base(int l) { int s = 0; for (int k = 0; k < l; ++k) s += pn(k); return s; }
};

// ---- one dense layer -------------------------------------------------------------
// out[p][r] = neuron 32p + nloc(hi,r) of sample j (lane = sample).  `frag_off` / `bias_off` are compile-time byte offsets of the
// layer's fragments / biases inside the dynamic LDS block, so every ds_read is base-register + immediate.
// SPLIT (16-bit operand types only): both operands are carried as hi + lo = value rounded to the operand type + the rounded
// residual, and the product is the three MFMAs hi*hi + hi*lo + lo*hi (the lo*lo term is below fp32 rounding): twice the
// operand mantissa (fp16: 22 bits), i.e. fp32-class outputs from the 16-bit matrix cores at 3x the (idle) MFMA work.
// `lo_off` = byte offset of the layer's residual fragments (same layout as the main ones).
template <class P, int QN, int PN, bool SPLIT = false>
__device__ __forceinline__ void dense_o1(const char* smem, int frag_off, int bias_off, const float (&in)[QN][16],
                                         float (&out)[PN][16], int lane, int lo_off = 0) {
  constexpr int KR = P::KR, NSTEP = 16 / KR;
  constexpr int FB = 64 * KR * (int)sizeof(typename P::elem);          // bytes of one fragment (all 64 lanes)
  static_assert(!SPLIT || KR == 8, "the operand split is for the 16-bit operand types");
  const int hi = lane >> 5;
  typename P::frag bop[QN][NSTEP];
  typename P::frag blo[SPLIT ? QN : 1][SPLIT ? NSTEP : 1];
#pragma unroll
  for (int q = 0; q < QN; ++q)
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      bop[q][s] = P::pack(&in[q][KR * s]);
      if constexpr (SPLIT) {
        float res[KR];
#pragma unroll
        for (int t = 0; t < KR; ++t) res[t] = in[q][KR * s + t] - (float)bop[q][s][t];
        blo[q][s] = P::pack(res);
      }
    }
  const char* fl = smem + lane * (KR * (int)sizeof(typename P::elem));
  const char* bl = smem + bias_off + hi * 16;
#pragma unroll
  for (int p = 0; p < PN; ++p) {
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bv = *(const float4*)(bl + (32 * p + 8 * g) * 4);
      acc[4 * g] = bv.x; acc[4 * g + 1] = bv.y; acc[4 * g + 2] = bv.z; acc[4 * g + 3] = bv.w;
    }
    if constexpr (!SPLIT && QN * NSTEP >= 8) {
      // a long accumulator chain (128-wide layers): all of the block's weight fragments are requested before the chain starts,
      // so that the MFMAs do not each wait for an LDS read issued just in front of them
      typename P::frag a[QN * NSTEP];
#pragma unroll
      for (int t = 0; t < QN * NSTEP; ++t) a[t] = *(const typename P::frag*)(fl + frag_off + (p * QN * NSTEP + t) * FB);
      asm volatile("" ::: "memory");                                   // keeps the eight reads in front of the chain
#pragma unroll
      for (int q = 0; q < QN; ++q)
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) acc = P::mma(a[q * NSTEP + s], bop[q][s], acc);
    } else {
#pragma unroll
      for (int q = 0; q < QN; ++q)
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
          const typename P::frag a = *(const typename P::frag*)(fl + frag_off + ((p * QN + q) * NSTEP + s) * FB);
          if constexpr (SPLIT) {
            const typename P::frag al = *(const typename P::frag*)(fl + lo_off + ((p * QN + q) * NSTEP + s) * FB);
            acc = P::mma(al, bop[q][s], acc);
            acc = P::mma(a, blo[q][s], acc);
          }
          acc = P::mma(a, bop[q][s], acc);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) out[p][r] = acc[r];
  }
}

// ReLU in place + the derivative mask for the backward, without compares and without SGPR lane masks (32 v_cmp results per
// block spilled the scalar file in the backward kernels).  Per element two instructions: v_alignbit shifts the SIGN bit of the
// pre-activation into the mask ({mask, h} >> 31 = mask << 1 | sign), and the ReLU itself is a signed-INTEGER max with 0 on the
// float's bits (negative floats are negative integers; -0.0 -> +0.0; no canonicalising second v_max as fmaxf needs on an MFMA
// result).  Element i = 16 p + r of n = 16 PN therefore sits at bit n-1-i; the mask is returned inverted (1 = the unit is ON).
// A pre-activation of exactly +0.0 counts as on (its output is 0 either way; PyTorch's relu'(0) = 0 differs only there, a
// measure-zero event for trained units).
// NO INLINE ASSEMBLY here (rounds 1-2 had v_lshl_or / v_bfe_i32 as asm statements): hipcc's hazard recogniser does not see an
// asm statement as a VALU instruction, so the wait states between an in-flight MFMA and a VALU instruction that reads its
// result or overwrites one of its operands are not inserted.  Whether that bites depends on register allocation: two
// unrelated changes of the backward kernels (loading the next tile's inputs early) made the 3-layer variants compute dfeat
// 3-10 % wrong with the asm in place.  Builtins and plain C++ only.
template <int PN>
__device__ __forceinline__ uint32_t relu_mask(float (&h)[PN][16]) {
  uint32_t off = 0;
#pragma unroll
  for (int p = 0; p < PN; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int bits = __float_as_int(h[p][r]);
      off = __builtin_amdgcn_alignbit(off, (uint32_t)bits, 31);
      h[p][r] = __int_as_float(bits > 0 ? bits : 0);
    }
  return ~off;
}
// g = unit on ? g : 0   (sign-extended one-bit field = all-ones where on, then and)
template <int PN>
__device__ __forceinline__ void apply_mask(float (&g)[PN][16], uint32_t on) {
#pragma unroll
  for (int p = 0; p < PN; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const uint32_t keep = (uint32_t)__builtin_amdgcn_sbfe((int)on, 16 * PN - 1 - (p * 16 + r), 1);
      g[p][r] = __uint_as_float(__float_as_uint(g[p][r]) & keep);
    }
}

// ---- tile I/O ------------------------------------------------------------------------------------
// orientation 1 features: lane (sample j, hi) slot r = feature 16hi + r = (level 8hi + r/2, ch r&1).
// Addressing: ONE 32-bit byte offset per lane ((8 hi B + b) * 8) + a wave-uniform base per k (feat + k B * 8, in SGPRs), the
// form global_load takes directly (saddr + voffset); eight per-lane 64-bit addresses cost 16 VGPRs, and in the kernels that run
// at their register cap those were spilled and reloaded behind s_waitcnt vmcnt(0), one load at a time.  The entry points
// check that the level-major arrays stay below 4 GiB (L * B * 8 bytes).
__device__ __forceinline__ uint32_t feat_lane_offset(int64_t B, int64_t b, int hi) {
  return (uint32_t)(((int64_t)(8 * hi) * B + b) * 8);
}
__device__ __forceinline__ void load_feat_o1(const float2* __restrict__ feat, int L, int64_t B, int64_t b, int hi,
                                             float (&x)[1][16]) {
  const uint32_t voff = feat_lane_offset(B, b, hi);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int level = 8 * hi + k;
    const char* base = reinterpret_cast<const char*>(feat) + (size_t)k * (size_t)B * 8;      // wave-uniform
    float2 v = make_float2(0.f, 0.f);
    if (level < L && b < B) v = *reinterpret_cast<const float2*>(base + voff);
    x[0][2 * k] = v.x;
    x[0][2 * k + 1] = v.y;
  }
}
// the same 16 features from the operand-precision copy the fused forward leaves (k_enc_mlp_fwd: [B][hi][16] elements): two 16-byte
// loads per lane instead of eight 8-byte ones; exact (the backward rounds the fp32 features to the operand type first thing)
template <class P>
__device__ __forceinline__ void load_featq_o1(const typename P::elem* __restrict__ featq, int64_t B, int64_t b, int hi, float (&x)[1][16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) x[0][r] = 0.0f;
  if constexpr (P::KR == 8) {
    if (b < B) {
      const typename P::frag* q = reinterpret_cast<const typename P::frag*>(featq + (b * 2 + hi) * 16);
      const typename P::frag q0 = q[0], q1 = q[1];
#pragma unroll
      for (int t = 0; t < 8; ++t) { x[0][t] = (float)q0[t]; x[0][8 + t] = (float)q1[t]; }
    }
  }
}
// dfeat[level 8hi + k][b] = (df[2k], df[2k+1]) * scale, same addressing
__device__ __forceinline__ void store_dfeat_o1(float2* __restrict__ dfeat, int L, int64_t B, int64_t b, int hi,
                                               const float (&df)[16], float scale) {
  if (b >= B) return;
  const uint32_t voff = feat_lane_offset(B, b, hi);
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int level = 8 * hi + k;
    char* base = reinterpret_cast<char*>(dfeat) + (size_t)k * (size_t)B * 8;
    if (level < L) *reinterpret_cast<float2*>(base + voff) = make_float2(df[2 * k] * scale, df[2 * k + 1] * scale);
  }
}
// the lane id recomputed on the spot (v_mbcnt on an opaque zero: three instructions).  Everything derived from threadIdx is
// loop-invariant, and in a kernel that sits at its register cap the compiler hoists such values out of the persistent loop and
// then SPILLS them instead of recomputing them; what hangs off this cannot be hoisted.
__device__ __forceinline__ uint32_t lane_id_here() {
  uint32_t zero = 0u;
  asm volatile("" : "+v"(zero));
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, zero));
}
__device__ __forceinline__ void load_view_o1(const float* __restrict__ view, int S, int64_t B, int64_t b, int hi,
                                             float (&x)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) x[r] = 0.0f;
  if (b < B) {                                          // slot (hi, r < 8) = view column 8 hi + r (inmap)
    const float4* v = (const float4*)(view + (b / S) * NOF_VIEW_COLS + 8 * hi);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float4 t = v[g];
      x[4 * g] = t.x; x[4 * g + 1] = t.y; x[4 * g + 2] = t.z; x[4 * g + 3] = t.w;
    }
  }
}
// the same with a uniform base + ONE 32-bit lane offset (see load_sig_tile_o1): the view rows of a batch are far below 4 GiB
__device__ __forceinline__ void load_view_off_o1(const float* __restrict__ view, int S, int64_t B, int64_t b, int hi,
                                                 float (&x)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) x[r] = 0.0f;
  if (b < B) {
    const uint32_t off = (uint32_t)(b / S) * (uint32_t)(NOF_VIEW_COLS * 4) + (lane_id_here() & 32u);      // (+ 32 hi bytes)
    (void)hi;
    const float4* v = reinterpret_cast<const float4*>(reinterpret_cast<const char*>(view) + off);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      const float4 t = v[g];
      x[4 * g] = t.x; x[4 * g + 1] = t.y; x[4 * g + 2] = t.z; x[4 * g + 3] = t.w;
    }
  }
}
// view column held by lane j of a slot-per-lane (transposed) view block, -1 for the unused slots: lane j = slot (hi_j, r_j)
__device__ __forceinline__ int view_col_of_lane(int j) {
  const int hi_j = (j >> 2) & 1, r_j = (j & 3) + 4 * (j >> 3);
  return r_j < 8 ? 8 * hi_j + r_j : -1;
}

// The sigma head's 16 outputs (sdf + geo_feat) of a sample in operand precision: [B][hi][8] elements; lane (j, hi) holds
// rows nloc(hi, r), r < 8 (r >= 8 are the padded rows >= 16, structurally zero).  Written by the forward kernel and read back
// as the colour net's input by the split backward; the same layout carries dL/d(sigma out) between its two kernels.  The
// values are rounded exactly where the fused kernel rounds them (P::pack of the MFMA operand), so nothing changes numerically.
template <class P>
__device__ __forceinline__ void store_sig_o1(typename P::elem* __restrict__ sig, int64_t B, int64_t b, int hi, const float (&x)[16]) {
  if (b < B) *reinterpret_cast<typename P::frag*>(sig + (b * 2 + hi) * 8) = P::pack(&x[0]);
}
// "These 16 values are in registers NOW": an empty asm statement that takes them as read-write operands.  Placed between
// `x = x_next` and the request of the tile after next, it makes the compiler wait for the previous look-ahead loads there (and
// keep x apart from the registers the new loads land in) instead of right after issuing the new ones, which is what it did in
// the kernels that run at their register cap (s_waitcnt vmcnt(1) behind the eight new loads: no look-ahead at all).  No
// instruction is emitted, so there is no MFMA hazard to miss.
__device__ __forceinline__ void pin16(float (&x)[16]) {
  asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]), "+v"(x[8]),
               "+v"(x[9]), "+v"(x[10]), "+v"(x[11]), "+v"(x[12]), "+v"(x[13]), "+v"(x[14]), "+v"(x[15]));
}
template <class P>
__device__ __forceinline__ typename P::frag load_sig_raw(const typename P::elem* __restrict__ sig, int64_t B, int64_t b, int hi) {
  typename P::frag f;
#pragma unroll
  for (int t = 0; t < 8; ++t) f[t] = (typename P::elem)0.0f;
  if (b < B) f = *reinterpret_cast<const typename P::frag*>(sig + (b * 2 + hi) * 8);
  return f;
}
template <class P>
__device__ __forceinline__ void sig_to_o1(const typename P::frag& f, float (&x)[16]) {
#pragma unroll
  for (int t = 0; t < 8; ++t) x[t] = (float)f[t];
#pragma unroll
  for (int r = 8; r < 16; ++r) x[r] = 0.0f;
}
template <class P>
__device__ __forceinline__ void load_sig_o1(const typename P::elem* __restrict__ sig, int64_t B, int64_t b, int hi, float (&x)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) x[r] = 0.0f;
  if (b < B) {
    const typename P::frag f = *reinterpret_cast<const typename P::frag*>(sig + (b * 2 + hi) * 8);
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = (float)f[t];
  }
}
// the same for sample j of the (wave-uniform) tile: a uniform tile base + ONE 32-bit lane offset -- the form global_load takes
// directly (saddr + voffset).  With `sig + (b * 2 + hi) * 8` the compiler keeps a 64-bit per-lane base (sig + 16 hi) alive across
// the persistent loop: two registers the three-colour-layer backward, which sits at its 256, does not have (it spilled them).
template <class P>
__device__ __forceinline__ void load_sig_tile_o1(const typename P::elem* __restrict__ sig, int64_t B, int64_t tile, int j, int hi,
                                                 float (&x)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) x[r] = 0.0f;
  if (tile * 32 + j < B) {
    const char* base = reinterpret_cast<const char*>(sig) + tile * (int64_t)(32 * 2 * 8 * sizeof(typename P::elem));
    const uint32_t l = lane_id_here();                                 // (j = l & 31, hi = l >> 5: recomputed here, see lane_id_here)
    const uint32_t off = (((l & 31u) << 1) | (l >> 5)) * (uint32_t)(8 * sizeof(typename P::elem));
    (void)hi;
    const typename P::frag f = *reinterpret_cast<const typename P::frag*>(base + off);
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = (float)f[t];
  }
}


// ---- work list of the backward (NofTileList, include/nof_hip.h; built by nof_composite_loss_fwd_bwd / nof_tile_list_build) ----
// head[0] = number of listed tiles, head + 4 = their ids (ascending).  A persistent wave takes entries wave, wave + n_waves, ...
// of the LIST, so every wave gets the same number of tiles that have work (+-1) whatever their position in the batch -- with
// the tiles strided over the waves in batch order the slowest wave set the kernel's time (63 % of the tiles skipped bought 16 %).
// Without a list (NULL): every tile of the batch, in order, each tested for an all-zero gradient in place.
struct TileWork {
  const uint32_t* tiles;                              // NULL: identity
  int64_t n, ntiles;
  __device__ __forceinline__ TileWork(const void* tile_list, int64_t ntiles_) : ntiles(ntiles_) {
    const uint32_t* head = (const uint32_t*)tile_list;
    tiles = head ? head + 4 : nullptr;
    n = head ? (int64_t)__builtin_amdgcn_readfirstlane((int)head[0]) : ntiles_;
  }
  // tile id of work item i; past the end: a tile that does not exist (every guarded load / store of it is a no-op)
  __device__ __forceinline__ int64_t at(int64_t i) const {
    if (i >= n) return ntiles;
    return tiles ? (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)tiles[i]) : i;
  }
};

// compile-time byte offsets inside the dynamic LDS block: [fw frags | (bw frags) | bias | ...]
#define PAIR_BYTES (16 * 64 * (int)sizeof(typename P::elem))
#define FW_OFF(l) (SH::pair_base(l) * PAIR_BYTES)
#define BW_OFF(l) (BW_BASE + SH::pair_base(l) * PAIR_BYTES)
#define BIAS_OFF(l) (BIAS_BASE + SH::oblk_base(l) * 32 * 4)
#define LO_OFF(l) (LO_BASE + SH::pair_base(l) * PAIR_BYTES)

// =====================================================================================================
// forward: raw[b] = (rgb_raw[3], sdf)
// =====================================================================================================
// Waves per workgroup of the forward kernel.  The LDS image of the weight fragments (57 KB with the operand split at 3 + 2
// layers) limits a CU to two workgroups: with 4 waves each that is 2 waves per SIMD, and the three dependent MFMAs of a split
// product leave the matrix pipe idle 60 % of the time.  The 16-bit kernels need <= 120 registers, so 8 waves share one image:
// 4 waves per SIMD from the same LDS.  (fp32 fragments: 127-129 registers, stays at 4 waves per workgroup.)
template <class P> struct FwdWaves { static constexpr int value = P::KR == 8 ? 8 : 4; };

template <class P, int NS, int NC, bool SDF_ONLY, bool SPLIT>
__global__ __launch_bounds__(64 * FwdWaves<P>::value, FwdWaves<P>::value == 8 ? 4 : 2) void k_mlp_fwd(   // no AGPRs: MFMA results land in VGPRs
NofMlpDesc d, const char* __restrict__ image,
                                                  const float2* __restrict__ feat, int L, const float* __restrict__ view,
                                                  int S, float* __restrict__ out, typename P::elem* __restrict__ sig, int64_t B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef Shp<NS, NC> SH;
  constexpr int NL = SDF_ONLY ? NS : NS + NC;
  constexpr int BIAS_BASE = SH::pair_base(NL) * PAIR_BYTES;           // this kernel keeps only the first NL layers' fragments
  constexpr int LO_BASE = BIAS_BASE + SH::oblk_base(NL) * 32 * 4;     // residual fragments of the split forward
  copy16(smem, image, (size_t)BIAS_BASE);
  copy16(smem + BIAS_BASE, image + 2 * (size_t)SH::pair_base(NS + NC) * PAIR_BYTES, (size_t)SH::oblk_base(NL) * 32 * 4);
  if constexpr (SPLIT)
    copy16(smem + LO_BASE, image + 2 * (size_t)SH::pair_base(NS + NC) * PAIR_BYTES + (size_t)SH::oblk_base(NS + NC) * 32 * 4,
           (size_t)BIAS_BASE);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hi = lane >> 5, j = lane & 31;
  constexpr int NW = FwdWaves<P>::value;
  const int64_t ntiles = (B + 31) / 32;
  const int64_t tstride = (int64_t)gridDim.x * NW;
  float xn[1][16];                                    // the NEXT tile's features: loaded a whole tile ahead (latency hidden)
  load_feat_o1(feat, L, B, ((int64_t)blockIdx.x * NW + wave) * 32 + j, hi, xn);
  for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < ntiles; tile += tstride) {
    asm volatile("" ::: "memory");                    // keep the weight fragments in LDS (no hoisting into VGPRs)
    const int64_t b = tile * 32 + j;
    float x[1][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[0][r] = xn[0][r];
    pin16(x[0]);
    load_feat_o1(feat, L, B, (tile + tstride) * 32 + j, hi, xn);      // out-of-range tiles load nothing (b >= B -> zeros)
    float h[2][16], so[1][16];
    dense_o1<P, 1, 2, SPLIT>(smem, FW_OFF(0), BIAS_OFF(0), x, h, lane, LO_OFF(0));
    relu_mask<2>(h);
#pragma unroll
    for (int l = 1; l < NS - 1; ++l) {
      float h2[2][16];
      dense_o1<P, 2, 2, SPLIT>(smem, FW_OFF(l), BIAS_OFF(l), h, h2, lane, LO_OFF(l));
      relu_mask<2>(h2);
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[p][r] = h2[p][r];
    }
    dense_o1<P, 2, 1, SPLIT>(smem, FW_OFF(NS - 1), BIAS_OFF(NS - 1), h, so, lane, LO_OFF(NS - 1));
    if constexpr (SDF_ONLY) {
      if (hi == 0 && b < B) out[b] = so[0][0];
    } else {
      float cin[2][16];
#pragma unroll
      for (int r = 0; r < 16; ++r) cin[0][r] = so[0][r];
      if constexpr (P::KR == 8) {
        if (sig != nullptr) store_sig_o1<P>(sig, B, b, hi, so[0]);     // the colour net's operand, kept for the split backward
      }
      load_view_o1(view, S, B, b, hi, cin[1]);        // (a tile ahead like the features: measured 2 % slower)
      dense_o1<P, 2, 2, SPLIT>(smem, FW_OFF(NS), BIAS_OFF(NS), cin, h, lane, LO_OFF(NS));
      relu_mask<2>(h);
#pragma unroll
      for (int l = NS + 1; l < NS + NC - 1; ++l) {
        float h2[2][16];
        dense_o1<P, 2, 2, SPLIT>(smem, FW_OFF(l), BIAS_OFF(l), h, h2, lane, LO_OFF(l));
        relu_mask<2>(h2);
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int r = 0; r < 16; ++r) h[p][r] = h2[p][r];
      }
      float co[1][16];
      dense_o1<P, 2, 1, SPLIT>(smem, FW_OFF(NS + NC - 1), BIAS_OFF(NS + NC - 1), h, co, lane, LO_OFF(NS + NC - 1));
      if (hi == 0 && b < B) ((float4*)out)[b] = make_float4(co[0][0], co[0][1], co[0][2], so[0][0]);
    }
  }
}

// =====================================================================================================
// forward with the hash encode fused in ("LDS-staged features": the encoded features never leave the CU).
//   pts_w [B,3] -> multiresolution encode (gridencoder.cu:107-200) -> both MLPs -> raw [B,4]
// The reference materialises the [B,32] embedding (nerf_runner.py:1255-1267) and so did k_hash_fwd + k_mlp_fwd: 100.7 MB written
// and read back per cfg2 step.  Here a wave owns 64 consecutive samples = TWO 32-sample tiles:
//   * encode with one SAMPLE per lane and the level uniform over the wave -- the form the stand-alone encoder is fastest in: level
//     constants in SGPRs (staged in LDS once per workgroup, read per level: as plain kernel arguments all 80 of them were held in
//     SGPRs across the loop and 285 of those were spilled to VGPR lanes), no divergence between dense and hashed levels, the
//     x-neighbour pair of a dense level in one 16-byte load (level_pairs), 92 gather instructions per 64 samples at cfg2, rows
//     computed per level without a branch per corner (enc_prep), gathers in global_load's saddr + 32-bit-offset form;
//   * lane s then holds the features of sample s, while the first layer's B operand wants lane (j, hi) to hold features
//     16 hi .. 16 hi + 15 of sample j: the features go through a wave-private LDS stage, feature-major [32][64] floats -- one
//     conflict-free ds_write per feature as it is produced, one conflict-free ds_read per operand register -- which also frees the
//     registers while the next level's gathers are in flight.  (Round 4's first version exchanged the halves with 16
//     v_permlane32_swap and parked tile B's operand in LDS: the same time, 32 more live registers.)
//   * the two tiles go through the MLP chain one after the other (dense_o1, the same fragments in LDS as k_mlp_fwd).  ONE
//     workgroup of up to 12 waves per CU around one fragment image (3 waves per SIMD; 8 KB of stage per wave bound the count).
// `featq` (may be NULL): the features in MFMA operand precision and operand order, [B][hi][16] elements = 64 B per sample -- what
// the split backward's sigma kernel needs of them (it rounds them to the operand type first thing): half the bytes of the fp32
// level-major array, written with two 16-byte stores per lane.  NULL: nothing but raw (and the sigma hand-off) is written.
// Same values as k_hash_fwd + k_mlp_fwd, bit for bit (tests/test_gpu_ops.py): the encode is encode_level()'s arithmetic, the
// chain is dense_o1().
//
// ONE level's gathers in flight per wave (NOF_ENC_GROUP = 1): the kernel is bound by gather issue and the matrix pipe taking turns,
// not by gather latency, so more levels in flight buy nothing (measured: the same 105-115 us with 1, 2 and 4).
// Round 4 met a fault here with two or more levels in flight -- wrong features of one level of a group in lanes 48-63, in a few
// 16-sample groups per million samples, differently on every run (profiles/r04_fused_forward_race.txt) -- and shipped GROUP = 1 as a
// workaround without knowing why.  Root cause (round 5; DESIGN 2.10, profiles/r05_*_fault_*.txt): with GROUP >= 2 clang's SLP
// vectoriser blended that level with packed-fp32 instructions, among them `v_pk_mul_f32 vD, vA, vB op_sel:[0,1]`; on gfx950 a packed
// fp32 instruction that reads SOURCE 1 through op_sel = 1 returns a wrong low result in its last quarter-wave whenever another wave
// of the SIMD executes an MFMA at that moment (stand-alone: tools/repro/pk_swap_repro.hip, 1.6 % of the executions under MFMA load,
// none without).  Bisected on this kernel's own assembly (tools/asm_variant.sh): rewriting that ONE instruction as two scalar
// multiplies cures it, rewriting any other class of packed instruction does not.  The library is therefore built with
// -fno-slp-vectorize (bundlesdf_amd/build.py; no measurable cost) and tests/test_capi.py disassembles every built library to
// prove the form is absent; NOF_ENC_BLEND = 1 additionally keeps the two channels of enc_blend apart at the source level.
// tests/test_gpu_ops.py::test_fused_forward_is_repeatable stays as the product-level guard.
// =====================================================================================================
#ifndef NOF_ENC_WAVES
#define NOF_ENC_WAVES 12                                  // most waves per workgroup (one workgroup per CU, 3 waves per SIMD)
#endif
#ifndef NOF_ENC_GROUP
#define NOF_ENC_GROUP 1                                   // levels whose gathers are in flight together (see above)
#endif
#ifndef NOF_ENC_ROLLED
#define NOF_ENC_ROLLED 1
#endif
#ifndef NOF_ENC_PRIO
#define NOF_ENC_PRIO 1                                    // s_setprio by phase (below); 0: none
#endif
#ifndef NOF_GRID_PRIO
#define NOF_GRID_PRIO 0                                   // the same in k_sdf_grid (A/B: profiles/r05_u_*)
#endif
#ifndef NOF_ENC_PRIO_ENC
#define NOF_ENC_PRIO_ENC 3
#endif
#ifndef NOF_ENC_PRIO_CHAIN
#define NOF_ENC_PRIO_CHAIN 0
#endif
#ifndef NOF_ENC_DEBUG_FEAT
#define NOF_ENC_DEBUG_FEAT 0                              // tools/fused_debug*.py: the features as computed, before the LDS stage
#endif

// indices and fractions of one level for one point; the loads and the blend are separate steps so that a GROUP of levels has all
// its gathers in flight before the first one is waited for
struct EncCell {
  uint32_t idx[8];
  float f[3];
  bool oob;
  const char* base;                                    // the level's first table row (wave-uniform: an SGPR pair)
};
// The rows are grid_index()'s (gridencoder.cu:66-83) with the terms the eight corners share computed once and every decision taken
// per LEVEL (wave-uniform here), so that the lanes run straight-line code: grid_index() per corner tests `index >= size` per lane,
// which costs a branch per corner.  (The same rule as make_scatter in nof_hash.hip, whose rows the scatter tests pin.)
__device__ __forceinline__ EncCell enc_prep(const HashLevel& lv, const float (&p)[3]) {
  const CellPos c = locate3(p, lv.scale);
  EncCell e;
  e.oob = c.oob;
#pragma unroll
  for (int d = 0; d < 3; ++d) e.f[d] = c.f[d];
  if (lv.hashed) {
    const uint32_t hy0 = c.g[1] * 2654435761u, hy1 = hy0 + 2654435761u, hz0 = c.g[2] * 805459861u, hz1 = hz0 + 805459861u;
    const uint32_t yz[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
#pragma unroll
    for (int k = 0; k < 8; ++k) e.idx[k] = (c.g[0] + (k & 1)) ^ yz[k >> 1];
    if ((lv.size & (lv.size - 1u)) == 0u) {
#pragma unroll
      for (int k = 0; k < 8; ++k) e.idx[k] &= lv.size - 1u;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) e.idx[k] %= lv.size;
    }
  } else {
    const uint32_t r1 = lv.res + 1u, r2 = r1 * r1;
    const uint32_t base = c.g[0] + c.g[1] * r1 + c.g[2] * r2;
#pragma unroll
    for (int k = 0; k < 8; ++k) e.idx[k] = base + (k & 1) + ((k >> 1) & 1) * r1 + (k >> 2) * r2;
    if (!level_pairs(lv)) {                                  // a dense level whose linear index can reach the modulo wrap
#pragma unroll
      for (int k = 0; k < 8; ++k) e.idx[k] %= lv.size;
    }
  }
  return e;
}
// The gathers of one level: a wave-uniform base (SGPR pair) + one 32-bit byte offset per lane and corner (global_load's
// saddr + voffset form).  The offsets are turned into byte offsets in place and stay in e.idx: enc_keep() below holds them live
// until the group's gathers have landed.
template <bool PAIRS>
__device__ __forceinline__ void enc_load(const HashLevel& lv, const float2* __restrict__ table, EncCell& e, float2 (&v)[8]) {
  // (an out-of-range point still loads: grid_index wraps every row into the level, and enc_blend returns zeros for it)
  const char* __restrict__ base = reinterpret_cast<const char*>(table + lv.offset);
  e.base = base;
#pragma unroll
  for (int k = 0; k < 8; ++k) e.idx[k] *= 8u;                        // rows -> bytes (a level is far below 4 GiB)
  if constexpr (PAIRS) {
#pragma unroll
    for (int k = 0; k < 8; k += 2) {
      const RowPair t = *reinterpret_cast<const RowPair*>(base + e.idx[k]);
      v[k] = make_float2(t.x, t.y);
      v[k + 1] = make_float2(t.z, t.w);
    }
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const float2*>(base + e.idx[k]);
  }
}
// "These address registers are read HERE": placed behind a group's gathers, it keeps the offsets and the base
// apart from the gathers' destination registers (the register allocator otherwise hands a gather's address registers to a later
// gather as its destination: legal, and not the cause of the quarter-wave fault described at the kernel -- kept as cheap insurance).
__device__ __forceinline__ void enc_keep(const EncCell& e) {
  asm volatile("" :: "v"(e.idx[0]), "v"(e.idx[1]), "v"(e.idx[2]), "v"(e.idx[3]), "v"(e.idx[4]), "v"(e.idx[5]), "v"(e.idx[6]), "v"(e.idx[7]),
               "s"(e.base));
}
#ifndef NOF_ENC_BLEND
#define NOF_ENC_BLEND 0                                   // experiments (tools/fused_fault.sh): 1 = the two channels kept apart (no packed-fp32 pairing), 2 = forced packed
#endif
__device__ __forceinline__ float2 enc_blend(const EncCell& e, const float2 (&v)[8]) {   // encode_level's own weights and order
  float2 acc = make_float2(0.f, 0.f);
#if NOF_ENC_BLEND == 2
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f a2 = {0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float wk = 1.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) wk *= (k & (1 << d)) ? e.f[d] : 1.0f - e.f[d];
    const v2f vk = {v[k].x, v[k].y}, w2 = {wk, wk};
    a2 = a2 + w2 * vk;
  }
  acc.x = a2.x;
  acc.y = a2.y;
#else
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float wk = 1.0f;
#pragma unroll
    for (int d = 0; d < 3; ++d) wk *= (k & (1 << d)) ? e.f[d] : 1.0f - e.f[d];
#if NOF_ENC_BLEND == 1
    float px = wk * v[k].x, py = wk * v[k].y;
    asm volatile("" : "+v"(px));                       // an opaque value: the x and y products cannot be paired into one v_pk_mul_f32
    acc.x += px;
    asm volatile("" : "+v"(acc.x));
    acc.y += py;
#else
    acc.x += wk * v[k].x;
    acc.y += wk * v[k].y;
#endif
  }
#endif
  // an out-of-range point: zeros (gridencoder.cu:131-139), by a select -- no lane is switched off around the blend, so no
  // execution mask per level in flight has to be kept (they were being spilled to VGPR lanes)
  acc.x = e.oob ? 0.0f : acc.x;
  acc.y = e.oob ? 0.0f : acc.y;
  return acc;
}

// one 32-sample tile through both networks: x = the first layer's B operand (lane (j, hi): features 16 hi .. 16 hi + 15 of sample b)
template <class P, int NS, int NC, bool SPLIT>
__device__ __forceinline__ void enc_chain_tile(const NofMlpDesc& d, const char* smem, const float (&xin)[16], int64_t b,
                                               const float* __restrict__ view, int S, float* __restrict__ out,
                                               typename P::elem* __restrict__ sig, typename P::elem* __restrict__ featq, int64_t B,
                                               int lane) {
  typedef Shp<NS, NC> SH;
  constexpr int NL = NS + NC;
  constexpr int BIAS_BASE = SH::pair_base(NL) * PAIR_BYTES;
  constexpr int LO_BASE = BIAS_BASE + SH::oblk_base(NL) * 32 * 4;
  const int hi = lane >> 5;
  float x[1][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) x[0][r] = xin[r];
  if (featq != nullptr && b < B) {
    typename P::frag* q = reinterpret_cast<typename P::frag*>(featq + (b * 2 + hi) * 16);
#pragma unroll
    for (int s2 = 0; s2 < 16 / P::KR; ++s2) q[s2] = P::pack(&x[0][P::KR * s2]);
  }
  float h[2][16], so[1][16];
  dense_o1<P, 1, 2, SPLIT>(smem, FW_OFF(0), BIAS_OFF(0), x, h, lane, LO_OFF(0));
  relu_mask<2>(h);
#pragma unroll
  for (int l = 1; l < NS - 1; ++l) {
    float h2[2][16];
    dense_o1<P, 2, 2, SPLIT>(smem, FW_OFF(l), BIAS_OFF(l), h, h2, lane, LO_OFF(l));
    relu_mask<2>(h2);
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int r = 0; r < 16; ++r) h[pp][r] = h2[pp][r];
  }
  dense_o1<P, 2, 1, SPLIT>(smem, FW_OFF(NS - 1), BIAS_OFF(NS - 1), h, so, lane, LO_OFF(NS - 1));
  float cin[2][16];
#pragma unroll
  for (int r = 0; r < 16; ++r) cin[0][r] = so[0][r];
  if (sig != nullptr) store_sig_o1<P>(sig, B, b, hi, so[0]);
  load_view_o1(view, S, B, b, hi, cin[1]);
  dense_o1<P, 2, 2, SPLIT>(smem, FW_OFF(NS), BIAS_OFF(NS), cin, h, lane, LO_OFF(NS));
  relu_mask<2>(h);
#pragma unroll
  for (int l = NS + 1; l < NS + NC - 1; ++l) {
    float h2[2][16];
    dense_o1<P, 2, 2, SPLIT>(smem, FW_OFF(l), BIAS_OFF(l), h, h2, lane, LO_OFF(l));
    relu_mask<2>(h2);
#pragma unroll
    for (int pp = 0; pp < 2; ++pp)
#pragma unroll
      for (int r = 0; r < 16; ++r) h[pp][r] = h2[pp][r];
  }
  float co[1][16];
  dense_o1<P, 2, 1, SPLIT>(smem, FW_OFF(NS + NC - 1), BIAS_OFF(NS + NC - 1), h, co, lane, LO_OFF(NS + NC - 1));
  if (hi == 0 && b < B) ((float4*)out)[b] = make_float4(co[0][0], co[0][1], co[0][2], so[0][0]);
}

template <class P, int NS, int NC, bool SPLIT>
__global__ __launch_bounds__(64 * NOF_ENC_WAVES, (NOF_ENC_WAVES + 3) / 4) void k_enc_mlp_fwd(
    NofMlpDesc d, const char* __restrict__ image, NofHashGrid g, const float2* __restrict__ table, const float* __restrict__ pts_w,
    const float* __restrict__ view, int S, float* __restrict__ out, typename P::elem* __restrict__ sig,
    typename P::elem* __restrict__ featq, int64_t B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef Shp<NS, NC> SH;
  constexpr int NL = NS + NC;
  constexpr int BIAS_BASE = SH::pair_base(NL) * PAIR_BYTES;
  constexpr int LO_BASE = BIAS_BASE + SH::oblk_base(NL) * 32 * 4;
  constexpr int PARK_BASE = LO_BASE + (SPLIT ? BIAS_BASE : 0);        // [wave][32][64] floats: the wave's feature stage (see below)
  copy16(smem, image, (size_t)BIAS_BASE);
  copy16(smem + BIAS_BASE, image + 2 * (size_t)SH::pair_base(NL) * PAIR_BYTES, (size_t)SH::oblk_base(NL) * 32 * 4);
  if constexpr (SPLIT)
    copy16(smem + LO_BASE, image + 2 * (size_t)SH::pair_base(NL) * PAIR_BYTES + (size_t)SH::oblk_base(NL) * 32 * 4, (size_t)BIAS_BASE);
  // the 16 levels' constants (80 scalars of the kernel argument) in LDS: read per level where they are needed.  As plain kernel
  // arguments they were all loaded up front and kept in SGPRs across the persistent loop -- 285 SGPR spills to VGPR lanes.
  uint32_t* lvl = reinterpret_cast<uint32_t*>(smem + PARK_BASE + (blockDim.x >> 6) * 8192);     // [16][8] words behind the stages
  if (threadIdx.x < NOF_MAX_LEVELS) {
    const int l = threadIdx.x;
    lvl[l * 8 + 0] = __float_as_uint(g.scale[l]); lvl[l * 8 + 1] = g.resolution[l]; lvl[l * 8 + 2] = g.offset[l];
    lvl[l * 8 + 3] = g.size[l]; lvl[l * 8 + 4] = g.hashed[l];
  }
  const int n_levels = g.L;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hi = lane >> 5, j = lane & 31;
  constexpr int GR = NOF_ENC_GROUP;
  auto level_at = [&](int l) {                          // wave-uniform: the LDS words go through readfirstlane into SGPRs
    HashLevel lv;
    const uint4 q = *reinterpret_cast<const uint4*>(lvl + l * 8);
    lv.scale = __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)q.x));
    lv.res = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.y);
    lv.offset = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.z);
    lv.size = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.w);
    lv.hashed = (uint32_t)__builtin_amdgcn_readfirstlane((int)lvl[l * 8 + 4]);
    return lv;
  };
  const int NW = blockDim.x >> 6;                     // (the host sizes the workgroup by what LDS admits: 8 KB of stage per wave)
  const int64_t npairs = (B + 63) / 64;
  for (int64_t tp = (int64_t)blockIdx.x * NW + wave; tp < npairs; tp += (int64_t)gridDim.x * NW) {
    asm volatile("" ::: "memory");                    // keep the weight fragments in LDS (no hoisting into VGPRs)
#if NOF_ENC_PRIO
    __builtin_amdgcn_s_setprio(NOF_ENC_PRIO_ENC);      // (see NOF_ENC_PRIO)
#endif
    // ---------------- encode: lane = sample tp*64 + lane, all levels, GR levels' gathers in flight together ----------------
    // The features go to a wave-private LDS stage as they are produced, feature-major: stage[feature][sample] -- one conflict-free
    // ds_write_b32 per feature -- so that no register holds them while the next group's gathers are in flight (32 accumulated
    // features + four levels in flight did not fit 168 registers), and the chain reads them back in operand order below.
    const int64_t bs = tp * 64 + lane;
    const int64_t bb = bs < B ? bs : B - 1;           // (a lane past the end encodes the last sample: nothing of it is stored)
    const float p[3] = {pts_w[bb * 3], pts_w[bb * 3 + 1], pts_w[bb * 3 + 2]};
    float* stage = reinterpret_cast<float*>(smem + PARK_BASE + wave * 8192);              // [32][64] floats
#if NOF_ENC_ROLLED
#pragma unroll 1                                      // one copy of the level body: the unrolled kernel is 76 KB of code (> the 64 KB instruction cache)
#else
#pragma unroll
#endif
    for (int l0 = 0; l0 < NOF_MAX_LEVELS; l0 += GR) {
      if (l0 < n_levels) {                            // (uniform)
        EncCell e[GR];
        float2 v[GR][8];
#pragma unroll
        for (int u = 0; u < GR; ++u) {
          if (l0 + u < NOF_MAX_LEVELS && l0 + u < n_levels) {
            const HashLevel lv = level_at(l0 + u);
            e[u] = enc_prep(lv, p);
            if (level_pairs(lv)) enc_load<true>(lv, table, e[u], v[u]);
            else enc_load<false>(lv, table, e[u], v[u]);
            asm volatile("" ::: "memory");            // this level's gathers are issued before the next level's rows are computed
          }                                           // (the scheduler otherwise computes all 32 rows first: 32 more live registers)
        }
#pragma unroll
        for (int u = 0; u < GR; ++u)
          if (l0 + u < NOF_MAX_LEVELS && l0 + u < n_levels) enc_keep(e[u]);       // (the gathers' address registers stay apart from their destinations)
#pragma unroll
        for (int u = 0; u < GR; ++u) {
          float2 a = make_float2(0.f, 0.f);
          if (l0 + u < NOF_MAX_LEVELS && l0 + u < n_levels) a = enc_blend(e[u], v[u]);
          if (l0 + u < NOF_MAX_LEVELS) {
            stage[(2 * (l0 + u)) * 64 + lane] = a.x;
            stage[(2 * (l0 + u) + 1) * 64 + lane] = a.y;
#if NOF_ENC_DEBUG_FEAT
            if (bs < B) {                             // debug build: the features as computed, before the LDS stage ([B][32] floats behind featq)
              float* dbg = reinterpret_cast<float*>(featq + B * 32);
              dbg[bs * 32 + 2 * (l0 + u)] = a.x;
              dbg[bs * 32 + 2 * (l0 + u) + 1] = a.y;
            }
#endif
          }
        }
      } else {
#pragma unroll
        for (int u = 0; u < 2 * GR; ++u)
          if (2 * l0 + u < 32) stage[(2 * l0 + u) * 64 + lane] = 0.0f;                     // levels the grid does not have
      }
    }
#if NOF_ENC_PRIO
    __builtin_amdgcn_s_setprio(NOF_ENC_PRIO_CHAIN);
#endif
    // ---------------- the two tiles through the chain: lane (j, hi) of tile t reads features 16 hi .. 16 hi + 15 of sample
    //                  32 t + j = stage[16 hi + r][32 t + j] (conflict-free: a half-wave reads 32 consecutive words) ----------------
#pragma unroll
    for (int t = 0; t < 2; ++t) {                     // (two inlined copies: as a rolled loop the split variants spilled)
      asm volatile("" ::: "memory");
      float xa[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) xa[r] = stage[(16 * hi + r) * 64 + 32 * t + j];
      enc_chain_tile<P, NS, NC, SPLIT>(d, smem, xa, tp * 64 + 32 * t + j, view, S, out, sig, featq, B, lane);
    }
  }
}

// =====================================================================================================
// dense SDF grid for mesh extraction (extract_mesh + run_network_density, nerf_runner.py:1307-1386), fused:
// voxel centre -> octree mask -> hash encode in registers -> sigma net on MFMA -> sdf[nx,ny,nz].  Nothing per point ever goes
// to HBM except the 4-byte result (the reference materialises query_pts, the [N,32] embedding and every activation).
// A wave owns 32 consecutive voxels of one z column (lane = (voxel, hi), hi picks which 8 levels the lane encodes = exactly
// the B-operand layout of the first layer); occupancy is spatially coherent (level <= 6 cells vs 1/512 voxels), so a
// wave-uniform skip of all-empty tiles is the whole compaction that is needed.
// =====================================================================================================
template <class P, int NS, int NC, bool SPLIT>
__global__ __launch_bounds__(256, 2) void k_sdf_grid(NofMlpDesc d, const char* __restrict__ image, NofHashGrid g,
                                                      const float2* __restrict__ table, const uint32_t* __restrict__ occ_bits,
                                                      int occ_n, const float* __restrict__ tx, const float* __restrict__ ty,
                                                      const float* __restrict__ tz, int nx, int ny, int nz, float outside,
                                                      float* __restrict__ sdf) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef Shp<NS, NC> SH;
  constexpr int BIAS_BASE = SH::pair_base(NS) * PAIR_BYTES;
  constexpr int LO_BASE = BIAS_BASE + SH::oblk_base(NS) * 32 * 4;
  copy16(smem, image, (size_t)BIAS_BASE);
  copy16(smem + BIAS_BASE, image + 2 * (size_t)SH::pair_base(NS + NC) * PAIR_BYTES, (size_t)SH::oblk_base(NS) * 32 * 4);
  if constexpr (SPLIT)
    copy16(smem + LO_BASE, image + 2 * (size_t)SH::pair_base(NS + NC) * PAIR_BYTES + (size_t)SH::oblk_base(NS + NC) * 32 * 4,
           (size_t)BIAS_BASE);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hi = lane >> 5, j = lane & 31;
  const int ntz = (nz + 31) / 32;
  const int64_t ntiles = (int64_t)nx * ny * ntz;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    asm volatile("" ::: "memory");
    const int64_t col = tile / ntz;
    const int k = (int)(tile - col * ntz) * 32 + j;
    const int ix = (int)(col / ny), iy = (int)(col - (int64_t)ix * ny);
    const bool in_range = k < nz;
    float p[3] = {tx[ix], ty[iy], in_range ? tz[k] : 0.0f};
    const bool inside = in_range && (occ_bits == nullptr || occ_point_test(occ_bits, occ_n, p[0], p[1], p[2]));
    const int64_t vox = col * nz + k;
    if (__ballot(inside) == 0ull) {
      if (hi == 0 && in_range) sdf[vox] = outside;
      continue;
    }
#pragma unroll
    for (int dd = 0; dd < 3; ++dd) p[dd] = fminf(fmaxf(p[dd], -1.0f), 1.0f);      // run_network_density clips (nerf_runner.py:1313)
    float x[1][16];
#if NOF_GRID_PRIO
    __builtin_amdgcn_s_setprio(3);                     // (as in k_enc_mlp_fwd: the gather phase issues first)
#endif
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const int level = 8 * hi + kk;
      float2 v = make_float2(0.f, 0.f);
      if (inside && level < g.L) {
        const HashLevel lv = load_level(g, level);
        v = encode_level(lv, table, locate3(p, lv.scale));
      }
      x[0][2 * kk] = v.x;
      x[0][2 * kk + 1] = v.y;
    }
#if NOF_GRID_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    float h[2][16], so[1][16];
    dense_o1<P, 1, 2, SPLIT>(smem, FW_OFF(0), BIAS_OFF(0), x, h, lane, LO_OFF(0));
    relu_mask<2>(h);
#pragma unroll
    for (int l = 1; l < NS - 1; ++l) {
      float h2[2][16];
      dense_o1<P, 2, 2, SPLIT>(smem, FW_OFF(l), BIAS_OFF(l), h, h2, lane, LO_OFF(l));
      relu_mask<2>(h2);
#pragma unroll
      for (int pp = 0; pp < 2; ++pp)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[pp][r] = h2[pp][r];
    }
    dense_o1<P, 2, 1, SPLIT>(smem, FW_OFF(NS - 1), BIAS_OFF(NS - 1), h, so, lane, LO_OFF(NS - 1));
    if (hi == 0 && in_range) sdf[vox] = inside ? so[0][0] : outside;
  }
}

// =====================================================================================================
// backward (forward recomputed): dfeat, dview, per-workgroup dW/db partials
// =====================================================================================================
// dIn of input block q, sample-per-lane (reg r = input slot (q,hi,r)); bw_off = compile-time byte offset of the layer's fragments
template <class P, int PN>
__device__ __forceinline__ void bwd_data(const char* smem, int bw_off, int q, const float (&dout1)[PN][16], float (&din1)[16],
                                         int lane) {
  constexpr int KR = P::KR, NSTEP = 16 / KR;
  constexpr int FB = 64 * KR * (int)sizeof(typename P::elem);
  const char* fl = smem + lane * (KR * (int)sizeof(typename P::elem));
  f32x16 a1;
#pragma unroll
  for (int r = 0; r < 16; ++r) a1[r] = 0.0f;
  if constexpr (PN * NSTEP >= 8) {                                     // long chain: fragments requested up front (see dense_o1)
    typename P::frag w[PN * NSTEP];
#pragma unroll
    for (int t = 0; t < PN * NSTEP; ++t) w[t] = *(const typename P::frag*)(fl + bw_off + (q * PN * NSTEP + t) * FB);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int p = 0; p < PN; ++p)
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) a1 = P::mma(w[p * NSTEP + s], P::pack(&dout1[p][KR * s]), a1);
  } else {
#pragma unroll
    for (int p = 0; p < PN; ++p)
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) {
        const typename P::frag w = *(const typename P::frag*)(fl + bw_off + ((q * PN + p) * NSTEP + s) * FB);
        a1 = P::mma(w, P::pack(&dout1[p][KR * s]), a1);
      }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) din1[r] = a1[r];
}

// The matrix core as a transpose engine: x is a 32-sample x 32-slot block held sample-per-lane (reg r = slot (hi,r));
// D[sample][n] = sum_k X[sample][k] I[k][n] with I = identity returns it slot-per-lane (lane n = slot with
// nloc(hi,r) == n, reg r' = sample nloc(hi',r')), i.e. exactly the A/B operand layout of the sample-contracted dW MFMA.
// Multiplying by 1 and adding 0 is exact: y holds the operand-rounded values of x.  `ident` = the identity fragments,
// built once per wave.
template <class P>
struct Ident {
  typename P::frag f[16 / P::KR];
  __device__ __forceinline__ void init(int lane) {
    const int hi = lane >> 5, j = lane & 31;
#pragma unroll
    for (int s = 0; s < 16 / P::KR; ++s) {
      float id[P::KR];
#pragma unroll
      for (int t = 0; t < P::KR; ++t) id[t] = (nloc(hi, P::KR * s + t) == j) ? 1.0f : 0.0f;
      f[s] = P::pack(id);
    }
  }
};

// The same fragments kept in LDS ([step][lane], written once per workgroup by build()) and read where they are used: eight
// registers less across the persistent loop of a kernel that runs at its register cap (three colour layers).
template <class P>
struct IdentLds {
  const typename P::frag* base;                                      // + lane
  __device__ __forceinline__ void build(char* smem_at, int lane) {
    Ident<P> I;
    I.init(lane);
    typename P::frag* w = reinterpret_cast<typename P::frag*>(smem_at) + lane;
    if (threadIdx.x < 64) {
#pragma unroll
      for (int s = 0; s < 16 / P::KR; ++s) w[s * 64] = I.f[s];
    }
    base = w;
  }
  __device__ __forceinline__ typename P::frag get(int s) const { return base[s * 64]; }
};
template <class P> __device__ __forceinline__ typename P::frag ident_frag(const Ident<P>& I, int s) { return I.f[s]; }
template <class P> __device__ __forceinline__ typename P::frag ident_frag(const IdentLds<P>& I, int s) { return I.get(s); }

template <class P, class ID>
__device__ __forceinline__ void transpose32(const ID& I, const float (&x)[16], float (&y)[16]) {
  constexpr int KR = P::KR, NSTEP = 16 / KR;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) acc = P::mma(P::pack(&x[KR * s]), ident_frag<P>(I, s), acc);
#pragma unroll
  for (int r = 0; r < 16; ++r) y[r] = acc[r];
}

// Orientation-2 (slot-per-lane) copies of every layer INPUT = the B operands of the dW MFMAs.  In the 16-bit modes they
// live in lane-private LDS slots (written and read by the same lane: no barrier); in fp32 (parity) mode in registers.
template <class P, int NSLOT, bool IN_LDS>
struct In2Store;
template <class P, int NSLOT>
struct In2Store<P, NSLOT, true> {
  static constexpr int NSTEP = 16 / P::KR;
  typename P::frag* base;                         // wave-private region + lane, indexed [slot][step] with stride 64 fragments
  __device__ __forceinline__ void put(int slot, int s, typename P::frag f) { base[(slot * NSTEP + s) * 64] = f; }
  __device__ __forceinline__ typename P::frag get(int slot, int s) const { return base[(slot * NSTEP + s) * 64]; }
};
template <class P, int NSLOT>
struct In2Store<P, NSLOT, false> {
  static constexpr int NSTEP = 16 / P::KR;
  typename P::frag r[NSLOT][NSTEP];
  __device__ __forceinline__ void put(int slot, int s, typename P::frag f) { r[slot][s] = f; }
  __device__ __forceinline__ typename P::frag get(int slot, int s) const { return r[slot][s]; }
};

// transpose one sample-per-lane block and park it as MFMA operands in slot `slot`
template <class P, class ST, class ID>
__device__ __forceinline__ void park_o2(ST& st, const ID& I, int slot, const float (&x)[16]) {
  float y[16];
  transpose32<P>(I, x, y);
#pragma unroll
  for (int s = 0; s < 16 / P::KR; ++s) st.put(slot, s, P::pack(&y[P::KR * s]));
}

// one output block p of layer l:  g2 = T(g1[p]);  db += sum_samples g2;  dW[p][q] += g2 (x) in2(l,q)
// TR: the operands swapped -- dW^T[slot of block q][neuron of block p], registers = input slots (used where only the first NACC
// slot rows of every input block can be non-zero: colour layer 0)
template <class P, int QN, int NACC, class ST, bool TR = false, class ID = Ident<P>>
__device__ __forceinline__ void dw_block(float (&dw)[2][16], float* db_lane, const ID& I, const float (&g1p)[16],
                                         const ST& st, int slot0) {
  constexpr int KR = P::KR, NSTEP = 16 / KR;
  float g2[16];
  transpose32<P>(I, g1p, g2);
  float sdb = 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) sdb += g2[r];
  *db_lane += sdb;                                   // lane-private LDS word (bias gradients need no register)
  typename P::frag ga[NSTEP];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) ga[s] = P::pack(&g2[KR * s]);
#pragma unroll
  for (int q = 0; q < QN; ++q) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = r < NACC ? dw[q][r] : 0.0f;
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) acc = TR ? P::mma(st.get(slot0 + q, s), ga[s], acc) : P::mma(ga[s], st.get(slot0 + q, s), acc);
#pragma unroll
    for (int r = 0; r < NACC; ++r) dw[q][r] = acc[r];
  }
}

// The workgroup's dW / db accumulators of layers [LA, LB) -> ITS row of `partials` (row = workgroup).  The four waves' registers are
// summed through LDS first -- plain stores and lane-private read-modify-writes, one wave after the other with a barrier between
// (the fragments at the start of the LDS block are dead once every wave has left the tile loop; LDS float ATOMICS cost ~70 us here:
// they retire ~1 lane per 4.5 cycles on gfx950) -- and wave 0 writes the totals with plain global stores: for one register every
// lane owns a distinct (row, column) and the (lane, register) pairs cover every parameter exactly once, so no zero-fill and no
// global atomics are needed.  One row per workgroup instead of one per wave is a quarter of the bytes nof_reduce_partials reads
// (cfg2: 75 MB -> 19 MB per step).  `db_stride`: floats between two waves' lane-private bias sums.  Deterministic: ((w0+w1)+w2)+w3.
// floats of LDS flush_dw's reduction needs per group of four waves (x 256 bytes): must end below the lane-private bias sums
template <class SH, int LA, int LB>
constexpr int flush_dw_accs() {
  int a = 0;
  for (int l = LA; l < LB; ++l) a += SH::pn(l) * SH::qn(l) * SH::nacc(l);
  return a;
}
// NW waves per workgroup (4 or 8): every GROUP of four consecutive waves is reduced on its own (its own LDS region, the same
// barriers) and writes its own row, 4 groups' worth of rows per launch being what the host sized `partials` for.
template <class SH, int LA, int LB, int NW = 4>
__device__ __forceinline__ void flush_dw(const NofMlpDesc& d, float (&dw)[SH::NL][2][2][16], const float* dbw, int db_stride,
                                         char* smem, float* __restrict__ partials, float unscale, int wave_s) {
  static_assert(NW % 4 == 0, "waves are reduced in groups of four");
  // lane id from mbcnt, not from threadIdx.x: the work-item id register is long overwritten by the end of the persistent loop
  // and a copy kept for this epilogue was the one value the three-colour-layer kernel spilled
  const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
  const int hi = lane >> 5, j = lane & 31;
  const int hi_j = (j >> 2) & 1, r_j = (j & 3) + 4 * (j >> 3);      // dW column lane j = input slot (hi_j, r_j)
  const int grp = wave_s >> 2, wv = wave_s & 3;                     // wave_s: SCALAR wave index (the phases below are whole-wave branches)
  float* red = reinterpret_cast<float*>(smem) + grp * (flush_dw_accs<SH, LA, LB>() * 64) + lane;
  // phase W: 0 = store, 1 / 2 = add into LDS, 3 = add LDS into the registers (the last wave keeps the totals).  Straight-line per
  // phase, so that a wave's 70-110 LDS reads are all in flight together (one branch per element serialised their latencies: 30 us)
  auto phase = [&](auto W) {
    constexpr int w = decltype(W)::value;
    int a = 0;
#pragma unroll
    for (int l = LA; l < LB; ++l)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (p < SH::pn(l) && q < SH::qn(l) && r < SH::nacc(l)) {
              if constexpr (w == 0) red[a * 64] = dw[l][p][q][r];
              else if constexpr (w < 3) red[a * 64] += dw[l][p][q][r];
              else dw[l][p][q][r] += red[a * 64];
              ++a;
            }
  };
  __syncthreads();                                                  // every wave is out of the tile loop
  if (wv == 0) phase(std::integral_constant<int, 0>());
  __syncthreads();
  if (wv == 1) phase(std::integral_constant<int, 1>());
  __syncthreads();
  if (wv == 2) phase(std::integral_constant<int, 2>());
  __syncthreads();
  if (wv == 3) phase(std::integral_constant<int, 3>());
  if (wv != 3) return;
  float* __restrict__ dst = partials + ((size_t)blockIdx.x * (NW / 4) + grp) * d.n_params;
  const float* db0 = dbw - 3 * db_stride;                          // the group's first wave's lane-private sums (this is its fourth)
#pragma unroll
  for (int l = LA; l < LB; ++l) {
    const int in_dim = d.in_dim[l], out_dim = d.out_dim[l];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      if (p < SH::pn(l)) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          if (q < SH::qn(l)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              if (r < SH::nacc(l)) {
                // normal block: lane = input slot (hi_j, r_j), register = neuron row nloc(hi, r); transposed block (SH::tr): lane =
                // neuron row j, register = input slot (hi, r)
                const int col = SH::tr(l) ? inmap(d, l, q, hi, r) : inmap(d, l, q, hi_j, r_j);
                const int row = 32 * p + (SH::tr(l) ? j : nloc(hi, r));
                if (col >= 0 && row < out_dim) dst[d.w_off[l] + row * in_dim + col] = dw[l][p][q][r] * unscale;
              }
            }
          }
        }
        const float* dbp = db0 + (2 * l + p) * 64;                   // the four waves' sums of lanes (hi, j): the pair shares row j
        float v = ((dbp[0] + dbp[db_stride]) + dbp[2 * db_stride]) + dbp[3 * db_stride];
        v += __shfl_xor(v, 32, 64);
        const int row = 32 * p + j;
        if (hi == 0 && row < out_dim) dst[d.b_off[l] + row] = v * unscale;
      }
    }
  }
}

template <class P, int NS, int NC>
__global__ __launch_bounds__(256) void k_mlp_bwd(NofMlpDesc d, const char* __restrict__ image,
                                                  const float2* __restrict__ feat, int L, const float* __restrict__ view,
                                                  int S, const float4* __restrict__ draw, float2* __restrict__ dfeat,
                                                  float* __restrict__ dview, float* __restrict__ partials, int64_t B,
                                                  const void* __restrict__ tile_list) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef Shp<NS, NC> SH;
  constexpr int NL = NS + NC;
  constexpr int KR = P::KR, NSTEP = 16 / KR;
  typedef typename P::frag frag;
  constexpr int BW_BASE = SH::pair_base(NL) * PAIR_BYTES;
  constexpr int BIAS_BASE = 2 * BW_BASE;
  constexpr int IN2_BASE = BIAS_BASE + SH::oblk_base(NL) * 32 * 4;
  constexpr bool IN2_LDS = (KR == 8);
  constexpr int NSLOT = 2 * NL;                       // slot(l, q) = 2 l + q : input block q of layer l
  constexpr int IN2_WAVE = NSLOT * NSTEP * 64 * (int)sizeof(frag);
  constexpr int DB_BASE = IN2_BASE + (IN2_LDS ? 4 * IN2_WAVE : 0);
  copy16(smem, image, (size_t)IN2_BASE);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hi = lane >> 5, j = lane & 31;
  typedef In2Store<P, NSLOT, IN2_LDS> Store;
  Store st;
  if constexpr (IN2_LDS) st.base = (frag*)(smem + IN2_BASE + wave * IN2_WAVE) + lane;
  float* dbw = (float*)(smem + DB_BASE) + wave * NSLOT * 64 + lane;   // bias-gradient partial sums: [2 l + p][lane], lane-private
#pragma unroll
  for (int k = 0; k < NSLOT; ++k) dbw[k * 64] = 0.0f;
  Ident<P> I;
  I.init(lane);
  // loss scaling of the 16-bit backward (the reference's GradScaler, nerf_runner.py:159,758): the loss gradient is multiplied
  // by a power of two where it enters and every fp32 output is divided by it where it leaves -- exact in fp32, and it keeps
  // the ~1e-7 gradients of a 1/(R*S)-normalised loss out of binary16's subnormal range inside the MFMA operands
  const float gscale = d.grad_scale > 0.0f ? d.grad_scale : 1.0f, gunscale = 1.0f / gscale;

  float dw[NL][2][2][16];                             // persistent per-wave dW accumulators (only the live entries are touched)
#pragma unroll
  for (int l = 0; l < NL; ++l)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (p < SH::pn(l) && q < SH::qn(l) && r < SH::nacc(l)) dw[l][p][q][r] = 0.0f;

  const int64_t ntiles = (B + 31) / 32;
  const TileWork work(tile_list, ntiles);
  for (int64_t wi = (int64_t)blockIdx.x * 4 + wave; wi < work.n; wi += (int64_t)gridDim.x * 4) {
    asm volatile("" ::: "memory");                    // keep the weight fragments in LDS (no hoisting into VGPRs)
    const int64_t tile = work.at(wi);
    const int64_t t0 = tile * 32;
    const int64_t b = t0 + j;
    // ---------------- forward recompute (sample-per-lane), parking the transposed layer inputs ----------------
    uint32_t m1[NL];                                  // ReLU masks of the hidden layers' outputs
    float h[2][16];
    {
      float x[1][16];
      load_feat_o1(feat, L, B, b, hi, x);
      park_o2<P>(st, I, 0, x[0]);
      dense_o1<P, 1, 2>(smem, FW_OFF(0), BIAS_OFF(0), x, h, lane);
      m1[0] = relu_mask<2>(h);
    }
#pragma unroll
    for (int l = 1; l < NS - 1; ++l) {
      park_o2<P>(st, I, 2 * l, h[0]);
      park_o2<P>(st, I, 2 * l + 1, h[1]);
      float hn[2][16];
      dense_o1<P, 2, 2>(smem, FW_OFF(l), BIAS_OFF(l), h, hn, lane);
      m1[l] = relu_mask<2>(hn);
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[p][r] = hn[p][r];
    }
    park_o2<P>(st, I, 2 * (NS - 1), h[0]);
    park_o2<P>(st, I, 2 * (NS - 1) + 1, h[1]);
    {
      float cin[2][16], so[1][16];
      dense_o1<P, 2, 1>(smem, FW_OFF(NS - 1), BIAS_OFF(NS - 1), h, so, lane);
#pragma unroll
      for (int r = 0; r < 16; ++r) cin[0][r] = so[0][r];
      load_view_o1(view, S, B, b, hi, cin[1]);
      park_o2<P>(st, I, 2 * NS, cin[0]);
      park_o2<P>(st, I, 2 * NS + 1, cin[1]);
      dense_o1<P, 2, 2>(smem, FW_OFF(NS), BIAS_OFF(NS), cin, h, lane);
      m1[NS] = relu_mask<2>(h);
    }
#pragma unroll
    for (int l = NS + 1; l < NL - 1; ++l) {
      park_o2<P>(st, I, 2 * l, h[0]);
      park_o2<P>(st, I, 2 * l + 1, h[1]);
      float hn[2][16];
      dense_o1<P, 2, 2>(smem, FW_OFF(l), BIAS_OFF(l), h, hn, lane);
      m1[l] = relu_mask<2>(hn);
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[p][r] = hn[p][r];
    }
    park_o2<P>(st, I, 2 * (NL - 1), h[0]);
    park_o2<P>(st, I, 2 * (NL - 1) + 1, h[1]);
    // (the last colour layer's output is not needed: its gradient comes from draw)

    // ---------------- backward ----------------
    float g1[2][16];                                  // dOut of the current layer, sample-per-lane (block 1 unused when PN = 1)
    float dsdf1 = 0.0f;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) g1[p][r] = 0.0f;
    if (hi == 0 && b < B) {                           // draw[b] = (d rgb_raw[3], d sdf)
      const float4 t = draw[b];
      g1[0][0] = t.x * gscale; g1[0][1] = t.y * gscale; g1[0][2] = t.z * gscale;
      dsdf1 = t.w * gscale;
    }
    // ---- colour net: head down to colour layer 1 ----
#pragma unroll
    for (int l = NL - 1; l > NS; --l) {
      float d1[2][16];
      if (l == NL - 1) {
        dw_block<P, 2, 4>(dw[l][0], dbw + (2 * l) * 64, I, g1[0], st, 2 * l);
        float ga[1][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ga[0][r] = g1[0][r];
        bwd_data<P, 1>(smem, BW_OFF(l), 0, ga, d1[0], lane);
        bwd_data<P, 1>(smem, BW_OFF(l), 1, ga, d1[1], lane);
      } else {
        dw_block<P, 2, 16>(dw[l][0], dbw + (2 * l) * 64, I, g1[0], st, 2 * l);
        dw_block<P, 2, 16>(dw[l][1], dbw + (2 * l + 1) * 64, I, g1[1], st, 2 * l);
        bwd_data<P, 2>(smem, BW_OFF(l), 0, g1, d1[0], lane);
        bwd_data<P, 2>(smem, BW_OFF(l), 1, g1, d1[1], lane);
      }
      apply_mask<2>(d1, m1[l - 1]);
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) g1[p][r] = d1[p][r];
    }
    // ---- colour layer 0: inputs = [sigma-out block | view block] ----
    {
      dw_block<P, 2, 8, Store, true>(dw[NS][0], dbw + (2 * NS) * 64, I, g1[0], st, 2 * NS);
      dw_block<P, 2, 8, Store, true>(dw[NS][1], dbw + (2 * NS + 1) * 64, I, g1[1], st, 2 * NS);
      float ds1[16], dv1[16], dv2[16];
      bwd_data<P, 2>(smem, BW_OFF(NS), 0, g1, ds1, lane);
      bwd_data<P, 2>(smem, BW_OFF(NS), 1, g1, dv1, lane);
      transpose32<P>(I, dv1, dv2);
      // dview[ray][u] += sum over the tile's samples (lane = view slot, regs <-> samples; a tile may straddle two rays)
      {
        const int64_t ray0 = t0 / S;                                 // one (wave-uniform) division per tile
        const int64_t end0 = (ray0 + 1) * S, endB = end0 < B ? end0 : B;
        float sa = 0.0f, sb = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t bs = t0 + nloc(hi, r);
          if (bs < endB) sa += dv2[r];
          else if (bs < B) sb += dv2[r];
        }
        sa += __shfl_xor(sa, 32, 64);
        sb += __shfl_xor(sb, 32, 64);
        const int u = view_col_of_lane(j);
        if (hi == 0 && u >= 0 && u < d.n_view) {
          if (sa != 0.0f) atomicAdd(&dview[ray0 * NOF_VIEW_COLS + u], sa * gunscale);
          if (sb != 0.0f) atomicAdd(&dview[(ray0 + 1) * NOF_VIEW_COLS + u], sb * gunscale);
        }
      }
      // gradient of the sigma net output block: geo_feat grads + the loss' own d sdf (output 0 = hi 0, reg 0)
#pragma unroll
      for (int r = 0; r < 16; ++r) { g1[0][r] = ds1[r]; g1[1][r] = 0.0f; }
      if (hi == 0) g1[0][0] += dsdf1;
    }
    // ---- sigma net: head down to layer 1 ----
#pragma unroll
    for (int l = NS - 1; l >= 1; --l) {
      float d1[2][16];
      if (l == NS - 1) {
        dw_block<P, 2, 8>(dw[l][0], dbw + (2 * l) * 64, I, g1[0], st, 2 * l);
        float ga[1][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ga[0][r] = g1[0][r];
        bwd_data<P, 1>(smem, BW_OFF(l), 0, ga, d1[0], lane);
        bwd_data<P, 1>(smem, BW_OFF(l), 1, ga, d1[1], lane);
      } else {
        dw_block<P, 2, 16>(dw[l][0], dbw + (2 * l) * 64, I, g1[0], st, 2 * l);
        dw_block<P, 2, 16>(dw[l][1], dbw + (2 * l + 1) * 64, I, g1[1], st, 2 * l);
        bwd_data<P, 2>(smem, BW_OFF(l), 0, g1, d1[0], lane);
        bwd_data<P, 2>(smem, BW_OFF(l), 1, g1, d1[1], lane);
      }
      apply_mask<2>(d1, m1[l - 1]);
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) g1[p][r] = d1[p][r];
    }
    // ---- sigma layer 0 ----
    {
      dw_block<P, 1, 16>(dw[0][0], dbw, I, g1[0], st, 0);
      dw_block<P, 1, 16>(dw[0][1], dbw + 64, I, g1[1], st, 0);
      float df1[16];
      bwd_data<P, 2>(smem, BW_OFF(0), 0, g1, df1, lane);
      store_dfeat_o1(dfeat, L, B, b, hi, df1, gunscale);
    }
  }

  // ---------------- reduce the workgroup's dW/db and write its row of `partials` ----------------
  static_assert(flush_dw_accs<SH, 0, NL>() * 256 <= DB_BASE, "flush_dw's LDS reduction would overwrite the bias sums");
  flush_dw<SH, 0, NL>(d, dw, dbw, NSLOT * 64, smem, partials, gunscale, __builtin_amdgcn_readfirstlane(wave));
}

// =====================================================================================================
// backward split by network (16-bit modes): k_mlp_bwd_color then k_mlp_bwd_sigma.
// The fused kernel above keeps all 150 dW registers of both networks and therefore runs 1 wave/SIMD, where hipcc puts every
// MFMA result in AGPRs (824 of the 3450 instructions per tile were AGPR<->VGPR moves) and nothing overlaps a wave's LDS
// round trips.  Each half needs fewer accumulators (colour 72, sigma 112 for 3x64 + 2x64), fits 256 registers and runs
// 2 waves/SIMD with the MFMA results in VGPRs.  The halves exchange two [B,16] operand-precision arrays: the sigma head's
// output (written by the forward kernel) and its gradient -- 64 B/sample of extra traffic against ~300 B/sample saved
// instructions' worth of time.  Numerically identical to the fused kernel (same operand roundings, same MFMA chains).
// =====================================================================================================
// Waves per workgroup of the two split-backward kernels.  Three colour layers (the reference's own shape, nerf_runner.py:221): the
// fragments (40 KB) + four waves' parked operands (48 KB) + bias sums are 95 KB, i.e. ONE 4-wave workgroup per CU = one wave per
// SIMD; eight waves around ONE copy of the fragments are 148 KB: one workgroup per CU, two waves per SIMD.
#ifndef NOF_BWD_PRIO
#define NOF_BWD_PRIO 0                                    // s_setprio around the loads of a tile in the split backward kernels (A/B: profiles/r05_t_*)
#endif
#ifndef NOF_BWD_WAVES_C2
#define NOF_BWD_WAVES_C2 4
#endif
#ifndef NOF_BWD_WAVES_S
#define NOF_BWD_WAVES_S 4
#endif
template <int NC> struct ColorWaves { static constexpr int value = NC >= 3 ? 8 : NOF_BWD_WAVES_C2; };
template <int NS> struct SigmaWaves { static constexpr int value = NOF_BWD_WAVES_S; };

template <class P, int NS, int NC>
__global__ __launch_bounds__(64 * ColorWaves<NC>::value, 2) void k_mlp_bwd_color(NofMlpDesc d, const char* __restrict__ image,
                                                           const typename P::elem* __restrict__ sig,
                                                           const float* __restrict__ view, int S,
                                                           const float4* __restrict__ draw, typename P::elem* __restrict__ dsig,
                                                           float* __restrict__ dview, float* __restrict__ partials, int64_t B,
                                                           const void* __restrict__ tile_list) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef Shp<NS, NC> SH;
  constexpr int NL = NS + NC;
  constexpr int KR = P::KR, NSTEP = 16 / KR;
  typedef typename P::frag frag;
  constexpr int PA = SH::pair_base(NS), PB = SH::pair_base(NL), NP = PB - PA;       // the colour layers' fragment pairs
  constexpr int OA = SH::oblk_base(NS), OB = SH::oblk_base(NL);
  constexpr int BWB = NP * PAIR_BYTES, BIASB = 2 * NP * PAIR_BYTES;
  constexpr int IN2_BASE = BIASB + (OB - OA) * 128;
  constexpr int NSLOT = 2 * NC;                                                     // slot(l, q) = 2 (l - NS) + q
  constexpr int IN2_WAVE = NSLOT * NSTEP * 64 * (int)sizeof(frag);
  constexpr int NW = ColorWaves<NC>::value;
  constexpr int DB_BASE = IN2_BASE + NW * IN2_WAVE;
  copy16(smem, image + (size_t)PA * PAIR_BYTES, (size_t)NP * PAIR_BYTES);
  copy16(smem + BWB, image + (size_t)(PB + PA) * PAIR_BYTES, (size_t)NP * PAIR_BYTES);
  copy16(smem + BIASB, image + 2 * (size_t)PB * PAIR_BYTES + OA * 128, (size_t)(OB - OA) * 128);
  // identity fragments of the MFMA transposes: in LDS where the registers are all taken (three colour layers), else in registers
  constexpr int ID_BASE = DB_BASE + NW * (2 * NC) * 64 * 4;
  typedef typename std::conditional<(NC >= 3), IdentLds<P>, Ident<P>>::type IdT;
  IdT I;
  if constexpr (NC >= 3) I.build(smem + ID_BASE, threadIdx.x & 63);
  else I.init(threadIdx.x & 63);
  __syncthreads();
#define CFW(l) ((SH::pair_base(l) - PA) * PAIR_BYTES)
#define CBW(l) (BWB + (SH::pair_base(l) - PA) * PAIR_BYTES)
#define CBIAS(l) (BIASB + (SH::oblk_base(l) - OA) * 128)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hi = lane >> 5, j = lane & 31;
  typedef In2Store<P, NSLOT, true> Store;
  Store st;
  st.base = (frag*)(smem + IN2_BASE + wave * IN2_WAVE) + lane;
  float* dbl = (float*)(smem + DB_BASE) + wave * (2 * NC) * 64 + lane;              // [2 (l - NS) + p][lane]
#pragma unroll
  for (int k = 0; k < 2 * NC; ++k) dbl[k * 64] = 0.0f;
  float* dbw = dbl - (2 * NS) * 64;                                                 // so that [2 l + p] addresses it (never dereferenced below 2 NS)
  // loss scaling of the 16-bit backward (the reference's GradScaler, nerf_runner.py:159,758): the loss gradient is multiplied
  // by a power of two where it enters and every fp32 output is divided by it where it leaves -- exact in fp32, and it keeps
  // the ~1e-7 gradients of a 1/(R*S)-normalised loss out of binary16's subnormal range inside the MFMA operands
  const float gscale = d.grad_scale > 0.0f ? d.grad_scale : 1.0f, gunscale = 1.0f / gscale;
  float dw[NL][2][2][16];
#pragma unroll
  for (int l = NS; l < NL; ++l)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (p < SH::pn(l) && q < SH::qn(l) && r < SH::nacc(l)) dw[l][p][q][r] = 0.0f;

  const int64_t ntiles = (B + 31) / 32;
  const int64_t tstride = (int64_t)gridDim.x * NW;
  // the NEXT tile's inputs are loaded a whole tile ahead (latency hidden behind this tile's MFMA chain) where the 20 registers
  // cost no heavy spilling (two colour layers; with three: 244 B of scratch per lane), draw at the top of the tile
  constexpr bool AHEAD = NC == 2;
  const TileWork work(tile_list, ntiles);
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int64_t w0 = (int64_t)blockIdx.x * NW + wave_s;
  int64_t tile_n = work.at(w0);                       // the tile whose inputs are in flight
  typename P::frag sign = load_sig_raw<P>(sig, B, AHEAD ? tile_n * 32 + j : B, hi);
  float viewn[16];
  load_view_o1(view, S, B, AHEAD ? tile_n * 32 + j : B, hi, viewn);
  // draw[b] = (d rgb_raw[3], d sdf), also a tile ahead: it decides whether the tile has anything to do
  float4 drn = make_float4(0.f, 0.f, 0.f, 0.f);
  if (hi == 0 && tile_n * 32 + j < B) drn = draw[tile_n * 32 + j];
  for (int64_t wi = w0; wi < work.n; wi += tstride) {
    asm volatile("" ::: "memory");
#if NOF_BWD_PRIO
    __builtin_amdgcn_s_setprio(NOF_BWD_PRIO == 1 ? 3 : 0);   // 1: the tile's loads (and the next tile's prefetch) ahead of the other wave's chain; 2: the chain ahead
#endif
    const int64_t tile = tile_n;
    tile_n = work.at(wi + tstride);
    const int64_t t0 = tile * 32;
    const int64_t b = t0 + j;
    uint32_t m1[NL];
    float h[2][16];
    float4 dr = drn;
    if constexpr (AHEAD) {
      asm volatile("" : "+v"(dr.x), "+v"(dr.y), "+v"(dr.z), "+v"(dr.w));
      drn = make_float4(0.f, 0.f, 0.f, 0.f);
      if (hi == 0 && tile_n * 32 + j < B) drn = draw[tile_n * 32 + j];
    } else {                                           // (three colour layers: no register to carry it across a tile)
      dr = make_float4(0.f, 0.f, 0.f, 0.f);
      if (hi == 0 && b < B) dr = draw[b];
    }
    float cin[2][16];
    if constexpr (AHEAD) {
      sig_to_o1<P>(sign, cin[0]);
#pragma unroll
      for (int r = 0; r < 16; ++r) cin[1][r] = viewn[r];
      pin16(cin[0]);
      pin16(cin[1]);
      sign = load_sig_raw<P>(sig, B, tile_n * 32 + j, hi);
      load_view_o1(view, S, B, tile_n * 32 + j, hi, viewn);
    }
    // A tile whose 32 loss gradients are all EXACTLY zero (background rays, free-space samples whose loss has saturated: two
    // thirds of a cfg2 batch once the field has settled) contributes exactly nothing to dW, db, dview and dsig: skipped, with
    // dsig = 0 written for the sigma kernel.  Same sums, less work (north_star's per-wavefront compaction, applied where
    // the zeros are: tools/zero_grad_probe.py).
    // (with a work list the test is never true: the list holds exactly the tiles with a non-zero row)
    const bool skip = tile_list == nullptr &&
                      __builtin_amdgcn_ballot_w64(dr.x != 0.0f || dr.y != 0.0f || dr.z != 0.0f || dr.w != 0.0f) == 0ull;
    float ds1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) ds1[r] = 0.0f;
#if NOF_BWD_PRIO
    __builtin_amdgcn_s_setprio(NOF_BWD_PRIO == 1 ? 0 : 3);
#endif
    if (!skip) {
    {
      if constexpr (!AHEAD) {
        load_sig_tile_o1<P>(sig, B, tile, j, hi, cin[0]);
        load_view_off_o1(view, S, B, b, hi, cin[1]);
      }
      park_o2<P>(st, I, 0, cin[0]);
      park_o2<P>(st, I, 1, cin[1]);
      dense_o1<P, 2, 2>(smem, CFW(NS), CBIAS(NS), cin, h, lane);
      m1[NS] = relu_mask<2>(h);
    }
#pragma unroll
    for (int l = NS + 1; l < NL - 1; ++l) {
      park_o2<P>(st, I, 2 * (l - NS), h[0]);
      park_o2<P>(st, I, 2 * (l - NS) + 1, h[1]);
      float hn[2][16];
      dense_o1<P, 2, 2>(smem, CFW(l), CBIAS(l), h, hn, lane);
      m1[l] = relu_mask<2>(hn);
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[p][r] = hn[p][r];
    }
    park_o2<P>(st, I, 2 * (NL - 1 - NS), h[0]);
    park_o2<P>(st, I, 2 * (NL - 1 - NS) + 1, h[1]);

    float g1[2][16];
    float dsdf1 = 0.0f;
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) g1[p][r] = 0.0f;
    if (hi == 0 && b < B) {
      g1[0][0] = dr.x * gscale; g1[0][1] = dr.y * gscale; g1[0][2] = dr.z * gscale;
      dsdf1 = dr.w * gscale;
    }
#pragma unroll
    for (int l = NL - 1; l > NS; --l) {
      float d1[2][16];
      if (l == NL - 1) {
        dw_block<P, 2, 4>(dw[l][0], dbw + (2 * l) * 64, I, g1[0], st, 2 * (l - NS));
        float ga[1][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ga[0][r] = g1[0][r];
        bwd_data<P, 1>(smem, CBW(l), 0, ga, d1[0], lane);
        bwd_data<P, 1>(smem, CBW(l), 1, ga, d1[1], lane);
      } else {
        dw_block<P, 2, 16>(dw[l][0], dbw + (2 * l) * 64, I, g1[0], st, 2 * (l - NS));
        dw_block<P, 2, 16>(dw[l][1], dbw + (2 * l + 1) * 64, I, g1[1], st, 2 * (l - NS));
        bwd_data<P, 2>(smem, CBW(l), 0, g1, d1[0], lane);
        bwd_data<P, 2>(smem, CBW(l), 1, g1, d1[1], lane);
      }
      apply_mask<2>(d1, m1[l - 1]);
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) g1[p][r] = d1[p][r];
    }
    {
      dw_block<P, 2, 8, Store, true>(dw[NS][0], dbw + (2 * NS) * 64, I, g1[0], st, 0);
      dw_block<P, 2, 8, Store, true>(dw[NS][1], dbw + (2 * NS + 1) * 64, I, g1[1], st, 0);
      float dv1[16], dv2[16];
      bwd_data<P, 2>(smem, CBW(NS), 0, g1, ds1, lane);
      bwd_data<P, 2>(smem, CBW(NS), 1, g1, dv1, lane);
      transpose32<P>(I, dv1, dv2);
      {
        const int64_t ray0 = t0 / S;
        const int64_t end0 = (ray0 + 1) * S, endB = end0 < B ? end0 : B;
        float sa = 0.0f, sb = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t bs = t0 + nloc(hi, r);
          if (bs < endB) sa += dv2[r];
          else if (bs < B) sb += dv2[r];
        }
        sa += __shfl_xor(sa, 32, 64);
        sb += __shfl_xor(sb, 32, 64);
        const int u = view_col_of_lane(j);
        if (hi == 0 && u >= 0 && u < d.n_view) {
          if (sa != 0.0f) atomicAdd(&dview[ray0 * NOF_VIEW_COLS + u], sa * gunscale);
          if (sb != 0.0f) atomicAdd(&dview[(ray0 + 1) * NOF_VIEW_COLS + u], sb * gunscale);
        }
      }
      if (hi == 0) ds1[0] += dsdf1;                    // the loss' own d sdf joins the geo_feat gradients (output 0 = hi 0, reg 0)
    }
    }                                                    // if (!skip)
    store_sig_o1<P>(dsig, B, b, hi, ds1);                // zeros for a skipped tile
  }
  static_assert(flush_dw_accs<SH, NS, NL>() * 256 * (NW / 4) <= DB_BASE, "flush_dw's LDS reduction would overwrite the bias sums");
  flush_dw<SH, NS, NL, NW>(d, dw, dbw, (2 * NC) * 64, smem, partials, gunscale, wave_s);
#undef CFW
#undef CBW
#undef CBIAS
}

template <class P, int NS, int NC>
__global__ __launch_bounds__(64 * SigmaWaves<NS>::value, 2) void k_mlp_bwd_sigma(NofMlpDesc d, const char* __restrict__ image,
                                                           const float2* __restrict__ feat, int L,
                                                           const typename P::elem* __restrict__ dsig, float2* __restrict__ dfeat,
                                                           float* __restrict__ partials, int64_t B,
                                                           const void* __restrict__ tile_list,
                                                           const typename P::elem* __restrict__ featq) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef Shp<NS, NC> SH;
  constexpr int NL = NS + NC;
  constexpr int KR = P::KR, NSTEP = 16 / KR;
  typedef typename P::frag frag;
  constexpr int FWN = SH::pair_base(NS - 1), NP = SH::pair_base(NS);               // the head's forward is not recomputed
  constexpr int BWB = FWN * PAIR_BYTES, BIASB = BWB + NP * PAIR_BYTES;
  constexpr int IN2_BASE = BIASB + SH::oblk_base(NS - 1) * 128;
  constexpr int NSLOT = 2 * NS - 1;                                                 // slot(0, 0) = 0, slot(l, q) = 2 l + q - 1
  constexpr int IN2_WAVE = NSLOT * NSTEP * 64 * (int)sizeof(frag);
  constexpr int NW = SigmaWaves<NS>::value;
  constexpr int DB_BASE = IN2_BASE + NW * IN2_WAVE;
  copy16(smem, image, (size_t)FWN * PAIR_BYTES);
  copy16(smem + BWB, image + (size_t)SH::pair_base(NL) * PAIR_BYTES, (size_t)NP * PAIR_BYTES);
  copy16(smem + BIASB, image + 2 * (size_t)SH::pair_base(NL) * PAIR_BYTES, (size_t)SH::oblk_base(NS - 1) * 128);
  __syncthreads();
#define SFW(l) (SH::pair_base(l) * PAIR_BYTES)
#define SBW(l) (BWB + SH::pair_base(l) * PAIR_BYTES)
#define SBIAS(l) (BIASB + SH::oblk_base(l) * 128)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hi = lane >> 5, j = lane & 31;
  typedef In2Store<P, NSLOT, true> Store;
  Store st;
  st.base = (frag*)(smem + IN2_BASE + wave * IN2_WAVE) + lane;
  float* dbw = (float*)(smem + DB_BASE) + wave * (2 * NS) * 64 + lane;              // [2 l + p][lane]
#pragma unroll
  for (int k = 0; k < 2 * NS; ++k) dbw[k * 64] = 0.0f;
  Ident<P> I;
  I.init(lane);
  // loss scaling of the 16-bit backward (the reference's GradScaler, nerf_runner.py:159,758): the loss gradient is multiplied
  // by a power of two where it enters and every fp32 output is divided by it where it leaves -- exact in fp32, and it keeps
  // the ~1e-7 gradients of a 1/(R*S)-normalised loss out of binary16's subnormal range inside the MFMA operands
  const float gscale = d.grad_scale > 0.0f ? d.grad_scale : 1.0f, gunscale = 1.0f / gscale;
  float dw[NL][2][2][16];
#pragma unroll
  for (int l = 0; l < NS; ++l)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (p < SH::pn(l) && q < SH::qn(l) && r < SH::nacc(l)) dw[l][p][q][r] = 0.0f;

  const int64_t ntiles = (B + 31) / 32;
  const int64_t tstride = (int64_t)gridDim.x * NW;
  // 64 % of this kernel's wave cycles used to be s_waitcnt on global memory (SQ_WAIT_ANY): the tile's feature loads queued
  // behind the previous tile's dfeat stores (vmcnt retires in order) and were waited for where they were issued, like the dsig
  // load.  Now the NEXT tile's features are requested a whole tile ahead and dsig at the top of the tile: 111 -> 82 us at cfg2.
  const TileWork work(tile_list, ntiles);
  const int wave_s = __builtin_amdgcn_readfirstlane(wave);
  const int64_t w0 = (int64_t)blockIdx.x * NW + wave_s;
  int64_t tile_n = work.at(w0);
  float xn[1][16];
  // features: the fp32 level-major array of nof_hash_encode_fwd, or (featq != NULL, uniform) the operand-precision copy of the fused forward
  if (featq != nullptr) load_featq_o1<P>(featq, B, tile_n * 32 + j, hi, xn);
  else load_feat_o1(feat, L, B, tile_n * 32 + j, hi, xn);
  typename P::frag dsn = load_sig_raw<P>(dsig, B, tile_n * 32 + j, hi);
  for (int64_t wi = w0; wi < work.n; wi += tstride) {
    asm volatile("" ::: "memory");
#if NOF_BWD_PRIO
    __builtin_amdgcn_s_setprio(NOF_BWD_PRIO == 1 ? 3 : 0);   // 1: the tile's loads (and the next tile's prefetch) ahead of the other wave's chain; 2: the chain ahead
#endif
    const int64_t tile = tile_n;
    tile_n = work.at(wi + tstride);
    const int64_t b = tile * 32 + j;
    uint32_t m1[NS];
    float h[2][16];
    const typename P::frag dsr = dsn;                   // dL/d(sigma out) of this tile, requested a tile ahead like the features
    dsn = load_sig_raw<P>(dsig, B, tile_n * 32 + j, hi);
    float x[1][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[0][r] = xn[0][r];
    pin16(x[0]);
    if (featq != nullptr) load_featq_o1<P>(featq, B, tile_n * 32 + j, hi, xn);
    else load_feat_o1(feat, L, B, tile_n * 32 + j, hi, xn);
    bool skip;
    {
      // all 32 x 16 gradients exactly zero (the colour kernel skipped the tile, see there): dfeat = 0, nothing else to do
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 bits = __builtin_bit_cast(u32x4, dsr);
      skip = tile_list == nullptr && __builtin_amdgcn_ballot_w64(((bits.x | bits.y | bits.z | bits.w) & 0x7FFF7FFFu) != 0u) == 0ull;
    }
    float df1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) df1[r] = 0.0f;
#if NOF_BWD_PRIO
    __builtin_amdgcn_s_setprio(NOF_BWD_PRIO == 1 ? 0 : 3);
#endif
    if (!skip) {
    {
      park_o2<P>(st, I, 0, x[0]);
      dense_o1<P, 1, 2>(smem, SFW(0), SBIAS(0), x, h, lane);
      m1[0] = relu_mask<2>(h);
    }
#pragma unroll
    for (int l = 1; l < NS - 1; ++l) {
      park_o2<P>(st, I, 2 * l - 1, h[0]);
      park_o2<P>(st, I, 2 * l, h[1]);
      float hn[2][16];
      dense_o1<P, 2, 2>(smem, SFW(l), SBIAS(l), h, hn, lane);
      m1[l] = relu_mask<2>(hn);
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[p][r] = hn[p][r];
    }
    park_o2<P>(st, I, 2 * (NS - 1) - 1, h[0]);
    park_o2<P>(st, I, 2 * (NS - 1), h[1]);

    float g1[2][16];
    sig_to_o1<P>(dsr, g1[0]);
#pragma unroll
    for (int r = 0; r < 16; ++r) g1[1][r] = 0.0f;
#pragma unroll
    for (int l = NS - 1; l >= 1; --l) {
      float d1[2][16];
      if (l == NS - 1) {
        dw_block<P, 2, 8>(dw[l][0], dbw + (2 * l) * 64, I, g1[0], st, 2 * l - 1);
        float ga[1][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ga[0][r] = g1[0][r];
        bwd_data<P, 1>(smem, SBW(l), 0, ga, d1[0], lane);
        bwd_data<P, 1>(smem, SBW(l), 1, ga, d1[1], lane);
      } else {
        dw_block<P, 2, 16>(dw[l][0], dbw + (2 * l) * 64, I, g1[0], st, 2 * l - 1);
        dw_block<P, 2, 16>(dw[l][1], dbw + (2 * l + 1) * 64, I, g1[1], st, 2 * l - 1);
        bwd_data<P, 2>(smem, SBW(l), 0, g1, d1[0], lane);
        bwd_data<P, 2>(smem, SBW(l), 1, g1, d1[1], lane);
      }
      apply_mask<2>(d1, m1[l - 1]);
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) g1[p][r] = d1[p][r];
    }
    {
      dw_block<P, 1, 16>(dw[0][0], dbw, I, g1[0], st, 0);
      dw_block<P, 1, 16>(dw[0][1], dbw + 64, I, g1[1], st, 0);
      bwd_data<P, 2>(smem, SBW(0), 0, g1, df1, lane);
    }
    }                                                                    // if (!skip)
    store_dfeat_o1(dfeat, L, B, b, hi, df1, gunscale);                   // zeros for a skipped tile
  }
  static_assert(flush_dw_accs<SH, 0, NS>() * 256 * (NW / 4) <= DB_BASE, "flush_dw's LDS reduction would overwrite the bias sums");
  flush_dw<SH, 0, NS, NW>(d, dw, dbw, (2 * NS) * 64, smem, partials, gunscale, wave_s);
#undef SFW
#undef SBW
#undef SBIAS
}

// =====================================================================================================
// Eikonal option (cfg eikonal_weight > 0): E = w * mean over {sdf < 1} of (|n| - 1)^2 with n = d sdf / d x (nerf_runner.py:
// 734-738; the normal as run_network_density defines it, :1342-1345 -- train_loop's own normal path is dead code).
// One fused pass per 32-sample tile, exact-fp32 MFMA, everything the second-order term needs in registers:
//   a. hash gathers of the lane's 8 levels: features and dy_dx (gridencoder.cu:160-245);
//   b. sigma net forward (ReLU masks);
//   c. its backward with a unit gradient on the sdf output: delta_l per layer and g = d sdf / d feature;
//   d. n = 0.5 * sum g . dy_dx, dE/dn, the loss; g and dE/dn are stored for the table / input gradients, which ride in the
//      hash backward (nof_hash_encode_bwd_eik);
//   e. the weight gradient: with q_0 = dE/dg and the tangents q_{l+1} = relu'(z_l) . (W_l q_l), dE/dW_l = delta_l (x) q_l
//      (forward-over-reverse); the sample contraction runs on the matrix core exactly like dW of the ordinary backward.
// `selected` / `n_sel`: s_b = sdf_b < 1 is re-derived from this kernel's own sdf; n_sel[0] = their number (device scalar,
// computed by the caller from the forward's raw output) normalises the mean.
// =====================================================================================================
template <int NS, int NC>
__global__ __launch_bounds__(256) void k_eikonal(NofMlpDesc d, const char* __restrict__ image, NofHashGrid g,
                                                  const float2* __restrict__ table, const float* __restrict__ pts_w,
                                                  const uint8_t* __restrict__ valid, const float* __restrict__ n_sel,
                                                  float weight, float gscale, float2* __restrict__ geik, float* __restrict__ dedn,
                                                  float* __restrict__ partials, float* __restrict__ loss_out, int64_t B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef PrecF32 P;
  typedef Shp<NS, NC> SH;
  constexpr int NL = NS + NC;
  constexpr int NPS = SH::pair_base(NS);                               // the sigma layers' fragment pairs
  constexpr int BW_BASE = NPS * PAIR_BYTES;
  constexpr int BIAS_BASE = 2 * NPS * PAIR_BYTES;
  constexpr int ZB_BASE = BIAS_BASE + SH::oblk_base(NS) * 128;         // a zero "bias" for the tangent pass
  constexpr int DB_BASE = ZB_BASE + 2 * 128;
  copy16(smem, image, (size_t)NPS * PAIR_BYTES);
  copy16(smem + BW_BASE, image + (size_t)SH::pair_base(NL) * PAIR_BYTES, (size_t)NPS * PAIR_BYTES);
  copy16(smem + BIAS_BASE, image + 2 * (size_t)SH::pair_base(NL) * PAIR_BYTES, (size_t)SH::oblk_base(NS) * 128);
  for (int e = threadIdx.x; e < 64; e += blockDim.x) ((float*)(smem + ZB_BASE))[e] = 0.0f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hi = lane >> 5, j = lane & 31;
  float* dbz = (float*)(smem + DB_BASE) + wave * (2 * NS + 1) * 64 + lane;      // [2 l + p][lane] zeros (no bias gradient) + 1 dummy
#pragma unroll
  for (int k = 0; k < 2 * NS + 1; ++k) dbz[k * 64] = 0.0f;
  float* dummy = dbz + 2 * NS * 64;
  __syncthreads();
  Ident<P> I;
  I.init(lane);
  typedef In2Store<P, 2, false> Store;
  float dw[NL][2][2][16];
#pragma unroll
  for (int l = 0; l < NS; ++l)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (p < SH::pn(l) && q < SH::qn(l) && r < SH::nacc(l)) dw[l][p][q][r] = 0.0f;
  float loss_acc = 0.0f;
  const float nsel = n_sel[0];
  const float ke = nsel > 0.0f ? gscale * weight / nsel : 0.0f;      // gradients carry the data-parallel 1/world, the loss value does not
  const int64_t ntiles = (B + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    asm volatile("" ::: "memory");
    const int64_t b = tile * 32 + j;
    const bool ok = b < B;
    // ---- a. features and dy_dx of this lane's 8 levels ----
    float x[1][16], dy[8][3][2];
    {
      float pt[3] = {0.f, 0.f, 0.f};
      const bool in = ok && valid[b] != 0;                             // outside [-1,1]^3: zero features, zero normal (nerf_runner.py:1245-1266)
      if (ok) { pt[0] = pts_w[b * 3]; pt[1] = pts_w[b * 3 + 1]; pt[2] = pts_w[b * 3 + 2]; }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int level = 8 * hi + k;
        float2 f = make_float2(0.f, 0.f);
#pragma unroll
        for (int dd = 0; dd < 3; ++dd) { dy[k][dd][0] = 0.f; dy[k][dd][1] = 0.f; }
        if (in && level < g.L) {
          const HashLevel lv = load_level(g, level);
          const CellPos c = locate3(pt, lv.scale);
          if (!c.oob) {
            const float2* __restrict__ tl = table + lv.offset;
            float2 v[8];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk)
              v[kk] = tl[grid_index(lv, c.g[0] + (kk & 1), c.g[1] + ((kk >> 1) & 1), c.g[2] + ((kk >> 2) & 1))];
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
              float w = 1.0f;
#pragma unroll
              for (int dd = 0; dd < 3; ++dd) w *= (kk & (1 << dd)) ? c.f[dd] : 1.0f - c.f[dd];
              f.x += w * v[kk].x; f.y += w * v[kk].y;
            }
#pragma unroll
            for (int gd = 0; gd < 3; ++gd)
#pragma unroll
              for (int kk = 0; kk < 8; ++kk) {
                if (kk & (1 << gd)) continue;
                float w = lv.scale;
#pragma unroll
                for (int dd = 0; dd < 3; ++dd)
                  if (dd != gd) w *= (kk & (1 << dd)) ? c.f[dd] : 1.0f - c.f[dd];
                dy[k][gd][0] += w * (v[kk | (1 << gd)].x - v[kk].x);
                dy[k][gd][1] += w * (v[kk | (1 << gd)].y - v[kk].y);
              }
          }
        }
        x[0][2 * k] = f.x; x[0][2 * k + 1] = f.y;
      }
    }
    // ---- b. forward ----
    uint32_t m1[NS];
    float h[2][16], so[1][16];
    dense_o1<P, 1, 2>(smem, FW_OFF(0), BIAS_OFF(0), x, h, lane);
    m1[0] = relu_mask<2>(h);
#pragma unroll
    for (int l = 1; l < NS - 1; ++l) {
      float hn[2][16];
      dense_o1<P, 2, 2>(smem, FW_OFF(l), BIAS_OFF(l), h, hn, lane);
      m1[l] = relu_mask<2>(hn);
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[p][r] = hn[p][r];
    }
    dense_o1<P, 2, 1>(smem, FW_OFF(NS - 1), BIAS_OFF(NS - 1), h, so, lane);
    // ---- c. backward with d sdf = 1: delta_l (gradient at the pre-activation of layer l) and g ----
    float delta[NS][2][16];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) delta[NS - 1][p][r] = 0.0f;
    if (hi == 0 && ok) delta[NS - 1][0][0] = 1.0f;
#pragma unroll
    for (int l = NS - 1; l >= 1; --l) {
      if (l == NS - 1) {
        float ga[1][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) ga[0][r] = delta[l][0][r];
        bwd_data<P, 1>(smem, BW_OFF(l), 0, ga, delta[l - 1][0], lane);
        bwd_data<P, 1>(smem, BW_OFF(l), 1, ga, delta[l - 1][1], lane);
      } else {
        bwd_data<P, 2>(smem, BW_OFF(l), 0, delta[l], delta[l - 1][0], lane);
        bwd_data<P, 2>(smem, BW_OFF(l), 1, delta[l], delta[l - 1][1], lane);
      }
      apply_mask<2>(delta[l - 1], m1[l - 1]);
    }
    float gf[16];
    bwd_data<P, 2>(smem, BW_OFF(0), 0, delta[0], gf, lane);
    // ---- d. normal, loss, dE/dn, q_0 ----
    float n[3];
#pragma unroll
    for (int dd = 0; dd < 3; ++dd) {
      float a = 0.0f;
#pragma unroll
      for (int k = 0; k < 8; ++k) a += gf[2 * k] * dy[k][dd][0] + gf[2 * k + 1] * dy[k][dd][1];
      a += __shfl_xor(a, 32, 64);
      n[dd] = 0.5f * a;                                                 // d x01 / d x = 1/2 (grid.py:160)
    }
    const float sdf = __shfl(so[0][0], j, 64);                          // lane (j, hi = 0) holds the sdf of sample j
    const float nrm = sqrtf((n[0] * n[0] + n[1] * n[1]) + n[2] * n[2]);
    const bool sel = ok && sdf < 1.0f;
    float dn[3];
#pragma unroll
    for (int dd = 0; dd < 3; ++dd) dn[dd] = (sel && nrm > 0.0f) ? ke * 2.0f * (nrm - 1.0f) * n[dd] / nrm : 0.0f;
    if (sel && hi == 0) loss_acc += (nrm - 1.0f) * (nrm - 1.0f);
    if (ok) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int level = 8 * hi + k;
        if (level < g.L) geik[(int64_t)level * B + b] = make_float2(gf[2 * k], gf[2 * k + 1]);
      }
      if (hi == 0) { dedn[b * 3] = dn[0]; dedn[b * 3 + 1] = dn[1]; dedn[b * 3 + 2] = dn[2]; }
    }
    float q0[1][16];
#pragma unroll
    for (int k = 0; k < 8; ++k)
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2)
        q0[0][2 * k + c2] = 0.5f * ((dn[0] * dy[k][0][c2] + dn[1] * dy[k][1][c2]) + dn[2] * dy[k][2][c2]);
    // ---- e. tangent pass: dW_l += delta_l (x) q_l, q_{l+1} = relu'(z_l) . (W_l q_l) ----
    Store st;
    {
      park_o2<P>(st, I, 0, q0[0]);
      dw_block<P, 1, 16>(dw[0][0], dummy, I, delta[0][0], st, 0);
      dw_block<P, 1, 16>(dw[0][1], dummy, I, delta[0][1], st, 0);
    }
    float q[2][16];
    dense_o1<P, 1, 2>(smem, FW_OFF(0), ZB_BASE, q0, q, lane);
    apply_mask<2>(q, m1[0]);
#pragma unroll
    for (int l = 1; l < NS; ++l) {
      park_o2<P>(st, I, 0, q[0]);
      park_o2<P>(st, I, 1, q[1]);
      if (l == NS - 1) {
        dw_block<P, 2, 8>(dw[l][0], dummy, I, delta[l][0], st, 0);
      } else {
        dw_block<P, 2, 16>(dw[l][0], dummy, I, delta[l][0], st, 0);
        dw_block<P, 2, 16>(dw[l][1], dummy, I, delta[l][1], st, 0);
        float qn[2][16];
        dense_o1<P, 2, 2>(smem, FW_OFF(l), ZB_BASE, q, qn, lane);
        apply_mask<2>(qn, m1[l]);
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
          for (int r = 0; r < 16; ++r) q[p][r] = qn[p][r];
      }
    }
  }
  static_assert(flush_dw_accs<SH, 0, NS>() * 256 <= DB_BASE, "flush_dw's LDS reduction would overwrite the bias sums");
  flush_dw<SH, 0, NS>(d, dw, dbz, (2 * NS + 1) * 64, smem, partials, 1.0f, __builtin_amdgcn_readfirstlane(wave));
  // loss: one atomic pair per wave
  loss_acc += __shfl_xor(loss_acc, 1, 64);  loss_acc += __shfl_xor(loss_acc, 2, 64);  loss_acc += __shfl_xor(loss_acc, 4, 64);
  loss_acc += __shfl_xor(loss_acc, 8, 64);  loss_acc += __shfl_xor(loss_acc, 16, 64); loss_acc += __shfl_xor(loss_acc, 32, 64);
  if (lane == 0 && loss_acc != 0.0f && nsel > 0.0f) {
    const float e = weight * loss_acc / nsel;
    atomicAdd(&loss_out[0], e);
    atomicAdd(&loss_out[7], e);
  }
}

// =====================================================================================================
// host side
// =====================================================================================================
static int check_desc(const NofMlpDesc* d) {
  if (!d) return nof_set_error(-1, "mlp descriptor is NULL");
  if (d->hidden != 64 && d->hidden != 128) return nof_set_error(-1, "mlp: hidden width must be 64 or 128 (got %d)", d->hidden);
  if (d->n_sigma < 2 || d->n_sigma > 4 || d->n_color < 2 || d->n_color > 4)
    return nof_set_error(-1, "mlp: supported depths are num_layers in {2,3,4}, num_layers_color in {2,3,4} (got %d,%d)",
                         d->n_sigma, d->n_color);
  if (d->in_feat < 1 || d->in_feat > 32) return nof_set_error(-1, "mlp: L*C must be <= 32 (got %d)", d->in_feat);
  if (d->n_view < 0 || d->n_view > NOF_VIEW_COLS) return nof_set_error(-1, "mlp: n_view must be <= 16 (got %d)", d->n_view);
  if (d->geo != 15) return nof_set_error(-1, "mlp: geo_feat_dim must be 15 (got %d)", d->geo);
  if (d->precision < 0 || d->precision > 4)
    return nof_set_error(-1, "mlp: precision must be 0 (fp32), 1 (bf16), 2 (fp16), 3 (fp16, split forward) or 4 (bf16, split forward)");
  const int nl = d->n_sigma + d->n_color;
  for (int l = 0; l < nl; ++l) {
    const int exp_in = l == 0 ? d->in_feat : (l == d->n_sigma ? d->n_view + d->geo : d->hidden);
    const int exp_out = l == d->n_sigma - 1 ? 1 + d->geo : (l == nl - 1 ? 3 : d->hidden);
    if (d->in_dim[l] != exp_in || d->out_dim[l] != exp_out)
      return nof_set_error(-1, "mlp: layer %d is %dx%d, expected %dx%d", l, d->out_dim[l], d->in_dim[l], exp_out, exp_in);
  }
  return 0;
}

// Networks the register-resident kernels above are instantiated for: hidden 64, depths {2,3}.  Everything else (hidden 128,
// depth 4: BASELINE cfg5's 4x128 + 4x128) runs through the per-network kernels of nof_mlp_wide.h (nof_mlp_wide_* entry points).
static bool is_wide(const NofMlpDesc* d) { return d->hidden != 64 || d->n_sigma > 3 || d->n_color > 3; }
static int check_narrow(const NofMlpDesc* d) {
  if (int e = check_desc(d)) return e;
  if (is_wide(d))
    return nof_set_error(-1, "mlp: hidden %d / depths (%d,%d) run through nof_mlp_wide_fwd / nof_mlp_wide_bwd / nof_mlp_wide_sdf",
                         d->hidden, d->n_sigma, d->n_color);
  return 0;
}
static size_t elem_size(int precision) { return precision == 0 ? 4 : 2; }
static bool is_split(int precision) { return precision >= 3; }            // 3-term operand split in the forward kernels
static bool is_bf16(int precision) { return precision == 1 || precision == 4; }
static int n_pairs(const NofMlpDesc& d, int nl) { return pair_base(d, nl); }
static int n_oblk(const NofMlpDesc& d, int nl) { return oblk_base(d, nl); }

template <class K>
static int set_smem(K kernel, size_t bytes) {
  if (bytes > 160 * 1024)                                              // gfx950: 160 KB LDS per CU; refuse before HIP sees it
    return nof_set_error(-1, "mlp: this shape/precision needs %zu bytes of LDS per workgroup (limit 163840)", bytes);
  if (bytes > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
      (void)hipGetLastError();                                         // do not leave a sticky error for the caller's next HIP call
      return nof_set_error((int)e, "hipFuncSetAttribute(%zu B LDS): %s", bytes, hipGetErrorString(e));
    }
  }
  return 0;
}

#define DISPATCH_SHAPE(P, FN, ...)                                                                        \
  if (d->n_sigma == 2 && d->n_color == 3) { FN(P, 2, 3, __VA_ARGS__) }                                    \
  else if (d->n_sigma == 3 && d->n_color == 2) { FN(P, 3, 2, __VA_ARGS__) }                               \
  else if (d->n_sigma == 2 && d->n_color == 2) { FN(P, 2, 2, __VA_ARGS__) }                               \
  else { FN(P, 3, 3, __VA_ARGS__) }
#define DISPATCH_PREC(FN, ...)                                                                            \
  if (d->precision == 0) { DISPATCH_SHAPE(PrecF32, FN, __VA_ARGS__) }                                     \
  else if (is_bf16(d->precision)) { DISPATCH_SHAPE(PrecBF16, FN, __VA_ARGS__) }                           \
  else { DISPATCH_SHAPE(PrecF16, FN, __VA_ARGS__) }
// forward-type kernels: the 16-bit types also exist with the 3-term operand split
#define DISPATCH_PREC_FWD(FN)                                                                             \
  if (d->precision == 0) { DISPATCH_SHAPE(PrecF32, FN, false) }                                           \
  else if (d->precision == 1) { DISPATCH_SHAPE(PrecBF16, FN, false) }                                     \
  else if (d->precision == 2) { DISPATCH_SHAPE(PrecF16, FN, false) }                                      \
  else if (d->precision == 3) { DISPATCH_SHAPE(PrecF16, FN, true) }                                       \
  else { DISPATCH_SHAPE(PrecBF16, FN, true) }

extern "C" int64_t nof_mlp_packed_bytes(const NofMlpDesc* d) {
  if (check_desc(d)) return -1;
  const int nl = d->n_sigma + d->n_color;
  return (is_split(d->precision) ? 3 : 2) * (int64_t)n_pairs(*d, nl) * 16 * 64 * (int64_t)elem_size(d->precision) +
         (int64_t)n_oblk(*d, nl) * 32 * 4;
}

static int mlp_pack_launch(const NofMlpDesc* d, const float* mlp_params, void* packed, const float* pose, const float* c2w,
                           float max_trans, float max_rot, float* tf, int F, void* stream) {
  if (int e = check_desc(d)) return e;
  NOF_ARG(mlp_params && packed);
  const int lo = is_split(d->precision) ? 1 : 0;
  const int pack_blocks = (int)nof_div_up((int64_t)n_pairs(*d, d->n_sigma + d->n_color) * 1024, 256);   // one element per thread
  const dim3 grid((unsigned)(pack_blocks + (F > 0 ? 1 : 0)));
  if (d->precision == 0) hipLaunchKernelGGL(k_mlp_pack<PrecF32>, grid, dim3(256), 0, (hipStream_t)stream, *d, mlp_params, (char*)packed, 0, pack_blocks, pose, c2w, max_trans, max_rot, tf, F);
  else if (is_bf16(d->precision)) hipLaunchKernelGGL(k_mlp_pack<PrecBF16>, grid, dim3(256), 0, (hipStream_t)stream, *d, mlp_params, (char*)packed, lo, pack_blocks, pose, c2w, max_trans, max_rot, tf, F);
  else hipLaunchKernelGGL(k_mlp_pack<PrecF16>, grid, dim3(256), 0, (hipStream_t)stream, *d, mlp_params, (char*)packed, lo, pack_blocks, pose, c2w, max_trans, max_rot, tf, F);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_mlp_pack(const NofMlpDesc* d, const float* mlp_params, void* packed, void* stream) {
  return mlp_pack_launch(d, mlp_params, packed, nullptr, nullptr, 0.0f, 0.0f, nullptr, 0, stream);
}

// nof_mlp_pack and nof_pose_fwd (same arguments, same results) as one launch: what a training step calls before its ray marcher
extern "C" int nof_mlp_pack_pose(const NofMlpDesc* d, const float* mlp_params, void* packed, const float* pose_data, const float* c2w,
                                 float max_trans, float max_rot_rad, float* tf, int32_t F, void* stream) {
  NOF_ARG(c2w && tf && F >= 0);
  return mlp_pack_launch(d, mlp_params, packed, pose_data, c2w, max_trans, max_rot_rad, tf, (int)F, stream);
}

static int g_bwd_blocks = 0;
extern "C" int nof_mlp_bwd_blocks(void) {
  if (g_bwd_blocks == 0) {
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
      hipDeviceProp_t prop;
      if (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) cus = prop.multiProcessorCount;
    }
    (void)hipGetLastError();
    g_bwd_blocks = 2 * cus;                                           // one partial row per workgroup: 2 persistent workgroups per CU
  }
  return g_bwd_blocks;
}

// workgroups of the forward kernel: one tile per wave up to 4 workgroups of 4 waves / 2 of 8 per CU (what LDS or registers admit)
template <class P> static unsigned fwd_blocks(int64_t ntiles) {
  constexpr int NW = FwdWaves<P>::value;
  const int64_t cap = NW == 8 ? 512 : 1024;
  return (unsigned)(nof_div_up(ntiles, NW) < cap ? nof_div_up(ntiles, NW) : cap);
}

extern "C" int nof_mlp_fwd(const NofMlpDesc* d, const void* packed, const float* feat, int32_t L, const float* view,
                            int32_t S, float* raw, void* sigma_out, int64_t B, void* stream) {
  if (int e = check_narrow(d)) return e;
  NOF_ARG(packed && feat && view && raw && B >= 0 && S >= 1 && L >= 1 && L * 2 == d->in_feat);
  NOF_ARG((int64_t)L * B * 8 < (1ll << 32));                   // level-major arrays are addressed with 32-bit lane offsets
  if (B == 0) return 0;
  const int nl = d->n_sigma + d->n_color;
  const size_t shm = (is_split(d->precision) ? 2 : 1) * (size_t)n_pairs(*d, nl) * 16 * 64 * elem_size(d->precision) +
                     (size_t)n_oblk(*d, nl) * 32 * 4;
  const int64_t ntiles = (B + 31) / 32;
#define LAUNCH_FWD(P, NS_, NC_, SPLIT_)                                                                   \
  {                                                                                                       \
    auto kern = k_mlp_fwd<P, NS_, NC_, false, SPLIT_>;                                                    \
    if (int e = set_smem(kern, shm)) return e;                                                            \
    hipLaunchKernelGGL(kern, dim3(fwd_blocks<P>(ntiles)), dim3(64 * FwdWaves<P>::value), shm, (hipStream_t)stream, *d, \
                       (const char*)packed, (const float2*)feat, (int)L, view, (int)S, raw, (typename P::elem*)sigma_out, B); \
  }
  DISPATCH_PREC_FWD(LAUNCH_FWD)
#undef LAUNCH_FWD
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_mlp_sdf(const NofMlpDesc* d, const void* packed, const float* feat, int32_t L, float* sdf,
                            int64_t B, void* stream) {
  if (int e = check_narrow(d)) return e;
  NOF_ARG(packed && feat && sdf && B >= 0 && L >= 1 && L * 2 == d->in_feat);
  NOF_ARG((int64_t)L * B * 8 < (1ll << 32));                   // level-major arrays are addressed with 32-bit lane offsets
  if (B == 0) return 0;
  const int nl = d->n_sigma;
  const size_t shm = (is_split(d->precision) ? 2 : 1) * (size_t)n_pairs(*d, nl) * 16 * 64 * elem_size(d->precision) +
                     (size_t)n_oblk(*d, nl) * 32 * 4;
  const int64_t ntiles = (B + 31) / 32;
#define LAUNCH_SDF(P, NS_, NC_, SPLIT_)                                                                   \
  {                                                                                                       \
    auto kern = k_mlp_fwd<P, NS_, NC_, true, SPLIT_>;                                                     \
    if (int e = set_smem(kern, shm)) return e;                                                            \
    hipLaunchKernelGGL(kern, dim3(fwd_blocks<P>(ntiles)), dim3(64 * FwdWaves<P>::value), shm, (hipStream_t)stream, *d, \
                       (const char*)packed, (const float2*)feat, (int)L, (const float*)nullptr, 1, sdf, (typename P::elem*)nullptr, B); \
  }
  DISPATCH_PREC_FWD(LAUNCH_SDF)
#undef LAUNCH_SDF
  NOF_LAUNCH_OK();
  return 0;
}

// Hash encode + both MLPs in ONE launch (16-bit operand types): pts_w [B,3] -> raw [B,4]; the [B,32] embedding stays on chip.
// sigma_out: as in nof_mlp_fwd.  featq (may be NULL): [B][2][16] operand-type elements, the features as the sigma backward wants
// them (nof_mlp_bwd_featq).
extern "C" int nof_encode_mlp_fwd(const NofHashGrid* g, const NofMlpDesc* d, const void* packed, const float* table,
                                   const float* pts_w, const float* view, int32_t S, float* raw, void* sigma_out, void* featq,
                                   int64_t B, void* stream) {
  if (int e = check_narrow(d)) return e;
  NOF_ARG(g && g->C == 2 && g->L >= 1 && g->L <= NOF_MAX_LEVELS && g->L * 2 == d->in_feat);
  NOF_ARG(packed && table && pts_w && view && raw && B >= 0 && S >= 1);
  if (d->precision == 0) return nof_set_error(-1, "nof_encode_mlp_fwd: 16-bit operand types only (fp32: nof_hash_encode_fwd + nof_mlp_fwd)");
  if (B == 0) return 0;
  const int nl = d->n_sigma + d->n_color;
  const size_t img = (is_split(d->precision) ? 2 : 1) * (size_t)n_pairs(*d, nl) * 16 * 64 * elem_size(d->precision) +
                     (size_t)n_oblk(*d, nl) * 32 * 4;
  int waves = (int)((160 * 1024 - img - 512) / 8192);                   // one workgroup per CU: the image + 8 KB of stage per wave + the level table
  if (waves > NOF_ENC_WAVES) waves = NOF_ENC_WAVES;
  NOF_ARG(waves >= 4);
  const size_t shm = img + (size_t)waves * 8192 + 512;
  const int64_t npairs = (B + 63) / 64;
  const int64_t want = nof_div_up(npairs, waves);
  const int64_t cap = (int64_t)nof_mlp_bwd_blocks() / 2;               // one resident workgroup per CU
  const unsigned blocks = (unsigned)(want < cap ? want : cap);
#define LAUNCH_ENC(P, NS_, NC_, SPLIT_)                                                                   \
  {                                                                                                       \
    auto kern = k_enc_mlp_fwd<P, NS_, NC_, SPLIT_>;                                                       \
    if (int e = set_smem(kern, shm)) return e;                                                            \
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * waves), shm, (hipStream_t)stream, *d, (const char*)packed, *g, \
                       (const float2*)table, pts_w, view, (int)S, raw, (typename P::elem*)sigma_out,      \
                       (typename P::elem*)featq, B);                                                      \
  }
  if (d->precision == 1) { DISPATCH_SHAPE(PrecBF16, LAUNCH_ENC, false) }
  else if (d->precision == 2) { DISPATCH_SHAPE(PrecF16, LAUNCH_ENC, false) }
  else if (d->precision == 3) { DISPATCH_SHAPE(PrecF16, LAUNCH_ENC, true) }
  else { DISPATCH_SHAPE(PrecBF16, LAUNCH_ENC, true) }
#undef LAUNCH_ENC
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_mlp_bwd(const NofMlpDesc* d, const void* packed, const float* feat, int32_t L, const float* view,
                            int32_t S, const float* draw, const void* sigma_out, void* dsigma_ws, float* dfeat, float* dview,
                            float* partials, int64_t B, void* stream) {
  return nof_mlp_bwd_tiles(d, packed, feat, L, view, S, draw, sigma_out, dsigma_ws, dfeat, dview, partials, nullptr, B, stream);
}

// The same over a work list (NofTileList): only the listed 32-sample tiles are computed, dealt evenly to the persistent waves.
// dfeat (and the dsigma workspace) of unlisted tiles is NOT written -- the consumers of the same step take the same list.
static int mlp_bwd_tiles(const NofMlpDesc* d, const void* packed, const float* feat, const void* featq, int32_t L, const float* view,
                         int32_t S, const float* draw, const void* sigma_out, void* dsigma_ws, float* dfeat, float* dview,
                         float* partials, const void* tile_list, int64_t B, void* stream);
extern "C" int nof_mlp_bwd_tiles(const NofMlpDesc* d, const void* packed, const float* feat, int32_t L, const float* view,
                                  int32_t S, const float* draw, const void* sigma_out, void* dsigma_ws, float* dfeat, float* dview,
                                  float* partials, const void* tile_list, int64_t B, void* stream) {
  NOF_ARG(feat);
  return mlp_bwd_tiles(d, packed, feat, nullptr, L, view, S, draw, sigma_out, dsigma_ws, dfeat, dview, partials, tile_list, B, stream);
}
// The same with the features taken from `featq`, the operand-precision copy nof_encode_mlp_fwd leaves ([B][2][16] elements): the
// backward of the fused forward.  16-bit operand types, split workspace required.
extern "C" int nof_mlp_bwd_featq(const NofMlpDesc* d, const void* packed, const void* featq, int32_t L, const float* view,
                                  int32_t S, const float* draw, const void* sigma_out, void* dsigma_ws, float* dfeat, float* dview,
                                  float* partials, const void* tile_list, int64_t B, void* stream) {
  NOF_ARG(featq && d && d->precision != 0 && sigma_out && dsigma_ws);
  return mlp_bwd_tiles(d, packed, nullptr, featq, L, view, S, draw, sigma_out, dsigma_ws, dfeat, dview, partials, tile_list, B, stream);
}
static int mlp_bwd_tiles(const NofMlpDesc* d, const void* packed, const float* feat, const void* featq, int32_t L, const float* view,
                         int32_t S, const float* draw, const void* sigma_out, void* dsigma_ws, float* dfeat, float* dview,
                         float* partials, const void* tile_list, int64_t B, void* stream) {
  if (int e = check_narrow(d)) return e;
  NOF_ARG(packed && (feat || featq) && view && draw && dfeat && dview && partials && B >= 0 && S >= 32 && L * 2 == d->in_feat);
  NOF_ARG((int64_t)L * B * 8 < (1ll << 32));                   // level-major arrays are addressed with 32-bit lane offsets
  const int nl = d->n_sigma + d->n_color, ns = d->n_sigma;
  const size_t es = elem_size(d->precision), pair_bytes = 16 * 64 * es;
  const unsigned rows = (unsigned)nof_mlp_bwd_blocks();
  if (d->precision != 0 && sigma_out != nullptr && dsigma_ws != nullptr) {
    // split path: colour net, then sigma net; 2 workgroups per CU each, one partial row per wave
    // `rows` partial rows = one per group of four waves (flush_dw): rows * 4 / waves-per-workgroup workgroups
    const size_t wc = d->n_color >= 3 ? 8 : NOF_BWD_WAVES_C2, ws = NOF_BWD_WAVES_S;
    const size_t shm_c = 2 * (size_t)(n_pairs(*d, nl) - n_pairs(*d, ns)) * pair_bytes + (size_t)(n_oblk(*d, nl) - n_oblk(*d, ns)) * 128 +
                         wc * (2 * d->n_color) * 2 * 64 * 16 + wc * (2 * d->n_color) * 64 * 4 + 2048;   // (+ identity fragments)
    const size_t shm_s = (size_t)(n_pairs(*d, ns - 1) + n_pairs(*d, ns)) * pair_bytes + (size_t)n_oblk(*d, ns - 1) * 128 +
                         ws * (2 * ns - 1) * 2 * 64 * 16 + ws * (2 * ns) * 64 * 4;
    const unsigned blocks_c = rows * 4 / (unsigned)wc, blocks_s = rows * 4 / (unsigned)ws;
#define LAUNCH_SPLIT(P, NS_, NC_, dummy)                                                                  \
  {                                                                                                       \
    auto kc = k_mlp_bwd_color<P, NS_, NC_>;                                                               \
    auto ks = k_mlp_bwd_sigma<P, NS_, NC_>;                                                               \
    if (int e = set_smem(kc, shm_c)) return e;                                                            \
    if (int e = set_smem(ks, shm_s)) return e;                                                            \
    hipLaunchKernelGGL(kc, dim3(blocks_c), dim3(64 * (unsigned)wc), shm_c, (hipStream_t)stream, *d, (const char*)packed,  \
                       (const typename P::elem*)sigma_out, view, (int)S, (const float4*)draw,             \
                       (typename P::elem*)dsigma_ws, dview, partials, B, tile_list);                      \
    hipLaunchKernelGGL(ks, dim3(blocks_s), dim3(64 * (unsigned)ws), shm_s, (hipStream_t)stream, *d, (const char*)packed,  \
                       (const float2*)feat, (int)L, (const typename P::elem*)dsigma_ws, (float2*)dfeat,   \
                       partials, B, tile_list, (const typename P::elem*)featq);                           \
  }
    if (is_bf16(d->precision)) { DISPATCH_SHAPE(PrecBF16, LAUNCH_SPLIT, 0) }
    else { DISPATCH_SHAPE(PrecF16, LAUNCH_SPLIT, 0) }
#undef LAUNCH_SPLIT
    NOF_LAUNCH_OK();
    return 0;
  }
  size_t shm = 2 * (size_t)n_pairs(*d, nl) * pair_bytes + (size_t)n_oblk(*d, nl) * 32 * 4;
  if (d->precision != 0) shm += (size_t)4 * (2 * nl) * 2 * 64 * 16;      // lane-private orientation-2 slots (16-bit modes)
  shm += (size_t)4 * (2 * nl) * 64 * 4;                                   // lane-private bias-gradient sums
  const unsigned blocks = rows;
#define LAUNCH_BWD(P, NS_, NC_, dummy)                                                                    \
  {                                                                                                       \
    auto kern = k_mlp_bwd<P, NS_, NC_>;                                                                   \
    if (int e = set_smem(kern, shm)) return e;                                                            \
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), shm, (hipStream_t)stream, *d, (const char*)packed,  \
                       (const float2*)feat, (int)L, view, (int)S, (const float4*)draw, (float2*)dfeat,    \
                       dview, partials, B, tile_list);                                                    \
  }
  DISPATCH_PREC(LAUNCH_BWD, 0)
#undef LAUNCH_BWD
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_sdf_grid_query(const NofHashGrid* g, const NofMlpDesc* d, const void* packed, const float* table,
                                   const uint32_t* occ_bits, int32_t level, const float* tx, const float* ty, const float* tz,
                                   int32_t nx, int32_t ny, int32_t nz, float outside_value, float* sdf, void* stream) {
  if (int e = check_narrow(d)) return e;
  NOF_ARG(g && g->C == 2 && g->L >= 1 && g->L <= NOF_MAX_LEVELS && g->L * 2 == d->in_feat);
  NOF_ARG(packed && table && tx && ty && tz && sdf && nx >= 0 && ny >= 0 && nz >= 0 && level >= 0 && level <= 8);
  if (nx == 0 || ny == 0 || nz == 0) return 0;
  const int nl = d->n_sigma;
  const size_t shm = (is_split(d->precision) ? 2 : 1) * (size_t)n_pairs(*d, nl) * 16 * 64 * elem_size(d->precision) +
                     (size_t)n_oblk(*d, nl) * 32 * 4;
  const int64_t ntiles = (int64_t)nx * ny * ((nz + 31) / 32);
  const unsigned blocks = (unsigned)(nof_div_up(ntiles, 4) < 2048 ? nof_div_up(ntiles, 4) : 2048);
#define LAUNCH_GRID(P, NS_, NC_, SPLIT_)                                                                  \
  {                                                                                                       \
    auto kern = k_sdf_grid<P, NS_, NC_, SPLIT_>;                                                          \
    if (int e = set_smem(kern, shm)) return e;                                                            \
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), shm, (hipStream_t)stream, *d, (const char*)packed, *g, \
                       (const float2*)table, occ_bits, 1 << level, tx, ty, tz, (int)nx, (int)ny, (int)nz, \
                       outside_value, sdf);                                                               \
  }
  DISPATCH_PREC_FWD(LAUNCH_GRID)
#undef LAUNCH_GRID
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int64_t nof_mlp_bwd_workspace_bytes(const NofMlpDesc* d) {
  if (check_desc(d)) return -1;
  return (int64_t)nof_mlp_bwd_blocks() * d->n_params * 4;
}

/* Eikonal term (cfg eikonal_weight > 0; nerf_runner.py:734-738 with the normal of run_network_density, :1342-1345).
 * `desc32` / `packed32`: the SAME network packed with precision 0 (the term is evaluated with the exact-fp32 MFMA whatever the
 * training precision).  pts_w [B,3], valid [B] u8, n_sel: device scalar = number of samples with sdf < 1 (from the forward's raw),
 * weight = eikonal_weight, grad_scale = 1/world_size (gradients only).  Writes geik [L,B,2] and dedn [B,3] (consumed by nof_hash_encode_bwd_eik), the sigma
 * layers' weight gradient as per-wave rows of partials_e [nof_mlp_bwd_blocks(), n_params] (other entries untouched: zero them
 * once), and ADDS the term to loss_out[0] and loss_out[7]. */
extern "C" int nof_eikonal(const NofMlpDesc* d, const void* packed32, const NofHashGrid* g, const float* table, const float* pts_w,
                            const uint8_t* valid, const float* n_sel, float weight, float grad_scale, float* geik, float* dedn,
                            float* partials_e, float* loss_out, int64_t B, void* stream) {
  if (int e = check_narrow(d)) return e;
  NOF_ARG(d->precision == 0 && g && g->C == 2 && g->L * 2 == d->in_feat);
  NOF_ARG(packed32 && table && pts_w && valid && n_sel && geik && dedn && partials_e && loss_out && B >= 0);
  if (B == 0) return 0;
  const int ns = d->n_sigma;
  const size_t shm = 2 * (size_t)n_pairs(*d, ns) * 16 * 64 * 4 + (size_t)n_oblk(*d, ns) * 128 + 2 * 128 + (size_t)4 * (2 * ns + 1) * 64 * 4;
  const unsigned blocks = (unsigned)nof_mlp_bwd_blocks();
#define LAUNCH_EIK(P_, NS_, NC_, dummy)                                                                   \
  {                                                                                                       \
    auto kern = k_eikonal<NS_, NC_>;                                                                      \
    if (int e = set_smem(kern, shm)) return e;                                                            \
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), shm, (hipStream_t)stream, *d, (const char*)packed32, *g, \
                       (const float2*)table, pts_w, valid, n_sel, weight, grad_scale, (float2*)geik, dedn, partials_e, loss_out, B); \
  }
  DISPATCH_SHAPE(PrecF32, LAUNCH_EIK, 0)
#undef LAUNCH_EIK
  NOF_LAUNCH_OK();
  return 0;
}

#include "nof_mlp_wide.h"
