// Shared device primitives and host helpers of the MLP kernels (nof_mlp.hip: hidden 64, register-resident; nof_mlp_wide.hip:
// hidden 128 / 4 layers).  Moved out of nof_mlp.hip in round 6 so that the wide networks are their own translation unit.
#pragma once
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
#ifndef NOF_SPLIT_FMA_MIX
#define NOF_SPLIT_FMA_MIX 1
#endif
template <class P, int QN, int PN, bool SPLIT = false>
__device__ __forceinline__ void dense_o1(const char* smem, int frag_off, int bias_off, const float (&in)[QN][16],
                                         float (&out)[PN][16], int lane, int lo_off = 0) {
  constexpr int KR = P::KR, NSTEP = 16 / KR;
  constexpr int FB = 64 * KR * (int)sizeof(typename P::elem);          // bytes of one fragment (all 64 lanes)
  static_assert(!SPLIT || KR == 8, "the operand split is for the 16-bit operand types");
  const int hi = lane >> 5;
  typename P::frag bop[QN][NSTEP];
  typename P::frag blo[SPLIT ? QN : 1][SPLIT ? NSTEP : 1];
  float negone = -1.0f;
  if constexpr (SPLIT) asm volatile("" : "+s"(negone));               // (no instruction: see the residual below)
#pragma unroll
  for (int q = 0; q < QN; ++q)
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
      bop[q][s] = P::pack(&in[q][KR * s]);
      if constexpr (SPLIT) {
        float res[KR];
        if constexpr (NOF_SPLIT_FMA_MIX && sizeof(typename P::elem) == 2 && P::KR == 8 && __is_same(typename P::elem, _Float16)) {
          // residual lo = round16(x - float(hi)) as ONE v_fma_mixlo_f16 / v_fma_mixhi_f16 per element (f16 source, fp32 addend, f16
          // result into its half of the operand register): float(hi) * (-1) + x is exact -- x - round16(x) always fits fp32 -- so
          // the single rounding to f16 gives the bits of the three-instruction form v_cvt_f32_f16, v_sub_f32, v_cvt_pk_f16_f32.  The
          // -1 is opaque to the optimiser, which otherwise folds the fma back into the subtraction (round 6: 4 -> 2.5 VALU
          // instructions per split element, 12 % of the fused forward's instructions).  tests/test_gpu_erratum.py: both forms are
          // exact under MFMA load (forms 7, 8); the bitwise fused == two-launch and repeatability tests run through them.
#pragma unroll
          for (int t = 0; t < KR; ++t) res[t] = __builtin_fmaf((float)bop[q][s][t], negone, in[q][KR * s + t]);
        } else {
#pragma unroll
          for (int t = 0; t < KR; ++t) res[t] = in[q][KR * s + t] - (float)bop[q][s][t];
        }
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

