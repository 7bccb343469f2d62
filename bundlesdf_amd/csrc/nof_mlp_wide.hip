// Wide / deep NeRFSmall shapes (hidden 128 and/or 4 layers per network: BASELINE.json cfg5's "MLP 4x128", nerf_helpers.py:243-321
// is parameterised in hidden_dim / num_layers) on the gfx950 matrix cores -- ON CHIP since round 6.  Its own translation unit on
// top of nof_mlp_dev.h (operand layouts, the fragment image of k_mlp_pack, the hash-encode pieces).
//
// Why a second set of kernels: the register-resident design of nof_mlp.hip keeps every weight fragment of both orientations in LDS
// and every dW accumulator in one wave's registers.  A 128x128 layer is 32 KB of fragments per orientation and 256 accumulator
// registers per wave; one 4x128 network is 78 KB per orientation and 608 accumulator registers.
//
// Rounds 2-5 staged everything through HBM (the forward stored six hidden activation rows per sample, the backward eight gradient
// rows, eight split-K weight-gradient launches read fourteen: 12.6 GB per cfg5 step, HBM-bound by construction).  Round 6:
//   * the forward stores NOTHING but raw, the sigma head's 16 outputs and (fused-encode entry point) the operand-precision
//     embedding -- 112 B per sample;
//   * the backward of one network is ONE kernel, k_wide_bwd_net: per 32-sample tile the forward is recomputed in registers, the
//     data gradients walk back through the layers, and the weight gradient is accumulated COOPERATIVELY by the four waves of a
//     workgroup: every wave transposes its tile's (delta_l, a_{l-1}) blocks on the matrix core (identity multiply, exact) and
//     parks the operand fragments in an LDS exchange buffer; behind a workgroup barrier every wave adds the four tiles' products
//     to the dW blocks it OWNS (block idx = wave + 4 i of the layer's PN x QN blocks: 38.9 k fp32 accumulators of a 4x128 network =
//     152 registers per lane spread over the workgroup's four waves; one wave per SIMD, 512-register budget);
//   * neither orientation of the weights is resident: the layer the workgroup is at streams through a two-slot LDS ring (one
//     chunk = one layer's fragments of one orientation, <= 32 KB, requested one phase ahead with global_load ... lds, shared by the
//     four waves), so LDS holds ring 64 KB + exchange 64 KB + biases;
//   * per workgroup ONE partial row of dW / db -> nof_reduce_partials, like the narrow path.
// Per tile and 128x128 layer: 32 (forward) + 32 (data gradient) + 16 (transposes) + 32 (dW) MFMAs against 64 KB of weight reads,
// 16 KB of exchange writes and 40 KB of exchange reads from LDS: matrix-pipe-bound on paper.
// 16-bit operand types only, no operand split: precisions 3 / 4 are refused (the Python host maps fp16x3 / bf16x3 to fp16 / bf16
// for these shapes and says so: NeuralObjectField.precision_effective).
#include "nof_mlp_dev.h"
#include <utility>

// ---- ReLU and its derivative on PACKED 16-bit operands (round 6).  relu(round16(x)) == round16(relu(x)) for both operand types
//      (rounding keeps sign and zero), so the backward kernel rounds first and clamps the packed halves: one v_pk_ashrrev_i16 +
//      one v_and (v_bfi) per PAIR instead of a v_max + a v_alignbit per element, and no derivative bit words -- the recomputed
//      activation itself (kept for the weight gradient anyway) says which units were on: a > 0 <=> the half's bits are non-zero.
typedef short s16x8 __attribute__((ext_vector_type(8)));
template <class F>
__device__ __forceinline__ F relu_pk(F x) {
  asm volatile("" : "+v"(x));        // (no instruction: the packed conversion result as four registers -- otherwise the combiner converts
                                     //  every half on its own and re-assembles the pairs with v_perm: 28 instead of 8 instructions per block)
  const s16x8 b = __builtin_bit_cast(s16x8, x);
  const s16x8 neg = b >> 15;                                             // 0xffff where the half is negative (incl. -0)
  return __builtin_bit_cast(F, (s16x8)(b & ~neg));
}
// g where the (post-ReLU, non-negative) activation a is non-zero, else +0
template <class F>
__device__ __forceinline__ F mask_pk(F g, F a) {
  asm volatile("" : "+v"(g));
  const s16x8 ab = __builtin_bit_cast(s16x8, a);
  // a in 0x0001 .. 0x7fff -> max(-a, -1) = -1 = 0xffff; a = 0 -> 0   (v_pk_sub_i16 + v_pk_max_i16 + v_and per pair)
  const s16x8 on = __builtin_elementwise_max((s16x8)((s16x8)(0) - ab), (s16x8)(-1));
  return __builtin_bit_cast(F, (s16x8)(__builtin_bit_cast(s16x8, g) & on));
}

#define WPAIR ((int)(16 * 64 * sizeof(typename P::elem)))

// =====================================================================================================
// backward of ONE network, everything on chip (header).  NET: 0 = sigma net (x0 = hash features, head gradient = dsig, input
// gradient = dfeat), 1 = colour net (x0 = [sigma head | view], head gradient = draw.xyz, input gradient = dsig (+ draw.w) and
// dview).  N = the network's layers (2..4), local layer k = 0..N-1 = global layer lbase + k.
// =====================================================================================================
template <int HB, int NET, int N>
struct WNet {
  static constexpr int QN0 = NET ? 2 : 1;
  static constexpr __host__ __device__ int qn(int k) { return k == 0 ? QN0 : HB; }
  static constexpr __host__ __device__ int pn(int k) { return k == N - 1 ? 1 : HB; }
  static constexpr __host__ __device__ int rel_pair(int k) { int s = 0; for (int i = 0; i < k; ++i) s += pn(i) * qn(i); return s; }
  static constexpr __host__ __device__ int rel_oblk(int k) { int s = 0; for (int i = 0; i < k; ++i) s += pn(i); return s; }
  static constexpr __host__ __device__ int nb(int k) { return (pn(k) * qn(k) + 3) / 4; }            // dW blocks a wave owns of layer k
  static constexpr __host__ __device__ int aoff(int k) { int s = 0; for (int i = 0; i < k; ++i) s += nb(i); return s; }
  static constexpr __host__ __device__ int slot_pairs() { int m = 0; for (int k = 0; k < N; ++k) m = pn(k) * qn(k) > m ? pn(k) * qn(k) : m; return m; }
  static constexpr int NACC = aoff(N);                                  // accumulator blocks per wave (x 16 registers)
  static constexpr int NDB = rel_oblk(N);                               // bias-gradient sums per wave (one register each)
  static constexpr int SLOTB = slot_pairs() * 2048;                     // bytes of one ring slot
  static constexpr int XWAVE = 2 * HB * 2048;                           // exchange bytes per wave: HB delta blocks + HB input blocks
  static constexpr int XB = 2 * SLOTB, BIASB = XB + 4 * XWAVE;
  static constexpr int IDB = BIASB + (N - 1) * HB * 128;                 // identity fragments of the MFMA transposes (2 KB)
  static constexpr int LDS_BYTES = IDB + 2048;
  static_assert(NDB <= 32 && HB <= 4, "one accumulator column per (layer, output block) of the bias gradient; block p belongs to owner p");
};

// the scheduler of a 512-register kernel hoists every LDS read it can see (all blocks' weight fragments, all tiles' operands) in
// front of the first MFMA and then spills what it hoisted: a scheduling fence per block keeps a block's reads beside its MFMAs
#ifndef NOF_WIDE_FENCE
#define NOF_WIDE_FENCE 1                                  // (without: the colour net's 4 x 128 data role spills five registers; 1.67 vs 1.70 ms at cfg5, r06_h)
#endif
#if NOF_WIDE_FENCE
#define WIDE_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define WIDE_FENCE() ((void)0)
#endif
#ifndef NOF_WIDE_DATA_PRIO
#define NOF_WIDE_DATA_PRIO 2                               // s_setprio of the data waves (the owners stay at 0)
#endif
#ifndef NOF_WIDE_DMA
#define NOF_WIDE_DMA 1                                    // weight chunks global -> LDS with global_load ... lds (0: through registers)
#endif

// one weight chunk (BYTES of the fragment image, contiguous) -> an LDS ring slot, by the workgroup's 256 threads.
// issue(): requests it; commit(): the requesting thread's part has landed (a workgroup barrier then publishes the slot).
template <int BYTES>
struct ChunkLoad {
#if NOF_WIDE_DMA
  __device__ __forceinline__ void issue(const char* __restrict__ src, char* slot, int wave_s, int lane) {
    // a wave instruction moves one contiguous kilobyte: LDS address = M0 (wave-uniform) + 16 * lane
    uint32_t lane16 = (uint32_t)lane * 16u;
    asm volatile("" : "+v"(lane16));      // (no instruction: the lane offset re-enters here, so the per-chunk 64-bit addresses are formed on
                                          //  the spot instead of being hoisted out of the persistent loop -- nine register pairs the colour
                                          //  net's owners, at 176 accumulator registers, spilled to scratch)
#pragma unroll
    for (int t = 0; t < BYTES / 4096; ++t) {
      const int piece = wave_s + 4 * t;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + piece * 1024 + lane16),
                                       (__attribute__((address_space(3))) void*)(slot + piece * 1024), 16, 0, 0);
    }
  }
  __device__ __forceinline__ void commit(char*, int, int) { __builtin_amdgcn_s_waitcnt(0x0f70); /* vmcnt(0) */ }
#else
  uint4 v[BYTES / 4096];
  __device__ __forceinline__ void issue(const char* __restrict__ src, char*, int wave_s, int lane) {
#pragma unroll
    for (int t = 0; t < BYTES / 4096; ++t) v[t] = *reinterpret_cast<const uint4*>(src + (wave_s + 4 * t) * 1024 + lane * 16);
  }
  __device__ __forceinline__ void commit(char* slot, int wave_s, int lane) {
#pragma unroll
    for (int t = 0; t < BYTES / 4096; ++t) *reinterpret_cast<uint4*>(slot + (wave_s + 4 * t) * 1024 + lane * 16) = v[t];
  }
#endif
};

template <class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>()), ...);
}
template <int NN, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, NN>()); }

// one dense layer on packed operands: out[p] = bias + W[p][:] in   (fragments of the layer at `wl` = slot + 16 * lane)
template <class P, int QN, int PN>
__device__ __forceinline__ void dense_pk(const char* wl, const char* bias_hi, const typename P::frag (&in)[QN][2], float (&out)[PN][16]) {
#pragma unroll
  for (int p = 0; p < PN; ++p) {
    f32x16 acc;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 bv = *(const float4*)(bias_hi + (32 * p + 8 * g) * 4);
      acc[4 * g] = bv.x; acc[4 * g + 1] = bv.y; acc[4 * g + 2] = bv.z; acc[4 * g + 3] = bv.w;
    }
    typename P::frag a[QN * 2];
#pragma unroll
    for (int t = 0; t < QN * 2; ++t) a[t] = *(const typename P::frag*)(wl + (p * QN * 2 + t) * 1024);
#pragma unroll
    for (int q = 0; q < QN; ++q)
#pragma unroll
      for (int s = 0; s < 2; ++s) acc = P::mma(a[q * 2 + s], in[q][s], acc);
#pragma unroll
    for (int r = 0; r < 16; ++r) out[p][r] = acc[r];
    WIDE_FENCE();
  }
}
// gradient of input block q: din = W^T[q][:] dout   (bw fragments of the layer at `wl`)
template <class P, int PN, int GB>
__device__ __forceinline__ void bwd_pk(const char* wl, int q, const typename P::frag (&dout)[GB][2], float (&din)[16]) {
  static_assert(PN <= GB, "the layer's output blocks are the first PN of the array");
  f32x16 a1;
#pragma unroll
  for (int r = 0; r < 16; ++r) a1[r] = 0.0f;
  typename P::frag w[PN * 2];
#pragma unroll
  for (int t = 0; t < PN * 2; ++t) w[t] = *(const typename P::frag*)(wl + (q * PN * 2 + t) * 1024);
#pragma unroll
  for (int p = 0; p < PN; ++p)
#pragma unroll
    for (int s = 0; s < 2; ++s) a1 = P::mma(w[p * 2 + s], dout[p][s], a1);
#pragma unroll
  for (int r = 0; r < 16; ++r) din[r] = a1[r];
}
template <class P>
__device__ __forceinline__ void pack_blk(const float (&x)[16], typename P::frag (&f)[2]) {
  f[0] = P::pack(&x[0]);
  f[1] = P::pack(&x[8]);
}
// a sample-per-lane block (two operand fragments) -> slot-per-lane on the matrix core (transpose32 of nof_mlp_dev.h on packed
// operands: x 1 + 0 is exact), as two operand fragments of the sample-contracted MFMA
template <class P, class ID>
__device__ __forceinline__ void transpose_pk(const ID& I, const typename P::frag (&x)[2], typename P::frag (&y)[2]) {
  f32x16 t;
#pragma unroll
  for (int r = 0; r < 16; ++r) t[r] = 0.0f;
  t = P::mma(x[0], ident_frag<P>(I, 0), t);
  t = P::mma(x[1], ident_frag<P>(I, 1), t);
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = t[r];
  y[0] = P::pack(&v[0]);
  y[1] = P::pack(&v[8]);
}

// The raw inputs of one tile as global memory delivers them -- the network's input block(s) and the head's gradient -- requested a
// step of the walk AHEAD of the pass that uses them (nothing touches the registers until the next pass unpacks them, so the loads
// stay in flight across the workgroup barriers; round 6's first form requested them at the top of the pass and waited right there,
// and read the input a second time for layer 0's weight gradient).
template <class P, int NET>
struct WideIn {
  typename P::frag q[2];                                                // NET 0: featq (two operand fragments); NET 1: q[0] = sigma head's output
  float f[16];                                                          // NET 0: the fp32 embedding (feat path); NET 1: f[0..7] = view row half
  typename P::frag hq;                                                  // NET 0: the head's gradient (dsig)
  float4 hd;                                                            // NET 1: draw
  __device__ __forceinline__ void fetch(const float2* __restrict__ feat, const typename P::elem* __restrict__ featq, int L,
                                        const typename P::elem* __restrict__ sig, const float* __restrict__ view, int S,
                                        const float4* __restrict__ draw, const typename P::elem* __restrict__ dsig, int64_t B,
                                        int64_t b, int hi) {
    typedef typename P::frag frag;
    const bool ok = b < B;
    frag z;
#pragma unroll
    for (int t = 0; t < 8; ++t) z[t] = (typename P::elem)0.0f;
    if constexpr (NET == 0) {
      if (featq != nullptr) {
        q[0] = z; q[1] = z;
        if (ok) {
          const frag* src = reinterpret_cast<const frag*>(featq + (b * 2 + hi) * 16);
          q[0] = src[0]; q[1] = src[1];
        }
      } else {
        float x[1][16];
        load_feat_o1(feat, L, B, b, hi, x);
#pragma unroll
        for (int r = 0; r < 16; ++r) f[r] = x[0][r];
      }
      hq = load_sig_raw<P>(dsig, B, b, hi);
    } else {
      q[0] = load_sig_raw<P>(sig, B, b, hi);
      float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
      if (ok) {                                                         // slot (hi, r < 8) = view column 8 hi + r (load_view_o1)
        const float4* v = (const float4*)(view + (b / S) * NOF_VIEW_COLS + 8 * hi);
        v0 = v[0]; v1 = v[1];
      }
      f[0] = v0.x; f[1] = v0.y; f[2] = v0.z; f[3] = v0.w; f[4] = v1.x; f[5] = v1.y; f[6] = v1.z; f[7] = v1.w;
      hd = make_float4(0.f, 0.f, 0.f, 0.f);
      if (hi == 0 && ok) hd = draw[b];
    }
  }
  // -> the input block(s) as MFMA operands (rounded where the forward kernel rounds them)
  template <int QN0>
  __device__ __forceinline__ void unpack(bool from_featq, typename P::frag (&x0)[QN0][2]) const {
    if constexpr (NET == 0) {
      if (from_featq) { x0[0][0] = q[0]; x0[0][1] = q[1]; }
      else pack_blk<P>(f, x0[0]);
    } else {
      typename P::frag z;
#pragma unroll
      for (int t = 0; t < 8; ++t) z[t] = (typename P::elem)0.0f;
      x0[0][0] = q[0]; x0[0][1] = z;                                    // the sigma head's 16 outputs: rows r < 8 of the block
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = r < 8 ? f[r] : 0.0f;
      pack_blk<P>(v, x0[1]);
    }
  }
};

// ---- software-pipelined block loops (round 6, second form).  The first on-chip form ran every block as read -> wait -> MFMA
//      chain -> wait -> pack, fenced block by block: the matrix pipe idled through every LDS latency and every result drain (the
//      data role alone, no owners and no exchange, ran at 42 % of its MFMA time; profiles/r06_l_wide_x.txt).  Here block p + 1's
//      fragments are requested BEFORE block p's chain and block p - 1's packing sits behind it, inside one fenced region, so the
//      reads land and the VALU work issues in the chain's shadow; two fragment sets and two accumulators alive (+ 48 registers).
template <class P, int T>
struct FragSet { typename P::frag a[T]; };
template <class P, int T>
__device__ __forceinline__ void load_frags(const char* base, FragSet<P, T>& w) {
#pragma unroll
  for (int t = 0; t < T; ++t) w.a[t] = *(const typename P::frag*)(base + t * 1024);
}
__device__ __forceinline__ f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int r = 0; r < 16; ++r) z[r] = 0.0f;
  return z;
}
template <class P>
__device__ __forceinline__ void pack_acc(const f32x16& acc, typename P::frag (&f)[2]) {
  float v[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = acc[r];
  f[0] = P::pack(&v[0]);
  f[1] = P::pack(&v[8]);
}

// out[p] = relu(round16(bias_p + W[p][:] in)), p < PN: the recompute of one hidden layer (fragments of the layer at `wl` = slot +
// 16 * lane, block p's QN * 2 fragments contiguous; `bias_hi` = the layer's first bias block + 16 * hi).  Same MFMA sequence per
// block as dense_pk (one accumulator chain over q, s): the same bits.
template <class P, int QN, int PN>
__device__ __forceinline__ void dense_relu_pipe(const char* wl, const char* bias_hi, const typename P::frag (&in)[QN][2],
                                                typename P::frag (&out)[PN][2]) {
  FragSet<P, QN * 2> w[2];
  float4 bv[2][4];
  f32x16 acc[2];
  auto load = [&](auto PC, auto BC) __attribute__((always_inline)) {
    constexpr int p = decltype(PC)::value, bf = decltype(BC)::value;
    load_frags<P, QN * 2>(wl + p * QN * 2048, w[bf]);
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[bf][g] = *(const float4*)(bias_hi + p * 128 + 32 * g);
  };
  auto finish = [&](auto PC, auto BC) __attribute__((always_inline)) {
    constexpr int p = decltype(PC)::value, bf = decltype(BC)::value;
    pack_acc<P>(acc[bf], out[p]);                                       // round, then ReLU on the packed halves (relu_pk)
    out[p][0] = relu_pk(out[p][0]);
    out[p][1] = relu_pk(out[p][1]);
  };
  load(std::integral_constant<int, 0>(), std::integral_constant<int, 0>());
  static_for<PN>([&](auto PC) __attribute__((always_inline)) {
    constexpr int p = decltype(PC)::value, cur = p & 1, nxt = cur ^ 1;
    if constexpr (p + 1 < PN) load(std::integral_constant<int, p + 1>(), std::integral_constant<int, nxt>());
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      acc[cur][4 * g] = bv[cur][g].x; acc[cur][4 * g + 1] = bv[cur][g].y; acc[cur][4 * g + 2] = bv[cur][g].z; acc[cur][4 * g + 3] = bv[cur][g].w;
    }
#pragma unroll
    for (int q = 0; q < QN; ++q)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) acc[cur] = P::mma(w[cur].a[q * 2 + s2], in[q][s2], acc[cur]);
    if constexpr (p > 0) finish(std::integral_constant<int, p - 1>(), std::integral_constant<int, nxt>());
    WIDE_FENCE();
  });
  finish(std::integral_constant<int, PN - 1>(), std::integral_constant<int, (PN - 1) & 1>());
  WIDE_FENCE();
}

// din[q] = W^T[q][:] dout for q = Q0 .. Q0 + NQ - 1 (bw fragments of the layer at `wl`, block q's PN * 2 fragments contiguous);
// done(q, acc) consumes block q's sums one region later.  Same chain per block as bwd_pk.
template <class P, int PN, int Q0, int NQ, int GB, class DONE>
__device__ __forceinline__ void bwd_pipe(const char* wl, const typename P::frag (&dout)[GB][2], DONE&& done) {
  static_assert(PN <= GB, "the layer's output blocks are the first PN of the array");
  FragSet<P, PN * 2> w[2];
  f32x16 acc[2];
  load_frags<P, PN * 2>(wl + Q0 * PN * 2048, w[0]);
  static_for<NQ>([&](auto QC) __attribute__((always_inline)) {
    constexpr int i = decltype(QC)::value, cur = i & 1, nxt = cur ^ 1;
    if constexpr (i + 1 < NQ) load_frags<P, PN * 2>(wl + (Q0 + i + 1) * PN * 2048, w[nxt]);
    acc[cur] = zero16();
#pragma unroll
    for (int p = 0; p < PN; ++p)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) acc[cur] = P::mma(w[cur].a[p * 2 + s2], dout[p][s2], acc[cur]);
    if constexpr (i > 0) done(std::integral_constant<int, Q0 + i - 1>(), acc[nxt]);
    WIDE_FENCE();
  });
  done(std::integral_constant<int, Q0 + NQ - 1>(), acc[(NQ - 1) & 1]);
  WIDE_FENCE();
}

// NB sample-per-lane blocks -> slot-per-lane (x 1 + 0 on the matrix core, exact: transpose_pk) -> the exchange region:
// src(i) = block i's two operand fragments, dst(i) = where its two transposed fragments go (+ 0 and + 1024)
template <class P, int NB, class SRC, class DST>
__device__ __forceinline__ void transpose_pipe(const typename P::frag (&I)[2], SRC&& src, DST&& dst) {
  typedef typename P::frag frag;
  f32x16 t[2];
  auto run = [&](auto IC, auto BC) __attribute__((always_inline)) {
    constexpr int bf = decltype(BC)::value;
    const frag (&x)[2] = src(IC);
    t[bf] = P::mma(x[0], I[0], zero16());
    t[bf] = P::mma(x[1], I[1], t[bf]);
  };
  run(std::integral_constant<int, 0>(), std::integral_constant<int, 0>());
  static_for<NB>([&](auto IC) __attribute__((always_inline)) {
    constexpr int i = decltype(IC)::value, cur = i & 1, nxt = cur ^ 1;
    if constexpr (i + 1 < NB) run(std::integral_constant<int, i + 1>(), std::integral_constant<int, nxt>());
    frag y[2];
    pack_acc<P>(t[cur], y);
    char* d = dst(IC);
    *reinterpret_cast<frag*>(d) = y[0];
    *reinterpret_cast<frag*>(d + 1024) = y[1];
    WIDE_FENCE();
  });
}

// =====================================================================================================
// forward, sigma net: features -> hidden layers -> head: sdf -> raw[b].w (or sdf[b]), sig[b] = 16 head outputs
// =====================================================================================================
// threads per workgroup of the forward kernels: 768 (3 waves per SIMD, 168 registers) where that fits without scratch (hidden 64);
// hidden 128 needs ~190 (two 64-register activation sets + the operand and weight fragments of a chain): 512 threads
template <int HB> struct WideFwdThreads { static constexpr int value = HB >= 4 ? 512 : 768; };
#ifndef NOF_WIDE_ENC_T
#define NOF_WIDE_ENC_T 768
#endif
// the fused-encode forward: its stage holds the features in OPERAND precision (4 KB per wave: what the chain's first layer and
// featq take anyway), so twelve waves fit beside the 78 KB sigma image -- a third wave per SIMD to hide the gathers of a 236 MB table
template <int HB> struct WideEncThreads { static constexpr int value = NOF_WIDE_ENC_T; };
#ifndef NOF_WIDE_COLOR_T4
#define NOF_WIDE_COLOR_T4 768
#endif
// the colour forward (no stage buffers, 161 registers on packed operands) has room for a third wave per SIMD
template <int HB> struct WideColorThreads { static constexpr int value = HB >= 4 ? NOF_WIDE_COLOR_T4 : 768; };

// One network's chain on PACKED operands (round 6: the forward kernels ran dense_o1 block by block on fp32 arrays -- read, wait,
// chain, drain, ReLU -- at 52-55 % matrix-pipe busy): hidden layers through dense_relu_pipe (the backward's recompute: the same
// bits by construction), then the head's single block.  `wl0` = the network's first fragment + 16 * lane, `bias_hi` = its first
// bias block + 16 * hi.
template <class P, int HB, int QN0>
__device__ __forceinline__ void wide_chain(const char* wl0, const char* bias_hi, int nlayers, const typename P::frag (&x0)[QN0][2],
                                           float (&out)[1][16]) {
  typename P::frag h[HB][2];
  dense_relu_pipe<P, QN0, HB>(wl0, bias_hi, x0, h);
  int foff = QN0 * HB * 2048, boff = HB * 128;
  for (int l = 1; l < nlayers - 1; ++l) {
    typename P::frag h2[HB][2];
    dense_relu_pipe<P, HB, HB>(wl0 + foff, bias_hi + boff, h, h2);
#pragma unroll
    for (int p = 0; p < HB; ++p) { h[p][0] = h2[p][0]; h[p][1] = h2[p][1]; }
    foff += HB * HB * 2048;
    boff += HB * 128;
  }
  dense_pk<P, HB, 1>(wl0 + foff, bias_hi + boff, h, out);
}

template <class P, int HB>
__global__ __launch_bounds__(WideFwdThreads<HB>::value) void k_wide_fwd_sigma(NofMlpDesc d, const char* __restrict__ image,
                                                         const float2* __restrict__ feat, int L,
                                                         float* __restrict__ out, int out_stride, int out_off,
                                                         typename P::elem* __restrict__ sig, int64_t B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NS = d.n_sigma, NL = d.n_sigma + d.n_color;
  const int bias_base = pair_base(d, NS) * WPAIR;
  copy16(smem, image, (size_t)bias_base);
  copy16(smem + bias_base, image + 2 * (size_t)pair_base(d, NL) * WPAIR, (size_t)oblk_base(d, NS) * 128);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int hi = lane >> 5, j = lane & 31;
  const int64_t ntiles = (B + 31) / 32, tstride = (int64_t)gridDim.x * nw;
  float xn[1][16];                                                      // the NEXT tile's features, a whole tile ahead
  load_feat_o1(feat, L, B, ((int64_t)blockIdx.x * nw + wave) * 32 + j, hi, xn);
  for (int64_t tile = (int64_t)blockIdx.x * nw + wave; tile < ntiles; tile += tstride) {
    asm volatile("" ::: "memory");
    const int64_t b = tile * 32 + j;
    const bool ok = b < B;
    float x[1][16], so[1][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) x[0][r] = xn[0][r];
    pin16(x[0]);
    load_feat_o1(feat, L, B, (tile + tstride) * 32 + j, hi, xn);
    typename P::frag x0[1][2];
    pack_blk<P>(x[0], x0[0]);
    wide_chain<P, HB, 1>(smem + lane * 16, smem + bias_base + hi * 16, NS, x0, so);
    if (sig != nullptr) store_sig_o1<P>(sig, B, b, hi, so[0]);
    if (hi == 0 && ok) out[b * out_stride + out_off] = so[0][0];
  }
}

// =====================================================================================================
// forward, sigma net WITH the hash encode in the launch (round 6; the narrow networks' k_enc_mlp_fwd, DESIGN 2.10, at width 128):
// pts_w -> 16 levels gathered, lane = sample, the level wave-uniform -> a wave-private LDS stage, feature-major -> the chain of
// two 32-sample tiles.  The fp32 embedding [L,B,2] (cfg5: 403 MB written by k_hash_fwd, read here, read again by the backward) is
// gone; what the backward needs of it is its value in operand precision, featq [B][2][16] (64 B per sample).
// =====================================================================================================
template <class P, int HB>
__global__ __launch_bounds__(WideEncThreads<HB>::value) void k_wide_enc_fwd_sigma(NofMlpDesc d, const char* __restrict__ image,
                                                             NofHashGrid g, const float2* __restrict__ table,
                                                             const float* __restrict__ pts_w, float* __restrict__ out, int out_stride,
                                                             int out_off, typename P::elem* __restrict__ sig,
                                                             typename P::elem* __restrict__ featq, int64_t B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NS = d.n_sigma, NL = d.n_sigma + d.n_color;
  const int bias_base = pair_base(d, NS) * WPAIR;
  const int stage_base = (bias_base + oblk_base(d, NS) * 128 + 15) & ~15;
  copy16(smem, image, (size_t)bias_base);
  copy16(smem + bias_base, image + 2 * (size_t)pair_base(d, NL) * WPAIR, (size_t)oblk_base(d, NS) * 128);
  const int NW = blockDim.x >> 6;
  uint32_t* lvl = reinterpret_cast<uint32_t*>(smem + stage_base + NW * 4096);       // [16][8] words behind the stages (see k_enc_mlp_fwd)
  if (threadIdx.x < NOF_MAX_LEVELS) {
    const int l = threadIdx.x;
    lvl[l * 8 + 0] = __float_as_uint(g.scale[l]); lvl[l * 8 + 1] = g.resolution[l]; lvl[l * 8 + 2] = g.offset[l];
    lvl[l * 8 + 3] = g.size[l]; lvl[l * 8 + 4] = g.hashed[l];
  }
  const int n_levels = g.L;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hi = lane >> 5, j = lane & 31;
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
  // [16 levels][64 samples] words: a level's two features of a sample, rounded to the operand type where they are parked (the
  // rounding the chain's first operand and featq apply anyway: same bits), one conflict-free ds_write_b32 per level
  uint32_t* const stage = reinterpret_cast<uint32_t*>(smem + stage_base + wave * 4096);
  typedef typename P::elem elem2 __attribute__((ext_vector_type(2)));
  const int64_t npairs = (B + 63) / 64;
  for (int64_t tp = (int64_t)blockIdx.x * NW + wave; tp < npairs; tp += (int64_t)gridDim.x * NW) {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_setprio(3);                      // (the gather phase issues first: DESIGN 2.10)
    const int64_t bs = tp * 64 + lane;
    const int64_t bb = bs < B ? bs : B - 1;             // (a lane past the end encodes the last sample: nothing of it is stored)
    const float p[3] = {pts_w[bb * 3], pts_w[bb * 3 + 1], pts_w[bb * 3 + 2]};
#pragma unroll 1
    for (int l0 = 0; l0 < NOF_MAX_LEVELS; ++l0) {
      float2 a = make_float2(0.f, 0.f);
      if (l0 < n_levels) {                              // (uniform)
        const HashLevel lv = level_at(l0);
        EncCell e = enc_prep(lv, p);
        float2 v[8];
        if (level_pairs(lv)) enc_load<true>(lv, table, e, v);
        else enc_load<false>(lv, table, e, v);
        enc_keep(e);
        a = enc_blend(e, v);
      }
      elem2 pr;
      pr[0] = (typename P::elem)a.x;
      pr[1] = (typename P::elem)a.y;
      stage[l0 * 64 + lane] = __builtin_bit_cast(uint32_t, pr);
    }
    __builtin_amdgcn_s_setprio(0);
    // the two tiles through the chain: lane (j, hi) of tile t reads features 16 hi .. 16 hi + 15 of sample 32 t + j
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      asm volatile("" ::: "memory");
      float so[1][16];
      // lane (j, hi) of tile t: features 16 hi .. 16 hi + 15 of sample 32 t + j = the words of levels 8 hi .. 8 hi + 7 (operand order)
      uint32_t w[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) w[r] = stage[(8 * hi + r) * 64 + 32 * t + j];
      const int64_t b = tp * 64 + 32 * t + j;
      const bool ok = b < B;
      typename P::frag x0[1][2];
      x0[0][0] = __builtin_bit_cast(typename P::frag, make_uint4(w[0], w[1], w[2], w[3]));
      x0[0][1] = __builtin_bit_cast(typename P::frag, make_uint4(w[4], w[5], w[6], w[7]));
      if (featq != nullptr && ok) {                     // the embedding as the backward reads it: rounded to the operand type, operand order
        typename P::frag* q = reinterpret_cast<typename P::frag*>(featq + (b * 2 + hi) * 16);
        q[0] = x0[0][0];
        q[1] = x0[0][1];
      }
      wide_chain<P, HB, 1>(smem + lane * 16, smem + bias_base + hi * 16, NS, x0, so);
      if (sig != nullptr) store_sig_o1<P>(sig, B, b, hi, so[0]);
      if (hi == 0 && ok) out[b * out_stride + out_off] = so[0][0];
    }
  }
}

// =====================================================================================================
// forward, colour net: [sig | view] -> hidden layers -> rgb_raw -> raw[b].xyz
// =====================================================================================================
template <class P, int HB>
__global__ __launch_bounds__(WideColorThreads<HB>::value) void k_wide_fwd_color(NofMlpDesc d, const char* __restrict__ image,
                                                         const typename P::elem* __restrict__ sig,
                                                         const float* __restrict__ view, int S,
                                                         float* __restrict__ raw, int64_t B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NS = d.n_sigma, NC = d.n_color, NL = NS + NC;
  const int PA = pair_base(d, NS), PB = pair_base(d, NL), OA = oblk_base(d, NS), OB = oblk_base(d, NL);
  const int bias_base = (PB - PA) * WPAIR;
  copy16(smem, image + (size_t)PA * WPAIR, (size_t)bias_base);
  copy16(smem + bias_base, image + 2 * (size_t)PB * WPAIR + OA * 128, (size_t)(OB - OA) * 128);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int hi = lane >> 5, j = lane & 31;
  const int64_t ntiles = (B + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * nw + wave; tile < ntiles; tile += (int64_t)gridDim.x * nw) {
    asm volatile("" ::: "memory");
    const int64_t b = tile * 32 + j;
    const bool ok = b < B;
    float vw[16], co[1][16];
    typename P::frag x0[2][2];                         // [sigma head | view] as the backward's WideIn<P, 1>::unpack builds them
    x0[0][0] = load_sig_raw<P>(sig, B, b, hi);        // (requested a tile ahead: 4-10 % slower at cfg5, measured twice)
#pragma unroll
    for (int t = 0; t < 8; ++t) x0[0][1][t] = (typename P::elem)0.0f;
    load_view_o1(view, S, B, b, hi, vw);
    pack_blk<P>(vw, x0[1]);
    wide_chain<P, HB, 2>(smem + lane * 16, smem + bias_base + hi * 16, NC, x0, co);
    if (hi == 0 && ok) { raw[b * 4] = co[0][0]; raw[b * 4 + 1] = co[0][1]; raw[b * 4 + 2] = co[0][2]; }
  }
}

// ---- the DATA role (waves 0-3 of the workgroup): one 32-sample tile per wave and pass -- forward recompute, data gradients,
//      the exchange operands.  Barriers: one per recomputed layer, two per layer of the walk -- the OWNER role runs the same sequence.
template <class P, int HB, int NET, int N>
__device__ __forceinline__ void wide_data_role(const NofMlpDesc& d, char* smem, int wave_s, int lane,
                                               const float2* __restrict__ feat, const typename P::elem* __restrict__ featq, int L,
                                               const typename P::elem* __restrict__ sig, const float* __restrict__ view, int S,
                                               const float4* __restrict__ draw, typename P::elem* __restrict__ dsig,
                                               float2* __restrict__ dfeat, float* __restrict__ dview, int64_t B,
                                               const TileWork& work, int64_t nbatch) {
  typedef WNet<HB, NET, N> W;
  typedef typename P::frag frag;
  const int hi = lane >> 5, j = lane & 31;
  IdentLds<P> I;                                                        // (in LDS: read once per step of the walk)
  I.base = reinterpret_cast<const typename P::frag*>(smem + W::IDB) + lane;
  const float gscale = d.grad_scale > 0.0f ? d.grad_scale : 1.0f, gunscale = 1.0f / gscale;
  char* const xw = smem + W::XB + wave_s * W::XWAVE + lane * 16;        // this wave's exchange region (+ its lane)
  const char* const bias_hi = smem + W::BIASB + hi * 16;
  int par = 0;                                                          // ring slot of the pass's first chunk (see the kernel)
  auto slot_of = [&](int c) { return smem + ((par + c) & 1) * W::SLOTB; };
  WideIn<P, NET> in;                                                    // the NEXT pass's raw inputs (in flight during the walk's last step)
  in.fetch(feat, featq, L, sig, view, S, draw, dsig, B, work.at((int64_t)blockIdx.x * 4 + wave_s) * 32 + j, hi);
  __builtin_amdgcn_s_setprio(NOF_WIDE_DATA_PRIO);                       // the data waves are the pass's critical path: they issue first
  for (int64_t bi = blockIdx.x; bi < nbatch; bi += gridDim.x) {
    asm volatile("" ::: "memory");
    const int64_t tile = work.at(bi * 4 + wave_s);                       // (past the end: a tile that does not exist -> zeros everywhere)
    const int64_t t0 = tile * 32, b = t0 + j;
    const bool ok = b < B;
    (void)ok;
    // ---------------- inputs: unpacked once, kept for layer 0's exchange at the end of the walk ----------------
    frag x0[W::QN0][2];
    in.template unpack<W::QN0>(featq != nullptr, x0);
    const float4 t_draw = in.hd;
    const frag f_dsig = in.hq;
    // ---------------- forward recompute: a[k] = relu(W_k a[k-1] + b_k), k < N - 1 ----------------
    frag a[N - 1][HB][2];
    static_for<N - 1>([&](auto K) __attribute__((always_inline)) {
      constexpr int k = decltype(K)::value;
      __syncthreads();                                                  // chunk k is in its slot
      const char* wl = slot_of(k) + lane * 16;
      if constexpr (k == 0) dense_relu_pipe<P, W::QN0, HB>(wl, bias_hi + W::rel_oblk(k) * 128, x0, a[k]);
      else dense_relu_pipe<P, HB, HB>(wl, bias_hi + W::rel_oblk(k) * 128, a[k - 1], a[k]);
    });
    // ---------------- the head's gradient ----------------
    frag g[HB][2];                                                      // delta of the layer the walk is at (block 0 only for the head)
    float dsdf1 = 0.0f;
    {
      float gh[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) gh[r] = 0.0f;
      if constexpr (NET == 1) {
        gh[0] = t_draw.x * gscale; gh[1] = t_draw.y * gscale; gh[2] = t_draw.z * gscale;
        dsdf1 = t_draw.w * gscale;
        pack_blk<P>(gh, g[0]);
      } else {
        sig_to_o1<P>(f_dsig, gh);
        pack_blk<P>(gh, g[0]);
      }
    }
    // ---------------- backward walk: layer k = N - 1 .. 0 ----------------
    static_for<N>([&](auto KK) __attribute__((always_inline)) {
      constexpr int k = N - 1 - decltype(KK)::value;
      constexpr int PN = W::pn(k), QN = W::qn(k), c = 2 * N - 2 - k;
      // exchange: delta_k (PN blocks) and the layer's input (QN blocks), transposed on the matrix core
      {
        const frag Id[2] = {I.get(0), I.get(1)};
        transpose_pipe<P, PN + QN>(Id,
          [&](auto IC) __attribute__((always_inline)) -> const frag (&)[2] {
            constexpr int i = decltype(IC)::value;
            if constexpr (i < PN) return g[i];
            else if constexpr (k == 0) return x0[i - PN];
            else return a[k > 0 ? k - 1 : 0][i - PN];
          },
          [&](auto IC) __attribute__((always_inline)) -> char* {
            constexpr int i = decltype(IC)::value;
            return xw + ((i < PN ? i : HB + i - PN) * 2) * 1024;
          });
      }
      __syncthreads();                                                  // (A) the four tiles' operands are in place; chunk c too
      if constexpr (k == 0) {                                           // the next pass's inputs: in flight from here
        if (bi + gridDim.x < nbatch)
          in.fetch(feat, featq, L, sig, view, S, draw, dsig, B, work.at((bi + gridDim.x) * 4 + wave_s) * 32 + j, hi);
      }
      // the data gradient of the layer's input
      const char* wl = slot_of(c) + lane * 16;
      if constexpr (k > 0) {
        frag gn[HB][2];
        bwd_pipe<P, PN, 0, HB>(wl, g, [&](auto QC, const f32x16& acc) __attribute__((always_inline)) {
          constexpr int q = decltype(QC)::value;
          pack_acc<P>(acc, gn[q]);                                      // round, then the ReLU derivative from the activation itself
          gn[q][0] = mask_pk(gn[q][0], a[k > 0 ? k - 1 : 0][q][0]);
          gn[q][1] = mask_pk(gn[q][1], a[k > 0 ? k - 1 : 0][q][1]);
        });
#pragma unroll
        for (int q = 0; q < HB; ++q) { g[q][0] = gn[q][0]; g[q][1] = gn[q][1]; }
      } else if constexpr (NET == 0) {
        float df1[16];
        bwd_pk<P, PN>(wl, 0, g, df1);
        store_dfeat_o1(dfeat, L, B, b, hi, df1, gunscale);
      } else {
        float ds1[16], dv1[16], dv2[16];
        bwd_pk<P, PN>(wl, 1, g, dv1);
        transpose32<P>(I, dv1, dv2);
        __builtin_amdgcn_sched_barrier(0);                              // (the view block first and done with, then the sigma block)
        bwd_pk<P, PN>(wl, 0, g, ds1);
        // dview[ray][u] += sum over the tile's samples (lane = view slot, regs <-> samples; a tile may straddle two rays)
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
        if (hi == 0) ds1[0] += dsdf1;                                   // the loss' own d sdf rides with the sigma head's gradient
        store_sig_o1<P>(dsig, B, b, hi, ds1);
      }
      __syncthreads();                                                  // (B) the exchange buffer is free again
    });
    par ^= 1;
  }
}

// ---- the OWNER role (waves 4-7): owner OW = wave - 4 (a template argument: which blocks a wave owns folds at compile time, no
//      wave-uniform branches around single MFMAs) holds the dW blocks idx = OW + 4 i of every layer, adds the four tiles' products
//      between the barriers (A) and (B) of a layer's step, and streams the weight chunks into the ring for everybody.
template <class P, int HB, int NET, int N, int OW>
__device__ __forceinline__ void wide_owner_role(const NofMlpDesc& d, char* smem, int lane, const char* __restrict__ fw_img,
                                                const char* __restrict__ bw_img, float* __restrict__ dst, int lbase, int64_t nbatch) {
  typedef WNet<HB, NET, N> W;
  typedef typename P::frag frag;
  constexpr int ow = OW;
  const int hi = lane >> 5, j = lane & 31;
  const float gunscale = d.grad_scale > 0.0f ? 1.0f / d.grad_scale : 1.0f;
  f32x16 acc[W::NACC];                                                  // the dW blocks this wave owns, all layers
#pragma unroll
  for (int a = 0; a < W::NACC; ++a) acc[a] = zero16();
  // The bias gradient db[m] = sum over samples of delta[m][sample] rides on the matrix core too: delta^T (the A operand of the dW
  // MFMA, already in the exchange) times a ONE-HOT column selector gives D[m][c] = db[m] in column c = the (layer, block)'s number --
  // one accumulator holds every block's sums side by side (NDB <= 32 columns), the data waves add nothing per tile (round 6: they
  // summed 16 registers per delta block and kept lane-private LDS words: 208 v_add + 26 LDS read-modify-writes per tile on the
  // critical path).  Block p of a layer belongs to owner p (PN <= 4).
  f32x16 accb = zero16();
  float one8[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) one8[t] = 1.0f;
  const frag ones = P::pack(one8);
  const char* const xr = smem + W::XB + lane * 16;                      // the exchange buffer as the owners read it
  int par = 0;
  auto slot_of = [&](int c) { return smem + ((par + c) & 1) * W::SLOTB; };
  for (int64_t bi = blockIdx.x; bi < nbatch; bi += gridDim.x) {
    asm volatile("" ::: "memory");
    static_for<N - 1>([&](auto K) __attribute__((always_inline)) {
      constexpr int k = decltype(K)::value;
      __syncthreads();                                                  // chunk k is published; the other slot's readers are done
      constexpr int kn = k + 1 < N - 1 ? k + 1 : N - 1;                 // next chunk: fw of layer k + 1, or bw of the head
      ChunkLoad<W::pn(kn) * W::qn(kn) * 2048> cl;
      cl.issue((k + 1 < N - 1 ? fw_img : bw_img) + W::rel_pair(kn) * 2048, slot_of(k + 1), ow, lane);
      cl.commit(slot_of(k + 1), ow, lane);
    });
    static_for<N>([&](auto KK) __attribute__((always_inline)) {
      constexpr int k = N - 1 - decltype(KK)::value;
      constexpr int PN = W::pn(k), QN = W::qn(k), c = 2 * N - 2 - k;
      __syncthreads();                                                  // (A)
      // next chunk: bw of layer k - 1, or -- the last step -- fw of layer 0 for the workgroup's next pass
      constexpr int kn = k > 0 ? k - 1 : 0;
      ChunkLoad<W::pn(kn) * W::qn(kn) * 2048> cl;
      const bool more = k > 0 || bi + gridDim.x < nbatch;
      if (more) cl.issue((k > 0 ? bw_img : fw_img) + W::rel_pair(kn) * 2048, slot_of(c + 1), ow, lane);
      // idx = OW + 4 i -> (p, q) = (idx / QN, idx % QN).  One unit of work = one (tile, K half): the wave's NBK blocks' products
      // (consecutive MFMAs go to different accumulators) + the bias selector; unit u + 1's operands are requested before unit
      // u's MFMAs (two operand sets alive).
      {
        constexpr int NBK = (PN * QN - ow + 3) / 4 > 0 ? (PN * QN - ow + 3) / 4 : 0;      // blocks of this layer the wave owns
        constexpr bool mine = ow < PN;                                  // the bias block p = OW
        frag sel;
#pragma unroll
        for (int t = 0; t < 8; ++t) sel[t] = j == W::rel_oblk(k) + ow ? ones[t] : (typename P::elem)0.0f;
        struct Ops { frag dl[NBK > 0 ? NBK : 1], av[NBK > 0 ? NBK : 1], db; };
        Ops op[2];
        auto fetch = [&](auto UC, auto BC) __attribute__((always_inline)) {
          constexpr int u = decltype(UC)::value, bf = decltype(BC)::value, w4 = u >> 1, s2 = u & 1;
          const char* base = xr + w4 * W::XWAVE + s2 * 1024;
#pragma unroll
          for (int i = 0; i < NBK; ++i) {
            const int idx = ow + 4 * i;
            op[bf].dl[i] = *reinterpret_cast<const frag*>(base + ((idx / QN) * 2) * 1024);
            op[bf].av[i] = *reinterpret_cast<const frag*>(base + ((HB + idx % QN) * 2) * 1024);
          }
          if constexpr (mine) op[bf].db = *reinterpret_cast<const frag*>(base + (ow * 2) * 1024);
        };
        if constexpr (NBK > 0 || mine) {
          fetch(std::integral_constant<int, 0>(), std::integral_constant<int, 0>());
          static_for<8>([&](auto UC) __attribute__((always_inline)) {
            constexpr int u = decltype(UC)::value, cur = u & 1, nxt = cur ^ 1;
            if constexpr (u + 1 < 8) fetch(std::integral_constant<int, u + 1>(), std::integral_constant<int, nxt>());
#pragma unroll
            for (int i = 0; i < NBK; ++i) acc[W::aoff(k) + i] = P::mma(op[cur].dl[i], op[cur].av[i], acc[W::aoff(k) + i]);
            if constexpr (mine) accb = P::mma(op[cur].db, sel, accb);
            __builtin_amdgcn_sched_barrier(0);
          });
        }
      }
      if (more) cl.commit(slot_of(c + 1), ow, lane);
      __syncthreads();                                                  // (B)
    });
    par ^= 1;
  }
  // ---------------- the workgroup's row of `partials`: every (layer, block) has exactly one owner ----------------
  const int hi_j = (j >> 2) & 1, r_j = (j & 3) + 4 * (j >> 3);          // lane j = input slot (hi_j, r_j) of block q
  static_for<N>([&](auto K) __attribute__((always_inline)) {
    constexpr int k = decltype(K)::value;
    constexpr int PN = W::pn(k), QN = W::qn(k);
    const int l = lbase + k;
    const int in_dim = d.in_dim[l], out_dim = d.out_dim[l];
#pragma unroll
    for (int i = 0; i < W::nb(k); ++i) {
      constexpr int dummy = 0; (void)dummy;
      const int idx = ow + 4 * i;
      if (idx < PN * QN) {
        const int p = idx / QN, q = idx % QN;
        const int col = inmap(d, l, q, hi_j, r_j);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int orow = 32 * p + nloc(hi, r);
          if (col >= 0 && orow < out_dim) dst[d.w_off[l] + orow * in_dim + col] = acc[W::aoff(k) + i][r] * gunscale;
        }
      }
    }
    // bias gradients: column rel_oblk(k) + p of accb, rows = the block's neurons (lane j = column, hi = row half)
    if (ow < PN && j == W::rel_oblk(k) + ow) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int orow = 32 * ow + nloc(hi, r);
        if (orow < out_dim) dst[d.b_off[l] + orow] = accb[r] * gunscale;
      }
    }
  });
}

template <class P, int HB, int NET, int N>
__global__ __launch_bounds__(512, 2) void k_wide_bwd_net(NofMlpDesc d, const char* __restrict__ image,
                                                          const float2* __restrict__ feat, const typename P::elem* __restrict__ featq,
                                                          int L, const typename P::elem* __restrict__ sig,
                                                          const float* __restrict__ view, int S, const float4* __restrict__ draw,
                                                          typename P::elem* __restrict__ dsig, float2* __restrict__ dfeat,
                                                          float* __restrict__ dview, float* __restrict__ partials, int64_t B,
                                                          const void* __restrict__ tile_list) {
  typedef WNet<HB, NET, N> W;
  static_assert(P::KR == 8, "16-bit operand types");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave_s = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lbase = NET ? d.n_sigma : 0, NL = d.n_sigma + d.n_color;
  const int PA = pair_base(d, lbase), NPALL = pair_base(d, NL), OA = oblk_base(d, lbase);
  const char* __restrict__ fw_img = image + (size_t)PA * 2048;
  const char* __restrict__ bw_img = image + (size_t)(NPALL + PA) * 2048;
  // biases of the layers the recompute runs (k < N - 1)
  copy16(smem + W::BIASB, image + 2 * (size_t)NPALL * 2048 + (size_t)OA * 128, (size_t)(N - 1) * HB * 128);
  const int64_t ntiles = (B + 31) / 32;
  const TileWork work(tile_list, ntiles);
  const int64_t nbatch = (work.n + 3) / 4;                              // four tiles per workgroup pass: one per data wave
  // Weight chunks: chunk c of a pass = the fw fragments of layer c (c < N - 1), else the bw fragments of layer 2 N - 2 - c; global
  // chunk g = pass * (2 N - 1) + c sits in ring slot g & 1 = (par + c) & 1 (2 N - 1 is odd: par flips every pass).  Chunk g + 1 is
  // requested while chunk g is in use, into the slot chunk g - 1 was read from one barrier ago.
  float* __restrict__ dst = partials + (size_t)blockIdx.x * d.n_params;
#ifndef NOF_WIDE_ROLES
#define NOF_WIDE_ROLES 3                                  // (register-budget experiments: 1 = data role only, 2 = owner role only)
#endif
  { IdentLds<P> I0; I0.build(smem + W::IDB, lane); }                     // (wave 0 writes; the roles' first barrier publishes)
  if (wave_s < 4) {
    if (NOF_WIDE_ROLES & 1) wide_data_role<P, HB, NET, N>(d, smem, wave_s, lane, feat, featq, L, sig, view, S, draw, dsig, dfeat, dview, B, work, nbatch);
  } else {
    if ((int64_t)blockIdx.x < nbatch) {
      ChunkLoad<W::pn(0) * W::qn(0) * 2048> c0;
      c0.issue(fw_img, smem, wave_s - 4, lane);
      c0.commit(smem, wave_s - 4, lane);
    }
    if (NOF_WIDE_ROLES & 2) {
      switch (wave_s) {                                                 // (wave-uniform)
        case 4: wide_owner_role<P, HB, NET, N, 0>(d, smem, lane, fw_img, bw_img, dst, lbase, nbatch); break;
        case 5: wide_owner_role<P, HB, NET, N, 1>(d, smem, lane, fw_img, bw_img, dst, lbase, nbatch); break;
        case 6: wide_owner_role<P, HB, NET, N, 2>(d, smem, lane, fw_img, bw_img, dst, lbase, nbatch); break;
        default: wide_owner_role<P, HB, NET, N, 3>(d, smem, lane, fw_img, bw_img, dst, lbase, nbatch); break;
      }
    }
  }
}

// =====================================================================================================
// host side
// =====================================================================================================
static int check_wide(const NofMlpDesc* d) {
  if (int e = check_desc(d)) return e;
  if (d->precision == 0)
    return nof_set_error(-1, "mlp (wide path, hidden %d depths %d,%d): 16-bit operand types only (fp32 fragments do not fit LDS)",
                         d->hidden, d->n_sigma, d->n_color);
  if (d->precision != 1 && d->precision != 2)      // (rounds 3-5 ran 3 / 4 silently as 2 / 1: the name promised a split that was not there)
    return nof_set_error(-1, "mlp (wide path, hidden %d depths %d,%d): no hi+lo operand split here -- precision %d is not available, "
                         "pass 2 (fp16) or 1 (bf16)", d->hidden, d->n_sigma, d->n_color, d->precision);
  return 0;
}

// workspace layout (bytes, 256-aligned): [sig : B * 16 elems][dsig : B * 16 elems] -- the sigma head's output (forward -> colour
// forward, colour backward) and its gradient (colour backward -> sigma backward), operand precision
struct WideWs { char *sig, *dsig; int64_t total; };
static WideWs wide_ws(const NofMlpDesc*, void* base, int64_t B) {
  auto up = [](int64_t x) { return (x + 255) / 256 * 256; };
  WideWs w;
  int64_t off = 0;
  w.sig = (char*)base + off; off += up(B * 32);
  w.dsig = (char*)base + off; off += up(B * 32);
  w.total = off;
  return w;
}
extern "C" int64_t nof_mlp_wide_workspace_bytes(const NofMlpDesc* d, int64_t B) {
  if (check_wide(d) || B < 0) return -1;
  return wide_ws(d, nullptr, B).total;
}
// rows of `partials`: one per workgroup of the backward kernels = one per compute unit
extern "C" int nof_mlp_wide_partial_rows(void) { return nof_cu_count(); }

template <class P, int HB>
static int wide_fwd_launch(const NofMlpDesc* d, const void* packed, const float* feat, int32_t L, const float* view, int32_t S,
                           float* out, int out_stride, int out_off, const WideWs* ws, bool sdf_only, int64_t B, hipStream_t st) {
  const int ns = d->n_sigma, nl = d->n_sigma + d->n_color;
  const size_t pair_bytes = 16 * 64 * 2;
  const size_t shm_s = (size_t)pair_base(*d, ns) * pair_bytes + (size_t)oblk_base(*d, ns) * 128;
  const size_t shm_c = (size_t)(pair_base(*d, nl) - pair_base(*d, ns)) * pair_bytes + (size_t)(oblk_base(*d, nl) - oblk_base(*d, ns)) * 128;
  const int64_t ntiles = (B + 31) / 32;
  constexpr int NT = WideFwdThreads<HB>::value;
  const unsigned blocks = (unsigned)(nof_div_up(ntiles, NT / 64) < (int64_t)nof_cu_count() ? nof_div_up(ntiles, NT / 64) : nof_cu_count());
  typedef typename P::elem elem;
  auto ks = k_wide_fwd_sigma<P, HB>;
  if (int e = set_smem(ks, shm_s)) return e;
  hipLaunchKernelGGL(ks, dim3(blocks), dim3(NT), shm_s, st, *d, (const char*)packed, (const float2*)feat, (int)L, out, out_stride,
                     out_off, sdf_only ? (elem*)nullptr : (elem*)ws->sig, B);
  if (!sdf_only) {
    constexpr int NTC = WideColorThreads<HB>::value;
    const unsigned blocks_c = (unsigned)(nof_div_up(ntiles, NTC / 64) < (int64_t)nof_cu_count() ? nof_div_up(ntiles, NTC / 64) : nof_cu_count());
    auto kc = k_wide_fwd_color<P, HB>;
    if (int e = set_smem(kc, shm_c)) return e;
    hipLaunchKernelGGL(kc, dim3(blocks_c), dim3(NTC), shm_c, st, *d, (const char*)packed, (const elem*)ws->sig, view, (int)S, out, B);
  }
  return 0;
}

#define WIDE_DISPATCH(FN, ...)                                                                            \
  if (is_bf16(d->precision)) {                                                                            \
    if (d->hidden == 128) { if (int e = FN<PrecBF16, 4>(__VA_ARGS__)) return e; }                         \
    else { if (int e = FN<PrecBF16, 2>(__VA_ARGS__)) return e; }                                          \
  } else {                                                                                                \
    if (d->hidden == 128) { if (int e = FN<PrecF16, 4>(__VA_ARGS__)) return e; }                          \
    else { if (int e = FN<PrecF16, 2>(__VA_ARGS__)) return e; }                                           \
  }

/* feat [L,B,2], view [R,16] -> raw [B,4]; the sigma head's output stays in `workspace` for nof_mlp_wide_bwd
 * (workspace: nof_mlp_wide_workspace_bytes(desc, B) bytes, caller-allocated). */
extern "C" int nof_mlp_wide_fwd(const NofMlpDesc* d, const void* packed, const float* feat, int32_t L, const float* view,
                                 int32_t S, float* raw, void* workspace, int64_t B, void* stream) {
  if (int e = check_wide(d)) return e;
  NOF_ARG(packed && feat && view && raw && workspace && B >= 0 && S >= 1 && L >= 1 && L * 2 == d->in_feat);
  NOF_ARG((int64_t)L * B * 8 < (1ll << 32));                   // level-major arrays are addressed with 32-bit lane offsets
  if (B == 0) return 0;
  const WideWs ws = wide_ws(d, workspace, B);
  WIDE_DISPATCH(wide_fwd_launch, d, packed, feat, L, view, S, raw, 4, 3, &ws, false, B, (hipStream_t)stream)
  NOF_LAUNCH_OK();
  return 0;
}

/* sigma net only: feat [L,B,2] -> sdf [B] (NeRFSmall.forward_sdf); needs no workspace */
extern "C" int nof_mlp_wide_sdf(const NofMlpDesc* d, const void* packed, const float* feat, int32_t L, float* sdf, int64_t B,
                                 void* stream) {
  if (int e = check_wide(d)) return e;
  NOF_ARG(packed && feat && sdf && B >= 0 && L >= 1 && L * 2 == d->in_feat);
  NOF_ARG((int64_t)L * B * 8 < (1ll << 32));                   // level-major arrays are addressed with 32-bit lane offsets
  if (B == 0) return 0;
  WIDE_DISPATCH(wide_fwd_launch, d, packed, feat, L, (const float*)nullptr, 1, sdf, 1, 0, (const WideWs*)nullptr, true, B,
                (hipStream_t)stream)
  NOF_LAUNCH_OK();
  return 0;
}

template <class P, int HB>
static int wide_enc_fwd_launch(const NofHashGrid* g, const NofMlpDesc* d, const void* packed, const float* table, const float* pts_w,
                               const float* view, int32_t S, float* raw, const WideWs* ws, void* featq, int64_t B, hipStream_t st) {
  const int ns = d->n_sigma, nl = d->n_sigma + d->n_color;
  const size_t pair_bytes = 16 * 64 * 2;
  constexpr int NT = WideEncThreads<HB>::value, NWV = NT / 64;
  const size_t img_s = (((size_t)pair_base(*d, ns) * pair_bytes + (size_t)oblk_base(*d, ns) * 128) + 15) & ~(size_t)15;
  const size_t shm_s = img_s + (size_t)NWV * 4096 + 512;               // fragments + biases | 4 KB of stage per wave | the level table
  if (shm_s > 160 * 1024) return nof_set_error(-1, "nof_encode_mlp_wide_fwd: %zu bytes of LDS", shm_s);
  const size_t shm_c = (size_t)(pair_base(*d, nl) - pair_base(*d, ns)) * pair_bytes + (size_t)(oblk_base(*d, nl) - oblk_base(*d, ns)) * 128;
  typedef typename P::elem elem;
  const int64_t npairs = (B + 63) / 64, ntiles = (B + 31) / 32;
  const unsigned blocks_s = (unsigned)(nof_div_up(npairs, NWV) < (int64_t)nof_cu_count() ? nof_div_up(npairs, NWV) : nof_cu_count());
  constexpr int NTC = WideColorThreads<HB>::value;
  const unsigned blocks_c = (unsigned)(nof_div_up(ntiles, NTC / 64) < (int64_t)nof_cu_count() ? nof_div_up(ntiles, NTC / 64) : nof_cu_count());
  auto ks = k_wide_enc_fwd_sigma<P, HB>;
  if (int e = set_smem(ks, shm_s)) return e;
  hipLaunchKernelGGL(ks, dim3(blocks_s), dim3(NT), shm_s, st, *d, (const char*)packed, *g, (const float2*)table, pts_w, raw, 4, 3,
                     (elem*)ws->sig, (elem*)featq, B);
  auto kc = k_wide_fwd_color<P, HB>;
  if (int e = set_smem(kc, shm_c)) return e;
  hipLaunchKernelGGL(kc, dim3(blocks_c), dim3(NTC), shm_c, st, *d, (const char*)packed, (const elem*)ws->sig, view, (int)S, raw, B);
  return 0;
}

/* Hash encode + both wide networks, the fp32 embedding never in HBM (the wide counterpart of nof_encode_mlp_fwd; replaces the pair
 * nof_hash_encode_fwd + nof_mlp_wide_fwd of the training forward, reference nerf_runner.py:1255-1294): pts_w [B,3], table [rows,2],
 * view [R,16] -> raw [B,4]; the sigma head's output stays in `workspace` (nof_mlp_wide_workspace_bytes); featq (may be NULL):
 * [B][2][16] operand-type elements, the embedding as nof_mlp_wide_bwd_parts reads it. */
extern "C" int nof_encode_mlp_wide_fwd(const NofHashGrid* g, const NofMlpDesc* d, const void* packed, const float* table,
                                        const float* pts_w, const float* view, int32_t S, float* raw, void* workspace, void* featq,
                                        int64_t B, void* stream) {
  if (int e = check_wide(d)) return e;
  NOF_ARG(g && g->C == 2 && g->L >= 1 && g->L <= NOF_MAX_LEVELS && g->L * 2 == d->in_feat);
  NOF_ARG(packed && table && pts_w && view && raw && workspace && B >= 0 && S >= 1);
  if (B == 0) return 0;
  const WideWs ws = wide_ws(d, workspace, B);
  WIDE_DISPATCH(wide_enc_fwd_launch, g, d, packed, table, pts_w, view, S, raw, &ws, featq, B, (hipStream_t)stream)
  NOF_LAUNCH_OK();
  return 0;
}

template <class P, int HB, int NET, int N>
static int wide_bwd_net_launch(const NofMlpDesc* d, const void* packed, const float* feat, const void* featq, int32_t L,
                               const float* view, int32_t S, const float* draw, const WideWs* ws, float* dfeat, float* dview,
                               float* partials, int64_t B, hipStream_t st, const void* tile_list) {
  typedef typename P::elem elem;
  typedef WNet<HB, NET, N> W;
  auto k = k_wide_bwd_net<P, HB, NET, N>;
  if (int e = set_smem(k, (size_t)W::LDS_BYTES)) return e;
  hipLaunchKernelGGL(k, dim3((unsigned)nof_cu_count()), dim3(512), (size_t)W::LDS_BYTES, st, *d, (const char*)packed,
                     (const float2*)feat, (const elem*)featq, (int)L, (const elem*)ws->sig, view, (int)S, (const float4*)draw,
                     (elem*)ws->dsig, (float2*)dfeat, dview, partials, B, tile_list);
  return 0;
}
template <class P, int HB>
static int wide_bwd_launch(const NofMlpDesc* d, const void* packed, const float* feat, const void* featq, int32_t L, const float* view,
                           int32_t S, const float* draw, const WideWs* ws, float* dfeat, float* dview, float* partials, int64_t B,
                           hipStream_t st, const void* tile_list, int parts) {
#define WIDE_NET(NET_, n_)                                                                                               \
  switch (n_) {                                                                                                          \
    case 2: if (int e = wide_bwd_net_launch<P, HB, NET_, 2>(d, packed, feat, featq, L, view, S, draw, ws, dfeat, dview, partials, B, st, tile_list)) return e; break; \
    case 3: if (int e = wide_bwd_net_launch<P, HB, NET_, 3>(d, packed, feat, featq, L, view, S, draw, ws, dfeat, dview, partials, B, st, tile_list)) return e; break; \
    default: if (int e = wide_bwd_net_launch<P, HB, NET_, 4>(d, packed, feat, featq, L, view, S, draw, ws, dfeat, dview, partials, B, st, tile_list)) return e; break; \
  }
  if (parts & NOF_WIDE_BWD_COLOR) { WIDE_NET(1, d->n_color) }
  if (parts & NOF_WIDE_BWD_SIGMA) { WIDE_NET(0, d->n_sigma) }
#undef WIDE_NET
  return 0;
}

/* draw [B,4] -> dfeat [L,B,2] (overwritten), dview [R,16] ACCUMULATED, partials [nof_mlp_wide_partial_rows(), n_params]
 * overwritten (sum the rows with nof_reduce_partials).  `workspace` as left by nof_mlp_wide_fwd of the same batch. */
extern "C" int nof_mlp_wide_bwd(const NofMlpDesc* d, const void* packed, const float* feat, int32_t L, const float* view,
                                 int32_t S, const float* draw, void* workspace, float* dfeat, float* dview, float* partials,
                                 int64_t B, void* stream) {
  return nof_mlp_wide_bwd_tiles(d, packed, feat, L, view, S, draw, workspace, dfeat, dview, partials, nullptr, B, stream);
}

/* the same over a work list (NofTileList): only the listed tiles are computed, four at a time per workgroup; dfeat of unlisted
 * tiles is not written */
extern "C" int nof_mlp_wide_bwd_tiles(const NofMlpDesc* d, const void* packed, const float* feat, int32_t L, const float* view,
                                       int32_t S, const float* draw, void* workspace, float* dfeat, float* dview, float* partials,
                                       const void* tile_list, int64_t B, void* stream) {
  return nof_mlp_wide_bwd_parts(d, packed, feat, nullptr, L, view, S, draw, workspace, dfeat, dview, partials, tile_list,
                                NOF_WIDE_BWD_ALL, B, stream);
}

/* The same, restricted to `parts` (all on `stream`): the colour net's kernel (needs draw; writes the sigma head's gradient into the
 * workspace, dview and the colour layers' entries of every partial row) and the sigma net's (needs the colour part; writes dfeat
 * and the sigma layers' entries).  `featq` (may be NULL): the embedding in operand precision as nof_encode_mlp_wide_fwd leaves it,
 * [B][2][16] elements, read instead of `feat` (which may then be NULL). */
extern "C" int nof_mlp_wide_bwd_parts(const NofMlpDesc* d, const void* packed, const float* feat, const void* featq, int32_t L,
                                       const float* view, int32_t S, const float* draw, void* workspace, float* dfeat, float* dview,
                                       float* partials, const void* tile_list, int32_t parts, int64_t B, void* stream) {
  if (int e = check_wide(d)) return e;
  NOF_ARG(parts >= 0 && parts <= NOF_WIDE_BWD_ALL);
  NOF_ARG(packed && (feat || featq) && view && draw && workspace && dfeat && dview && partials && B >= 0 && S >= 32 && L * 2 == d->in_feat);
  NOF_ARG((int64_t)L * B * 8 < (1ll << 32));                   // level-major arrays are addressed with 32-bit lane offsets
  if (B == 0) return 0;
  const WideWs ws = wide_ws(d, workspace, B);
  WIDE_DISPATCH(wide_bwd_launch, d, packed, feat, featq, L, view, S, draw, &ws, dfeat, dview, partials, B, (hipStream_t)stream, tile_list, (int)parts)
  NOF_LAUNCH_OK();
  return 0;
}
#undef WPAIR
