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
#include "nof_mlp_dev.h"
#include "nof_pose_dev.h"
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
// Round 6: the encode of the TRAINING forward (k_enc_mlp_fwd) -- lane = voxel, the level wave-uniform (its constants in SGPRs, one
// saddr gather instruction per corner row for 64 voxels), the features parked feature-major in a wave-private 8 KB LDS stage -- then
// the sigma net over the stage's two 32-voxel tiles.  Rounds 2-5 put (voxel, 8 of the 16 levels) on a lane: 32 voxels per wave pass,
// every level's constants per lane, 64 gathers per lane; 5.5 ms for 512^3 against the 2.2 ms the training forward needs for as many
// points.  A wave owns 64 consecutive voxels of one z column; occupancy is spatially coherent (level <= 6 cells vs 1/512 voxels), so a
// wave-uniform skip of all-empty columns (and of an empty half) is the whole compaction that is needed; voxels outside the octree
// issue no gathers (their lanes are masked).
// =====================================================================================================
template <class P, int NS, int NC, bool SPLIT>
__global__ __launch_bounds__(64 * NOF_ENC_WAVES, (NOF_ENC_WAVES + 3) / 4) void k_sdf_grid(
    NofMlpDesc d, const char* __restrict__ image, NofHashGrid g, const float2* __restrict__ table, const uint32_t* __restrict__ occ_bits,
    int occ_n, const float* __restrict__ tx, const float* __restrict__ ty, const float* __restrict__ tz, int nx, int ny, int nz,
    float outside, float* __restrict__ sdf) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef Shp<NS, NC> SH;
  constexpr int BIAS_BASE = SH::pair_base(NS) * PAIR_BYTES;
  constexpr int LO_BASE = BIAS_BASE + SH::oblk_base(NS) * 32 * 4;
  constexpr int PARK_BASE = (LO_BASE + (SPLIT ? BIAS_BASE : 0) + 15) & ~15;     // [wave][32][64] floats: the wave's feature stage
  copy16(smem, image, (size_t)BIAS_BASE);
  copy16(smem + BIAS_BASE, image + 2 * (size_t)SH::pair_base(NS + NC) * PAIR_BYTES, (size_t)SH::oblk_base(NS) * 32 * 4);
  if constexpr (SPLIT)
    copy16(smem + LO_BASE, image + 2 * (size_t)SH::pair_base(NS + NC) * PAIR_BYTES + (size_t)SH::oblk_base(NS + NC) * 32 * 4,
           (size_t)BIAS_BASE);
  const int NW = blockDim.x >> 6;
  uint32_t* lvl = reinterpret_cast<uint32_t*>(smem + PARK_BASE + NW * 8192);    // [16][8] words behind the stages (see k_enc_mlp_fwd)
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
  float* const stage = reinterpret_cast<float*>(smem + PARK_BASE + wave * 8192);       // [32][64] floats
  const int ntz = (nz + 63) / 64;
  const int64_t ntiles = (int64_t)nx * ny * ntz;
  for (int64_t tile = (int64_t)blockIdx.x * NW + wave; tile < ntiles; tile += (int64_t)gridDim.x * NW) {
    asm volatile("" ::: "memory");
    const int64_t col = tile / ntz;
    const int k0 = (int)(tile - col * ntz) * 64;
    const int k = k0 + lane;
    const int ix = (int)(col / ny), iy = (int)(col - (int64_t)ix * ny);
    const bool in_range = k < nz;
    float p[3] = {tx[ix], ty[iy], in_range ? tz[k] : 0.0f};
    const bool inside = in_range && (occ_bits == nullptr || occ_point_test(occ_bits, occ_n, p[0], p[1], p[2]));
    const uint64_t mask = __ballot(inside);
    if (mask == 0ull) {
      if (in_range) sdf[col * nz + k] = outside;
      continue;
    }
#pragma unroll
    for (int dd = 0; dd < 3; ++dd) p[dd] = fminf(fmaxf(p[dd], -1.0f), 1.0f);      // run_network_density clips (nerf_runner.py:1313)
#pragma unroll 1
    for (int l0 = 0; l0 < NOF_MAX_LEVELS; ++l0) {
      float2 a = make_float2(0.f, 0.f);
      if (l0 < n_levels && inside) {                    // (the level test is uniform; lanes outside the octree gather nothing)
        const HashLevel lv = level_at(l0);
        EncCell e = enc_prep(lv, p);
        float2 v[8];
        if (level_pairs(lv)) enc_load<true>(lv, table, e, v);
        else enc_load<false>(lv, table, e, v);
        enc_keep(e);
        a = enc_blend(e, v);
      }
      stage[(2 * l0) * 64 + lane] = a.x;
      stage[(2 * l0 + 1) * 64 + lane] = a.y;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      asm volatile("" ::: "memory");
      const int kt = k0 + 32 * t + j;                   // the voxel of lane (j, hi) in tile t
      const uint32_t mt = (uint32_t)(mask >> (32 * t));
      if (mt == 0u) {                                   // (uniform) nothing of this half is inside
        if (hi == 0 && kt < nz) sdf[col * nz + kt] = outside;
        continue;
      }
      float x[1][16];
#pragma unroll
      for (int r = 0; r < 16; ++r) x[0][r] = stage[(16 * hi + r) * 64 + 32 * t + j];
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
      if (hi == 0 && kt < nz) sdf[col * nz + kt] = ((mt >> j) & 1u) ? so[0][0] : outside;
    }
  }
}

// =====================================================================================================
// backward (forward recomputed): dfeat, dview, per-workgroup dW/db partials
// =====================================================================================================

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

// (the body of k_mlp_bwd_color as a device function: the stand-alone kernel and the merged launch k_mlp_bwd_both call it)
template <class P, int NS, int NC>
__device__ __forceinline__ void mlp_bwd_color_body(const NofMlpDesc& d, const char* __restrict__ image,
                                                   const typename P::elem* __restrict__ sig,
                                                   const float* __restrict__ view, int S,
                                                   const float4* __restrict__ draw, typename P::elem* dsig,
                                                   float* __restrict__ dview, float* __restrict__ partials, int64_t B,
                                                   const void* __restrict__ tile_list, char* smem) {
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
__global__ __launch_bounds__(64 * ColorWaves<NC>::value, 2) void k_mlp_bwd_color(NofMlpDesc d, const char* __restrict__ image,
                                                           const typename P::elem* __restrict__ sig,
                                                           const float* __restrict__ view, int S,
                                                           const float4* __restrict__ draw, typename P::elem* __restrict__ dsig,
                                                           float* __restrict__ dview, float* __restrict__ partials, int64_t B,
                                                           const void* __restrict__ tile_list) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  mlp_bwd_color_body<P, NS, NC>(d, image, sig, view, S, draw, dsig, dview, partials, B, tile_list, smem);
}

// (the body of k_mlp_bwd_sigma as a device function, like the colour half)
template <class P, int NS, int NC>
__device__ __forceinline__ void mlp_bwd_sigma_body(const NofMlpDesc& d, const char* __restrict__ image,
                                                   const float2* __restrict__ feat, int L,
                                                   const typename P::elem* dsig, float2* __restrict__ dfeat,
                                                   float* __restrict__ partials, int64_t B,
                                                   const void* __restrict__ tile_list,
                                                   const typename P::elem* __restrict__ featq, char* smem) {
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

template <class P, int NS, int NC>
__global__ __launch_bounds__(64 * SigmaWaves<NS>::value, 2) void k_mlp_bwd_sigma(NofMlpDesc d, const char* __restrict__ image,
                                                           const float2* __restrict__ feat, int L,
                                                           const typename P::elem* __restrict__ dsig, float2* __restrict__ dfeat,
                                                           float* __restrict__ partials, int64_t B,
                                                           const void* __restrict__ tile_list,
                                                           const typename P::elem* __restrict__ featq) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  mlp_bwd_sigma_body<P, NS, NC>(d, image, feat, L, dsig, dfeat, partials, B, tile_list, featq, smem);
}

// Both halves in ONE launch (round 6; shapes whose halves use the same number of waves per workgroup): a workgroup walks its share
// of the work list through the colour net, then -- new fragments in the same LDS -- the SAME tiles through the sigma net.  Every wave
// reads back the dsig rows it wrote itself, so no workgroup waits for another; the second half's fragments load while other
// workgroups still finish their first half, and the step loses a launch, a ramp and a drain.  Same code as the two kernels: same bits.
template <class P, int NS, int NC>
__global__ __launch_bounds__(64 * SigmaWaves<NS>::value, 2) void k_mlp_bwd_both(NofMlpDesc d, const char* __restrict__ image,
                                                           const typename P::elem* __restrict__ sig, const float* __restrict__ view,
                                                           int S, const float4* __restrict__ draw, typename P::elem* dsig,
                                                           float* __restrict__ dview, float* __restrict__ partials, int64_t B,
                                                           const void* __restrict__ tile_list, const float2* __restrict__ feat, int L,
                                                           float2* __restrict__ dfeat, const typename P::elem* __restrict__ featq) {
  static_assert(ColorWaves<NC>::value == SigmaWaves<NS>::value, "the halves must deal the work list to the same waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  mlp_bwd_color_body<P, NS, NC>(d, image, sig, view, S, draw, dsig, dview, partials, B, tile_list, smem);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");               // (this workgroup's dsig rows, before it reads them back)
  __syncthreads();                                                      // (and nobody still reads the colour fragments)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  asm volatile("" ::: "memory");
  mlp_bwd_sigma_body<P, NS, NC>(d, image, feat, L, dsig, dfeat, partials, B, tile_list, featq, smem);
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
                         float* partials, const void* tile_list, int64_t B, void* stream, bool one_launch = true);
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
// The same as two launches, colour half then sigma half, whatever the shape (nof_mlp_bwd_featq merges them where it can: the A/B
// and the bit-equality test of the merged launch).
extern "C" int nof_mlp_bwd_featq_two_launches(const NofMlpDesc* d, const void* packed, const void* featq, int32_t L, const float* view,
                                               int32_t S, const float* draw, const void* sigma_out, void* dsigma_ws, float* dfeat,
                                               float* dview, float* partials, const void* tile_list, int64_t B, void* stream) {
  NOF_ARG(featq && d && d->precision != 0 && sigma_out && dsigma_ws);
  return mlp_bwd_tiles(d, packed, nullptr, featq, L, view, S, draw, sigma_out, dsigma_ws, dfeat, dview, partials, tile_list, B, stream,
                       false);
}
static int mlp_bwd_tiles(const NofMlpDesc* d, const void* packed, const float* feat, const void* featq, int32_t L, const float* view,
                         int32_t S, const float* draw, const void* sigma_out, void* dsigma_ws, float* dfeat, float* dview,
                         float* partials, const void* tile_list, int64_t B, void* stream, bool one_launch) {
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
  if constexpr (ColorWaves<NC_>::value == SigmaWaves<NS_>::value) {                                       \
    if (one_launch) {                                                                                     \
      auto kb = k_mlp_bwd_both<P, NS_, NC_>;                                                              \
      const size_t shm_b = shm_c > shm_s ? shm_c : shm_s;                                                 \
      if (int e = set_smem(kb, shm_b)) return e;                                                          \
      hipLaunchKernelGGL(kb, dim3(blocks_c), dim3(64 * (unsigned)wc), shm_b, (hipStream_t)stream, *d, (const char*)packed, \
                         (const typename P::elem*)sigma_out, view, (int)S, (const float4*)draw,           \
                         (typename P::elem*)dsigma_ws, dview, partials, B, tile_list, (const float2*)feat, (int)L, \
                         (float2*)dfeat, (const typename P::elem*)featq);                                 \
      NOF_LAUNCH_OK();                                                                                    \
      return 0;                                                                                           \
    }                                                                                                     \
  }                                                                                                       \
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
  const size_t img = ((is_split(d->precision) ? 2 : 1) * (size_t)n_pairs(*d, nl) * 16 * 64 * elem_size(d->precision) +
                      (size_t)n_oblk(*d, nl) * 32 * 4 + 15) & ~(size_t)15;
  int waves = (int)((160 * 1024 - img - 512) / 8192);                   // one workgroup per CU: the sigma image + 8 KB of stage per wave + the level table
  if (waves > NOF_ENC_WAVES) waves = NOF_ENC_WAVES;
  NOF_ARG(waves >= 4);
  const size_t shm = img + (size_t)waves * 8192 + 512;
  const int64_t ntiles = (int64_t)nx * ny * ((nz + 63) / 64);
  const int64_t want = nof_div_up(ntiles, waves), cap = (int64_t)nof_cu_count();
  const unsigned blocks = (unsigned)(want < cap ? want : cap);
#define LAUNCH_GRID(P, NS_, NC_, SPLIT_)                                                                  \
  {                                                                                                       \
    auto kern = k_sdf_grid<P, NS_, NC_, SPLIT_>;                                                          \
    if (int e = set_smem(kern, shm)) return e;                                                            \
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(64 * waves), shm, (hipStream_t)stream, *d, (const char*)packed, *g, \
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

