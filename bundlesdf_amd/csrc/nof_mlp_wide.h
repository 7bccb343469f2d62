// Wide / deep NeRFSmall shapes (hidden 128 and/or 4 layers per network: BASELINE.json cfg5's "MLP 4x128", nerf_helpers.py:243-294
// is parameterised in hidden_dim / num_layers) on the gfx950 matrix cores.  Included at the end of nof_mlp.hip: it reuses
// that file's operand layouts, fragment image (k_mlp_pack), dense_o1 / bwd_data / transpose32 primitives and host helpers.
//
// Why a second set of kernels: the register-resident design of nof_mlp.hip keeps every weight fragment of both orientations
// in LDS and every dW accumulator in registers.  A 128x128 layer is 32 KB of fragments per orientation and 256 accumulator
// registers per wave; 4+4 such layers are 168 KB per orientation.  So here
//   * each network (sigma, colour) and each direction is its own kernel with ONE orientation of ONE network resident in LDS
//     (<= 88 KB for 4x128, 16-bit operands), shared by a 768-thread workgroup (12 waves, 3 per SIMD, <= 168 VGPRs: these
//     kernels wait on every staged row they load, so occupancy is what hides the latency: 512 threads were 1.6x slower);
//   * the forward kernels store the hidden activations (post-ReLU, operand precision, [layer][B][H] sample-major) instead of the
//     backward recomputing them, and the backward-data kernels store the pre-activation gradients the same way;
//   * the weight gradients are a separate split-K pass per layer (k_wide_dw): a wave owns ONE 32-neuron output block (16 * QN
//     accumulator registers), transposes its 32-sample tile of gradients / inputs on the matrix core (transpose32) and
//     accumulates across its share of the batch; per-wave partial rows are summed by nof_reduce_partials like the narrow path.
// Algorithmic HBM traffic of the staging per hidden layer and sample: H*2 B written + read twice for the activations, H*2 B
// written + read for the gradients (~1.3 KB at H = 128): this path is HBM-bound by construction (DESIGN.md "wide network").
// 16-bit operand types only (fp32 fragments of a 128-wide network do not fit LDS); precisions 3 / 4 run as 2 / 1 here (no
// operand split: the residual fragments would double the LDS image).
#pragma once

// ---- staging buffers: [tile][block p][half][lane][8 elements] -- FRAGMENT order -------------------------------------------------
// The buffers are private to these kernels and every kernel touches them with the same lane <-> (sample j, hi) mapping, so a
// 32-sample tile's block p is laid out exactly as the wave holds it: the 8 values lane (hi, j) keeps in registers r = 8 half ..
// 8 half + 7 of block p (neurons 32p + nloc(hi, r)) sit at fragment (p, half), lane hi * 32 + j.  One store / load instruction of a
// wave is then ONE contiguous kilobyte (sample-major rows made it 64 separate 16-byte pieces of 64 different 256-byte rows: the
// forward kernels ran at 2.5 TB/s of write traffic).  `tile_row` = the address of the lane's first fragment of its tile; the
// stride between a tile's fragments is 64 lanes.  A layer's buffer holds ceil(B / 32) * 32 * H elements.
// Addressing: ONE 32-bit byte offset per lane (the lane's first fragment of its tile) + a wave-uniform layer base + an immediate
// per fragment -- the form global_load / global_store take directly (saddr + voffset + imm).  64-bit per-lane pointers per layer
// cost these kernels, which sit at their register cap, 15-20 VGPRs and sent the sigma forward to scratch.  A layer's buffer stays
// below 4 GiB (checked on the host).
struct TileRow {
  char* base;                                                          // wave-uniform: the layer's buffer
  uint32_t voff;                                                       // this lane's byte offset of fragment (0, 0) of its tile
};
template <class P>
__device__ __forceinline__ TileRow tile_row(const typename P::elem* __restrict__ base, int64_t b, int hi, int hb) {
  TileRow t;
  t.base = reinterpret_cast<char*>(const_cast<typename P::elem*>(base));
  t.voff = (uint32_t)(((b >> 5) * (int64_t)(hb * 2 * 64) + (hi * 32 + (int)(b & 31))) * 16);
  return t;
}
template <class P>
__device__ __forceinline__ void store_blk(const TileRow& row, int p, const float (&v)[16]) {
  *reinterpret_cast<typename P::frag*>(row.base + row.voff + (2 * p) * 1024) = P::pack(&v[0]);
  *reinterpret_cast<typename P::frag*>(row.base + row.voff + (2 * p + 1) * 1024) = P::pack(&v[8]);
}
template <class P>
__device__ __forceinline__ void load_blk(const TileRow& row, int p, bool ok, float (&v)[16]) {
  typename P::frag a, b;
#pragma unroll
  for (int k = 0; k < 8; ++k) { a[k] = (typename P::elem)0.0f; b[k] = (typename P::elem)0.0f; }
  if (ok) {
    a = *reinterpret_cast<const typename P::frag*>(row.base + row.voff + (2 * p) * 1024);
    b = *reinterpret_cast<const typename P::frag*>(row.base + row.voff + (2 * p + 1) * 1024);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { v[k] = (float)a[k]; v[8 + k] = (float)b[k]; }
}

// ReLU in place + the 16 PN derivative bits of the lane (relu_mask of nof_mlp.hip, two blocks per 32-bit word): the backward
// kernels read these 8 bytes per lane and hidden layer instead of the whole stored activation row (256 bytes per sample at H = 128)
// just to know which units were on.
template <int HB>
__device__ __forceinline__ uint2 relu_bits(float (&h)[HB][16]) {
  static_assert(HB == 2 || HB == 4, "hidden width 64 or 128");
  uint32_t off[2] = {0u, 0u};
#pragma unroll
  for (int p = 0; p < HB; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) {                                     // (plain indexing: a reinterpret_cast of the array sent it to scratch)
      const int bits = __float_as_int(h[p][r]);
      off[p >> 1] = __builtin_amdgcn_alignbit(off[p >> 1], (uint32_t)bits, 31);
      h[p][r] = __int_as_float(bits > 0 ? bits : 0);
    }
  return make_uint2(~off[0], HB == 4 ? ~off[1] : 0u);
}
// element (p, r) of word p >> 1 sits at bit 31 - (16 (p & 1) + r), 1 = the unit was on
template <int HB>
__device__ __forceinline__ void apply_bits(float (&g)[HB][16], uint2 m) {
#pragma unroll
  for (int p = 0; p < HB; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const uint32_t keep = (uint32_t)__builtin_amdgcn_sbfe((int)(p < 2 ? m.x : m.y), 31 - (16 * (p & 1) + r), 1);
      g[p][r] = __uint_as_float(__float_as_uint(g[p][r]) & keep);
    }
}
// [hidden layer][tile][lane] uint2: uniform layer base + 32-bit lane offset
__device__ __forceinline__ uint2* bits_at(uint2* __restrict__ base, int64_t ntiles, int layer, int64_t tile, int lane) {
  char* lb = reinterpret_cast<char*>(base + (int64_t)layer * ntiles * 64);
  return reinterpret_cast<uint2*>(lb + (uint32_t)((tile * 64 + lane) * 8));
}
__device__ __forceinline__ const uint2* bits_at(const uint2* __restrict__ base, int64_t ntiles, int layer, int64_t tile, int lane) {
  const char* lb = reinterpret_cast<const char*>(base + (int64_t)layer * ntiles * 64);
  return reinterpret_cast<const uint2*>(lb + (uint32_t)((tile * 64 + lane) * 8));
}

template <int PN>
__device__ __forceinline__ void relu_inplace(float (&h)[PN][16]) {
#pragma unroll
  for (int p = 0; p < PN; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) {                                     // signed-integer max on the bits: one instruction (see relu_mask)
      const int bits = __float_as_int(h[p][r]);
      h[p][r] = __int_as_float(bits > 0 ? bits : 0);
    }
}

#define WPAIR ((int)(16 * 64 * sizeof(typename P::elem)))

// =====================================================================================================
// forward, sigma net: features -> hidden layers (stored) -> head: sdf -> raw[b].w (or sdf[b]), sig[b] = 16 head outputs
// =====================================================================================================
// threads per workgroup of the forward kernels: 768 (3 waves per SIMD, 168 registers) where that fits without scratch (hidden 64);
// hidden 128 needs ~190 (two 64-register activation sets + the operand and weight fragments of a chain): 512 threads
template <int HB> struct WideFwdThreads { static constexpr int value = HB >= 4 ? 512 : 768; };

template <class P, int HB>
__global__ __launch_bounds__(WideFwdThreads<HB>::value) void k_wide_fwd_sigma(NofMlpDesc d, const char* __restrict__ image,
                                                         const float2* __restrict__ feat, int L,
                                                         typename P::elem* __restrict__ hid, int64_t hid_stride,
                                                         float* __restrict__ out, int out_stride, int out_off,
                                                         typename P::elem* __restrict__ sig, uint2* __restrict__ bits, int64_t B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NS = d.n_sigma, NL = d.n_sigma + d.n_color;
  const int bias_base = pair_base(d, NS) * WPAIR;
  copy16(smem, image, (size_t)bias_base);
  copy16(smem + bias_base, image + 2 * (size_t)pair_base(d, NL) * WPAIR, (size_t)oblk_base(d, NS) * 128);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int hi = lane >> 5, j = lane & 31;
  const int H = 32 * HB;
  const int64_t ntiles = (B + 31) / 32, tstride = (int64_t)gridDim.x * nw;
  // the NEXT tile's features are requested a whole tile ahead where the 16 registers are to be had (hidden 64); at hidden 128 the
  // kernel sits at its 168-register cap (768 threads) and the look-ahead spilled 42 of them
  constexpr bool AHEAD = true;
  float xn[1][16];
  if constexpr (AHEAD) load_feat_o1(feat, L, B, ((int64_t)blockIdx.x * nw + wave) * 32 + j, hi, xn);
  for (int64_t tile = (int64_t)blockIdx.x * nw + wave; tile < ntiles; tile += tstride) {
    asm volatile("" ::: "memory");
    const int64_t b = tile * 32 + j;
    const bool ok = b < B;
    float x[1][16], h[HB][16], so[1][16];
    if constexpr (AHEAD) {
#pragma unroll
      for (int r = 0; r < 16; ++r) x[0][r] = xn[0][r];
      pin16(x[0]);
      load_feat_o1(feat, L, B, (tile + tstride) * 32 + j, hi, xn);
    } else {
      load_feat_o1(feat, L, B, b, hi, x);
    }
    dense_o1<P, 1, HB>(smem, 0, bias_base, x, h, lane);
    uint2 mb = relu_bits<HB>(h);
    int foff = HB * WPAIR, boff = bias_base + HB * 128;
    for (int l = 1; l < NS - 1; ++l) {
      if (hid != nullptr && ok) {
        *bits_at(bits, ntiles, l - 1, tile, lane) = mb;
#pragma unroll
        for (int p = 0; p < HB; ++p) store_blk<P>(tile_row<P>(hid + (int64_t)(l - 1) * hid_stride, b, hi, HB), p, h[p]);
      }
      float h2[HB][16];
      dense_o1<P, HB, HB>(smem, foff, boff, h, h2, lane);
      mb = relu_bits<HB>(h2);
#pragma unroll
      for (int p = 0; p < HB; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[p][r] = h2[p][r];
      foff += HB * HB * WPAIR;
      boff += HB * 128;
    }
    if (hid != nullptr && ok) {
      *bits_at(bits, ntiles, NS - 2, tile, lane) = mb;
#pragma unroll
      for (int p = 0; p < HB; ++p) store_blk<P>(tile_row<P>(hid + (int64_t)(NS - 2) * hid_stride, b, hi, HB), p, h[p]);
    }
    dense_o1<P, HB, 1>(smem, foff, boff, h, so, lane);
    if (sig != nullptr) store_sig_o1<P>(sig, B, b, hi, so[0]);
    if (hi == 0 && ok) out[b * out_stride + out_off] = so[0][0];
  }
}

// =====================================================================================================
// forward, colour net: [sig | view] -> hidden layers (stored) -> rgb_raw -> raw[b].xyz
// =====================================================================================================
template <class P, int HB>
__global__ __launch_bounds__(WideFwdThreads<HB>::value) void k_wide_fwd_color(NofMlpDesc d, const char* __restrict__ image,
                                                         const typename P::elem* __restrict__ sig,
                                                         const float* __restrict__ view, int S,
                                                         typename P::elem* __restrict__ hid, int64_t hid_stride,
                                                         float* __restrict__ raw, uint2* __restrict__ bits, int64_t B) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NS = d.n_sigma, NC = d.n_color, NL = NS + NC;
  const int PA = pair_base(d, NS), PB = pair_base(d, NL), OA = oblk_base(d, NS), OB = oblk_base(d, NL);
  const int bias_base = (PB - PA) * WPAIR;
  copy16(smem, image + (size_t)PA * WPAIR, (size_t)bias_base);
  copy16(smem + bias_base, image + 2 * (size_t)PB * WPAIR + OA * 128, (size_t)(OB - OA) * 128);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int hi = lane >> 5, j = lane & 31;
  const int H = 32 * HB;
  const int64_t ntiles = (B + 31) / 32;
  for (int64_t tile = (int64_t)blockIdx.x * nw + wave; tile < ntiles; tile += (int64_t)gridDim.x * nw) {
    asm volatile("" ::: "memory");
    const int64_t b = tile * 32 + j;
    const bool ok = b < B;
    float cin[2][16], h[HB][16], co[1][16];
    load_sig_o1<P>(sig, B, b, hi, cin[0]);            // (requested a tile ahead: 4-10 % slower at cfg5, measured twice)
    load_view_o1(view, S, B, b, hi, cin[1]);
    dense_o1<P, 2, HB>(smem, 0, bias_base, cin, h, lane);
    uint2 mb = relu_bits<HB>(h);
    int foff = 2 * HB * WPAIR, boff = bias_base + HB * 128;
    for (int l = 1; l < NC - 1; ++l) {
      if (hid != nullptr && ok) {
        *bits_at(bits, ntiles, NS - 1 + l - 1, tile, lane) = mb;
#pragma unroll
        for (int p = 0; p < HB; ++p) store_blk<P>(tile_row<P>(hid + (int64_t)(NS - 1 + l - 1) * hid_stride, b, hi, HB), p, h[p]);
      }
      float h2[HB][16];
      dense_o1<P, HB, HB>(smem, foff, boff, h, h2, lane);
      mb = relu_bits<HB>(h2);
#pragma unroll
      for (int p = 0; p < HB; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) h[p][r] = h2[p][r];
      foff += HB * HB * WPAIR;
      boff += HB * 128;
    }
    if (hid != nullptr && ok) {
      *bits_at(bits, ntiles, NS - 1 + NC - 2, tile, lane) = mb;
#pragma unroll
      for (int p = 0; p < HB; ++p) store_blk<P>(tile_row<P>(hid + (int64_t)(NS - 1 + NC - 2) * hid_stride, b, hi, HB), p, h[p]);
    }
    dense_o1<P, HB, 1>(smem, foff, boff, h, co, lane);
    if (hi == 0 && ok) { raw[b * 4] = co[0][0]; raw[b * 4 + 1] = co[0][1]; raw[b * 4 + 2] = co[0][2]; }
  }
}

// =====================================================================================================
// backward (data path), colour net: draw -> pre-activation gradients of every colour layer (stored), dsig, dview
// gbuf[l] rows are [B][H]; a head's gradient occupies block 0 of its row.
// =====================================================================================================
template <class P, int HB>
__global__ __launch_bounds__(768) void k_wide_bwd_color(NofMlpDesc d, const char* __restrict__ image,
                                                         const typename P::elem* __restrict__ hid, int64_t hid_stride,
                                                         int S, const float4* __restrict__ draw,
                                                         typename P::elem* __restrict__ gbuf, int64_t g_stride,
                                                         typename P::elem* __restrict__ dsig, float* __restrict__ dview,
                                                         uint8_t* __restrict__ zflag, int64_t B, const void* __restrict__ tile_list,
                                                         const uint2* __restrict__ bits) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NS = d.n_sigma, NC = d.n_color, NL = NS + NC;
  const int PA = pair_base(d, NS), PB = pair_base(d, NL);
  copy16(smem, image + (size_t)(PB + PA) * WPAIR, (size_t)(PB - PA) * WPAIR);       // the bw orientation of the colour layers
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int hi = lane >> 5, j = lane & 31;
  const int H = 32 * HB;
  Ident<P> I;
  I.init(lane);
  const float gscale = d.grad_scale > 0.0f ? d.grad_scale : 1.0f, gunscale = 1.0f / gscale;
  const int64_t ntiles = (B + 31) / 32;
  const TileWork work(tile_list, ntiles);              // the backward's work list (NofTileList) or every tile of the batch
  for (int64_t wi = (int64_t)blockIdx.x * nw + __builtin_amdgcn_readfirstlane(wave); wi < work.n; wi += (int64_t)gridDim.x * nw) {
    asm volatile("" ::: "memory");
    const int64_t tile = work.at(wi);
    const int64_t t0 = tile * 32, b = t0 + j;
    const bool ok = b < B;
    float gh[1][16], g[HB][16];
    float dsdf1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) gh[0][r] = 0.0f;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (hi == 0 && ok) t = draw[b];
    // a tile whose 32 loss gradients are all exactly zero contributes nothing to any gradient: flagged for the
    // sigma kernel and the weight-gradient passes, nothing stored, nothing computed
    // (with a work list every listed tile has a non-zero row and zflag is not used)
    if (tile_list == nullptr) {
      const bool skip = __builtin_amdgcn_ballot_w64(t.x != 0.0f || t.y != 0.0f || t.z != 0.0f || t.w != 0.0f) == 0ull;
      if (lane == 0) zflag[tile] = skip ? 1 : 0;
      if (skip) continue;
    }
    gh[0][0] = t.x * gscale; gh[0][1] = t.y * gscale; gh[0][2] = t.z * gscale;
    dsdf1 = t.w * gscale;
    if (ok) store_blk<P>(tile_row<P>(gbuf + (int64_t)(NL - 1) * g_stride, b, hi, HB), 0, gh[0]);
    // head -> last hidden colour layer
    int woff = (pair_base(d, NL - 1) - PA) * WPAIR;
    {
#pragma unroll
      for (int q = 0; q < HB; ++q) bwd_data<P, 1>(smem, woff, q, gh, g[q], lane);
      apply_bits<HB>(g, *bits_at(bits, ntiles, NS - 1 + NC - 2, tile, lane));
    }
    for (int l = NL - 2; l > NS; --l) {                               // hidden colour layers above layer 0
      if (ok) {
#pragma unroll
        for (int p = 0; p < HB; ++p) store_blk<P>(tile_row<P>(gbuf + (int64_t)l * g_stride, b, hi, HB), p, g[p]);
      }
      woff = (pair_base(d, l) - PA) * WPAIR;
      float g2[HB][16];
#pragma unroll
      for (int q = 0; q < HB; ++q) bwd_data<P, HB>(smem, woff, q, g, g2[q], lane);
      apply_bits<HB>(g2, *bits_at(bits, ntiles, NS - 1 + (l - 1 - NS), tile, lane));
#pragma unroll
      for (int p = 0; p < HB; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[p][r] = g2[p][r];
    }
    // colour layer 0: inputs [sigma-out block | view block]
    if (ok) {
#pragma unroll
      for (int p = 0; p < HB; ++p) store_blk<P>(tile_row<P>(gbuf + (int64_t)NS * g_stride, b, hi, HB), p, g[p]);
    }
    float ds1[16], dv1[16], dv2[16];
    bwd_data<P, HB>(smem, 0, 0, g, ds1, lane);
    bwd_data<P, HB>(smem, 0, 1, g, dv1, lane);
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
    if (hi == 0) ds1[0] += dsdf1;
    store_sig_o1<P>(dsig, B, b, hi, ds1);
  }
}

// =====================================================================================================
// backward (data path), sigma net: dsig -> pre-activation gradients of every sigma layer (stored), dfeat
// =====================================================================================================
template <class P, int HB>
__global__ __launch_bounds__(768) void k_wide_bwd_sigma(NofMlpDesc d, const char* __restrict__ image,
                                                         const typename P::elem* __restrict__ hid, int64_t hid_stride,
                                                         const typename P::elem* __restrict__ dsig,
                                                         typename P::elem* __restrict__ gbuf, int64_t g_stride,
                                                         float2* __restrict__ dfeat, int L,
                                                         const uint8_t* __restrict__ zflag, int64_t B,
                                                         const void* __restrict__ tile_list, const uint2* __restrict__ bits) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int NS = d.n_sigma, NL = d.n_sigma + d.n_color;
  const int PB = pair_base(d, NL), PS = pair_base(d, NS);
  copy16(smem, image + (size_t)PB * WPAIR, (size_t)PS * WPAIR);                       // the bw orientation of the sigma layers
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  const int hi = lane >> 5, j = lane & 31;
  const int H = 32 * HB;
  const float gunscale = d.grad_scale > 0.0f ? 1.0f / d.grad_scale : 1.0f;
  const int64_t ntiles = (B + 31) / 32, tstride = (int64_t)gridDim.x * nw;
  const TileWork work(tile_list, ntiles);
  const int64_t w0 = (int64_t)blockIdx.x * nw + __builtin_amdgcn_readfirstlane(wave);
  int64_t tile_n = work.at(w0);
  typename P::frag dsn = load_sig_raw<P>(dsig, B, tile_n * 32 + j, hi);   // a tile ahead
  for (int64_t wi = w0; wi < work.n; wi += tstride) {
    asm volatile("" ::: "memory");
    const int64_t tile = tile_n;
    tile_n = work.at(wi + tstride);
    const int64_t b = tile * 32 + j;
    const bool ok = b < B;
    float gh[1][16], g[HB][16];
    sig_to_o1<P>(dsn, gh[0]);
    pin16(gh[0]);
    dsn = load_sig_raw<P>(dsig, B, tile_n * 32 + j, hi);
    float df1[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) df1[r] = 0.0f;
    if (tile_list != nullptr || zflag[tile] == 0) {                       // (flagged by k_wide_bwd_color: dfeat = 0, nothing else)
    if (ok) store_blk<P>(tile_row<P>(gbuf + (int64_t)(NS - 1) * g_stride, b, hi, HB), 0, gh[0]);
    int woff = pair_base(d, NS - 1) * WPAIR;
    {
#pragma unroll
      for (int q = 0; q < HB; ++q) bwd_data<P, 1>(smem, woff, q, gh, g[q], lane);
      apply_bits<HB>(g, *bits_at(bits, ntiles, NS - 2, tile, lane));
    }
    for (int l = NS - 2; l >= 1; --l) {
      if (ok) {
#pragma unroll
        for (int p = 0; p < HB; ++p) store_blk<P>(tile_row<P>(gbuf + (int64_t)l * g_stride, b, hi, HB), p, g[p]);
      }
      woff = pair_base(d, l) * WPAIR;
      float g2[HB][16];
#pragma unroll
      for (int q = 0; q < HB; ++q) bwd_data<P, HB>(smem, woff, q, g, g2[q], lane);
      apply_bits<HB>(g2, *bits_at(bits, ntiles, l - 1, tile, lane));
#pragma unroll
      for (int p = 0; p < HB; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) g[p][r] = g2[p][r];
    }
    if (ok) {
#pragma unroll
      for (int p = 0; p < HB; ++p) store_blk<P>(tile_row<P>(gbuf, b, hi, HB), p, g[p]);
    }
    bwd_data<P, HB>(smem, 0, 0, g, df1, lane);
    }
    store_dfeat_o1(dfeat, L, B, b, hi, df1, gunscale);
  }
}

// =====================================================================================================
// weight gradients of ONE layer: dW[32p + n][k] = sum_b G[b][32p + n] X[b][k], db likewise.  The four waves of a workgroup
// split into PN output blocks x NST = 4/PN tile streams: wave w owns output block p = w % PN of the tiles of partial row
// blockIdx.x * NST + w / PN.  The PN waves of a stream need the same QN input blocks of the same tile, so each of them loads
// and transposes (on the matrix core) only the blocks q = p, p + PN, ... and they exchange the transposed operand fragments
// through LDS (double-buffered, one workgroup barrier per tile): 2 + 2 QN/PN sixteen-byte loads per lane and tile instead of
// 2 + 2 QN.  `partials` is [rows][n_params]: every (layer, block) writes its own entries of every row, so after all layers
// each row is completely defined.
// KIND: 0 = hash features (QN = 1), 1 = a stored hidden activation row (QN = H/32), 2 = colour layer 0's [sig | view] (QN = 2)
// =====================================================================================================
template <class P, int QN, int KIND, int PN>
__global__ __launch_bounds__(256) void k_wide_dw(NofMlpDesc d, int l, const typename P::elem* __restrict__ grow,
                                                  int H, const float2* __restrict__ feat, int L,
                                                  const typename P::elem* __restrict__ xrow,
                                                  const typename P::elem* __restrict__ sig, const float* __restrict__ view,
                                                  int S, float* __restrict__ partials, int rows,
                                                  const uint8_t* __restrict__ zflag, int64_t B,
                                                  const void* __restrict__ tile_list) {
  constexpr int KR = P::KR, NSTEP = 16 / KR, NST = 4 / PN, MAXQ = (QN + PN - 1) / PN;
  __shared__ typename P::frag xs[2][NST][QN][NSTEP][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hi = lane >> 5, j = lane & 31;
  const int p = wave % PN, stream = wave / PN;
  const int row = blockIdx.x * NST + stream;
  Ident<P> I;
  I.init(lane);
  const float gunscale = d.grad_scale > 0.0f ? 1.0f / d.grad_scale : 1.0f;
  f32x16 acc[QN];
#pragma unroll
  for (int q = 0; q < QN; ++q)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[q][r] = 0.0f;
  float db = 0.0f;
  const int64_t ntiles = (B + 31) / 32;
  const TileWork work(tile_list, ntiles);
  const int n_it = (int)((work.n + rows - 1) / rows);                  // the same trip count for every wave (barrier inside)
  // The gradient block and this wave's share of the input blocks of the tiles TWO iterations ahead are in flight while a tile is
  // computed, kept as the raw 16-byte fragments they are stored as (one tile ahead, held as 32 floats, left ~4 KB per wave in
  // flight: 3.6 TB/s of a pass that does nothing but stream its two operand rows); the stored fragments are the A operands of the
  // transposing MFMA as they are.
  typedef typename P::frag frag;
  struct Pre {
    frag g[2];
    frag xf[KIND == 1 ? MAXQ : 1][2];                                    // KIND 1: stored activation blocks
    float xr[KIND == 1 ? 1 : MAXQ][16];                                  // KIND 0 / 2: hash features, [sig | view]
    bool z;                                                              // flagged all-zero (or past the end)
  };
  auto zero_frag = [] { frag f;
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] = (typename P::elem)0.0f;
    return f; };
  auto fetch = [&](int64_t wi, Pre& t) {
    const int64_t tile = work.at(wi);
    const int64_t b = tile * 32 + j;
    t.z = tile >= ntiles || (tile_list == nullptr && zflag[tile] != 0);
    const bool ok = !t.z && b < B;
    const TileRow gr = tile_row<P>(grow, b, hi, H / 32);
    t.g[0] = t.g[1] = zero_frag();
    if (ok) {
      t.g[0] = *reinterpret_cast<const frag*>(gr.base + gr.voff + (2 * p) * 1024);
      t.g[1] = *reinterpret_cast<const frag*>(gr.base + gr.voff + (2 * p + 1) * 1024);
    }
#pragma unroll
    for (int m = 0; m < MAXQ; ++m) {
      const int q = p + m * PN;
      if constexpr (KIND == 1) {
        t.xf[m][0] = t.xf[m][1] = zero_frag();
        if (q < QN && ok) {
          const TileRow xr = tile_row<P>(xrow, b, hi, H / 32);
          t.xf[m][0] = *reinterpret_cast<const frag*>(xr.base + xr.voff + (2 * q) * 1024);
          t.xf[m][1] = *reinterpret_cast<const frag*>(xr.base + xr.voff + (2 * q + 1) * 1024);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) t.xr[m][r] = 0.0f;
        if (q < QN) {
          if constexpr (KIND == 0) {
            float xf[1][16];
            load_feat_o1(feat, L, ok ? B : 0, b, hi, xf);
#pragma unroll
            for (int r = 0; r < 16; ++r) t.xr[m][r] = xf[0][r];
          } else {
            if (q == 0) load_sig_o1<P>(sig, ok ? B : 0, b, hi, t.xr[m]);
            else load_view_o1(view, S, ok ? B : 0, b, hi, t.xr[m]);
          }
        }
      }
    }
  };
  // block held sample-per-lane as two operand fragments -> slot-per-lane (see transpose32)
  auto transpose_raw = [&](frag a, frag b, float (&y)[16]) {
    static_assert(NSTEP == 2, "16-bit operand types");
    f32x16 t;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = 0.0f;
    t = P::mma(a, I.f[0], t);
    t = P::mma(b, I.f[1], t);
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r] = t[r];
  };
  auto step = [&](const Pre& t, int it) {
    const int buf = it & 1;
    if (__builtin_amdgcn_readfirstlane(t.z ? 1 : 0) != 0) { __syncthreads(); return; }   // adds nothing: only the barrier is kept
#pragma unroll
    for (int m = 0; m < MAXQ; ++m) {
      const int q = p + m * PN;
      if (q < QN) {
        float x2[16];
        if constexpr (KIND == 1) transpose_raw(t.xf[m][0], t.xf[m][1], x2);
        else transpose32<P>(I, t.xr[m], x2);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) xs[buf][stream][q][s][lane] = P::pack(&x2[KR * s]);
      }
    }
    float g2[16];
    transpose_raw(t.g[0], t.g[1], g2);
#pragma unroll
    for (int r = 0; r < 16; ++r) db += g2[r];
    frag ga[NSTEP];
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) ga[s] = P::pack(&g2[KR * s]);
    __syncthreads();
#pragma unroll
    for (int q = 0; q < QN; ++q)
#pragma unroll
      for (int s = 0; s < NSTEP; ++s) acc[q] = P::mma(ga[s], xs[buf][stream][q][s][lane], acc[q]);
  };
  Pre ta, tb;
  fetch(row, ta);
  fetch((int64_t)row + rows, tb);
  for (int it = 0; it < n_it; it += 2) {                                 // (n_it is the same for every wave: the barriers match)
    step(ta, it);
    fetch((int64_t)row + (int64_t)(it + 2) * rows, ta);
    if (it + 1 < n_it) {
      step(tb, it + 1);
      fetch((int64_t)row + (int64_t)(it + 3) * rows, tb);
    }
  }
  // flush: lane j = input slot (hi_j, r_j) of block q, register r = neuron 32p + nloc(hi, r)
  const int hi_j = (j >> 2) & 1, r_j = (j & 3) + 4 * (j >> 3);
  float* __restrict__ dst = partials + (size_t)row * d.n_params;
  const int in_dim = d.in_dim[l], out_dim = d.out_dim[l];
#pragma unroll
  for (int q = 0; q < QN; ++q) {
    const int col = inmap(d, l, q, hi_j, r_j);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int orow = 32 * p + nloc(hi, r);
      if (col >= 0 && orow < out_dim) dst[d.w_off[l] + orow * in_dim + col] = acc[q][r] * gunscale;
    }
  }
  db += __shfl_xor(db, 32, 64);
  if (hi == 0 && 32 * p + j < out_dim) dst[d.b_off[l] + 32 * p + j] = db * gunscale;
}

// =====================================================================================================
// host side
// =====================================================================================================
static int check_wide(const NofMlpDesc* d) {
  if (int e = check_desc(d)) return e;
  if (d->precision == 0)
    return nof_set_error(-1, "mlp (wide path, hidden %d depths %d,%d): 16-bit operand types only (fp32 fragments do not fit LDS)",
                         d->hidden, d->n_sigma, d->n_color);
  return 0;
}
static int64_t wide_hid_layers(const NofMlpDesc* d) { return (d->n_sigma - 1) + (d->n_color - 1); }
static const int kWideRows = 768;                                         // wave-rows of `partials`: the weight-gradient pass runs 3 waves per SIMD (164 registers), all resident in one round

// workspace layout (bytes, 256-aligned; Bt = B rounded up to whole tiles): [hid : n_hid * Bt * H elems][gbuf : NL * Bt * H elems][sig : B * 16 elems][dsig : B * 16 elems][zflag : B/32 bytes][bits : n_hid * Bt/32 * 64 * 8 bytes]
struct WideWs { char *hid, *gbuf, *sig, *dsig; uint8_t* zflag; uint2* bits; int64_t total; };
static WideWs wide_ws(const NofMlpDesc* d, void* base, int64_t B) {
  auto up = [](int64_t x) { return (x + 255) / 256 * 256; };
  const int64_t row = (int64_t)d->hidden * 2, nl = d->n_sigma + d->n_color;
  const int64_t Bt = (B + 31) / 32 * 32;                                // staging buffers hold whole 32-sample tiles
  WideWs w;
  int64_t off = 0;
  w.hid = (char*)base + off; off += up(wide_hid_layers(d) * Bt * row);
  w.gbuf = (char*)base + off; off += up(nl * Bt * row);
  w.sig = (char*)base + off; off += up(B * 32);
  w.dsig = (char*)base + off; off += up(B * 32);
  w.zflag = (uint8_t*)base + off; off += up((B + 31) / 32);             // one byte per 32-sample tile: 1 = every loss gradient is exactly zero
  w.bits = (uint2*)((char*)base + off); off += up(wide_hid_layers(d) * (Bt / 32) * 64 * 8);   // ReLU derivative bits: 8 B per lane, tile and hidden layer
  w.total = off;
  return w;
}
extern "C" int64_t nof_mlp_wide_workspace_bytes(const NofMlpDesc* d, int64_t B) {
  if (check_wide(d) || B < 0) return -1;
  return wide_ws(d, nullptr, B).total;
}
extern "C" int nof_mlp_wide_partial_rows(void) { return kWideRows; }

template <class P, int HB>
static int wide_fwd_launch(const NofMlpDesc* d, const void* packed, const float* feat, int32_t L, const float* view, int32_t S,
                           float* out, int out_stride, int out_off, const WideWs* ws, bool store_hidden, bool sdf_only,
                           int64_t B, hipStream_t st) {
  const int ns = d->n_sigma, nl = d->n_sigma + d->n_color;
  const size_t pair_bytes = 16 * 64 * 2;
  const size_t shm_s = (size_t)pair_base(*d, ns) * pair_bytes + (size_t)oblk_base(*d, ns) * 128;
  const size_t shm_c = (size_t)(pair_base(*d, nl) - pair_base(*d, ns)) * pair_bytes + (size_t)(oblk_base(*d, nl) - oblk_base(*d, ns)) * 128;
  const int64_t ntiles = (B + 31) / 32;
  constexpr int NT = WideFwdThreads<HB>::value;
  const unsigned blocks = (unsigned)(nof_div_up(ntiles, NT / 64) < (int64_t)nof_cu_count() ? nof_div_up(ntiles, NT / 64) : nof_cu_count());
  typedef typename P::elem elem;
  const int64_t hs = (B + 31) / 32 * 32 * (int64_t)d->hidden;
  auto ks = k_wide_fwd_sigma<P, HB>;
  if (int e = set_smem(ks, shm_s)) return e;
  hipLaunchKernelGGL(ks, dim3(blocks), dim3(NT), shm_s, st, *d, (const char*)packed, (const float2*)feat, (int)L,
                     store_hidden ? (elem*)ws->hid : (elem*)nullptr, hs, out, out_stride, out_off,
                     sdf_only ? (elem*)nullptr : (elem*)ws->sig, store_hidden ? ws->bits : (uint2*)nullptr, B);
  if (!sdf_only) {
    auto kc = k_wide_fwd_color<P, HB>;
    if (int e = set_smem(kc, shm_c)) return e;
    hipLaunchKernelGGL(kc, dim3(blocks), dim3(NT), shm_c, st, *d, (const char*)packed, (const elem*)ws->sig, view, (int)S,
                       store_hidden ? (elem*)ws->hid : (elem*)nullptr, hs, out, store_hidden ? ws->bits : (uint2*)nullptr, B);
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

/* feat [L,B,2], view [R,16] -> raw [B,4]; the hidden activations and the sigma head's output stay in `workspace` for
 * nof_mlp_wide_bwd (workspace: nof_mlp_wide_workspace_bytes(desc, B) bytes, caller-allocated). */
extern "C" int nof_mlp_wide_fwd(const NofMlpDesc* d, const void* packed, const float* feat, int32_t L, const float* view,
                                 int32_t S, float* raw, void* workspace, int64_t B, void* stream) {
  if (int e = check_wide(d)) return e;
  NOF_ARG(packed && feat && view && raw && workspace && B >= 0 && S >= 1 && L >= 1 && L * 2 == d->in_feat);
  NOF_ARG((int64_t)L * B * 8 < (1ll << 32));                   // level-major arrays are addressed with 32-bit lane offsets
  if (B == 0) return 0;
  const WideWs ws = wide_ws(d, workspace, B);
  WIDE_DISPATCH(wide_fwd_launch, d, packed, feat, L, view, S, raw, 4, 3, &ws, true, false, B, (hipStream_t)stream)
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
  WIDE_DISPATCH(wide_fwd_launch, d, packed, feat, L, (const float*)nullptr, 1, sdf, 1, 0, (const WideWs*)nullptr, false, true, B,
                (hipStream_t)stream)
  NOF_LAUNCH_OK();
  return 0;
}

template <class P, int HB>
static int wide_bwd_launch(const NofMlpDesc* d, const void* packed, const float* feat, int32_t L, const float* view, int32_t S,
                           const float* draw, const WideWs* ws, float* dfeat, float* dview, float* partials, int64_t B,
                           hipStream_t st, const void* tile_list, int parts) {
  typedef typename P::elem elem;
  const int ns = d->n_sigma, nc = d->n_color, nl = ns + nc, H = d->hidden;
  const size_t pair_bytes = 16 * 64 * 2;
  const size_t shm_s = (size_t)pair_base(*d, ns) * pair_bytes;
  const size_t shm_c = (size_t)(pair_base(*d, nl) - pair_base(*d, ns)) * pair_bytes;
  const int64_t ntiles = (B + 31) / 32;
  const unsigned blocks = (unsigned)(nof_div_up(ntiles, 12) < (int64_t)nof_cu_count() ? nof_div_up(ntiles, 12) : nof_cu_count());
  const int64_t hs = (B + 31) / 32 * 32 * (int64_t)H;
  elem *hid = (elem*)ws->hid, *gbuf = (elem*)ws->gbuf, *sig = (elem*)ws->sig, *dsig = (elem*)ws->dsig;
  auto kc = k_wide_bwd_color<P, HB>;
  auto ks = k_wide_bwd_sigma<P, HB>;
  if (int e = set_smem(kc, shm_c)) return e;
  if (int e = set_smem(ks, shm_s)) return e;
  if (parts & NOF_WIDE_BWD_DATA_COLOR)
    hipLaunchKernelGGL(kc, dim3(blocks), dim3(768), shm_c, st, *d, (const char*)packed, (const elem*)hid, hs, (int)S,
                       (const float4*)draw, gbuf, hs, dsig, dview, ws->zflag, B, tile_list, (const uint2*)ws->bits);
  if (parts & NOF_WIDE_BWD_DATA_SIGMA)
    hipLaunchKernelGGL(ks, dim3(blocks), dim3(768), shm_s, st, *d, (const char*)packed, (const elem*)hid, hs, (const elem*)dsig, gbuf,
                       hs, (float2*)dfeat, (int)L, (const uint8_t*)ws->zflag, B, tile_list, (const uint2*)ws->bits);
  // weight gradients, layer by layer (PN = output blocks of the layer: HB for the hidden layers, 1 for the two heads)
  for (int l = 0; l < nl; ++l) {
    if (!(parts & (l < ns ? NOF_WIDE_BWD_DW_SIGMA : NOF_WIDE_BWD_DW_COLOR))) continue;
    const int PN = lay_pn(*d, l);
    const unsigned grid = (unsigned)(kWideRows * PN / 4);              // 4 / PN tile streams per workgroup
    const elem* grow = gbuf + (int64_t)l * hs;
    const int hidx = l < ns ? l - 1 : (ns - 1) + (l - 1 - ns);         // the stored activation that is a hidden layer's input
    const elem* xin = (l == 0 || l == ns) ? (const elem*)nullptr : (const elem*)(hid + (int64_t)hidx * hs);
#define WIDE_DW(QN_, KIND_, PN_)                                                                          \
    hipLaunchKernelGGL((k_wide_dw<P, QN_, KIND_, PN_>), dim3(grid), dim3(256), 0, st, *d, l, grow, H, (const float2*)feat, \
                       (int)L, xin, (const elem*)sig, view, (int)S, partials, kWideRows, (const uint8_t*)ws->zflag, B, tile_list)
    if (l == 0) { WIDE_DW(1, 0, HB); }
    else if (l == ns) { WIDE_DW(2, 2, HB); }
    else if (PN == 1) { WIDE_DW(HB, 1, 1); }
    else { WIDE_DW(HB, 1, HB); }
#undef WIDE_DW
  }
  return 0;
}

/* draw [B,4] -> dfeat [L,B,2] (overwritten), dview [R,16] ACCUMULATED, partials [nof_mlp_wide_partial_rows(), n_params]
 * overwritten (sum the rows with nof_reduce_partials).  `workspace` as left by nof_mlp_wide_fwd of the same batch. */
extern "C" int nof_mlp_wide_bwd(const NofMlpDesc* d, const void* packed, const float* feat, int32_t L, const float* view,
                                 int32_t S, const float* draw, void* workspace, float* dfeat, float* dview, float* partials,
                                 int64_t B, void* stream) {
  return nof_mlp_wide_bwd_tiles(d, packed, feat, L, view, S, draw, workspace, dfeat, dview, partials, nullptr, B, stream);
}

/* the same over a work list (NofTileList): only the listed tiles are computed (data path and weight-gradient passes), dealt evenly
 * to the waves; dfeat of unlisted tiles is not written */
extern "C" int nof_mlp_wide_bwd_tiles(const NofMlpDesc* d, const void* packed, const float* feat, int32_t L, const float* view,
                                       int32_t S, const float* draw, void* workspace, float* dfeat, float* dview, float* partials,
                                       const void* tile_list, int64_t B, void* stream) {
  return nof_mlp_wide_bwd_parts(d, packed, feat, L, view, S, draw, workspace, dfeat, dview, partials, tile_list, NOF_WIDE_BWD_ALL, B, stream);
}

/* The same, restricted to `parts` (all on `stream`): the data path of the colour net (needs draw; writes dsigma / dview), of the
 * sigma net (needs the colour part; writes dfeat), and the two nets' weight-gradient passes, each of which only needs its own
 * net's data part.  What runs beside what is the caller's business: the training step starts the colour net's weight gradients
 * beside the sigma net's data path, and both nets' beside the hash backward (which only needs dfeat). */
extern "C" int nof_mlp_wide_bwd_parts(const NofMlpDesc* d, const void* packed, const float* feat, int32_t L, const float* view,
                                       int32_t S, const float* draw, void* workspace, float* dfeat, float* dview, float* partials,
                                       const void* tile_list, int32_t parts, int64_t B, void* stream) {
  if (int e = check_wide(d)) return e;
  NOF_ARG(parts >= 0 && parts <= NOF_WIDE_BWD_ALL);
  NOF_ARG(packed && feat && view && draw && workspace && dfeat && dview && partials && B >= 0 && S >= 32 && L * 2 == d->in_feat);
  NOF_ARG((int64_t)L * B * 8 < (1ll << 32));                   // level-major arrays are addressed with 32-bit lane offsets
  if (B == 0) return 0;
  const WideWs ws = wide_ws(d, workspace, B);
  WIDE_DISPATCH(wide_bwd_launch, d, packed, feat, L, view, S, draw, &ws, dfeat, dview, partials, B, (hipStream_t)stream, tile_list, (int)parts)
  NOF_LAUNCH_OK();
  return 0;
}
#undef WPAIR
