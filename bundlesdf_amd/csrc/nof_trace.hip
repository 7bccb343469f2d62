// Occupancy grid, ray / occupied-voxel intersection, z sampling and point generation for gfx950.
//
// Replaces, on a dense 2^level bitfield (<= 64^3 bits = 32 KiB, L2/L1 resident):
//   kaolin SPC octree build + unbatched_query + unbatched_raytrace      Utils.py:362-371,393,457
//   common.postprocessOctreeRayTracing                                  common.cu:129-167
//   common.sampleRaysUniformOccupiedVoxels                              common.cu:41-125
//   sample_rays_uniform / sample_rays_uniform_occupied_voxels / render_rays point generation
//                                                                       nerf_runner.py:67-87,979-1011,1044-1083,1242-1245
// Intersection definition (shared with oracle/nof_oracle.py:trace_rays): float32 slab test per cell with
// plane coordinates i*cs-1 (exact), t = (plane - o) * (1/d); 3-D DDA visits cells in plane-crossing order and
// evaluates every interval from the integer cell index (never incrementally), so that a cell's [t_in,t_out]
// is bit-identical to the brute-force slab test of the oracle.
#include "nof_common.h"
#include "nof_adam_tail_dev.h"
#pragma clang fp contract(off)

#define ZERO_DIR 1e-20f
#define MIN_LEN 1e-4f

// ------------------------------------------------------------------------------------------------
__global__ void k_occ_build(const int32_t* __restrict__ coords, int64_t P, int shift, int n, uint32_t* __restrict__ bits) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P) return;
  const int nmax = n << shift;
  int x = coords[i * 3 + 0], y = coords[i * 3 + 1], z = coords[i * 3 + 2];
  x = min(max(x, 0), nmax - 1) >> shift;
  y = min(max(y, 0), nmax - 1) >> shift;
  z = min(max(z, 0), nmax - 1) >> shift;
  const uint32_t id = ((uint32_t)x * n + y) * n + z;
  atomicOr(&bits[id >> 5], 1u << (id & 31));
}

__device__ __forceinline__ bool occ_test(const uint32_t* __restrict__ bits, int n, int x, int y, int z) {
  const uint32_t id = ((uint32_t)x * n + y) * n + z;
  return (bits[id >> 5] >> (id & 31)) & 1u;
}

__global__ void k_occ_query(const uint32_t* __restrict__ bits, int n, const float* __restrict__ pts,
                            uint8_t* __restrict__ inside, int64_t N) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  int c[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) {                                     // kaolin quantize_points: floor(clamp(n*(x+1)/2, 0, n-1))
    float q = (float)n * (pts[i * 3 + d] + 1.0f) / 2.0f;
    q = fminf(fmaxf(q, 0.0f), (float)n - 1.0f);
    c[d] = (int)floorf(q);
  }
  inside[i] = occ_test(bits, n, c[0], c[1], c[2]) ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
struct Axis {
  float o, inv;
  bool zero;
  int step;
};

__device__ __forceinline__ void cell_slab(const Axis& a, int i, float cs, float& tmin, float& tmax) {
  const float lo = (float)i * cs - 1.0f;
  const float hi = (float)(i + 1) * cs - 1.0f;
  if (a.zero) {
    const bool in = (lo <= a.o) && (a.o < hi);
    tmin = in ? -NOF_INF : NOF_INF;
    tmax = in ? NOF_INF : -NOF_INF;
  } else {
    const float t0 = (lo - a.o) * a.inv;
    const float t1 = (hi - a.o) * a.inv;
    tmin = fminf(t0, t1);
    tmax = fmaxf(t0, t1);
  }
}

// The DDA's occupancy test.  LDS: the bitfield staged by stage_occ, read through the dynamic LDS block ITSELF, i.e. a ds_read.  (A
// pointer that is "the LDS copy or the global one" is a flat pointer: the test then was a flat load per DDA step -- one vector-memory
// round trip through the texture path per cell, 47 of them per wave and zero LDS instructions in profiles/r04_z_pmc_vmem.txt -- on a
// kernel that is nothing but one dependent chain per ray.)
template <bool LDS>
__device__ __forceinline__ bool occ_test_t(const uint32_t* __restrict__ bits, int n, int x, int y, int z) {
  const uint32_t id = ((uint32_t)x * n + y) * n + z;
  if constexpr (LDS) {
    extern __shared__ uint32_t occ_lds_words[];
    return (occ_lds_words[id >> 5] >> (id & 31)) & 1u;
  } else {
    return (bits[id >> 5] >> (id & 31)) & 1u;
  }
}

// Walks the occupied cells of one ray; writes at most max_hits intervals. Returns the number written.
template <bool LDS>
__device__ __forceinline__ int trace_one_t(const uint32_t* __restrict__ bits, int n, const float o[3], const float d[3], int max_hits,
                         float* __restrict__ tio, int32_t* __restrict__ cid, int* overflow) {
  const float cs = 2.0f / (float)n;
  Axis ax[3];
  float tenter = 0.0f, texit = NOF_INF;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    ax[a].o = o[a];
    ax[a].zero = fabsf(d[a]) < ZERO_DIR;
    ax[a].inv = ax[a].zero ? 0.0f : 1.0f / d[a];
    ax[a].step = d[a] > 0.0f ? 1 : -1;
    if (ax[a].zero) {
      if (!(-1.0f <= o[a] && o[a] < 1.0f)) texit = -NOF_INF;
    } else {
      const float t0 = (-1.0f - o[a]) * ax[a].inv, t1 = (1.0f - o[a]) * ax[a].inv;
      tenter = fmaxf(tenter, fminf(t0, t1));
      texit = fminf(texit, fmaxf(t0, t1));
    }
  }
  if (!(tenter <= texit)) return 0;
  int c[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float p = ax[a].zero ? o[a] : (o[a] + tenter * d[a]);
    int i = (int)floorf((p + 1.0f) / cs);
    i = min(max(i, 0), n - 1);
    if (!ax[a].zero) {
      // make the start cell consistent with the exact plane times: tmin <= tenter <= tmax
      for (int it = 0; it < 4; ++it) {
        float tmin, tmax;
        cell_slab(ax[a], i, cs, tmin, tmax);
        if (tmax < tenter && i + ax[a].step >= 0 && i + ax[a].step < n) i += ax[a].step;
        else if (tmin > tenter && i - ax[a].step >= 0 && i - ax[a].step < n) i -= ax[a].step;
        else break;
      }
    }
    c[a] = i;
  }
  int nh = 0;
  const int max_iter = 3 * n + 8;
  for (int it = 0; it < max_iter; ++it) {
    float tmin[3], tmax[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) cell_slab(ax[a], c[a], cs, tmin[a], tmax[a]);
    const float tin = fmaxf(fmaxf(fmaxf(tmin[0], tmin[1]), tmin[2]), 0.0f);
    const float tout = fminf(fminf(tmax[0], tmax[1]), tmax[2]);
    if (tin <= tout && occ_test_t<LDS>(bits, n, c[0], c[1], c[2])) {
      if (tin == 0.0f || tout == 0.0f) break;                       // common.cu:140 (terminator)
      if (!(fabsf(tout - tin) < MIN_LEN)) {                          // common.cu:142
        if (nh < max_hits) {
          tio[2 * nh] = tin;
          tio[2 * nh + 1] = tout;
          if (cid) cid[nh] = (int32_t)(((uint32_t)c[0] * n + c[1]) * n + c[2]);
          ++nh;
        } else {
          *overflow = 1;
        }
      }
    }
    // leave through the nearest exit plane (ties: x, y, z).  The axis is picked with selects, not by indexing ax[] / c[] with it:
    // a run-time index sends both arrays to scratch memory, and this loop then waits for a scratch round trip per step
    const bool y_first = tmax[1] < tmax[0];
    const float t01 = y_first ? tmax[1] : tmax[0];
    const bool z_first = tmax[2] < t01;
    const bool zero_a = z_first ? ax[2].zero : (y_first ? ax[1].zero : ax[0].zero);
    const int step_a = z_first ? ax[2].step : (y_first ? ax[1].step : ax[0].step);
    if (zero_a) break;
    c[0] += (!z_first && !y_first) ? step_a : 0;
    c[1] += (!z_first && y_first) ? step_a : 0;
    c[2] += z_first ? step_a : 0;
    const int ca = z_first ? c[2] : (y_first ? c[1] : c[0]);
    if (ca < 0 || ca >= n) break;
  }
  return nh;
}

// The occupancy bitfield of the ray-tracing level is tiny (level 4: 512 B, level 6: 32 KB): every workgroup stages it in LDS
// so that the DDA's dependent occupancy tests cost an LDS read instead of a global round trip (the kernel is one latency
// chain per ray: 64 waves for 4096 rays).  Levels above 6 read the global copy.
#define OCC_LDS_WORDS 8192
__device__ __forceinline__ const uint32_t* stage_occ(const uint32_t* __restrict__ bits, int n, uint32_t* lds) {
  const int words = (n * n * n + 31) / 32;
  if (words > OCC_LDS_WORDS) return bits;
  const int quads = words >> 2;                                         // 16-byte copies (the bitfield is 256-byte aligned)
  for (int i = threadIdx.x; i < quads; i += blockDim.x)
    reinterpret_cast<uint4*>(lds)[i] = reinterpret_cast<const uint4*>(bits)[i];
  for (int i = (quads << 2) + threadIdx.x; i < words; i += blockDim.x) lds[i] = bits[i];
  __syncthreads();
  return lds;
}

// `occ`: what stage_occ returned (the LDS copy, or `bits` itself for levels that do not fit)
__device__ __forceinline__ int trace_one(const uint32_t* __restrict__ occ, const uint32_t* __restrict__ bits, int n, const float o[3],
                                         const float d[3], int max_hits, float* __restrict__ tio, int32_t* __restrict__ cid, int* overflow) {
  return occ != bits ? trace_one_t<true>(occ, n, o, d, max_hits, tio, cid, overflow)       // (wave-uniform)
                     : trace_one_t<false>(bits, n, o, d, max_hits, tio, cid, overflow);
}

__global__ __launch_bounds__(64) void k_trace_rays(const uint32_t* __restrict__ bits, int n, const float* __restrict__ rays_o,
                                                    const float* __restrict__ rays_d, int64_t R, int max_hits,
                                                    float* __restrict__ t_in_out, int32_t* __restrict__ cell_ids,
                                                    int32_t* __restrict__ n_hits, int32_t* __restrict__ flags) {
  extern __shared__ uint32_t occ_lds[];
  const uint32_t* occ = stage_occ(bits, n, occ_lds);
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const float o[3] = {rays_o[r * 3], rays_o[r * 3 + 1], rays_o[r * 3 + 2]};
  const float d[3] = {rays_d[r * 3], rays_d[r * 3 + 1], rays_d[r * 3 + 2]};
  float* tio = t_in_out + r * max_hits * 2;
  int32_t* cid = cell_ids ? cell_ids + r * max_hits : nullptr;
  int overflow = 0;
  const int nh = trace_one(occ, bits, n, o, d, max_hits, tio, cid, &overflow);
  for (int k = nh; k < max_hits; ++k) reinterpret_cast<float2*>(tio)[k] = make_float2(0.f, 0.f);   // zero padding (at::zeros, common.cu:158)
  if (cid)
    for (int k = nh; k < max_hits; ++k) cid[k] = -1;
  n_hits[r] = nh;
  if (overflow && flags) atomicOr(&flags[0], 1);
}

// ------------------------------------------------------------------------------------------------
// SH of the world view direction (nerf_helpers.py:67-105), degree <= 4
__device__ __forceinline__ void sh_eval(int degree, float x, float y, float z, float* out) {
  out[0] = 0.28209479177387814f;
  if (degree > 1) {
    out[1] = -0.4886025119029199f * y;
    out[2] = 0.4886025119029199f * z;
    out[3] = -0.4886025119029199f * x;
  }
  if (degree > 2) {
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    out[4] = 1.0925484305920792f * xy;
    out[5] = -1.0925484305920792f * yz;
    out[6] = 0.31539156525252005f * (2.0f * zz - xx - yy);
    out[7] = -1.0925484305920792f * xz;
    out[8] = 0.5462742152960396f * (xx - yy);
    if (degree > 3) {
      out[9] = -0.5900435899266435f * y * (3.0f * xx - yy);
      out[10] = 2.890611442640554f * xy * z;
      out[11] = -0.4570457994644658f * y * (4.0f * zz - xx - yy);
      out[12] = 0.3731763325901154f * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
      out[13] = -0.4570457994644658f * x * (4.0f * zz - xx - yy);
      out[14] = 1.445305721320277f * z * (xx - yy);
      out[15] = -0.5900435899266435f * x * (xx - 3.0f * yy);
    }
  }
}

__global__ __launch_bounds__(64) void k_batch_trace(const float* __restrict__ pool, const int64_t* __restrict__ ids,
                                                     const float* __restrict__ tf, const float* __restrict__ frame_feat, int ff,
                                                     int sh_degree, const uint32_t* __restrict__ bits, int n, int64_t R,
                                                     int max_hits, float* __restrict__ batch, float* __restrict__ rays_o_w,
                                                     float* __restrict__ viewdirs_w, float* __restrict__ view,
                                                     float* __restrict__ t_in_out, int32_t* __restrict__ cell_ids,
                                                     int32_t* __restrict__ n_hits, int32_t* __restrict__ flags) {
  extern __shared__ uint32_t occ_lds[];
  const uint32_t* occ = stage_occ(bits, n, occ_lds);
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (16 rays per wave in 4x the waves: 28 -> 33 us, each
  if (r >= R) return;                                                     //  workgroup stages the 32 KB bitfield)
  const int64_t src = ids ? ids[r] : r;
  float row[NOF_RAY_COLS];
#pragma unroll
  for (int k = 0; k < NOF_RAY_COLS; ++k) {
    row[k] = pool[src * NOF_RAY_COLS + k];
    batch[r * NOF_RAY_COLS + k] = row[k];
  }
  const int f = (int)row[8];
  const float* T = tf + (int64_t)f * 12;
  // viewdirs = rays_d / |rays_d| (nerf_runner.py:1047); rays_o = 0 so rays_o_w = translation (:1056)
  const float nrm = sqrtf(row[0] * row[0] + row[1] * row[1] + row[2] * row[2]);
  const float v[3] = {row[0] / nrm, row[1] / nrm, row[2] / nrm};
  float o[3], d[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    o[i] = T[i * 4 + 3];
    d[i] = (T[i * 4 + 0] * v[0] + T[i * 4 + 1] * v[1]) + T[i * 4 + 2] * v[2];
    rays_o_w[r * 3 + i] = o[i];
    viewdirs_w[r * 3 + i] = d[i];
  }
  float vw[NOF_VIEW_COLS];
#pragma unroll
  for (int k = 0; k < NOF_VIEW_COLS; ++k) vw[k] = 0.0f;
  float sh[16];
  sh_eval(sh_degree, d[0], d[1], d[2], sh);
  const int nsh = sh_degree * sh_degree;
  for (int k = 0; k < ff; ++k) vw[k] = frame_feat[(int64_t)f * ff + k];      // [frame_features | SH] (nerf_runner.py:1270-1286)
  for (int k = 0; k < nsh && ff + k < NOF_VIEW_COLS; ++k) vw[ff + k] = sh[k];
#pragma unroll
  for (int k = 0; k < NOF_VIEW_COLS; ++k) view[r * NOF_VIEW_COLS + k] = vw[k];

  float* tio = t_in_out + r * max_hits * 2;
  int32_t* cid = cell_ids ? cell_ids + r * max_hits : nullptr;
  int overflow = 0;
  const int nh = trace_one(occ, bits, n, o, d, max_hits, tio, cid, &overflow);
  // zero padding of the interval list (at::zeros in common.cu:158) by the ray's own lane: a memset launch in front of this kernel
  // cost 8 us of the step
  for (int k = nh; k < max_hits; ++k) reinterpret_cast<float2*>(tio)[k] = make_float2(0.f, 0.f);
  if (cid)
    for (int k = nh; k < max_hits; ++k) cid[k] = -1;
  n_hits[r] = nh;
  if (overflow && flags) atomicOr(&flags[0], 1);
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al. 2011) -> one uniform in [0,1)
__device__ __forceinline__ float philox_uniform(uint64_t seed, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3) {
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  uint32_t x0 = c0, x1 = c1, x2 = c2, x3 = c3;
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * x0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * x2;
    const uint32_t y0 = (uint32_t)(p1 >> 32) ^ x1 ^ k0;
    const uint32_t y1 = (uint32_t)p1;
    const uint32_t y2 = (uint32_t)(p0 >> 32) ^ x3 ^ k1;
    const uint32_t y3 = (uint32_t)p0;
    x0 = y0; x1 = y1; x2 = y2; x3 = y3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return (float)(x0 >> 8) * (1.0f / 16777216.0f);
}

__device__ __forceinline__ float lin01(int i, int N) {               // torch.linspace(0,1,N)[i] in float32
  const float step = 1.0f / (float)(N - 1);                          // upper half: one rounding (fused), as ATen computes it
  return (i < N / 2) ? (float)i * step : __builtin_fmaf(-step, (float)(N - 1 - i), 1.0f);
}

// sample_rays_uniform (nerf_runner.py:67-87) for one sample index
__device__ __forceinline__ float stratified(int i, int N, float near, float far, float u) {
  const float t = lin01(i, N);
  const float zi = near * (1.0f - t) + far * t;
  float lower = zi, upper = zi;
  if (i > 0) {
    const float tp = lin01(i - 1, N);
    const float zp = near * (1.0f - tp) + far * tp;
    lower = 0.5f * (zi + zp);
  }
  if (i < N - 1) {
    const float tn = lin01(i + 1, N);
    const float zn = near * (1.0f - tn) + far * tn;
    upper = 0.5f * (zn + zi);
  }
  float z = lower + (upper - lower) * u;
  return fminf(fmaxf(z, near), far);
}
// the same with perturb=False (render_images, nerf_runner.py:597): the linspace itself, no jitter and no clip (:78-85 are skipped)
__device__ __forceinline__ float unperturbed(int i, int N, float near, float far) {
  const float t = lin01(i, N);
  return near * (1.0f - t) + far * t;
}

// sample s of ray r: its z by the reference's sequential subtraction walk over the ray's clipped intervals (zin / zout, `total` their
// summed length), the sample point in world coordinates and its validity.  Shared by k_sample_points (a workgroup per ray) and the
// fused wave-per-ray kernel (k_raymarch_wave<true>): the same code, the same bits.
__device__ __forceinline__ void sample_place(const NofSampleCfg& cfg, int64_t r, int s, int S, int nh, float total,
                                             const float* zin, const float* zout, bool valid_depth, float depth, float dx, float dy,
                                             float dz, int f, const float* __restrict__ tf, const float* __restrict__ u_occ,
                                             const float* __restrict__ u_dep, float* __restrict__ z_vals, float* __restrict__ pts_w,
                                             uint8_t* __restrict__ valid, int32_t* __restrict__ flags) {
  float z;
  bool occupied_mode;
  int N, i;
  float u;
  if (s < cfg.n_samples) {
    occupied_mode = true; N = cfg.n_samples; i = s;
    u = u_occ ? u_occ[r * cfg.n_samples + i] : philox_uniform(cfg.seed, (uint32_t)r, (uint32_t)s, cfg.d_step ? *cfg.d_step : cfg.step, 0u);
  } else {
    N = cfg.n_around; i = s - cfg.n_samples;
    occupied_mode = !valid_depth;                                       // nerf_runner.py:1072-1076
    u = u_dep ? u_dep[r * cfg.n_around + i] : philox_uniform(cfg.seed, (uint32_t)r, (uint32_t)s, cfg.d_step ? *cfg.d_step : cfg.step, 0u);
  }
  if (!occupied_mode) {
    const float nd = depth - cfg.trunc;                                  // nerf_runner.py:1067-1071
    const float fd = depth + cfg.trunc * cfg.neg_trunc_ratio;
    z = cfg.deterministic ? unperturbed(i, N, nd, fd) : stratified(i, N, nd, fd, u);
  } else if (nh == 0) {
    z = 0.0f;                                                            // common.cu:54 (no box -> z_vals stays 0)
  } else {
    float zr = cfg.deterministic ? unperturbed(i, N, 0.0f, total) : stratified(i, N, 0.0f, total, u);
    int ib = 0;
    for (;;) {                                                           // common.cu:56-104
      if (ib >= nh) {
        z = zout[nh - 1];
        if (zr > 1e-4f && flags) atomicOr(&flags[0], 2);                // the reference would print + spin here
        break;
      }
      const float bl = zout[ib] - zin[ib];
      if (zr <= bl) { z = zin[ib] + zr; break; }
      zr -= bl;
      ++ib;
    }
  }
  const int64_t b = r * S + s;
  z_vals[b] = z;
  const float px = dx * z, py = dy * z, pz = dz * z;                     // pts = rays_d * z (nerf_runner.py:1083)
  const float* T = tf + (int64_t)f * 12;
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 3; ++k) {                                           // transform_pts (Utils.py:253-257)
    const float x = ((T[k * 4 + 0] * px + T[k * 4 + 1] * py) + T[k * 4 + 2] * pz) + T[k * 4 + 3];
    pts_w[b * 3 + k] = x;
    ok = ok && (fabsf(x) <= 1.0f);                                         // nerf_runner.py:1245
  }
  valid[b] = ok ? 1 : 0;
}

// One workgroup per ray: lanes stage the (clipped) z intervals in LDS, lane 0 sums their lengths in
// order, then every lane places its sample by the reference's sequential subtraction walk.
__global__ void k_sample_points(NofSampleCfg cfg, const float* __restrict__ batch, const float* __restrict__ tf,
                                const float* __restrict__ t_in_out, const int32_t* __restrict__ n_hits, int max_hits,
                                const float* __restrict__ u_occ, const float* __restrict__ u_dep,
                                float* __restrict__ z_vals, float* __restrict__ pts_w, uint8_t* __restrict__ valid,
                                int32_t* __restrict__ flags) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* zin = smem;
  float* zout = smem + max_hits;
  __shared__ float s_total;
  const int64_t r = blockIdx.x;
  // a new batch starts here: a "this step's weight gradient is not finite" mark left by the PREVIOUS step (bit 2, raised by
  // nof_reduce_partials / nof_grad_check and consumed by that step's Adam launches, which skipped) becomes the sticky bit 3 that
  // the host polls to lower the loss scale (field.poll_flags) -- so that exactly the offending step is skipped, like GradScaler
  if (blockIdx.x == 0 && threadIdx.x == 0 && flags != nullptr) {
    if (atomicAnd(&flags[0], ~4) & 4) atomicOr(&flags[0], 8);
  }
  const int S = cfg.n_samples + cfg.n_around;
  const float* row = batch + r * NOF_RAY_COLS;
  const float dx = row[0], dy = row[1], dz = row[2];
  const float depth = row[6];
  const int f = (int)row[8];
  const float nrm = sqrtf(dx * dx + dy * dy + dz * dz);
  const float vz = fabsf(dz / nrm);                                   // |viewdirs_z| (nerf_runner.py:987-990)
  const bool valid_depth = (depth >= cfg.near_sc) && (depth <= cfg.far_sc);
  const int nh = n_hits[r];
  for (int h = threadIdx.x; h < nh; h += blockDim.x) {
    float zi = t_in_out[(r * max_hits + h) * 2] * vz;
    float zo = t_in_out[(r * max_hits + h) * 2 + 1] * vz;
    if (valid_depth && zi > 0.0f && zo > 0.0f) {                       // nerf_runner.py:995-999
      const float cap = depth + cfg.trunc;
      zi = fminf(fmaxf(zi, 0.0f), cap);
      zo = fminf(fmaxf(zo, 0.0f), cap);
    }
    zin[h] = zi;
    zout[h] = zo;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.0f;
    for (int h = 0; h < nh; ++h) tot += zout[h] - zin[h];              // depths_lens.sum (:1002-1003), in order
    s_total = tot;
  }
  __syncthreads();
  const int s = threadIdx.x;
  if (s >= S) return;
  sample_place(cfg, r, s, S, nh, s_total, zin, zout, valid_depth, depth, dx, dy, dz, f, tf, u_occ, u_dep, z_vals, pts_w, valid, flags);
}

// ------------------------------------------------------------------------------------------------
// The same ray marcher with ONE WAVE PER RAY (nof_set_trace_kernel(1); levels <= 6): no walk.  Along axis a the ray leaves cell
// index i at tmax_a(i) -- cell_slab's closed form, non-decreasing along the direction of travel -- so the walk's sequence of exit
// axes is the 3-way merge of three sorted lists under the walk's tie rule (x, then y, then z), and crossing j of axis a has the rank
//     j + sum over the other axes b of #{m : tmax_b(m) < tmax_a(j)}   (b > a: b loses ties)   or   #{... <= ...}   (b < a)
// -- two binary searches over closed forms.  The cell entered through it is the start cell moved by (j + 1) steps along a and by
// those counts along the other axes; the walk ends at the first crossing that is the last of its axis.  One lane per crossing
// computes its cell, the cell's interval and occupancy bit on its own; the cells go to a wave-private LDS table in rank order, a
// ballot finds the terminator (common.cu:140) and a prefix count compacts the hits.  Same float32 expressions as trace_one_t, hence
// the same bits: tools/dda_closed_form.py is this algorithm in NumPy, checked against the walk on the CPU
// (tests/test_oracle.py::test_dda_closed_form_enumeration_equals_the_walk), and tests/test_gpu_ops.py compares the two kernels.
#define NOF_TW_SLOTS 196                                               // 3 * 64 crossings + the start cell, rounded up
struct WaveCells {
  float tin[NOF_TW_SLOTS], tout[NOF_TW_SLOTS];
  int32_t code[NOF_TW_SLOTS];                                          // cell id of a hit to emit, -2: nothing, -3: the terminator
  float zin[NOF_TW_SLOTS], zout[NOF_TW_SLOTS];                         // the sampler's clipped intervals, in hit order (k_raymarch_wave<true>)
};

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}

// returns the number of hits written (wave-uniform); tio / cid rows are this ray's
__device__ __forceinline__ int trace_wave(int n, const float o[3], const float d[3], int max_hits, float* __restrict__ tio,
                                          int32_t* __restrict__ cid, int* overflow, WaveCells* wc, int lane) {
  const float cs = 2.0f / (float)n;
  Axis ax[3];
  float tenter = 0.0f, texit = NOF_INF;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    ax[a].o = o[a];
    ax[a].zero = fabsf(d[a]) < ZERO_DIR;
    ax[a].inv = ax[a].zero ? 0.0f : 1.0f / d[a];
    ax[a].step = d[a] > 0.0f ? 1 : -1;
    if (ax[a].zero) {
      if (!(-1.0f <= o[a] && o[a] < 1.0f)) texit = -NOF_INF;
    } else {
      const float t0 = (-1.0f - o[a]) * ax[a].inv, t1 = (1.0f - o[a]) * ax[a].inv;
      tenter = fmaxf(tenter, fminf(t0, t1));
      texit = fminf(texit, fmaxf(t0, t1));
    }
  }
  if (!(tenter <= texit)) return 0;
  int c0[3], m[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {                                         // the walk's start cell, unchanged
    const float p = ax[a].zero ? o[a] : (o[a] + tenter * d[a]);
    int i = (int)floorf((p + 1.0f) / cs);
    i = min(max(i, 0), n - 1);
    if (!ax[a].zero) {
      for (int it = 0; it < 4; ++it) {
        float tmin, tmax;
        cell_slab(ax[a], i, cs, tmin, tmax);
        if (tmax < tenter && i + ax[a].step >= 0 && i + ax[a].step < n) i += ax[a].step;
        else if (tmin > tenter && i - ax[a].step >= 0 && i - ax[a].step < n) i -= ax[a].step;
        else break;
      }
    }
    c0[a] = i;
    m[a] = ax[a].zero ? 0 : (ax[a].step > 0 ? n - i : i + 1);           // crossings of axis a until the ray leaves the grid
  }
  const int M = m[0] + m[1] + m[2];
  int k_end = M == 0 ? 0 : 0x7fffffff;                                  // rank of the first crossing that leaves the grid
  for (int q0 = 0; q0 <= M; q0 += 64) {
    const int q = q0 + lane;                                            // 0: the start cell; q >= 1: crossing q - 1
    const bool live = q <= M;
    const int e = q - 1;
    const int a = e < m[0] ? 0 : (e < m[0] + m[1] ? 1 : 2);             // (selects, not indexing: ax[] / c0[] stay in registers)
    const int j = e - (a == 0 ? 0 : (a == 1 ? m[0] : m[0] + m[1]));
    float t = 0.0f;
    if (q >= 1) {
      float tmin0, tmax0, tmin1, tmax1, tmin2, tmax2;
      cell_slab(ax[0], c0[0] + j * ax[0].step, cs, tmin0, tmax0);
      cell_slab(ax[1], c0[1] + j * ax[1].step, cs, tmin1, tmax1);
      cell_slab(ax[2], c0[2] + j * ax[2].step, cs, tmin2, tmax2);
      t = a == 0 ? tmax0 : (a == 1 ? tmax1 : tmax2);
    }
    int cnt[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      int lo = 0, hi = m[b];
      const bool use_le = b < a;                                        // b wins ties against a
#pragma unroll
      for (int it = 0; it < 7; ++it) {                                  // m[b] <= 64
        if (lo < hi) {
          const int mid = (lo + hi) >> 1;
          float tmn, tmx;
          cell_slab(ax[b], c0[b] + mid * ax[b].step, cs, tmn, tmx);
          const bool before = use_le ? tmx <= t : tmx < t;
          lo = before ? mid + 1 : lo;
          hi = before ? hi : mid;
        }
      }
      cnt[b] = q >= 1 ? (b == a ? j : lo) : 0;
    }
    const int rank = cnt[0] + cnt[1] + cnt[2];
    const int m_a = a == 0 ? m[0] : (a == 1 ? m[1] : m[2]);
    const bool leaves = q >= 1 && live && j == m_a - 1;
    k_end = min(k_end, wave_min_i(leaves ? rank : 0x7fffffff));
    if (live && !leaves) {
      const int cc0 = c0[0] + (cnt[0] + (q >= 1 && a == 0 ? 1 : 0)) * ax[0].step;
      const int cc1 = c0[1] + (cnt[1] + (q >= 1 && a == 1 ? 1 : 0)) * ax[1].step;
      const int cc2 = c0[2] + (cnt[2] + (q >= 1 && a == 2 ? 1 : 0)) * ax[2].step;
      // a crossing ranked behind the one that leaves the grid (rank > k_end: cnt[b] can reach m[b]) names a cell outside [0, n)^3.
      // Its table entry is never read (the compaction below stops at k_end), so it is not computed either: no occupancy read
      // outside the bitfield, no LDS write
      const bool in_grid = (unsigned)cc0 < (unsigned)n && (unsigned)cc1 < (unsigned)n && (unsigned)cc2 < (unsigned)n;
      const int k = q >= 1 ? rank + 1 : 0;
      if (in_grid) {
        float tmin[3], tmax[3];
        cell_slab(ax[0], cc0, cs, tmin[0], tmax[0]);
        cell_slab(ax[1], cc1, cs, tmin[1], tmax[1]);
        cell_slab(ax[2], cc2, cs, tmin[2], tmax[2]);
        const float tin = fmaxf(fmaxf(fmaxf(tmin[0], tmin[1]), tmin[2]), 0.0f);
        const float tout = fminf(fminf(tmax[0], tmax[1]), tmax[2]);
        int code = -2;
        if (tin <= tout && occ_test_t<true>(nullptr, n, cc0, cc1, cc2)) {
          if (tin == 0.0f || tout == 0.0f) code = -3;                    // common.cu:140 (terminator)
          else if (!(fabsf(tout - tin) < MIN_LEN)) code = (int32_t)(((uint32_t)cc0 * n + cc1) * n + cc2);   // common.cu:142
        }
        wc->tin[k] = tin;
        wc->tout[k] = tout;
        wc->code[k] = code;
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");               // (one wave: LDS operations retire in order; nothing to wait for)
  __builtin_amdgcn_wave_barrier();
  int nh = 0;
  for (int k0 = 0; k0 <= k_end; k0 += 64) {
    const int k = k0 + lane;
    const bool valid = k <= k_end;
    const int code = valid ? wc->code[k] : -2;
    const unsigned long long term = __builtin_amdgcn_ballot_w64(code == -3);
    const int k_term = term ? k0 + __builtin_ctzll(term) : 0x7fffffff;
    const bool emit = code >= 0 && k < k_term;
    const unsigned long long em = __builtin_amdgcn_ballot_w64(emit);
    const int pos = nh + __builtin_popcountll(em & ((1ull << lane) - 1ull));
    if (emit) {
      if (pos < max_hits) {
        const float2 io = make_float2(wc->tin[k], wc->tout[k]);
        reinterpret_cast<float2*>(tio)[pos] = io;
        if (cid) cid[pos] = code;
        if (pos < NOF_TW_SLOTS) { wc->zin[pos] = io.x; wc->zout[pos] = io.y; }      // (raw; the sampler half scales and clips them)
      } else {
        *overflow = 1;
      }
    }
    nh += __builtin_popcountll(em);
    if (term) break;
  }
  return min(nh, max_hits);
}

// SAMPLE: followed, in the same wave, by the sampler (k_sample_points' arithmetic through sample_place, the intervals taken from the
// LDS table instead of being read back): nof_raymarch_sample as ONE launch.
// (the kernel's body as a device function of the workgroup's number: the stand-alone kernel passes blockIdx.x; the optimiser launch of
//  the step BEFORE carries the same workgroups as one of its roles, nof_adam_step_tail_march.  NEWBATCH: this launch is also where a
//  batch starts for the overflow mark of the device flags -- not so inside the optimiser launch, whose other workgroups still read it.)
template <bool SAMPLE, bool NEWBATCH>
__device__ __forceinline__ void raymarch_wave_block(const float* __restrict__ pool, const int64_t* __restrict__ ids,
                                                    const float* __restrict__ tf, const float* __restrict__ frame_feat, int ff,
                                                    int sh_degree, const uint32_t* __restrict__ bits, int n, int64_t R,
                                                    int max_hits, float* __restrict__ batch, float* __restrict__ rays_o_w,
                                                    float* __restrict__ viewdirs_w, float* __restrict__ view,
                                                    float* __restrict__ t_in_out, int32_t* __restrict__ cell_ids,
                                                    int32_t* __restrict__ n_hits, int32_t* __restrict__ flags, int cells_off,
                                                    const NofSampleCfg cfg, const float* __restrict__ u_occ,
                                                    const float* __restrict__ u_dep, float* __restrict__ z_vals,
                                                    float* __restrict__ pts_w, uint8_t* __restrict__ valid, const uint32_t block_id,
                                                    uint32_t* occ_lds) {
  if (SAMPLE && NEWBATCH && block_id == 0 && threadIdx.x == 0 && flags != nullptr) {   // a new batch starts here: see k_sample_points
    if (atomicAnd(&flags[0], ~4) & 4) atomicOr(&flags[0], 8);
  }
  stage_occ(bits, n, occ_lds);                                           // (levels <= 6 only: the bitfield is in LDS)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r = (int64_t)block_id * 4 + wave;
  if (r >= R) return;                                                    // (wave-uniform)
  WaveCells* wc = reinterpret_cast<WaveCells*>(reinterpret_cast<char*>(occ_lds) + cells_off) + wave;
  // every lane holds the ray (uniform loads); lanes 0..11 / 0..15 / 0..2 write its rows
  const int64_t src = ids ? ids[r] : r;
  float row[NOF_RAY_COLS];
#pragma unroll
  for (int k = 0; k < NOF_RAY_COLS; ++k) row[k] = pool[src * NOF_RAY_COLS + k];
  const int f = (int)row[8];
  const float* T = tf + (int64_t)f * 12;
  const float nrm = sqrtf(row[0] * row[0] + row[1] * row[1] + row[2] * row[2]);
  const float v[3] = {row[0] / nrm, row[1] / nrm, row[2] / nrm};
  float o[3], d[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    o[i] = T[i * 4 + 3];
    d[i] = (T[i * 4 + 0] * v[0] + T[i * 4 + 1] * v[1]) + T[i * 4 + 2] * v[2];
  }
  float sh[16];
  sh_eval(sh_degree, d[0], d[1], d[2], sh);
  const int nsh = sh_degree * sh_degree;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NOF_RAY_COLS; ++k) batch[r * NOF_RAY_COLS + k] = row[k];
#pragma unroll
    for (int i = 0; i < 3; ++i) { rays_o_w[r * 3 + i] = o[i]; viewdirs_w[r * 3 + i] = d[i]; }
    // the view row [frame features | SH | 0]: placed by ADDRESS (a register array indexed by ff went to scratch)
    float* vrow = view + r * NOF_VIEW_COLS;
    for (int k = 0; k < ff; ++k) vrow[k] = frame_feat[(int64_t)f * ff + k];
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if (k < nsh && ff + k < NOF_VIEW_COLS) vrow[ff + k] = sh[k];
    for (int k = ff + nsh; k < NOF_VIEW_COLS; ++k) vrow[k] = 0.0f;
  }
  float* tio = t_in_out + r * max_hits * 2;
  int32_t* cid = cell_ids ? cell_ids + r * max_hits : nullptr;
  int overflow = 0;
  const int nh = trace_wave(n, o, d, max_hits, tio, cid, &overflow, wc, lane);
  for (int k = nh + lane; k < max_hits; k += 64) {                       // zero padding (at::zeros, common.cu:158), the wave together
    reinterpret_cast<float2*>(tio)[k] = make_float2(0.f, 0.f);
    if (cid) cid[k] = -1;
  }
  if (lane == 0) n_hits[r] = nh;
  if (__builtin_amdgcn_ballot_w64(overflow != 0) != 0ull && lane == 0 && flags) atomicOr(&flags[0], 1);
  if constexpr (SAMPLE) {
    // k_sample_points for this ray, the wave instead of a workgroup: clip the intervals (nerf_runner.py:995-999), sum their lengths
    // in order (:1002-1003), place the S samples
    const int S = cfg.n_samples + cfg.n_around;
    const float dx = row[0], dy = row[1], dz = row[2];
    const float depth = row[6];
    const float vz = fabsf(dz / nrm);                                   // |viewdirs_z| (nerf_runner.py:987-990)
    const bool valid_depth = (depth >= cfg.near_sc) && (depth <= cfg.far_sc);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int h = lane; h < nh; h += 64) {
      float zi = wc->zin[h] * vz;
      float zo = wc->zout[h] * vz;
      if (valid_depth && zi > 0.0f && zo > 0.0f) {
        const float cap = depth + cfg.trunc;
        zi = fminf(fmaxf(zi, 0.0f), cap);
        zo = fminf(fmaxf(zo, 0.0f), cap);
      }
      wc->zin[h] = zi;
      wc->zout[h] = zo;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    float tot = 0.0f;
    for (int h = 0; h < nh; ++h) tot += wc->zout[h] - wc->zin[h];       // (every lane, the same order as the workgroup's thread 0)
    for (int sidx = lane; sidx < S; sidx += 64)
      sample_place(cfg, r, sidx, S, nh, tot, wc->zin, wc->zout, valid_depth, depth, dx, dy, dz, f, tf, u_occ, u_dep, z_vals, pts_w,
                   valid, flags);
  }
}

template <bool SAMPLE>
__global__ __launch_bounds__(256) void k_raymarch_wave(const float* __restrict__ pool, const int64_t* __restrict__ ids,
                                                           const float* __restrict__ tf, const float* __restrict__ frame_feat, int ff,
                                                           int sh_degree, const uint32_t* __restrict__ bits, int n, int64_t R,
                                                           int max_hits, float* __restrict__ batch, float* __restrict__ rays_o_w,
                                                           float* __restrict__ viewdirs_w, float* __restrict__ view,
                                                           float* __restrict__ t_in_out, int32_t* __restrict__ cell_ids,
                                                           int32_t* __restrict__ n_hits, int32_t* __restrict__ flags, int cells_off,
                                                           NofSampleCfg cfg, const float* __restrict__ u_occ,
                                                           const float* __restrict__ u_dep, float* __restrict__ z_vals,
                                                           float* __restrict__ pts_w, uint8_t* __restrict__ valid) {
  extern __shared__ uint32_t occ_lds[];
  raymarch_wave_block<SAMPLE, true>(pool, ids, tf, frame_feat, ff, sh_degree, bits, n, R, max_hits, batch, rays_o_w, viewdirs_w, view,
                                    t_in_out, cell_ids, n_hits, flags, cells_off, cfg, u_occ, u_dep, z_vals, pts_w, valid, blockIdx.x,
                                    occ_lds);
}

// =====================================================================================================
// nof_adam_step_tail_march (round 6): the optimiser launch of step N (nof_adam_step_tail: pose sums, Adam, the next operand image and
// pose table) with the RAY MARCHER OF STEP N + 1 as one more role.  The marcher is a latency chain per ray (20 us for 4096 rays on a
// chip that Adam keeps busy with streaming for 37): side by side in one launch they take what the longer one takes.  The only thing
// the marcher needs from the optimiser is the pose table, a row per frame -- written by the launch's first F workgroups, which are
// resident before any other.  Each of them publishes its row with a release increment of a device counter (`epoch`, which only ever
// grows: the host passes the value it reaches when all F rows of THIS launch are out); a marcher workgroup's first thread waits for
// it -- bounded, a time-out raises flag bit 2 instead of hanging --, and an acquire fence makes the rows visible.
// MEASURED AND NOT THE DEFAULT (field.march_ahead = False): the merged launch takes 86-88 us where Adam's takes 37 and the marcher's
// 20 -- with or without the fences and the wait (profiles/r06_ak_march_x.txt) --: under Adam's streaming the marcher's dependent
// loads take four times as long, the same finding as round 3's side-stream prologue.  Kept as an option with its tests.
// =====================================================================================================
static size_t occ_lds_bytes(int level);
struct MarchArgs {
  const float* pool; const int64_t* ids; const uint32_t* bits; int n, sh_degree, max_hits, cells_off; int64_t R;
  float* batch; float* rays_o_w; float* viewdirs_w; float* view; float* t_in_out; int32_t* n_hits; int32_t* flags;
  NofSampleCfg cfg; float* z_vals; float* pts_w; uint8_t* valid;
};

template <class P>
__global__ __launch_bounds__(256) void k_adam_tail_march(NofMlpDesc d, float* __restrict__ p, float* __restrict__ g, float* __restrict__ m,
                                                          float* __restrict__ v, AdamK k, const int32_t* __restrict__ skip_flags,
                                                          TailArgs a, MarchArgs q, uint32_t n_march, uint32_t* __restrict__ epoch,
                                                          uint32_t target) {
  extern __shared__ uint32_t occ_lds[];
  const uint32_t F = (uint32_t)a.F, nb = gridDim.x - n_march;
  if (blockIdx.x < F) {                                                // (workgroup-uniform)
    adam_tail_roles<P>(d, p, g, m, v, k, skip_flags, a, blockIdx.x, nb);
    if (threadIdx.x == 0)                                              // (the thread that wrote the frame's row)
      __hip_atomic_fetch_add(epoch, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (blockIdx.x < F + n_march) {
    if (threadIdx.x == 0) {
      int spins = 0;
      while ((int32_t)(__hip_atomic_load(epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1 << 20)) {                                     // (~a second: never in a healthy launch)
          if (q.flags) atomicOr(&q.flags[0], 2);
          break;
        }
      }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    raymarch_wave_block<true, false>(q.pool, q.ids, a.tf, nullptr, 0, q.sh_degree, q.bits, q.n, q.R, q.max_hits, q.batch, q.rays_o_w,
                                     q.viewdirs_w, q.view, q.t_in_out, nullptr, q.n_hits, q.flags, q.cells_off, q.cfg, nullptr, nullptr,
                                     q.z_vals, q.pts_w, q.valid, blockIdx.x - F, occ_lds);
    return;
  }
  adam_tail_roles<P>(d, p, g, m, v, k, skip_flags, a, blockIdx.x - n_march, nb);
}

extern "C" int nof_adam_step_tail_march(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t n_basic,
                                         float lr, float lr_pose, float beta1, float beta2, float eps, int32_t step,
                                         const int32_t* skip_flags, const NofAdamTail* t, const NofMarchNext* nx, uint32_t* d_epoch,
                                         uint32_t epoch_target, void* stream) {
  TailArgs a;
  dim3 grid;
  NOF_ARG(step >= 1 && nx && d_epoch);
  if (int e = adam_tail_check(params, grads, exp_avg, exp_avg_sq, n, n_basic, t, &a, &grid)) return e;
  const NofSampleCfg* cfg = nx->cfg;
  // the next batch's ray marcher as nof_raymarch_sample would launch it in its one-launch form (and only that form)
  NOF_ARG(cfg && cfg->marcher == NOF_MARCHER_WAVE && cfg->d_step == nullptr && nx->level >= 0 && nx->level <= 6 &&
          nx->max_hits >= 1 && nx->max_hits <= NOF_TW_SLOTS && nx->R >= 1 && nx->sh_degree >= 1 && nx->sh_degree <= 4 &&
          nx->sh_degree * nx->sh_degree <= NOF_VIEW_COLS);
  NOF_ARG(nx->pool && nx->occ_bits && nx->batch && nx->rays_o_w && nx->viewdirs_w && nx->view && nx->t_in_out && nx->n_hits &&
          nx->z_vals && nx->pts_w && nx->valid);
  NOF_ARG(cfg->n_samples >= 2 && cfg->n_around >= 0 && cfg->n_around != 1 && cfg->n_samples + cfg->n_around <= 1024);
  const NofMlpDesc& d = *t->desc;
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  const AdamK k{(float)(lr / bc1), (float)(lr_pose / bc1), beta1, beta2, eps, (float)(1.0 / sqrt(bc2))};
  const size_t cells_off = (occ_lds_bytes(nx->level) + 15) & ~(size_t)15;
  const unsigned n_march = (unsigned)nof_div_up(nx->R, 4);
  MarchArgs q{nx->pool, nx->ids, nx->occ_bits, 1 << nx->level, nx->sh_degree, nx->max_hits, (int)cells_off, nx->R,
              nx->batch, nx->rays_o_w, nx->viewdirs_w, nx->view, nx->t_in_out, nx->n_hits, nx->flags, *cfg, nx->z_vals, nx->pts_w,
              nx->valid};
  grid.x += n_march;
#define NOF_TAIL(P)                                                                                                          \
  hipLaunchKernelGGL(k_adam_tail_march<P>, grid, dim3(256), cells_off + 4 * sizeof(WaveCells), (hipStream_t)stream, d, params,   \
                     grads, exp_avg, exp_avg_sq, k, skip_flags, a, q, n_march, d_epoch, epoch_target)
  if (d.precision == 0) NOF_TAIL(PrecF32);
  else if (is_bf16(d.precision)) NOF_TAIL(PrecBF16);
  else NOF_TAIL(PrecF16);
#undef NOF_TAIL
  NOF_LAUNCH_OK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
static size_t occ_lds_bytes(int level) {
  const size_t n = (size_t)1 << level;
  const size_t words = (n * n * n + 31) / 32;
  return words <= OCC_LDS_WORDS ? words * 4 : 0;
}

extern "C" int nof_occgrid_build(const int32_t* coords, int64_t P, int32_t max_level, int32_t level,
                                  uint32_t* occ_bits, void* stream) {
  NOF_ARG(occ_bits && level >= 0 && level <= 8 && max_level >= level && max_level <= 10 && P >= 0);
  const int n = 1 << level;
  const size_t words = ((size_t)n * n * n + 31) / 32;
  NOF_HIP(hipMemsetAsync(occ_bits, 0, words * 4, (hipStream_t)stream));
  if (P == 0) return 0;
  NOF_ARG(coords);
  hipLaunchKernelGGL(k_occ_build, dim3((unsigned)nof_div_up(P, 256)), dim3(256), 0, (hipStream_t)stream, coords, P,
                     max_level - level, n, occ_bits);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_occgrid_query(const uint32_t* occ_bits, int32_t level, const float* pts, uint8_t* inside,
                                  int64_t N, void* stream) {
  NOF_ARG(occ_bits && pts && inside && level >= 0 && level <= 8 && N >= 0);
  if (N == 0) return 0;
  hipLaunchKernelGGL(k_occ_query, dim3((unsigned)nof_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, occ_bits,
                     1 << level, pts, inside, N);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_trace_rays(const uint32_t* occ_bits, int32_t level, const float* rays_o, const float* rays_d, int64_t R,
                               int32_t max_hits, float* t_in_out, int32_t* cell_ids, int32_t* n_hits, int32_t* flags,
                               void* stream) {
  NOF_ARG(occ_bits && rays_o && rays_d && t_in_out && n_hits && level >= 0 && level <= 8 && max_hits >= 1 && R >= 0);
  if (R == 0) return 0;
  hipLaunchKernelGGL(k_trace_rays, dim3((unsigned)nof_div_up(R, 64)), dim3(64), occ_lds_bytes(level), (hipStream_t)stream, occ_bits,
                     1 << level, rays_o, rays_d, R, max_hits, t_in_out, cell_ids, n_hits, flags);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_batch_trace(const float* pool, const int64_t* ids, const float* tf, const float* frame_feat, int32_t ff,
                                int32_t sh_degree, const uint32_t* occ_bits, int32_t level, int64_t R, int32_t max_hits,
                                int32_t marcher, float* batch, float* rays_o_w, float* viewdirs_w, float* view, float* t_in_out,
                                int32_t* cell_ids, int32_t* n_hits, int32_t* flags, void* stream) {
  NOF_ARG(pool && tf && occ_bits && batch && rays_o_w && viewdirs_w && view && t_in_out && n_hits);
  NOF_ARG(marcher == NOF_MARCHER_WAVE || marcher == NOF_MARCHER_WALK);
  NOF_ARG(level >= 0 && level <= 8 && max_hits >= 1 && R >= 0 && sh_degree >= 1 && sh_degree <= 4);
  NOF_ARG(ff >= 0 && ff + sh_degree * sh_degree <= NOF_VIEW_COLS && (ff == 0 || frame_feat));
  if (R == 0) return 0;
  if (marcher == NOF_MARCHER_WAVE && level <= 6) {                      // one wave per ray (k_raymarch_wave)
    const size_t cells_off = (occ_lds_bytes(level) + 15) & ~(size_t)15;
    hipLaunchKernelGGL(k_raymarch_wave<false>, dim3((unsigned)nof_div_up(R, 4)), dim3(256), cells_off + 4 * sizeof(WaveCells),
                       (hipStream_t)stream, pool, ids, tf, frame_feat, ff, sh_degree, occ_bits, 1 << level, R, max_hits, batch,
                       rays_o_w, viewdirs_w, view, t_in_out, cell_ids, n_hits, flags, (int)cells_off, NofSampleCfg{}, nullptr, nullptr,
                       nullptr, nullptr, nullptr);
    NOF_LAUNCH_OK();
    return 0;
  }
  hipLaunchKernelGGL(k_batch_trace, dim3((unsigned)nof_div_up(R, 64)), dim3(64), occ_lds_bytes(level), (hipStream_t)stream, pool, ids, tf,
                     frame_feat, ff, sh_degree, occ_bits, 1 << level, R, max_hits, batch, rays_o_w, viewdirs_w, view,
                     t_in_out, cell_ids, n_hits, flags);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_sample_points(const NofSampleCfg* cfg, const float* batch, const float* tf, const float* t_in_out,
                                  const int32_t* n_hits, int64_t R, int32_t max_hits, const float* u_occ, const float* u_dep,
                                  float* z_vals, float* pts_w, uint8_t* valid, int32_t* flags, void* stream) {
  NOF_ARG(cfg && batch && tf && t_in_out && n_hits && z_vals && pts_w && valid && R >= 0 && max_hits >= 1);
  NOF_ARG(cfg->n_samples >= 2 && cfg->n_around >= 0 && cfg->n_around != 1);
  const int S = cfg->n_samples + cfg->n_around;
  NOF_ARG(S <= 1024);
  if (R == 0) return 0;
  const int threads = (int)nof_div_up(S, 64) * 64;
  const size_t shm = sizeof(float) * 2 * (size_t)max_hits;
  hipLaunchKernelGGL(k_sample_points, dim3((unsigned)R), dim3(threads), shm, (hipStream_t)stream, *cfg, batch, tf, t_in_out,
                     n_hits, max_hits, u_occ, u_dep, z_vals, pts_w, valid, flags);
  NOF_LAUNCH_OK();
  return 0;
}

// render_rays up to the sample points in ONE call (SURVEY.md 8b `nof_raymarch_sample`): ray gather + pose transform + SH +
// occupancy DDA (k_batch_trace) and stratified z sampling + point generation (k_sample_points), launched back to back.
extern "C" int nof_raymarch_sample(const NofSampleCfg* cfg, const float* pool, const int64_t* ids, const float* tf,
                                    const float* frame_feat, int32_t ff, int32_t sh_degree, const uint32_t* occ_bits,
                                    int32_t level, int64_t R, int32_t max_hits, const float* u_occ, const float* u_dep,
                                    float* batch, float* rays_o_w, float* viewdirs_w, float* view, float* t_in_out,
                                    int32_t* cell_ids, int32_t* n_hits, float* z_vals, float* pts_w, uint8_t* valid,
                                    int32_t* flags, void* stream) {
  if (cfg && cfg->marcher == NOF_MARCHER_WAVE && level >= 0 && level <= 6 && max_hits <= NOF_TW_SLOTS) {
    // both halves in ONE launch: the wave that enumerated a ray's cells places its samples (k_raymarch_wave<true>)
    NOF_ARG(pool && tf && occ_bits && batch && rays_o_w && viewdirs_w && view && t_in_out && n_hits && z_vals && pts_w && valid);
    NOF_ARG(max_hits >= 1 && R >= 0 && sh_degree >= 1 && sh_degree <= 4);
    NOF_ARG(ff >= 0 && ff + sh_degree * sh_degree <= NOF_VIEW_COLS && (ff == 0 || frame_feat));
    NOF_ARG(cfg->n_samples >= 2 && cfg->n_around >= 0 && cfg->n_around != 1 && cfg->n_samples + cfg->n_around <= 1024);
    if (R == 0) return 0;
    const size_t cells_off = (occ_lds_bytes(level) + 15) & ~(size_t)15;
    hipLaunchKernelGGL(k_raymarch_wave<true>, dim3((unsigned)nof_div_up(R, 4)), dim3(256), cells_off + 4 * sizeof(WaveCells),
                       (hipStream_t)stream, pool, ids, tf, frame_feat, ff, sh_degree, occ_bits, 1 << level, R, max_hits, batch,
                       rays_o_w, viewdirs_w, view, t_in_out, cell_ids, n_hits, flags, (int)cells_off, *cfg, u_occ, u_dep, z_vals,
                       pts_w, valid);
    NOF_LAUNCH_OK();
    return 0;
  }
  NOF_ARG(cfg);
  if (int e = nof_batch_trace(pool, ids, tf, frame_feat, ff, sh_degree, occ_bits, level, R, max_hits, cfg->marcher, batch, rays_o_w,
                              viewdirs_w, view, t_in_out, cell_ids, n_hits, flags, stream))
    return e;
  return nof_sample_points(cfg, batch, tf, t_in_out, n_hits, R, max_hits, u_occ, u_dep, z_vals, pts_w, valid, flags, stream);
}

// ================================================================================================================
// Ray-pool construction on the device: the per-keyframe ray table of
// NerfRunner.make_frame_rays (nerf_runner.py:246-316) + compute_near_far_and_filter_rays (:39-65) +
// ray_box_intersection_batch (nerf_helpers.py:403-446) + the octree-miss filter (:302-314) + the cloud-based depth
// denoise (:178-195).  The reference does all of it in float64 NumPy on the host for every pixel of every keyframe; here
// it is one pass per frame in float64 on the GPU (same operations, same order -- bundlesdf_amd/rays.py is the host
// restatement the tests compare against), a brute-force nearest-cloud-point test, and a stable compaction.
// ================================================================================================================
// does the ray hit any occupied cell (n_hits > 0 of the tracer, with its terminator / minimum-length filters)?
__device__ __forceinline__ bool trace_hits_any(const uint32_t* __restrict__ occ, const uint32_t* __restrict__ bits, int n,
                                               const float (&o)[3], const float (&d)[3]) {
  float tio[2];
  int overflow = 0;
  return trace_one(occ, bits, n, o, d, 1, tio, nullptr, &overflow) > 0;
}

// cv2.dilate with a k x k all-ones kernel (anchor k/2): window offsets -k/2 ... k-1-k/2, borders ignored; separable.
__global__ __launch_bounds__(256) void k_dilate_1d(const uint8_t* __restrict__ in, int H, int W, int k, int horizontal,
                                                    uint8_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)H * W) return;
  const int v = (int)(i / W), u = (int)(i % W);
  const int lo = -(k / 2), hi = k - 1 - k / 2;
  uint8_t m = 0;
  if (horizontal) {
    for (int d = lo; d <= hi; ++d) {
      const int uu = u + d;
      if (uu >= 0 && uu < W) { const uint8_t x = in[(int64_t)v * W + uu]; m = x > m ? x : m; }
    }
  } else {
    for (int d = lo; d <= hi; ++d) {
      const int vv = v + d;
      if (vv >= 0 && vv < H) { const uint8_t x = in[(int64_t)vv * W + u]; m = x > m ? x : m; }
    }
  }
  out[i] = m;
}

__device__ __forceinline__ void box_axis(double o, double inv, bool neg, double lo, double hi, double& tn, double& tf_) {
  const double near_plane = neg ? hi : lo, far_plane = neg ? lo : hi;      // nerf_helpers.py:414-420
  tn = (near_plane - o) * inv;
  if (tn < 0.0) tn = 0.0;
  tf_ = (far_plane - o) * inv;
}

__global__ __launch_bounds__(64) void k_frame_rays(NofFrameRaysCfg c, const float* __restrict__ image,
                                                    const float* __restrict__ depth, const uint8_t* __restrict__ mask_in,
                                                    const uint8_t* __restrict__ mask_sel, const uint8_t* __restrict__ occ_mask,
                                                    const uint32_t* __restrict__ bits, int n, int H, int W,
                                                    float* __restrict__ rows, uint8_t* __restrict__ keep) {
  extern __shared__ uint32_t occ_lds[];
  const uint32_t* occ = bits ? stage_occ(bits, n, occ_lds) : nullptr;
  const int64_t i = (int64_t)blockIdx.x * 64 + threadIdx.x;
  if (i >= (int64_t)H * W) return;
  const int v = (int)(i / W), u = (int)(i % W);
  const float d32 = depth[i];
  const bool min_ = mask_in[i] > 0;
  const bool invalid_depth = (((double)d32 < c.near_thr) || ((double)d32 > c.far_thr)) && min_;   // nerf_runner.py:268
  bool sel = mask_sel[i] > 0;
  if (occ_mask && occ_mask[i] > 0) sel = false;
  if (c.valid_depth_only && invalid_depth) sel = false;
  sel = sel && !invalid_depth;                                                     // ray type 0 only (:296)
  // get_camera_rays_np: float32 pixel grid, float64 intrinsics
  const double dx = ((double)(float)u - c.cx) / c.fx, dy = -((double)(float)v - c.cy) / c.fy, dz = -1.0;
  float* r = rows + i * NOF_RAY_COLS;
  r[0] = (float)dx; r[1] = (float)dy; r[2] = (float)dz;
  r[3] = image[i * 3]; r[4] = image[i * 3 + 1]; r[5] = image[i * 3 + 2];
  r[6] = d32; r[7] = min_ ? 1.0f : 0.0f; r[8] = (float)c.frame_id; r[9] = 0.0f;
  bool ok = sel;
  float nearf = 0.0f, farf = 0.0f;
  if (sel) {
    const double nrm = sqrt((dx * dx + dy * dy) + dz * dz);
    const double ux = dx / nrm, uy = dy / nrm, uz = dz / nrm;                      // dirs_unit (:44)
    const double* P = c.pose;
    double dw[3], ow[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      dw[a] = (P[a * 4] * dx + P[a * 4 + 1] * dy) + P[a * 4 + 2] * dz;            // cam_in_world[:3,:3] @ dir (:45)
      ow[a] = P[a * 4 + 3];
    }
    const double n2 = sqrt((dw[0] * dw[0] + dw[1] * dw[1]) + dw[2] * dw[2]) + 1e-10;   // nerf_helpers.py:408
    double inv[3];
    bool neg[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { inv[a] = 1.0 / (dw[a] / n2); neg[a] = inv[a] < 0.0; }
    double tmin, tmax, tymin, tymax, tzmin, tzmax;
    box_axis(ow[0], inv[0], neg[0], c.box_lo[0], c.box_hi[0], tmin, tmax);
    box_axis(ow[1], inv[1], neg[1], c.box_lo[1], c.box_hi[1], tymin, tymax);
    bool ishit = !((tmin > tymax) || (tymin > tmax));
    if (tymin > tmin) tmin = tymin;
    if (tymax < tmax) tmax = tymax;
    box_axis(ow[2], inv[2], neg[2], c.box_lo[2], c.box_hi[2], tzmin, tzmax);
    ishit = ishit && !((tmin > tzmax) || (tzmin > tmax));
    if (tzmin > tmin) tmin = tzmin;
    if (tzmax < tmax) tmax = tzmax;
    if (!ishit) { tmin = -1.0; tmax = -1.0; }
    ok = tmin >= 0.0;                                                              // nerf_runner.py:57
    nearf = (float)fabs(uz * tmin);                                                // |z| of the entry / exit point (:59-62)
    farf = (float)fabs(uz * tmax);
    if (ok && occ) {                                                               // rays that miss the octree (:302-314)
      const double wx = (P[0] * ux + P[1] * uy) + P[2] * uz, wy = (P[4] * ux + P[5] * uy) + P[6] * uz,
                   wz = (P[8] * ux + P[9] * uy) + P[10] * uz;
      const float o32[3] = {(float)ow[0], (float)ow[1], (float)ow[2]}, d32v[3] = {(float)wx, (float)wy, (float)wz};
      ok = trace_hits_any(occ, bits, n, o32, d32v);
    }
  }
  r[10] = nearf; r[11] = farf;
  keep[i] = ok ? 1 : 0;
}

// depth denoise (nerf_runner.py:178-195): a kept ray with mask > 0 and depth <= far whose back-projected point is farther
// than dist_thr from every cloud point is dropped.  Brute force over the (voxel-down-sampled, <= a few 10^4 points) cloud,
// staged through LDS in tiles; float64 like cKDTree.
__global__ __launch_bounds__(256) void k_cloud_filter(const float* __restrict__ rows, int64_t N, uint8_t* __restrict__ keep,
                                                       const double* __restrict__ poses, const double* __restrict__ cloud,
                                                       int64_t P, double far_thr, double dist_thr) {
  __shared__ double tile[256 * 3];
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool active = false;
  double pw[3] = {0.0, 0.0, 0.0};
  if (i < N && keep[i]) {
    const float* r = rows + i * NOF_RAY_COLS;
    if (r[7] > 0.0f && (double)r[6] <= far_thr) {
      active = true;
      const double dep = (double)r[6];
      const double px = (double)r[0] * dep, py = (double)r[1] * dep, pz = (double)r[2] * dep;
      const double* T = poses + (int64_t)r[8] * 16;
#pragma unroll
      for (int a = 0; a < 3; ++a) pw[a] = ((T[a * 4] * px + T[a * 4 + 1] * py) + T[a * 4 + 2] * pz) + T[a * 4 + 3];
    }
  }
  double best = 1e300;
  for (int64_t base = 0; base < P; base += 256) {
    const int64_t j = base + threadIdx.x;
    if (j < P) { tile[threadIdx.x * 3] = cloud[j * 3]; tile[threadIdx.x * 3 + 1] = cloud[j * 3 + 1]; tile[threadIdx.x * 3 + 2] = cloud[j * 3 + 2]; }
    __syncthreads();
    const int cnt = (int)((P - base) < 256 ? (P - base) : 256);
    if (active) {
      for (int t = 0; t < cnt; ++t) {
        const double ax = pw[0] - tile[t * 3], ay = pw[1] - tile[t * 3 + 1], az = pw[2] - tile[t * 3 + 2];
        const double d2 = (ax * ax + ay * ay) + az * az;
        best = d2 < best ? d2 : best;
      }
    }
    __syncthreads();
  }
  if (active && sqrt(best) > dist_thr) keep[i] = 0;
}

__global__ __launch_bounds__(256) void k_compact_rows(const float* __restrict__ rows, const uint8_t* __restrict__ keep,
                                                       const int64_t* __restrict__ offsets, int64_t N, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i = t / NOF_RAY_COLS;
  const int cidx = (int)(t % NOF_RAY_COLS);
  if (i >= N || !keep[i]) return;
  out[offsets[i] * NOF_RAY_COLS + cidx] = rows[t];
}

extern "C" int nof_mask_dilate(const uint8_t* mask, int32_t H, int32_t W, int32_t k, uint8_t* tmp, uint8_t* out, void* stream) {
  NOF_ARG(mask && tmp && out && H > 0 && W > 0 && k >= 1 && k <= 1024);
  const unsigned blocks = (unsigned)nof_div_up((int64_t)H * W, 256);
  hipLaunchKernelGGL(k_dilate_1d, dim3(blocks), dim3(256), 0, (hipStream_t)stream, mask, H, W, k, 1, tmp);
  hipLaunchKernelGGL(k_dilate_1d, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)tmp, H, W, k, 0, out);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_frame_rays(const NofFrameRaysCfg* cfg, const float* image, const float* depth, const uint8_t* mask_in,
                               const uint8_t* mask_sel, const uint8_t* occ_mask, const uint32_t* occ_bits, int32_t level,
                               int32_t H, int32_t W, float* rows, uint8_t* keep, void* stream) {
  NOF_ARG(cfg && image && depth && mask_in && mask_sel && rows && keep && H > 0 && W > 0 && level >= 0 && level <= 8);
  const unsigned blocks = (unsigned)nof_div_up((int64_t)H * W, 64);
  hipLaunchKernelGGL(k_frame_rays, dim3(blocks), dim3(64), occ_bits ? occ_lds_bytes(level) : 0, (hipStream_t)stream, *cfg, image,
                     depth, mask_in, mask_sel, occ_mask, occ_bits, 1 << level, (int)H, (int)W, rows, keep);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_cloud_filter(const float* rows, int64_t N, uint8_t* keep, const double* poses, const double* cloud, int64_t P,
                                 double far_thr, double dist_thr, void* stream) {
  NOF_ARG(N >= 0 && P >= 0);
  if (N == 0 || P == 0) return 0;
  NOF_ARG(rows && keep && poses && cloud);
  hipLaunchKernelGGL(k_cloud_filter, dim3((unsigned)nof_div_up(N, 256)), dim3(256), 0, (hipStream_t)stream, rows, N, keep, poses,
                     cloud, P, far_thr, dist_thr);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_compact_rows(const float* rows, const uint8_t* keep, const int64_t* offsets, int64_t N, float* out,
                                 void* stream) {
  NOF_ARG(N >= 0);
  if (N == 0) return 0;
  NOF_ARG(rows && keep && offsets && out);
  hipLaunchKernelGGL(k_compact_rows, dim3((unsigned)nof_div_up(N * NOF_RAY_COLS, 256)), dim3(256), 0, (hipStream_t)stream, rows,
                     keep, offsets, N, out);
  NOF_LAUNCH_OK();
  return 0;
}
