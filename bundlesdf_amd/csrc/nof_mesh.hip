// Iso-surface extraction from the dense SDF grid on the GPU: marching cubes (nof_mc_*, further down: what the reference's
// skimage call computes) and marching tetrahedra (6 tetrahedra per cell around the main diagonal), each as three launches around
// two device-side primitives the host supplies (exclusive scan of the per-cell triangle counts, sort/unique of the edge keys):
//     nof_mt_count     per cell: number of triangles (0..12)
//     nof_mt_emit      per cell: its triangles as three EDGE KEYS each (key = lo * npts + hi of the edge's grid points),
//                      oriented so that the normal points from the tetrahedron's inside (value < iso) to its outside
//     nof_mt_vertices  per unique key: the vertex on that edge (float64 linear interpolation, index coordinates)
// Replaces skimage.measure.marching_cubes on the host (nerf_runner.py:1388-1394); bundlesdf_amd/mesh.py holds the same
// algorithm in numpy (the unit tests compare the two vertex for vertex).  Marching tetrahedra has no ambiguous cases and
// its vertices lie on grid edges by linear interpolation exactly like marching cubes, so the surfaces agree to well
// below a voxel -- the quantity the Chamfer parity metric measures.
#include "nof_common.h"

__device__ __constant__ int kTets[6][4] = {{0, 1, 3, 7}, {0, 1, 5, 7}, {0, 2, 3, 7}, {0, 2, 6, 7}, {0, 4, 5, 7}, {0, 4, 6, 7}};

typedef float mt_f32x8 __attribute__((ext_vector_type(8)));
struct MtCell {
  mt_f32x8 f;                                                         // corner values: ONE vector value (a run-time element pick is a register
  int64_t base, si, sj;                                               //  move; as float f[8] + int64 id[8] the record lived in scratch: 112 B)
  uint32_t in;                                                        // bit c: corner c is inside (value < iso)
  __device__ __forceinline__ int64_t id(int c) const { return base + (c & 1) * si + ((c >> 1) & 1) * sj + (c >> 2); }
};

// corner c = (dx, dy, dz) = (c & 1, (c >> 1) & 1, c >> 2)
__device__ __forceinline__ bool mt_load(const float* __restrict__ vol, int nx, int ny, int nz, float iso, int64_t cell, MtCell& m) {
  const int cz = nz - 1, cy = ny - 1;
  const int k = (int)(cell % cz);
  const int64_t t = cell / cz;
  const int j = (int)(t % cy), i = (int)(t / cy);
  m.in = 0;
  m.base = ((int64_t)i * ny + j) * nz + k;
  m.si = (int64_t)ny * nz;
  m.sj = nz;
  mt_f32x8 f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    f[c] = vol[m.id(c)];
    if (f[c] < iso) m.in |= 1u << c;
  }
  m.f = f;
  return m.in != 0u && m.in != 0xFFu;
}

__global__ __launch_bounds__(256) void k_mt_count(const float* __restrict__ vol, int nx, int ny, int nz, float iso,
                                                   int64_t ncell, int32_t* __restrict__ counts) {
  const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (cell >= ncell) return;
  MtCell m;
  int n = 0;
  if (mt_load(vol, nx, ny, nz, iso, cell, m)) {
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      int cnt = 0;
#pragma unroll
      for (int v = 0; v < 4; ++v) cnt += (m.in >> kTets[t][v]) & 1u;
      n += (cnt == 1 || cnt == 3) ? 1 : (cnt == 2 ? 2 : 0);
    }
  }
  counts[cell] = n;
}

__device__ __forceinline__ void mt_point(int64_t id, int ny, int nz, double (&p)[3]) {
  p[2] = (double)(id % nz);
  const int64_t t = id / nz;
  p[1] = (double)(t % ny);
  p[0] = (double)(t / ny);
}

// the vertex on edge (a, b): interpolation is done from the endpoint with the smaller id, as in mesh.py
__device__ __forceinline__ void mt_edge_vertex(int64_t a, int64_t b, float fa, float fb, double iso, int ny, int nz, double (&v)[3]) {
  if (a > b) { const int64_t t = a; a = b; b = t; const float s = fa; fa = fb; fb = s; }
  const double da = (double)fa, db = (double)fb;
  const double t = db != da ? (iso - da) / (db - da) : 0.5;
  double pa[3], pb[3];
  mt_point(a, ny, nz, pa);
  mt_point(b, ny, nz, pb);
#pragma unroll
  for (int d = 0; d < 3; ++d) v[d] = pa[d] + t * (pb[d] - pa[d]);
}

__device__ __forceinline__ void mt_write_tri(const MtCell& m, const int (&ea)[3], const int (&eb)[3], const double (&dir)[3],
                                             double iso, int ny, int nz, int64_t npts, int64_t* __restrict__ out) {
  double p[3][3];
  int64_t key[3];
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const int64_t a = m.id(ea[e]), b = m.id(eb[e]);
    mt_edge_vertex(a, b, m.f[ea[e]], m.f[eb[e]], iso, ny, nz, p[e]);
    key[e] = (a < b ? a : b) * npts + (a < b ? b : a);
  }
  const double u[3] = {p[1][0] - p[0][0], p[1][1] - p[0][1], p[1][2] - p[0][2]};
  const double w[3] = {p[2][0] - p[0][0], p[2][1] - p[0][1], p[2][2] - p[0][2]};
  const double n[3] = {u[1] * w[2] - u[2] * w[1], u[2] * w[0] - u[0] * w[2], u[0] * w[1] - u[1] * w[0]};
  const bool flip = (n[0] * dir[0] + n[1] * dir[1]) + n[2] * dir[2] < 0.0;
  out[0] = key[0];
  out[1] = flip ? key[2] : key[1];
  out[2] = flip ? key[1] : key[2];
}

__global__ __launch_bounds__(256) void k_mt_emit(const float* __restrict__ vol, int nx, int ny, int nz, float iso, int64_t ncell,
                                                  const int64_t* __restrict__ offsets, int64_t* __restrict__ keys) {
  const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (cell >= ncell) return;
  MtCell m;
  if (!mt_load(vol, nx, ny, nz, iso, cell, m)) return;
  const int64_t npts = (int64_t)nx * ny * nz;
  int64_t* out = keys + offsets[cell] * 3;
  const double isod = (double)iso;
  for (int t = 0; t < 6; ++t) {
    int c[4], cnt = 0;
    bool in[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      c[v] = kTets[t][v];
      in[v] = (m.in >> c[v]) & 1u;
      cnt += in[v] ? 1 : 0;
    }
    if (cnt == 0 || cnt == 4) continue;
    double pos[4][3];
#pragma unroll
    for (int v = 0; v < 4; ++v) mt_point(m.id(c[v]), ny, nz, pos[v]);
    if (cnt == 1 || cnt == 3) {                                       // one vertex on its own side -> one triangle
      int lone = 0;
#pragma unroll
      for (int v = 3; v >= 0; --v)
        if (in[v] == (cnt == 1)) lone = v;
      const int o1 = (lone + 1) & 3, o2 = (lone + 2) & 3, o3 = (lone + 3) & 3;
      double dir[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        dir[d] = (pos[o1][d] + pos[o2][d] + pos[o3][d]) / 3.0 - pos[lone][d];
        if (cnt == 3) dir[d] = -dir[d];
      }
      const int ea[3] = {c[lone], c[lone], c[lone]}, eb[3] = {c[o1], c[o2], c[o3]};
      mt_write_tri(m, ea, eb, dir, isod, ny, nz, npts, out);
      out += 3;
    } else {                                                          // two / two -> a quad (two triangles)
      int ins[2], outs[2], ni = 0, no = 0;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        if (in[v]) ins[ni++] = v;
        else outs[no++] = v;
      }
      double dir[3];
#pragma unroll
      for (int d = 0; d < 3; ++d) dir[d] = (pos[outs[0]][d] + pos[outs[1]][d]) / 2.0 - (pos[ins[0]][d] + pos[ins[1]][d]) / 2.0;
      const int i0 = c[ins[0]], i1 = c[ins[1]], q0 = c[outs[0]], q1 = c[outs[1]];
      {
        const int ea[3] = {i0, i0, i1}, eb[3] = {q0, q1, q1};
        mt_write_tri(m, ea, eb, dir, isod, ny, nz, npts, out);
      }
      {
        const int ea[3] = {i0, i1, i1}, eb[3] = {q0, q1, q0};
        mt_write_tri(m, ea, eb, dir, isod, ny, nz, npts, out + 3);
      }
      out += 6;
    }
  }
}

__global__ __launch_bounds__(256) void k_mt_vertices(const float* __restrict__ vol, int ny, int nz, int64_t npts, float iso,
                                                      const int64_t* __restrict__ keys, int64_t V, double* __restrict__ verts) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= V) return;
  const int64_t a = keys[i] / npts, b = keys[i] % npts;
  double v[3];
  mt_edge_vertex(a, b, vol[a], vol[b], (double)iso, ny, nz, v);
  verts[i * 3] = v[0];
  verts[i * 3 + 1] = v[1];
  verts[i * 3 + 2] = v[2];
}

// ---- marching cubes: the same three-launch scheme with a 256-row case table the HOST derives (bundlesdf_amd/mesh.py:
// mc_case_table; row = [T, 3 T cube-edge ids], edge e joins corners kMcEdge[e], corner c = x + 2 y + 4 z).  Keys, key order and the
// vertex rule are the ones of marching tetrahedra above, so nof_mt_vertices serves both.
__device__ __constant__ int kMcEdge[12][2] = {{0, 1}, {0, 2}, {0, 4}, {1, 3}, {1, 5}, {2, 3}, {2, 6}, {3, 7}, {4, 5}, {4, 6}, {5, 7}, {6, 7}};

__global__ __launch_bounds__(256) void k_mc_count(const float* __restrict__ vol, int nx, int ny, int nz, float iso,
                                                   const int8_t* __restrict__ table, int64_t ncell, int32_t* __restrict__ counts) {
  const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (cell >= ncell) return;
  MtCell m;
  mt_load(vol, nx, ny, nz, iso, cell, m);
  counts[cell] = table[m.in * 16];
}

__global__ __launch_bounds__(256) void k_mc_emit(const float* __restrict__ vol, int nx, int ny, int nz, float iso,
                                                  const int8_t* __restrict__ table, int64_t ncell,
                                                  const int64_t* __restrict__ offsets, int64_t* __restrict__ keys) {
  const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (cell >= ncell) return;
  MtCell m;
  if (!mt_load(vol, nx, ny, nz, iso, cell, m)) return;
  const int8_t* row = table + m.in * 16;
  const int n = row[0];
  const int64_t npts = (int64_t)nx * ny * nz;
  int64_t* out = keys + offsets[cell] * 3;
  for (int t = 0; t < 3 * n; ++t) {
    const int e = row[1 + t];
    out[t] = m.id(kMcEdge[e][0]) * npts + m.id(kMcEdge[e][1]);        // the lower corner has the lower grid id
  }
}

static int mt_dims_ok(int nx, int ny, int nz) { return nx >= 2 && ny >= 2 && nz >= 2 && nx <= 2048 && ny <= 2048 && nz <= 2048; }

extern "C" int nof_mt_count(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, int32_t* counts, void* stream) {
  NOF_ARG(vol && counts && mt_dims_ok(nx, ny, nz));
  const int64_t ncell = (int64_t)(nx - 1) * (ny - 1) * (nz - 1);
  hipLaunchKernelGGL(k_mt_count, dim3((unsigned)nof_div_up(ncell, 256)), dim3(256), 0, (hipStream_t)stream, vol, nx, ny, nz, iso,
                     ncell, counts);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_mt_emit(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, const int64_t* offsets,
                            int64_t* keys, void* stream) {
  NOF_ARG(vol && offsets && keys && mt_dims_ok(nx, ny, nz));
  const int64_t ncell = (int64_t)(nx - 1) * (ny - 1) * (nz - 1);
  hipLaunchKernelGGL(k_mt_emit, dim3((unsigned)nof_div_up(ncell, 256)), dim3(256), 0, (hipStream_t)stream, vol, nx, ny, nz, iso,
                     ncell, offsets, keys);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_mc_count(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, const int8_t* case_table,
                             int32_t* counts, void* stream) {
  NOF_ARG(vol && counts && case_table && mt_dims_ok(nx, ny, nz));
  const int64_t ncell = (int64_t)(nx - 1) * (ny - 1) * (nz - 1);
  hipLaunchKernelGGL(k_mc_count, dim3((unsigned)nof_div_up(ncell, 256)), dim3(256), 0, (hipStream_t)stream, vol, nx, ny, nz, iso,
                     case_table, ncell, counts);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_mc_emit(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, const int8_t* case_table,
                            const int64_t* offsets, int64_t* keys, void* stream) {
  NOF_ARG(vol && offsets && keys && case_table && mt_dims_ok(nx, ny, nz));
  const int64_t ncell = (int64_t)(nx - 1) * (ny - 1) * (nz - 1);
  hipLaunchKernelGGL(k_mc_emit, dim3((unsigned)nof_div_up(ncell, 256)), dim3(256), 0, (hipStream_t)stream, vol, nx, ny, nz, iso,
                     case_table, ncell, offsets, keys);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_mt_vertices(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, const int64_t* keys, int64_t V,
                                double* verts, void* stream) {
  NOF_ARG(vol && mt_dims_ok(nx, ny, nz) && V >= 0);
  if (V == 0) return 0;
  NOF_ARG(keys && verts);
  hipLaunchKernelGGL(k_mt_vertices, dim3((unsigned)nof_div_up(V, 256)), dim3(256), 0, (hipStream_t)stream, vol, ny, nz,
                     (int64_t)nx * ny * nz, iso, keys, V, verts);
  NOF_LAUNCH_OK();
  return 0;
}

// =====================================================================================================
// Marching cubes with Lewiner's topological disambiguation -- what skimage.measure.marching_cubes(volume, level) computes by default
// (method='lewiner'; the reference's call, nerf_runner.py:1388-1394): the 33 topological cases behind the 256 corner-sign
// configurations, chosen per cell by the paper's face tests and interior test (Lewiner, Lopes, Vieira, Tavares, JGT 8(2) 2003), with
// the paper's lookup tables handed in by the host as one packed int8 buffer (bundlesdf_amd/mesh.py:lewiner_lut_pack; table order =
// the enum below).  Same three-launch scheme and edge keys as above; a tiling's "edge 12" is the cell's CENTRE vertex, key
// -(cell + 1), placed like scikit-image places it (the corners weighted by 1 / (eps + |value - iso|)).  The tests run in double
// like scikit-image's.  oracle/marching_cubes_lewiner.py is the restatement this is checked against, itself pinned on scikit-image's
// outputs (tests/golden/mc_skimage_vectors.npz).
// Lewiner's cube: corner p at array offset kMclCorner[p] = (di, dj, dk) -- x = the last array axis --, edge e joins the corner pair mcl_write spells out.
// =====================================================================================================
enum { L_CASES, L_T1, L_T2, L_T3_1, L_T3_2, L_T4_1, L_T4_2, L_T5, L_T6_1_1, L_T6_1_2, L_T6_2, L_T7_1, L_T7_2, L_T7_3, L_T7_4_1, L_T7_4_2,
       L_T8, L_T9, L_T10_1_1, L_T10_1_1_, L_T10_1_2, L_T10_2, L_T10_2_, L_T11, L_T12_1_1, L_T12_1_1_, L_T12_1_2, L_T12_2, L_T12_2_,
       L_T13_1, L_T13_1_, L_T13_2, L_T13_2_, L_T13_3, L_T13_3_, L_T13_4, L_T13_5_1, L_T13_5_2, L_T14, L_TEST3, L_TEST4, L_TEST6,
       L_TEST7, L_TEST10, L_TEST12, L_TEST13, L_SUB13, L_COUNT };
static_assert(L_COUNT == NOF_MCL_TABLES, "table order of include/nof_hip.h");
__device__ __constant__ int kMclCorner[8][3] = {{0, 0, 0}, {0, 0, 1}, {0, 1, 1}, {0, 1, 0}, {1, 0, 0}, {1, 0, 1}, {1, 1, 1}, {1, 1, 0}};
// interior test: the reference edge (p, q) and the three edges parallel to it in the order the paper's code walks them
__device__ __constant__ int kMclPar[12][8] = {{0, 1, 3, 2, 7, 6, 4, 5}, {1, 2, 0, 3, 4, 7, 5, 6}, {2, 3, 1, 0, 5, 4, 6, 7}, {3, 0, 2, 1, 6, 5, 7, 4},
                                             {4, 5, 0, 1, 3, 2, 7, 6}, {5, 6, 1, 2, 0, 3, 4, 7}, {6, 7, 2, 3, 1, 0, 5, 4}, {7, 4, 3, 0, 2, 1, 6, 5},
                                             {0, 4, 3, 7, 2, 6, 1, 5}, {1, 5, 0, 4, 3, 7, 2, 6}, {2, 6, 1, 5, 0, 4, 3, 7}, {3, 7, 2, 6, 1, 5, 0, 4}};
#define MCL_EPS 2.220446049250313e-16   /* scikit-image's 'FLT_EPSILON' = np.spacing(1.0): the weights 1 / (eps + |value|) are an exact linear interpolation */

typedef double mcl_f64x8 __attribute__((ext_vector_type(8)));
struct MclCell {
  mcl_f64x8 c;                                                         // corner values minus the iso value, Lewiner's corner order
  int64_t base;                                                        // array index of corner 0; corner p sits at base + mcl_corner_off(p)
  int idx;                                                             // bit p: c_p > 0
};
// array offset of Lewiner's corner p from corner 0 (kMclCorner as arithmetic).  The corner values are ONE vector value (a run-time
// element pick of a register vector is a register move, not a memory access): round 5's record -- c[8], id[8] in private memory,
// picked by table entries -- cost 144 B of scratch (or 20 KB of LDS where the compiler promoted it); as eight named fields behind a
// select chain the compiler re-merged the loads into one load through a selected POINTER and kept the record in scratch
__device__ __forceinline__ int64_t mcl_corner_off(int p, int64_t si, int64_t sj) {
  return ((p & 4) ? si : 0) + (((p & 3) == 2 || (p & 3) == 3) ? sj : 0) + (((p & 3) == 1 || (p & 3) == 2) ? 1 : 0);
}
__device__ __forceinline__ bool mcl_load(const float* __restrict__ vol, int nx, int ny, int nz, float iso, int64_t cell, MclCell& m) {
  const int cz = nz - 1, cy = ny - 1;
  const int k = (int)(cell % cz);
  const int64_t t = cell / cz;
  const int j = (int)(t % cy), i = (int)(t / cy);
  // first the signs alone ((double)v - (double)iso > 0 <=> v > iso): 99 % of the cells of a dense grid end here
  const int64_t base = ((int64_t)i * ny + j) * nz + k;
  const int64_t sj = nz, si = (int64_t)ny * nz;
  const float v0 = vol[base], v1 = vol[base + 1], v2 = vol[base + sj + 1], v3 = vol[base + sj];
  const float v4 = vol[base + si], v5 = vol[base + si + 1], v6 = vol[base + si + sj + 1], v7 = vol[base + si + sj];
  const int idx = (v0 > iso ? 1 : 0) | (v1 > iso ? 2 : 0) | (v2 > iso ? 4 : 0) | (v3 > iso ? 8 : 0) | (v4 > iso ? 16 : 0) |
                  (v5 > iso ? 32 : 0) | (v6 > iso ? 64 : 0) | (v7 > iso ? 128 : 0);
  m.idx = idx;
  if (idx == 0 || idx == 0xFF) return false;
  m.base = base;
  const double d = (double)iso;
  const mcl_f64x8 c = {(double)v0 - d, (double)v1 - d, (double)v2 - d, (double)v3 - d, (double)v4 - d, (double)v5 - d, (double)v6 - d, (double)v7 - d};
  m.c = c;
  return true;
}
__device__ __forceinline__ double mcl_pick(const MclCell& m, int p) { return m.c[p & 7]; }
__device__ __forceinline__ bool mcl_test_face(const MclCell& m, int face) {
  const int f = face < 0 ? -face : face;
  const int a = f == 1 ? 0 : f == 2 ? 1 : f == 3 ? 2 : f == 4 ? 3 : f == 5 ? 0 : 4;
  const int b = f == 1 ? 4 : f == 2 ? 5 : f == 3 ? 6 : f == 4 ? 7 : f == 5 ? 3 : 7;
  const int c = f == 1 ? 5 : f == 2 ? 6 : f == 3 ? 7 : f == 4 ? 4 : f == 5 ? 2 : 6;
  const int d = f == 1 ? 1 : f == 2 ? 2 : f == 3 ? 3 : f == 4 ? 0 : f == 5 ? 1 : 5;
  const double A = mcl_pick(m, a), B = mcl_pick(m, b), C = mcl_pick(m, c), D = mcl_pick(m, d);
  return (double)face * A * (A * C - B * D) >= 0.0;                   // (no epsilon branch: scikit-image has none, see the oracle)
}
__device__ __forceinline__ bool mcl_test_interior(const MclCell& m, int s, int cs, int edge) {
  double At, Bt, Ct, Dt;
  if (cs == 4 || cs == 10) {
    const double c0 = m.c[0], c1 = m.c[1], c2 = m.c[2], c3 = m.c[3], c4 = m.c[4], c5 = m.c[5], c6 = m.c[6], c7 = m.c[7];
    const double a = (c4 - c0) * (c6 - c2) - (c7 - c3) * (c5 - c1);
    const double b = c2 * (c4 - c0) + c0 * (c6 - c2) - c1 * (c7 - c3) - c3 * (c5 - c1);
    if (a == 0.0) return s > 0;
    const double t = -b / (2.0 * a);
    if (!(t >= 0.0 && t <= 1.0)) return s > 0;
    At = c0 + (c4 - c0) * t;
    Bt = c3 + (c7 - c3) * t;
    Ct = c2 + (c6 - c2) * t;
    Dt = c1 + (c5 - c1) * t;
  } else {
    const int* q = kMclPar[edge];
    const double cp = mcl_pick(m, q[0]), cq = mcl_pick(m, q[1]);
    const double t = cp / (cp - cq);
    At = 0.0;
    Bt = mcl_pick(m, q[2]) + (mcl_pick(m, q[3]) - mcl_pick(m, q[2])) * t;
    Ct = mcl_pick(m, q[4]) + (mcl_pick(m, q[5]) - mcl_pick(m, q[4])) * t;
    Dt = mcl_pick(m, q[6]) + (mcl_pick(m, q[7]) - mcl_pick(m, q[6])) * t;
  }
  const int test = (At >= 0.0 ? 1 : 0) + (Bt >= 0.0 ? 2 : 0) + (Ct >= 0.0 ? 4 : 0) + (Dt >= 0.0 ? 8 : 0);
  switch (test) {
    case 7: case 11: case 13: case 14: case 15: return s < 0;
    case 5: return At * Ct - Bt * Dt < 0.0 && s > 0;                   // (scikit-image's form of the paper's determinant test: against
    case 10: return At * Ct - Bt * Dt >= 0.0 && s > 0;                 //  zero, and false whatever the sign of s -- see the oracle)
    default: return s > 0;
  }
}
// the cell's tiling: pointer to its row of 3 * ntri edge ids (12 = the centre vertex), ntri = 0 for a cell without surface
__device__ __forceinline__ const int8_t* mcl_tiling(const MclCell& m, const int8_t* __restrict__ L, const NofMclLuts& O, int& ntri) {
#define TAB(id) (L + O.off[id])
  const int cs = TAB(L_CASES)[2 * m.idx], cfg = TAB(L_CASES)[2 * m.idx + 1];
  switch (cs) {
    case 1: ntri = 1; return TAB(L_T1) + cfg * 3;
    case 2: ntri = 2; return TAB(L_T2) + cfg * 6;
    case 3:
      if (mcl_test_face(m, TAB(L_TEST3)[cfg])) { ntri = 4; return TAB(L_T3_2) + cfg * 12; }
      ntri = 2; return TAB(L_T3_1) + cfg * 6;
    case 4:
      if (mcl_test_interior(m, TAB(L_TEST4)[cfg], 4, 0)) { ntri = 2; return TAB(L_T4_1) + cfg * 6; }
      ntri = 6; return TAB(L_T4_2) + cfg * 18;
    case 5: ntri = 3; return TAB(L_T5) + cfg * 9;
    case 6: {
      const int8_t* t = TAB(L_TEST6) + cfg * 3;
      if (mcl_test_face(m, t[0])) { ntri = 5; return TAB(L_T6_2) + cfg * 15; }
      if (mcl_test_interior(m, t[1], 6, t[2])) { ntri = 3; return TAB(L_T6_1_1) + cfg * 9; }
      ntri = 9; return TAB(L_T6_1_2) + cfg * 27;
    }
    case 7: {
      const int8_t* t = TAB(L_TEST7) + cfg * 5;
      const int sub = (mcl_test_face(m, t[0]) ? 1 : 0) + (mcl_test_face(m, t[1]) ? 2 : 0) + (mcl_test_face(m, t[2]) ? 4 : 0);
      switch (sub) {
        case 0: ntri = 3; return TAB(L_T7_1) + cfg * 9;
        case 1: ntri = 5; return TAB(L_T7_2) + (cfg * 3 + 0) * 15;
        case 2: ntri = 5; return TAB(L_T7_2) + (cfg * 3 + 1) * 15;
        case 4: ntri = 5; return TAB(L_T7_2) + (cfg * 3 + 2) * 15;
        case 3: ntri = 9; return TAB(L_T7_3) + (cfg * 3 + 0) * 27;
        case 5: ntri = 9; return TAB(L_T7_3) + (cfg * 3 + 1) * 27;
        case 6: ntri = 9; return TAB(L_T7_3) + (cfg * 3 + 2) * 27;
        default:
          if (mcl_test_interior(m, t[3], 7, t[4])) { ntri = 9; return TAB(L_T7_4_2) + cfg * 27; }
          ntri = 5; return TAB(L_T7_4_1) + cfg * 15;
      }
    }
    case 8: ntri = 2; return TAB(L_T8) + cfg * 6;
    case 9: ntri = 4; return TAB(L_T9) + cfg * 12;
    case 10: {
      const int8_t* t = TAB(L_TEST10) + cfg * 3;
      if (mcl_test_face(m, t[0])) {
        if (mcl_test_face(m, t[1])) { ntri = 4; return TAB(L_T10_1_1_) + cfg * 12; }
        ntri = 8; return TAB(L_T10_2) + cfg * 24;
      }
      if (mcl_test_face(m, t[1])) { ntri = 8; return TAB(L_T10_2_) + cfg * 24; }
      if (mcl_test_interior(m, t[2], 10, 0)) { ntri = 4; return TAB(L_T10_1_1) + cfg * 12; }
      ntri = 8; return TAB(L_T10_1_2) + cfg * 24;
    }
    case 11: ntri = 4; return TAB(L_T11) + cfg * 12;
    case 12: {
      const int8_t* t = TAB(L_TEST12) + cfg * 4;
      if (mcl_test_face(m, t[0])) {
        if (mcl_test_face(m, t[1])) { ntri = 4; return TAB(L_T12_1_1_) + cfg * 12; }
        ntri = 8; return TAB(L_T12_2) + cfg * 24;
      }
      if (mcl_test_face(m, t[1])) { ntri = 8; return TAB(L_T12_2_) + cfg * 24; }
      if (mcl_test_interior(m, t[2], 12, t[3])) { ntri = 4; return TAB(L_T12_1_1) + cfg * 12; }
      ntri = 8; return TAB(L_T12_1_2) + cfg * 24;
    }
    case 13: {
      const int8_t* t = TAB(L_TEST13) + cfg * 7;
      int sub = 0;
#pragma unroll
      for (int b = 0; b < 6; ++b) sub |= mcl_test_face(m, t[b]) ? (1 << b) : 0;
      const int k = TAB(L_SUB13)[sub];
      if (k == 0) { ntri = 4; return TAB(L_T13_1) + cfg * 12; }
      if (k <= 6) { ntri = 6; return TAB(L_T13_2) + (cfg * 6 + (k - 1)) * 18; }
      if (k <= 18) { ntri = 10; return TAB(L_T13_3) + (cfg * 12 + (k - 7)) * 30; }
      if (k <= 22) { ntri = 12; return TAB(L_T13_4) + (cfg * 4 + (k - 19)) * 36; }
      if (k <= 26) {
        const int sc = k - 23;
        const int8_t* r51 = TAB(L_T13_5_1) + (cfg * 4 + sc) * 18;
        if (mcl_test_interior(m, t[6], 13, r51[0])) { ntri = 6; return r51; }
        ntri = 10; return TAB(L_T13_5_2) + (cfg * 4 + sc) * 30;
      }
      if (k <= 38) { ntri = 10; return TAB(L_T13_3_) + (cfg * 12 + (k - 27)) * 30; }
      if (k <= 44) { ntri = 6; return TAB(L_T13_2_) + (cfg * 6 + (k - 39)) * 18; }
      ntri = 4; return TAB(L_T13_1_) + cfg * 12;
    }
    case 14: ntri = 4; return TAB(L_T14) + cfg * 12;
    default: ntri = 0; return L;
  }
#undef TAB
}

__global__ __launch_bounds__(256) void k_mcl_count(const float* __restrict__ vol, int nx, int ny, int nz, float iso,
                                                    const int8_t* __restrict__ luts, NofMclLuts offs, int64_t ncell,
                                                    int32_t* __restrict__ counts) {
  const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (cell >= ncell) return;
  MclCell m;
  int n = 0;
  if (mcl_load(vol, nx, ny, nz, iso, cell, m)) (void)mcl_tiling(m, luts, offs, n);
  counts[cell] = n;
}

// the n triangles of a cell's tiling as vertex keys
__device__ __forceinline__ void mcl_write(const MclCell& m, const int8_t* __restrict__ row, int n, int64_t cell, int ny, int nz,
                                          int64_t npts, int64_t* __restrict__ out) {
  const int64_t sj = nz, si = (int64_t)ny * nz;
  for (int t = 0; t < n; ++t) {
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const int e = row[3 * t + v];
      int64_t key = -(cell + 1);                                       // the centre vertex of the cell
      if (e < 12) {
        // Lewiner's edge e joins corners (e, e + 1 mod 4) on the bottom face, (e, 4 + (e + 1) mod 4) on the top, (e - 8, e - 4) upright
        const int p0 = e < 8 ? e : e - 8, p1 = e < 4 ? ((e + 1) & 3) : e < 8 ? 4 + ((e + 1) & 3) : e - 4;
        const int64_t a = m.base + mcl_corner_off(p0, si, sj), b = m.base + mcl_corner_off(p1, si, sj);
        key = a < b ? a * npts + b : b * npts + a;
      }
      out[3 * t + (2 - v)] = key;                                      // scikit-image's default winding (gradient_direction='descent')
    }
  }
}

__global__ __launch_bounds__(256) void k_mcl_emit(const float* __restrict__ vol, int nx, int ny, int nz, float iso,
                                                   const int8_t* __restrict__ luts, NofMclLuts offs, int64_t ncell,
                                                   const int64_t* __restrict__ offsets, int64_t* __restrict__ keys) {
  const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (cell >= ncell) return;
  MclCell m;
  if (!mcl_load(vol, nx, ny, nz, iso, cell, m)) return;
  int n = 0;
  const int8_t* row = mcl_tiling(m, luts, offs, n);
  mcl_write(m, row, n, cell, ny, nz, (int64_t)nx * ny * nz, keys + offsets[cell] * 3);
}

// ---- the two-level form (round 6): at 512^3 the per-cell counts (0.5 GB), their 64-bit inclusive scan (1 GB) and the exclusive offsets
//      (1 GB) cost 2.1 of the extraction's 4.9 ms in torch launches -- for 99.5 % cells that emit nothing.  Here the first launch
//      leaves ONE number per 256 cells (the workgroup's triangle count), the host scans those (0.5 M entries), and the second launch
//      -- workgroups without surface return at once -- recomputes its cells' tilings and places them with a workgroup-local scan.
__device__ __forceinline__ int mcl_block_scan(int n, int* total) {       // exclusive prefix of n over the workgroup's 256 threads
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = n;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d, 64);
    if (lane >= d) incl += up;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int before = 0, all = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) { before += w < wave ? wsum[w] : 0; all += wsum[w]; }
  *total = all;
  return before + incl - n;
}

__global__ __launch_bounds__(256) void k_mcl_count_blocks(const float* __restrict__ vol, int nx, int ny, int nz, float iso,
                                                           const int8_t* __restrict__ luts, NofMclLuts offs, int64_t ncell,
                                                           int32_t* __restrict__ block_counts) {
  const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
  MclCell m;
  int n = 0;
  if (cell < ncell && mcl_load(vol, nx, ny, nz, iso, cell, m)) (void)mcl_tiling(m, luts, offs, n);
  if (__syncthreads_or(n) == 0) {                                      // (the common case: no cell of the workgroup meets the surface)
    if (threadIdx.x == 0) block_counts[blockIdx.x] = 0;
    return;
  }
  int total;
  (void)mcl_block_scan(n, &total);
  if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
}

__global__ __launch_bounds__(256) void k_mcl_emit_blocks(const float* __restrict__ vol, int nx, int ny, int nz, float iso,
                                                          const int8_t* __restrict__ luts, NofMclLuts offs, int64_t ncell,
                                                          const int64_t* __restrict__ block_end, int64_t* __restrict__ keys) {
  // block_end = the INCLUSIVE scan of block_counts: this workgroup's triangles are [block_end[b - 1], block_end[b])
  const int64_t first = blockIdx.x == 0 ? 0 : block_end[blockIdx.x - 1];
  if (block_end[blockIdx.x] == first) return;                          // (uniform)
  const int64_t cell = (int64_t)blockIdx.x * 256 + threadIdx.x;
  MclCell m;
  int n = 0;
  const int8_t* row = luts;
  if (cell < ncell && mcl_load(vol, nx, ny, nz, iso, cell, m)) row = mcl_tiling(m, luts, offs, n);
  int total;
  const int before = mcl_block_scan(n, &total);
  if (n > 0) mcl_write(m, row, n, cell, ny, nz, (int64_t)nx * ny * nz, keys + (first + before) * 3);
}

__global__ __launch_bounds__(256) void k_mcl_vertices(const float* __restrict__ vol, int nx, int ny, int nz, float iso,
                                                       const int64_t* __restrict__ keys, int64_t V, double* __restrict__ verts) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= V) return;
  const int64_t npts = (int64_t)nx * ny * nz;
  double v[3];
  if (keys[i] >= 0) {
    // scikit-image's edge vertex: the two grid points weighted by 1 / (eps + |value - iso|) -- linear interpolation up to eps
    // (the two part by 1e-5 ... 1e-4 voxel where a corner value is within 1e-3 of the iso value; this extractor IS skimage's)
    const int64_t a = keys[i] / npts, b = keys[i] % npts;
    const double wa = 1.0 / (MCL_EPS + fabs((double)vol[a] - (double)iso)), wb = 1.0 / (MCL_EPS + fabs((double)vol[b] - (double)iso));
    double pa[3], pb[3];
    mt_point(a, ny, nz, pa);
    mt_point(b, ny, nz, pb);
#pragma unroll
    for (int d = 0; d < 3; ++d) v[d] = pa[d] + (pb[d] - pa[d]) * (wb / (wa + wb));
  } else {
    // scikit-image's centre vertex: the cell's corners weighted by 1 / (eps + |value - iso|)
    const int64_t cell = -(keys[i] + 1);
    const int cz = nz - 1, cy = ny - 1;
    const int k = (int)(cell % cz);
    const int64_t t = cell / cz;
    const int j = (int)(t % cy), ii = (int)(t / cy);
    double w = 0.0, f[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int64_t id = ((int64_t)(ii + kMclCorner[p][0]) * ny + (j + kMclCorner[p][1])) * nz + (k + kMclCorner[p][2]);
      const double wp = 1.0 / (MCL_EPS + fabs((double)vol[id] - (double)iso));
      w += wp;
#pragma unroll
      for (int d = 0; d < 3; ++d) f[d] += wp * (double)kMclCorner[p][d];
    }
    v[0] = (double)ii + f[0] / w;
    v[1] = (double)j + f[1] / w;
    v[2] = (double)k + f[2] / w;
  }
  verts[i * 3] = v[0];
  verts[i * 3 + 1] = v[1];
  verts[i * 3 + 2] = v[2];
}

extern "C" int nof_mcl_count(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, const int8_t* luts, const NofMclLuts* offs,
                              int32_t* counts, void* stream) {
  NOF_ARG(vol && counts && luts && offs && mt_dims_ok(nx, ny, nz));
  const int64_t ncell = (int64_t)(nx - 1) * (ny - 1) * (nz - 1);
  hipLaunchKernelGGL(k_mcl_count, dim3((unsigned)nof_div_up(ncell, 256)), dim3(256), 0, (hipStream_t)stream, vol, nx, ny, nz, iso,
                     luts, *offs, ncell, counts);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_mcl_emit(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, const int8_t* luts, const NofMclLuts* offs,
                             const int64_t* offsets, int64_t* keys, void* stream) {
  NOF_ARG(vol && offsets && keys && luts && offs && mt_dims_ok(nx, ny, nz));
  const int64_t ncell = (int64_t)(nx - 1) * (ny - 1) * (nz - 1);
  hipLaunchKernelGGL(k_mcl_emit, dim3((unsigned)nof_div_up(ncell, 256)), dim3(256), 0, (hipStream_t)stream, vol, nx, ny, nz, iso,
                     luts, *offs, ncell, offsets, keys);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_mcl_count_blocks(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, const int8_t* luts,
                                     const NofMclLuts* offs, int32_t* block_counts, void* stream) {
  NOF_ARG(vol && block_counts && luts && offs && mt_dims_ok(nx, ny, nz));
  const int64_t ncell = (int64_t)(nx - 1) * (ny - 1) * (nz - 1);
  hipLaunchKernelGGL(k_mcl_count_blocks, dim3((unsigned)nof_div_up(ncell, 256)), dim3(256), 0, (hipStream_t)stream, vol, nx, ny, nz, iso,
                     luts, *offs, ncell, block_counts);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_mcl_emit_blocks(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, const int8_t* luts,
                                    const NofMclLuts* offs, const int64_t* block_end, int64_t* keys, void* stream) {
  NOF_ARG(vol && block_end && keys && luts && offs && mt_dims_ok(nx, ny, nz));
  const int64_t ncell = (int64_t)(nx - 1) * (ny - 1) * (nz - 1);
  hipLaunchKernelGGL(k_mcl_emit_blocks, dim3((unsigned)nof_div_up(ncell, 256)), dim3(256), 0, (hipStream_t)stream, vol, nx, ny, nz, iso,
                     luts, *offs, ncell, block_end, keys);
  NOF_LAUNCH_OK();
  return 0;
}

extern "C" int nof_mcl_vertices(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, const int64_t* keys, int64_t V,
                                 double* verts, void* stream) {
  NOF_ARG(vol && mt_dims_ok(nx, ny, nz) && V >= 0);
  if (V == 0) return 0;
  NOF_ARG(keys && verts);
  hipLaunchKernelGGL(k_mcl_vertices, dim3((unsigned)nof_div_up(V, 256)), dim3(256), 0, (hipStream_t)stream, vol, nx, ny, nz, iso, keys,
                     V, verts);
  NOF_LAUNCH_OK();
  return 0;
}

// ---- texture bake helper: UV of every ray/mesh hit (replaces common.rayColorToTextureImageCUDA, common.cu:171-238) -------
// Barycentric weights of the hit point in its triangle from signed-area ratios projected on the triangle normal
// (w0 = [P,B,C]/[A,B,C], w1 = [P,C,A]/[A,B,C], w2 = 1 - w0 - w1), then the vertices' texture coordinates blended with them.
__device__ __forceinline__ void cross3(const float (&a)[3], const float (&b)[3], float (&o)[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
__device__ __forceinline__ float dot3(const float (&a)[3], const float (&b)[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

__global__ __launch_bounds__(256) void k_bary_uv(const int64_t* __restrict__ faces, const float* __restrict__ verts,
                                                  const float* __restrict__ hit_locations, const int64_t* __restrict__ hit_face_ids,
                                                  const float* __restrict__ uvs_tex, int64_t n_hits, float* __restrict__ uvs) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_hits) return;
  const int64_t* f = faces + hit_face_ids[i] * 3;
  float A[3], Bv[3], Cv[3], p[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    A[c] = verts[f[0] * 3 + c];
    Bv[c] = verts[f[1] * 3 + c];
    Cv[c] = verts[f[2] * 3 + c];
    p[c] = hit_locations[i * 3 + c];
  }
  float bc[3], ba[3], ca[3], pb[3], pc[3], pa[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    bc[c] = Bv[c] - Cv[c]; ba[c] = Bv[c] - A[c]; ca[c] = Cv[c] - A[c];
    pb[c] = Bv[c] - p[c]; pc[c] = Cv[c] - p[c]; pa[c] = A[c] - p[c];
  }
  float nrm[3], t[3];
  cross3(bc, ba, nrm);
  cross3(ba, ca, t);
  const float area = dot3(nrm, t);
  cross3(pb, pc, t);
  const float w0 = dot3(nrm, t) / area;
  cross3(pc, pa, t);
  const float w1 = dot3(nrm, t) / area;
  const float w2 = 1.0f - w0 - w1;
#pragma unroll
  for (int c = 0; c < 2; ++c)
    uvs[i * 2 + c] = (uvs_tex[f[0] * 2 + c] * w0 + uvs_tex[f[1] * 2 + c] * w1) + uvs_tex[f[2] * 2 + c] * w2;
}

extern "C" int nof_bary_uv(const int64_t* faces, const float* verts, const float* hit_locations, const int64_t* hit_face_ids,
                            const float* uvs_tex, int64_t n_hits, float* uvs, void* stream) {
  NOF_ARG(n_hits >= 0);
  if (n_hits == 0) return 0;
  NOF_ARG(faces && verts && hit_locations && hit_face_ids && uvs_tex && uvs);
  hipLaunchKernelGGL(k_bary_uv, dim3((unsigned)nof_div_up(n_hits, 256)), dim3(256), 0, (hipStream_t)stream, faces, verts,
                     hit_locations, hit_face_ids, uvs_tex, n_hits, uvs);
  NOF_LAUNCH_OK();
  return 0;
}
