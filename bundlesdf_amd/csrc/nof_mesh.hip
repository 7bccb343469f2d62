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

struct MtCell {
  float f[8];
  int64_t id[8];
  uint32_t in;                                                        // bit c: corner c is inside (value < iso)
};

// corner c = (dx, dy, dz) = (c & 1, (c >> 1) & 1, c >> 2)
__device__ __forceinline__ bool mt_load(const float* __restrict__ vol, int nx, int ny, int nz, float iso, int64_t cell, MtCell& m) {
  const int cz = nz - 1, cy = ny - 1;
  const int k = (int)(cell % cz);
  const int64_t t = cell / cz;
  const int j = (int)(t % cy), i = (int)(t / cy);
  m.in = 0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const int64_t id = ((int64_t)(i + (c & 1)) * ny + (j + ((c >> 1) & 1))) * nz + (k + (c >> 2));
    m.id[c] = id;
    m.f[c] = vol[id];
    if (m.f[c] < iso) m.in |= 1u << c;
  }
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
    const int64_t a = m.id[ea[e]], b = m.id[eb[e]];
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
    for (int v = 0; v < 4; ++v) mt_point(m.id[c[v]], ny, nz, pos[v]);
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
    out[t] = m.id[kMcEdge[e][0]] * npts + m.id[kMcEdge[e][1]];        // the lower corner has the lower grid id
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
