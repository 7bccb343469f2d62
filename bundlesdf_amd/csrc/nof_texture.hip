// Texture bake from the training images on the device (NerfRunner.mesh_texture_from_train_images, nerf_runner.py:1468-1542).
// The reference renders the mesh's depth with pyrender (EGL), back-projects it, asks trimesh for the closest surface point and
// triangle of every pixel, interpolates the UV of the hit (common.cu:171-238) and lets every texel take ONE colour per frame.
// Here, per keyframe, three launches over resident buffers:
//   k_tex_raster   a z-buffer rasteriser: thread = triangle, 64-bit atomicMin of (depth bits << 32 | triangle id) per pixel --
//                  what the offscreen render + closest_point pair computes for a closed mesh (visible point and its triangle);
//   k_tex_owner    thread = pixel: surface point = pixel ray x triangle plane, barycentric UV (the formula of
//                  calculateBarycentricCoordinate3DKernel, common.cu:171-185), rounded texel; atomicMin of the pixel index per
//                  texel = the first pixel in row-major order (nerf_runner.py:1527-1531 keeps one colour per texel and frame);
//   k_tex_accum    the owning pixels add their raw colour and weight 1 (nerf_runner.py:1533-1535).
#include "nof_common.h"
#pragma clang fp contract(off)

struct TexCam {
  float R[9], t[3];              // object(normalised) -> OpenCV camera:  p_cam = R p + t
  float fx, fy, cx, cy;
  int H, W;
};

__global__ __launch_bounds__(256) void k_tex_raster(TexCam c, const float* __restrict__ verts, const int64_t* __restrict__ faces,
                                                     int64_t n_faces, unsigned long long* __restrict__ zbuf) {
  const int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (f >= n_faces) return;
  float u[3], v[3], z[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float* p = verts + faces[f * 3 + k] * 3;
    const float X = (c.R[0] * p[0] + c.R[1] * p[1]) + c.R[2] * p[2] + c.t[0];
    const float Y = (c.R[3] * p[0] + c.R[4] * p[1]) + c.R[5] * p[2] + c.t[1];
    const float Z = (c.R[6] * p[0] + c.R[7] * p[1]) + c.R[8] * p[2] + c.t[2];
    if (Z < 1e-6f) return;                                         // behind the camera: no clipping (the object is in front)
    u[k] = c.fx * X / Z + c.cx;
    v[k] = c.fy * Y / Z + c.cy;
    z[k] = Z;
  }
  const float area = (u[1] - u[0]) * (v[2] - v[0]) - (u[2] - u[0]) * (v[1] - v[0]);
  if (area == 0.0f) return;
  int x0 = (int)ceilf(fminf(fminf(u[0], u[1]), u[2])), x1 = (int)floorf(fmaxf(fmaxf(u[0], u[1]), u[2]));
  int y0 = (int)ceilf(fminf(fminf(v[0], v[1]), v[2])), y1 = (int)floorf(fmaxf(fmaxf(v[0], v[1]), v[2]));
  x0 = x0 < 0 ? 0 : x0; y0 = y0 < 0 ? 0 : y0;
  x1 = x1 >= c.W ? c.W - 1 : x1; y1 = y1 >= c.H ? c.H - 1 : y1;
  if (x1 - x0 > 4096 || y1 - y0 > 4096) return;
  const float inv = 1.0f / area;
  for (int py = y0; py <= y1; ++py)
    for (int px = x0; px <= x1; ++px) {
      const float fxp = (float)px, fyp = (float)py;
      const float w0 = ((u[1] - fxp) * (v[2] - fyp) - (u[2] - fxp) * (v[1] - fyp)) * inv;
      const float w1 = ((u[2] - fxp) * (v[0] - fyp) - (u[0] - fxp) * (v[2] - fyp)) * inv;
      const float w2 = 1.0f - w0 - w1;
      if (w0 < 0.0f || w1 < 0.0f || w2 < 0.0f) continue;
      const float zi = 1.0f / (w0 / z[0] + w1 / z[1] + w2 / z[2]);  // perspective-correct depth
      const unsigned long long key = ((unsigned long long)__float_as_uint(zi) << 32) | (unsigned long long)(uint32_t)f;
      atomicMin(&zbuf[(size_t)py * c.W + px], key);
    }
}

// texel of a pixel's visible surface point, or -1
__device__ __forceinline__ int64_t pixel_texel(const TexCam& c, const float* __restrict__ verts, const int64_t* __restrict__ faces,
                                               const float* __restrict__ uvs_tex, const unsigned long long* __restrict__ zbuf,
                                               const uint8_t* __restrict__ mask, float min_depth, int tex_w, int64_t pix) {
  const unsigned long long key = zbuf[pix];
  if (key == ~0ull || mask[pix] == 0) return -1;
  const float depth = __uint_as_float((uint32_t)(key >> 32));
  if (!(depth >= min_depth)) return -1;                             // nerf_runner.py:1506
  const int64_t f = (int64_t)(uint32_t)key;
  const int px = (int)(pix % c.W), py = (int)(pix / c.W);
  // camera ray in the object frame: o = -R^T t, d = R^T ((px-cx)/fx, (py-cy)/fy, 1)
  const float dc[3] = {((float)px - c.cx) / c.fx, ((float)py - c.cy) / c.fy, 1.0f};
  float o[3], d[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    o[k] = -((c.R[k] * c.t[0] + c.R[3 + k] * c.t[1]) + c.R[6 + k] * c.t[2]);
    d[k] = (c.R[k] * dc[0] + c.R[3 + k] * dc[1]) + c.R[6 + k] * dc[2];
  }
  float A[3], Bv[3], Cv[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    A[k] = verts[faces[f * 3] * 3 + k];
    Bv[k] = verts[faces[f * 3 + 1] * 3 + k];
    Cv[k] = verts[faces[f * 3 + 2] * 3 + k];
  }
  float e1[3], e2[3], n[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { e1[k] = Bv[k] - A[k]; e2[k] = Cv[k] - A[k]; }
  n[0] = e1[1] * e2[2] - e1[2] * e2[1]; n[1] = e1[2] * e2[0] - e1[0] * e2[2]; n[2] = e1[0] * e2[1] - e1[1] * e2[0];
  const float den = (n[0] * d[0] + n[1] * d[1]) + n[2] * d[2];
  if (den == 0.0f) return -1;
  const float tt = ((n[0] * (A[0] - o[0]) + n[1] * (A[1] - o[1])) + n[2] * (A[2] - o[2])) / den;
  float p[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) p[k] = o[k] + tt * d[k];
  // barycentric weights exactly as calculateBarycentricCoordinate3DKernel (common.cu:171-185)
  float bc[3], ba[3], ca[3], pb[3], pc[3], pa[3], t3[3], nr[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    bc[k] = Bv[k] - Cv[k]; ba[k] = Bv[k] - A[k]; ca[k] = Cv[k] - A[k];
    pb[k] = Bv[k] - p[k]; pc[k] = Cv[k] - p[k]; pa[k] = A[k] - p[k];
  }
  nr[0] = bc[1] * ba[2] - bc[2] * ba[1]; nr[1] = bc[2] * ba[0] - bc[0] * ba[2]; nr[2] = bc[0] * ba[1] - bc[1] * ba[0];
  t3[0] = ba[1] * ca[2] - ba[2] * ca[1]; t3[1] = ba[2] * ca[0] - ba[0] * ca[2]; t3[2] = ba[0] * ca[1] - ba[1] * ca[0];
  const float area = (nr[0] * t3[0] + nr[1] * t3[1]) + nr[2] * t3[2];
  t3[0] = pb[1] * pc[2] - pb[2] * pc[1]; t3[1] = pb[2] * pc[0] - pb[0] * pc[2]; t3[2] = pb[0] * pc[1] - pb[1] * pc[0];
  const float w0 = ((nr[0] * t3[0] + nr[1] * t3[1]) + nr[2] * t3[2]) / area;
  t3[0] = pc[1] * pa[2] - pc[2] * pa[1]; t3[1] = pc[2] * pa[0] - pc[0] * pa[2]; t3[2] = pc[0] * pa[1] - pc[1] * pa[0];
  const float w1 = ((nr[0] * t3[0] + nr[1] * t3[1]) + nr[2] * t3[2]) / area;
  const float w2 = 1.0f - w0 - w1;
  const float uu = (uvs_tex[faces[f * 3] * 2] * w0 + uvs_tex[faces[f * 3 + 1] * 2] * w1) + uvs_tex[faces[f * 3 + 2] * 2] * w2;
  const float vv = (uvs_tex[faces[f * 3] * 2 + 1] * w0 + uvs_tex[faces[f * 3 + 1] * 2 + 1] * w1) + uvs_tex[faces[f * 3 + 2] * 2 + 1] * w2;
  const long iu = lrintf(uu), iv = lrintf(vv);                      // torch.round: half to even
  if (iu < 0 || iv < 0 || iu >= tex_w || iv >= tex_w) return -1;
  // the reference flattens with (W-1) and decodes with (W-1) (nerf_runner.py:1528,1532): u == W-1 lands on (0, v+1)
  const int64_t flat = iv * (int64_t)(tex_w - 1) + iu;
  const int64_t du = flat % (tex_w - 1), dv = flat / (tex_w - 1);
  if (dv >= tex_w) return -1;
  return dv * tex_w + du;
}

__global__ __launch_bounds__(256) void k_tex_owner(TexCam c, const float* __restrict__ verts, const int64_t* __restrict__ faces,
                                                    const float* __restrict__ uvs_tex, const unsigned long long* __restrict__ zbuf,
                                                    const uint8_t* __restrict__ mask, float min_depth, int tex_w,
                                                    int32_t* __restrict__ owner) {
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= (int64_t)c.H * c.W) return;
  const int64_t t = pixel_texel(c, verts, faces, uvs_tex, zbuf, mask, min_depth, tex_w, pix);
  if (t >= 0) atomicMin(&owner[t], (int32_t)pix);
}

__global__ __launch_bounds__(256) void k_tex_accum(TexCam c, const float* __restrict__ verts, const int64_t* __restrict__ faces,
                                                    const float* __restrict__ uvs_tex, const unsigned long long* __restrict__ zbuf,
                                                    const uint8_t* __restrict__ mask, float min_depth, int tex_w,
                                                    const int32_t* __restrict__ owner, const float* __restrict__ rgb,
                                                    float* __restrict__ tex, float* __restrict__ wtex) {
  const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (pix >= (int64_t)c.H * c.W) return;
  const int64_t t = pixel_texel(c, verts, faces, uvs_tex, zbuf, mask, min_depth, tex_w, pix);
  if (t < 0 || owner[t] != (int32_t)pix) return;
  tex[t * 3] += rgb[pix * 3]; tex[t * 3 + 1] += rgb[pix * 3 + 1]; tex[t * 3 + 2] += rgb[pix * 3 + 2];   // one owner per texel: no atomics
  wtex[t] += 1.0f;
}

static int make_cam(const float* ob_in_cam12, const float* K4, int H, int W, TexCam* c) {
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) c->R[i * 3 + j] = ob_in_cam12[i * 4 + j];
    c->t[i] = ob_in_cam12[i * 4 + 3];
  }
  c->fx = K4[0]; c->fy = K4[1]; c->cx = K4[2]; c->cy = K4[3];
  c->H = H; c->W = W;
  return 0;
}

/* One keyframe of the texture bake.  ob_in_cam: HOST pointer to 12 floats (rows of the 3x4 normalised-object -> OpenCV-camera
 * transform); K4: HOST pointer to (fx, fy, cx, cy).  verts [nv,3] f32, faces [nf,3] i64, uvs_tex [nv,2] f32 (texel units),
 * mask [H,W] u8, rgb [H,W,3] f32 (raw colours), zbuf [H*W] u64 and owner [tex_res*tex_res] i32 scratch (overwritten),
 * tex [tex_res,tex_res,3] and wtex [tex_res,tex_res] f32 ACCUMULATED. */
extern "C" int nof_texture_bake_frame(const float* ob_in_cam, const float* K4, int32_t H, int32_t W, const float* verts,
                                       const int64_t* faces, int64_t n_faces, const float* uvs_tex, const uint8_t* mask,
                                       const float* rgb, float min_depth, int32_t tex_res, uint64_t* zbuf, int32_t* owner,
                                       float* tex, float* wtex, void* stream) {
  NOF_ARG(ob_in_cam && K4 && verts && faces && uvs_tex && mask && rgb && zbuf && owner && tex && wtex);
  NOF_ARG(H > 0 && W > 0 && tex_res > 1 && n_faces >= 0 && n_faces < (1ll << 32));
  TexCam c;
  make_cam(ob_in_cam, K4, H, W, &c);
  hipStream_t st = (hipStream_t)stream;
  NOF_HIP(hipMemsetAsync(zbuf, 0xFF, (size_t)H * W * 8, st));
  NOF_HIP(hipMemsetAsync(owner, 0x7F, (size_t)tex_res * tex_res * 4, st));
  if (n_faces == 0) return 0;
  hipLaunchKernelGGL(k_tex_raster, dim3((unsigned)nof_div_up(n_faces, 256)), dim3(256), 0, st, c, verts, faces, n_faces,
                     (unsigned long long*)zbuf);
  const unsigned pb = (unsigned)nof_div_up((int64_t)H * W, 256);
  hipLaunchKernelGGL(k_tex_owner, dim3(pb), dim3(256), 0, st, c, verts, faces, uvs_tex, (const unsigned long long*)zbuf, mask,
                     min_depth, (int)tex_res, owner);
  hipLaunchKernelGGL(k_tex_accum, dim3(pb), dim3(256), 0, st, c, verts, faces, uvs_tex, (const unsigned long long*)zbuf, mask,
                     min_depth, (int)tex_res, (const int32_t*)owner, rgb, tex, wtex);
  NOF_LAUNCH_OK();
  return 0;
}
