/* libnof_hip.so -- C ABI of the MI355X (gfx950) Neural Object Field hot path.
 *
 * Drop-in boundary for the native operators the reference (NVlabs/BundleSDF) binds through
 * pybind11/ATen (file:line into /root/reference):
 *   gridencoder.grid_encode_forward / grid_encode_backward   mycuda/torch_ngp_grid_encoder/bindings.cpp:16-19,
 *                                                            gridencoder.h:23-24, gridencoder.cu:447-502
 *   common.sampleRaysUniformOccupiedVoxels                   mycuda/bindings.cpp:15, common.cu:107-125
 *   common.postprocessOctreeRayTracing                       mycuda/bindings.cpp:17, common.cu:151-167
 *   kaolin.render.spc.unbatched_raytrace (+ octree build/query)   Utils.py:362-371,393,457
 *   pytorch3d.transforms.se3_exp_map                         nerf_helpers.py:15,150
 *   NeRFSmall Linear/ReLU GEMMs (cuBLAS under autocast)      nerf_helpers.py:243-321, nerf_runner.py:1289-1294
 *   raw2outputs + loss assembly (eager PyTorch)              nerf_runner.py:1132-1169,679-752, nerf_helpers.py:367-399
 *   torch.optim.Adam + schedule                              nerf_runner.py:492-504,756-763
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless its name starts with h_ ; the caller (PyTorch) owns and
 *     allocates every buffer, including scratch; nothing here allocates or synchronises the host;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream);
 *   - all entry points are hipGraph-capturable (kernel launches + hipMemsetAsync only);
 *   - return 0 on success, <0 for an argument error, >0 = hipError_t; text via nof_last_error();
 *   - float == IEEE binary32; row-major; sample index b = ray*S + s.
 */
#ifndef NOF_HIP_H
#define NOF_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define NOF_MAX_LEVELS 16
#define NOF_MAX_LAYERS 8
#define NOF_RAY_COLS 12          /* dir 0-2, rgb 3-5, depth 6, mask 7, frame 8, type 9, near 10, far 11 (nerf_runner.py:259-300) */
#define NOF_VIEW_COLS 16         /* per-ray view vector: [frame_features(ff) | SH(9) | 0 pad] */

const char* nof_last_error(void);
/* ABI version of THIS header (nof_version() returns the library's): bumped whenever a signature or a struct layout changes, so that
 * an out-of-tree caller built against another header can refuse to run instead of passing misaligned arguments.
 *   100  rounds 1-4
 *   110  round 5: nof_batch_trace gained `marcher` (in the middle of its argument list); NofSampleCfg gained `marcher` (trailing)
 *   120  round 6: nof_mlp_wide_bwd_parts gained `featq`; the wide networks' workspace holds the sigma head's hand-off only
 *   121  round 6: new entry points nof_encode_mlp_wide_fwd, nof_mcl_count_blocks, nof_mcl_emit_blocks (nothing existing changed);
 *        the wide entry points refuse precisions 3 / 4 instead of running them as 2 / 1 */
#define NOF_ABI_VERSION 122
int nof_version(void);

/* ---- multires hash grid (replaces gridencoder.*) --------------------------------------------- */
typedef struct {
  int32_t  L, C;                          /* levels, features per level (C must be 2) */
  float    scale[NOF_MAX_LEVELS];         /* exp2f(l*S)*H-1 evaluated on the host in float32 (gridencoder.cu:155) */
  uint32_t resolution[NOF_MAX_LEVELS];    /* ceil(scale)+1                      (gridencoder.cu:156) */
  uint32_t offset[NOF_MAX_LEVELS];        /* first table row of the level       (grid.py:127-134) */
  uint32_t size[NOF_MAX_LEVELS];          /* rows in the level (hashmap_size)   (gridencoder.cu:154) */
  uint32_t hashed[NOF_MAX_LEVELS];        /* 1: fast_hash, 0: dense stride index (gridencoder.cu:66-83) */
} NofHashGrid;

/* pts_w [B,3] in [-1,1] (the module maps (x+1)/2, grid.py:160); table [rows,2]; feat [L,B,2] (level-major,
 * the reference's own kernel layout gridencoder.cu:384).  Out-of-range points give zeros. */
int nof_hash_encode_fwd(const NofHashGrid* h_grid, const float* pts_w, const float* table,
                        float* feat, int64_t B, void* stream);
/* dfeat [L,B,2]; grad_table [rows,2] is ACCUMULATED into (caller zeroes it once per step);
 * dpts [B,3] (may be NULL) is overwritten with dL/dpts_w (kernel_input_backward, gridencoder.cu:340-365). */
int nof_hash_encode_bwd(const NofHashGrid* h_grid, const float* pts_w, const float* table, const float* dfeat,
                        float* grad_table, float* dpts, int64_t B, void* stream);
/* debug/parity: the 8 absolute table rows each (point, level) touches -> idx [B,L,8] int32 (-1 if out of range) */
/* Same, restricted to the table rows of levels [level_lo, level_hi); dpts (if given) still covers all levels. */
int nof_hash_encode_bwd_levels(const NofHashGrid* h_grid, const float* pts_w, const float* table, const float* dfeat,
                               float* grad_table, float* dpts, int32_t level_lo, int32_t level_hi, int64_t B, void* stream);
/* The same with the eikonal term (cfg eikonal_weight > 0; nerf_runner.py:734-738 with the normal of run_network_density,
 * :1342-1345): geik [L,B,2] = d sdf / d feature and dedn [B,3] = dE/dn as written by nof_eikonal (both NULL = plain backward).
 * The normal is linear in the table (finite differences, gridencoder.cu:202-245), so its table gradient rides in the same
 * scatter; its input gradient (mixed second derivatives of the trilinear blend) is added to dpts. */
int nof_hash_encode_bwd_eik(const NofHashGrid* h_grid, const float* pts_w, const float* table, const float* dfeat,
                            const float* geik, const float* dedn, float* grad_table, float* dpts, int32_t level_lo,
                            int32_t level_hi, int64_t B, void* stream);
/* The full-featured form.  The backward is three independent kernels; `parts` selects which of them this call launches -- all on
 * `stream`, one after the other.  Running them beside each other is the caller's business (it owns the streams; this library
 * creates no stream and no event and reads no environment variable): the training step launches TABLE_BIG on its main stream and
 * the other two on its side stream.  tile_list: NofTileList (below) or NULL; with a list only the listed tiles' dfeat is read,
 * and dpts of the unlisted tiles is written as 0.  wgs_per_cu: persistent workgroups per CU of the TABLE_BIG kernel, 0 = default. */
#define NOF_HASH_BWD_TABLE_BIG 1        /* levels larger than 48 KiB: run-merged global atomics */
#define NOF_HASH_BWD_TABLE_SMALL 2      /* levels accumulated in LDS and flushed once per workgroup */
#define NOF_HASH_BWD_INPUT 4            /* dL/dpts over all levels (needs dpts) */
#define NOF_HASH_BWD_ALL 7
#define NOF_HASH_BWD_MERGE_INPUT 8      /* with TABLE_BIG | INPUT: the large levels' scatter and dL/dpts as two roles of ONE launch (round 6) */
#define NOF_HASH_BWD_NEW_BATCH 16       /* nof_hash_encode_bwd_step with MERGE_INPUT: the merged launch also moves the overflow mark of `flags`
                                         * (bit 2 -> the sticky bit 3), which the batch's ray marcher does unless it ran inside the
                                         * previous optimiser launch (nof_adam_step_tail_march) */
int nof_hash_encode_bwd_parts(const NofHashGrid* h_grid, const float* pts_w, const float* table, const float* dfeat,
                              const float* geik, const float* dedn, float* grad_table, float* dpts, int32_t level_lo,
                              int32_t level_hi, const void* tile_list, int32_t parts, int32_t wgs_per_cu, int64_t B, void* stream);
/* The same followed by nof_reduce_partials(partials, n_rows, n_cols, grad_mlp, flags) (below; same results), the row reduction of
 * the MLP backward riding inside the launch of the LDS-accumulated levels: what the training step calls for its table gradient. */
int nof_hash_encode_bwd_parts_reduce(const NofHashGrid* h_grid, const float* pts_w, const float* table, const float* dfeat,
                                     const float* geik, const float* dedn, float* grad_table, float* dpts, int32_t level_lo,
                                     int32_t level_hi, const void* tile_list, int32_t parts, int32_t workgroups_per_cu, int64_t B,
                                     const float* partials, int32_t n_rows, int32_t n_cols, float* grad_mlp, int32_t* flags,
                                     void* stream);
int nof_hash_corner_indices(const NofHashGrid* h_grid, const float* pts_w, int32_t* idx, int64_t B, void* stream);

/* ---- pose corrections (replaces PoseArray.get_matrices + pytorch3d se3_exp_map) ---------------- */
/* pose_data [F,6] (may be NULL: identity), c2w [F,16] row-major 4x4 -> tf [F,12] = (Delta_i @ c2w_i)[:3,:4];
 * Delta_0 = I (nerf_helpers.py:151-153).  max_rot in radians. */
int nof_pose_fwd(const float* pose_data, const float* c2w, float max_trans, float max_rot_rad,
                 float* tf, int32_t F, void* stream);
/* g_delta [F,12] = dL/dDelta_i[:3,:4] -> grad_pose [F,6] ACCUMULATED (frame 0 gets 0). */
int nof_pose_bwd(const float* pose_data, const float* g_delta, float max_trans, float max_rot_rad,
                 float* grad_pose, int32_t F, void* stream);

/* ---- occupancy grid + ray tracing (replaces the kaolin SPC octree + postprocessOctreeRayTracing) -- */
/* coords [P,3] int32 occupied cells at max_level (already dilated/clamped, nerf_runner.py:453-465);
 * occ_bits: ceil(n^3/32) uint32 words of the level-`level` grid, bit id = (x*n+y)*n+z, n = 2^level.
 * The function clears and fills occ_bits. */
int nof_occgrid_build(const int32_t* coords, int64_t P, int32_t max_level, int32_t level,
                      uint32_t* occ_bits, void* stream);
/* pts [N,3] -> inside [N] uint8: 1 if the level cell containing the point is occupied
 * (OctreeManager.get_center_ids >= 0, Utils.py:392-394). */
int nof_occgrid_query(const uint32_t* occ_bits, int32_t level, const float* pts, uint8_t* inside,
                      int64_t N, void* stream);
/* rays_o [R,3], rays_d [R,3] (unit, world) -> t_in_out [R,max_hits,2] zero padded, cell_ids [R,max_hits]
 * (may be NULL), n_hits [R]; flags[0] |= 1 if any ray overflowed max_hits.
 * Semantics = OctreeManager.ray_trace (Utils.py:443-475) + common.cu:129-149. */
int nof_trace_rays(const uint32_t* occ_bits, int32_t level, const float* rays_o, const float* rays_d,
                   int64_t R, int32_t max_hits, float* t_in_out, int32_t* cell_ids, int32_t* n_hits,
                   int32_t* flags, void* stream);

/* ---- ray-pool construction on the device (make_frame_rays nerf_runner.py:246-316, compute_near_far_and_filter_rays :39-65,
 *      ray_box_intersection_batch nerf_helpers.py:403-446, octree-miss filter :302-314, cloud denoise :178-195) ------------ */
/* cv2.dilate(mask, ones(k,k)): window offsets -k/2 .. k-1-k/2, borders ignored.  mask/tmp/out [H,W] uint8. */
int nof_mask_dilate(const uint8_t* mask, int32_t H, int32_t W, int32_t k, uint8_t* tmp, uint8_t* out, void* stream);
typedef struct {
  double  fx, fy, cx, cy;                 /* K (float64 like the reference's numpy path) */
  double  near_thr, far_thr;              /* near*sc, far*sc, already rounded to the dtype numpy would compare in */
  double  box_lo[3], box_hi[3];           /* cfg['bounding_box'] */
  double  pose[16];                       /* cam_in_world of this frame, row-major 4x4 */
  int32_t frame_id;                       /* written to column 8 */
  int32_t valid_depth_only;               /* cfg['rays_valid_depth_only'] */
} NofFrameRaysCfg;
/* One row per pixel, rows [H*W,12] (column layout above), keep [H*W] = 1 for the rays make_frame_rays returns: selected by
 * mask_sel (the dilated mask; occ_mask [H,W] may be NULL), usable depth, inside the bounding box, and -- when occ_bits is
 * given -- hitting an occupied octree cell.  image [H,W,3] f32, depth [H,W] f32, mask_in [H,W] u8. */
int nof_frame_rays(const NofFrameRaysCfg* h_cfg, const float* image, const float* depth, const uint8_t* mask_in,
                   const uint8_t* mask_sel, const uint8_t* occ_mask, const uint32_t* occ_bits, int32_t level,
                   int32_t H, int32_t W, float* rows, uint8_t* keep, void* stream);
/* clears keep[i] for kept rays (mask > 0, depth <= far_thr) whose back-projected point is farther than dist_thr from every
 * cloud point; poses [F,16] f64, cloud [P,3] f64. */
int nof_cloud_filter(const float* rows, int64_t N, uint8_t* keep, const double* poses, const double* cloud, int64_t P,
                     double far_thr, double dist_thr, void* stream);
/* out[offsets[i]] = rows[i] for keep[i] != 0 (offsets = exclusive prefix sum of keep, supplied by the caller). */
int nof_compact_rows(const float* rows, const uint8_t* keep, const int64_t* offsets, int64_t N, float* out, void* stream);

/* ---- one training batch: gather + ray setup + trace (render_rays nerf_runner.py:1044-1060) ------ */
/* pool [N,12]; ids [R] int64 rows of the pool (NULL: rows 0..R-1); tf [F,12].
 * Outputs: batch [R,12] gathered rows; rays_o_w [R,3]; viewdirs_w [R,3]; view [R,16] = [ff|SH9|0];
 * frame_feat [F,ff] may be NULL when ff == 0. */
/* `marcher` (an ARGUMENT, not library state: two callers in one process, or a captured graph's owner, each get what they asked for):
 * NOF_MARCHER_WAVE = one wave per ray, the ray's cells enumerated from the ranks of its plane crossings (levels <= 6; above, the
 * walk is used); NOF_MARCHER_WALK = one lane per ray walking its cells.  Same results, bit for bit. */
#define NOF_MARCHER_WAVE 0
#define NOF_MARCHER_WALK 1
int nof_batch_trace(const float* pool, const int64_t* ids, const float* tf, const float* frame_feat, int32_t ff,
                    int32_t sh_degree, const uint32_t* occ_bits, int32_t level, int64_t R, int32_t max_hits, int32_t marcher,
                    float* batch, float* rays_o_w, float* viewdirs_w, float* view,
                    float* t_in_out, int32_t* cell_ids, int32_t* n_hits, int32_t* flags, void* stream);
typedef struct {
  int32_t  n_samples, n_around;           /* N_samples, N_samples_around_depth (config.yml:18-19) */
  float    near_sc, far_sc;               /* near*sc_factor, far*sc_factor */
  float    trunc;                         /* get_truncation() (nerf_runner.py:663-676) */
  float    neg_trunc_ratio;
  uint64_t seed;                          /* Philox key when u_occ/u_dep are NULL */
  uint32_t step;                          /* Philox counter word 2 */
  const uint32_t* d_step;                 /* DEVICE pointer (or NULL): when set, the Philox step is read from it at run time
                                           * instead of `step` -- what lets a captured step be replayed (NofStepState.step) */
  int32_t  deterministic;                 /* != 0: perturb=False of sample_rays_uniform (nerf_runner.py:67-87) -- the linspace itself,
                                           * no jitter, no clip; u_occ / u_dep / seed are ignored (render_images, :597) */
  int32_t  marcher;                       /* which ray marcher nof_raymarch_sample launches: NOF_MARCHER_WAVE (0, the default of a
                                           * zero-initialised struct) or NOF_MARCHER_WALK; same bits either way (see nof_batch_trace) */
} NofSampleCfg;
/* z sampling + point generation (nerf_runner.py:979-1011,1063-1083,1242-1245; common.cu:41-105).
 * u_occ [R,n_samples], u_dep [R,n_around] injected uniforms or NULL (Philox4x32-10).
 * Outputs z_vals [R,S], pts_w [R*S,3], valid [R*S] uint8. */
int nof_sample_points(const NofSampleCfg* h_cfg, const float* batch, const float* tf, const float* t_in_out,
                      const int32_t* n_hits, int64_t R, int32_t max_hits, const float* u_occ, const float* u_dep,
                      float* z_vals, float* pts_w, uint8_t* valid, int32_t* flags, void* stream);

/* The two calls above as one entry point (render_rays up to the sample points, nerf_runner.py:1044-1083): this is what the
 * training step calls.  Arguments as in nof_batch_trace / nof_sample_points. */
int nof_raymarch_sample(const NofSampleCfg* h_cfg, const float* pool, const int64_t* ids, const float* tf,
                        const float* frame_feat, int32_t ff, int32_t sh_degree, const uint32_t* occ_bits, int32_t level,
                        int64_t R, int32_t max_hits, const float* u_occ, const float* u_dep,
                        float* batch, float* rays_o_w, float* viewdirs_w, float* view, float* t_in_out,
                        int32_t* cell_ids, int32_t* n_hits, float* z_vals, float* pts_w, uint8_t* valid,
                        int32_t* flags, void* stream);

/* ---- device-resident step state: the per-step scalars of a captured (hipGraph) step --------------------------------------
 * A captured step bakes every by-value argument into its launches.  The three that change every step -- the Philox step of the
 * sampler, Adam's step count and the scheduled learning rates -- therefore live in this 16-byte device struct: the sampler reads
 * `step` through NofSampleCfg.d_step, nof_adam_step_dyn reads the step sizes, and nof_step_state_advance (the last launch of the
 * captured step) increments `step` and recomputes the rest for the next replay (schedule_lr, nerf_runner.py:579-583: the rate
 * changes after steps g with g % 10 == 0, g > 0: lr = lrate * decay_rate^(g / n_iters)). */
typedef struct {
  uint32_t step;                          /* optimiser steps taken so far (= global_step) */
  float step_basic, step_pose;            /* lr / (1 - beta1^t), t = step + 1, for the two param groups */
  float inv_sqrt_bc2;                     /* 1 / sqrt(1 - beta2^t) */
} NofStepState;
/* set_step < 0: step += 1; otherwise step = set_step.  Then the Adam constants of the NEXT optimiser step are recomputed. */
int nof_step_state_advance(NofStepState* d_state, float lrate, float lrate_pose, float decay_rate, int32_t n_iters, float beta1,
                           float beta2, int32_t set_step, void* stream);
/* nof_adam_step with lr / step taken from the device state */
int nof_adam_step_dyn(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n_total, int64_t n_basic,
                      const NofStepState* d_state, float beta1, float beta2, float eps, const int32_t* skip_flags, void* stream);

/* ---- SDF + colour tiny-MLPs on MFMA (replaces NeRFSmall's cuBLAS GEMMs) ------------------------ */
typedef struct {
  int32_t n_sigma, n_color;               /* NeRFSmall(num_layers, num_layers_color) nerf_helpers.py:244 */
  int32_t hidden;                         /* 64 */
  int32_t in_feat;                        /* L*C <= 32 */
  int32_t n_view;                         /* ff + SH coeffs <= 16 */
  int32_t geo;                            /* geo_feat_dim = 15 */
  int32_t w_off[NOF_MAX_LAYERS];          /* float offsets of W_l [out,in] inside `mlp_params` (PyTorch parameter order) */
  int32_t b_off[NOF_MAX_LAYERS];
  int32_t out_dim[NOF_MAX_LAYERS], in_dim[NOF_MAX_LAYERS];
  int32_t n_params;
  int32_t precision;                      /* 0: fp32 MFMA (exact), 1: bf16 MFMA, 2: fp16 MFMA */
  float grad_scale;                       /* nof_mlp_bwd only: loss scale (a power of two; 0 = 1).  draw is multiplied by it on
                                           * entry and dfeat / dview / partials are divided by it on exit, so callers never see
                                           * it: what the reference's GradScaler does for its fp16 autocast path */
} NofMlpDesc;
/* Weights are consumed as an MFMA-fragment image: nof_mlp_pack converts the fp32 PyTorch-layout parameters (once per
 * optimiser step) into `packed` (nof_mlp_packed_bytes() bytes, caller-allocated); fwd / bwd / sdf read only the image. */
int64_t nof_mlp_packed_bytes(const NofMlpDesc* h_desc);
int nof_mlp_pack(const NofMlpDesc* h_desc, const float* mlp_params, void* packed, void* stream);
/* nof_mlp_pack and nof_pose_fwd (above: same arguments, same results) as ONE launch -- the two things a training step needs before
 * its ray marcher (the reference rebuilds neither explicitly: autograd re-reads the nn.Linear weights, and PoseArray.get_matrices
 * runs inside render_rays, nerf_runner.py:1051-1053).  F == 0: nof_mlp_pack alone. */
int nof_mlp_pack_pose(const NofMlpDesc* h_desc, const float* mlp_params, void* packed, const float* pose_data, const float* c2w,
                      float max_trans, float max_rot_rad, float* tf, int32_t F, void* stream);
/* feat [L,B,2]; view [R,16]; raw [B,4] = (rgb_raw[3], sdf)  (nerf_helpers.py:319).
 * sigma_out (may be NULL; ignored in fp32 mode): [B,16] elements of the MFMA operand type (2 bytes) = the sigma head's
 * output as the colour net consumes it; nof_mlp_bwd's split path reads it back instead of recomputing the sigma net twice. */
int nof_mlp_fwd(const NofMlpDesc* h_desc, const void* packed, const float* feat, int32_t L,
                const float* view, int32_t S, float* raw, void* sigma_out, int64_t B, void* stream);
/* draw [B,4] -> dfeat [L,B,2] (overwritten), dview [R,16] ACCUMULATED, partials [n_rows, n_params]
 * overwritten with per-workgroup weight-gradient partial sums (reduce with nof_reduce_partials).
 * n_rows must equal nof_mlp_bwd_blocks() (= the 2 persistent workgroups per CU of the grid; their four waves are summed in LDS).
 * 16-bit modes: with sigma_out (as written by nof_mlp_fwd) and dsigma_ws (scratch of the same size, [B,16] x 2 bytes) the
 * backward runs as two kernels (colour net, sigma net) at twice the occupancy; with either NULL, and always in fp32 mode,
 * one fused kernel recomputes everything.  Both paths produce the same values. */
int nof_mlp_bwd_blocks(void);
int nof_mlp_bwd(const NofMlpDesc* h_desc, const void* packed, const float* feat, int32_t L,
                const float* view, int32_t S, const float* draw, const void* sigma_out, void* dsigma_ws,
                float* dfeat, float* dview, float* partials, int64_t B, void* stream);
/* The same over a work list (NofTileList, below): only the listed 32-sample tiles are computed, dealt evenly to the persistent
 * waves.  dfeat (and dsigma_ws) of UNLISTED tiles is not written; the hash backward of the same step takes the same list. */
int nof_mlp_bwd_tiles(const NofMlpDesc* h_desc, const void* packed, const float* feat, int32_t L,
                      const float* view, int32_t S, const float* draw, const void* sigma_out, void* dsigma_ws,
                      float* dfeat, float* dview, float* partials, const void* tile_list, int64_t B, void* stream);
/* Hash encode + both MLPs in ONE launch, the embedding kept on chip (north_star: "LDS-staged features"; replaces the pair
 * nof_hash_encode_fwd + nof_mlp_fwd of the training forward, reference nerf_runner.py:1255-1294 which materialises `embedded`).
 * 16-bit operand precisions.  pts_w [B,3] world points, table [rows,2], view [R,16], raw [B,4]; sigma_out as in nof_mlp_fwd;
 * featq (may be NULL): [B][2][16] operand-type elements (64 B per sample) -- the features as nof_mlp_bwd_featq reads them. */
int nof_encode_mlp_fwd(const NofHashGrid* h_grid, const NofMlpDesc* h_desc, const void* packed, const float* table,
                       const float* pts_w, const float* view, int32_t S, float* raw, void* sigma_out, void* featq,
                       int64_t B, void* stream);
/* nof_mlp_bwd_tiles with the features read from featq instead of the fp32 level-major array (split workspace required). */
int nof_mlp_bwd_featq(const NofMlpDesc* h_desc, const void* packed, const void* featq, int32_t L,
                      const float* view, int32_t S, const float* draw, const void* sigma_out, void* dsigma_ws,
                      float* dfeat, float* dview, float* partials, const void* tile_list, int64_t B, void* stream);
/* The 16-bit backward runs per network, colour then sigma.  Where both halves use the same workgroup shape (two colour layers) they are
 * ONE launch since round 6 -- a workgroup walks its tiles through the colour net, then the same tiles through the sigma net --;
 * this entry point keeps them as two launches whatever the shape (same bits: the A/B and the parity test of the merged launch). */
int nof_mlp_bwd_featq_two_launches(const NofMlpDesc* h_desc, const void* packed, const void* featq, int32_t L,
                                   const float* view, int32_t S, const float* draw, const void* sigma_out, void* dsigma_ws,
                                   float* dfeat, float* dview, float* partials, const void* tile_list, int64_t B, void* stream);
/* out[j] += sum_i partials[i,j].  flags (int32, may be NULL): flags[0] |= 4 when a column sum is not finite -- an overflow inside the
 * 16-bit backward, where the reference's GradScaler skips the step and backs off (nerf_runner.py:756-761): nof_adam_step[_dyn]
 * given the same flags skips the update, the next batch's nof_sample_points turns the mark into the sticky bit 3 (value 8), the
 * host polls that and lowers NofMlpDesc.grad_scale (bundlesdf_amd/field.py); the step itself never synchronises. */
int nof_reduce_partials(const float* partials, int32_t n_rows, int32_t n_cols, float* out, int32_t* flags, void* stream);
/* sigma_net only: feat [L,B,2] -> sdf [B]  (NeRFSmall.forward_sdf, nerf_helpers.py:296-302) */
int nof_mlp_sdf(const NofMlpDesc* h_desc, const void* packed, const float* feat, int32_t L,
                float* sdf, int64_t B, void* stream);

/* ---- wide / deep networks: hidden 128 and/or 4 layers per network (BASELINE cfg5: SDF 4x128 + colour 4x128; NeRFSmall is
 * parameterised in hidden_dim / num_layers, nerf_helpers.py:243-294).  nof_mlp_fwd / _bwd / _sdf / nof_sdf_grid_query accept
 * hidden 64 with depths {2,3} and reject everything else with a message naming these entry points.  16-bit operand types only.
 * One network and one fragment orientation per kernel is resident in LDS; hidden activations and pre-activation gradients are
 * staged in `workspace` (nof_mlp_wide_workspace_bytes(desc, B) bytes, caller-allocated, must survive from the forward to the
 * backward call of the same batch).  Same tensors as the narrow entry points otherwise; `partials` is
 * [nof_mlp_wide_partial_rows(), n_params] floats, overwritten, to be summed with nof_reduce_partials. */
int64_t nof_mlp_wide_workspace_bytes(const NofMlpDesc* h_desc, int64_t B);
int nof_mlp_wide_partial_rows(void);
int nof_mlp_wide_fwd(const NofMlpDesc* h_desc, const void* packed, const float* feat, int32_t L, const float* view, int32_t S,
                     float* raw, void* workspace, int64_t B, void* stream);
int nof_mlp_wide_sdf(const NofMlpDesc* h_desc, const void* packed, const float* feat, int32_t L, float* sdf, int64_t B,
                     void* stream);
int nof_mlp_wide_bwd(const NofMlpDesc* h_desc, const void* packed, const float* feat, int32_t L, const float* view, int32_t S,
                     const float* draw, void* workspace, float* dfeat, float* dview, float* partials, int64_t B, void* stream);
/* over a work list (NofTileList): data path and weight-gradient passes of the listed tiles only; dfeat of the others is not written */
int nof_mlp_wide_bwd_tiles(const NofMlpDesc* h_desc, const void* packed, const float* feat, int32_t L, const float* view, int32_t S,
                           const float* draw, void* workspace, float* dfeat, float* dview, float* partials, const void* tile_list,
                           int64_t B, void* stream);
/* Hash encode + both wide networks in two launches with the fp32 embedding never in HBM (the wide counterpart of nof_encode_mlp_fwd;
 * replaces nof_hash_encode_fwd + nof_mlp_wide_fwd in the training forward; reference nerf_runner.py:1255-1294 materialises `embedded`).
 * pts_w [B,3], table [rows,2], view [R,16] -> raw [B,4]; `workspace` as for nof_mlp_wide_fwd; featq (may be NULL): [B][2][16]
 * operand-type elements (64 B per sample) = the embedding as nof_mlp_wide_bwd_parts reads it. */
int nof_encode_mlp_wide_fwd(const NofHashGrid* h_grid, const NofMlpDesc* h_desc, const void* packed, const float* table,
                            const float* pts_w, const float* view, int32_t S, float* raw, void* workspace, void* featq,
                            int64_t B, void* stream);
/* restricted to `parts` (all on `stream`): the colour net's kernel (needs draw; writes dview, the sigma head's gradient and the
 * colour layers' entries of every partial row) and / or the sigma net's (needs the colour part; writes dfeat and the sigma layers'
 * entries).  `featq` (may be NULL): the embedding in MFMA operand precision and order as nof_encode_mlp_wide_fwd leaves it
 * ([B][2][16] elements), read instead of the fp32 `feat` [L,B,2] (which may then be NULL).  Round 6: the four-way split of round 3
 * (data path / weight-gradient passes per network) is gone -- a network's backward is one kernel. */
#define NOF_WIDE_BWD_COLOR 1
#define NOF_WIDE_BWD_SIGMA 2
#define NOF_WIDE_BWD_ALL 3
int nof_mlp_wide_bwd_parts(const NofMlpDesc* h_desc, const void* packed, const float* feat, const void* featq, int32_t L,
                           const float* view, int32_t S, const float* draw, void* workspace, float* dfeat, float* dview,
                           float* partials, const void* tile_list, int32_t parts, int64_t B, void* stream);

/* ---- eikonal option (cfg eikonal_weight > 0; nerf_runner.py:734-738 with the normal of run_network_density, :1342-1345):
 * E = w * mean over {sdf < 1} of (|d sdf / d x| - 1)^2, evaluated with the exact-fp32 MFMA.  h_desc32 / packed32: the network
 * packed with precision 0.  pts_w [B,3], valid [B] u8, n_sel: DEVICE scalar (float) = number of samples with sdf < 1, weight =
 * eikonal_weight, grad_scale = 1/world_size (applied to the gradients only).  Writes geik [L,B,2] (d sdf / d feature) and dedn [B,3] (dE/dn) for nof_hash_encode_bwd_eik,
 * the sigma layers' weight gradient as per-workgroup rows of partials_e [nof_mlp_bwd_blocks(), n_params] (the colour layers'
 * entries are never written: zero the buffer once), and ADDS the term to loss_out[0] and loss_out[7]. */
int nof_eikonal(const NofMlpDesc* h_desc32, const void* packed32, const NofHashGrid* h_grid, const float* table,
                const float* pts_w, const uint8_t* valid, const float* n_sel, float weight, float grad_scale, float* geik,
                float* dedn, float* partials_e, float* loss_out, int64_t B, void* stream);

/* bytes of the `partials` workspace of nof_mlp_bwd (= nof_mlp_bwd_blocks() * n_params * 4); -1 on a bad descriptor */
int64_t nof_mlp_bwd_workspace_bytes(const NofMlpDesc* h_desc);

/* ---- dense SDF grid for mesh extraction (extract_mesh + run_network_density, nerf_runner.py:1307-1386) ------------
 * Fused: voxel centre (tx[i], ty[j], tz[k]) -> occupancy mask (get_center_ids >= 0, Utils.py:393-398; occ_bits may be NULL =
 * every voxel valid) -> clip to [-1,1] -> hash encode -> sigma net -> sdf[(i*ny + j)*nz + k]; voxels outside the octree get
 * `outside_value` (1.0 in the reference, nerf_runner.py:1384-1385).  tx/ty/tz are the float32 device copies of the host's
 * np.arange axes, so the query points are bit-identical to the reference's meshgrid.  No per-point buffer in HBM. */
int nof_sdf_grid_query(const NofHashGrid* h_grid, const NofMlpDesc* h_desc, const void* packed, const float* table,
                       const uint32_t* occ_bits, int32_t level, const float* tx, const float* ty, const float* tz,
                       int32_t nx, int32_t ny, int32_t nz, float outside_value, float* sdf, void* stream);

/* ---- iso-surface extraction on the device (replaces skimage.measure.marching_cubes, nerf_runner.py:1388-1394) -----
 * Marching tetrahedra, 6 per cell.  vol [nx,ny,nz]; cells are (nx-1)(ny-1)(nz-1), z fastest.
 *   nof_mt_count     counts [ncell] int32 = triangles per cell;
 *   (caller: exclusive prefix sum of counts -> offsets [ncell] int64, T = total)
 *   nof_mt_emit      keys [T,3] int64, one EDGE KEY per triangle corner: lo * (nx*ny*nz) + hi of the edge's two grid points;
 *                    triangles are oriented from value < iso towards value >= iso;
 *   (caller: sort/unique of the keys -> V unique keys, faces = inverse indices)
 *   nof_mt_vertices  verts [V,3] float64 = interpolated vertex of each unique key, in index coordinates. */
int nof_mt_count(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, int32_t* counts, void* stream);
int nof_mt_emit(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, const int64_t* offsets, int64_t* keys,
                void* stream);
int nof_mt_vertices(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, const int64_t* keys, int64_t V,
                    double* verts, void* stream);

/* Marching cubes, the extractor the reference calls (skimage.measure.marching_cubes, nerf_runner.py:1388-1394): same three steps,
 * same edge keys, vertices through nof_mt_vertices.  case_table [256,16] int8 (device): row `case` (bit c set = corner c = x + 2y +
 * 4z of the cell has value < iso) = [T <= 5, 3 T cube-edge ids]; edge e joins the corners (0,1) (0,2) (0,4) (1,3) (1,5) (2,3) (2,6)
 * (3,7) (4,5) (4,6) (5,7) (6,7)[e].  The host derives the table (bundlesdf_amd/mesh.py:mc_case_table: oriented, watertight). */
int nof_mc_count(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, const int8_t* case_table, int32_t* counts,
                 void* stream);
int nof_mc_emit(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, const int8_t* case_table,
                const int64_t* offsets, int64_t* keys, void* stream);

/* Marching cubes with Lewiner's topological disambiguation = skimage.measure.marching_cubes(volume, level) with its default
 * method='lewiner', the call the reference makes (nerf_runner.py:1388-1394): per cell the tiling of one of the 33 topological cases,
 * chosen by the paper's face tests and interior test (Lewiner, Lopes, Vieira, Tavares, JGT 8(2) 2003).  Same scheme as nof_mc_*:
 * counts -> (host: exclusive scan) -> keys -> (host: sort / unique) -> vertices.  `luts` = the paper's lookup tables as ONE packed int8
 * buffer on the device, `offs->off[t]` = byte offset of table t in it, tables in the order
 *   CASES, TILING1, 2, 3_1, 3_2, 4_1, 4_2, 5, 6_1_1, 6_1_2, 6_2, 7_1, 7_2, 7_3, 7_4_1, 7_4_2, 8, 9, 10_1_1, 10_1_1_, 10_1_2, 10_2, 10_2_,
 *   11, 12_1_1, 12_1_1_, 12_1_2, 12_2, 12_2_, 13_1, 13_1_, 13_2, 13_2_, 13_3, 13_3_, 13_4, 13_5_1, 13_5_2, 14, TEST3, 4, 6, 7, 10, 12,
 *   13, SUBCONFIG13                                                       (bundlesdf_amd/mesh.py:lewiner_lut_pack builds both).
 * A key >= 0 is an edge key as above; a key < 0 is the CENTRE vertex of cell -(key + 1) (tilings 6.1.2, 7.3, 10.2, 12.2, 13.3,
 * 13.4), which nof_mcl_vertices places like scikit-image does (the cell's corners weighted by 1 / (eps + |value - iso|)); the vertices on
 * grid edges likewise (the edge's two points with those weights: linear interpolation up to eps), in float64.  Triangles are wound like
 * scikit-image's default (gradient_direction='descent'). */
#define NOF_MCL_TABLES 47
typedef struct { int32_t off[NOF_MCL_TABLES]; } NofMclLuts;
int nof_mcl_count(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, const int8_t* luts, const NofMclLuts* offs,
                  int32_t* counts, void* stream);
int nof_mcl_emit(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, const int8_t* luts, const NofMclLuts* offs,
                 const int64_t* offsets, int64_t* keys, void* stream);
/* The two-level form of count / emit (what bundlesdf_amd/mesh_gpu.py calls): ONE triangle count per workgroup of 256 consecutive
 * cells -- block_counts [ceil(ncell / 256)] --, the host's inclusive 64-bit scan of those (block_end), and an emit launch in which
 * every workgroup with surface recomputes its cells' tilings and places them behind block_end[b - 1] with a workgroup-local scan:
 * the same keys in the same order as nof_mcl_count -> scan -> nof_mcl_emit, without the per-cell arrays (2.5 GB at 512^3). */
int nof_mcl_count_blocks(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, const int8_t* luts, const NofMclLuts* offs,
                         int32_t* block_counts, void* stream);
int nof_mcl_emit_blocks(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, const int8_t* luts, const NofMclLuts* offs,
                        const int64_t* block_end, int64_t* keys, void* stream);
int nof_mcl_vertices(const float* vol, int32_t nx, int32_t ny, int32_t nz, float iso, const int64_t* keys, int64_t V,
                     double* verts, void* stream);

/* ---- texture bake helper (replaces common.rayColorToTextureImageCUDA, mycuda/common.h:30, common.cu:171-238) ----------
 * faces [nf,3] int64, verts [nv,3] f32, hit_locations [n,3] f32 (points on the mesh), hit_face_ids [n] int64,
 * uvs_tex [nv,2] f32 per-vertex texture coordinates -> uvs [n,2]: barycentric blend of the hit triangle's uvs. */
int nof_bary_uv(const int64_t* faces, const float* verts, const float* hit_locations, const int64_t* hit_face_ids,
                const float* uvs_tex, int64_t n_hits, float* uvs, void* stream);
/* ---- texture bake, one keyframe (NerfRunner.mesh_texture_from_train_images, nerf_runner.py:1499-1535) ------------------
 * Replaces, per frame: the pyrender depth render + trimesh.proximity.closest_point (visible surface point and triangle of
 * every pixel: z-buffer rasteriser), common.rayColorToTextureImageCUDA (barycentric UV) and the one-colour-per-texel-and-frame
 * accumulation.  ob_in_cam (HOST, 12 floats: rows of the 3x4 normalised-object -> OpenCV-camera transform), K4 (HOST: fx, fy, cx,
 * cy); verts [nv,3] f32, faces [nf,3] i64, uvs_tex [nv,2] f32 in texel units (uv * (tex_res-1)), mask [H,W] u8, rgb [H,W,3]
 * f32 raw colours; pixels whose rendered depth is below min_depth are skipped; zbuf [H*W] u64 and owner [tex_res^2] i32 are
 * scratch; tex [tex_res,tex_res,3] / wtex [tex_res,tex_res] f32 are ACCUMULATED (divide at the end). */
int nof_texture_bake_frame(const float* h_ob_in_cam, const float* h_K4, int32_t H, int32_t W, const float* verts,
                           const int64_t* faces, int64_t n_faces, const float* uvs_tex, const uint8_t* mask, const float* rgb,
                           float min_depth, int32_t tex_res, uint64_t* zbuf, int32_t* owner, float* tex, float* wtex, void* stream);

/* ---- compositing + losses + dL/draw (raw2outputs, train_loop, get_sdf_loss) ---------------------- */
typedef struct {
  float trunc, neg_trunc_ratio, sdf_lambda;
  float near_sc, far_sc;
  float rgb_weight, fs_weight, trunc_weight, empty_weight, fs_sdf, fs_rgb_weight;
  float first_frame_weight;
  float grad_scale;                       /* multiplies every gradient (1/world_size for DP averaging) */
} NofLossCfg;
/* raw [R,S,4], z_vals [R,S], valid [R,S] u8, batch [R,12] ->
 * rgb_map [R,3], weights [R,S] (may be NULL), draw [R,S,4], loss_out [8] ACCUMULATED (may be NULL):
 * 0 total, 1 rgb, 2 fs(+empty), 3 sdf, 4 fs_rgb, 5 n_valid_samples, 6 n_valid_rays.
 * loss_rows [R,8] is scratch for the per-ray terms (required when loss_out is given). */
int nof_composite_loss(const NofLossCfg* h_cfg, const float* raw, const float* z_vals, const uint8_t* valid,
                       const float* batch, int64_t R, int32_t S, float* rgb_map, float* weights, float* draw,
                       float* loss_rows, float* loss_out, void* stream);

/* ---- work list of the backward: north_star's per-wavefront compaction ----------------------------------------------------
 * A ray-sample whose row of dL/draw is EXACTLY zero (rays without a loss term; free-space samples whose loss has saturated: two
 * thirds of a settled cfg2 batch) contributes exactly nothing to any gradient.  NofTileList names the 32-sample tiles (tile t =
 * samples 32t .. 32t+31) that hold at least one non-zero row, in ascending order; the backward entry points that take one
 * (nof_mlp_bwd_tiles, nof_hash_encode_bwd_parts) deal the LISTED tiles evenly to their persistent waves and touch nothing else:
 * the same sums as the whole batch, none of the work of the zeros, balanced waves.  Device memory, caller-allocated,
 * nof_tile_list_bytes(B) bytes:  uint32 count, n_tiles, 0, 0 | uint32 tiles[n_tiles (+ pad)] | uint8 flags[n_tiles].
 * It lives on the device only (the count never visits the host: capturable). */
int64_t nof_tile_list_bytes(int64_t B);
/* the list from an existing dL/draw [B,4]; or, with all != 0 (draw may be NULL), every tile of the batch: the list that makes the
 * backward entry points do the whole batch without looking for zeros */
int nof_tile_list_build(const float* draw, int64_t B, int32_t all, void* tile_list, void* stream);
/* nof_composite_loss + the work list of its dL/draw (tile_list may be NULL), with loss_out OVERWRITTEN instead of accumulated
 * (the first writer of a step's loss terms: no zero-fill launch in front of it).  When S % 32 == 0 the list costs no launch and
 * no pass over draw: the flags come out of the loss kernel and the scan rides beside the loss reduction. */
int nof_composite_loss_fwd_bwd(const NofLossCfg* h_cfg, const float* raw, const float* z_vals, const uint8_t* valid,
                               const float* batch, int64_t R, int32_t S, float* rgb_map, float* weights, float* draw,
                               float* loss_rows, float* loss_out, void* tile_list, void* stream);

/* ---- pose / feature gradients of a batch ---------------------------------------------------------- */
/* dpts [R*S,3] (from nof_hash_encode_bwd, may be NULL), dview [R,16] (from nof_mlp_bwd), batch, z_vals, c2w [F,16], tf [F,12]
 * -> g_ray [R,12] = this ray's contribution to dL/dDelta_frame (rows of frame-0 rays are 0).
 * frame_slots (may be NULL): [F, NOF_POSE_SLOTS, NOF_POSE_SLOT_W] floats, all zero on entry.  When given, the ray's 12 values and
 * its dview[:ff] are ALSO added (one fp32 atomic instruction per ray) to slot (frame, ray % NOF_POSE_SLOTS), and the ray's dview
 * row is set to 0 (this is then its last reader in a step) -- nof_pose_reduce_bwd sums a frame's slots instead of searching the
 * batch for the frame's rays. */
#define NOF_POSE_SLOTS 16
#define NOF_POSE_SLOT_W 28                  /* 12 + NOF_VIEW_COLS */
/* nof_pose_grad_accum's arguments as a struct (nof_hash_encode_bwd_step carries that kernel as a passenger of another launch) */
typedef struct {
  const float* dpts; float* dview; const float* batch; const float* z_vals; const float* c2w; const float* tf;
  int32_t ff, sh_degree; int64_t R; int32_t S; float* g_ray; float* frame_slots;
} NofPoseAccum;
/* The hash side of the training step's backward in two launches: nof_hash_encode_bwd_parts_reduce with
 * NOF_HASH_BWD_MERGE_INPUT, and -- pose != NULL -- nof_pose_grad_accum(pose->...) as a passenger of the LDS levels' launch (it
 * needs dL/dpts, which the merged launch in front of it has finished).  Same results as the separate calls. */
int nof_hash_encode_bwd_step(const NofHashGrid* g, const float* pts_w, const float* table, const float* dfeat, const float* geik,
                             const float* dedn, float* grad_table, float* dpts, int32_t level_lo, int32_t level_hi,
                             const void* tile_list, int32_t parts, int32_t wgs_per_cu, int64_t B, const float* partials,
                             int32_t n_rows, int32_t n_cols, float* grad_mlp, int32_t* flags, const NofPoseAccum* pose, void* stream);
int nof_pose_grad_accum(const float* dpts, float* dview, const float* batch, const float* z_vals,
                        const float* c2w, const float* tf, int32_t ff, int32_t sh_degree, int64_t R, int32_t S,
                        float* g_ray, float* frame_slots, void* stream);
/* one workgroup per frame: g_delta[f] = sum of its rays' rows (written when non-NULL), grad_pose [F,6] += se3 backward,
 * grad_feat [F,ff] += sum of its rays' dview[:, :ff] (either gradient pointer may be NULL).
 * frame_slots == NULL: the frame's rays are found in `batch` and their rows of g_ray / dview summed in a fixed order;
 * zero_dview != 0: the rows of dview are set to 0 once read (this is their last reader in a step; every ray's frame must lie in
 * [0, F)): ready for the next step's nof_mlp_bwd, which accumulates into them.
 * frame_slots != NULL (filled by nof_pose_grad_accum): the frame's NOF_POSE_SLOTS partial sums are added in slot order and set
 * back to 0; g_ray / dview / batch are not read. */
int nof_pose_reduce_bwd(const float* pose_data, const float* g_ray, float* dview, const float* batch, int64_t R,
                        int32_t ff, float max_trans, float max_rot_rad, float* grad_pose, float* grad_feat,
                        float* g_delta, int32_t F, int32_t zero_dview, float* frame_slots, void* stream);
/* grad += 2*w*data/numel  (feature_reg, nerf_runner.py:745-747) and pose_reg (:749-752) */
int nof_small_regs(const float* feat_data, float* grad_feat, int32_t n_feat, float feature_reg_weight,
                   float grad_scale, void* stream);

/* pose regulariser (nerf_runner.py:749-752): loss += w * ||pose_data[1:]||_2; grad_pose += grad_scale * d/dpose; loss_out[0]
 * (may be NULL) += the term. */
int nof_pose_reg(const float* pose_data, float* grad_pose, int32_t F, float pose_reg_weight, float grad_scale,
                 float* loss_out, void* stream);

/* render_images' depth map values (nerf_runner.py:604-612): per ray the z of the first sample pair whose SDFs (raw[..,3]) differ in
 * sign; `far` (= cfg far * sc_factor) when every pair's product is > 0; z_vals[r,0] otherwise.  raw [R,S,4], z_vals [R,S] -> depth [R]. */
int nof_render_depth(const float* raw, const float* z_vals, int64_t R, int32_t S, float far, float* depth, void* stream);

/* ---- optimiser ---------------------------------------------------------------------------------- */
/* torch.optim.Adam(betas, eps=1e-15, weight_decay=0) over the flat buffer; entries [0,n_basic) use lr,
 * [n_basic,n) use lr_pose (param groups nerf_runner.py:498-500).  step is 1-based.  Grads are zeroed.
 * skip_flags (device int32, may be NULL): when bit 2 of skip_flags[0] is set -- this step's weight gradient came out of the 16-bit
 * backward non-finite (nof_reduce_partials / nof_grad_check) -- the update is SKIPPED like torch's GradScaler.step skips it
 * (nerf_runner.py:756-761): params / exp_avg / exp_avg_sq untouched, grads zeroed.  The next step's nof_sample_points turns the
 * mark into the sticky bit 3 (value 8) that the host polls to lower NofMlpDesc.grad_scale. */
int nof_adam_step(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t n_basic,
                  float lr, float lr_pose, float beta1, float beta2, float eps, int32_t step, const int32_t* skip_flags,
                  void* stream);
/* The optimiser launch of a single-GPU training step with its two neighbours inside (round 6): nof_pose_reduce_bwd (slot mode, no
 * frame features) in front -- grads[pose_off + 6 f ..] += SE(3) backward of frame f's summed slots --, nof_adam_step over the flat
 * buffers [table | MLP | 6 F pose entries] (mlp_off + n_mlp == pose_off == n_basic, pose_off + 6 F == n), and nof_mlp_pack_pose
 * behind: `packed` (packed once before by nof_mlp_pack for the same desc) and the pose table `tf` [F,12] hold the UPDATED weights and
 * poses when the launch ends, so the next step starts at its ray marcher.  Bit-identical to the three calls. */
typedef struct {
  const NofMlpDesc* desc;                 /* host pointer, like every descriptor argument */
  void* packed;                           /* the MFMA operand image of nof_mlp_pack (device) */
  int64_t mlp_off, n_mlp, pose_off;       /* the MLP segment and the first pose entry in the flat buffers */
  int32_t F;
  float max_trans, max_rot;
  const float* c2w;                       /* [F,16] */
  float* tf;                              /* [F,12] */
  float* frame_slots;                     /* nof_pose_grad_accum's partial sums, handed back zeroed */
} NofAdamTail;
int nof_adam_step_tail(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t n_basic,
                       float lr, float lr_pose, float beta1, float beta2, float eps, int32_t step, const int32_t* skip_flags,
                       const NofAdamTail* tail, void* stream);
/* ... and with the NEXT batch's ray marcher as one more role of the launch (nof_raymarch_sample in its one-launch form: the wave
 * marcher, level <= 6, injected uniforms NULL, no frame features, no cell ids): a latency chain per ray beside Adam's streaming, for
 * the time of the longer one.  The marcher's workgroups wait inside the launch for the pose-table rows the launch's first F
 * workgroups write: d_epoch is one device uint32 that only grows -- by F per call --, epoch_target the value it has when this
 * call's rows are out (the caller keeps the running sum, starting from the counter's initial 0).  The overflow mark of the device
 * flags (bit 2 -> bit 3 when a new batch starts) is NOT moved by this marcher -- the launch's other workgroups still read bit 2;
 * the following step passes NOF_HASH_BWD_NEW_BATCH to nof_hash_encode_bwd_step instead.  Same results as nof_adam_step_tail
 * followed by nof_raymarch_sample. */
typedef struct {
  const NofSampleCfg* cfg;                /* host pointer: seed / step of the NEXT batch (d_step NULL) */
  const float* pool; const int64_t* ids; const uint32_t* occ_bits;
  int32_t sh_degree, level, max_hits, reserved;
  int64_t R;
  float* batch; float* rays_o_w; float* viewdirs_w; float* view; float* t_in_out; int32_t* n_hits;
  float* z_vals; float* pts_w; uint8_t* valid; int32_t* flags;
} NofMarchNext;
int nof_adam_step_tail_march(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t n_basic,
                             float lr, float lr_pose, float beta1, float beta2, float eps, int32_t step, const int32_t* skip_flags,
                             const NofAdamTail* tail, const NofMarchNext* next, uint32_t* d_epoch, uint32_t epoch_target,
                             void* stream);
/* ... with the step's scalars read from the device state (a captured, replayable step: nof_adam_step_dyn's arguments), and
 * nof_step_state_advance(set_step = -1) inside as well: the last workgroup to finish advances *d_state.  d_done:
 * NOF_ADAM_TAIL_DONE_WORDS device uint32, zero before the launch and zero again after it. */
#define NOF_ADAM_TAIL_DONE_WORDS 1040       /* 65 counters, one per 64-byte line */
int nof_adam_step_tail_dyn(float* params, float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t n_basic,
                           NofStepState* d_state, float lrate, float lrate_pose, float decay_rate, int32_t n_iters, float beta1,
                           float beta2, float eps, const int32_t* skip_flags, const NofAdamTail* tail, uint32_t* d_done, void* stream);
/* flags[0] |= 4 when any of grad[0, n) is not finite (the check of nof_reduce_partials, for a gradient that was summed over the
 * data-parallel ranks afterwards: every rank must skip the same step). */
int nof_grad_check(const float* grad, int64_t n, int32_t* flags, void* stream);

#ifdef __cplusplus
}
#endif
#endif
