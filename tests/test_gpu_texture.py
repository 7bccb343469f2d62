"""Texture bake: nof_bary_uv against the reference's own kernel compiled as host code (common.cu:171-238
through oracle/_ref), nof_texture_bake_frame against the CPU restatement of one keyframe of mesh_texture_from_train_images
(nerf_runner.py:1499-1535), and the runner's method end to end on an analytic textured ellipsoid."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import ref_native as RN
from oracle import texture_oracle as TO
from tests import util as U

pytestmark = pytest.mark.gpu


def _ellipsoid_mesh(n_lat=10, n_lon=16, axes=(0.5, 0.35, 0.42)):
    th = np.linspace(0, np.pi, n_lat + 1)[1:-1]
    ph = np.linspace(0, 2 * np.pi, n_lon, endpoint=False)
    v = [[0, 0, 1.0]] + [[np.sin(t) * np.cos(p), np.sin(t) * np.sin(p), np.cos(t)] for t in th for p in ph] + [[0, 0, -1.0]]
    v = np.array(v) * np.array(axes)
    f = []
    ring = lambda i: 1 + i * n_lon
    for k in range(n_lon):
        f.append([0, ring(0) + k, ring(0) + (k + 1) % n_lon])
        f.append([len(v) - 1, ring(n_lat - 2) + (k + 1) % n_lon, ring(n_lat - 2) + k])
    for i in range(n_lat - 2):
        for k in range(n_lon):
            a, b = ring(i) + k, ring(i) + (k + 1) % n_lon
            c, d = ring(i + 1) + k, ring(i + 1) + (k + 1) % n_lon
            f += [[a, c, b], [b, c, d]]
    return v, np.array(f, dtype=np.int64)


@pytest.mark.skipif(RN.load() is None, reason='oracle/_ref not built and /root/reference absent')
def test_bary_uv_matches_reference_kernel(nof):
    rng = np.random.default_rng(0)
    V, F = _ellipsoid_mesh()
    V = V.astype(np.float32)
    uvt = (rng.random((len(V), 2)) * 511).astype(np.float32)
    n = 4000
    fid = rng.integers(0, len(F), n)
    w = rng.dirichlet([1, 1, 1], n)
    P = np.einsum('nk,nkd->nd', w, V[F[fid]].astype(np.float64)).astype(np.float32)
    want = RN.ray_color_to_texture_image(F, V, P, fid, uvt)
    uvs = torch.zeros(n, 2, device='cuda')
    nof.call('nof_bary_uv', U.dev(F), U.dev(V), U.dev(P), U.dev(fid.astype(np.int64)), U.dev(uvt), n, uvs)
    torch.cuda.synchronize()
    assert np.abs(uvs.cpu().numpy() - want).max() < 2e-3             # texel units (|uv| up to 511): float32 rounding of the blend


def _ray_bary(o, d, A, B, Cc):
    """float64 barycentric coordinates (and the ray parameter) of the ray o + t d on the plane of triangle A B C"""
    e1, e2 = B - A, Cc - A
    p = np.cross(d, e2)
    det = p @ e1
    tv = o - A
    u = (p @ tv) / det
    q = np.cross(tv, e1)
    w = (d @ q) / det
    return min(u, w, 1 - u - w), (e2 @ q) / det


def test_texture_bake_frame_matches_oracle(nof):
    from bundlesdf_amd.mesh import Mesh
    from bundlesdf_amd.synthetic import look_at_cv
    V, F = _ellipsoid_mesh()
    T, H, W = 96, 60, 80
    m = Mesh(V, F).unwrap(T)
    verts, faces = m.vertices.astype(np.float32), m.faces
    uvs_tex = (m.uv * (T - 1)).astype(np.float32)
    K = np.array([[90.0, 0, 40.0], [0, 90.0, 30.0], [0, 0, 1]])
    rng = np.random.default_rng(1)
    tex = torch.zeros(T, T, 3, device='cuda')
    wtex = torch.zeros(T, T, device='cuda')
    zbuf = torch.empty(H * W, dtype=torch.int64, device='cuda')
    owner = torch.empty(T * T, dtype=torch.int32, device='cuda')
    tex_o, wtex_o = np.zeros((T, T, 3)), np.zeros((T, T))
    K4 = (C.c_float * 4)(90.0, 90.0, 40.0, 30.0)
    n_edge_pixels = 0
    for cam in ([1.6, 0.2, 0.3], [-0.4, 1.5, -0.5]):
        cam_in_ob = look_at_cv(np.array(cam))
        ob_in_cam = np.linalg.inv(cam_in_ob)
        rgb = rng.integers(0, 255, size=(H, W, 3)).astype(np.float32)
        mask = (rng.random((H, W)) > 0.1).astype(np.uint8)
        nof.call('nof_texture_bake_frame', (C.c_float * 12)(*ob_in_cam[:3, :4].astype(np.float32).reshape(-1)), K4, H, W,
                 U.dev(verts), U.dev(faces), faces.shape[0], U.dev(uvs_tex), U.dev(mask), U.dev(rgb), C.c_float(0.05), T, zbuf,
                 owner, tex, wtex)
        torch.cuda.synchronize()
        tri, depth = TO.bake_frame(ob_in_cam, K, H, W, verts, faces, uvs_tex, mask, rgb, 0.05, T, tex_o, wtex_o)
        z = zbuf.cpu().numpy().astype(np.uint64)
        got_tri = np.where(z == np.uint64(0xFFFFFFFFFFFFFFFF), -1, (z & np.uint64(0xFFFFFFFF)).astype(np.int64)).reshape(H, W)
        got_depth = (z >> np.uint64(32)).astype(np.uint32).view(np.float32).reshape(H, W)
        same = got_tri == tri
        # Round 6 ("passes at 97 %, the 3 % never characterised"): measured, the float32 rasteriser and the
        # float64 ray caster agree on EVERY pixel of both views (tools/texture_edge_probe.py, profiles/r06_q_texture_probe.txt).
        # What may legitimately differ is a pixel whose ray passes within float32 rounding of a triangle edge: any such pixel must
        # be exactly that -- on the triangle the device chose, the float64 ray lies within 1e-4 (barycentric) of the boundary at
        # the oracle's depth -- and there may be at most a handful.
        n_edge_pixels += int((~same).sum())
        assert (~same).sum() <= 4, int((~same).sum())
        o64 = -ob_in_cam[:3, :3].T @ ob_in_cam[:3, 3]
        for (y, x) in np.argwhere(~same):
            d64 = ob_in_cam[:3, :3].T @ np.array([(x - K[0, 2]) / K[0, 0], (y - K[1, 2]) / K[1, 1], 1.0])
            f_dev, f_orc = got_tri[y, x], tri[y, x]
            f = f_dev if f_dev >= 0 else f_orc                       # (the device saw nothing: the ray grazes the oracle's triangle)
            mb, tt = _ray_bary(o64, d64, *(verts[faces[f, k]].astype(np.float64) for k in range(3)))
            assert abs(mb) < 1e-4, (y, x, f_dev, f_orc, mb)
            if f_dev >= 0 and f_orc >= 0:
                assert abs(tt - depth[y, x]) < 1e-3, (y, x, tt, depth[y, x])
        hit = same & (tri >= 0)
        assert hit.sum() > 500 and np.abs(got_depth[hit] - depth[hit]).max() < 1e-4
    w_got, t_got = wtex.cpu().numpy(), tex.cpu().numpy()
    cover = (w_got > 0) | (wtex_o > 0)
    agree = (w_got == wtex_o) & (np.abs(t_got - tex_o).max(-1) < 1e-3)
    # every texel's weight and colour equal the oracle's, except the (at most four) texels an edge pixel above would have painted
    assert cover.sum() > 800 and int((~agree[cover]).sum()) <= 2 * n_edge_pixels, (cover.sum(), int((~agree[cover]).sum()), n_edge_pixels)


def test_runner_texture_bake_end_to_end(nof):
    """NerfRunner.mesh_texture_from_train_images on the analytic textured ellipsoid (no training needed: the mesh is the
    analytic surface in normalised space): baked texel colours against the procedural texture evaluated at the texel's point."""
    from bundlesdf_amd import synthetic
    from bundlesdf_amd.config import default_cfg
    from bundlesdf_amd.mesh import Mesh
    from bundlesdf_amd.nerf_runner import NerfRunner
    pool = synthetic.make_pool(n_frames=8, H=240, W=320, fx=300.0, seed=0, pose_noise=False)
    cfg = default_cfg(n_step=10, N_rand=512, num_levels=8, log2_hashmap_size=14, finest_res=128, base_res=16, far=1.0,
                      sc_factor=pool['sc_factor'], translation=pool['translation'])
    runner = NerfRunner(cfg, pool['rgbs'], depths=pool['depths'], masks=pool['masks'], normal_maps=None, poses=pool['poses'],
                        K=pool['K'], build_octree_pcd=synthetic.PointCloud(pool['pcd_normalized']), precision='fp16x3')
    V, F = _ellipsoid_mesh(24, 48, axes=synthetic.SEMI_AXES)
    Vn = (V + pool['translation']) * pool['sc_factor']               # normalised object space (nerf_helpers.py:236-239)
    rgbs_raw = (pool['rgbs'] * 255.0).astype(np.float32)             # what bundlesdf.py keeps as rgbs_raw (bundlesdf.py:711)
    tm = runner.mesh_texture_from_train_images(Mesh(Vn, F), rgbs_raw, tex_res=256)
    assert tm.uv.shape == (len(tm.vertices), 2) and len(tm.vertices) == 3 * len(tm.faces)
    assert tm.texture.shape == (256, 256, 3) and tm.texture.dtype == np.uint8
    # texels inside triangles: colour vs the analytic texture at the texel's 3-D point
    img = tm.texture[::-1].astype(np.float64)                        # back to v-down rows (row = v * (T-1))
    rng = np.random.default_rng(0)
    fsel = rng.integers(0, len(tm.faces), 4000)
    w = rng.dirichlet([4, 4, 4], 4000)
    P = np.einsum('nk,nkd->nd', w, tm.vertices[tm.faces[fsel]])
    uv = np.einsum('nk,nkd->nd', w, tm.uv[tm.faces[fsel]]) * 255
    col = img[np.rint(uv[:, 1]).astype(int), np.rint(uv[:, 0]).astype(int)]
    p = P / pool['sc_factor'] - pool['translation']
    want = 255 * (0.5 + 0.5 * np.stack([np.sin(40 * p[:, 0] + 1.0) * np.cos(25 * p[:, 1]), np.sin(30 * p[:, 1] + 2.0) * np.cos(35 * p[:, 2]),
                                        np.sin(45 * p[:, 2] + 0.5) * np.cos(20 * p[:, 0])], -1))
    painted = col.sum(-1) > 0
    # like the reference's splat, a texel is painted only when some pixel's surface point rounds to it: coverage follows the
    # ratio of image pixels on the object to texels on the mesh
    assert painted.mean() > 0.3, painted.mean()
    err = np.abs(col[painted] - want[painted]).mean()
    print(f'texture bake: {painted.mean():.2f} of the sampled surface painted, mean colour error {err:.1f}/255, coverage {tm.texture_coverage:.3f}')
    assert err < 30.0
    out = tm.export('/tmp/nof_textured_test.obj')
    assert open(out).read().count('vt ') == len(tm.vertices)
