"""Renderer side on the MI355X: the fused dense SDF-grid query against the oracle (hash encode + sigma net per voxel,
octree mask) and the device iso-surface extractor against the numpy restatement in bundlesdf_amd/mesh.py."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import nof_oracle as O
from tests import util as U
from tests.test_gpu_ops import _mlp_setup, _pack, ODT

pytestmark = pytest.mark.gpu


def _canon(faces):
    """rotation-invariant, order-invariant face list (orientation preserved)"""
    f = np.asarray(faces)
    r = np.argmin(f, 1)
    f = np.stack([np.take_along_axis(f, ((r + k) % 3)[:, None], 1)[:, 0] for k in range(3)], 1)
    return f[np.lexsort((f[:, 2], f[:, 1], f[:, 0]))]


@pytest.mark.parametrize("ns,nc,precision", [(2, 3, 0), (3, 2, 1), (2, 3, 2)])
def test_sdf_grid_query_matches_oracle(nof, ns, nc, precision):
    L = 16
    g, geo = U.make_grids(nof, L=L, T=14)
    shape, params, desc, flat = _mlp_setup(nof, ns, nc, 0, L, precision, seed=3)
    rng = np.random.default_rng(7)
    table = (rng.uniform(-1, 1, size=(geo.n_entries, 2)) * 0.5).astype(np.float32)
    level = 3
    occ = U.random_occ(1 << level, 0.4, seed=11)
    bits = torch.zeros(((1 << level) ** 3 + 31) // 32, dtype=torch.int32, device='cuda')
    nof.call('nof_occgrid_build', U.dev(U.occ_to_coords(occ)), int(occ.sum()), level, level, bits)
    # axes like extract_mesh: np.arange(lo + 0.5 v, hi, v); nz = 45 exercises a ragged z tile, one axis pokes outside [-1,1]
    tx = np.arange(-0.93 + 0.5 * 0.11, 0.95, 0.11)
    ty = np.arange(-1.08 + 0.5 * 0.13, 1.1, 0.13)
    tz = np.arange(-0.97 + 0.5 * 0.043, 0.97, 0.043)
    nx, ny, nz = len(tx), len(ty), len(tz)
    assert nz % 32 != 0
    packed = _pack(nof, desc, flat)
    out = torch.full((nx, ny, nz), -7.0, device='cuda')
    axes = [U.dev(a.astype(np.float32)) for a in (tx, ty, tz)]
    nof.call('nof_sdf_grid_query', C.byref(g), C.byref(desc), packed, U.dev(table), bits, level, axes[0], axes[1], axes[2],
             nx, ny, nz, C.c_float(1.0), out)
    got = out.cpu().numpy()
    # oracle: the reference's sequence (nerf_runner.py:1363-1386, 1307-1347)
    q = np.stack(np.meshgrid(tx, ty, tz, indexing='ij'), -1).astype(np.float32).reshape(-1, 3)
    n = 1 << level
    c = np.floor(np.clip(np.float32(n) * (q + np.float32(1)) / np.float32(2), 0, n - 1)).astype(np.int64)
    valid = occ[c[:, 0], c[:, 1], c[:, 2]]
    qc = np.clip(q[valid], -1, 1)
    x01 = (torch.from_numpy(qc) + 1) * 0.5
    feat = O.hash_encode(x01, torch.from_numpy(table), geo)
    ref = np.ones(len(q), dtype=np.float32)
    ref[valid] = O.mlp_forward_sdf(shape, params, feat, ODT[precision]).detach().numpy().reshape(-1)
    ref = ref.reshape(nx, ny, nz)
    assert ((got == 1.0) == (ref == 1.0)).all()                       # the octree mask is index work: exact
    scale = np.abs(ref[ref != 1.0]).max()
    err = np.abs(got - ref).max() / scale
    assert err < {0: 2e-5, 1: 4e-3, 2: 5e-4}[precision], err
    # no octree: every voxel is evaluated
    nof.call('nof_sdf_grid_query', C.byref(g), C.byref(desc), packed, U.dev(table), None, level, axes[0], axes[1], axes[2],
             nx, ny, nz, C.c_float(1.0), out)
    got2 = out.cpu().numpy()
    assert np.abs(got2[ref != 1.0] - got[ref != 1.0]).max() == 0.0 and (got2 != 1.0).all()


@pytest.mark.parametrize("shape_", [(33, 29, 41), (64, 64, 64)])
def test_marching_tetrahedra_gpu_matches_numpy(nof, shape_):
    from bundlesdf_amd.mesh import marching_tetrahedra
    from bundlesdf_amd.mesh_gpu import marching_tetrahedra_gpu
    nx, ny, nz = shape_
    g = np.stack(np.meshgrid(np.linspace(-1, 1, nx), np.linspace(-1, 1, ny), np.linspace(-1, 1, nz), indexing='ij'), -1)
    rng = np.random.default_rng(1)
    vol = (np.sqrt((g[..., 0] / 0.8) ** 2 + (g[..., 1] / 0.55) ** 2 + (g[..., 2] / 0.7) ** 2) - 1.0).astype(np.float32)
    vol += (rng.normal(size=vol.shape) * 0.02).astype(np.float32)    # noise: all tetrahedron cases, small components
    vol[5, 5, 5] = 0.0                                                # a value exactly on the iso level
    vol[:3] = 1.0                                                     # the 'outside the octree' plateau
    v_ref, f_ref = marching_tetrahedra(vol, 0.0)
    v_gpu, f_gpu = marching_tetrahedra_gpu(torch.from_numpy(vol).cuda(), 0.0)
    assert v_gpu.shape == v_ref.shape and np.abs(v_gpu - v_ref).max() == 0.0        # same keys, same fp64 interpolation
    assert f_gpu.shape == f_ref.shape and (_canon(f_gpu) == _canon(f_ref)).all()
    with pytest.raises(ValueError):
        marching_tetrahedra_gpu(torch.ones(8, 8, 8, device='cuda'), 0.0)


@pytest.mark.parametrize("shape_", [(33, 29, 41), (64, 64, 64), (3, 3, 3), (97, 5, 3)])
def test_marching_cubes_gpu_equals_the_oracle(nof, shape_):
    """The device extractor the runner uses by default against oracle/marching_cubes.py (the restatement of what the reference's
    skimage call computes, nerf_runner.py:1388-1394): the SAME vertices in the same order (sorted edge keys, float64
    interpolation) and the SAME triangles with the same winding (compared as a set of rows: the oracle lists them case by case,
    the device cell by cell)."""
    from bundlesdf_amd.mesh_gpu import marching_cubes_gpu
    from oracle import marching_cubes as MC
    nx, ny, nz = shape_
    g = np.stack(np.meshgrid(np.linspace(-1, 1, nx), np.linspace(-1, 1, ny), np.linspace(-1, 1, nz), indexing='ij'), -1)
    rng = np.random.default_rng(nx)
    vol = (np.sqrt((g[..., 0] / 0.8) ** 2 + (g[..., 1] / 0.55) ** 2 + (g[..., 2] / 0.7) ** 2) - 1.0).astype(np.float32)
    vol += (rng.normal(size=vol.shape) * 0.05).astype(np.float32)    # noise: every sign configuration, ambiguous faces included
    if nx > 8:
        vol[5, min(5, ny - 2), 2] = 0.0                                # a value exactly on the iso level
        vol[:3] = 1.0                                                 # the 'outside the octree' plateau
    v_ref, f_ref = MC.marching_cubes(vol, 0.0)
    v_gpu, f_gpu = marching_cubes_gpu(torch.from_numpy(vol).cuda(), 0.0)
    assert v_gpu.shape == v_ref.shape and np.abs(v_gpu - v_ref).max() == 0.0
    order = lambda f: f[np.lexsort(f.T[::-1])]
    assert f_gpu.shape == f_ref.shape and np.array_equal(order(f_gpu), order(f_ref))
    with pytest.raises(ValueError):
        marching_cubes_gpu(torch.ones(8, 8, 8, device='cuda'), 0.0)


def test_marching_cubes_gpu_covers_all_256_cases(nof):
    """one 2x2x2 volume per sign configuration, values of random magnitude: every row of the case table is exercised"""
    from bundlesdf_amd.mesh_gpu import marching_cubes_gpu
    from oracle import marching_cubes as MC
    rng = np.random.default_rng(0)
    for case in range(1, 255):
        mag = rng.uniform(0.1, 1.0, 8).astype(np.float32)
        sign = np.array([-1.0 if (case >> c) & 1 else 1.0 for c in range(8)], np.float32)
        vol = np.zeros((2, 2, 2), np.float32)
        for c in range(8):
            vol[c & 1, (c >> 1) & 1, c >> 2] = sign[c] * mag[c]
        v_ref, f_ref = MC.marching_cubes(vol, 0.0)
        v_gpu, f_gpu = marching_cubes_gpu(torch.from_numpy(vol).cuda(), 0.0)
        assert np.array_equal(v_gpu, v_ref) and np.array_equal(f_gpu, f_ref), case


def test_extract_mesh_sphere_end_to_end(nof):
    """A field whose SDF grid is replaced by an analytic sphere is not reachable through NerfRunner, so this drives the
    two device stages directly at a 192^3 grid and checks the surface: vertices within half a voxel of the sphere,
    closed orientable surface (every edge shared by exactly two faces, opposite directions), outward normals."""
    from bundlesdf_amd.mesh_gpu import marching_cubes_gpu, marching_tetrahedra_gpu
    n = 192
    ax = torch.linspace(-1, 1, n, device='cuda')
    gx, gy, gz = torch.meshgrid(ax, ax, ax, indexing='ij')
    vol = (torch.sqrt(gx * gx + gy * gy + gz * gz) - 0.6).contiguous()
    for extract in (marching_cubes_gpu, marching_tetrahedra_gpu):
        v, f = extract(vol, 0.0)
        _check_sphere(v, f, n)


def _check_sphere(v, f, n):
    p = v * (2.0 / (n - 1)) - 1.0
    assert np.abs(np.linalg.norm(p, axis=1) - 0.6).max() < 0.5 * 2.0 / (n - 1)
    e = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], 0)
    key = e[:, 0] * (len(v) + 1) + e[:, 1]
    rkey = e[:, 1] * (len(v) + 1) + e[:, 0]
    assert len(np.unique(key)) == len(key) and np.isin(rkey, key).all()
    nrm = np.cross(p[f[:, 1]] - p[f[:, 0]], p[f[:, 2]] - p[f[:, 0]])
    assert ((nrm * p[f].mean(1)).sum(1) > 0).mean() > 0.999


def test_bary_uv_matches_numpy(nof):
    """common.rayColorToTextureImageCUDA's arithmetic (common.cu:171-216) restated in NumPy float64: points sampled inside
    random triangles must get the barycentric blend of the triangle's texture coordinates."""
    rng = np.random.default_rng(4)
    nv, nf, n = 500, 900, 20000
    V = rng.normal(size=(nv, 3)).astype(np.float32)
    F = np.stack([rng.permutation(nv)[:3] for _ in range(nf)]).astype(np.int64)
    uv = rng.random((nv, 2)).astype(np.float32)
    fid = rng.integers(0, nf, size=n).astype(np.int64)
    w = rng.dirichlet([1, 1, 1], size=n)
    tri = V[F[fid]].astype(np.float64)                               # [n,3,3]
    P = (w[:, :, None] * tri).sum(1).astype(np.float32)
    out = torch.zeros(n, 2, device='cuda')
    nof.call('nof_bary_uv', U.dev(F), U.dev(V), U.dev(P), U.dev(fid), U.dev(uv), n, out)
    A, B, Cc = tri[:, 0], tri[:, 1], tri[:, 2]
    p = P.astype(np.float64)
    nrm = np.cross(B - Cc, B - A)
    area = (nrm * np.cross(B - A, Cc - A)).sum(1)
    w0 = (nrm * np.cross(B - p, Cc - p)).sum(1) / area
    w1 = (nrm * np.cross(Cc - p, A - p)).sum(1) / area
    ww = np.stack([w0, w1, 1 - w0 - w1], 1)
    ref = (ww[:, :, None] * uv[F[fid]].astype(np.float64)).sum(1)
    assert np.abs(ww - w).max() < 1e-4                                # the restatement recovers the sampling weights
    assert np.abs(out.cpu().numpy() - ref).max() < 2e-4


def test_marching_cubes_gpu_on_hand_checkable_cells_and_vertex_set(nof):
    """The device extractor against the paper-derived fixture of tests/test_mesh.py (20 classical cells: E vertices on the crossed
    edges, E - 2 (I + O - 1) triangles) and against the property it shares with skimage's 'lewiner', the extractor the reference
    calls (nerf_runner.py:1389): the vertex set is exactly the set of grid edges whose end points straddle the iso level."""
    from bundlesdf_amd.mesh_gpu import marching_cubes_gpu
    from tests.test_mesh import MC_HAND_CASES, _cell_volume, _crossed_edges, _vertex_edges, _noisy_sdf
    for name, inside, E, I, O in MC_HAND_CASES:
        vol = _cell_volume(inside)
        v, f = marching_cubes_gpu(torch.from_numpy(vol).cuda(), 0.0)
        assert len(v) == E and len(f) == E - 2 * (I + O - 1), name
        assert _vertex_edges(v, vol.shape) == _crossed_edges(vol), name
    for seed, noise in ((0, 0.3), (2, 3.0)):
        vol = _noisy_sdf(40, seed, noise).astype(np.float32)
        v, f = marching_cubes_gpu(torch.from_numpy(vol).cuda(), 0.0)
        edges = _crossed_edges(vol)
        assert len(v) == len(edges) and _vertex_edges(v, vol.shape) == edges


def test_wide_network_grid_query_matches_oracle(nof):
    """NeuralObjectField.query_sdf_grid for the wide networks (hidden 128 / 4 layers, BASELINE cfg5: no fused grid kernel; octree
    mask of every voxel in one launch, one compaction, encode + sigma net of the voxels inside) against the oracle's sequence
    (nerf_runner.py:1363-1386, 1307-1347) -- with and without the octree, with x slabs smaller than the grid."""
    from tests.test_gpu_step import _pair
    cfg, fld, orc, batch, rng = _pair(nof, 'fp16', 0, 4, 4, hidden=128)
    assert fld.wide
    tx = np.arange(-0.93 + 0.5 * 0.11, 0.95, 0.11)
    ty = np.arange(-1.08 + 0.5 * 0.13, 1.1, 0.13)
    tz = np.arange(-0.97 + 0.5 * 0.043, 0.97, 0.043)
    q = np.stack(np.meshgrid(tx, ty, tz, indexing='ij'), -1).astype(np.float32).reshape(-1, 3)
    n = 1 << fld.level
    occ = np.zeros((n, n, n), bool)
    bits = fld.occ_bits.cpu().numpy().view(np.uint32)
    ids = np.arange(n ** 3)
    occ.reshape(-1)[:] = (bits[ids >> 5] >> (ids & 31)) & 1
    c = np.floor(np.clip(np.float32(n) * (q + np.float32(1)) / np.float32(2), 0, n - 1)).astype(np.int64)
    valid = occ[c[:, 0], c[:, 1], c[:, 2]]
    assert 0.05 < valid.mean() < 0.95
    ref_all = orc.query_sdf(q).numpy().reshape(-1)
    for use_octree in (True, False):
        got = fld.query_sdf_grid(tx, ty, tz, outside_value=1.0, use_octree=use_octree).cpu().numpy().reshape(-1)
        m = valid if use_octree else np.ones_like(valid)
        assert (got[~m] == 1.0).all()
        err = np.abs(got[m] - ref_all[m]).max() / np.abs(ref_all[m]).max()
        assert err < 2e-3, err                                         # plain fp16 wide forward vs the oracle with fp16 operand rounding


# ----------------------------------------------------------------------------------------------------------------------
# The runner's default extractor: marching cubes with Lewiner's disambiguation on the device (nof_mcl_*) = the reference's
# skimage.measure.marching_cubes call.  Against scikit-image 0.18.3's own outputs (tests/golden/mc_skimage_vectors.npz) and against
# the oracle's restatement on volumes the fixture does not hold.
# ----------------------------------------------------------------------------------------------------------------------
def test_lewiner_gpu_equals_skimage_on_volumes(nof):
    from bundlesdf_amd.mesh_gpu import marching_cubes_lewiner_gpu
    from tests.test_mesh import _mc_golden, same_mesh
    G = _mc_golden()
    for name in ('sphere', 'blobs', 'smooth_noise', 'rough_noise', 'sdf_noisy', 'slab'):
        v, f = marching_cubes_lewiner_gpu(torch.from_numpy(G['vol_' + name]).cuda(), 0.0)
        ok, why = same_mesh(v, f, G[f'lewiner_{name}_v'], G[f'lewiner_{name}_f'])
        assert ok, (name, why)


def test_lewiner_gpu_equals_skimage_on_single_cells(nof):
    """every corner-sign configuration x 24 magnitude sets, and 4000 cells of the ambiguous configurations incl. every rare tiling the
    fixture holds, each as its own 2x2x2 volume"""
    from bundlesdf_amd.mesh_gpu import marching_cubes_lewiner_gpu
    from oracle import marching_cubes_lewiner as ML
    from tests.test_mesh import _mc_golden, same_mesh, _canon_tris
    G = _mc_golden()
    vals, gv, gnv, gf, gnf = G['cell_values'], G['cell_verts'], G['cell_nverts'], G['cell_faces'], G['cell_nfaces']
    for i in range(len(vals)):
        v, f = marching_cubes_lewiner_gpu(torch.from_numpy(vals[i].reshape(2, 2, 2).copy()).cuda(), 0.0)
        ok, why = same_mesh(v, f, gv[i][:gnv[i]], gf[i][:gnf[i]], tol=2e-5)
        assert ok, (i, why)
    V, T, N = G['amb_values'], G['amb_tris'], G['amb_ntris']
    pick = np.unique(np.concatenate([np.arange(0, len(V), 11), np.arange(len(V) - 64, len(V))]))       # (the rare tilings are at the end)
    EDGE_MID = np.array([(ML.CORNER[a] + ML.CORNER[b]) / 2.0 for a, b in ML.EDGE])
    for i in pick:
        vol = V[i].reshape(2, 2, 2).copy()
        v, f = marching_cubes_lewiner_gpu(torch.from_numpy(vol).cuda(), 0.0)
        # a vertex -> the cube edge it lies on (two integral coordinates), 12 = the centre vertex
        eid = []
        for p in v:
            integral = np.abs(p - np.round(p)) < 1e-9
            if integral.sum() < 2:
                eid.append(12)
            else:
                d = np.abs(EDGE_MID - np.where(integral, np.round(p), 0.5)).sum(1)
                eid.append(int(np.argmin(d)))
        assert _canon_tris(np.array(eid)[f]) == _canon_tris(T[i][:N[i]]), i


def test_lewiner_gpu_vertex_positions_close_to_the_iso_value(nof):
    """corner values of 1e-6 ... 1e-3 (every configuration x 6): scikit-image's vertex positions to 2e-7 voxel (its interpolation
    epsilon is np.spacing(1.0), not binary32's)"""
    from bundlesdf_amd.mesh_gpu import marching_cubes_lewiner_gpu
    from tests.test_mesh import _mc_golden, same_mesh
    G = _mc_golden()
    vals, gv, gnv, gf, gnf = G['near_values'], G['near_verts'], G['near_nverts'], G['near_faces'], G['near_nfaces']
    for i in range(0, len(vals), 2):
        v, f = marching_cubes_lewiner_gpu(torch.from_numpy(vals[i].reshape(2, 2, 2).copy()).cuda(), 0.0)
        ok, why = same_mesh(v, f, gv[i][:gnv[i]], gf[i][:gnf[i]], tol=2e-7)
        assert ok, (i, why)


@pytest.mark.parametrize("shape_", [(33, 29, 31), (64, 64, 64)])
def test_lewiner_gpu_equals_the_oracle_on_rough_volumes(nof, shape_):
    """random fields with many ambiguous cells (volumes the fixture does not hold): device == oracle, vertex for vertex and triangle
    for triangle, incl. the centre vertices; iso value 0 and a non-zero iso value"""
    from scipy.ndimage import gaussian_filter
    from bundlesdf_amd.mesh_gpu import marching_cubes_lewiner_gpu
    from oracle import marching_cubes_lewiner as ML
    from tests.test_mesh import same_mesh
    rng = np.random.default_rng(shape_[0])
    vol = gaussian_filter(rng.normal(size=shape_), 0.7).astype(np.float32)
    for iso in (0.0, 0.05):
        v, f = marching_cubes_lewiner_gpu(torch.from_numpy(vol).cuda(), iso)
        vr, fr = ML.marching_cubes(vol, float(np.float32(iso)))     # (the C ABI takes the iso value as a float32)
        ok, why = same_mesh(v, f, vr, fr, tol=1e-9)
        assert ok, (shape_, iso, why)
        assert (np.abs(v - np.round(v)) > 1e-9).all(1).sum() > 10                                   # centre vertices are present


def test_extract_mesh_uses_the_lewiner_extractor_by_default(nof):
    from bundlesdf_amd import nerf_runner
    import inspect
    src = inspect.getsource(nerf_runner.NerfRunner.extract_mesh)
    assert "self.cfg.get('mesh_extractor', 'lewiner')" in src
