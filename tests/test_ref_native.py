"""The oracle's restatement of the reference's NATIVE arithmetic against the reference's own kernels compiled as host
C++ (oracle/_ref/libnof_ref.so, recipe oracle/ref_build.py): multires hash encode forward / dy_dx / table-gradient
scatter / input gradient (gridencoder.cu:47-365), the occupied-voxel sampler walk (common.cu:41-105), the octree
ray-trace post-process (common.cu:129-149) and the texture kernel's barycentric UV (common.cu:171-219).

These are CPU tests.  Where /root/reference is mounted the library is (re)built on demand; elsewhere (the GPU box) the
prebuilt .so that travelled with the snapshot is used; with neither, the module is skipped.
"""
import numpy as np
import pytest
import torch

from oracle import nof_oracle as O
from oracle import ref_native as RN

pytestmark = pytest.mark.skipif(RN.load() is None, reason='oracle/_ref not built and /root/reference absent')

GEOS = {
    'L16_T14_256': dict(n_levels=16, log2_hashmap_size=14, desired_resolution=256),
    'L16_T19_256': dict(n_levels=16, log2_hashmap_size=19, desired_resolution=256),
    'L16_T19_512': dict(n_levels=16, log2_hashmap_size=19, desired_resolution=512),   # levels 12/15 land on 257/513
    'L4_T22_128': dict(n_levels=4, log2_hashmap_size=22, desired_resolution=128),
    'L16_T22_512': dict(n_levels=16, log2_hashmap_size=22, desired_resolution=512),
}


def _points(rng, B):
    x = rng.random((B, 3)).astype(np.float32)
    x[0] = 0.0
    x[1] = 1.0
    x[2] = [0.0, 1.0, 0.5]
    x[3] = [-1e-6, 0.5, 0.5]                    # out of range: zero output, zero gradient
    x[4] = [0.5, 1.0000001, 0.5]
    x[5] = [0.25, 0.5, 0.75]                    # exact binary fractions land on cell boundaries at power-of-two levels
    x[6] = np.float32(1.0) - np.float32(2 ** -24)
    x[7] = np.float32(2 ** -20)
    return x


@pytest.mark.parametrize('name', list(GEOS))
def test_level_constants_bitwise(name):
    """scale = exp2f(level*S)*H - 1, resolution = ceil(scale)+1 (gridencoder.cu:155-156) -- the oracle's (and the
    product's) host-evaluated table equals what the reference's code yields with this host's exp2f."""
    geo = O.HashGeometry(**GEOS[name])
    scale, res = RN.level_constants(geo.L, float(geo.S), geo.H)
    assert np.array_equal(scale.view(np.uint32), geo.scale.view(np.uint32))
    assert np.array_equal(res.astype(np.int64), geo.resolution)


@pytest.mark.parametrize('name', list(GEOS))
def test_forward_and_dydx_match_reference_kernel(name):
    geo = O.HashGeometry(**GEOS[name])
    rng = np.random.default_rng(1)
    B = 600
    x = _points(rng, B)
    table = ((rng.random((geo.n_entries, 2)) * 2 - 1) * 0.5).astype(np.float32)
    out_ref, dy_ref = RN.grid_encode_forward(x, table, geo.offsets, geo.L, float(geo.S), geo.H, calc_grad_inputs=True)
    xt = torch.from_numpy(x).requires_grad_(True)
    tt = torch.from_numpy(table)
    out = O.hash_encode(xt, tt, geo)                                  # [B, L*C]
    want = np.transpose(out_ref, (1, 0, 2)).reshape(B, geo.L * geo.C)  # grid.py:64 permute
    # same operation order in float32, no contraction: the restatement is bit-identical to the reference kernel
    assert np.array_equal(out.detach().numpy().view(np.uint32), want.view(np.uint32))
    assert np.all(want[3] == 0) and np.all(want[4] == 0)
    # dy_dx [B, L, D, C] (gridencoder.cu:204) through kernel_input_backward (:340-365) == autograd of the restatement
    g = rng.standard_normal((geo.L, B, geo.C)).astype(np.float32)
    _, gi_ref = RN.grid_encode_backward(g, x, table, geo.offsets, geo.L, float(geo.S), geo.H, dy_dx=dy_ref)
    gt = torch.from_numpy(np.transpose(g, (1, 0, 2)).reshape(B, -1).copy())
    (gx,) = torch.autograd.grad(out, xt, gt)
    scale = np.abs(gi_ref).max()
    assert np.abs(gx.numpy() - gi_ref).max() <= 2e-6 * scale
    assert np.all(gi_ref[3] == 0) and np.all(gx.numpy()[3] == 0)


@pytest.mark.parametrize('name', ['L16_T14_256', 'L16_T19_512', 'L4_T22_128'])
def test_table_gradient_matches_reference_scatter(name):
    geo = O.HashGeometry(**GEOS[name])
    rng = np.random.default_rng(2)
    B = 400
    x = _points(rng, B)
    x[8:200] = x[200:201] + rng.random((192, 3)).astype(np.float32) * 1e-3   # a run of samples inside one fine cell
    x = np.clip(x, -1, 2).astype(np.float32)
    table = ((rng.random((geo.n_entries, 2)) * 2 - 1) * 1e-2).astype(np.float32)
    g = rng.standard_normal((geo.L, B, geo.C)).astype(np.float32)
    ge_ref, _ = RN.grid_encode_backward(g, x, table, geo.offsets, geo.L, float(geo.S), geo.H)
    tt = torch.from_numpy(table).requires_grad_(True)
    out = O.hash_encode(torch.from_numpy(x), tt, geo)
    (gt,) = torch.autograd.grad(out, tt, torch.from_numpy(np.transpose(g, (1, 0, 2)).reshape(B, -1).copy()))
    gt = gt.numpy()
    assert np.array_equal(gt != 0, ge_ref != 0)                      # the same rows are touched
    # summation ORDER differs (sequential atomics vs index_add): float32 rounding of sums only
    assert np.abs(gt - ge_ref).max() <= 2e-6 * np.abs(ge_ref).max()


@pytest.mark.parametrize('name', ['L16_T14_256', 'L16_T19_256', 'L16_T19_512'])
def test_corner_rows_bit_identical(name):
    """The integer work (cell, dense index / hash, modulo) of get_grid_index, read off the reference's scatter: with a
    one-hot gradient the touched rows of level l ARE the eight corner rows of that sample."""
    geo = O.HashGeometry(**GEOS[name])
    rng = np.random.default_rng(3)
    x = _points(rng, 40)
    x = x[[0, 1, 2, 5, 6, 7] + list(range(8, 40))]
    table = np.zeros((geo.n_entries, 2), np.float32)
    rows = O.hash_corner_indices(x, geo)                             # [B, L, 8] absolute rows
    for b in range(x.shape[0]):
        g = np.zeros((geo.L, 1, 2), np.float32)
        g[:, 0, 0] = 1.0
        ge, _ = RN.grid_encode_backward(g, x[b:b + 1], table, geo.offsets, geo.L, float(geo.S), geo.H)
        touched = np.flatnonzero(ge[:, 0])
        pos = (x[b] * geo.scale[:, None]).astype(np.float32) + np.float32(0.5)
        frac = pos - np.floor(pos)
        for l in range(geo.L):
            lo, hi = geo.offsets[l], geo.offsets[l + 1]
            got = set(touched[(touched >= lo) & (touched < hi)].tolist())
            want = set()
            for idx in range(8):
                w = np.prod([frac[l, d] if (idx >> d) & 1 else 1 - frac[l, d] for d in range(3)])
                if w != 0:                                           # zero-weight corners leave no trace
                    want.add(int(rows[b, l, idx]))
            assert got == want, (name, b, l)


def test_half_type_is_ieee_binary16():
    rng = np.random.default_rng(4)
    x = np.concatenate([rng.standard_normal(20000) * 10 ** rng.uniform(-9, 5, 20000),
                        [0.0, -0.0, 65504.0, 65520.0, 65519.99, 6e-8, 2.98e-8, 2.9802322e-8, 1e-10, np.inf, -np.inf,
                         0.00006103515625, 0.000061005353927612305]]).astype(np.float32)
    bits, back = RN.half_roundtrip(x)
    want = x.astype(np.float16)
    assert np.array_equal(bits, want.view(np.uint16))
    assert np.array_equal(back, want.astype(np.float32))


def test_reference_fp16_table_path_vs_fp32():
    """What the reference's own autocast path (fp16 table, fp16 features, gridencoder.cu with scalar_t = at::Half) does
    to the encode relative to fp32: the size of the reference's own 16-bit deviation, recorded as the yardstick for the
    product's 1e-3 target (the product keeps the table and the features in fp32)."""
    geo = O.HashGeometry(**GEOS['L16_T19_256'])
    rng = np.random.default_rng(5)
    x = rng.random((2000, 3)).astype(np.float32)
    table = ((rng.random((geo.n_entries, 2)) * 2 - 1) * 0.3).astype(np.float32)
    o32, _ = RN.grid_encode_forward(x, table, geo.offsets, geo.L, float(geo.S), geo.H, calc_grad_inputs=False)
    o16, _ = RN.grid_encode_forward(x, table.astype(np.float16), geo.offsets, geo.L, float(geo.S), geo.H, calc_grad_inputs=False)
    rel = np.abs(o16.astype(np.float32) - o32).max() / np.abs(o32).max()
    assert 1e-4 < rel < 4e-3, rel          # binary16 has 11 significant bits: ~5e-4 per rounding, a few roundings


# ------------------------------------------------------------------------------------------------ sampler walk (K4)
def _boxes(rng, R, H):
    z = np.zeros((R, H, 2), np.float32)
    for r in range(R):
        n = rng.integers(1, H + 1)
        edges = np.sort(rng.random(2 * n).astype(np.float32) * 2 + np.float32(0.3))
        z[r, :n] = edges.reshape(n, 2)
        if n > 2 and r % 5 == 0:
            z[r, 1, 1] = z[r, 1, 0]                                   # a zero-length (clipped) box in the middle
    return z


def test_sampler_walk_matches_reference_kernel():
    rng = np.random.default_rng(6)
    R, H, S = 64, 7, 33
    z = _boxes(rng, R, H)
    z[3, 0] = 0                                                       # first box is a terminator: kernel returns at once
    total = (z[:, :, 1] - z[:, :, 0]).sum(-1, dtype=np.float32)
    zc = (rng.random((R, S)).astype(np.float32) * total[:, None]).astype(np.float32)
    zc[:, 0] = 0
    zc[:, 1] = z[:, 0, 1] - z[:, 0, 0]                                # exactly the first box's length (<= comparison)
    zc[:, 2] = np.minimum(total, total * np.float32(0.999999))
    want, spun = RN.sample_rays_uniform_occupied_voxels(z, zc)
    got, spin = O.walk_boxes(z, zc, return_spin=True)
    assert not spun and not spin.any()
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.all(want[3] == 0)


def test_sampler_walk_end_of_list_and_spin_cases():
    """common.cu:57-95: a remaining distance within eps=1e-4 past the last box returns that box's far end; a larger one
    is the reference's printf + `while(1){}` hang, which the oracle reports as a flag (the product raises its error flag)."""
    z = np.zeros((4, 3, 2), np.float32)
    z[0] = [[1.0, 1.5], [2.0, 2.25], [0, 0]]          # terminated list
    z[1] = [[1.0, 1.5], [2.0, 2.25], [3.0, 3.5]]      # full list
    z[2] = z[0]
    z[3] = z[1]
    zc = np.array([[0.75 + 5e-5], [1.25 + 5e-5], [0.75 + 1e-3], [1.25 + 1e-3]], np.float32)
    want, spun = RN.sample_rays_uniform_occupied_voxels(z[:2], zc[:2])
    got, spin = O.walk_boxes(z[:2], zc[:2], return_spin=True)
    assert not spun and not spin.any()
    assert np.array_equal(got, want) and got[0, 0] == np.float32(2.25) and got[1, 0] == np.float32(3.5)
    for r in (2, 3):
        _, spun = RN.sample_rays_uniform_occupied_voxels(z[r:r + 1], zc[r:r + 1])
        _, spin = O.walk_boxes(z[r:r + 1], zc[r:r + 1], return_spin=True)
        assert spun and spin.all()


# --------------------------------------------------------------------------------------- ray-trace post-process (K5)
def _kaolin_call(ray_index, dio, N_rays):
    """Utils.py:466-470 around the kernel."""
    ri = torch.from_numpy(ray_index)
    ids, counts = torch.unique_consecutive(ri, return_counts=True)
    max_int = int(counts.max())
    start = torch.cat([torch.tensor([0]), torch.cumsum(counts[:-1], 0)])
    return RN.postprocess_octree_ray_tracing(ray_index, dio, ids.numpy(), start.numpy(), max_int, N_rays)


def test_postprocess_matches_reference_kernel_on_traced_rays():
    rng = np.random.default_rng(7)
    occ = rng.random((16, 16, 16)) < 0.2
    R = 300
    o = (rng.standard_normal((R, 3)) * 0.2 + np.array([0, 0, 2.5])).astype(np.float32)
    tgt = (rng.random((R, 3)) * 1.6 - 0.8).astype(np.float32)
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).astype(np.float32)
    d[:10] = [0, 0, -1]                                               # axis-aligned rays (zero direction components)
    o[:10, :2] = (np.arange(10)[:, None] / 8.0 - 0.6 + np.array([0, 0.03])).astype(np.float32)
    fr, fio, fc = O.trace_rays_flat(occ, o, d)
    assert len(fr) > 800
    want = _kaolin_call(fr, fio, R)
    got = O.postprocess_hits(fr, fio, R)
    assert got.shape == want.shape and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    tio, cid, nh = O.trace_rays(occ, o, d)
    H = tio.shape[1]
    assert np.array_equal(tio, want[:, :H]) and not want[:, H:].any()


def test_postprocess_filters_match_reference_kernel():
    """in==0 / out==0 terminate the run, in>out and |out-in|<1e-4 are skipped (common.cu:139-142)."""
    ray_index = np.array([0, 0, 0, 0, 2, 2, 2, 5, 5, 5, 5, 5], np.int64)
    dio = np.array([[1.0, 1.2], [1.2, 1.20005], [1.5, 1.4], [1.6, 1.9],
                    [0.0, 0.4], [0.5, 0.6], [0.7, 0.8],
                    [0.3, 0.5], [0.5, 0.0], [0.6, 0.7], [0.8, 0.9], [1.0, 1.1]], np.float32)
    want = _kaolin_call(ray_index, dio, 7)
    got = O.postprocess_hits(ray_index, dio, 7)
    assert np.array_equal(got, want)
    assert np.array_equal(got[0, :2], np.array([[1.0, 1.2], [1.6, 1.9]], np.float32)) and not got[2].any()
    assert np.array_equal(got[5, 0], np.array([0.3, 0.5], np.float32)) and not got[5, 1:].any()


# ----------------------------------------------------------------------------------------- texture kernel (K12)
def test_barycentric_uv_reference_kernel_interpolates():
    """rayColorToTextureImageKernel (common.cu:171-219) compiled as host code: for points inside a triangle the UV is the
    barycentric interpolation of the corner UVs (float64 NumPy as the yardstick)."""
    rng = np.random.default_rng(8)
    V = rng.standard_normal((30, 3)).astype(np.float32)
    F = np.stack([rng.permutation(30)[:3] for _ in range(50)]).astype(np.int64)
    uvt = rng.random((30, 2)).astype(np.float32)
    n = 500
    fid = rng.integers(0, 50, n)
    w = rng.dirichlet([1, 1, 1], n)
    P = np.einsum('nk,nkd->nd', w, V[F[fid]].astype(np.float64)).astype(np.float32)
    uv = RN.ray_color_to_texture_image(F, V, P, fid, uvt)
    want = np.einsum('nk,nkd->nd', w, uvt[F[fid]].astype(np.float64))
    assert np.abs(uv - want).max() < 2e-4
