"""Device ray-pool construction against its NumPy restatement (bundlesdf_amd/rays.py, itself pinned to the reference's own
functions by tests/test_oracle.py): same rays in the same order; direction / colour / depth / mask / frame columns
bit-equal, near / far equal up to the last float32 digit (NumPy's BLAS matmul may fuse multiply-adds in float64)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _runner(nof, n_frames=3, device_pool=False, **over):
    from bundlesdf_amd import synthetic
    from bundlesdf_amd.config import default_cfg
    from bundlesdf_amd.nerf_runner import NerfRunner
    pool = synthetic.make_pool(n_frames=n_frames, H=240, W=320, fx=300.0, seed=2)
    cfg = default_cfg(n_step=10, N_rand=512, num_levels=8, log2_hashmap_size=14, finest_res=128, far=1.0,
                      sc_factor=pool['sc_factor'], translation=pool['translation'], device_ray_pool=device_pool, **over)
    r = NerfRunner(cfg, pool['rgbs'], depths=pool['depths'], masks=pool['masks'], normal_maps=None, poses=pool['poses'],
                   K=pool['K'], build_octree_pcd=synthetic.PointCloud(pool['pcd_normalized']), precision='bf16')
    return r, pool


@pytest.mark.parametrize("denoise", [True, False])
def test_device_ray_pool_matches_numpy(nof, denoise):
    host, _ = _runner(nof, device_pool=False, denoise_depth_use_octree_cloud=denoise)
    dev, _ = _runner(nof, device_pool=True, denoise_depth_use_octree_cloud=denoise)
    a, b = host.rays.cpu().numpy(), dev.rays.cpu().numpy()
    assert a.shape == b.shape and a.shape[0] > 10000, (a.shape, b.shape)
    assert np.array_equal(a[:, :10], b[:, :10])                      # dir, rgb, depth, mask, frame, type: bit-equal
    assert np.abs(a[:, 10:] - b[:, 10:]).max() <= 2e-7 * np.abs(a[:, 10:]).max()
    assert (a[:, 10:] != b[:, 10:]).mean() < 1e-3


@pytest.mark.timeout(900)
def test_device_ray_pool_at_the_bench_size_matches_the_restatement(nof):
    """The pool bench.py builds -- 64 keyframes of 640x480, every stage at its real size (k_dilate_1d with k = 100 / 60, k_frame_rays
    over 19.7 M pixels, the octree trace of every masked ray, k_cloud_filter's nearest-point search of 2.4 M back-projected points,
    k_compact_rows) -- against rays.py, the NumPy restatement of nerf_runner.py:178-195,246-316 that tests/test_host_logic.py pins on
    a reference-driven run of make_frame_rays: the same rays in the same order, columns 0-9 bit-equal (this
    was tested on 3 frames of 320x240 only)."""
    from bundlesdf_amd import synthetic
    from bundlesdf_amd.config import default_cfg
    from bundlesdf_amd.nerf_runner import NerfRunner
    pool = synthetic.make_pool(n_frames=64, H=480, W=640, fx=600.0, seed=0, analytic_bounds=True)
    out = []
    for device_pool in (True, False):
        cfg = default_cfg(n_step=10, N_rand=4096, num_levels=8, log2_hashmap_size=14, finest_res=128, sc_factor=pool['sc_factor'],
                          translation=pool['translation'], far=1.0, save_octree_clouds=False,
                          device_ray_pool=device_pool)
        r = NerfRunner(cfg, pool['rgbs'], depths=pool['depths'], masks=pool['masks'], normal_maps=None, poses=pool['poses'],
                       K=pool['K'], build_octree_pcd=synthetic.PointCloud(pool['pcd_normalized']), precision='fp16x3')
        out.append(r.rays.cpu().numpy())
        del r
        torch.cuda.empty_cache()
    b, a = out
    assert a.shape == b.shape and a.shape[0] > 2_000_000, (a.shape, b.shape)
    assert np.array_equal(a[:, :10], b[:, :10])                      # dir, rgb, depth, mask, frame, type: bit-equal, same order
    assert np.abs(a[:, 10:] - b[:, 10:]).max() <= 2e-7 * np.abs(a[:, 10:]).max()
    assert (a[:, 10:] != b[:, 10:]).mean() < 1e-3
    print(f'ray pool at the bench size: {a.shape[0]} rays of 64 x 640 x 480 pixels, columns 0-9 bit-equal; near / far differ in '
          f'{(a[:, 10:] != b[:, 10:]).mean():.2e} of the entries (last float32 digit)')


def test_mask_dilate_matches_scipy(nof):
    from bundlesdf_amd.rays import dilate_mask
    rng = np.random.default_rng(0)
    m = (rng.random((97, 131)) < 0.01).astype(np.uint8)              # asymmetric, sparse: anchor conventions show
    m[0, 0] = m[-1, -1] = 1
    for k in (2, 5, 30, 60, 100):
        tmp = torch.empty(m.size, dtype=torch.uint8, device='cuda')
        out = torch.empty(m.size, dtype=torch.uint8, device='cuda')
        nof.call('nof_mask_dilate', torch.from_numpy(m).cuda(), m.shape[0], m.shape[1], k, tmp, out)
        assert np.array_equal(out.cpu().numpy().reshape(m.shape), dilate_mask(m, k)), k
