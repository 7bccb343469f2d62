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
