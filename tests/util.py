"""Shared builders for the parity tests (seeded synthetic inputs)."""
import ctypes as C

import numpy as np
import torch

from oracle import nof_oracle as O


def dev(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x) if isinstance(x, np.ndarray) else x)
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def make_grids(lib, L=16, T=14, base=16, finest=256):
    g, offsets, n, pls = lib.make_hash_grid(L, 2, base, T, finest)
    geo = O.HashGeometry(n_levels=L, level_dim=2, base_resolution=base, log2_hashmap_size=T, desired_resolution=finest)
    assert geo.n_entries == n
    for l in range(L):      # product and oracle must share bit-identical level constants
        assert np.float32(g.scale[l]) == geo.scale[l] and g.resolution[l] == geo.resolution[l]
        assert g.size[l] == geo.size[l] and g.offset[l] == geo.offsets[l] and bool(g.hashed[l]) == bool(geo.hashed[l])
    return g, geo


def test_points(n, seed=0):
    rng = np.random.default_rng(seed)
    p = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    p[0] = [-1, -1, -1]
    p[1] = [1, 1, 1]
    p[2] = [0, 0, 0]
    p[3] = [1.5, 0, 0]          # out of range
    p[4] = [0.2, -1.0000001, 0.3]
    p[5] = [1, -1, 0.5]
    return p


def random_occ(n, fill, seed):
    rng = np.random.default_rng(seed)
    return rng.random((n, n, n)) < fill


def occ_to_coords(occ):
    return np.argwhere(occ).astype(np.int32)


def random_rays(R, seed, radius=2.5):
    """Origins on a sphere outside the unit cube, directions towards a jittered point inside it."""
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(R, 3))
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * radius
    tgt = rng.uniform(-0.9, 0.9, size=(R, 3))
    d = tgt - o
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    return o.astype(np.float32), d.astype(np.float32)
