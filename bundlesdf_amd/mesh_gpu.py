"""Iso-surface extraction on the MI355X (csrc/nof_mesh.hip), replacing skimage.measure.marching_cubes on the host
(nerf_runner.py:1388-1394):

    marching_cubes_gpu         marching cubes through nof_mc_* with the case table bundlesdf_amd/mesh.py derives -- the default;
                               tests/test_gpu_mesh.py: the same vertices and the same triangles as oracle/marching_cubes.py
    marching_tetrahedra_gpu    marching tetrahedra through nof_mt_* (cfg mesh_extractor: 'tetrahedra'): same algorithm, keys and
                               orientation rule as bundlesdf_amd/mesh.py:marching_tetrahedra

torch supplies the two device-side primitives between the launches: the exclusive scan of the per-cell triangle counts and the
sort/unique that welds the edge vertices.
"""
import ctypes as C

import numpy as np
import torch

from . import lib


def marching_tetrahedra_gpu(vol, iso=0.0):
    """vol [nx,ny,nz] float32 CUDA tensor -> (vertices [V,3] float64 numpy, index coordinates; faces [T,3] int64 numpy).
    Raises ValueError when the level set is empty (like skimage / mesh.py)."""
    assert vol.is_cuda and vol.dtype == torch.float32 and vol.dim() == 3
    vol = vol.contiguous()
    nx, ny, nz = vol.shape
    ncell = (nx - 1) * (ny - 1) * (nz - 1)
    iso32 = C.c_float(float(np.float32(iso)))
    counts = torch.empty(ncell, dtype=torch.int32, device=vol.device)
    lib.call('nof_mt_count', vol, nx, ny, nz, iso32, counts)
    incl = torch.cumsum(counts, 0, dtype=torch.int64)
    T = int(incl[-1].item()) if ncell > 0 else 0
    if T == 0:
        raise ValueError('Surface level must be within volume data range.')
    offsets = (incl - counts).contiguous()
    keys = torch.empty(T, 3, dtype=torch.int64, device=vol.device)
    lib.call('nof_mt_emit', vol, nx, ny, nz, iso32, offsets, keys)
    del offsets, incl, counts
    uniq, inv = torch.unique(keys.view(-1), sorted=True, return_inverse=True)
    faces = inv.view(-1, 3)
    verts = torch.empty(uniq.numel(), 3, dtype=torch.float64, device=vol.device)
    lib.call('nof_mt_vertices', vol, nx, ny, nz, iso32, uniq.contiguous(), int(uniq.numel()), verts)
    ok = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])
    return verts.cpu().numpy(), faces[ok].cpu().numpy()


_MC_TABLE = {}


def marching_cubes_gpu(vol, iso=0.0):
    """vol [nx,ny,nz] float32 CUDA tensor -> (vertices [V,3] float64 numpy in index coordinates, sorted by edge key; faces [T,3]
    int64 numpy, cell by cell, normals from value < iso to value >= iso).  Raises ValueError when the level set is empty."""
    assert vol.is_cuda and vol.dtype == torch.float32 and vol.dim() == 3
    from .mesh import mc_case_table
    vol = vol.contiguous()
    if vol.device not in _MC_TABLE:
        _MC_TABLE[vol.device] = torch.from_numpy(mc_case_table()).to(vol.device).contiguous()
    table = _MC_TABLE[vol.device]
    nx, ny, nz = vol.shape
    ncell = (nx - 1) * (ny - 1) * (nz - 1)
    iso32 = C.c_float(float(np.float32(iso)))
    counts = torch.empty(ncell, dtype=torch.int32, device=vol.device)
    lib.call('nof_mc_count', vol, nx, ny, nz, iso32, table, counts)
    incl = torch.cumsum(counts, 0, dtype=torch.int64)
    T = int(incl[-1].item()) if ncell > 0 else 0
    if T == 0:
        raise ValueError('Surface level must be within volume data range.')
    offsets = (incl - counts).contiguous()
    keys = torch.empty(T, 3, dtype=torch.int64, device=vol.device)
    lib.call('nof_mc_emit', vol, nx, ny, nz, iso32, table, offsets, keys)
    del offsets, incl, counts
    uniq, inv = torch.unique(keys.view(-1), sorted=True, return_inverse=True)
    faces = inv.view(-1, 3)
    verts = torch.empty(uniq.numel(), 3, dtype=torch.float64, device=vol.device)
    lib.call('nof_mt_vertices', vol, nx, ny, nz, iso32, uniq.contiguous(), int(uniq.numel()), verts)
    ok = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 0] != faces[:, 2])
    return verts.cpu().numpy(), faces[ok].cpu().numpy()
