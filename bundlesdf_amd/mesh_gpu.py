"""Iso-surface extraction on the MI355X (csrc/nof_mesh.hip), replacing skimage.measure.marching_cubes on the host
(nerf_runner.py:1388-1394):

    marching_cubes_lewiner_gpu marching cubes with Lewiner's disambiguation through nof_mcl_* = skimage's default method, the call the
                               reference makes -- the default (cfg mesh_extractor: 'lewiner'); tests/test_gpu_mesh.py: triangle for
                               triangle skimage 0.18.3's output on the committed fixture (tests/golden/mc_skimage_vectors.npz)
    marching_cubes_gpu         classic marching cubes through nof_mc_* with the case table bundlesdf_amd/mesh.py derives
                               (cfg mesh_extractor: 'cubes'); the same vertices and triangles as oracle/marching_cubes.py
    marching_tetrahedra_gpu    marching tetrahedra through nof_mt_* (cfg mesh_extractor: 'tetrahedra'): same algorithm, keys and
                               orientation rule as bundlesdf_amd/mesh.py:marching_tetrahedra

torch supplies the two device-side primitives between the launches: the exclusive scan of the per-cell triangle counts and the
sort/unique that welds the edge vertices.
"""
import ctypes as C

import numpy as np
import torch

from . import lib


def _to_host(*tensors):
    """device tensors -> NumPy arrays through PINNED staging buffers, one synchronisation for all of them.  `tensor.cpu()` goes through
    pageable memory: 3.4-25 ms for a 512^3 mesh (27 MB) against 0.55 ms this way (tools/pin_probe.py, profiles/r06_s_pin_probe.txt);
    torch's caching host allocator hands the same pinned blocks out again once the arrays of an earlier call are gone (each array
    keeps its buffer alive: `Tensor.numpy()` shares memory), so only a process's first extraction pays for the page locking."""
    out = []
    for t in tensors:
        h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        h.copy_(t, non_blocking=True)
        out.append(h)
    torch.cuda.current_stream().synchronize()
    return tuple(h.numpy() for h in out)


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
    return _to_host(verts, faces[ok])


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
    return _to_host(verts, faces[ok])


_MCL_LUTS = {}


def marching_cubes_lewiner_gpu(vol, iso=0.0):
    """skimage.measure.marching_cubes(vol, iso) -- method 'lewiner', gradient_direction 'descent', allow_degenerate=True: the
    defaults the reference leaves in place (nerf_runner.py:1388-1394) -- on the device.  vol [nx,ny,nz] float32 CUDA tensor ->
    (vertices [V,3] float64 numpy in index coordinates: the cells' centre vertices first (cell order), then the edge vertices sorted by
    edge key; faces [T,3] int64 numpy, cell by cell).  Raises ValueError when the level set is empty."""
    assert vol.is_cuda and vol.dtype == torch.float32 and vol.dim() == 3
    from .mesh import lewiner_lut_pack
    vol = vol.contiguous()
    if vol.device not in _MCL_LUTS:
        packed, offs = lewiner_lut_pack()
        o = lib.NofMclLuts()
        for t, v in enumerate(offs):
            o.off[t] = int(v)
        _MCL_LUTS[vol.device] = (torch.from_numpy(packed).to(vol.device).contiguous(), o)
    luts, offs = _MCL_LUTS[vol.device]
    nx, ny, nz = vol.shape
    ncell = (nx - 1) * (ny - 1) * (nz - 1)
    iso32 = C.c_float(float(np.float32(iso)))
    # two-level scan (round 6): one count per workgroup of 256 cells, their scan here, placement inside the emit launch -- the
    # per-cell counts / scan / offsets of the three-launch scheme (2.5 GB at 512^3, 2.1 of 4.9 ms) are gone
    nblk = (ncell + 255) // 256
    block_counts = torch.empty(max(nblk, 1), dtype=torch.int32, device=vol.device)
    if ncell > 0:
        lib.call('nof_mcl_count_blocks', vol, nx, ny, nz, iso32, luts, C.byref(offs), block_counts)
    block_end = torch.cumsum(block_counts, 0, dtype=torch.int64)
    T = int(block_end[-1].item()) if ncell > 0 else 0
    if T == 0:
        raise ValueError('Surface level must be within volume data range.')
    keys = torch.empty(T, 3, dtype=torch.int64, device=vol.device)
    lib.call('nof_mcl_emit_blocks', vol, nx, ny, nz, iso32, luts, C.byref(offs), block_end, keys)
    del block_end, block_counts
    uniq, inv = torch.unique(keys.view(-1), sorted=True, return_inverse=True)
    faces = inv.view(-1, 3)
    verts = torch.empty(uniq.numel(), 3, dtype=torch.float64, device=vol.device)
    lib.call('nof_mcl_vertices', vol, nx, ny, nz, iso32, uniq.contiguous(), int(uniq.numel()), verts)
    return _to_host(verts, faces)
