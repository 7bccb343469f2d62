"""Where the device Lewiner extraction of a 512^3 volume spends its time, stage by stage (events on the current stream):
   python tools/mc_stage_probe.py [n]     (an analytic ellipsoid's SDF, masked to 1.0 away from the surface like the octree query does)"""
import ctypes as C, sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bundlesdf_amd import lib, mesh_gpu
from bundlesdf_amd.mesh import lewiner_lut_pack

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
lib.load()
ax = torch.linspace(-1, 1, n, device='cuda')
X, Y, Z = torch.meshgrid(ax, ax, ax, indexing='ij')
vol = (torch.sqrt((X / 0.5) ** 2 + (Y / 0.35) ** 2 + (Z / 0.42) ** 2) - 1.0) * 0.35
vol = torch.where(vol.abs() < 0.08, vol, torch.ones_like(vol)).contiguous()
del X, Y, Z
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    v, f = mesh_gpu.marching_cubes_lewiner_gpu(vol, 0.0)
    t1 = time.perf_counter()
    print(f'whole call incl. D2H: {(t1 - t0) * 1e3:.2f} ms, {len(v)} vertices, {len(f)} triangles')
if hasattr(mesh_gpu, 'STAGE_MS'):
    print(mesh_gpu.STAGE_MS)
# stage by stage, the call's own sequence
packed, offs = lewiner_lut_pack()
o = lib.NofMclLuts()
for t, val in enumerate(offs):
    o.off[t] = int(val)
luts = torch.from_numpy(packed).cuda().contiguous()
nx = ny = nz = n
ncell = (n - 1) ** 3
iso32 = C.c_float(0.0)
def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e
for rep in range(2):
    marks = [('start', ev())]
    counts = torch.empty(ncell, dtype=torch.int32, device='cuda')
    lib.call('nof_mcl_count', vol, nx, ny, nz, iso32, luts, C.byref(o), counts); marks.append(('count', ev()))
    incl = torch.cumsum(counts, 0, dtype=torch.int64); marks.append(('cumsum', ev()))
    T = int(incl[-1].item()); marks.append(('item (sync)', ev()))
    offsets = (incl - counts).contiguous(); marks.append(('offsets', ev()))
    keys = torch.empty(T, 3, dtype=torch.int64, device='cuda')
    lib.call('nof_mcl_emit', vol, nx, ny, nz, iso32, luts, C.byref(o), offsets, keys); marks.append(('emit', ev()))
    uniq, inv = torch.unique(keys.view(-1), sorted=True, return_inverse=True); marks.append(('unique', ev()))
    verts = torch.empty(uniq.numel(), 3, dtype=torch.float64, device='cuda')
    lib.call('nof_mcl_vertices', vol, nx, ny, nz, iso32, uniq.contiguous(), int(uniq.numel()), verts); marks.append(('vertices', ev()))
    vh, fh = verts.cpu().numpy(), inv.view(-1, 3).cpu().numpy(); marks.append(('D2H', ev()))
    torch.cuda.synchronize()
    print(' | '.join(f'{b[0]} {a[1].elapsed_time(b[1]):.3f}' for a, b in zip(marks[:-1], marks[1:])), '| total', f'{marks[0][1].elapsed_time(marks[-1][1]):.3f} ms')
