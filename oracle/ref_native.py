"""ctypes view of oracle/_ref/libnof_ref.so: the reference's own native arithmetic compiled as host C++
(recipe: oracle/ref_build.py).  TEST INFRASTRUCTURE -- see oracle/__init__.py for who may import this.

    grid_encode_forward / grid_encode_backward      gridencoder.cu:447,472 (float32, float16 or float64 tables)
    sample_rays_uniform_occupied_voxels             common.cu:107
    postprocess_octree_ray_tracing                  common.cu:151
    ray_color_to_texture_image                      common.cu:223
    level_constants                                 gridencoder.cu:155-156 evaluated with the host's exp2f
"""
import ctypes as C
import numpy as np

from . import ref_build

_lib = None
_DT = {np.dtype(np.float32): 0, np.dtype(np.float16): 1, np.dtype(np.float64): 2}


def load(build=True):
    """The library, built on demand where /root/reference exists; None when neither it nor a prebuilt .so is there."""
    global _lib
    if _lib is None:
        path = ref_build.build() if build else (ref_build.LIB if ref_build.available() else None)
        if path is None:
            return None
        _lib = C.CDLL(path)
        _lib.ref_grid_last_error.restype = C.c_char_p
        _lib.ref_common_last_error.restype = C.c_char_p
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _chk(rc, which):
    if rc < 0:
        raise RuntimeError(getattr(_lib, f'ref_{which}_last_error')().decode())
    return rc


def grid_encode_forward(inputs01, embeddings, offsets, L, S, H, calc_grad_inputs=True, gridtype=0, align_corners=False):
    """inputs01 [B,3] float32 in [0,1]; embeddings [N,C] f32/f16/f64; offsets [L+1] int32.
    Returns (outputs [L,B,C], dy_dx [B, L*D*C] or None) in the embeddings' dtype -- layouts of gridencoder.cu:384-388."""
    lib = load()
    x = np.ascontiguousarray(inputs01, np.float32)
    emb = np.ascontiguousarray(embeddings)
    off = np.ascontiguousarray(offsets, np.int32)
    B, D = x.shape
    Cc = emb.shape[1]
    out = np.zeros((L, B, Cc), emb.dtype)
    dy = np.zeros((B, L * D * Cc), emb.dtype)
    _chk(lib.ref_grid_encode_forward(_p(x), _p(emb), _p(off), _p(out), C.c_uint32(B), C.c_uint32(D), C.c_uint32(Cc),
                                     C.c_uint32(L), C.c_float(S), C.c_uint32(H), int(calc_grad_inputs), _p(dy),
                                     C.c_uint32(gridtype), int(align_corners), _DT[emb.dtype]), 'grid')
    return out, (dy if calc_grad_inputs else None)


def grid_encode_backward(grad, inputs01, embeddings, offsets, L, S, H, dy_dx=None, gridtype=0, align_corners=False):
    """grad [L,B,C]; returns (grad_embeddings [N,C] accumulated from zero, grad_inputs [B,D] or None)."""
    lib = load()
    x = np.ascontiguousarray(inputs01, np.float32)
    emb = np.ascontiguousarray(embeddings)
    g = np.ascontiguousarray(grad, emb.dtype)
    off = np.ascontiguousarray(offsets, np.int32)
    B, D = x.shape
    Cc = emb.shape[1]
    ge = np.zeros_like(emb)
    calc = dy_dx is not None
    dy = np.ascontiguousarray(dy_dx, emb.dtype) if calc else np.zeros((1,), emb.dtype)
    gi = np.zeros((B, D), emb.dtype)
    _chk(lib.ref_grid_encode_backward(_p(g), _p(x), _p(emb), _p(off), _p(ge), C.c_uint32(B), C.c_uint32(D), C.c_uint32(Cc),
                                      C.c_uint32(L), C.c_float(S), C.c_uint32(H), int(calc), _p(dy), _p(gi),
                                      C.c_uint32(gridtype), int(align_corners), _DT[emb.dtype]), 'grid')
    return ge, (gi if calc else None)


def level_constants(L, S, H):
    lib = load()
    scale = np.zeros(L, np.float32)
    res = np.zeros(L, np.uint32)
    lib.ref_grid_level_constants(C.c_uint32(L), C.c_float(S), C.c_uint32(H), _p(scale), _p(res))
    return scale, res


def half_roundtrip(x):
    lib = load()
    x = np.ascontiguousarray(x, np.float32)
    bits = np.zeros(x.shape, np.uint16)
    back = np.zeros(x.shape, np.float32)
    lib.ref_half_roundtrip(_p(x), _p(bits), _p(back), C.c_int64(x.size))
    return bits, back


def sample_rays_uniform_occupied_voxels(z_in_out, z_sampled, z_vals=None):
    """Returns (z_vals, spun): spun is True when the device code would have hung (common.cu:66-72,87-93)."""
    lib = load()
    # the kernel's own error print reads box [max_n_box] (one past the end, common.cu:69): give it a padded buffer
    a = np.concatenate([np.asarray(z_in_out, np.float32).ravel(), np.zeros(8, np.float32)])[:np.asarray(z_in_out).size].reshape(np.asarray(z_in_out).shape)
    b = np.ascontiguousarray(z_sampled, np.float32)
    out = np.zeros_like(b) if z_vals is None else np.ascontiguousarray(z_vals, np.float32).copy()
    R, H, _ = a.shape
    rc = _chk(lib.ref_sample_rays_uniform_occupied_voxels(_p(a), _p(b), _p(out), C.c_int64(R), C.c_int64(H),
                                                          C.c_int64(b.shape[1])), 'common')
    return out, rc == 1


def postprocess_octree_ray_tracing(ray_index, depth_in_out, unique_ids, start_poss, max_intersections, N_rays):
    lib = load()
    ri = np.ascontiguousarray(ray_index, np.int64)
    d = np.ascontiguousarray(depth_in_out, np.float32)
    u = np.ascontiguousarray(unique_ids, np.int64)
    s = np.ascontiguousarray(start_poss, np.int64)
    out = np.zeros((N_rays, max_intersections, 2), np.float32)
    _chk(lib.ref_postprocess_octree_ray_tracing(_p(ri), _p(d), _p(u), _p(s), C.c_int64(ri.shape[0]), C.c_int64(u.shape[0]),
                                                int(max_intersections), int(N_rays), _p(out)), 'common')
    return out


def ray_color_to_texture_image(F, V, hit_locations, hit_face_ids, uvs_tex):
    lib = load()
    F = np.ascontiguousarray(F, np.int64)
    V = np.ascontiguousarray(V, np.float32)
    h = np.ascontiguousarray(hit_locations, np.float32)
    i = np.ascontiguousarray(hit_face_ids, np.int64)
    t = np.ascontiguousarray(uvs_tex, np.float32)
    uvs = np.zeros((h.shape[0], 2), np.float32)
    _chk(lib.ref_ray_color_to_texture_image(_p(F), _p(V), _p(h), _p(i), _p(t), _p(uvs), C.c_int64(F.shape[0]),
                                            C.c_int64(V.shape[0]), C.c_int64(h.shape[0])), 'common')
    return uvs


# ----------------------------------------------------------------------------------------------------------------------
# Stand-ins for the reference's two pybind modules (`gridencoder`, `common`) over CPU torch tensors, same signatures as
# mycuda/torch_ngp_grid_encoder/bindings.cpp:16-19 and mycuda/bindings.cpp:15-19, executing the reference's own code.
# Used by tests/golden/make_golden.py to run the reference's grid.py / nerf_runner.py / Utils.py on CPU.
class _Gridencoder:
    @staticmethod
    def _dt(t):
        import torch
        return {torch.float32: 0, torch.float16: 1, torch.float64: 2}[t.dtype]

    def grid_encode_forward(self, inputs, embeddings, offsets, outputs, B, D, Cc, L, S, H, calc_grad_inputs, dy_dx,
                            gridtype, align_corners):
        lib = load()
        for t in (inputs, embeddings, offsets, outputs, dy_dx):
            assert t.is_contiguous() and t.device.type == 'cpu'
        assert offsets.dtype.is_floating_point is False and offsets.element_size() == 4
        _chk(lib.ref_grid_encode_forward(C.c_void_p(inputs.data_ptr()), C.c_void_p(embeddings.data_ptr()),
                                         C.c_void_p(offsets.data_ptr()), C.c_void_p(outputs.data_ptr()), C.c_uint32(B),
                                         C.c_uint32(D), C.c_uint32(Cc), C.c_uint32(L), C.c_float(S), C.c_uint32(H),
                                         int(bool(calc_grad_inputs)), C.c_void_p(dy_dx.data_ptr()), C.c_uint32(gridtype),
                                         int(bool(align_corners)), self._dt(embeddings)), 'grid')

    def grid_encode_backward(self, grad, inputs, embeddings, offsets, grad_embeddings, B, D, Cc, L, S, H, calc_grad_inputs,
                             dy_dx, grad_inputs, gridtype, align_corners):
        lib = load()
        for t in (grad, inputs, embeddings, offsets, grad_embeddings, dy_dx, grad_inputs):
            assert t.is_contiguous() and t.device.type == 'cpu'
        assert grad.dtype == embeddings.dtype
        _chk(lib.ref_grid_encode_backward(C.c_void_p(grad.data_ptr()), C.c_void_p(inputs.data_ptr()),
                                          C.c_void_p(embeddings.data_ptr()), C.c_void_p(offsets.data_ptr()),
                                          C.c_void_p(grad_embeddings.data_ptr()), C.c_uint32(B), C.c_uint32(D),
                                          C.c_uint32(Cc), C.c_uint32(L), C.c_float(S), C.c_uint32(H),
                                          int(bool(calc_grad_inputs)), C.c_void_p(dy_dx.data_ptr()),
                                          C.c_void_p(grad_inputs.data_ptr()), C.c_uint32(gridtype), int(bool(align_corners)),
                                          self._dt(embeddings)), 'grid')


class _Common:
    def sampleRaysUniformOccupiedVoxels(self, z_in_out, z_sampled, z_vals):
        import torch
        out, spun = sample_rays_uniform_occupied_voxels(z_in_out.detach().numpy(), z_sampled.detach().numpy(),
                                                        z_vals.detach().numpy())
        if spun:
            raise RuntimeError('the reference kernel would hang here (common.cu:66-72,87-93)')
        z_vals.copy_(torch.from_numpy(out))
        return z_vals

    def postprocessOctreeRayTracing(self, ray_index, depth_in_out, unique_ids, start_poss, max_intersections, N_rays):
        import torch
        return torch.from_numpy(postprocess_octree_ray_tracing(ray_index.numpy(), depth_in_out.numpy(), unique_ids.numpy(),
                                                               start_poss.numpy(), max_intersections, N_rays))


def gridencoder_module():
    return _Gridencoder()


def common_module():
    return _Common()
