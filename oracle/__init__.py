"""CPU oracle for the Neural Object Field hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (PyTorch fp32 + NumPy) of the
reference algorithm on the path named by BASELINE.json:north_star.  Every function
cites the reference file:line it follows.

Rules (enforced by tests/test_capi.py::test_product_never_imports_the_oracle):
  * only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
    may import anything from here -- and there only as the checker, never as the thing
    being measured or shipped;
  * nothing under ``bundlesdf_amd/`` imports it; the product fails loudly when the HIP
    extension is missing.

Pinning status (see DESIGN.md "Oracle"):
  * NeRFSmall, SHEncoder, raw2outputs, get_sdf_loss/get_masks, sample_rays_uniform,
    ray_box_intersection_batch, get_camera_rays_np, get_truncation and the train_loop
    loss assembly are PINNED against outputs of the reference's own pure-PyTorch code
    executed on CPU (tests/golden/make_golden.py -> tests/golden/*.npz).
  * the multires hash encoder (forward, dy_dx, table-gradient scatter, input gradient:
    gridencoder.cu), the occupied-voxel sampler walk and the ray-trace post-process
    (common.cu) are PINNED against the reference's own kernels compiled as host C++
    (oracle/ref_build.py -> oracle/_ref/libnof_ref.so, tests/test_ref_native.py), and the
    committed train_loop fixture was generated with those compiled kernels under the
    reference's own grid.py / OctreeManager.ray_trace.
  * skimage.measure.marching_cubes (default method 'lewiner'; nerf_runner.py:1388-1394) is third-party code absent
    from the reference tree but importable by the build container's Anaconda interpreter (scikit-image 0.18.3):
    oracle/marching_cubes_lewiner.py restates Lewiner et al. 2003 and is PINNED on scikit-image's own outputs
    (tests/golden/make_mc_golden.py -> tests/golden/mc_skimage_vectors.npz, tests/test_mesh.py).
  * kaolin's octree ray tracer (unbatched_raytrace) and pytorch3d's se3_exp_map are
    third-party code absent from the reference tree AND from this container: for those two the oracle is a
    restatement of the published algorithm and parity is UNPINNED.
"""
