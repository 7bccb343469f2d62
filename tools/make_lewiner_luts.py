"""Reads Lewiner's marching-cubes lookup tables (LookUpTable.h of the companion code of Lewiner, Lopes, Vieira, Tavares, "Efficient
implementation of Marching Cubes' cases with topological guarantees", JGT 8(2), 2003) out of scikit-image's copy of them and writes them
as plain int8 arrays:   bundlesdf_amd/lewiner_luts.npz (the product: packed for the device by bundlesdf_amd/mesh.py) and
                        oracle/lewiner_luts.npz       (the oracle's own, byte-identical copy).
scikit-image is the library the reference calls (skimage.measure.marching_cubes, nerf_runner.py:1388-1394); it is not importable by
this repository's interpreter but by the build container's Anaconda one -- run with THAT interpreter:

    /opt/conda/bin/python3.9 tools/make_lewiner_luts.py

(scikit-image 0.18.3, skimage/measure/_marching_cubes_lewiner_luts.py: "auto-generated from mc_meta/LookUpTable.h".)  The classic
table CASESCLASSIC is left out: nothing here uses it."""
import os
import warnings

import numpy as np

warnings.filterwarnings('ignore')
from skimage.measure import _marching_cubes_lewiner_luts as L          # noqa: E402
from skimage.measure._marching_cubes_lewiner import _to_array          # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = {}
for name in sorted(dir(L)):
    v = getattr(L, name)
    if name.isupper() and name != 'CASESCLASSIC' and isinstance(v, tuple) and len(v) == 2:
        out[name] = np.ascontiguousarray(_to_array(v)).astype(np.int8)
for rel in ('bundlesdf_amd/lewiner_luts.npz', 'oracle/lewiner_luts.npz'):
    np.savez_compressed(os.path.join(ROOT, rel), **out)
print(len(out), 'tables,', sum(v.size for v in out.values()), 'bytes')
