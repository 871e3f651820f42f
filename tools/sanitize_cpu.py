"""Driver of tools/sanitize_cpu.sh: runs the oracle, the CPU twin of the device arithmetic and the host state
machine (built with -fsanitize=address,undefined) over the shapes / parameter combinations the tests use."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
D = os.environ["RPF_ASAN_DIR"]
import oracle as O  # noqa: E402

O.build = lambda force=False: os.path.join(D, "librpf_oracle_asan.so")
from tests import util  # noqa: E402

util.build_twin = lambda force=False: os.path.join(D, "librpf_twin_asan.so")
rng = np.random.default_rng(0)
for _ in range(6):
    ps = [util.random_params(rng) for _ in range(int(rng.integers(1, 4)))]
    rgb = rng.random((5000, 3), dtype=np.float32) * np.float32(1.3) - np.float32(0.1)
    img = np.concatenate([rgb, np.ones((5000, 1), np.float32)], 1).reshape(1, 5000, 4)
    assert np.array_equal(util.twin_apply(rgb, ps).view(np.uint32), O.apply(img, ps).reshape(5000, 4).view(np.uint32))
for shp in ((3, 3), (5, 8), (33, 47)):
    for pat in range(4):
        O.develop(util.bayer_uniform(*shp), O.RawParams(cfa_pattern=pat), [util.params_S()])
try:
    O.pchip_interpolate([0, 5, 5], [0, 1, 2], [1])
except O.PchipError:
    pass
for bits in (10, 12, 14, 16):
    a = rng.integers(0, 1 << bits, (3, 37), dtype=np.uint16)
    assert np.array_equal(O.unpack_bayer(O.pack_bayer(a, bits), 37, bits), a)
O.resize_long_edge(rng.random((33, 5, 3), dtype=np.float32), 7)
O.legacy_tone_lut(0.5, 20, 30, -20, 5, -5)
O.legacy_apply_gamma(rng.random(1000, dtype=np.float32), (2.222, 4.5))
assert util.twin().twin_state_exercise() == 0
print("sanitizer run complete: no ASan / UBSan report")
