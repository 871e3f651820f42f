#!/usr/bin/env python3
"""Write tests/golden/chain_golden.npz: oracle outputs on the seeded 64x64 crops of SURVEY.md §8(d)
((U) and (N) frames under parameter sets D and S) plus SHA-256 digests of the 256x256 cases.

The reference has no golden vectors of its own and cannot run here, so these pin the
ORACLE (oracle/rpf_oracle.c) against regressions and give the GPU tests committed expected outputs.
"""
import hashlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import oracle as O  # noqa: E402
from tests import util  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "chain_golden.npz")


def main():
    out = {}
    rp = util.raw_params()
    for tag, gen in (("U", util.bayer_uniform), ("N", util.bayer_natural)):
        for pname, pf in (("D", util.params_D), ("S", util.params_S)):
            cfa = gen(64, 64)
            out[f"cfa64_{tag}"] = cfa
            out[f"develop64_{tag}_{pname}"] = O.develop(cfa, rp, [pf()])
            big = O.develop(gen(256, 256), rp, [pf()])
            out[f"sha256_develop256_{tag}_{pname}"] = np.frombuffer(hashlib.sha256(big.tobytes()).digest(), np.uint8)
        out[f"demosaic64_{tag}"] = O.demosaic(gen(64, 64), rp)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
