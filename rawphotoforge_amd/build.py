"""Build rawphotoforge_amd/librpf_hip.so (gfx950 kernels + the extern "C" boundary) in-tree.

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the parity contract: the
oracle rounds after every binary32 operation (oracle/rpf_oracle.c), so must the kernels and
the host-side f32 arithmetic (PCHIP, parameter conversion).  -fno-slp-vectorize: on gfx950 a v_pk_*_f32
costs two plain f32 instructions' issue time (tools/microbench.hip), so SLP packing only adds the v_mov
traffic that builds the register pairs.  The rolling-window kernels (the fused chain) are their own translation unit,
compiled with -mllvm -amdgpu-sched-strategy=max-ilp: same instructions, 93 instead of 78 VGPRs, fewer waits, -2 %
(profiles/r02_experiments.md); instruction scheduling does not touch the arithmetic, and the parity tests run on this build.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librpf_hip.so")
# (source, object, extra flags).  rpf_kernels.hip is built as two translation units (see RPF_TU in the file): the
# rolling-window kernels under the scheduler's max-ILP strategy, the rest under the default one.
UNITS = [("rpf_kernels.hip", "rpf_kernels_stream.o", ["-DRPF_TU=1", "-mllvm", "-amdgpu-sched-strategy=max-ilp"]),
         ("rpf_kernels.hip", "rpf_kernels_rest.o", ["-DRPF_TU=2"]),
         ("rpf_api.cpp", "rpf_api.o", []),
         ("rpf_shard.cpp", "rpf_shard.o", []),
         ("rpf_tables.cpp", "rpf_tables.o", [])]
SOURCES = sorted({u[0] for u in UNITS})
HEADERS = ["rpf_math.h", "rpf_chain.h", "rpf_host.h", "rpf_launch.h", "rpf_internal.h", os.path.join("..", "..", "include", "rpf.h")]
ARCH = "gfx950"


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    for src, objname, extra in UNITS:
        obj = os.path.join(CSRC, objname)
        cmd = [hipcc, "-x", "hip", f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
               "-Wall", "-Wno-unused-function"] + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.run(cmd, check=True)
        objs.append(obj)
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl", "-pthread"]
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
