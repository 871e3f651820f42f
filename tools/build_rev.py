#!/usr/bin/env python3
"""Build librpf_hip.so of a git revision into tmp_libs/librpf_<name>.so (git-ignored; travels to the GPU box) for same-box A/B
against the working tree (tools/ab_bench.sh):   python tools/build_rev.py <rev> <name>"""
import os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rev, name = sys.argv[1], sys.argv[2]
top = os.path.join("/tmp", "rev_" + name)
shutil.rmtree(top, ignore_errors=True)
os.makedirs(top)
subprocess.run(f"git -C {ROOT} archive {rev} rawphotoforge_amd/csrc rawphotoforge_amd/build.py include | tar -x -C {top}", shell=True, check=True)
sys.path.insert(0, os.path.join(top, "rawphotoforge_amd"))
import build as b   # the revision's own build recipe
dst = os.path.join(top, "rawphotoforge_amd", "csrc")
objs = []
for srcf, objname, extra in b.UNITS:
    obj = os.path.join(dst, objname)
    subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", f"--offload-arch={b.ARCH}", "-O3", "-std=c++17", "-ffp-contract=off",
                    "-fno-slp-vectorize", "-fPIC", "-Wno-unused-function"] + extra + ["-c", os.path.join(dst, srcf), "-o", obj], check=True)
    objs.append(obj)
os.makedirs(os.path.join(ROOT, "tmp_libs"), exist_ok=True)
out = os.path.join(ROOT, "tmp_libs", f"librpf_{name}.so")
subprocess.run(["/opt/rocm/bin/hipcc", f"--offload-arch={b.ARCH}", "-shared", "-fPIC", "-o", out] + objs + ["-ldl", "-pthread"], check=True)
print(out)
