#!/usr/bin/env python3
"""Timing-only / A-B builds of librpf_hip.so: a copy of rawphotoforge_amd/csrc with textual patches applied, built into
tmp_libs/librpf_<name>.so (git-ignored; it travels to the GPU box) and run with RPF_HIP_LIB=... (tools/ab_bench.sh).
  python tools/exp_build.py <name> 'file::old text::new text' ['file::old::new' ...]
Results of such builds are NOT correct images unless the patch says so; they exist to attribute kernel time."""
import os, shutil, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name, patches = sys.argv[1], sys.argv[2:]
src = os.path.join(ROOT, "rawphotoforge_amd", "csrc")
dst = os.path.join("/tmp", "exp_" + name, "rawphotoforge_amd", "csrc")
shutil.rmtree(os.path.join("/tmp", "exp_" + name), ignore_errors=True)
os.makedirs(os.path.dirname(dst))
shutil.copytree(src, dst, ignore=shutil.ignore_patterns("*.o"))
shutil.copytree(os.path.join(ROOT, "include"), os.path.join("/tmp", "exp_" + name, "include"))
for p in patches:
    f, old, new = p.split("::", 2)
    path = os.path.join(dst, f)
    s = open(path).read()
    if old not in s:
        raise SystemExit(f"patch text not found in {f}: {old[:60]!r}")
    open(path, "w").write(s.replace(old, new))
sys.path.insert(0, ROOT)
from rawphotoforge_amd import build as b
objs = []
for srcf, objname, extra in b.UNITS:
    obj = os.path.join(dst, objname)
    subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", f"--offload-arch={b.ARCH}", "-O3", "-std=c++17", "-ffp-contract=off",
                    "-fno-slp-vectorize", "-fPIC", "-Wno-unused-function"] + extra + ["-c", os.path.join(dst, srcf), "-o", obj], check=True)
    objs.append(obj)
os.makedirs(os.path.join(ROOT, "tmp_libs"), exist_ok=True)
out = os.path.join(ROOT, "tmp_libs", f"librpf_{name}.so")
subprocess.run(["/opt/rocm/bin/hipcc", f"--offload-arch={b.ARCH}", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"], check=True)
print(out)
