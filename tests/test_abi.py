"""The C-ABI library loads, exports every symbol include/rpf.h declares, and has no CPU compute path."""
import ctypes as C
import os
import re
import subprocess
import sys

import pytest

from rawphotoforge_amd import _native as N
from tests import util

HEADER = os.path.join(util.ROOT, "include", "rpf.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(rpf_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    names = _declared()
    assert len(names) >= 40
    lib = C.CDLL(N.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/rpf.h but not exported"
        assert n in N.SYMBOLS, f"{n} has no ctypes binding"
    assert sorted(N.SYMBOLS) == names


def test_library_does_not_link_the_oracle():
    out = subprocess.run(["nm", "-D", "--defined-only", N.LIB_PATH], capture_output=True, text=True).stdout
    assert "orc_" not in out and "twin_" not in out
    ldd = subprocess.run(["ldd", N.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in ldd and "quadmath" not in ldd and "libamdhip64" in ldd


def test_product_package_never_imports_the_oracle():
    code = ("import sys; import rawphotoforge_amd as R; R.PhotoEditor.headless(2, 2); "
            "assert 'oracle' not in sys.modules and 'torch' not in sys.modules; print('clean')")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=util.ROOT)
    assert out.stdout.strip() == "clean", out.stderr
    for root, _, files in os.walk(os.path.join(util.ROOT, "rawphotoforge_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip")):
                text = open(os.path.join(root, f)).read()
                for needle in ("import oracle", "from oracle", "librpf_oracle", '#include "../../oracle', "_build/librpf"):
                    assert needle not in text, (f, needle)


def test_missing_extension_fails_loudly(tmp_path, monkeypatch):
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", str(tmp_path / "librpf_hip.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        N.lib()


def test_version_and_last_error():
    assert b"gfx950" in N.lib().rpf_version()
    h = C.c_void_p()
    assert N.lib().rpf_editor_create_headless(2, 2, None) == N.ERR_INVALID_ARGUMENT
    assert "null" in N.last_error()
    del h


def _build_c_host(tmp_path):
    """tests/c_host/host_check.c as strict C99 against include/rpf.h, linked with the in-tree library."""
    from rawphotoforge_amd import build as b
    exe = str(tmp_path / "host_check")
    libdir = os.path.dirname(b.LIB)
    subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(util.ROOT, "include"),
                    os.path.join(util.ROOT, "tests", "c_host", "host_check.c"), "-o", exe, "-L", libdir, "-lrpf_hip",
                    f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_header_is_plain_c_and_a_c_host_drives_the_device_free_part(tmp_path):
    """The boundary as a cgo / JNI / Rust binding sees it: the header compiles as pedantic C99 and a C program exercises
    operator state, PCHIP, the PPM header grammar and the error reporting — none of which needs a GPU."""
    out = subprocess.run([_build_c_host(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0 and "host_check ok" in out.stdout, out.stderr


@pytest.mark.gpu
def test_c_host_develops_a_frame(tmp_path):
    out = subprocess.run([_build_c_host(tmp_path), "gpu"], capture_output=True, text=True)
    assert out.returncode == 0 and "host_check ok" in out.stdout, out.stderr


@pytest.mark.gpu
def test_c_host_runs_two_sharded_step_pipelines(tmp_path):
    """tests/c_host/shard_check.c: rpf_shard_create x 2, rpf_shard_develop_async on two streams, flush, timing — from C, with
    HIP's C API for buffers and streams, checked against rpf_dev_develop on self-exchanged halos."""
    from rawphotoforge_amd import build as b
    exe = str(tmp_path / "shard_check")
    libdir = os.path.dirname(b.LIB)
    subprocess.run(["gcc", "-std=gnu11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(util.ROOT, "include"), "-I", "/opt/rocm/include",
                    os.path.join(util.ROOT, "tests", "c_host", "shard_check.c"), "-o", exe, "-L", libdir, "-lrpf_hip",
                    "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "shard_check ok" in out.stdout, out.stderr + out.stdout
