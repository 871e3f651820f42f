#!/usr/bin/env python3
"""Instruction census PER STAGE of a fused kernel's inner loop, by differencing: the device assembly is generated for the
shipped source and for copies with ONE stage replaced by a pass-through, and tools/isa_census.py counts each inner loop;
a stage's share is what disappears with it (stages interact a little through scheduling and register allocation, so the
shares do not add up exactly).  Nothing here runs on a GPU.
    python tools/stage_census.py '<kernel name substring>' [...]  >  profiles/r03_isa_census.md (part)"""
import os, shutil, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_census as IC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "rawphotoforge_amd", "csrc")
STAGES = [
    ("(nothing removed)", []),
    ("gamma: 3 correctly rounded pow(c, 1/2.4) + select + clamps", [
        ("rpf_chain.h", "clamp01(RPF_F(1.055) * powc<POW_GAMMA>(c, T.gamma) - chain_k_gamma_off(T))", "clamp01(RPF_F(1.055) * c - chain_k_gamma_off(T))")]),
    ("cube roots: 3 correctly rounded pow(x, 1/3)", [
        ("rpf_chain.h", "const float lc = powc<POW_CBRT>(l, T.cbrt);", "const float lc = l;"),
        ("rpf_chain.h", "const float mc = powc<POW_CBRT>(mm, T.cbrt);", "const float mc = mm;"),
        ("rpf_chain.h", "const float sc = powc<POW_CBRT>(s, T.cbrt);", "const float sc = s;")]),
    ("hue index + cos / sin (candidate polynomial, table gather, binary64 cross product)", [
        ("rpf_chain.h", "hu0[n] = hue_index_cs<false>(Bq, A, luts.hue_bounds_cs(), T.sc, cs0[n], sn0[n]);", "hu0[n] = 0u; cs0[n] = A; sn0[n] = Bq;"),
        ("rpf_chain.h", "hu0[n] = hue_index<false>(Bq, A, luts.hue_bounds());", "hu0[n] = (uint32_t)A;")]),
    ("chroma: correctly rounded sqrt(a^2 + b^2)", [
        ("rpf_chain.h", "C[n] = sqrt_cr(A * A + Bq * Bq);", "C[n] = A * A + Bq * Bq;")]),
    ("vignette (pow(t, 1.5), sqrt, division)", [
        ("rpf_chain.h", "    const float dist = sqrt_cr(cx2 + cy2);", "    return cx2 + cy2 + vign_strength;\n    const float dist = sqrt_cr(cx2 + cy2);")]),
]


def asm_for(patches):
    d = tempfile.mkdtemp(prefix="census_")
    dst = os.path.join(d, "rawphotoforge_amd", "csrc")
    os.makedirs(os.path.dirname(dst))
    shutil.copytree(SRC, dst, ignore=shutil.ignore_patterns("*.o"))
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(d, "include"))
    for f, old, new in patches:
        p = os.path.join(dst, f)
        s = open(p).read()
        if old not in s:
            raise SystemExit(f"patch text not found in {f}: {old[:50]!r}")
        open(p, "w").write(s.replace(old, new))
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "-x", "hip", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                    "-fno-slp-vectorize", "-DRPF_TU=1", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "--cuda-device-only", "-S",
                    os.path.join(dst, "rpf_kernels.hip"), "-o", out], check=True, stderr=subprocess.DEVNULL)
    return out, d


def main():
    kernels = sys.argv[1:]
    rows = {k: [] for k in kernels}
    for name, patches in STAGES:
        path, d = asm_for(patches)
        for k in kernels:
            cls, _ = IC.census(path, k)
            valu = {c: v for c, v in cls.items() if c in IC.COST}
            full = valu.get("full", 0)
            f64 = valu.get("f64", 0)
            quarter = valu.get("quarter", 0)
            half = sum(v for c, v in valu.items() if c not in ("full", "f64", "quarter"))
            ns = sum(IC.COST[c] * v for c, v in valu.items())
            rows[k].append((name, full, half, f64, quarter, cls.get("lds", 0), cls.get("vmem", 0), ns))
        shutil.rmtree(d, ignore_errors=True)
    for k in kernels:
        base = rows[k][0]
        print(f"### `{k}` — per stage, one inner-loop trip\n")
        print("| stage removed | full-rate | half-rate | binary64 | quarter-rate | LDS | VMEM | ns at issue cost | share |")
        print("|---|---|---|---|---|---|---|---|---|")
        print(f"| whole loop | {base[1]} | {base[2]} | {base[3]} | {base[4]} | {base[5]} | {base[6]} | {base[7]:.0f} | 100 % |")
        rest = list(base[1:])
        for r in rows[k][1:]:
            d = [b - x for b, x in zip(base[1:], r[1:])]
            rest = [a - b for a, b in zip(rest, d)]
            print(f"| {r[0]} | {d[0]} | {d[1]} | {d[2]} | {d[3]} | {d[4]} | {d[5]} | {d[6]:.0f} | {100 * d[6] / base[7]:.0f} % |")
        print(f"| everything else (demosaic, normalisation, taps, camera matrix, brightness / tone stage, 3x3 matrices, stores) | "
              f"{rest[0]} | {rest[1]} | {rest[2]} | {rest[3]} | {rest[4]} | {rest[5]} | {rest[6]:.0f} | {100 * rest[6] / base[7]:.0f} % |\n")


if __name__ == "__main__":
    main()
