#!/usr/bin/env python3
"""Condense one GPU pass (gpurun_out/<tag>/, written by tools/gpu_pmc.sh) into the tracked summaries
under profiles/: kernel-trace stats, PMC counters aggregated per kernel, derived figures, bench lines.

  python tools/summarize_profiles.py <tag> [round-label]

HBM traffic follows MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE come from separate --pmc passes,
in KiB; WRITE_SIZE is exact for 16-B-per-lane streaming stores; FETCH_SIZE is exact for narrow loads
and reports HALF the bytes of wide (16 B/lane) coalesced streams — the summary lists the raw value and
says which correction applies to each kernel's loads.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1]
    label = sys.argv[2] if len(sys.argv) > 2 else tag
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    lines = [f"# rocprofv3 summary — {label}", "",
             "Source: `tools/gpu_pmc.sh` on one MI355X (gfx950): the default `bench.py` command (steps 400, warmup 50, "
             "`--preheat 200` instead of the default 1000; `--no-cpu-baseline --no-secondary`: without the CPU leg and without the `secondary` block, whose launches of "
             "the same kernel on frames from HBM would be averaged into the headline kernel's row) under `rocprofv3 --kernel-trace --stats`, and, separately, "
             "`--kernel-trace --pmc ...` passes of `bench.py --steps 50 --warmup 10 --preheat 60`.", ""]

    lines += ["## Kernel-trace stats (`--kernel-trace --stats`)", "",
              "| run | kernel | calls | avg µs | min µs | max µs |", "|---|---|---|---|---|---|"]
    for d in sorted(glob.glob(os.path.join(src, "prof_*"))):
        for f in glob.glob(os.path.join(d, "*_kernel_stats.csv")):
            shutil.copy(f, os.path.join(dst, f"{label}_{os.path.basename(f)}"))
            for r in csv.DictReader(open(f)):
                lines.append(f"| {os.path.basename(d)[5:]} | `{r['Name']}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | "
                             f"{float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} |")
    lines.append("")

    traffic = {}
    for w in ("develop", "demosaic", "apply"):
        agg = collections.OrderedDict()
        dur = []
        for d in sorted(glob.glob(os.path.join(src, f"pmc_{w}_*"))):
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "rpf::k_" in r["Kernel_Name"] or "_ZN3rpf" in r["Kernel_Name"]:
                        agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "rpf" in r["Kernel_Name"]:
                        dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        if not agg:
            continue
        a = {k: sum(v) / len(v) for k, v in agg.items()}
        lines += [f"## PMC counters, `{w}` kernel (average per launch over {len(next(iter(agg.values())))} launches)", "",
                  "| counter | value |", "|---|---|"]
        lines += [f"| {k} | {v:.6g} |" for k, v in a.items()]
        lines.append("")
        mp = 24.0
        der = []
        if "GRBM_GUI_ACTIVE" in a and dur:
            avg_us = sum(dur) / len(dur)
            der.append(f"kernel duration under PMC passes: {avg_us:.1f} µs; effective clock = GRBM_GUI_ACTIVE/8/duration = "
                       f"{a['GRBM_GUI_ACTIVE'] / 8 / avg_us / 1e3:.2f} GHz")
        if "SQ_INSTS_VALU" in a:
            der.append(f"VALU wave-instructions per pixel-pair wave-iteration: {a['SQ_INSTS_VALU'] / (mp * 1e6 / 128):.0f} "
                       f"(SALU {a.get('SQ_INSTS_SALU', 0) / (mp * 1e6 / 128):.0f}, LDS {a.get('SQ_INSTS_LDS', 0) / (mp * 1e6 / 128):.0f})")
            if dur:
                der.append(f"VALU issue: {a['SQ_INSTS_VALU'] / 1024 / (sum(dur) / len(dur)) / 1e3:.3f} wave-instr/ns/SIMD "
                           f"= {(sum(dur) / len(dur)) * 1e3 * 1024 / a['SQ_INSTS_VALU']:.2f} ns per VALU instruction per SIMD "
                           "(tools/microbench: f32 add/mul/fma 1.1 ns, f64 / cvt / cndmask / shift-add 1.8 ns, rcp/sqrt 3.4 ns)")
        if "SQ_WAVE_CYCLES" in a and "SQ_WAIT_ANY" in a:
            der.append(f"wave time split: ACTIVE_INST_ANY {a.get('SQ_ACTIVE_INST_ANY', 0) / a['SQ_WAVE_CYCLES']:.0%}, "
                       f"WAIT_INST_ANY {a.get('SQ_WAIT_INST_ANY', 0) / a['SQ_WAVE_CYCLES']:.0%}, "
                       f"WAIT_ANY {a['SQ_WAIT_ANY'] / a['SQ_WAVE_CYCLES']:.0%}")
        if "SQ_LDS_BANK_CONFLICT" in a and a.get("SQ_LDS_IDX_ACTIVE"):
            der.append(f"LDS bank-conflict cycles / LDS active cycles: {a['SQ_LDS_BANK_CONFLICT'] / a['SQ_LDS_IDX_ACTIVE']:.0%}")
        if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
            fetch, write = a["FETCH_SIZE"] * 1024, a["WRITE_SIZE"] * 1024
            alg = 18 * mp * 1e6 if w != "apply" else 32 * mp * 1e6
            der.append(f"HBM traffic per launch: FETCH_SIZE {fetch / 1e6:.1f} MB (raw; this kernel's loads are 2 B/lane, "
                       f"not the 16-B/lane streams whose count gfx950 halves) + WRITE_SIZE {write / 1e6:.1f} MB = "
                       f"{(fetch + write) / 1e6:.1f} MB vs {alg / 1e6:.0f} MB algorithmic ({(fetch + write) / alg:.3f}x; "
                       f"{(2 * fetch + write) / alg:.3f}x if the read count were halved)")
            traffic[w] = {"fetch_bytes": fetch, "write_bytes": write, "traffic_bytes": fetch + write,
                          "algorithmic_bytes": alg, "source": f"profiles/{label}_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"}
            if "SQ_INSTS_VALU" in a and dur:
                avg_us = sum(dur) / len(dur)
                traffic[w]["valu"] = {
                    "wave_instructions_per_launch": a["SQ_INSTS_VALU"],
                    "per_pixel": a["SQ_INSTS_VALU"] * 64 / (mp * 1e6),
                    "kernel_us_under_pmc": avg_us,
                    "ns_per_instruction_per_simd": avg_us * 1e3 * 1024 / a["SQ_INSTS_VALU"],
                    "note": ("gfx950 issues one f32 add/mul/fma per ~1.1 ns per SIMD and one f64 / cvt / cndmask / cmp "
                             "per ~1.8 ns (profiles/*_microbench.txt): at this instruction count the kernel is VALU-bound")
                    if w != "demosaic" else "instruction issue is a fraction of the kernel time: HBM/store-bound"}
        lines += ["Derived:", ""] + [f"- {d}" for d in der] + [""]

    for name in ("bench_default.json", "bench_variants.log", "microbench.txt"):
        f = os.path.join(src, name)
        if not os.path.exists(f):
            f = os.path.join(ROOT, "gpurun_out", name)
        if os.path.exists(f):
            shutil.copy(f, os.path.join(dst, f"{label}_{name}"))
    f = os.path.join(src, "gpu_tests.log")
    if os.path.exists(f):
        tail = open(f).read().strip().splitlines()[-3:]
        lines += ["## GPU test run", "", "```"] + tail + ["```", ""]
    open(os.path.join(dst, f"{label}_summary.md"), "w").write("\n".join(lines))
    if traffic:
        json.dump(traffic, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
