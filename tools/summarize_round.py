#!/usr/bin/env python3
"""Condense one pass of tools/gpu_profile.sh (gpurun_out/<tag>/) into profiles/<label>_summary.md, so that the bench line's
`roofline.frac` follows from the committed files:

  python tools/summarize_round.py <tag> <label>

* kernel durations of the TIMED REGION only: the last `steps` launches of the fused kernel in the rocprofv3 kernel trace of the
  default command (preheat and warm-up launches dropped) — median, minimum, mean — next to the HIP-event figures bench.py printed
  in that same process and to the default command's own line on the same box;
* counters per launch (SQ / TCP / FETCH_SIZE / WRITE_SIZE passes), exact and fast kernel;
* FETCH_SIZE calibrated on a known byte count in this kernel's own access pattern (MI355X_MICROARCH.md, HBM section: "other
  access widths are uncalibrated"): the frame-pool pass, where every input byte must come from HBM.
Copies the rocprofv3 kernel_stats CSVs to profiles/ and rewrites profiles/hbm_traffic.json (read by bench.py)."""
import collections
import csv
import glob
import json
import os
import shutil
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MP = 24.0e6


def timed_region(trace_csv, steps):
    rows = [r for r in csv.DictReader(open(trace_csv)) if "rpf::k_hp" in r["Kernel_Name"] or "rpf::k_stream" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    return rows[0]["Kernel_Name"] if rows else "?", len(d), d[-steps:]


def counters(src, name):
    agg, dur = collections.OrderedDict(), []
    for d in sorted(glob.glob(os.path.join(src, f"pmc_{name}_*"))):
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "rpf::k_" in r["Kernel_Name"]:
                    agg.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if "rpf::k_" in r["Kernel_Name"]:
                    dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return {k: sum(v) / len(v) for k, v in agg.items()}, (sum(dur) / len(dur) if dur else None)


def main():
    tag, label = sys.argv[1], sys.argv[2]
    src, dst = os.path.join(ROOT, "gpurun_out", tag), os.path.join(ROOT, "profiles")
    L = [f"# rocprofv3 summary — {label}", "",
         "One MI355X box, one pass of `tools/gpu_profile.sh`: the default command `python bench.py` (its JSON line is "
         f"`profiles/{label}_bench_default.json`), the same command without its CPU leg and `secondary` block under "
         "`rocprofv3 --kernel-trace --stats` (exact and fast kernel, edit sets D and S), and counter passes "
         "(`--kernel-trace --pmc …`, counters only) of `bench.py --steps 50 --warmup 10 --preheat 60`.", ""]
    bd = json.load(open(os.path.join(src, "bench_default.json")))
    shutil.copy(os.path.join(src, "bench_default.json"), os.path.join(dst, f"{label}_bench_default.json"))
    L += ["## The default command on this box", "",
          f"`value` {bd['value']} MP/s, `ms_per_step` {bd['ms_per_step']}, HIP-event kernel mean {bd['roofline']['kernel_ms_avg']} ms "
          f"(median {bd['config']['kernel_ms_median']}, min {bd['config']['kernel_ms_min']}) -> `roofline.frac` {bd['roofline']['frac']}.", ""]
    sec = bd.get("secondary", {})
    if "fast_precision" in sec:
        fp = sec["fast_precision"]
        a = fp["against_exact_kernel"]
        L += [f"`secondary.fast_precision`: kernel {fp['kernel_ms_avg']} ms -> frac **{fp['frac']}** ({fp['MP/s']} MP/s); against the exact "
              f"kernel's frame: median {a['median_ulp']} ulp, {a['pixels_within_1e-5']:.4%} of the pixels within 1e-5, index-flip rate "
              f"{a['index_flip_rate']:.4%}, max |difference| {a['max_abs_difference']}.  `secondary.edit_set_S`: {sec['edit_set_S']['kernel_ms_avg']} ms "
              f"(frac {sec['edit_set_S']['frac']}); `fresh_input` with / without prefetch: {sec['fresh_input']['with_prefetch']['kernel_ms_avg']} / "
              f"{sec['fresh_input']['without_prefetch']['kernel_ms_avg']} ms (frac {sec['fresh_input']['with_prefetch']['frac']} / "
              f"{sec['fresh_input']['without_prefetch']['frac']}).", ""]

    L += ["## Kernel durations of the timed region (rocprofv3 kernel trace; preheat and warm-up launches dropped)", "",
          "| run | kernel | launches in the trace | timed launches | median µs | min µs | mean µs | mean over ALL launches µs | bench.py in that process: kernel mean ms / ms per step | frac from the trace's median |",
          "|---|---|---|---|---|---|---|---|---|---|"]
    for d in sorted(glob.glob(os.path.join(src, "prof_*"))):
        name = os.path.basename(d)[5:]
        for f in glob.glob(os.path.join(d, "*_kernel_stats.csv")):
            shutil.copy(f, os.path.join(dst, f"{label}_{os.path.basename(f)}"))
        tr = glob.glob(os.path.join(d, "*_kernel_trace.csv"))
        bj = os.path.join(src, f"bench_under_rocprof_{name}.json")
        if not tr or not os.path.exists(bj):
            continue
        b = json.load(open(bj))
        kern, n_all, d_timed = timed_region(tr[0], b["steps"])
        allmean = None
        for r in csv.DictReader(open(glob.glob(os.path.join(d, "*_kernel_stats.csv"))[0])):
            if "rpf::k_hp" in r["Name"] or "rpf::k_stream" in r["Name"]:
                allmean = float(r["AverageNs"]) / 1e3
        med = statistics.median(d_timed)
        L.append(f"| {name} | `{kern}` | {n_all} | {len(d_timed)} | **{med:.1f}** | {min(d_timed):.1f} | {sum(d_timed) / len(d_timed):.1f} | "
                 f"{allmean:.1f} | {b['roofline']['kernel_ms_avg']} / {b['ms_per_step']} | {18 * MP / (med * 1e-6) / 8e12:.4f} |")
    L += ["", "The trace's timed-region figures are the kernel alone (dispatch begin to end); bench.py's HIP-event mean adds the two "
          "event packets (~1-2 µs), and `ms_per_step` the launch gap.  The mean over ALL launches (what `--stats` prints) includes the "
          "1000 preheat launches, the first ~50 of which run below the settled clock.", ""]

    traffic = {}
    cal = None
    fb, _ = counters(src, "batch")
    if "FETCH_SIZE" in fb:
        true_in = 48.0e6 * (1 + 4.0 / 128)   # every strip of 128 columns re-reads 4 halo columns of its neighbours
        cal = true_in / (fb["FETCH_SIZE"] * 1024)
        L += ["## FETCH_SIZE calibrated on this kernel's own access pattern", "",
              f"Pool of 8 different frames (`--workload batch --frames 8 --no-prefetch`: 384 MB of input and 384 MB of output between two "
              f"uses of a frame — nothing stays in the 256-MiB Infinity Cache, every input byte comes from HBM): FETCH_SIZE "
              f"{fb['FETCH_SIZE'] * 1024 / 1e6:.1f} MB per launch for {true_in / 1e6:.1f} MB read (48 MB of samples + 4 halo columns per "
              f"128-column strip), WRITE_SIZE {fb.get('WRITE_SIZE', 0) * 1024 / 1e6:.1f} MB.  The kernel's 2-byte-per-lane row loads are "
              f"therefore tallied at **{1 / cal:.3f}** of their bytes (the guide calibrates only 16-B-per-lane streams: ½), and the same "
              f"FETCH_SIZE with the frame resident (below) does NOT mean that a quarter of the input stays in the L2: the counter sits "
              f"on the L2's memory side and sees the same requests whether the Infinity Cache or HBM serves them.", ""]
    for name, title in (("develop", "exact kernel (the headline)"), ("developfast", "RPF_PRECISION_FAST kernel")):
        a, avg_us = counters(src, name)
        if not a:
            continue
        L += [f"## PMC counters, {title}, per launch", "", "| counter | value |", "|---|---|"] + [f"| {k} | {v:.6g} |" for k, v in a.items()] + [""]
        der = []
        if "GRBM_GUI_ACTIVE" in a and avg_us:
            der.append(f"kernel duration under the counter passes: {avg_us:.1f} µs; clock = GRBM_GUI_ACTIVE / 8 / duration = {a['GRBM_GUI_ACTIVE'] / 8 / avg_us / 1e3:.2f} GHz")
        if "SQ_INSTS_VALU" in a:
            der.append(f"VALU wave-instructions per pixel: **{a['SQ_INSTS_VALU'] * 64 / MP:.1f}** ({a['SQ_INSTS_VALU'] / (MP / 128):.0f} per 128 pixels; "
                       f"SALU {a.get('SQ_INSTS_SALU', 0) / (MP / 128):.0f}, LDS {a.get('SQ_INSTS_LDS', 0) / (MP / 128):.0f} per 128 pixels)")
            if avg_us:
                der.append(f"VALU issue: {avg_us * 1e3 * 1024 / a['SQ_INSTS_VALU']:.2f} ns per VALU instruction per SIMD")
        if "SQ_WAVE_CYCLES" in a and "SQ_WAIT_ANY" in a:
            der.append(f"wave time: ACTIVE_INST_ANY {a.get('SQ_ACTIVE_INST_ANY', 0) / a['SQ_WAVE_CYCLES']:.0%}, WAIT_INST_ANY "
                       f"{a.get('SQ_WAIT_INST_ANY', 0) / a['SQ_WAVE_CYCLES']:.0%}, WAIT_ANY {a['SQ_WAIT_ANY'] / a['SQ_WAVE_CYCLES']:.0%}")
        if "SQ_LDS_BANK_CONFLICT" in a and a.get("SQ_LDS_IDX_ACTIVE"):
            der.append(f"LDS bank-conflict cycles / LDS active cycles: {a['SQ_LDS_BANK_CONFLICT'] / a['SQ_LDS_IDX_ACTIVE']:.0%}")
        if "TCP_TCC_READ_REQ_sum" in a:
            der.append(f"L1 -> L2 read requests per pixel: **{a['TCP_TCC_READ_REQ_sum'] / MP:.3f}**; TCP_PENDING_STALL_CYCLES / TCP_GATE_EN1 = "
                       f"{a.get('TCP_PENDING_STALL_CYCLES_sum', 0) / max(1.0, a.get('TCP_GATE_EN1_sum', 1)):.0%}")
        if "FETCH_SIZE" in a and "WRITE_SIZE" in a:
            fetch, write = a["FETCH_SIZE"] * 1024, a["WRITE_SIZE"] * 1024
            alg = 18 * MP
            s = f"HBM-side traffic per launch: FETCH_SIZE {fetch / 1e6:.1f} MB raw"
            if cal:
                s += f" = {fetch * cal / 1e6:.1f} MB calibrated (x {cal:.3f}, above)"
            s += f" + WRITE_SIZE {write / 1e6:.1f} MB = **{(fetch * (cal or 1) + write) / 1e6:.1f} MB** vs {alg / 1e6:.0f} MB algorithmic ({(fetch * (cal or 1) + write) / alg:.3f}x)"
            der.append(s)
            if name == "develop":
                traffic["develop"] = {"fetch_bytes_raw": fetch, "fetch_calibration": cal, "fetch_bytes": fetch * (cal or 1), "write_bytes": write,
                                      "traffic_bytes": fetch * (cal or 1) + write, "algorithmic_bytes": alg,
                                      "source": f"profiles/{label}_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE calibrated on the frame-pool pass)"}
                if "SQ_INSTS_VALU" in a and avg_us:
                    traffic["develop"]["valu"] = {"wave_instructions_per_launch": a["SQ_INSTS_VALU"], "per_pixel": a["SQ_INSTS_VALU"] * 64 / MP,
                                                  "kernel_us_under_pmc": avg_us, "ns_per_instruction_per_simd": avg_us * 1e3 * 1024 / a["SQ_INSTS_VALU"],
                                                  "note": "gfx950 issues one f32 add/mul/fma per ~1.1 ns per SIMD and one f64 / cvt / cndmask / cmp per ~1.8 ns "
                                                          "(profiles/*_microbench.txt): at this instruction count the kernel is VALU-bound"}
        L += ["Derived:", ""] + [f"- {x}" for x in der] + [""]
    for name in ("gpu_tests.log",):
        f = os.path.join(src, name)
        if os.path.exists(f):
            L += ["## GPU test run", "", "```"] + open(f).read().strip().splitlines()[-3:] + ["```", ""]
    open(os.path.join(dst, f"{label}_summary.md"), "w").write("\n".join(L))
    if traffic:
        old = {}
        try:
            old = json.load(open(os.path.join(dst, "hbm_traffic.json")))
        except (OSError, ValueError):
            pass
        old.update(traffic)
        json.dump(old, open(os.path.join(dst, "hbm_traffic.json"), "w"), indent=1)
    print("\n".join(L))


if __name__ == "__main__":
    main()
