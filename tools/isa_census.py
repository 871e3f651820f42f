#!/usr/bin/env python3
"""Static instruction census of a fused kernel's inner loop (one trip = 128 or 256 pixels per wavefront), from the device
assembly:   hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -DRPF_TU=1 \
                  -mllvm -amdgpu-sched-strategy=max-ilp --cuda-device-only -S rawphotoforge_amd/csrc/rpf_kernels.hip -o k.s
            python tools/isa_census.py k.s '<mangled kernel name or a substring of it>' [...]
Counts the instructions of the innermost loop by issue class (rates measured with tools/microbench.hip on gfx950, ns per
wave-instruction per SIMD at 8 waves per SIMD: profiles/r03_microbench.txt) and by the stage they belong to, which is
read off the instruction stream by the markers the source leaves in it (the order of stages inside a step is fixed).
Basic blocks that hold the IEEE square-root fallback (v_sqrt_f32: taken only outside [2^-96, 2^96)) are left out."""
import re
import sys
from collections import Counter, OrderedDict

# ns per wave-instruction per SIMD (profiles/r03_microbench.txt, 8 waves per SIMD)
FULL, HALF, QUARTER = 1.05, 1.76, 3.42
FULL_OPS = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32",
            "v_and_b32", "v_or_b32", "v_xor_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_lshrrev_b32", "v_ashrrev_i32",
            "v_mov_b32", "v_not_b32"}
QUARTER_OPS = {"v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_swap_b32", "v_permlane32_swap_b32",
               "v_permlane16_swap_b32", "v_rcp_iflag_f32"}


def classify(op: str, text: str) -> str:
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if op.startswith("s_"):
        return "salu/smem" if not op.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch", "s_barrier")) else "ctl"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("buffer_", "global_", "flat_")):
        return "vmem"
    if "_f64" in base or "_b64" in base and base.startswith("v_"):
        if base in ("v_mov_b64",):
            return "f64-rate (mov_b64)"
        return "f64" if not base.startswith("v_cvt") else "cvt"
    if base.startswith("v_cvt"):
        return "cvt"
    if "dpp" in text or op.endswith("_dpp"):
        return "dpp"
    if base in QUARTER_OPS:
        return "quarter"
    if base in FULL_OPS and "sdwa" not in text:
        # an SGPR source halves the rate of an otherwise full-rate f32 / int operation
        srcs = text.split(",")[1:]
        if base.startswith(("v_add_f32", "v_sub", "v_mul_f32", "v_fma", "v_fmac")) and any(re.match(r"\s*-?\|?s\d|\s*-?\|?s\[", x) for x in srcs):
            return "half (sgpr operand)"
        return "full"
    if base.startswith("v_cndmask"):
        return "select"
    if base.startswith("v_cmp"):
        return "compare"
    if base.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "readlane"
    return "half (other)"


COST = {"full": FULL, "select": HALF, "compare": HALF, "dpp": HALF, "cvt": HALF, "f64": 1.87, "f64-rate (mov_b64)": HALF,
        "half (other)": HALF, "half (sgpr operand)": HALF, "readlane": HALF, "quarter": QUARTER}


def kernel_body(lines, name):
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and name in l and l.rstrip().split(":")[0].endswith("E") or
                 (l.startswith("_Z") and name in l and ":" in l))
    end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def innermost_loop(body):
    """(first, last) line indices of the deepest loop: from its header label to the last branch back to it."""
    headers = [(i, int(m.group(1))) for i, l in enumerate(body) for m in [re.search(r"Loop Header: Depth=(\d+)", l)] if m]
    if not headers:
        raise SystemExit("no loop found")
    depth = max(d for _, d in headers)
    # the longest loop below the item loop (depth 1): the step loop — the prologue may hold short waterfall loops, and the
    # progress-priority bookkeeping a two-instruction loop of its own inside the step loop
    best = None
    for i, d in headers:
        if d != depth and not (depth > 2 and d == depth - 1):
            continue
        while not body[i].startswith(".LBB"):  # the comment may continue below the label's own line
            i -= 1
        label = body[i].split(":")[0].strip()
        targets = [label]
        # the loop's latch block (window slide) is usually laid out in FRONT of the header and falls through into it
        k = i - 1
        while k > 0 and not body[k].startswith(".LBB"):
            k -= 1
        if k > 0 and ("Header=" + label.lstrip(".L")) in body[k].replace(" ", ""):
            targets.append(body[k].split(":")[0].strip())
        pat = r"s_c?branch\w*\s+(" + "|".join(re.escape(t) for t in targets) + r")\b"
        backs = [j for j in range(i, len(body)) if re.search(pat, body[j])]
        if backs and (best is None or backs[-1] - i > best[1] - best[0]):
            best = (i, backs[-1])
    return best


def census(path, name):
    lines = open(path).read().splitlines()
    body = kernel_body(lines, name)
    lo, hi = innermost_loop(body)
    # the compiler may place the loop's latch block (window slide) in front of the header: include it — but only a block that
    # names THIS loop's header (the block in front of k_hp's step loop is the work item's prologue, a block of the outer loop:
    # counting it added 143 instructions per trip to every total of that kernel until round 4)
    header = body[lo].split(":")[0].strip().lstrip(".L")
    k = lo - 1
    while k > 0 and not body[k].startswith(".LBB"):
        k -= 1
    if k > 0 and ("Header=" + header) in body[k].replace(" ", ""):
        lo = k
    # split into basic blocks at labels; drop blocks with v_sqrt_f32 (the IEEE fallback of sqrt_cr)
    blocks, cur = [], []
    for l in body[lo:hi + 1]:
        if l.startswith(".LBB") or l.startswith("; %bb."):
            if cur:
                blocks.append(cur)
            cur = []
        cur.append(l)
    if cur:
        blocks.append(cur)
    cls, ops = Counter(), Counter()
    for b in blocks:
        if any("v_sqrt_f32" in l for l in b):
            continue
        for l in b:
            t = l.strip()
            if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
                continue
            op = t.split()[0]
            if not re.match(r"^[vsdgbf]\w+$", op) or op in ("s_waitcnt", "s_nop"):
                if op in ("s_waitcnt", "s_nop"):
                    cls["ctl"] += 1
                continue
            c = classify(op, t)
            cls[c] += 1
            if c not in ("ctl", "salu/smem", "lds", "vmem"):
                ops[re.sub(r"_(e32|e64)$", "", op) + (" dpp" if "dpp" in t and not op.endswith("dpp") else "")] += 1
    return cls, ops


def main():
    path = sys.argv[1]
    for name in sys.argv[2:]:
        cls, ops = census(path, name)
        valu = {k: v for k, v in cls.items() if k in COST}
        n = sum(valu.values())
        ns = sum(COST[k] * v for k, v in valu.items())
        px = 256 if "k_hp" in name else 128   # k_hp walks two rows of 128 columns per trip, k_stream two rows of 64
        print(f"## `{name}` — one trip of the inner loop = {px} pixels per wavefront\n")
        print("| class | instructions | ns per trip per SIMD at the measured issue cost |")
        print("|---|---|---|")
        for k in sorted(valu, key=lambda k: -COST[k] * valu[k]):
            print(f"| {k} | {valu[k]} | {COST[k] * valu[k]:.0f} |")
        print(f"| **VALU total** | **{n}** ({n * 64 / px:.0f} per pixel; {n * 128 / px:.0f} per 128 pixels) | **{ns:.0f}** ({ns * 128 / px:.0f} per 128 pixels) |")
        for k in ("salu/smem", "lds", "vmem", "ctl"):
            print(f"| {k} | {cls.get(k, 0)} | |")
        print("\nby opcode: " + ", ".join(f"{o} {c}" for o, c in ops.most_common()) + "\n")


if __name__ == "__main__":
    main()
