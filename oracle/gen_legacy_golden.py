#!/usr/bin/env python3
"""Generate tests/golden/legacy_helpers.npz from the reference's own Python code.

TEST INFRASTRUCTURE ONLY — run in the build container, where /root/reference is
mounted; the GPU box never sees the reference, only the committed .npz.

The reference package ``raw_image_editor`` cannot be imported (ordinary
ModuleNotFoundError: photo_metadata / slangpy / rawpy / sam2 are not installed,
SURVEY.md §8c), so the PURE numpy/scipy helpers are lifted out of
``python-legacy/raw_image_editor/editor.py`` by AST (function and assignment
nodes only, none of the module's imports or top-level side effects) and
executed as they stand:

    apply_gamma                       editor.py:47-76
    oklab_to_lms_m / lms_to_srgb      editor.py:82-92
    oklch_to_rgb_vectorized           editor.py:121-150
    RAWImageEditorBase._create_tone_lut_from_params   editor.py:712-778

Only inputs and outputs are written — no reference source text.
"""
import ast
import os
import sys

import numpy as np
from scipy import interpolate  # noqa: F401  (used by the lifted tone-LUT builder)

REF = "/root/reference/python-legacy/raw_image_editor/editor.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "legacy_helpers.npz")


def lift():
    tree = ast.parse(open(REF, encoding="utf-8").read())
    want_funcs = {"apply_gamma", "oklch_to_rgb_vectorized"}
    want_assign = {"oklab_to_lms_m", "lms_to_srgb"}
    body = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in want_funcs:
            body.append(node)
        elif isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id in want_assign for t in node.targets):
            body.append(node)
        elif isinstance(node, ast.ClassDef) and node.name == "RAWImageEditorBase":
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and sub.name == "_create_tone_lut_from_params":
                    body.append(sub)
    ns = {"np": np, "interpolate": interpolate}
    exec(compile(ast.Module(body=body, type_ignores=[]), REF, "exec"), ns)
    return ns


def main():
    if not os.path.exists(REF):
        sys.exit("reference not mounted; golden vectors are committed under tests/golden/")
    ns = lift()
    out = {}

    # apply_gamma on a 4097-point grid plus random values, both gamma tuples the reference uses:
    # the GUI's (2.222, 4.5) (raw_photo_forge.py:1727) and the function default (2.222, 4.5/255).
    rng = np.random.Generator(np.random.PCG64(20260630))
    grid = np.concatenate([np.linspace(0.0, 1.0, 4097, dtype=np.float32),
                           rng.random(4096, dtype=np.float32),
                           np.array([-0.25, 1.5, 1e-6, 1e-4, 3e-4, 1e-3], np.float32)])
    out["gamma_in"] = grid
    out["gamma_gui"] = ns["apply_gamma"](grid.copy(), (2.222, 4.5))
    out["gamma_default"] = ns["apply_gamma"](grid.copy())

    # oklch_to_rgb_vectorized on a 32x32x8 L/C/H grid (float64 in, float64 out)
    L, Cc, H = np.meshgrid(np.linspace(0.0, 1.0, 32), np.linspace(0.0, 0.4, 32), np.linspace(0.0, 1.0, 8),
                           indexing="ij")
    out["oklch_L"], out["oklch_C"], out["oklch_H"] = L, Cc, H
    out["oklch_rgb"] = ns["oklch_to_rgb_vectorized"](L, Cc, H)

    # tone LUT, sub-sampled every 16th entry, for 6 slider sets (exposure, contrast, shadow, highlight, black, white)
    sets = [(0.0, 0, 0, 0, 0, 0), (0.5, 20, 30, -20, 5, -5), (-1.0, -50, 0, 0, 0, 0), (1.5, 0, 100, -100, 0, 0),
            (0.0, 100, 0, 0, -100, 100), (-0.3, 10, -40, 60, 20, -30)]
    out["tone_sets"] = np.array(sets, np.float64)
    out["tone_lut_16"] = np.stack([ns["_create_tone_lut_from_params"](None, *s)[::16] for s in sets])
    out["tone_lut_neutral_is_arange"] = np.array(
        np.array_equal(ns["_create_tone_lut_from_params"](None), np.arange(65536, dtype=np.float32)))

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **out)
    print("wrote", os.path.normpath(OUT), {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
