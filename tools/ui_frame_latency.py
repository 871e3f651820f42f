#!/usr/bin/env python3
"""Latency of one interactive "UI frame" through the operator layer, as the reference's front-ends drive it
(web/main.ts:761-826: 4 scalar setters + 4 curve setters from control points, apply, readback) on the three
preview sizes of the reference's pyramid (400 / 1280 px long edge and the full 6000x4000 frame)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rawphotoforge_amd as R

gpu = R.GpuProcessor(0)
rng = np.random.default_rng(0)
for (w, h) in ((400, 267), (1280, 853), (6000, 4000)):
    img = rng.random((h, w, 3), dtype=np.float32)
    ed = R.PhotoEditor.new_from_rgb_f32(gpu, img, h, w)
    def frame(i):
        t0 = time.perf_counter()
        ed.set_whitebalance(10 + i % 3, -5)
        ed.set_vignette(-20)
        ed.set_lens_distortion_correction(0)
        ed.set_tone(0.3, 10, 20, -10, 0, 0)
        ed.set_brightness_tone_curve(control_points_x=[0, 16384, 49152, 65535], control_points_y=[0, 18000 + i, 50000, 65535])
        ed.set_oklch_hue_curve(control_points_x=[0, 65535], control_points_y=[0, 65535])
        ed.set_oklch_saturation_curve(control_points_x=[0, 65535], control_points_y=[34000, 34000])
        ed.set_oklch_lightness_curve(control_points_x=[0, 65535], control_points_y=[32767, 32767])
        t1 = time.perf_counter()
        ed.apply_adjustments(); ed.sync()
        t2 = time.perf_counter()
        u8 = ed.to_rgb_array3_u8()
        t3 = time.perf_counter()
        return t1 - t0, t2 - t1, t3 - t2
    for i in range(3): frame(i)
    ts = np.median(np.array([frame(i) for i in range(10)]), axis=0) * 1e3
    print(f"{w}x{h}: setters (4 PCHIP curves) {ts[0]:.2f} ms | apply incl. curve upload {ts[1]:.2f} ms | readback u8 {ts[2]:.2f} ms | total {ts.sum():.2f} ms")
