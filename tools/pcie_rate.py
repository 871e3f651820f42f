#!/usr/bin/env python3
"""PCIe-inclusive rate of the operator layer (host buffers in, host buffers out) — never bench.py's `value`.
Times, for one 6000x4000 frame: editor creation from a host u16 CFA (H2D 48 MB), apply, and the three
readbacks (RGBA32F 384 MB, RGB f32 288 MB, RGB u8 72 MB)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import rawphotoforge_amd as R
from rawphotoforge_amd import synthetic as syn

W, H = 6000, 4000
gpu = R.GpuProcessor(0)
cfa = syn.bayer_uniform(H, W)
raw = R.RawParams(**syn.RAW)
ed = R.PhotoEditor.new_from_bayer_u16(gpu, cfa, raw); ed.apply_adjustments(); ed.sync()   # warm
rows = []
for rep in range(3):
    t0 = time.perf_counter(); ed = R.PhotoEditor.new_from_bayer_u16(gpu, cfa, raw); t1 = time.perf_counter()
    ed.apply_adjustments(); ed.sync(); t2 = time.perf_counter()
    a = ed.to_flat_vec_rgba(); t3 = time.perf_counter()
    b = ed.to_flat_vec_rgb(); t4 = time.perf_counter()
    c = ed.to_rgb_array3_u8(); t5 = time.perf_counter()
    rows.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4))
# the same with caller-owned, already touched output buffers (what a caller that processes many frames does)
oa, ob, oc = np.zeros(W * H * 4, np.float32), np.zeros(W * H * 3, np.float32), np.zeros(W * H * 3, np.uint8)
rows2 = []
for rep in range(3):
    t2 = time.perf_counter(); ed.to_flat_vec_rgba(oa); t3 = time.perf_counter()
    ed.to_flat_vec_rgb(ob); t4 = time.perf_counter()
    ed.to_rgb_array3_u8(oc); t5 = time.perf_counter()
    rows2.append((t3 - t2, t4 - t3, t5 - t4))
m2 = np.median(np.array(rows2), axis=0)
print(f"into reused host buffers: get rgba f32 {m2[0]*1e3:.1f} ms ({384/m2[0]/1e3:.1f} GB/s) | get rgb f32 {m2[1]*1e3:.1f} ms | get rgb u8 {m2[2]*1e3:.1f} ms")
m = np.median(np.array(rows), axis=0)
mp = W * H / 1e6
print(f"create(H2D 48MB) {m[0]*1e3:.1f} ms | apply {m[1]*1e3:.2f} ms | get rgba f32 {m[2]*1e3:.1f} ms | get rgb f32 {m[3]*1e3:.1f} ms | get rgb u8 {m[4]*1e3:.1f} ms")
for name, t in (("rgba f32", m[0] + m[1] + m[2]), ("rgb f32", m[0] + m[1] + m[3]), ("rgb u8", m[0] + m[1] + m[4])):
    print(f"end to end, host in -> freshly allocated host out ({name}): {t*1e3:.1f} ms = {mp/t:.0f} MP/s")
for name, t in (("rgba f32", m[0] + m[1] + m2[0]), ("rgb f32", m[0] + m[1] + m2[1]), ("rgb u8", m[0] + m[1] + m2[2])):
    print(f"end to end, host in -> reused host out ({name}): {t*1e3:.1f} ms = {mp/t:.0f} MP/s")
