import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import rawphotoforge_amd as R
from rawphotoforge_amd import synthetic as syn, tiles
W, H = 6000, 4000
gpu = R.GpuProcessor(0)
praw = R.RawParams(**syn.RAW)
stream = torch.cuda.current_stream().cuda_stream
out = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
band = tiles.make_band_buffer(syn.bayer_uniform(H, W), device="cuda")
def run(label, masks):
    ed = R.PhotoEditor.for_tile(gpu, W, H, 0, H)
    syn.configure(ed, syn.PARAM_SETS["D"])
    for i, m in enumerate(masks):
        ed.add_mask(f"u{i}", m)
        syn.configure(ed, syn.PARAM_SETS["S"], mask_name=f"u{i}", main=False)
    for _ in range(60): ed.dev_develop(band.data_ptr(), praw, out.data_ptr(), R.OUT_RGBA_F32, stream)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100): ed.dev_develop(band.data_ptr(), praw, out.data_ptr(), R.OUT_RGBA_F32, stream)
    e1.record(); torch.cuda.synchronize()
    print(f"{label:40s} {e0.elapsed_time(e1)/100:.4f} ms", flush=True)
ones = np.ones((H, W), np.float32); neg = -ones
half = neg.copy(); half[:H//2] = 1.0
cols = neg.copy(); cols[:, ::2] = 1.0     # every other column: every wavefront half covered
run("no user mask", [])
run("1 user mask, covers nothing", [neg])
run("1 user mask, covers everything", [ones])
run("1 user mask, top half", [half])
run("1 user mask, every other column", [cols])
run("3 user masks, everything", [ones, ones, ones])
