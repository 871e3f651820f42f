"""Parity of the gfx950 kernels with the oracle, through the C ABI (run on the GPU box: -m gpu).

Tolerance (BASELINE.json north_star): <= 1 ULP per channel against the correctly rounded oracle,
bit-exact for the integer stages.  Because the kernels evaluate the transcendentals to < 2^-49 and
round once, the expectation is in fact bit-for-bit equality; `assert_parity` enforces the 1-ULP
bound and, on the small cases, exact equality.
"""
import hashlib
import os

import numpy as np
import pytest

import oracle as O
import rawphotoforge_amd as R
from tests import util

pytestmark = pytest.mark.gpu

TOL_ULP = 1


def assert_parity(got, want, exact=True):
    got = np.asarray(got, np.float32).reshape(want.shape)
    d = util.ulp_diff(got, want)
    if exact:
        assert d.max() == 0, f"{int((d > 0).sum())} of {d.size} values differ, max {int(d.max())} ulp"
    else:
        assert d.max() <= TOL_ULP, f"max {int(d.max())} ulp, {int((d > 0).sum())} of {d.size} differ"


def bayer_editor(gpu, cfa, rp):
    return R.PhotoEditor.new_from_bayer_u16(gpu, cfa, util.to_raw(rp))


# ---- BASELINE config 2: demosaic only ---------------------------------------------------------
@pytest.mark.parametrize("variant", [0, 1, 3, 4])
@pytest.mark.parametrize("shape", [(64, 64), (67, 130), (19, 257), (3, 3), (8, 5), (301, 6)])
def test_demosaic_bit_exact(gpu, shape, variant):
    gpu.set_kernel_variant(variant)
    try:
        for pat in (O.CFA_RGGB, O.CFA_BGGR, O.CFA_GRBG, O.CFA_GBRG):
            cfa = util.bayer_uniform(*shape, seed=100 + pat)
            rp = O.RawParams(cfa_pattern=pat, black=(500, 520, 530, 510), wb=(2.1, 1.0, 1.02, 1.6))
            ed = bayer_editor(gpu, cfa, rp)
            assert_parity(ed.rgba(), O.demosaic(cfa, rp))   # image before apply = linear frame
    finally:
        gpu.set_kernel_variant(0)


def test_demosaic_saturated_and_below_black(gpu):
    cfa = util.bayer_uniform(40, 72)
    cfa[::3, ::5] = 0
    cfa[1::4, 2::7] = 65535
    rp = util.raw_params()
    assert_parity(bayer_editor(gpu, cfa, rp).rgba(), O.demosaic(cfa, rp))


# ---- BASELINE config 3: fused full chain ------------------------------------------------------
GOLD = np.load(os.path.join(util.GOLDEN, "chain_golden.npz"))


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 6, 7, 8, 9, 10, 11, 20])   # every A/B variant of rpf_set_kernel_variant: same bits
@pytest.mark.parametrize("pname", ["D", "S"])
@pytest.mark.parametrize("tag", ["U", "N"])
def test_develop_matches_committed_golden(gpu, tag, pname, variant):
    gpu.set_kernel_variant(variant)
    try:
        cfa = GOLD[f"cfa64_{tag}"]
        p = util.params_D() if pname == "D" else util.params_S()
        ed = bayer_editor(gpu, cfa, util.raw_params())
        util.configure_editor(ed, p)
        ed.apply_adjustments()
        assert_parity(ed.rgba(), GOLD[f"develop64_{tag}_{pname}"])
        big = (util.bayer_uniform if tag == "U" else util.bayer_natural)(256, 256)
        ed = bayer_editor(gpu, big, util.raw_params())
        util.configure_editor(ed, p)
        ed.apply_adjustments()
        digest = np.frombuffer(hashlib.sha256(ed.to_flat_vec_rgba().tobytes()).digest(), np.uint8)
        assert np.array_equal(digest, GOLD[f"sha256_develop256_{tag}_{pname}"])
    finally:
        gpu.set_kernel_variant(0)


@pytest.mark.parametrize("shape", [(33, 47), (129, 131), (16, 1000), (500, 9)])
def test_develop_ragged_sizes_general_curves(gpu, shape):
    cfa = util.bayer_natural(*shape)
    rp = O.RawParams(cfa_pattern=O.CFA_GBRG)
    p = util.params_G()
    p.vignette = 45
    ed = bayer_editor(gpu, cfa, rp)
    util.configure_editor(ed, p)
    ed.apply_adjustments()
    assert_parity(ed.rgba(), O.develop(cfa, rp, [p]))


def test_develop_with_user_masks(gpu):
    h, w = 70, 150
    cfa = util.bayer_uniform(h, w)
    rp = util.raw_params()
    rng = np.random.default_rng(8)
    main, a, b = util.params_S(), util.params_G(), O.EditParameters(exposure=1.0, contrast=40, wb_temperature=60)
    b.lightness_tone_curve = np.full(65536, 20000, np.int32)
    ma = (rng.random((h, w)) > 0.4).astype(np.float32) * rng.choice([1.0, 0.7], (h, w)).astype(np.float32)
    mb = np.zeros((h, w), np.float32)
    mb[10:50, 20:120] = 1.0
    ed = bayer_editor(gpu, cfa, rp)
    ed.add_mask("a", ma)
    ed.add_mask("b", mb)
    util.configure_editor(ed, main)
    util.configure_editor(ed, a, "a", main=False)
    util.configure_editor(ed, b, "b", main=False)
    ed.apply_adjustments()
    masks = np.stack([np.ones((h, w), np.float32), O.binarize_mask(ma), O.binarize_mask(mb)])
    want = O.develop(cfa, rp, [main, a, b], masks=masks)
    assert_parity(ed.rgba(), want)
    ed.remove_mask("a")
    ed.apply_adjustments()
    assert_parity(ed.rgba(), O.develop(cfa, rp, [main, b], masks=masks[[0, 2]]))
    ed.reset()
    ed.apply_adjustments()
    assert_parity(ed.rgba(), O.develop(cfa, rp, [O.EditParameters()]))


def test_user_masks_over_a_pristine_main(gpu):
    """"main" untouched, only the user masks carry edits: the multi-mask kernels with a compile-time "main"
    (MODE_DEVELOP_MP; _MPH when every mask's hue curve is the identity, where cos / sin of the hue come with the index
    "main" settled) against the oracle, overlapping masks included."""
    h, w = 66, 200
    cfa = util.bayer_uniform(h, w, seed=31)
    rp = util.raw_params()
    rng = np.random.default_rng(9)
    ma = np.where(rng.random((h, w)) > 0.3, 1.0, -1.0).astype(np.float32)
    mb = np.full((h, w), -1.0, np.float32)
    mb[5:60, 40:190] = 1.0
    masks = np.stack([np.ones((h, w), np.float32), O.binarize_mask(ma), O.binarize_mask(mb)])
    s2 = util.params_S()
    s2.exposure, s2.contrast = -0.7, 35
    for pa, pb in ((util.params_S(), s2),                 # identity hue curves everywhere (_MPH)
                   (util.params_S(), util.params_G())):   # one mask with a hue curve (_MP)
        ed = bayer_editor(gpu, cfa, rp)
        ed.add_mask("a", ma)
        ed.add_mask("b", mb)
        util.configure_editor(ed, util.params_D())
        util.configure_editor(ed, pa, "a", main=False)
        util.configure_editor(ed, pb, "b", main=False)
        ed.apply_adjustments()
        assert_parity(ed.rgba(), O.develop(cfa, rp, [util.params_D(), pa, pb], masks=masks))


def test_fused_equals_demosaic_then_apply(gpu):
    cfa = util.bayer_natural(96, 200)
    rp = util.raw_params()
    p = util.params_S()
    fused = bayer_editor(gpu, cfa, rp)
    util.configure_editor(fused, p)
    fused.apply_adjustments()
    lin = bayer_editor(gpu, cfa, rp).rgba()
    two = R.PhotoEditor.new_from_rgb_f32(gpu, lin[..., :3].copy(), 96, 200)
    util.configure_editor(two, p)
    two.apply_adjustments()
    assert np.array_equal(fused.to_flat_vec_rgba(), two.to_flat_vec_rgba())


# ---- the reference-equivalent back half (wgsl main on RGBA32F) -------------------------------
def _rgb_image(h, w, seed=3):
    rng = np.random.default_rng(seed)
    img = rng.random((h, w, 3), dtype=np.float32) * np.float32(1.2) - np.float32(0.05)
    img[0, :8] = 0
    img[1, :8] = rng.random((8, 1), dtype=np.float32)
    return img


@pytest.mark.parametrize("pname", ["D", "S", "G"])
def test_apply_rgb_editor(gpu, pname):
    h, w = 61, 83
    img = _rgb_image(h, w)
    p = {"D": util.params_D, "S": util.params_S, "G": util.params_G}[pname]()
    ed = R.PhotoEditor.new_from_rgb_f32(gpu, img, h, w)
    assert np.array_equal(ed.to_flat_vec_rgb().reshape(h, w, 3), img)      # image.rs:135-139, 202-277
    assert (ed.rgba()[..., 3] == 1).all()
    util.configure_editor(ed, p)
    ed.apply_adjustments()
    rgba = np.concatenate([img, np.ones((h, w, 1), np.float32)], -1)
    want = O.apply(rgba, [p])
    assert_parity(ed.rgba(), want)
    assert np.array_equal(ed.to_flat_vec_rgb().reshape(h, w, 3), want[..., :3])
    assert np.array_equal(ed.to_rgb_array3_u8(), O.rgba_to_rgb_u8(want))   # image.rs:375-383


@pytest.mark.parametrize("dist", [-100, -20, 35, 100])
def test_lens_distortion(gpu, dist):
    h, w = 90, 140
    img = _rgb_image(h, w, seed=5)
    p = util.params_S()
    p.lens_distortion = dist
    ed = R.PhotoEditor.new_from_rgb_f32(gpu, img, h, w)
    util.configure_editor(ed, p)
    ed.apply_adjustments()
    rgba = np.concatenate([img, np.ones((h, w, 1), np.float32)], -1)
    assert_parity(ed.rgba(), O.apply(rgba, [p]))
    # Bayer source: demosaic pass + resampling pass
    cfa = util.bayer_natural(h, w)
    eb = bayer_editor(gpu, cfa, util.raw_params())
    util.configure_editor(eb, p)
    eb.apply_adjustments()
    assert_parity(eb.rgba(), O.develop(cfa, util.raw_params(), [p]))


def test_apply_is_repeatable_and_tracks_setters(gpu):
    img = _rgb_image(32, 40)
    ed = R.PhotoEditor.new_from_rgb_f32(gpu, img, 32, 40)
    rgba = np.concatenate([img, np.ones((32, 40, 1), np.float32)], -1)
    for p in (util.params_S(), util.params_D(), util.params_G(), util.params_S()):
        util.configure_editor(ed, p)
        ed.apply_adjustments()
        first = ed.to_flat_vec_rgba()
        ed.apply_adjustments()                     # non-destructive: always from original_image (lib.rs:507-515)
        assert np.array_equal(first, ed.to_flat_vec_rgba())
        assert_parity(first, O.apply(rgba, [p]))


def test_fused_hue_table_follows_the_curves(gpu):
    """One mask with hue / saturation / lightness curves: the fused develop kernel reads them folded into the editor's
    hue table (rpf_math.h HueFused), which has to be rebuilt exactly when what it is made of changes — another hue curve,
    an identity hue curve with the other two left, constant curves (no table), a second mask (no table) and back."""
    import copy
    h, w = 46, 130
    cfa = util.bayer_natural(h, w)
    rp = util.raw_params()
    ed = R.PhotoEditor.new_from_bayer_u16(gpu, cfa, util.to_raw(rp))
    g = util.params_G()
    steps = [g]
    p = copy.deepcopy(g); p.hue_tone_curve = np.roll(np.asarray(g.hue_tone_curve), 1234).tolist(); steps.append(p)      # another hue curve
    p = copy.deepcopy(g); p.hue_tone_curve = list(range(65536)); steps.append(p)                                       # identity hue, the others stay
    p = copy.deepcopy(g); p.saturation_tone_curve = [40000] * 65536; p.lightness_tone_curve = [30000] * 65536; steps.append(p)   # constant gains
    steps += [util.params_S(), g]                                                                                        # no curves at all, and back
    for p in steps:
        util.configure_editor(ed, p)
        ed.apply_adjustments()
        assert_parity(ed.rgba(), O.develop(cfa, rp, [p]))
    rng = np.random.default_rng(3)
    mask = rng.random((h, w), dtype=np.float32)
    ed.add_mask("m1", mask)
    util.configure_editor(ed, util.params_S(), mask_name="m1", main=False)
    ed.apply_adjustments()                                                   # two masks: the general kernel, no table
    masks = np.stack([np.ones((h, w), np.float32), O.binarize_mask(mask, g.mask_range)])
    assert_parity(ed.rgba(), O.develop(cfa, rp, [g, util.params_S()], masks=masks))
    ed.remove_mask("m1")
    ed.apply_adjustments()                                                   # one mask again: the table as it was
    assert_parity(ed.rgba(), O.develop(cfa, rp, [g]))


# ---- device-pointer layer: row tiles with halos, output formats -------------------------------
def test_row_tiles_with_halo_equal_whole_frame(gpu):
    torch = pytest.importorskip("torch")
    h, w = 150, 260
    cfa = util.bayer_uniform(h, w)
    rp = util.raw_params()
    p = util.params_S()
    want = O.develop(cfa, rp, [p])
    bounds = [0, 37, 38, 100, 150]   # odd offsets and a 1-row band on purpose
    pad = np.zeros((h + 4, w), np.uint16)
    pad[2:-2] = cfa
    for y0, y1 in zip(bounds[:-1], bounds[1:]):
        rows = y1 - y0
        band = torch.from_numpy(pad[y0:y0 + rows + 4].copy().view(np.int16)).cuda()   # rows y0-2 .. y1+1
        out = torch.empty((rows, w, 4), dtype=torch.float32, device="cuda")
        ed = R.PhotoEditor.for_tile(gpu, w, h, y0, rows)
        util.configure_editor(ed, p)
        ed.dev_develop(band.data_ptr(), util.to_raw(rp), out.data_ptr(), R.OUT_RGBA_F32,
                       torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert_parity(out.cpu().numpy(), want[y0:y1])
        lin = torch.empty((rows, w, 4), dtype=torch.float32, device="cuda")
        R.dev_demosaic(gpu, band.data_ptr(), util.to_raw(rp), w, h, y0, rows, lin.data_ptr(),
                       torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert_parity(lin.cpu().numpy(), O.demosaic(cfa, rp, y0=y0, rows=rows))


def test_bands_of_a_very_tall_frame(gpu):
    """The geometry of the 8-GPU weak-scaling run (bench.py --gpus 8: frame 6000 x 32000, band y0 = rank * 4000): a band
    far down a frame that is much taller than wide — row reflection at the frame's last rows, the vignette's global
    coordinates (wgsl:169 uses the FRAME's size) and 32-bit row offsets with a large y0 — against the oracle."""
    torch = pytest.importorskip("torch")
    h, w = 32000, 96
    cfa = util.bayer_uniform(h, w)
    rp = util.raw_params()
    p = util.params_S()   # vignette, every tone slider, a brightness curve
    want = O.develop(cfa, rp, [p])
    pad = np.zeros((h + 4, w), np.uint16)
    pad[2:-2] = cfa
    s = torch.cuda.current_stream().cuda_stream
    for y0, rows in ((28000, 4000), (0, 4000), (15998, 4), (31990, 10)):
        band = torch.from_numpy(pad[y0:y0 + rows + 4].copy().view(np.int16)).cuda()
        out = torch.empty((rows, w, 4), dtype=torch.float32, device="cuda")
        ed = R.PhotoEditor.for_tile(gpu, w, h, y0, rows)
        util.configure_editor(ed, p)
        ed.dev_develop(band.data_ptr(), util.to_raw(rp), out.data_ptr(), R.OUT_RGBA_F32, s)
        torch.cuda.synchronize()
        assert_parity(out.cpu().numpy(), want[y0:y0 + rows])


def test_output_formats_of_the_fused_kernel(gpu):
    torch = pytest.importorskip("torch")
    h, w = 64, 200
    cfa = util.bayer_natural(h, w)
    rp = util.raw_params()
    p = util.params_S()
    want = O.develop(cfa, rp, [p])
    pad = np.zeros((h + 4, w), np.uint16)
    pad[2:-2] = cfa
    d = torch.from_numpy(pad.view(np.int16)).cuda()
    ed = R.PhotoEditor.for_tile(gpu, w, h, 0, h)
    util.configure_editor(ed, p)
    s = torch.cuda.current_stream().cuda_stream
    rgb = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    ed.dev_develop(d.data_ptr(), util.to_raw(rp), rgb.data_ptr(), R.OUT_RGB_F32, s)
    u8 = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
    ed.dev_develop(d.data_ptr(), util.to_raw(rp), u8.data_ptr(), R.OUT_RGB_U8, s)
    torch.cuda.synchronize()
    assert np.array_equal(rgb.cpu().numpy(), want[..., :3])
    assert np.array_equal(u8.cpu().numpy(), O.rgba_to_rgb_u8(want))


# ---- BASELINE full size: size-independent properties + oracle on sampled bands ---------------
def test_full_size_6000x4000(gpu):
    torch = pytest.importorskip("torch")
    h, w = 4000, 6000
    cfa = util.bayer_uniform(h, w)
    rp = util.raw_params()
    s = torch.cuda.current_stream().cuda_stream
    pad = np.zeros((h + 4, w), np.uint16)
    pad[2:-2] = cfa
    d = torch.from_numpy(pad.view(np.int16)).cuda()
    for p in (util.params_D(), util.params_S()):
        ed = R.PhotoEditor.for_tile(gpu, w, h, 0, h)
        util.configure_editor(ed, p)
        fused = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
        ed.dev_develop(d.data_ptr(), util.to_raw(rp), fused.data_ptr(), R.OUT_RGBA_F32, s)
        # (1) fused pass == demosaic pass followed by the reference-equivalent pass
        lin = torch.empty_like(fused)
        R.dev_demosaic(gpu, d.data_ptr(), util.to_raw(rp), w, h, 0, h, lin.data_ptr(), s)
        two = torch.empty_like(fused)
        ed.dev_apply(lin.data_ptr(), two.data_ptr(), R.OUT_RGBA_F32, s)
        torch.cuda.synchronize()
        assert torch.equal(fused, two)
        # (2) two row bands with halos == the whole frame
        half = 1998
        for y0, rows in ((0, half), (half, h - half)):
            band = d[y0:y0 + rows + 4].contiguous()
            eb = R.PhotoEditor.for_tile(gpu, w, h, y0, rows)
            util.configure_editor(eb, p)
            ob = torch.empty((rows, w, 4), dtype=torch.float32, device="cuda")
            eb.dev_develop(band.data_ptr(), util.to_raw(rp), ob.data_ptr(), R.OUT_RGBA_F32, s)
            torch.cuda.synchronize()
            assert torch.equal(ob, fused[y0:y0 + rows])
        # (3) oracle on three 8-row bands (top edge, middle, bottom edge)
        host = fused.cpu().numpy()
        assert (host[..., 3] == 1).all() and host[..., :3].min() >= 0 and host[..., :3].max() <= 1
        for y0 in (0, 1996, h - 8):
            assert_parity(host[y0:y0 + 8], O.develop(cfa, rp, [p], y0=y0, rows=8))


def test_band_developer_overlap_split_equals_oracle(gpu):
    """The N > 1 step (interior tile on a side stream, edge tiles after the halo exchange) on one GPU: the
    halos are pre-filled instead of exchanged; the three tiles must tile the band exactly."""
    torch = pytest.importorskip("torch")
    from rawphotoforge_amd import synthetic as syn
    from rawphotoforge_amd import tiles
    h, w = 230, 300
    cfa = util.bayer_natural(h, w)
    rp = util.raw_params()
    want = O.develop(cfa, rp, [util.params_S()])
    pad = np.zeros((h + 4, w), np.uint16)
    pad[2:-2] = cfa
    for rank, (y0, rows) in enumerate(((0, 70), (70, 96), (166, 64))):
        band = torch.from_numpy(pad[y0:y0 + rows + 4].copy().view(np.int16)).cuda()
        out = torch.zeros((rows, w, 4), dtype=torch.float32, device="cuda")
        bd = tiles.BandDeveloper(gpu, w, h, y0, rows, util.to_raw(rp), lambda e: syn.configure(e, syn.PARAM_SETS["S"]),
                                 rank=rank, world=3, overlap=True)
        assert bd.split and [n for _, n, _ in bd.parts] == [rows - 16, 8, 8]
        ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        for _ in range(3):   # repeated steps: stream order covers the reuse of band/out
            bd.step(band, out, R.OUT_RGBA_F32, 16, exchange=False, events=ev)
        torch.cuda.synchronize()
        assert ev[0].elapsed_time(ev[1]) > 0
        assert_parity(out.cpu().numpy(), want[y0:y0 + rows])
        single = tiles.BandDeveloper(gpu, w, h, y0, rows, util.to_raw(rp), lambda e: syn.configure(e, syn.PARAM_SETS["S"]),
                                     rank=rank, world=3, overlap=False)
        out2 = torch.zeros_like(out)
        single.step(band, out2, R.OUT_RGBA_F32, 16, exchange=False)
        torch.cuda.synchronize()
        assert torch.equal(out, out2)


def test_band_developer_split_with_user_masks(gpu):
    """User masks on a split band: the configure callback adds BAND-sized masks, every part (interior, two edge
    tiles) gets its rows of them; the three launches must equal the oracle's whole-frame answer on the band."""
    torch = pytest.importorskip("torch")
    from rawphotoforge_amd import tiles
    h, w = 200, 140
    cfa = util.bayer_uniform(h, w, seed=5)
    rp = util.raw_params()
    rng = np.random.default_rng(21)
    main, a, b = util.params_D(), util.params_S(), util.params_G()
    ma = np.where(rng.random((h, w)) > 0.5, 1.0, -1.0).astype(np.float32)
    mb = np.full((h, w), -1.0, np.float32)
    mb[4:150, 30:120] = 1.0   # crosses the band boundary and both edge tiles of band 1
    masks = np.stack([np.ones((h, w), np.float32), O.binarize_mask(ma), O.binarize_mask(mb)])
    want = O.develop(cfa, rp, [main, a, b], masks=masks)
    pad = np.zeros((h + 4, w), np.uint16)
    pad[2:-2] = cfa
    for rank, (y0, rows) in enumerate(((0, 90), (90, 110))):
        def configure(e, y0=y0, rows=rows):
            e.add_mask("a", ma[y0:y0 + rows])
            e.add_mask("b", mb[y0:y0 + rows].reshape(-1))
            util.configure_editor(e, main)
            util.configure_editor(e, a, "a", main=False)
            util.configure_editor(e, b, "b", main=False)
        band = torch.from_numpy(pad[y0:y0 + rows + 4].copy().view(np.int16)).cuda()
        out = torch.zeros((rows, w, 4), dtype=torch.float32, device="cuda")
        bd = tiles.BandDeveloper(gpu, w, h, y0, rows, util.to_raw(rp), configure, rank=rank, world=2, overlap=True)
        assert bd.split and all(ed.num_masks == 3 for _, _, ed in bd.parts)
        bd.step(band, out, R.OUT_RGBA_F32, 16, exchange=False)
        torch.cuda.synchronize()
        assert_parity(out.cpu().numpy(), want[y0:y0 + rows])


def test_processor_outlives_its_handle_while_editors_exist():
    """Arc<GpuProcessor> (lib.rs:83): destroying the processor handle first must not pull the device tables from
    under an editor that still uses them (use-after-free for a C caller)."""
    import ctypes as C
    from rawphotoforge_amd import _native as N
    lib = N.lib()
    proc = C.c_void_p()
    assert lib.rpf_processor_create(0, C.byref(proc)) == 0
    h, w = 24, 40
    cfa = util.bayer_uniform(h, w, seed=3)
    rp = util.to_raw(util.raw_params()).to_c()
    ed = C.c_void_p()
    assert lib.rpf_editor_create_from_bayer_u16(proc, cfa.ctypes.data_as(C.POINTER(C.c_uint16)), w, h, C.byref(rp),
                                                C.byref(ed)) == 0
    lib.rpf_processor_destroy(proc)            # the caller's handle goes first
    assert lib.rpf_apply_adjustments(ed) == 0  # tables are still there
    out = np.zeros((h, w, 4), np.float32)
    assert lib.rpf_get_rgba_f32(ed, out.ctypes.data_as(C.POINTER(C.c_float))) == 0
    assert_parity(out, O.develop(cfa, util.raw_params(), [O.EditParameters()]))
    lib.rpf_editor_destroy(ed)                 # last reference: frees the processor


def test_dev_prefetch_is_value_neutral(gpu):
    """rpf_dev_prefetch only reads: any size / alignment (whole 16-byte words inside the range), nothing changes."""
    torch = pytest.importorskip("torch")
    buf = torch.arange(1 << 20, dtype=torch.int32, device="cuda")
    before = buf.clone()
    s = torch.cuda.current_stream().cuda_stream
    for off, nbytes in ((0, buf.numel() * 4), (0, 0), (4, 1000), (12, 15), (1, 17), (3, (1 << 22) - 7)):
        R.dev_prefetch(gpu, buf.data_ptr() + off, nbytes, s)
    torch.cuda.synchronize()
    assert torch.equal(buf, before)
    with pytest.raises(R.InvalidArgument):
        R.dev_prefetch(gpu, 0, 64, s)


def test_empty_and_invalid_inputs(gpu):
    with pytest.raises(R.FailedToReadImage):
        R.PhotoEditor.new_from_rgb_f32(gpu, np.zeros(0, np.float32), 0, 0)
    with pytest.raises(R.FailedToReadImage):
        R.PhotoEditor.new_from_bayer_u16(gpu, np.zeros((2, 2), np.uint16), util.to_raw(util.raw_params()))
    with pytest.raises(R.FailedToReadImage):   # white level must exceed black level
        R.PhotoEditor.new_from_bayer_u16(gpu, np.zeros((8, 8), np.uint16), R.RawParams(black=(512,) * 4, white=512))
    with pytest.raises(R.InvalidArgument):
        R.PhotoEditor.for_tile(gpu, 64, 64, 60, 8)      # tile leaves its frame
    ed = R.PhotoEditor.for_tile(gpu, 64, 64, 0, 64)
    with pytest.raises(R.GpuComputeError):
        ed.apply_adjustments()                          # a tile editor owns no image
    ed.set_lens_distortion_correction(10)
    with pytest.raises(R.InvalidArgument):
        ed.dev_develop(1, util.to_raw(util.raw_params()), 1)   # resampling cannot run on the fused tile path
    one = R.PhotoEditor.new_from_rgb_f32(gpu, np.float32([0.25, 0.5, 0.75]), 1, 1)   # 1x1 image
    one.apply_adjustments()
    want = O.apply(np.float32([[[0.25, 0.5, 0.75, 1.0]]]), [O.EditParameters()])
    assert_parity(one.rgba(), want)


def test_maximum_size_12000x8000(gpu):
    """BASELINE configs[3] frame on one GPU: bands with halos == whole frame, oracle on sampled bands."""
    torch = pytest.importorskip("torch")
    h, w = 8000, 12000
    cfa = util.bayer_uniform(h, w, seed=4)
    rp = util.raw_params()
    p = util.params_S()
    s = torch.cuda.current_stream().cuda_stream
    pad = np.zeros((h + 4, w), np.uint16)
    pad[2:-2] = cfa
    d = torch.from_numpy(pad.view(np.int16)).cuda()
    ed = R.PhotoEditor.for_tile(gpu, w, h, 0, h)
    util.configure_editor(ed, p)
    whole = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    ed.dev_develop(d.data_ptr(), util.to_raw(rp), whole.data_ptr(), R.OUT_RGBA_F32, s)
    torch.cuda.synchronize()
    from rawphotoforge_amd import tiles
    for y0, rows in tiles.split_rows(h, 8):
        eb = R.PhotoEditor.for_tile(gpu, w, h, y0, rows)
        util.configure_editor(eb, p)
        ob = torch.empty((rows, w, 4), dtype=torch.float32, device="cuda")
        eb.dev_develop(d[y0:y0 + rows + 4].data_ptr(), util.to_raw(rp), ob.data_ptr(), R.OUT_RGBA_F32, s)
        torch.cuda.synchronize()
        assert torch.equal(ob, whole[y0:y0 + rows])
    for y0 in (0, 3998, h - 4):
        assert_parity(whole[y0:y0 + 4].cpu().numpy(), O.develop(cfa, rp, [p], y0=y0, rows=4))


def test_random_parameter_space_on_gpu(gpu):
    """Random sliders / curves / masks (every combination of the identity- and constant-curve fast paths),
    Bayer and RGB sources, vignette on and off: bit-exact against the oracle."""
    rng = np.random.default_rng(777)
    h, w = 48, 88
    for it in range(24):
        n_masks = int(rng.integers(1, 4))
        params = [util.random_params(rng) for _ in range(n_masks)]
        params[0].vignette = 0 if rng.random() < 0.5 else int(rng.integers(-100, 101))
        masks = np.ones((n_masks, h, w), np.float32)
        raw_masks = []
        for m in range(1, n_masks):
            mm = rng.random((h, w)).astype(np.float32) - np.float32(0.3)
            raw_masks.append(mm)
            masks[m] = O.binarize_mask(mm)
        if it % 2 == 0:
            cfa = util.bayer_natural(h, w, seed=it)
            rp = O.RawParams(cfa_pattern=it % 4)
            ed = bayer_editor(gpu, cfa, rp)
            want = O.develop(cfa, rp, params, masks=masks)
        else:
            img = _rgb_image(h, w, seed=it)
            ed = R.PhotoEditor.new_from_rgb_f32(gpu, img, h, w)
            want = O.apply(np.concatenate([img, np.ones((h, w, 1), np.float32)], -1), params, masks=masks)
        for m in range(1, n_masks):
            ed.add_mask(f"m{m}", raw_masks[m - 1])
        for m, p in enumerate(params):
            util.configure_editor(ed, p, None if m == 0 else f"m{m}", main=(m == 0))
        ed.apply_adjustments()
        assert_parity(ed.rgba(), want)


def test_sixteen_masks(gpu):
    """RPF_FUSED_MASKS masks — the most the fused kernel takes: every mask's curves and its
    u8 plane are gathered from L2, only "main"'s brightness curve lives in LDS."""
    rng = np.random.default_rng(16)
    h, w = 40, 72
    n = R.FUSED_MASKS
    cfa = util.bayer_natural(h, w, seed=16)
    rp = util.raw_params()
    params = [util.random_params(rng, trivial_prob=0.3) for _ in range(n)]
    masks = np.ones((n, h, w), np.float32)
    ed = bayer_editor(gpu, cfa, rp)
    for m in range(1, n):
        mm = (rng.random((h, w)) > 0.7).astype(np.float32)
        masks[m] = mm
        ed.add_mask(f"m{m}", mm * 2.0 - 0.5)      # binarised at add_mask: >= 0 -> 1
        masks[m] = O.binarize_mask(mm * 2.0 - 0.5)
    for m, p in enumerate(params):
        util.configure_editor(ed, p, None if m == 0 else f"m{m}", main=(m == 0))
    ed.apply_adjustments()
    assert_parity(ed.rgba(), O.develop(cfa, rp, params, masks=masks))


@pytest.mark.parametrize("n", [17, 32, 33, 50])
def test_more_masks_than_the_fused_kernel_takes(gpu, n):
    """The reference has no limit on masks (lib.rs:481-499).  Past RPF_FUSED_MASKS the coverage lies in one 16-bit plane per
    16 masks and the parameters in device memory; a Bayer editor runs the demosaic-only kernel and the RGBA pass.  Counts
    on both sides of a plane boundary; masks that cover nothing, everything, and a pixel alone; removing masks brings the
    editor back under the boundary (the fused kernel again) with the slots moved."""
    rng = np.random.default_rng(1000 + n)
    h, w = 36, 70
    cfa = util.bayer_natural(h, w, seed=n)
    rp = util.raw_params()
    params = [util.random_params(rng, trivial_prob=0.4) for _ in range(n)]
    masks = np.ones((n, h, w), np.float32)
    ed = bayer_editor(gpu, cfa, rp)
    for m in range(1, n):
        if m == 20:
            mm = np.zeros((h, w), np.float32)                     # covers nothing
        elif m == 21:
            mm = np.ones((h, w), np.float32)                      # covers everything
        elif m == n - 1:
            mm = np.zeros((h, w), np.float32); mm[h // 2, w // 3] = 1.0   # the last mask: one pixel
        else:
            mm = (rng.random((h, w)) > 0.75).astype(np.float32)
        ed.add_mask(f"m{m}", mm)
        masks[m] = O.binarize_mask(mm)
    for m, p in enumerate(params):
        util.configure_editor(ed, p, None if m == 0 else f"m{m}", main=(m == 0))
    ed.apply_adjustments()
    assert_parity(ed.rgba(), O.develop(cfa, rp, params, masks=masks))
    # the other output layouts read the same frame
    want = O.develop(cfa, rp, params, masks=masks)
    np.testing.assert_array_equal(ed.to_flat_vec_rgb().reshape(h, w, 3), want[..., :3])
    np.testing.assert_array_equal(ed.to_rgb_array3_u8(), O.rgba_to_rgb_u8(want))
    # a setter on a mask of a further plane
    p_new = util.random_params(rng, trivial_prob=0.0)
    params[n - 2] = p_new
    util.configure_editor(ed, p_new, f"m{n - 2}", main=False)
    ed.apply_adjustments()
    assert_parity(ed.rgba(), O.develop(cfa, rp, params, masks=masks))
    # remove masks from the front until the editor fits the fused kernel again: every later slot moves down
    keep = [0] + list(range(n - (R.FUSED_MASKS - 1), n))
    for m in range(1, n):
        if m not in keep:
            ed.remove_mask(f"m{m}")
    assert ed.num_masks == R.FUSED_MASKS
    ed.apply_adjustments()
    assert_parity(ed.rgba(), O.develop(cfa, rp, [params[m] for m in keep], masks=masks[keep]))


def test_many_masks_with_their_parameters_in_device_memory(gpu):
    """Variant 24: the kernel editors with more than 1024 masks run (parameters read from device memory, not staged in LDS)."""
    rng = np.random.default_rng(2424)
    h, w, n = 30, 50, 21
    img = rng.random((h, w, 3), dtype=np.float32)
    rgba = np.concatenate([img, np.ones((h, w, 1), np.float32)], -1)
    params = [util.random_params(rng, trivial_prob=0.4) for _ in range(n)]
    masks = np.ones((n, h, w), np.float32)
    ed = R.PhotoEditor.new_from_rgb_f32(gpu, img, h, w)
    for m in range(1, n):
        mm = (rng.random((h, w)) > 0.7).astype(np.float32)
        ed.add_mask(f"m{m}", mm)
        masks[m] = O.binarize_mask(mm)
    for m, p in enumerate(params):
        util.configure_editor(ed, p, None if m == 0 else f"m{m}", main=(m == 0))
    gpu.set_kernel_variant(24)
    try:
        ed.apply_adjustments()
        got = ed.rgba()
    finally:
        gpu.set_kernel_variant(0)
    assert_parity(got, O.apply(rgba, params, masks=masks))


def test_many_masks_on_an_rgb_editor_with_lens_and_vignette(gpu):
    """40 masks through the reference's own pass (RGB input), with the lens resample and the vignette in front."""
    rng = np.random.default_rng(4040)
    h, w, n = 41, 67, 40
    img = rng.random((h, w, 3), dtype=np.float32)
    rgba = np.concatenate([img, np.ones((h, w, 1), np.float32)], -1)
    params = [util.random_params(rng, trivial_prob=0.4) for _ in range(n)]
    params[0].vignette = -40
    masks = np.ones((n, h, w), np.float32)
    ed = R.PhotoEditor.new_from_rgb_f32(gpu, img, h, w)
    for m in range(1, n):
        mm = (rng.random((h, w)) > 0.8).astype(np.float32)
        ed.add_mask(f"m{m}", mm)
        masks[m] = O.binarize_mask(mm)
    for lens in (0, 35):
        params[0].lens_distortion = lens
        for m, p in enumerate(params):
            util.configure_editor(ed, p, None if m == 0 else f"m{m}", main=(m == 0))
        ed.apply_adjustments()
        assert_parity(ed.rgba(), O.apply(rgba, params, masks=masks))


def test_tile_entry_point_with_many_masks(gpu):
    """rpf_dev_develop with more masks than the fused kernel takes: the demosaic-only kernel into the editor's linear tile, then the
    RGBA pass — and the caller's own two launches give the same bits; a packed band is refused with the remedy."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(77)
    h, w, n = 32, 64, 18
    cfa = util.bayer_natural(h, w, seed=77)
    rp = util.raw_params()
    params = [util.random_params(rng, trivial_prob=0.5) for _ in range(n)]
    masks = np.ones((n, h, w), np.float32)
    ed = R.PhotoEditor.for_tile(gpu, w, h, 0, h)
    for m in range(1, n):
        mm = (rng.random((h, w)) > 0.6).astype(np.float32)
        ed.add_mask(f"m{m}", mm)
        masks[m] = O.binarize_mask(mm)
    for m, p in enumerate(params):
        util.configure_editor(ed, p, None if m == 0 else f"m{m}", main=(m == 0))
    band = np.zeros((h + 4, w), np.uint16)   # rows -2 .. h + 1 (outside the frame: never read, the kernel reflects)
    band[2:-2] = cfa
    d_band = torch.from_numpy(band.view(np.int16)).cuda()
    d_out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    want = O.develop(cfa, rp, params, masks=masks)
    ed.dev_develop(d_band.data_ptr(), util.to_raw(rp), d_out.data_ptr(), R.OUT_RGBA_F32, s)
    torch.cuda.synchronize()
    assert_parity(d_out.cpu().numpy(), want)
    d_u8 = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
    ed.dev_develop(d_band.data_ptr(), util.to_raw(rp), d_u8.data_ptr(), R.OUT_RGB_U8, s)
    torch.cuda.synchronize()
    assert np.array_equal(d_u8.cpu().numpy(), O.rgba_to_rgb_u8(want))
    d_packed = torch.zeros((h + 4, (w * 12 + 7) // 8), dtype=torch.uint8, device="cuda")
    with pytest.raises(R.InvalidArgument, match="rpf_dev_unpack_bayer"):
        ed.dev_develop_packed(d_packed.data_ptr(), d_packed.shape[1], 12, util.to_raw(rp), d_out.data_ptr(), R.OUT_RGBA_F32, s)
    d_out.zero_()
    d_lin = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    R.dev_demosaic(gpu, d_band.data_ptr(), util.to_raw(rp), w, h, 0, h, d_lin.data_ptr(), s)
    ed.dev_apply(d_lin.data_ptr(), d_out.data_ptr(), R.OUT_RGBA_F32, s)
    torch.cuda.synchronize()
    assert_parity(d_out.cpu().numpy(), want)


def test_many_mask_band_launch_leaves_an_rgb_editors_image_alone(gpu):
    """An RGB editor's d_original is its IMAGE.  rpf_dev_develop with more than 16 masks needs a linear intermediate of the band: it
    must be a buffer of its own — the editor's next apply_adjustments still renders the photo it was given."""
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(808)
    h, w, n = 28, 60, 18
    img = rng.random((h, w, 3), dtype=np.float32)
    rgba = np.concatenate([img, np.ones((h, w, 1), np.float32)], -1)
    params = [util.random_params(rng, trivial_prob=0.5) for _ in range(n)]
    for p in params:
        p.lens_distortion = 0
    masks = np.ones((n, h, w), np.float32)
    ed = R.PhotoEditor.new_from_rgb_f32(gpu, img, h, w)
    for m in range(1, n):
        mm = (rng.random((h, w)) > 0.6).astype(np.float32)
        ed.add_mask(f"m{m}", mm)
        masks[m] = O.binarize_mask(mm)
    for m, p in enumerate(params):
        util.configure_editor(ed, p, None if m == 0 else f"m{m}", main=(m == 0))
    want = O.apply(rgba, params, masks=masks)
    ed.apply_adjustments()
    assert_parity(ed.rgba(), want)
    cfa = util.bayer_natural(h, w, seed=808)
    rp = util.raw_params()
    band = np.zeros((h + 4, w), np.uint16)
    band[2:-2] = cfa
    d_band = torch.from_numpy(band.view(np.int16)).cuda()
    d_out = torch.empty((h, w, 4), dtype=torch.float32, device="cuda")
    ed.dev_develop(d_band.data_ptr(), util.to_raw(rp), d_out.data_ptr(), R.OUT_RGBA_F32, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert_parity(d_out.cpu().numpy(), O.develop(cfa, rp, params, masks=masks))
    ed.set_vignette(0 if params[0].vignette else 10)   # any state change: the next apply really launches again
    ed.set_vignette(params[0].vignette)
    ed.apply_adjustments()
    assert_parity(ed.rgba(), want)


def test_editors_share_a_processor_and_interleave(gpu):
    """One GpuProcessor, several editors on their own streams (the reference's 3-editor preview pyramid,
    Arc<GpuProcessor>, lib.rs:83): interleaved setters / applies / readbacks stay independent."""
    rng = np.random.default_rng(99)
    eds, wants = [], []
    for i, (h, w) in enumerate(((40, 64), (33, 50), (64, 96))):
        img = rng.random((h, w, 3), dtype=np.float32)
        eds.append((R.PhotoEditor.new_from_rgb_f32(gpu, img, h, w), np.concatenate([img, np.ones((h, w, 1), np.float32)], -1)))
    for rep in range(5):
        params = [util.random_params(rng) for _ in eds]
        for (ed, _), p in zip(eds, params):
            util.configure_editor(ed, p)
            ed.apply_adjustments()            # enqueue all three before reading any back
        for (ed, rgba), p in zip(eds, params):
            assert_parity(ed.rgba(), O.apply(rgba, [p]))


def test_processor_shared_by_threads(gpu):
    """SURVEY 8(b) threading: an editor is externally synchronised, the processor is safe to share.  Four threads, each
    with its own editors (an RGB one and a Bayer one) on the ONE processor, create / configure / apply / read back
    concurrently (ctypes drops the GIL inside every call); every result is compared with the oracle, bit for bit, and
    an error on one thread (unknown mask name) leaves its message on that thread only."""
    import threading

    def worker(i, failures):
        try:
            rng = np.random.default_rng(1000 + i)
            h, w = 48 + 8 * i, 80 + 16 * i
            img = rng.random((h, w, 3), dtype=np.float32)
            rgba = np.concatenate([img, np.ones((h, w, 1), np.float32)], -1)
            cfa = rng.integers(0, 16384, (h, w), dtype=np.uint16)
            rp = O.RawParams(cfa_pattern=i % 4, black=(512,) * 4, white=16383, wb=(2.0, 1.0, 1.0, 1.5))
            for rep in range(6):
                ed_rgb = R.PhotoEditor.new_from_rgb_f32(gpu, img, h, w)
                ed_raw = bayer_editor(gpu, cfa, rp)
                p, q = util.random_params(rng), util.random_params(rng)
                util.configure_editor(ed_rgb, p)
                util.configure_editor(ed_raw, q)
                ed_rgb.apply_adjustments()
                ed_raw.apply_adjustments()
                with pytest.raises(R.MaskNotFound) as ei:
                    ed_rgb.set_whitebalance(1, 1, f"no such mask {i}")
                assert f"no such mask {i}" in str(ei.value)
                assert_parity(ed_rgb.rgba(), O.apply(rgba, [p]))
                assert_parity(ed_raw.rgba(), O.develop(cfa, rp, [q]))
        except BaseException as e:  # noqa: BLE001 - reported on the main thread
            failures.append((i, repr(e)))

    failures = []
    threads = [threading.Thread(target=worker, args=(i, failures)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not failures, failures


def test_incremental_setter_traffic(gpu):
    """Per-curve dirty tracking: after each small change (one curve, one slider, the same control points again, the
    same full curve again, a mask added / removed, reset) the next apply must equal the oracle on the full state."""
    h, w = 48, 96
    img = _rgb_image(h, w, seed=21)
    rgba = np.concatenate([img, np.ones((h, w, 1), np.float32)], -1)
    ed = R.PhotoEditor.new_from_rgb_f32(gpu, img, h, w)
    p = O.EditParameters()

    def check(masks_params=None, masks=None):
        ed.apply_adjustments()
        assert_parity(ed.to_flat_vec_rgba(), O.apply(rgba, masks_params or [p], masks=masks))

    check()
    xs, ys = [0, 20000, 45000, 65535], [0, 26000, 40000, 65535]
    ed.set_brightness_tone_curve(control_points_x=xs, control_points_y=ys)
    p.brightness_tone_curve = O.curve_from_control_points(xs, ys)
    check()
    ed.set_brightness_tone_curve(control_points_x=xs, control_points_y=ys)      # same points again: no work, same result
    check()
    ed.set_oklch_hue_curve(control_points_x=[0, 30000, 65535], control_points_y=[2000, 28000, 65535])
    p.hue_tone_curve = O.curve_from_control_points([0, 30000, 65535], [2000, 28000, 65535])
    check()
    ed.set_tone(0.4, 10, -20, 15, 0, 5)                                          # scalars only: no curve upload
    p.exposure, p.contrast, p.shadow, p.highlight, p.black, p.white = 0.4, 10, -20, 15, 0, 5
    check()
    sat = np.full(65536, 40000, np.int32)
    ed.set_oklch_saturation_curve(curve=sat)
    p.saturation_tone_curve = sat
    check()
    ed.set_oklch_saturation_curve(curve=sat)                                     # same 65536 values again
    check()
    ys2 = [0, 20000, 45000, 60000]
    ed.set_brightness_tone_curve(control_points_x=xs, control_points_y=ys2)     # same x, different y
    p.brightness_tone_curve = O.curve_from_control_points(xs, ys2)
    check()
    with pytest.raises(R.NotStrictlyIncreasing):                                 # a failed call changes nothing
        ed.set_brightness_tone_curve(control_points_x=[0, 5, 5], control_points_y=[0, 1, 2])
    check()
    rng = np.random.default_rng(3)
    m = rng.random((h, w), dtype=np.float32)
    ed.add_mask("sky", m)
    q = util.params_G()
    util.configure_editor(ed, q, mask_name="sky", main=False)
    masks = np.stack([np.ones((h, w), np.float32), O.binarize_mask(m, p.mask_range)])
    check([p, q], masks)
    ed.set_oklch_lightness_curve(control_points_x=[0, 65535], control_points_y=[30000, 36000], mask_name="sky")
    q.lightness_tone_curve = O.curve_from_control_points([0, 65535], [30000, 36000])
    check([p, q], masks)
    ed.remove_mask("sky")
    check()
    ed.reset()
    p = O.EditParameters()
    check()


@pytest.mark.parametrize("extra", [[], ["--workload", "batch", "--frames", "3"],
                                   ["--workload", "develop_packed", "--pack", "mipi"]])
def test_bench_prints_one_contract_line(gpu, capsys, monkeypatch, extra):
    """bench.py's output contract: ONE JSON line with the fields the driver and the judge read (the headline workload,
    and BASELINE configs[4]'s batch of different frames)."""
    import importlib
    import json
    import sys
    monkeypatch.setattr(sys, "argv", ["bench.py", "--steps", "6", "--warmup", "2", "--preheat", "4", "--cpu-seconds", "0.5", "--secondary-steps", "8"] + extra)
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    bench = importlib.import_module("bench")
    bench.main()
    lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    r = json.loads(lines[0])
    base = json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "BASELINE.json")))
    assert r["metric"] == base["metric"] and r["unit"] == "MP/s" and r["n_gpus"] == 1 and r["steps"] == 6 and r["warmup"] == 2
    batch = "batch" in extra
    assert r["higher_is_better"] is True and r["vs_baseline"] is None and r["data"] == "synthetic"
    # ONE frame however many GPUs (north_star) = strong scaling; the batch of different frames is replicas = weak
    assert r["scaling"] == ("weak" if batch else "strong") and r["config"]["mode"] == ("replicas" if batch else "strong")
    assert r["config"]["input_resident"] == ("hbm" if batch else "infinity_cache")
    assert r["config"]["world_size"] == 1 and len(r["config"]["ranks"]) == 1 and r["config"]["ranks"][0]["band"] == [0, 4000]
    assert r["value"] > 0 and abs(r["ms_per_step"] - 24.0 / r["value"] * 1e3) < 1e-2 * r["ms_per_step"]
    assert "workload" in r["config"] and "model" not in r["config"]
    rf = r["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert "traffic" in rf
    if rf["traffic"] is not None:   # replayed from the committed PMC pass of the same command, and labelled so
        assert rf["traffic_measured_in_this_run"] is False
    if not extra:   # the headline line also carries the kernel a user's edits dispatch and input that comes from HBM
        sec = r["secondary"]
        assert "pristine" in r["config"]["workload"] and "Infinity Cache" in r["config"]["workload"]
        for e in (sec["edit_set_S"], sec["fresh_input"]["with_prefetch"], sec["fresh_input"]["without_prefetch"]):
            assert e["kernel_ms_avg"] > 0 and 0 < e["frac"] < 1 and abs(e["frac"] - e["achieved_GB/s"] / 8000.0) < 1e-3
    else:
        assert "secondary" not in r
    cb = r["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "MP/s" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
