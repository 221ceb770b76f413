"""The C++ host facade (libmedseg.so: the reference's API names) against the oracle, the golden JSON bytes emitted by
the reference's own nlohmann header, and PIL for the PNG codec.  CPU only -- nothing here touches the engine."""
import json
import os
import subprocess

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st
from PIL import Image

import oracle_lib as orc
from miunet import hostlib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROBE = os.path.join(ROOT, "oracle", "_ref", "json_probe")


def test_libmedseg_exports_every_declared_symbol():
    import re
    src = open(os.path.join(ROOT, "include", "medseg_c.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    syms = sorted(set(re.findall(r"\b(medseg_[a-z0-9_]+)\s*\(", src)))
    L = hostlib.lib()
    for s in syms:
        assert hasattr(L, s), s
    assert sorted(hostlib.EXPORTS) == syms


# ---------------------------------------------------------------- preprocess
@pytest.mark.parametrize("h,w,seed", [(1536, 2048, 77), (200, 300, 78), (512, 512, 79), (700, 333, 80), (3, 5, 81)])
def test_resample_normalize_bit_exact_vs_oracle(h, w, seed):
    raw = synth.make_raw16(h, w, seed=seed)
    assert np.array_equal(hostlib.resample_normalize(raw), orc.preprocess_raw(raw))


def test_resample_normalize_edge_cases():
    for raw in (np.full((40, 30), 1234, np.uint16), np.full((8, 8), 65535, np.uint16), np.zeros((2, 2), np.uint16),
                np.array([[0, 65535]], np.uint16)):
        assert np.array_equal(hostlib.resample_normalize(raw), orc.preprocess_raw(raw))


def test_preprocess_raw_files(tmp_path, golden_dir):
    raw = synth.make_raw16(300, 400, seed=5)
    rp = tmp_path / "scan 01.raw"
    raw.tofile(rp)
    png, js = tmp_path / "out" / "scan 01_normalized.png", tmp_path / "scan 01_original_sizes.json"
    assert hostlib.preprocess_raw(str(rp), str(png), str(js), 400, 300)        # creates the missing parent dir (:121)
    assert js.read_bytes() == b'{"scan 01.raw":{"original_height":300,"original_width":400,"scaled_height":512,"scaled_width":512}}\n'
    assert np.array_equal(np.array(Image.open(png)), orc.preprocess_raw(raw))  # PIL decodes our level-0 PNG
    assert np.array_equal(hostlib.read_png(str(png)), orc.preprocess_raw(raw))
    # failure paths return false (src/preprocess.cpp:137-140): missing file, file shorter than w*h*2
    assert not hostlib.preprocess_raw(str(tmp_path / "nope.raw"), str(png), str(js), 400, 300)
    assert not hostlib.preprocess_raw(str(rp), str(png), str(js), 4000, 3000)


# ---------------------------------------------------------------- postprocess / LUT
def test_postprocess_matches_oracle_on_goldens(golden_dir):
    g = np.load(os.path.join(golden_dir, "imgproc.npz"))
    for i in range(5):
        assert np.array_equal(hostlib.postprocess_mask(g[f"mask{i}"]), g[f"final{i}"])


@settings(max_examples=60, deadline=None)
@given(st.integers(0, 2 ** 32 - 1), st.sampled_from([(24, 40), (64, 64), (50, 33)]), st.floats(0.2, 0.8))
def test_postprocess_random_blobs_vs_oracle(seed, shape, fill):
    rng = np.random.default_rng(seed)
    h, w = shape
    # smooth random field -> large blobs with holes, plus class-1 and speckle noise
    f = rng.random((h, w))
    for _ in range(3):
        f = (f + np.roll(f, 1, 0) + np.roll(f, -1, 0) + np.roll(f, 1, 1) + np.roll(f, -1, 1)) / 5
    m = np.where(f > np.quantile(f, 1 - fill), 2, 0).astype(np.uint8)
    m[rng.random((h, w)) < 0.03] = 1
    m[rng.random((h, w)) < 0.02] = 2
    assert np.array_equal(hostlib.postprocess_mask(m), orc.postprocess_mask(m))


def test_mask_to_image():
    m = np.arange(256, dtype=np.uint8).reshape(16, 16)
    assert np.array_equal(hostlib.mask_to_image(m), orc.mask_to_image(m))


# ---------------------------------------------------------------- contours
def _blob_mask(seed, h, w):
    rng = np.random.default_rng(seed)
    f = rng.random((h, w))
    for _ in range(2):
        f = (f + np.roll(f, 1, 0) + np.roll(f, -1, 0) + np.roll(f, 1, 1) + np.roll(f, -1, 1)) / 5
    m = (f > np.quantile(f, 0.55)).astype(np.uint8) * 255
    m[rng.random((h, w)) < 0.02] = 255
    m[rng.random((h, w)) < 0.02] = 0
    return m


@settings(max_examples=80, deadline=None)
@given(st.integers(0, 2 ** 32 - 1), st.sampled_from([(16, 16), (31, 47), (64, 64)]))
def test_extract_contours_random_vs_oracle(seed, shape):
    m = _blob_mask(seed, *shape)
    assert hostlib.extract_contours(m) == orc.find_contours(m)


def test_extract_contours_known_answers():
    m = np.zeros((12, 16), np.uint8); m[2:6, 3:9] = 255
    assert hostlib.extract_contours(m) == [[(3, 2), (3, 5), (8, 5), (8, 2)]]
    m[8, 1] = 255; m[10, 5:9] = 255
    assert hostlib.extract_contours(m) == [[(5, 10), (8, 10)], [(1, 8)], [(3, 2), (3, 5), (8, 5), (8, 2)]]
    full = np.full((512, 512), 255, np.uint8)
    assert hostlib.extract_contours(full) == [[(0, 0), (0, 511), (511, 511), (511, 0)]]
    assert hostlib.extract_contours(np.zeros((512, 512), np.uint8)) == []


def test_contours_are_closed_8_connected_loops_on_the_boundary():
    """size-independent properties at full size: every contour point is a foreground pixel with a background
    4/8-neighbour (or on the image edge), consecutive points are joined by horizontal, vertical or 45-degree runs."""
    m = _blob_mask(123, 512, 512)
    pad = np.pad(m > 127, 1)
    cs = hostlib.extract_contours(m)
    assert cs == orc.find_contours(m) and len(cs) > 3
    for c in cs:
        for k, (x, y) in enumerate(c):
            assert pad[y + 1, x + 1]
            assert not pad[y:y + 3, x:x + 3].all()
            if len(c) > 1:
                x2, y2 = c[(k + 1) % len(c)]
                dx, dy = abs(x2 - x), abs(y2 - y)
                assert dx == 0 or dy == 0 or dx == dy


def _polyline_pixels(c):
    """independent rasteriser: the closed polyline through the points of a contour as a set of pixels.  Consecutive points of
    a CHAIN_APPROX_SIMPLE contour are joined by horizontal, vertical or exact 45-degree runs, for which every 8-connected
    line algorithm (cv::line LINE_8 included) visits the same pixels: n = max(|dx|, |dy|) unit steps."""
    px = set()
    for k, (x0, y0) in enumerate(c):
        x1, y1 = c[(k + 1) % len(c)]
        n = max(abs(x1 - x0), abs(y1 - y0))
        assert abs(x1 - x0) in (0, n) and abs(y1 - y0) in (0, n)
        sx, sy = (x1 > x0) - (x1 < x0), (y1 > y0) - (y1 < y0)
        for t in range(n + 1):
            px.add((x0 + t * sx, y0 + t * sy))
    return px


def _adversarial_masks():
    """shapes that stress a border follower: 1-pixel spurs, diagonal-only links, nested holes, blobs on the frame"""
    out = []
    m = np.zeros((24, 24), np.uint8); m[4:12, 4:12] = 255; m[7, 12:18] = 255; m[12:16, 6] = 255; m[1, 1] = 255
    out.append(m)                                                        # spurs (traversed out and back) + an isolated pixel
    m = np.zeros((20, 20), np.uint8)
    for k in range(12):
        m[3 + k, 2 + k] = 255                                            # a pure diagonal chain
    m[15:18, 15:18] = 255; m[14, 14] = 255                               # ... linked to a block only through a corner
    out.append(m)
    m = np.zeros((40, 40), np.uint8); m[0:40, 0:40] = 255; m[5:35, 5:35] = 0; m[10:30, 10:30] = 255; m[15:25, 15:25] = 0; m[18:22, 18:22] = 255
    out.append(m)                                                        # nested rings touching the frame: only the outermost border is external
    m = np.zeros((16, 30), np.uint8); m[0, :] = 255; m[:, 0] = 255; m[15, 10:20] = 255; m[5:9, 29] = 255
    out.append(m)                                                        # 1-pixel-wide strokes lying on the frame
    m = np.zeros((18, 18), np.uint8); m[2::2, 2:16] = 255; m[2:16, 2] = 255
    out.append(m)                                                        # a comb: many parallel spurs off one spine
    m = (np.indices((17, 17)).sum(0) % 2 * 255).astype(np.uint8)
    out.append(m)                                                        # checkerboard: one 8-connected component, all links diagonal
    return out


def test_overlay_is_exactly_the_closed_polylines_of_the_contours():
    """A14 (src/mask2polygon.cpp:114-129): every pixel of the overlay is either the grey tile replicated to B,G,R or pure
    red, and the red set is EXACTLY the closed 8-connected polylines through the contour points -- checked against an
    independent rasteriser, including the wrap-around segment, the two-point out-and-back contour and the single point.
    For a SIMPLE-compressed border that polyline is the border itself: every red pixel is a foreground pixel with a
    background 4-neighbour (or on the frame)."""
    rng = np.random.default_rng(5)
    masks = _adversarial_masks() + [_blob_mask(s, 48, 64) for s in (1, 2, 3)] + [_blob_mask(9, 512, 512)]
    for m in masks:
        gray = rng.integers(0, 255, m.shape, dtype=np.uint8)             # never 255: red (0,0,255) cannot occur in the grey picture
        cs = hostlib.extract_contours(m)
        ov = hostlib.draw_overlay(gray, cs)
        red = (ov[..., 0] == 0) & (ov[..., 1] == 0) & (ov[..., 2] == 255)          # B, G, R
        want = set().union(*[_polyline_pixels(c) for c in cs]) if cs else set()
        assert {(int(x), int(y)) for y, x in zip(*np.nonzero(red))} == want
        assert np.array_equal(ov[~red], np.repeat(gray[~red][:, None], 3, axis=1))
        pad = np.pad(m > 127, 1)
        for (x, y) in want:
            assert pad[y + 1, x + 1]
            assert not (pad[y, x + 1] and pad[y + 2, x + 1] and pad[y + 1, x] and pad[y + 1, x + 2])
    # hand-written contours: a single point, an out-and-back pair, a triangle with two diagonals
    gray = np.full((10, 10), 7, np.uint8)
    ov = hostlib.draw_overlay(gray, [[(2, 3)], [(5, 1), (8, 1)], [(1, 6), (3, 8), (5, 6)]])
    red = (ov[..., 2] == 255) & (ov[..., 0] == 0)
    assert {(int(x), int(y)) for y, x in zip(*np.nonzero(red))} == \
        {(2, 3)} | {(x, 1) for x in range(5, 9)} | {(1, 6), (2, 7), (3, 8), (4, 7), (5, 6)} | {(x, 6) for x in range(1, 6)}


def test_contour_implementations_agree_on_adversarial_masks():
    """the oracle's Suzuki-Abe restatement and the host BorderTracer are two independently written followers; they must
    agree point for point (start pixel, direction, SIMPLE compression, newest-first order) on shapes built to break one"""
    for m in _adversarial_masks():
        a, b = hostlib.extract_contours(m), orc.find_contours(m)
        assert a == b and len(a) >= 1
    # known answers on two of them
    m = _adversarial_masks()[2]                                          # nested rings: ONE external contour, the frame rectangle
    assert hostlib.extract_contours(m) == [[(0, 0), (0, 39), (39, 39), (39, 0)]]
    m = _adversarial_masks()[5]                                          # checkerboard: ONE component through diagonal links only
    assert len(hostlib.extract_contours(m)) == 1


@settings(max_examples=120, deadline=None)
@given(st.integers(0, 2 ** 32 - 1), st.sampled_from([(9, 9), (12, 20), (24, 24)]), st.floats(0.2, 0.8))
def test_contours_random_dense_noise_vs_oracle(seed, shape, density):
    """unsmoothed noise: diagonal-only links, 1-pixel components and spurs everywhere, frame contacts on every side"""
    rng = np.random.default_rng(seed)
    m = (rng.random(shape) < density).astype(np.uint8) * 255
    a = hostlib.extract_contours(m)
    assert a == orc.find_contours(m)
    if a:
        gray = np.zeros(shape, np.uint8)
        red = hostlib.draw_overlay(gray, a)[..., 2] == 255
        assert {(int(x), int(y)) for y, x in zip(*np.nonzero(red))} == set().union(*[_polyline_pixels(c) for c in a])


def test_map_points():
    pts = [(511, 511), (3, 7), (0, 0), (255, 100)]
    for sx, sy in [(4.0, 3.0), (300 / 512.0, 200 / 512.0), (1.0, 1.0), (2047 / 512.0, 1.37)]:
        assert hostlib.map_points(pts, sx, sy) == orc.map_points(pts, sx, sy)


# ---------------------------------------------------------------- JSON bytes (pinned to the reference's nlohmann header)
def _cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "json", "cases.json")))


def test_polygon_json_bytes_match_reference_nlohmann(golden_dir, tmp_path):
    for c in _cases(golden_dir)["poly"]:
        out = tmp_path / (c["case"] + ".json")
        hostlib.generate_json([[tuple(p) for p in cc] for cc in c["contours"]], str(out), c["base_name"],
                              c["original_width"], c["original_height"])
        assert out.read_bytes() == open(os.path.join(golden_dir, "json", c["case"] + ".json"), "rb").read(), c["case"]


def test_size_json_bytes_match_reference_nlohmann(golden_dir, tmp_path):
    for c in _cases(golden_dir)["size"]:
        raw = np.zeros((2, 2), np.uint16)
        d = tmp_path / c["case"]
        d.mkdir()
        rp = d / c["raw_filename"]
        try:
            raw.tofile(rp)
        except OSError:
            pytest.skip("file name not representable on this filesystem")
        js = d / "sizes.json"
        # w*h*2 must not exceed the file: shrink the RAW claim, the JSON records whatever w,h the caller passed
        assert hostlib.preprocess_raw(str(rp), str(d / "n.png"), str(js), 2, 2)
        want = open(os.path.join(golden_dir, "json", c["case"] + ".json"), "rb").read()
        want = want.replace(f'"original_height":{c["h"]}'.encode(), b'"original_height":2').replace(
            f'"original_width":{c["w"]}'.encode(), b'"original_width":2')
        assert js.read_bytes() == want, c["case"]


@pytest.mark.skipif(not os.path.exists(PROBE), reason="oracle/_ref/json_probe only exists where /root/reference does")
@settings(max_examples=25, deadline=None)
@given(st.lists(st.lists(st.tuples(st.integers(-9999, 99999), st.integers(-9999, 99999)), min_size=1, max_size=6), min_size=1, max_size=4),
       st.text(alphabet=st.characters(blacklist_categories=("Cs",), blacklist_characters="\x00/"), min_size=1, max_size=8))
def test_polygon_json_live_against_probe(tmp_path_factory, contours, base):
    out = tmp_path_factory.mktemp("j") / "o.json"
    hostlib.generate_json(contours, str(out), base, 1234, 987)
    stdin = f"{len(contours)}\n" + "".join(f"{len(c)} " + " ".join(f"{x} {y}" for x, y in c) + "\n" for c in contours)
    want = subprocess.run([PROBE, "poly", base, "1234", "987"], input=stdin.encode(), check=True, capture_output=True).stdout
    assert out.read_bytes() == want


# ---------------------------------------------------------------- PNG codec vs PIL, and process_single_mask end to end
def test_png_roundtrip_against_pil(tmp_path):
    rng = np.random.default_rng(3)
    g = rng.integers(0, 256, (37, 53), dtype=np.uint8)
    assert hostlib.write_png(str(tmp_path / "g0.png"), g, level0=True)
    assert hostlib.write_png(str(tmp_path / "g9.png"), g, level0=False)
    for n in ("g0.png", "g9.png"):
        assert np.array_equal(np.array(Image.open(tmp_path / n)), g)
        assert np.array_equal(hostlib.read_png(str(tmp_path / n)), g)
    assert os.path.getsize(tmp_path / "g0.png") > g.size                 # stored, not deflated
    Image.fromarray(g).save(tmp_path / "pil.png", optimize=True)          # PIL picks adaptive filters: exercises all five
    assert np.array_equal(hostlib.read_png(str(tmp_path / "pil.png")), g)
    grad = (np.add.outer(np.arange(64), np.arange(64)) % 256).astype(np.uint8)
    Image.fromarray(grad).save(tmp_path / "grad.png")
    assert np.array_equal(hostlib.read_png(str(tmp_path / "grad.png")), grad)
    col = hostlib.read_png(str(tmp_path / "pil.png"), as_color=True)      # default cv::imread: gray replicated into B,G,R
    assert col.shape == (37, 53, 3) and all(np.array_equal(col[..., k], g) for k in range(3))
    bgr = rng.integers(0, 256, (9, 11, 3), dtype=np.uint8)
    assert hostlib.write_png(str(tmp_path / "c.png"), bgr, level0=False)
    assert np.array_equal(np.array(Image.open(tmp_path / "c.png")), bgr[..., ::-1])
    assert hostlib.read_png(str(tmp_path / "missing.png")) is None


def test_process_single_mask_end_to_end(tmp_path, capfd):
    """src/mask2polygon.cpp:134-222: size lookup, size check, contours, overlay with UN-mapped points, mapped JSON."""
    raw = synth.make_raw16(1536, 2048, seed=9)
    rp = tmp_path / "img7.raw"
    raw.tofile(rp)
    norm, sizes = tmp_path / "img7_normalized.png", tmp_path / "img7_original_sizes.json"
    assert hostlib.preprocess_raw(str(rp), str(norm), str(sizes), 2048, 1536)
    mask = np.zeros((512, 512), np.uint8)
    mask[100:300, 120:400] = 2
    mask[350:360, 10:30] = 2
    vis = hostlib.mask_to_image(mask)
    mp = tmp_path / "img7_mask.png"
    assert hostlib.write_png(str(mp), vis)
    hostlib.process_single_mask(str(mp), str(tmp_path), str(sizes), str(norm), "img7")
    doc = json.load(open(tmp_path / "img7.json"))
    assert doc["imagePath"] == "img7.raw" and doc["imageWidth"] == 2048 and doc["imageHeight"] == 1536
    pts = [[tuple(p) for p in s["points"]] for s in doc["shapes"]]
    want = [orc.map_points(c, 2048 / 512.0, 1536 / 512.0) for c in orc.find_contours(vis)]
    assert pts == want and len(pts) == 2
    assert pts[1] == [(480, 300), (480, 897), (1596, 897), (1596, 300)]
    ov = np.array(Image.open(tmp_path / "img7_contour_overlay.png"))
    base = orc.preprocess_raw(raw)
    red = (ov[..., 0] == 255) & (ov[..., 1] == 0) & (ov[..., 2] == 0)
    assert red[100, 120:400].all() and red[299, 120:400].all() and red[100:300, 120].all() and red[100:300, 399].all()
    assert red.sum() == 2 * (280 + 200) - 4 + 2 * (20 + 10) - 4          # two closed rectangles, thickness 1
    assert np.array_equal(ov[~red][:, 0], base[~red])                    # untouched pixels keep the gray value
    # no contours -> neither JSON nor overlay, only a warning (:183-186)
    d2 = tmp_path / "empty"
    d2.mkdir()
    assert hostlib.write_png(str(d2 / "img7_mask.png"), np.zeros((512, 512), np.uint8))
    hostlib.process_single_mask(str(d2 / "img7_mask.png"), str(d2), str(sizes), str(norm), "img7")
    assert not (d2 / "img7.json").exists() and not (d2 / "img7_contour_overlay.png").exists()
    assert "Warning: No Contours Detected" in capfd.readouterr().out
    # wrong mask size -> swallowed failure (:172-179, :219-221)
    assert hostlib.write_png(str(d2 / "small_mask.png"), np.zeros((64, 64), np.uint8))
    hostlib.process_single_mask(str(d2 / "small_mask.png"), str(d2), str(sizes), str(norm), "img7")
    assert "Mask size mismatch: 64x64 (actual) vs 512x512 (JSON)" in capfd.readouterr().err
    hostlib.process_single_mask(str(mp), str(d2), str(sizes), str(norm), "other")
    assert "Cannot Find Size Info in JSON: other.raw/.tif" in capfd.readouterr().err
