"""Size-independent properties at BASELINE.json's full sizes -- checks that need no oracle run, so they cover the sizes and batch
counts the oracle takes minutes for (512^2 x 16, 1024^2 x 3 x 8, 2048 x 1536 and larger RAW images):

  * the network is a per-image function: permuting the batch permutes the label maps and logits, bit for bit, in every plan;
  * min/max normalisation is invariant under an exact rescaling of the RAW values (x2 is exact in binary floating point, so
    Preprocess::preprocess_raw's (v - mn) * (255 / (mx - mn)), src/preprocess.cpp:92-116, gives the same byte);
  * postprocess_mask (src/postprocess.cpp:47-79) commutes with the symmetries of the square: 8-connectivity, the 3x3 structuring
    element, the border rule and the area thresholds are all symmetric, so flipping or transposing the input flips or
    transposes the output;
  * extract_contours (src/mask2polygon.cpp:29-36): every point is a foreground pixel with a background 4-neighbour (or on the
    frame), every contour starts at its component's first pixel in raster order, contours come newest (bottom-most) first, and
    the number of contours is the number of 8-connected components that are not enclosed by another one."""
import numpy as np
import pytest

from miunet import binding, synth
from miunet.spec import UNetSpec, pack_weights

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("algo,spec,size,batch", [("auto", UNetSpec(), 512, 16), ("bf16", UNetSpec(), 512, 16),
                                                  ("fp16", UNetSpec(in_ch=3, base=32, levels=5), 1024, 8)])
def test_batch_permutation_permutes_the_results(algo, spec, size, batch):
    blob = pack_weights(spec, synth.make_weights(spec, 99))
    imgs = synth.make_images(batch, size, size, spec.in_ch, 0xABCD, "blobs")
    perm = np.random.default_rng(5).permutation(batch)
    with binding.Engine(size, size, in_ch=spec.in_ch, base=spec.base, levels=spec.levels, max_batch=batch, conv_algo=algo) as eng:
        eng.load_weights(blob)
        lab, lg = eng.infer(imgs, want_logits=True)
        lab_p, lg_p = eng.infer(imgs[perm], want_logits=True)
    assert np.array_equal(lab_p, lab[perm]) and np.array_equal(lg_p, lg[perm])
    assert len(np.unique(lab)) >= 2                       # not a degenerate all-one-class output


@pytest.mark.parametrize("h,w", [(1536, 2048), (3000, 4096)])
def test_preprocess_is_invariant_under_exact_rescaling(h, w):
    raw = synth.make_raw16(h, w, seed=h + w, lo=10, hi=16000)          # x2 and x4 stay inside u16
    with binding.Engine(512, 512, max_batch=3) as eng:
        eng.load_weights(pack_weights(UNetSpec(), synth.make_weights(UNetSpec(), 1)))
        tiles, _, _ = eng.infer_raw16([raw, (raw * 2).astype(np.uint16), (raw * 4).astype(np.uint16)])
    assert np.array_equal(tiles[0], tiles[1]) and np.array_equal(tiles[0], tiles[2])
    assert tiles[0].min() == 0 and tiles[0].max() == 255


def _masks(n, size, seed):
    """label maps with blobs of class 2 around the 6 % area threshold, holes of both kinds, class-1 islands and speckle"""
    r = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size]
    out = np.zeros((n, size, size), np.uint8)
    for i in range(n):
        m = out[i]
        for _ in range(3):
            cy, cx = r.integers(size // 5, 4 * size // 5, 2)
            ry, rx = r.integers(size // 10, size // 3, 2)
            m[((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1.0] = 2
        for _ in range(4):                                 # holes: some below, some above 6 % of the image
            cy, cx = r.integers(size // 4, 3 * size // 4, 2)
            rr = r.integers(size // 40, size // 6)
            m[(yy - cy) ** 2 + (xx - cx) ** 2 <= rr * rr] = r.integers(0, 2)
        speck = r.random((size, size)) < 0.002
        m[speck] = 2 - m[speck] // 2 * 2
    return out


@pytest.mark.parametrize("size,n", [(512, 16), (1024, 4)])
def test_postprocess_commutes_with_the_symmetries_of_the_square(size, n):
    masks = _masks(n, size, 17 + size)
    with binding.Engine(size, size, 1, 16, 1, 3, max_batch=n) as eng:
        base = eng.postprocess_masks(masks)
        assert set(np.unique(base)) <= {0, 2} and (base == 2).any()
        for name, f in (("flip x", lambda a: a[:, :, ::-1]), ("flip y", lambda a: a[:, ::-1, :]),
                        ("transpose", lambda a: a.transpose(0, 2, 1)), ("rot180", lambda a: a[:, ::-1, ::-1])):
            got = eng.postprocess_masks(np.ascontiguousarray(f(masks)))
            assert np.array_equal(got, f(base)), name


def _components8(fg):
    from scipy import ndimage
    return ndimage.label(fg, structure=np.ones((3, 3), int))


@pytest.mark.parametrize("size,n", [(512, 16), (1024, 4)])
def test_contour_invariants_at_full_size(size, n):
    from scipy import ndimage
    masks = _masks(n, size, 3 + size)
    with binding.Engine(size, size, 1, 16, 1, 3, max_batch=n) as eng:
        vis = np.where(eng.postprocess_masks(masks) == 2, 255, 0).astype(np.uint8)
        vis[0, : size // 2, : size // 2] |= np.where(masks[0, : size // 2, : size // 2] == 2, 255, 0).astype(np.uint8)   # one unfiltered quadrant: many contours
        conts = eng.extract_contours(vis, cap_points=1 << 18, cap_contours=1 << 12)
    for i in range(n):
        fg = vis[i] > 127
        lab, ncomp = _components8(fg)
        # components not enclosed by another one: those whose pixels touch the background region connected to the frame
        outer_bg = ndimage.label(~np.pad(fg, 1))[0]
        outside = outer_bg == outer_bg[0, 0]
        near = ndimage.binary_dilation(outside, structure=np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], bool))[1:-1, 1:-1]
        external = set(np.unique(lab[near & fg])) - {0}
        cs = conts[i]
        assert cs is not None and len(cs) == len(external), (i, len(cs), len(external))
        starts = []
        for c in cs:
            pts = np.array(c)
            assert fg[pts[:, 1], pts[:, 0]].all()                                   # on foreground ...
            pad = np.pad(fg, 1)
            y, x = pts[:, 1] + 1, pts[:, 0] + 1
            assert (~pad[y - 1, x] | ~pad[y + 1, x] | ~pad[y, x - 1] | ~pad[y, x + 1]).all()     # ... with a background 4-neighbour
            comp = lab[pts[0, 1], pts[0, 0]]
            assert comp in external and (lab[pts[:, 1], pts[:, 0]] == comp).all()   # one component per contour
            ys, xs = np.nonzero(lab == comp)
            assert (pts[0, 1], pts[0, 0]) == (ys[0], xs[0])                         # starts at the component's first pixel in raster order
            starts.append(pts[0, 1] * size + pts[0, 0])
        assert starts == sorted(starts, reverse=True)                               # newest (bottom-most) first
        assert len({lab[c[0][1], c[0][0]] for c in cs}) == len(cs)
