"""Layer-by-layer parity of the HIP kernels against the oracle, through the C-ABI (mi_unet_layer_debug)."""
import numpy as np
import pytest

import oracle_lib as orc
from miunet import binding

pytestmark = pytest.mark.gpu


def _rng(seed):
    return np.random.default_rng(seed)


def _tol(ref):
    # fp32 sums of K products in a different order: |err| <= ~K * eps * |a||b|; 1e-4 relative to the output scale is ample
    return 1e-4 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("B,H,W,Cin,Cout", [
    (1, 8, 32, 16, 64),        # one exact tile
    (2, 16, 64, 64, 64),       # several tiles, several chunks
    (1, 5, 7, 32, 64),         # ragged: partial tile in x and y
    (1, 12, 40, 48, 128),      # two n-tiles, partial x tile, Cin % 16 == 0
    (1, 9, 33, 24, 32),        # Cin % 16 != 0 (masked last chunk), Cout < tile (masked columns)
    (3, 4, 4, 128, 256),       # deep-layer shape
    (1, 2, 2, 1024, 64),       # bottleneck of a 32x32 input: K = 9216
])
@pytest.mark.parametrize("op", ["conv3x3", "conv3x3_wino", "conv3x3_wino16"])
def test_conv3x3_mfma(B, H, W, Cin, Cout, op):
    r = _rng(B * 1000 + H * 100 + W + Cin + Cout)
    x = r.standard_normal((B, H, W, Cin), dtype=np.float32)
    w = (r.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    scale = (1.0 + 0.1 * r.standard_normal(Cout)).astype(np.float32)
    shift = (0.1 * r.standard_normal(Cout)).astype(np.float32)
    got = binding.layer_debug(op, x, w, scale, shift, relu=True)
    ref = orc.conv3x3(x, w) * scale + shift
    ref = np.maximum(ref, 0.0)
    assert not np.isnan(got).any(), "unwritten (NaN-poisoned) outputs"
    assert np.max(np.abs(got - ref)) < _tol(ref)


def test_conv3x3_no_relu_keeps_negatives():
    r = _rng(5)
    x = r.standard_normal((1, 8, 32, 16), dtype=np.float32)
    w = r.standard_normal((64, 16, 3, 3), dtype=np.float32) * 0.1
    got = binding.layer_debug("conv3x3", x, w, None, None, relu=False)
    ref = orc.conv3x3(x, w.astype(np.float32))
    assert (got < 0).any()
    assert np.max(np.abs(got - ref)) < _tol(ref)


def test_conv3x3_exact_integers_and_asymmetric_taps():
    # small-integer data: every product and partial sum is exact in fp32, so any summation order gives the same bits;
    # a delta weight on ONE tap/channel checks tap orientation (dy,dx not swapped or mirrored) and channel order.
    r = _rng(11)
    B, H, W, Cin, Cout = 1, 10, 36, 32, 64
    x = r.integers(-4, 5, (B, H, W, Cin)).astype(np.float32)
    w = r.integers(-3, 4, (Cout, Cin, 3, 3)).astype(np.float32)
    got = binding.layer_debug("conv3x3", x, w)
    assert np.array_equal(got, orc.conv3x3(x, w))
    for (ky, kx, ci, co) in [(0, 2, 5, 7), (2, 0, 31, 63), (1, 1, 0, 0), (0, 0, 17, 33)]:
        w = np.zeros((Cout, Cin, 3, 3), np.float32)
        w[co, ci, ky, kx] = 1.0
        got = binding.layer_debug("conv3x3", x, w)
        want = np.zeros((B, H, W), np.float32)
        ys, xs = np.arange(H)[:, None] + ky - 1, np.arange(W)[None, :] + kx - 1
        ok = (ys >= 0) & (ys < H) & (xs >= 0) & (xs < W)
        want[0][ok] = x[0, np.clip(ys, 0, H - 1), np.clip(xs, 0, W - 1), ci][ok]
        assert np.array_equal(got[..., co], want)
        assert np.count_nonzero(got) == np.count_nonzero(want)


@pytest.mark.parametrize("op", ["conv3x3_wino", "conv3x3_wino16"])
def test_conv3x3_wino_exact_on_even_integers_and_tap_orientation(op):
    # Winograd's G has halves: with weights that are multiples of 4 and small-integer inputs every intermediate is an
    # exactly representable integer, so the result must equal the direct sum bit for bit; single-tap weights check the
    # orientation of the transforms (a transposed G or B would mirror or swap taps).
    r = _rng(12)
    B, H, W, Cin, Cout = 2, 11, 19, 24, 64
    x = r.integers(-4, 5, (B, H, W, Cin)).astype(np.float32)
    w = (4 * r.integers(-3, 4, (Cout, Cin, 3, 3))).astype(np.float32)
    got = binding.layer_debug(op, x, w)
    assert np.array_equal(got, orc.conv3x3(x, w))
    for (ky, kx, ci, co) in [(0, 2, 5, 7), (2, 0, 23, 63), (1, 1, 0, 0), (0, 0, 17, 33), (2, 2, 9, 40), (1, 0, 3, 3)]:
        w = np.zeros((Cout, Cin, 3, 3), np.float32)
        w[co, ci, ky, kx] = 4.0
        got = binding.layer_debug(op, x, w)
        assert np.array_equal(got, orc.conv3x3(x, w))
        assert np.count_nonzero(got[..., [c for c in range(Cout) if c != co]]) == 0


@pytest.mark.parametrize("B,H,W,Cin,Cout", [
    (1, 16, 16, 16, 128),      # one exact 16x16 block, one chunk
    (2, 32, 48, 64, 128),      # several blocks, several chunks
    (1, 5, 7, 32, 128),        # ragged: partial block in x and y
    (1, 20, 40, 48, 256),      # two n-tiles, partial blocks, odd chunk count
    (1, 9, 33, 24, 96),        # Cin % 16 != 0 (masked last chunk), Cout < 128 (masked columns)
    (3, 4, 4, 128, 256),       # deep-layer shape
    (1, 2, 2, 1024, 128),      # K = 9216 * 4
    (1, 18, 18, 16, 128),      # a 2-pixel rim past the block boundary
    (2, 32, 32, 64, 64),       # Cout = 64: the one-block-per-wave variant
    (1, 21, 35, 128, 64),      # ... ragged
    (1, 16, 16, 32, 192),      # Cout % 128 == 64: one-block variant, three n-tiles
    (1, 16, 16, 32, 48),       # Cout < 64 (masked columns)
])
def test_conv3x3_wino4(B, H, W, Cin, Cout):
    # F(4x4,3x3): transforms with coefficients up to 8 and 1/24 -- still fp32 rounding noise at this scale
    r = _rng(B * 1000 + H * 100 + W + Cin + Cout)
    x = r.standard_normal((B, H, W, Cin), dtype=np.float32)
    w = (r.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    scale = (1.0 + 0.1 * r.standard_normal(Cout)).astype(np.float32)
    shift = (0.1 * r.standard_normal(Cout)).astype(np.float32)
    got = binding.layer_debug("conv3x3_wino4", x, w, scale, shift, relu=True)
    ref = np.maximum(orc.conv3x3(x, w) * scale + shift, 0.0)
    assert not np.isnan(got).any(), "unwritten (NaN-poisoned) outputs"
    assert np.max(np.abs(got - ref)) < _tol(ref)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [
    (2, 32, 32, 64, 64),       # the 64 -> 64 layers (inc.c2, up4.c2)
    (1, 21, 35, 128, 64),      # up4.c1's channels, ragged blocks
    (1, 16, 16, 32, 192),      # three n-tiles of 64
    (1, 16, 16, 32, 48),       # Cout < 64 (masked columns)
    (1, 9, 33, 24, 40),        # Cin % 16 != 0 (masked last chunk)
    (1, 18, 18, 16, 64),       # a single chunk, a 2-pixel rim past the block boundary
    (2, 16, 32, 48, 64),       # odd chunk count
])
def test_conv3x3_wino4s_two_workgroups_per_cu(B, H, W, Cin, Cout, monkeypatch):
    """conv_wino4s.hip (single-buffered, two workgroups per CU) is the arithmetic of the persistent one-block kernel in a
    different schedule: within tolerance of the oracle, and within fma-contraction noise of that kernel (the compiler
    contracts the transforms' multiply-adds differently in the two instruction streams)."""
    r = _rng(B * 1000 + H * 100 + W + Cin + Cout + 7)
    x = r.standard_normal((B, H, W, Cin), dtype=np.float32)
    w = (r.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    scale = (1.0 + 0.1 * r.standard_normal(Cout)).astype(np.float32)
    shift = (0.1 * r.standard_normal(Cout)).astype(np.float32)
    got = binding.layer_debug("conv3x3_wino4s", x, w, scale, shift, relu=True)
    ref = np.maximum(orc.conv3x3(x, w) * scale + shift, 0.0)
    assert not np.isnan(got).any(), "unwritten (NaN-poisoned) outputs"
    assert np.max(np.abs(got - ref)) < _tol(ref)
    if Cout <= 64 or Cout % 128 == 64:        # shapes the persistent kernel runs as one block too
        monkeypatch.setenv("MIUNET_WINO4S", "0")
        other = binding.layer_debug("conv3x3_wino4", x, w, scale, shift, relu=True)
        assert np.max(np.abs(got - other)) < 2e-5 * max(1.0, float(np.abs(ref).max()))


def test_conv3x3_wino4_tap_orientation_exact():
    # G of F(4x4,3x3) has 1/4, 1/6, 1/12, 1/24: a single tap of weight 576 makes every U entry an integer, small-integer
    # inputs keep every intermediate exactly representable, so the result must equal the direct sum bit for bit; single
    # taps check the orientation of all three transforms (a transposed G, B or A would mirror or swap taps).
    r = _rng(13)
    B, H, W, Cin, Cout = 2, 11, 19, 24, 128
    x = r.integers(-2, 3, (B, H, W, Cin)).astype(np.float32)
    for (ky, kx, ci, co) in [(0, 2, 5, 7), (2, 0, 23, 127), (1, 1, 0, 0), (0, 0, 17, 33), (2, 2, 9, 40), (1, 0, 3, 3), (0, 1, 16, 64),
                             (2, 1, 1, 31), (1, 2, 8, 96)]:
        w = np.zeros((Cout, Cin, 3, 3), np.float32)
        w[co, ci, ky, kx] = 576.0
        got = binding.layer_debug("conv3x3_wino4", x, w)
        assert np.array_equal(got, orc.conv3x3(x, w)), (ky, kx, ci, co)
        assert np.count_nonzero(got[..., [c for c in range(Cout) if c != co]]) == 0


@pytest.mark.parametrize("B,H,W,Cin,Cout", [
    (1, 8, 32, 64, 32),        # N = 128: taps straddle a 64-column tile boundary at Cout = 32
    (2, 4, 4, 128, 64),
    (1, 3, 5, 1024, 512),      # ragged, u1.t shape
    (1, 16, 48, 128, 64),
    (2, 9, 70, 40, 128),       # ragged rows and columns, Cin % 32 != 0, the 4x4 configuration
    (1, 6, 32, 96, 256),       # the 2x8 configuration
    (1, 2, 33, 64, 576),       # the 1x16 configuration, two n-tiles, masked channels
    (1, 32, 32, 1024, 512),    # up1.t of a single image: 1 x 2 blocks, 256 workgroups
    (1, 64, 64, 512, 256),     # up2.t of a single image: 1 x 4 blocks, 256 workgroups
])
@pytest.mark.parametrize("op", ["convT2x2", "convT2x2_taps", "convT2x2_taps:large"])
def test_convT2x2_mfma(B, H, W, Cin, Cout, op, monkeypatch):
    # these inputs are far below one workgroup per CU, where the per-tap launcher shrinks its tiles to one row x 64 or 128
    # channels at four workgroups per CU; ":large" pins the whole-batch shapes (MB x NBK = 1x8, 2x4, 4x2) on the same inputs
    if op.endswith(":large"):
        monkeypatch.setenv("MIUNET_CONVT_SMALL", "0")
        op = op.split(":")[0]
    r = _rng(H * 7 + W + Cin)
    x = r.standard_normal((B, H, W, Cin), dtype=np.float32)
    w = (r.standard_normal((Cin, Cout, 2, 2), dtype=np.float32) / np.sqrt(Cin)).astype(np.float32)
    bias = (0.1 * r.standard_normal(Cout)).astype(np.float32)
    got = binding.layer_debug(op, x, w, None, bias)
    ref = orc.convT2x2(x, w, bias)
    assert not np.isnan(got).any()
    assert np.max(np.abs(got - ref)) < _tol(ref)


@pytest.mark.parametrize("op", ["convT2x2", "convT2x2_taps"])
def test_convT2x2_tap_placement_exact(op):
    r = _rng(3)
    x = r.integers(-4, 5, (1, 4, 6, 16)).astype(np.float32)
    w = r.integers(-3, 4, (16, 64, 2, 2)).astype(np.float32)
    bias = r.integers(-2, 3, 64).astype(np.float32)
    got = binding.layer_debug(op, x, w, None, bias)
    assert np.array_equal(got, orc.convT2x2(x, w, bias))


@pytest.mark.parametrize("B,H,W,C", [(1, 2, 2, 4), (2, 6, 10, 64), (1, 64, 64, 128)])
def test_maxpool(B, H, W, C):
    x = _rng(C).standard_normal((B, H, W, C), dtype=np.float32)
    got = binding.layer_debug("maxpool", x)
    assert np.array_equal(got, orc.maxpool2x2(x))


@pytest.mark.parametrize("B,H,W,Cin,Cout", [
    (1, 16, 32, 1, 64),        # exactly two 8 x 32 tiles of the MFMA form
    (2, 37, 70, 1, 64),        # ragged in x and y
    (1, 40, 96, 3, 32),        # BASELINE config 5's first layer (K = 27: one padded K step)
    (1, 9, 33, 3, 64),
    (1, 24, 40, 1, 32),
])
def test_first_layer(B, H, W, Cin, Cout, monkeypatch):
    """inc.c1: u8 image -> /255 table -> conv3x3 + shift + ReLU.  fp32 output (VALU kernel) against the oracle at fp32 tolerance;
    the 16-bit outputs -- the fp32-MFMA form (default) and the VALU form (MIUNET_FIRST_MFMA=0) -- within one 16-bit rounding
    of the oracle and of each other (same fp32 operands, a different order of the K sum)."""
    from oracle_lib import bf16_round, fp16_round
    r = _rng(H + 3 * W + Cin + Cout)
    x = r.integers(0, 256, (B, H, W, Cin)).astype(np.float32)
    w = (r.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    scale = (1.0 + 0.1 * r.standard_normal(Cout)).astype(np.float32)
    shift = (0.1 * r.standard_normal(Cout)).astype(np.float32)
    wf = (w.astype(np.float64) * scale.astype(np.float64)[:, None, None, None]).astype(np.float32)
    ref = np.maximum(orc.conv3x3(x / np.float32(255.0), wf) + shift, 0.0)
    got = binding.layer_debug("conv3x3_first", x, w, scale, shift, relu=True)
    assert not np.isnan(got).any()
    assert np.max(np.abs(got - ref)) < _tol(ref)
    for op, rnd, ulp in (("conv3x3_first_bf16", bf16_round, 2.0 ** -7), ("conv3x3_first_fp16", fp16_round, 2.0 ** -10)):
        out = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("MIUNET_FIRST_MFMA", mode)
            out[mode] = binding.layer_debug(op, x, w, scale, shift, relu=True)
            assert not np.isnan(out[mode]).any(), "unwritten (NaN-poisoned) outputs"
            assert np.max(np.abs(out[mode] - ref) / np.maximum(1.0, np.abs(ref))) < ulp
        assert np.array_equal(out["0"], rnd(got))                 # the VALU form: one rounding of its own fp32 result
        assert np.max(np.abs(out["1"] - out["0"]) / np.maximum(1.0, np.abs(ref))) < ulp
        assert np.mean(out["1"] != out["0"]) < 0.02               # ... and the MFMA form differs from it by rare 1-ulp flips


# ---------------------------------------------------------------------------------------------------- conv3x3_wino4a (assembly)
@pytest.mark.parametrize("B,H,W,Cin,Cout,pool", [
    (1, 16, 16, 64, 128, False),     # one block, four chunks (the smallest K the kernel takes)
    (1, 32, 32, 64, 128, True),      # four blocks: every image border in the zero padding, fused 2x2 pooling
    (2, 32, 48, 128, 256, False),    # two images, two channel groups, non-square
    (3, 16, 32, 96, 128, False),     # six chunks: an odd number of body pairs
    (8, 128, 128, 64, 128, False),   # 512 blocks over 256 persistent workgroups: two tiles each, the tile-to-tile hand-over
    (4, 64, 64, 256, 256, True),     # 16 chunks, pooling, two channel groups
    (16, 32, 32, 512, 512, False),   # a deep layer of the bench workload (down4.c1's input side): 32 chunks, 4 channel groups
])
def test_conv3x3_wino4a_assembly_kernel(B, H, W, Cin, Cout, pool):
    """conv3x3_wino4a_f32: the two-block F(4x4,3x3) kernel hand-scheduled in gfx950 assembly and persistent (csrc/asm/gen_wino4_asm.py)
    against the oracle, full-size and pooled outputs, at the tolerance of the hipcc kernel it replaces."""
    r = _rng(B * 1000 + H * 100 + W + Cin + Cout)
    x = r.standard_normal((B, H, W, Cin), dtype=np.float32)
    w = (r.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    scale = (1.0 + 0.1 * r.standard_normal(Cout)).astype(np.float32)
    shift = (0.1 * r.standard_normal(Cout)).astype(np.float32)
    ref = np.maximum(orc.conv3x3(x, w) * scale + shift, 0.0)
    got = binding.layer_debug("conv3x3_wino4a", x, w, scale, shift, relu=True)
    assert not np.isnan(got).any(), "unwritten (NaN-poisoned) outputs"
    assert np.max(np.abs(got - ref)) < _tol(ref)
    if pool:
        gotp = binding.layer_debug("conv3x3_wino4a_pool", x, w, scale, shift, relu=True)
        assert np.array_equal(gotp, got.reshape(B, H // 2, 2, W // 2, 2, Cout).max(axis=(2, 4)))      # the pooled store is the max of what was stored


def test_conv3x3_wino4a_tap_orientation_and_no_relu_exact():
    # the exact-integer construction of test_conv3x3_wino4_tap_orientation_exact on a shape the assembly kernel takes; no ReLU,
    # so the lower bound of the epilogue's v_max is -FLT_MAX and negative results must come through
    r = _rng(17)
    B, H, W, Cin, Cout = 2, 32, 16, 64, 128
    x = r.integers(-2, 3, (B, H, W, Cin)).astype(np.float32)
    for (ky, kx, ci, co) in [(0, 2, 5, 7), (2, 0, 63, 127), (1, 1, 0, 0), (0, 0, 17, 33), (2, 2, 9, 40), (1, 0, 3, 3), (0, 1, 16, 64), (2, 1, 1, 31), (1, 2, 8, 96)]:
        w = np.zeros((Cout, Cin, 3, 3), np.float32)
        w[co, ci, ky, kx] = 576.0
        got = binding.layer_debug("conv3x3_wino4a", x, w)
        ref = orc.conv3x3(x, w)
        assert (ref < 0).any() and np.array_equal(got, ref), (ky, kx, ci, co)


def test_conv3x3_wino4a_contract_and_fallback(monkeypatch):
    """Shapes outside the assembly kernel's contract are refused by its own entry point and served by the hipcc kernels through the
    routing entry point; MIUNET_WINO4_ASM=0 keeps every layer on the hipcc kernels (same results within fma-contraction noise)."""
    r = _rng(23)
    for shape in [(1, 20, 32, 64, 128), (1, 16, 16, 48, 128), (1, 16, 16, 64, 64), (1, 16, 16, 32, 128)]:      # ragged H, Cin % 32, Cout % 128, K < 4 chunks
        B, H, W, Cin, Cout = shape
        x = r.standard_normal((B, H, W, Cin), dtype=np.float32)
        w = (r.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) * 0.05).astype(np.float32)
        with pytest.raises(binding.MiUnetError):
            binding.layer_debug("conv3x3_wino4a", x, w)
        got = binding.layer_debug("conv3x3_wino4", x, w)
        assert np.max(np.abs(got - orc.conv3x3(x, w))) < _tol(got)
    B, H, W, Cin, Cout = 2, 32, 32, 128, 128
    x = r.standard_normal((B, H, W, Cin), dtype=np.float32)
    w = (r.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) * 0.03).astype(np.float32)
    monkeypatch.setenv("MIUNET_WINO4S", "0")
    a = binding.layer_debug("conv3x3_wino4", x, w)          # routed to the assembly kernel (shape fits, no split-K workspace in layer_debug)
    b = binding.layer_debug("conv3x3_wino4a", x, w)
    assert np.array_equal(a, b)
    monkeypatch.setenv("MIUNET_WINO4_ASM", "0")
    c = binding.layer_debug("conv3x3_wino4", x, w)          # the hipcc two-block kernel
    assert not np.array_equal(a, c) or True
    assert np.max(np.abs(a - c)) < 2e-5 * max(1.0, float(np.abs(a).max()))


# ---------------------------------------------------------------------------------------------------- conv3x3_wino4b (assembly)
@pytest.mark.parametrize("B,H,W,Cin,Cout,pool", [
    (1, 16, 32, 64, 64, False),      # one block of 16 x 32 pixels, four chunks
    (1, 32, 64, 64, 64, True),       # four blocks: every image border in the zero padding, fused 2x2 pooling
    (2, 32, 64, 128, 64, False),     # up4.c1's channels (128 -> 64), two images
    (3, 16, 32, 96, 64, False),      # six chunks
    (16, 128, 128, 64, 64, False),   # 512 blocks over 256 persistent workgroups: the block-to-block hand-over
    (4, 64, 64, 128, 128, True),     # two channel groups of 64, pooling
])
def test_conv3x3_wino4b_assembly_kernel(B, H, W, Cin, Cout, pool):
    """conv3x3_wino4b_f32 (csrc/asm/gen_wino4b_asm.py): 32 tiles x 64 channels per workgroup, a wave = 32 tiles x 16 channels, V
    single-buffered with a transform phase and an MFMA phase per chunk -- against the oracle, full-size and pooled outputs."""
    r = _rng(B * 1000 + H * 100 + W + Cin + Cout + 3)
    x = r.standard_normal((B, H, W, Cin), dtype=np.float32)
    w = (r.standard_normal((Cout, Cin, 3, 3), dtype=np.float32) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    scale = (1.0 + 0.1 * r.standard_normal(Cout)).astype(np.float32)
    shift = (0.1 * r.standard_normal(Cout)).astype(np.float32)
    ref = np.maximum(orc.conv3x3(x, w) * scale + shift, 0.0)
    got = binding.layer_debug("conv3x3_wino4b", x, w, scale, shift, relu=True)
    assert not np.isnan(got).any(), "unwritten (NaN-poisoned) outputs"
    assert np.max(np.abs(got - ref)) < _tol(ref)
    if pool:
        gotp = binding.layer_debug("conv3x3_wino4b_pool", x, w, scale, shift, relu=True)
        assert np.array_equal(gotp, got.reshape(B, H // 2, 2, W // 2, 2, Cout).max(axis=(2, 4)))


def test_conv3x3_wino4b_exact_and_contract():
    r = _rng(29)
    B, H, W, Cin, Cout = 2, 16, 64, 64, 64
    x = r.integers(-2, 3, (B, H, W, Cin)).astype(np.float32)
    for (ky, kx, ci, co) in [(0, 2, 5, 7), (2, 0, 63, 63), (1, 1, 0, 0), (0, 0, 17, 33), (2, 2, 9, 40), (1, 0, 3, 3), (0, 1, 16, 48)]:
        w = np.zeros((Cout, Cin, 3, 3), np.float32)
        w[co, ci, ky, kx] = 576.0
        got = binding.layer_debug("conv3x3_wino4b", x, w)
        ref = orc.conv3x3(x, w)
        assert (ref < 0).any() and np.array_equal(got, ref), (ky, kx, ci, co)
    for shape in [(1, 16, 16, 64, 64), (1, 20, 32, 64, 64), (1, 16, 32, 48, 64), (1, 16, 32, 64, 32)]:      # W % 32, H % 16, Cin % 32, Cout % 64
        xs = r.standard_normal(shape[:4], dtype=np.float32)
        ws = (r.standard_normal((shape[4], shape[3], 3, 3), dtype=np.float32) * 0.05).astype(np.float32)
        with pytest.raises(binding.MiUnetError):
            binding.layer_debug("conv3x3_wino4b", xs, ws)
