"""The multi-device group, cloned contexts and the multi-plane RAW path of include/mi_unet.h.

The reference is one image on one implicit device with one context per host thread (src/process.cpp:15, :70;
src/main.cpp:148-164); these entry points are what its file loop and its thread-local context map onto.  The CPU tests
check the sharding arithmetic and the loud failure without a device; the GPU tests run on ONE card: a one-rank group
(optionally with a one-rank RCCL communicator, so the RCCL calls are exercised) and a two-rank group whose ranks share the
card (threads, shards and the peer-copy weight path).  A group over several distinct devices has never run here."""
import os
import threading

import numpy as np
import pytest

import oracle_lib as orc
from miunet import binding, shard, synth
from miunet.spec import UNetSpec, pack_weights


# ---------------------------------------------------------------------------------------------------------------- CPU
@pytest.mark.parametrize("n,world", [(0, 1), (1, 1), (5, 2), (1, 2), (512, 8), (7, 8), (16, 3), (1000, 7)])
def test_shard_range_is_the_contiguous_split(n, world):
    seen = []
    for r in range(world):
        lo, hi = binding.shard_range(n, r, world)
        assert (lo, hi) == shard.shard_range(n, r, world)              # same split as the torch.distributed helpers
        assert 0 <= lo <= hi <= n and hi - lo in (n // world, n // world + 1)
        seen.extend(range(lo, hi))
    assert seen == list(range(n))                                      # disjoint, ordered, complete


def test_shard_range_rejects_bad_arguments():
    with pytest.raises(binding.MiUnetError):
        binding.shard_range(4, 2, 2)
    with pytest.raises(binding.MiUnetError):
        binding.shard_range(-1, 0, 1)


def test_group_scheduling_on_cpu_with_a_stub_backend(tmp_path):
    """csrc/group_sched.h -- the shard split, the per-rank worker threads and the run-on-every-rank / first-failure logic of
    the multi-device group -- compiled with g++ against a stub backend and run here: no GPU involved (1, 2, 3 and 8 ranks;
    empty, ragged and full batches; concurrency; error propagation from a failing rank's own thread)."""
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "group_sched_test"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-o", str(exe), os.path.join(root, "tests", "cpu", "group_sched_test.cpp")])
    r = subprocess.run([str(exe)], capture_output=True, timeout=120)
    assert r.returncode == 0, r.stdout.decode() + r.stderr.decode()
    assert b"all group scheduling checks passed" in r.stdout


def test_group_without_device_fails_loudly():
    if binding.device_count() > 0:
        pytest.skip("a HIP device is visible")
    with pytest.raises(binding.MiUnetError) as ei:
        binding.Group(64, 64, max_batch=1, n_devices=1)
    assert ei.value.code == 2 and "no CPU fallback" in str(ei.value)


# ---------------------------------------------------------------------------------------------------------------- GPU
def _small():
    spec = UNetSpec(1, 16, 2, 3)
    blob = pack_weights(spec, synth.make_weights(spec, 77))
    return spec, blob


@pytest.mark.gpu
def test_group_rccl_code_path_executes_on_a_stand_in_library(tmp_path, monkeypatch):
    """The group's RCCL calls -- ncclCommInitAll, the grouped ncclBroadcast of the weight blob, the grouped ncclSend / ncclRecv
    gather of label maps into the first rank -- have never met a second GPU (gpurun offers one).  tests/cpu/fake_rccl.cpp is a
    stand-in library with the same entry points whose transfers are device-to-device copies, so that this code path EXECUTES here
    with two and three ranks sharing the card: a wrong peer, offset or count in csrc/group.cpp gives wrong label maps, an
    unmatched send / recv an error.  (It says nothing about RCCL itself.)"""
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    so = tmp_path / "libfake_rccl.so"
    subprocess.check_call([hipcc, "-O1", "-std=c++17", "-fPIC", "-shared", "-x", "hip", "--offload-arch=gfx950",
                           os.path.join(root, "tests", "cpu", "fake_rccl.cpp"), "-o", str(so)], stderr=subprocess.DEVNULL)
    monkeypatch.setenv("MIUNET_RCCL_LIB", str(so))
    monkeypatch.setenv("MIUNET_GROUP_RCCL", "2")
    spec, blob = _small()
    imgs = synth.make_images(7, 64, 96, 1, 0xC4, "blobs")
    with binding.Engine(64, 96, 1, 16, 2, 3, max_batch=2) as eng:
        eng.load_weights(blob)
        want, _ = eng.infer(imgs)
    for devices in ([0, 0], [0, 0, 0]):
        with binding.Group(64, 96, 1, 16, 2, 3, max_batch=2, devices=devices) as g:
            assert g.size == len(devices) and g.weight_transport == "rccl"
            g.load_weights(blob)                       # ranks > 0 receive the packed blob through the grouped broadcast
            assert g.weight_transport == "rccl"
            for b in (7, 5, 1, 2):                     # ragged shards; with one image the later ranks own nothing
                g.set_gather("host")
                lab_h, _ = g.infer(imgs[:b])
                g.set_gather("xgmi")
                lab_x, _ = g.infer(imgs[:b])
                assert np.array_equal(lab_h, want[:b]) and np.array_equal(lab_x, want[:b]), (devices, b)
            g.set_postprocess(True)
            g.set_gather("xgmi")
            post, _ = g.infer(imgs)
            g.set_postprocess(False)
        assert post.shape == want.shape and set(np.unique(post)) <= {0, 2}


@pytest.mark.gpu
def test_group_of_one_equals_the_engine():
    spec, blob = _small()
    imgs = synth.make_images(5, 64, 96, 1, 0xA1, "blobs")
    with binding.Engine(64, 96, 1, 16, 2, 3, max_batch=2) as eng:
        eng.load_weights(blob)
        lab0, log0 = eng.infer(imgs, want_logits=True)
    with binding.Group(64, 96, 1, 16, 2, 3, max_batch=2, n_devices=1) as g:
        assert g.size == 1 and g.weight_transport == "peer-copy"
        g.load_weights(blob)
        lab, log = g.infer(imgs, want_logits=True)
        with pytest.raises(binding.MiUnetError):                       # no communicator -> no xGMI gather
            g.set_gather("xgmi")
    assert np.array_equal(lab, lab0) and np.array_equal(log, log0)
    ref_log, ref_lab = orc.unet_forward(blob, imgs)
    assert np.max(np.abs(log - ref_log)) < 1e-3


@pytest.mark.gpu
def test_group_rccl_calls_on_a_one_rank_communicator(monkeypatch):
    """MIUNET_GROUP_RCCL=1: librccl is dlopen'ed, ncclCommInitAll builds a communicator for the single rank, the weights go
    through ncclBroadcast and the label maps through the xGMI gather path (no peers: the send / recv group is empty, the
    batch is read back from rank 0 in one D2H)."""
    monkeypatch.setenv("MIUNET_GROUP_RCCL", "1")
    spec, blob = _small()
    imgs = synth.make_images(3, 64, 64, 1, 0xA2, "blobs")
    with binding.Engine(64, 64, 1, 16, 2, 3, max_batch=2) as eng:
        eng.load_weights(blob)
        lab0, _ = eng.infer(imgs)
    with binding.Group(64, 64, 1, 16, 2, 3, max_batch=2, n_devices=1) as g:
        assert g.weight_transport == "rccl"
        g.load_weights(blob)
        g.set_gather("xgmi")
        lab, _ = g.infer(imgs)
        g.set_gather("host")
        lab2, _ = g.infer(imgs)
    assert np.array_equal(lab, lab0) and np.array_equal(lab2, lab0)


@pytest.mark.gpu
@pytest.mark.parametrize("B", [1, 2, 5])
def test_two_ranks_sharing_one_card(B):
    """devices = [0, 0]: two engines, two worker threads, contiguous shards (B = 1 leaves rank 1 idle, B = 5 is ragged),
    weights packed once and copied device-to-device; results identical to one engine."""
    spec, blob = _small()
    imgs = synth.make_images(B, 64, 64, 1, 0xB0 + B, "blobs")
    with binding.Engine(64, 64, 1, 16, 2, 3, max_batch=2) as eng:
        eng.load_weights(blob)
        lab0, log0 = eng.infer(imgs, want_logits=True)
    with binding.Group(64, 64, 1, 16, 2, 3, max_batch=2, devices=[0, 0]) as g:
        assert g.size == 2 and g.weight_transport == "peer-copy"
        g.load_weights(blob)
        lab, log = g.infer(imgs, want_logits=True)
        g.set_postprocess(True)
        post, _ = g.infer(imgs)
    assert np.array_equal(lab, lab0) and np.array_equal(log, log0)
    assert np.array_equal(post, np.stack([orc.postprocess_mask(m) for m in lab0]))


@pytest.mark.gpu
def test_group_segment_raw16_matches_the_single_engine():
    spec = UNetSpec()
    blob = pack_weights(spec, synth.make_threshold_weights(spec))
    raws = [synth.make_raw16(300 + 20 * i, 400, seed=60 + i) for i in range(3)]
    with binding.Engine(512, 512, max_batch=2) as eng:
        eng.load_weights(blob)
        t0, m0, c0 = eng.segment_raw16(raws)
    with binding.Group(512, 512, max_batch=2, devices=[0, 0]) as g:
        g.load_weights(blob)
        t, m, c = g.segment_raw16(raws)
    assert np.array_equal(t, t0) and np.array_equal(m, m0) and c == c0
    assert any(len(x) > 0 for x in c)


@pytest.mark.gpu
def test_clone_shares_the_weights_and_outlives_its_source():
    spec, blob = _small()
    imgs = synth.make_images(3, 64, 64, 1, 0xC1, "blobs")
    eng = binding.Engine(64, 64, 1, 16, 2, 3, max_batch=4)
    eng.load_weights(blob)
    lab0, log0 = eng.infer(imgs, want_logits=True)
    ctx = eng.clone(max_batch=1)                                       # the reference's per-thread context: batch 1
    eng.close()                                                        # the blob stays alive with its last holder
    lab, log = ctx.infer(imgs, want_logits=True)
    ctx.close()
    assert np.array_equal(lab, lab0)
    # batch 1 may choose split-K launches the batch-4 plan did not: equal to fp32 rounding, labels equal
    assert np.max(np.abs(log - log0)) < 1e-4


@pytest.mark.gpu
def test_cloned_contexts_run_concurrently_from_two_threads():
    spec, blob = _small()
    sets = [synth.make_images(4, 64, 64, 1, 0xD0 + t, "blobs") for t in range(2)]
    with binding.Engine(64, 64, 1, 16, 2, 3, max_batch=2) as eng:
        eng.load_weights(blob)
        want = [eng.infer(s)[0] for s in sets]
        ctxs = [eng.clone(max_batch=2) for _ in range(2)]
        got = [[None] * 8, [None] * 8]

        def work(t):
            for k in range(8):
                got[t][k] = ctxs[t].infer(sets[t])[0]

        th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        for c in ctxs:
            c.close()
    for t in range(2):
        for k in range(8):
            assert np.array_equal(got[t][k], want[t])


@pytest.mark.gpu
def test_config5_in_one_call_three_planes_fp16_device_contours():
    """BASELINE configs[4] as ONE call: fused preprocess (three RAW planes per image) -> 5-level base-32 fp16 network at
    1024x1024x3 -> device postprocess -> device contours.  Structured weights (an intensity classifier on plane 0) make the
    label maps exact under fp16 operands, so every stage compares bit for bit with the oracle chain."""
    spec = UNetSpec(3, 32, 5, 3)
    blob = pack_weights(spec, synth.make_threshold_weights(spec))
    B = 2
    planes = [synth.make_raw16(700 + 40 * k, 900, seed=90 + k) for k in range(3 * B)]
    with binding.Engine(1024, 1024, 3, 32, 5, 3, max_batch=1, conv_algo="fp16") as eng:
        eng.load_weights(blob)
        tiles, masks, cont = eng.segment_raw16(planes, cap_points=1 << 15, cap_contours=64)
    assert tiles.shape == (B, 1024, 1024, 3)
    for i in range(B):
        tile = np.stack([orc.preprocess_raw(planes[3 * i + c], 1024, 1024) for c in range(3)], axis=-1)
        assert np.array_equal(tiles[i], tile)
        _, lab = orc.unet_forward(blob, tile[None], want_logits=False, fp16=True)
        vis = orc.mask_to_image(orc.postprocess_mask(lab[0]))
        assert np.array_equal(masks[i], vis)
        assert vis.max() == 255                                        # something survived the 6 % area filter
        assert cont[i] == orc.find_contours(vis)


@pytest.mark.gpu
def test_config3_bf16_batch_32_in_micro_batches_of_16():
    """BASELINE configs[2] at its tile size: 32 images of 512x512 through max_batch-16 micro-batches.  Image 0 against the
    bf16-operand oracle at the size of the bf16 quantisation noise; every other image must not depend on
    its batch neighbours: three of them re-run alone give the same bits."""
    spec = UNetSpec()
    blob = pack_weights(spec, synth.make_weights(spec, 1234))
    imgs = synth.make_images(32, 512, 512, 1, 0x5EED, "blobs")
    ref16, _ = orc.unet_forward(blob, imgs[:1], bf16=True)
    ref32, lab32 = orc.unet_forward(blob, imgs[:1])
    with binding.Engine(512, 512, max_batch=16, conv_algo="bf16") as eng:
        eng.load_weights(blob)
        labels, logits = eng.infer(imgs, want_logits=True)
        for k in (5, 17, 31):
            lab1, log1 = eng.infer(imgs[k:k + 1], want_logits=True)
            assert np.array_equal(lab1[0], labels[k]) and np.array_equal(log1[0], logits[k])
    noise = float(np.max(np.abs(ref16 - ref32)))
    assert float(np.max(np.abs(logits[:1] - ref32))) < 1.5 * noise + 1e-3
    srt = np.sort(ref32, axis=1)
    safe = (srt[:, -1] - srt[:, -2]) > 0.1
    assert safe.mean() > 0.5 and np.array_equal(labels[:1][safe], lab32[safe])
    assert np.array_equal(labels, np.stack([orc.argmax_planar(l) for l in logits]))


@pytest.mark.gpu
def test_config5_fp16_batch_2():
    spec = UNetSpec(3, 32, 5, 3)
    blob = pack_weights(spec, synth.make_weights(spec, 555))
    imgs = synth.make_images(2, 1024, 1024, 3, 0xC6, "blobs")
    ref16, _ = orc.unet_forward(blob, imgs, fp16=True)
    ref32, lab32 = orc.unet_forward(blob, imgs)
    with binding.Engine(1024, 1024, 3, 32, 5, 3, max_batch=2, conv_algo="fp16") as eng:
        eng.load_weights(blob)
        labels, logits = eng.infer(imgs, want_logits=True)
    noise = float(np.max(np.abs(ref16 - ref32)))
    assert float(np.max(np.abs(logits - ref32))) < 1.5 * noise + 1e-3
    srt = np.sort(ref32, axis=1)
    safe = (srt[:, -1] - srt[:, -2]) > 2e-2
    assert np.array_equal(labels[safe], lab32[safe])


@pytest.mark.gpu
def test_group_clone_is_a_second_lane_with_shared_weights():
    """mi_unet_group_clone: a second set of contexts over the same devices (the facade's second device lane in directory
    mode).  Both groups run different batches at the same time from two threads and agree with the single engine; the clone
    outlives its source."""
    spec = UNetSpec()
    blob = pack_weights(spec, synth.make_threshold_weights(spec))
    sets = [[synth.make_raw16(300 + 16 * i, 420, seed=700 + 10 * t + i) for i in range(3)] for t in range(2)]
    with binding.Engine(512, 512, max_batch=2) as eng:
        eng.load_weights(blob)
        want = [eng.segment_raw16(s) for s in sets]
    g = binding.Group(512, 512, max_batch=2, devices=[0, 0])
    g.load_weights(blob)
    g2 = g.clone()
    got = [None, None]

    def work(t, grp):
        for _ in range(3):
            got[t] = grp.segment_raw16(sets[t])

    th = [threading.Thread(target=work, args=(0, g)), threading.Thread(target=work, args=(1, g2))]
    for x in th:
        x.start()
    for x in th:
        x.join()
    g.close()                                                          # the clone keeps the weights alive
    last = g2.segment_raw16(sets[0])
    g2.close()
    for t in range(2):
        assert np.array_equal(got[t][0], want[t][0]) and np.array_equal(got[t][1], want[t][1]) and got[t][2] == want[t][2]
    assert np.array_equal(last[1], want[0][1]) and last[2] == want[0][2]
