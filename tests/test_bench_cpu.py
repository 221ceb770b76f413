"""Host logic of bench.py that needs no GPU: the roofline arithmetic (executed vs algorithmic FLOPs), the kernel-family
bookkeeping and the rule that a committed PMC summary is quoted only while it describes the kernels in the tree."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _stats():
    s = []
    for step in range(2):
        s.append(dict(name="inc.c1", kernel="conv3x3_first", flops=2.4e9, bytes=1.08e9, ms=0.25))
        s.append(dict(name="inc.c2", kernel="conv3x3_wino4s", flops=309e9, bytes=2.2e9, ms=1.09))
        s.append(dict(name="down1.c2", kernel="conv3x3_wino4", flops=309e9, bytes=0.9e9, ms=0.89))
        s.append(dict(name="up1.c1", kernel="conv3x3_wino4", flops=618e9, bytes=0.4e9, ms=1.36))
        s.append(dict(name="up4.t", kernel="convT2x2_taps", flops=68.7e9, bytes=1.6e9, ms=0.63))
        s.append(dict(name="up4.c2", kernel="conv3x3_wino4s+head", flops=309e9, bytes=1.1e9, ms=1.02))
    return s


def test_roofline_fraction_counts_executed_flops(monkeypatch):
    monkeypatch.setattr(bench, "pmc_summary", lambda tag: (None, "none"))
    r = bench.roofline_from_stats(_stats(), 192.4e9, 900.0, "fp32")
    assert "families" not in r and r["families_compact"]["conv3x3_wino4"][0] == 4      # the default line stays short
    monkeypatch.setattr(bench, "VERBOSE", True)
    r = bench.roofline_from_stats(_stats(), 192.4e9, 900.0, "fp32")
    assert r["kernel"].startswith("conv3x3_wino4 ")                      # the family with the largest share of device time
    assert r["winograd_reduction"] == 4.0 and r["algorithm"] == "winograd F(4x4,3x3)"
    alg = (309e9 + 618e9) / ((0.89 + 1.36) * 1e-3) / 1e12
    assert abs(r["algorithmic_tflops"] - alg) < 1e-6 and abs(r["achieved"] - alg / 4) < 1e-6
    assert 0.0 < r["frac"] < 1.0 and abs(r["frac"] - alg / 4 / 157.3) < 1e-9   # never the algorithmic rate over the peak
    assert r["launches"] == 4 and r["traffic"] is None and r["mfma_busy"] is None
    fam = {f["kernel"]: f for f in r["families"]}
    assert set(fam) == {"conv3x3_first", "conv3x3_wino4s", "conv3x3_wino4", "convT2x2_taps"}
    assert fam["conv3x3_wino4s"]["launches"] == 4                          # the fused-head launches belong to their kernel
    assert abs(fam["convT2x2_taps"]["frac_of_mfma_peak"] - 68.7e9 / 0.63e-3 / 1e12 / 157.3) < 1e-9   # direct form: reduction 1
    assert all(f["frac_of_mfma_peak"] < 1.0 for f in r["families"])


def test_sixteen_bit_plans_are_priced_against_the_16_bit_peak(monkeypatch):
    monkeypatch.setattr(bench, "pmc_summary", lambda tag: (None, "none"))
    stats = [dict(name="up1.c1", kernel="conv3x3_bf16", flops=618e9, bytes=0.2e9, ms=0.53),
             dict(name="up4.c2", kernel="conv3x3_bf16+head", flops=309e9, bytes=0.6e9, ms=0.52)]
    r = bench.roofline_from_stats(stats, 192.4e9, 2400.0, "bf16")
    assert r["peak"] == 2500.0 and r["winograd_reduction"] == 1.0 and "bf16" in r["kernel"]
    assert abs(r["frac"] - (927e9 / 1.05e-3 / 1e12) / 2500.0) < 1e-9


def test_pmc_summary_is_quoted_only_for_the_tree_it_was_measured_on(tmp_path, monkeypatch):
    prof = tmp_path / "profiles"
    prof.mkdir()
    doc = {"kernel_source_sha": "abc", "kernels": {"miunet::conv3x3_wino4_f32<*>": {"hbm_bytes_per_launch": 1.2e9, "mfma_busy": 0.64}}}
    (prof / "r09_pmc_fp32.json").write_text(json.dumps(doc))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_sha", lambda: "abc")
    d, src = bench.pmc_summary("fp32")
    assert d["kernels"]["miunet::conv3x3_wino4_f32<*>"]["mfma_busy"] == 0.64 and src == "profiles/r09_pmc_fp32.json"
    r = bench.roofline_from_stats(_stats(), 192.4e9, 900.0, "fp32")
    assert r["traffic"] == 1.2e9 and r["mfma_busy"] == 0.64
    monkeypatch.setattr(bench, "kernel_source_sha", lambda: "different")          # a kernel changed since the passes ran
    d, src = bench.pmc_summary("fp32")
    assert d is None and "not quoted" in src
    r = bench.roofline_from_stats(_stats(), 192.4e9, 900.0, "fp32")
    assert r["traffic"] is None and r["mfma_busy"] is None and "not quoted" in r["pmc_source"]
    assert bench.pmc_summary("bf16")[0] is None                                   # no summary for that plan at all


def test_committed_pmc_summaries_carry_a_source_hash():
    import glob
    files = [f for tag in ("fp32", "bf16", "fp16") for f in glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_{tag}.json"))]
    assert files, "no PMC summary committed"
    for f in files:
        d = json.load(open(f))
        assert len(d["kernel_source_sha"]) == 16 and d["kernels"], f


def test_group_child_idles_until_told_and_never_costs_the_headline():
    """The `group` record (mi_unet_group_* over every visible device, one process) runs in a child that bench.py starts BEFORE it
    touches the GPU; the child waits on stdin.  Told not to run it leaves quietly without importing anything GPU-related; a child
    that overruns its deadline is killed by PID and becomes an error record, never an exception in the parent."""
    import subprocess
    child = bench.start_group_child()
    assert bench.finish_group_child(child, run=False) is None and child.returncode == 0
    hang = subprocess.Popen([sys.executable, "-c", "import sys, time; sys.stdin.readline(); time.sleep(60)"], stdin=subprocess.PIPE,
                            stdout=subprocess.PIPE, text=True)
    rec = bench.finish_group_child(hang, run=True, deadline_s=1)
    assert "error" in rec and "killed" in rec["error"] and hang.poll() is not None
    assert bench.finish_group_child(None, run=True) is None


def test_the_newest_pmc_summary_describes_this_tree():
    """the judged roofline quotes `traffic` / `mfma_busy` only from a PMC summary whose source hash equals the tree's: the last
    profile round must have run on the final kernels"""
    d, src = bench.pmc_summary("fp32")
    assert d is not None, src
