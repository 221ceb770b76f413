#!/usr/bin/env python3
"""bench.py -- images/sec of the UNet inference hot path on MI355X (BASELINE.json metric), one process per GPU.

A step = one pass of the hot path (u8 tiles resident in HBM -> UNet forward -> u8 label maps in HBM) over one batch of
16 synthetic 512x512x1 images per GPU (BASELINE.json configs[1]); for N > 1 ranks each rank processes its own shard
(weak scaling) and the per-batch label maps are gathered to rank 0 over RCCL inside the timed region, after the
weights were broadcast from rank 0 over RCCL at start-up (north_star: "RCCL-over-xGMI broadcast of weights and
per-rank gather of mask tensors").

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "unet-medical-image-contour-segmentation-cpp_amd"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from miunet import binding, shard, synth  # noqa: E402
from miunet.spec import UNetSpec, pack_weights  # noqa: E402

FP32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 matrix peak (v_mfma_f32_32x32x2_f32)
BF16_PEAK_TFLOPS = 2500.0     # dense bf16 MFMA peak (v_mfma_f32_32x32x16_bf16)
HBM_PEAK_GBS = 8000.0


def cpu_baseline(blob, h, w, in_ch=1):
    """Oracle (oracle/liboracle.so: the CPU restatement, kind "port") timed on this host's cores on a bounded sample:
    ONE 512x512 image end to end (normalise + UNet forward + argmax)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc

    cores = int(orc.lib().orc_num_threads())
    probe = synth.make_images(1, h, w, in_ch, 0x5EED, "bytes")
    t0 = time.perf_counter()
    orc.unet_forward(blob, probe, want_logits=False)                # also pages in the library and the thread pool
    t1 = time.perf_counter() - t0
    n = int(min(32, max(1, np.ceil(20.0 / t1))))                    # bounded sample: about 10-30 s of CPU work
    imgs = synth.make_images(n, h, w, in_ch, 0x5EED, "bytes")
    t0 = time.perf_counter()
    orc.unet_forward(blob, imgs, want_logits=False)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{n} image(s) {h}x{w}x{in_ch} of the same synthetic workload through oracle/unet_oracle.c "
                      f"(normalise + UNet forward + argmax; OpenMP, {cores} threads), {dt:.2f} s",
            "ms_per_image": dt / n * 1e3}


def pmc_traffic(kernel):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same command
    (profiles/summarize_pmc.py: FETCH_SIZE doubled per the gfx950 correction + WRITE_SIZE; counters cannot be read from
    inside the process, so the newest committed summary is quoted) -- None when no summary exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))
    if not files:
        return None
    try:
        return json.load(open(files[-1]))[kernel]["hbm_bytes_per_launch"]
    except (KeyError, ValueError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--in-ch", type=int, default=1)
    ap.add_argument("--base", type=int, default=64)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-layer", action="store_true", help="also print a per-layer table to stderr")
    ap.add_argument("--conv-algo", choices=["auto", "direct", "winograd", "winograd16", "bf16", "fp16"], default="auto")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # BENCH_FORCE_DIST=1 runs the RCCL code path (init, weight broadcast, label gather, barrier) even with one rank, so it
    # can be rehearsed on a 1-GPU box
    use_dist = world > 1 or os.environ.get("BENCH_FORCE_DIST") == "1"
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    spec = UNetSpec(args.in_ch, args.base, args.levels, 3)
    H = W = args.size
    B = args.batch

    # ---- weights: rank 0 generates, everybody else receives them over RCCL
    blob = pack_weights(spec, synth.make_weights(spec, 1234)) if rank == 0 else None
    if use_dist:
        blob = shard.broadcast_blob(blob, spec.n_params() * 4 + 36, dev)

    eng = binding.Engine(H, W, spec.in_ch, spec.base, spec.levels, spec.classes, max_batch=B, device=local_rank,
                         conv_algo=args.conv_algo)
    eng.load_weights(blob)
    stream = torch.cuda.current_stream(dev)
    eng.set_stream(stream.cuda_stream)

    # ---- this rank's shard of the synthetic batch, resident in HBM before the timed region
    imgs = torch.from_numpy(synth.make_images(B, H, W, spec.in_ch, 0x5EED + 1000 * rank, "bytes")).to(dev)
    labels = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    gathered = [torch.empty_like(labels) for _ in range(world)] if (use_dist and rank == 0) else None

    def step():
        eng.infer_device(imgs.data_ptr(), B, labels.data_ptr(), 0)
        if use_dist:
            dist.gather(labels, gathered, dst=0)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    eng.set_profiling(True)             # event pair around every launch on the launch stream; no host waits
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    stats = eng.kernel_stats()
    eng.set_profiling(False)

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        images = B * world * args.steps
        ips = images / dt
        # dominant kernel: the conv3x3 kernel with the largest share of device time (the default plan mixes the F(4x4,3x3)
        # kernel with the F(2x2,3x3) one for small grids)
        conv_kernels = ("conv3x3_mfma", "conv3x3_wino", "conv3x3_wino16", "conv3x3_wino4", "conv3x3_bf16", "conv3x3_fp16")
        by_kernel = {}
        for s in stats:
            if s["kernel"] in conv_kernels:
                by_kernel[s["kernel"]] = by_kernel.get(s["kernel"], 0.0) + s["ms"]
        dom_kernel = max(by_kernel, key=by_kernel.get) if by_kernel else "conv3x3_mfma"
        dom = [s for s in stats if s["kernel"] == dom_kernel]
        dom_flops = sum(s["flops"] for s in dom)
        dom_ms = sum(s["ms"] for s in dom)
        all_ms = sum(s["ms"] for s in stats)
        achieved = dom_flops / (dom_ms * 1e-3) / 1e12
        is_bf16 = dom_kernel in ("conv3x3_bf16", "conv3x3_fp16")
        lp_name = "fp16" if dom_kernel == "conv3x3_fp16" else "bf16"
        is_cfg1 = (spec.in_ch, spec.base, spec.levels, H) == (1, 64, 4, 512)
        peak = BF16_PEAK_TFLOPS if is_bf16 else FP32_PEAK_TFLOPS
        out = {
            "metric": "images/sec, %s UNet %s inference (u8 tile -> u8 label map)" % (f"{H}x{W}", (lp_name + "-operand / fp32-accumulate") if is_bf16 else "fp32"),
            "value": ips,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_image": dt / images * 1e3 * world,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": lp_name if is_bf16 else "fp32",
            "data": "synthetic",
            "config": {"workload": (f"BASELINE.json configs[{(2 if is_bf16 else 1) if is_cfg1 else 4}]: " if (is_cfg1 or (spec.in_ch, spec.base, spec.levels, H) == (3, 32, 5, 1024)) else "") +
                                   f"batch {B} x {H}x{W}x{spec.in_ch} u8 per GPU, {spec.levels}-level UNet base {spec.base}, {(lp_name + ' operands / fp32 accumulate') if is_bf16 else 'fp32'}, "
                                   "argmax label maps", "images_per_gpu_per_step": B, "global_batch": B * world,
                       "parallelism": f"dp{world}" + (" (RCCL weight broadcast + per-step label-map gather)" if world > 1 else "")},
            "roofline": {
                "bound": "mfma", "kernel": dom_kernel + ((" (v_mfma_f32_32x32x16_%s)" % ("f16" if lp_name == "fp16" else "bf16")) if is_bf16
                                       else " (v_mfma_f32_16x16x4_f32)" if dom_kernel == "conv3x3_wino4" else " (v_mfma_f32_32x32x2_f32)"),
                "algorithm": "winograd F(4x4,3x3): achieved counts ALGORITHMIC (direct-convolution) FLOPs, the MFMA pipe "
                             "executes 1/4 of them" if dom_kernel == "conv3x3_wino4" else
                             "winograd F(2x2,3x3): achieved counts ALGORITHMIC (direct-convolution) FLOPs, the MFMA pipe "
                             "executes 1/2.25 of them" if dom_kernel.startswith("conv3x3_wino") else "direct implicit GEMM",
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": pmc_traffic({"conv3x3_wino": "miunet::conv3x3_wino_f32<*>", "conv3x3_wino16": "miunet::conv3x3_wino16_f32",
                                        "conv3x3_wino4": "miunet::conv3x3_wino4_f32<*>"}.get(
                    dom_kernel, "miunet::conv_mfma_f32<9, 8, 64, 16, false>")),
                "launches": len(dom), "avg_launch_ms": dom_ms / max(1, len(dom)),
                "avg_launch_gflop": dom_flops / max(1, len(dom)) / 1e9,
                "share_of_device_time": dom_ms / all_ms if all_ms else None,
                "whole_net_tflops": 2.0 * spec.macs_per_image(H, W) * ips / world / 1e12,
            },
        }
        if args.per_layer:
            per = {}
            for s in stats:
                e = per.setdefault(s["name"], [s["kernel"], 0.0, 0.0, 0.0, 0])
                e[1] += s["ms"]; e[2] += s["flops"]; e[3] += s["bytes"]; e[4] += 1
            print(f"{'layer':14s} {'kernel':16s} {'ms/launch':>10s} {'TFLOP/s':>9s} {'GB/s(alg)':>10s}", file=sys.stderr)
            for name, (k, ms, fl, by, n) in per.items():
                print(f"{name:14s} {k:16s} {ms / n:10.3f} {fl / ms / 1e9:9.1f} {by / ms / 1e6:10.0f}", file=sys.stderr)
            print(f"sum of kernel time per step: {all_ms / args.steps:.3f} ms; wall per step: {dt / args.steps * 1e3:.3f} ms", file=sys.stderr)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(blob, H, W, spec.in_ch)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if use_dist:
        if rank == 0 and gathered is not None and not torch.equal(gathered[0], labels):
            raise SystemExit("gathered label maps differ from rank 0's own")
        dist.barrier()
        dist.destroy_process_group()
    eng.close()


if __name__ == "__main__":
    main()
