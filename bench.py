#!/usr/bin/env python
"""Headline benchmark: MC-dropout range-image scans/s (64x2048x5, T=8) on MI355X.

One "step" = one pass of the hot path over one batch of synthetic scans resident in HBM:
  T=8 stochastic SalsaNext forwards per scan (Dropout2d live, BatchNorm frozen)  ->  fused
  softmax / mean-over-T / predictive-entropy / mutual-information / argmax reduction  ->
  on-device confusion-matrix and ECE-bin accumulation.
`value` = scans of all ranks / max-over-ranks wall time.  Scans are independent, so ranks shard them
with no data-path collective ("weak" scaling); the 20x20 confusion matrix is all-reduced once after
the timed region, as an evaluation run would.

    python bench.py [--gpus N --steps K --warmup W]          # N>1: launched by torch.distributed.run
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W, T, NCLS = 64, 2048, 8, 20
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
F16_MFMA_PEAK_TFLOPS = 2500.0       # v_mfma_f32_32x32x16_f16, dense
F16X3_MFMA_PEAK_TFLOPS = 2500.0 / 3  # three dense f16 MFMAs (2.5 PFLOP/s) per fp32-class product
HBM_PEAK_GBS = 8000.0


def cpu_baseline(model_sd, x_cpu, threads, passes=T):
    """The oracle (CPU restatement, pinned against the reference) on the host cores: one MC scan
    (T=8 passes of one 64x2048 scan with dropout multipliers + the reduction), after one warm-up pass."""
    from oracle import salsanext as osalsa, uncertainty as ounc
    torch.set_num_threads(threads)
    x1 = x_cpu[:1]
    with torch.no_grad():
        osalsa.salsanext_forward(model_sd, x1)                       # warm-up (oneDNN primitive cache)
        t0 = time.perf_counter()
        outs = []
        g = torch.Generator().manual_seed(0)
        for _ in range(passes):
            outs.append(osalsa.salsanext_forward(model_sd, x1, osalsa.draw_dropout_scales(1, 0.2, g)))
        ounc.mc_reduce(torch.stack(outs, 0))
        dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 4), "unit": "scans/s", "cores": threads, "kind": "port",
            "sample": f"1 scan {x1.shape[2]}x{x1.shape[3]}x5, T={passes} passes + MC reduction, torch-CPU oracle, {dt:.2f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scans", type=int, default=8, help="scans per step per GPU (the reference config batch size for SalsaNext at 64x2048, SemanticKitti_default.yaml:75-78)")
    ap.add_argument("--height", type=int, default=H, help="range-image rows (default: the metric's 64; 128 = BASELINE configs[4])")
    ap.add_argument("--width", type=int, default=W, help="range-image columns (default: the metric's 2048; 4096 = configs[4])")
    ap.add_argument("--passes", type=int, default=T, help="MC passes T (default: the metric's 8; 16 = configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--shared-prefix", action="store_true",
                    help="MC schedule that computes the layers no active Dropout2d can reach once per scan instead of T times "
                         "(bit-identical outputs); default: every pass fully recomputed")
    ap.add_argument("--precision", default="f16", choices=["fp32", "f16x3", "f16"],
                    help="conv precision: exact fp32 MFMA; split-fp16 (fp32 storage, 3 f16 MFMAs, fp32 accumulate); "
                         "f16 = fp16 storage + 1 f16 MFMA, fp32 accumulate (BASELINE configs[2] names half-precision storage)")
    ap.add_argument("--breakdown", default=None, help="write a per-kernel / per-layer-shape timing table to this file")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:      # launched by torch.distributed.run: use RCCL even for a single rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL over xGMI

    from semanticlidarunc_amd import ops
    from semanticlidarunc_amd import salsanext as sn
    from semanticlidarunc_amd.metrics.ece import ECEAggregator
    from semanticlidarunc_amd.models.evaluator import IoUEvaluator
    from semanticlidarunc_amd.salsanext import SalsaNext
    from semanticlidarunc_amd.testing import seeded_model, synthetic_scan
    from semanticlidarunc_amd.utils.mc_dropout import mc_predict

    sn.set_conv_precision(args.precision)
    model = seeded_model(SalsaNext)
    sd_cpu = {k: v.clone() for k, v in model.state_dict().items()}
    model = model.to(dev)
    Hh, Ww, Tt = args.height, args.width, args.passes
    x_cpu, labels_cpu = synthetic_scan(args.scans, Hh, Ww, seed=1234 + rank)
    x, labels = x_cpu.to(dev), labels_cpu.to(dev)
    iou, ece = IoUEvaluator(NCLS), ECEAggregator(n_bins=15, mode="probs", ignore_index=0, max_samples=500000)
    torch.manual_seed(100 + rank)

    def step(share_prefix=args.shared_prefix):
        p_bar, h_norm, mi_norm, preds = mc_predict(model, [x], T=Tt, share_prefix=share_prefix)
        iou.update(preds, labels)
        ece.update(p_bar, labels)
        return h_norm

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    iou.reset(); ece.reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.all_reduce(iou.confmat, op=dist.ReduceOp.SUM)            # the evaluation's one exchange
    miou, _ = iou.compute([str(i) for i in range(NCLS)], test_mask=[0] + [1] * (NCLS - 1), ignore_gt=[0])
    (ece_v, _), _ = ece.compute()[:2]

    # ---- roofline of the dominant kernel: one extra step with HIP events around every conv launch ----
    ops.TIMING, ops.TIMING_TAGS = [], []
    step()
    torch.cuda.synchronize()
    per_kernel = {}
    for name, flops, nbytes, e0, e1 in ops.TIMING:
        k = per_kernel.setdefault(name, [0, 0.0, 0.0, 0.0])
        k[0] += 1; k[1] += flops; k[2] += nbytes; k[3] += e0.elapsed_time(e1) * 1e-3
    if args.breakdown and rank == 0:
        with open(args.breakdown, "w") as f:
            f.write("per kernel instantiation (one measured step)\n")
            for name, k in sorted(per_kernel.items(), key=lambda kv: -kv[1][3]):
                f.write(f"{k[3]*1e3:9.3f} ms  n={k[0]:3d}  {k[1]/k[3]/1e12:7.2f} TF/s  {k[2]/k[3]/1e9:8.1f} GB/s  {name}\n")
            f.write("\nper launch, in launch order (name, ms, TFLOP/s, GB/s algorithmic, shape tag)\n")
            for (name, flops, nbytes, e0, e1), tag in zip(ops.TIMING, ops.TIMING_TAGS):
                ms = e0.elapsed_time(e1)
                f.write(f"{ms:8.3f} ms {flops/ms/1e9:7.2f} TF/s {nbytes/ms/1e6:8.1f} GB/s  {tag}  {name}\n")
    ops.TIMING = None
    conv_s = sum(k[3] for k in per_kernel.values())
    conv_flops = sum(k[1] for k in per_kernel.values())
    ranked = sorted(per_kernel, key=lambda n: -per_kernel[n][3])
    dom = ranked[0]
    n_l, fl, by, sec = per_kernel[dom]
    conv_bytes = sum(k[2] for k in per_kernel.values())

    def roof(name):
        """Which roof binds a kernel (algorithmic intensity of its launches vs the ridge of its MFMA path) and how close it gets."""
        _, k_fl, k_by, k_sec = per_kernel[name]
        peak = F16_MFMA_PEAK_TFLOPS if "h8" in name else (F16X3_MFMA_PEAK_TFLOPS if "f16x3" in name else FP32_MFMA_PEAK_TFLOPS)
        if k_fl / k_by >= peak * 1e12 / (HBM_PEAK_GBS * 1e9):
            a = k_fl / k_sec / 1e12
            return {"bound": "mfma", "kernel": name, "achieved": round(a, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(a / peak, 4)}, peak
        a = k_by / k_sec / 1e9
        return {"bound": "hbm", "kernel": name, "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4)}, peak

    roofline, mfma_peak = roof(dom)
    ridge = mfma_peak * 1e12 / (HBM_PEAK_GBS * 1e9)
    # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes (cannot be collected inside this process):
    # 2 x FETCH_SIZE + WRITE_SIZE per launch (gfx950 correction of MI355X_MICROARCH.md), only when the workload matches
    traffic, traffic_src = None, None
    pmc_file = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r01", "pmc_f16_traffic_N64.json")
    if args.precision == "f16" and (Hh, Ww) == (H, W) and not args.shared_prefix and os.path.exists(pmc_file):
        with open(pmc_file) as f:
            pmc = json.load(f)
        if pmc.get("_meta", {}).get("images") == args.scans * Tt and dom in pmc and pmc[dom].get("launches") == n_l:
            traffic = int((2.0 * pmc[dom]["FETCH_SIZE"] + pmc[dom]["WRITE_SIZE"]) * 1024)
            traffic_src = "bytes per launch; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes in profiles/r01/pmc_f16_traffic_N64.json"
    roofline.update({"traffic": traffic, "traffic_note": traffic_src, "launches_per_step": n_l, "avg_launch_us": round(sec / n_l * 1e6, 1),
                     "algorithmic_gflop_per_launch": round(fl / n_l / 1e9, 3), "algorithmic_mb_per_launch": round(by / n_l / 1e6, 2),
                     "intensity_flop_per_byte": round(fl / by, 1), "ridge_flop_per_byte": round(ridge, 1),
                     "all_convs": {"tflops": round(conv_flops / conv_s / 1e12, 2), "gbs": round(conv_bytes / conv_s / 1e9, 1),
                                   "ms_per_step": round(conv_s * 1e3, 2),
                                   "hbm_frac": round(conv_bytes / conv_s / 1e9 / HBM_PEAK_GBS, 4)}})
    if len(ranked) > 1:      # the two largest kernels are within a few percent of each other in total time: report the runner-up as well
        second, _ = roof(ranked[1])
        second.update({"launches_per_step": per_kernel[ranked[1]][0], "avg_launch_us": round(per_kernel[ranked[1]][3] / per_kernel[ranked[1]][0] * 1e6, 1),
                       "ms_per_step": round(per_kernel[ranked[1]][3] * 1e3, 3)})
        roofline["ms_per_step"] = round(sec * 1e3, 3)
        roofline["second"] = second

    if rank == 0:
        scans = args.scans * world * args.steps
        out = {
            "metric": f"range-image scans/sec ({Hh}x{Ww}, T={Tt} MC)", "value": round(scans / dt, 3), "unit": "scans/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"fp32": "f32", "f16x3": "f16x3 (fp32 I/O and accumulate; products as 3 split-fp16 MFMAs)",
                      "f16": "f16 (fp16 activations/weights, fp32 accumulate + epilogue, fp32 logits)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": f"SalsaNext MC-dropout T={Tt} + entropy/MI map + IoU/ECE accumulation, "
                                   f"{args.scans} scans of {Hh}x{Ww}x5 per step per GPU "
                                   f"({'BASELINE configs[2] shape' if (Hh, Ww, Tt) == (H, W, T) else 'non-default shape'})",
                       "scans_per_step_per_gpu": args.scans, "T": Tt, "parallelism": f"scan-sharded x{world}",
                       "mc_schedule": "shared deterministic prefix (3 context blocks + resBlock1 + resBlock2 convs once per scan)"
                                      if args.shared_prefix else "every pass fully recomputed"},
            "parity": {"mIoU_random_labels": round(miou, 6), "ece": round(ece_v, 6)},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd_cpu, x_cpu, threads=min(16, os.cpu_count() or 1), passes=Tt)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
