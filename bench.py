#!/usr/bin/env python
"""Headline benchmark: MC-dropout range-image scans/s (64x2048x5, T=8) on MI355X.

One "step" = one pass of the hot path over one batch of synthetic scans resident in HBM:
  T=8 stochastic SalsaNext forwards per scan (Dropout2d live, BatchNorm frozen)  ->  fused
  softmax / mean-over-T / predictive-entropy / mutual-information / argmax reduction  ->
  on-device confusion-matrix and ECE-bin accumulation.
`value` = scans of all ranks / max-over-ranks wall time.  Scans are independent, so ranks shard them
with no data-path collective ("weak" scaling); the 20x20 confusion matrix and the ECE bins are all-reduced
(RCCL) once after the timed region, as an evaluation run would.

    python bench.py [--gpus N --steps K --warmup W]

`--gpus N` ALWAYS means N ranks: under torch.distributed.run (the driver's launch line) WORLD_SIZE must equal N; started
plainly with N > 1 this script launches `python -m torch.distributed.run --nproc-per-node N` on itself BEFORE touching the
GPU, relays rank 0's JSON line and exits with the launcher's code.  Any other combination exits non-zero.
"""
import argparse
import hashlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W, T, NCLS = 64, 2048, 8, 20
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
F16_MFMA_PEAK_TFLOPS = 2500.0       # v_mfma_f32_32x32x16_f16, dense
F16X3_MFMA_PEAK_TFLOPS = 2500.0 / 3  # three dense f16 MFMAs (2.5 PFLOP/s) per fp32-class product
HBM_PEAK_GBS = 8000.0
PMC_FILE = os.path.join(ROOT, "profiles", "r03", "pmc_f16_traffic_N64.json")
LIB_FILE = os.path.join(ROOT, "semanticlidarunc_amd", "libslu_hip.so")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--scans", type=int, default=8, help="scans per step per GPU (the reference config batch size for SalsaNext at 64x2048, SemanticKitti_default.yaml:75-78)")
    ap.add_argument("--height", type=int, default=H, help="range-image rows (default: the metric's 64; 128 = BASELINE configs[4])")
    ap.add_argument("--width", type=int, default=W, help="range-image columns (default: the metric's 2048; 4096 = configs[4])")
    ap.add_argument("--passes", type=int, default=T, help="MC passes T (default: the metric's 8; 16 = configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the host-CPU oracle timing (and the parity block that shares its pass)")
    ap.add_argument("--no-train-step", action="store_true", help="skip the extra BASELINE configs[1] training-step measurement")
    ap.add_argument("--no-shared-prefix", action="store_true",
                    help="skip the extra legs that launch the same kernels on other shapes -- shared prefix, B = 1 stream, capped ECE -- so that a "
                         "rocprofv3 --kernel-trace --stats run of this script contains strict steps only")
    ap.add_argument("--shared-prefix", action="store_true",
                    help="MC schedule that computes the layers no active Dropout2d can reach once per scan instead of T times "
                         "(bit-identical outputs); default: every pass fully recomputed")
    ap.add_argument("--precision", default="f16", choices=["fp32", "f16x3", "f16"],
                    help="conv precision: exact fp32 MFMA; split-fp16 (fp32 storage, 3 f16 MFMAs, fp32 accumulate); "
                         "f16 = fp16 storage + 1 f16 MFMA, fp32 accumulate (BASELINE configs[2] names half-precision storage)")
    ap.add_argument("--breakdown", default=None, help="write a per-kernel / per-layer-shape timing table to this file")
    ap.add_argument("--stub-cpu", action="store_true",
                    help="launcher self-test (tests/test_distributed_cpu.py): gloo ranks on the CPU and a trivial step; no GPU, no HIP library")
    return ap.parse_args(argv)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def ensure_ranks(args, argv):
    """Make `--gpus N` mean N ranks, decided before anything touches the GPU.  Returns (rank, local_rank, world) for a rank that
    should run; never returns in the launcher process (it exits with the children's code)."""
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit(f"--gpus {args.gpus}: need at least one rank")
    if env_world is None:
        if args.gpus == 1:
            return 0, 0, 1
        # plain `python bench.py --gpus N`: start N fresh rank processes (this process never initialises the GPU)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        sys.stderr.write(f"bench.py: --gpus {args.gpus} without a torchrun environment -> launching {args.gpus} ranks: {' '.join(cmd)}\n")
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    world = int(env_world)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to report a line for a different rank count")
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), world


# ---------------------------------------------------------------------------------------------------------------
# host CPU facts and the CPU baseline (the oracle, timed on the GPU box's own cores)
# ---------------------------------------------------------------------------------------------------------------
def host_cpu_info():
    """CPU model, physical cores (unique (socket, core) pairs), logical CPUs, the CPUs this process may use (affinity and cgroup
    quota), and the thread count the baseline therefore runs with."""
    model, pairs, logical = None, set(), 0
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("processor"):
                    logical += 1
                elif line.startswith("model name") and model is None:
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    logical = logical or (os.cpu_count() or 1)
    physical = len(pairs) or max(1, logical // 2)
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else logical
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()
            if q != "max":
                quota = max(1, int(int(q) / int(period)))
    except (OSError, ValueError):
        pass
    threads = max(1, min([physical, usable] + ([quota] if quota else [])))
    return {"cpu_model": model, "physical_cores": physical, "logical_cpus": logical, "usable_cpus": usable, "cgroup_cpu_quota": quota,
            "threads": threads}


def cpu_baseline_and_parity_reference(model_sd, x_cpu, passes):
    """The oracle (CPU restatement, pinned against the reference) on the host cores, as SURVEY 8(d) / BASELINE.md section 3
    prescribe: threads = the physical cores this process may use, 2 warm-up forwards, then the median of 5 for (i) one eval forward
    (BASELINE configs[0]) and (ii) one MC scan = T stochastic passes + the reduction (the metric's unit).
    Also returns the oracle's outputs of the first timed MC scan (with its dropout multipliers) for the parity block."""
    import torch
    from oracle import salsanext as osalsa, uncertainty as ounc
    info = host_cpu_info()
    torch.set_num_threads(info["threads"])
    x1 = x_cpu[:1]
    single, mc, ref = [], [], None
    with torch.no_grad():
        for _ in range(2):
            osalsa.salsanext_forward(model_sd, x1)                   # warm-ups (oneDNN primitive cache, allocator)
        for _ in range(5):
            t0 = time.perf_counter()
            osalsa.salsanext_forward(model_sd, x1)
            single.append(time.perf_counter() - t0)
        for rep in range(5):
            g = torch.Generator().manual_seed(rep)
            t0 = time.perf_counter()
            scales = [osalsa.draw_dropout_scales(1, 0.2, g) for _ in range(passes)]
            outs = [osalsa.salsanext_forward(model_sd, x1, s) for s in scales]
            red = ounc.mc_reduce(torch.stack(outs, 0))
            mc.append(time.perf_counter() - t0)
            if rep == 0:
                ref = (scales, red, osalsa.salsanext_forward(model_sd, x1).argmax(1))     # + the deterministic eval pass the parity labels use
    dt = statistics.median(mc)
    out = {"value": round(1.0 / dt, 4), "unit": "scans/s", "cores": info["threads"], "kind": "port",
           "sample": f"1 scan {x1.shape[2]}x{x1.shape[3]}x5: median of 5 MC scans (T={passes} oracle passes with dropout multipliers + MC reduction, "
                     f"{dt:.2f} s each) after 2 warm-up forwards; torch-CPU fp32 oracle",
           "single_pass_scans_per_s": round(1.0 / statistics.median(single), 3),
           "cpu_model": info["cpu_model"], "physical_cores": info["physical_cores"], "logical_cpus": info["logical_cpus"],
           "usable_cpus": info["usable_cpus"], "cgroup_cpu_quota": info["cgroup_cpu_quota"]}
    return out, ref


def parity_block(model, x_cpu, ref, passes, dev):
    """north_star parity on one full-size scan: (GPU network in the benchmark's precision -> GPU reduction -> GPU IoU / ECE) against
    (oracle fp32 network with the SAME dropout multipliers -> oracle reduction -> oracle IoU / ECE).  Labels = argmax of the oracle's
    deterministic eval pass with 30 % seeded label noise (empty returns -> class 0, ignored): far from trivial, and not tied to
    either MC path's own near-tie decisions."""
    import numpy as np
    import torch
    from oracle import metrics as ometrics
    from semanticlidarunc_amd.metrics.ece import ECEAggregator
    from semanticlidarunc_amd.models.evaluator import IoUEvaluator
    from semanticlidarunc_amd.utils.mc_dropout import dropout_sampling
    scales, (p_w, h_w, mi_w, pred_w), det = ref
    x1 = x_cpu[:1]
    g = torch.Generator().manual_seed(99)
    noisy = torch.rand(det.shape, generator=g) < 0.30
    labels = torch.where(noisy, torch.randint(1, NCLS, det.shape, generator=g), det)
    labels = labels.masked_fill(x1[:, 0] == 0, 0)
    stacked = {k: torch.cat([s[k] for s in scales], 0) for k in scales[0]}            # pass-major [T*1, C, 1, 1]
    model.eval()
    with torch.no_grad(), dropout_sampling(model, True):
        xg = x1.to(dev)
        if model.mc_fused_ok(xg, passes):
            p_g, h_g, mi_g, pred_g = model.mc_predict_fused(xg, passes, scales=stacked)
        else:
            from semanticlidarunc_amd import ops
            logits = model.forward_with_dropout_scales(xg.repeat(passes, 1, 1, 1), stacked)
            p_g, h_g, mi_g, pred_g = ops.mc_reduce(logits.reshape(passes, 1, *logits.shape[1:]).contiguous())
    iou, ece = IoUEvaluator(NCLS), ECEAggregator(n_bins=15, mode="probs", ignore_index=0, max_samples=500000)
    lab_g = labels.to(dev)
    iou.update(pred_g, lab_g)
    ece.update(p_g, lab_g)
    names, mask = [str(i) for i in range(NCLS)], [0] + [1] * (NCLS - 1)
    miou_g, _ = iou.compute(names, test_mask=mask, ignore_gt=[0])
    (ece_g, _), _ = ece.compute()[:2]
    cm = ometrics.confusion_matrix(pred_w.numpy(), labels.numpy(), NCLS)
    miou_w, _ = ometrics.iou_from_confusion(cm, mask, [0])
    conf, ok = ometrics.top_label(p_w.numpy(), labels.numpy(), 0, "probs")
    ece_w, _ = ometrics.ece_from_bins(*ometrics.ece_bins(conf, ok, 15))
    return {"scan": f"1x5x{x1.shape[2]}x{x1.shape[3]}, T={passes}, same dropout multipliers on both sides; labels = oracle eval-pass argmax + 30 % noise",
            "mIoU_gpu": round(miou_g, 6), "mIoU_oracle": round(miou_w, 6), "abs_dmIoU": round(abs(miou_g - miou_w), 7),
            "ece_gpu": round(ece_g, 6), "ece_oracle": round(ece_w, 6), "abs_dECE": round(abs(ece_g - ece_w), 7),
            "max_abs_dp_bar": round(float((p_g.cpu() - p_w).abs().max()), 7),
            "max_abs_dentropy_norm": round(float((h_g.cpu() - h_w).abs().max()), 7),
            "max_abs_dmi_norm": round(float((mi_g.cpu() - mi_w).abs().max()), 7),
            "argmax_flip_frac": round(float((pred_g.cpu() != pred_w).float().mean()), 7),
            "bar": 1e-3}


# ---------------------------------------------------------------------------------------------------------------
# roofline helpers
# ---------------------------------------------------------------------------------------------------------------
def lib_sha256():
    h = hashlib.sha256()
    with open(LIB_FILE, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def load_pmc(images):
    """Committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_table.py).  Only trusted when they were collected with
    THIS build of libslu_hip.so (sha256 recorded next to them) and the same number of stacked images."""
    if not os.path.exists(PMC_FILE):
        return None, "no PMC file committed for this round"
    with open(PMC_FILE) as f:
        pmc = json.load(f)
    meta = pmc.get("_meta", {})
    if meta.get("images") != images:
        return None, f"PMC passes were collected with {meta.get('images')} images per launch, this run uses {images}"
    if meta.get("lib_sha256") != lib_sha256():
        return None, "PMC passes were collected with a different build of libslu_hip.so (sha256 mismatch): traffic dropped"
    return pmc, os.path.relpath(PMC_FILE, ROOT)


def stub_main(args, rank, world):
    """Launcher self-test: N gloo ranks on the CPU, a trivial step, the same barrier / max-over-ranks / one-JSON-line protocol."""
    import torch
    import torch.distributed as dist
    if "RANK" in os.environ:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    x = torch.ones(1024)

    def barrier():
        if dist.is_initialized():
            dist.barrier()

    for _ in range(args.warmup):
        x = x * 1.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = x * 1.0
    barrier()
    dt = time.perf_counter() - t0
    got_world = 1
    if dist.is_initialized():
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        got_world = dist.get_world_size()
    if rank == 0:
        print(json.dumps({"metric": "stub steps/sec (launcher self-test, CPU, gloo)", "value": round(args.steps * world / max(dt, 1e-9), 3),
                          "unit": "steps/s", "n_gpus": world, "rccl_world_size": got_world, "steps": args.steps, "warmup": args.warmup,
                          "higher_is_better": True, "scaling": "weak", "data": "synthetic", "config": {"workload": "stub"}}), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    rank, local_rank, world = ensure_ranks(args, argv)
    if args.stub_cpu:
        return stub_main(args, rank, world)

    # stdout carries ONE JSON line (the driver's contract): libraries write banners there too (RCCL prints its version block on stdout at
    # communicator creation), so everything else that reaches file descriptor 1 in this process goes to stderr and the line is written to the
    # saved descriptor at the end
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if "RANK" in os.environ:      # launched by torch.distributed.run: use RCCL even for a single rank
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # RCCL over xGMI
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"RCCL communicator has {dist.get_world_size()} ranks, --gpus {args.gpus}")

    from semanticlidarunc_amd import ops
    from semanticlidarunc_amd import salsanext as sn
    from semanticlidarunc_amd.distributed import all_reduce_metrics
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
    # max_samples=None: the reference aggregator then keeps EVERY valid pixel (metrics/ece.py:88-91) -- the mirror's exact per-bin
    # accumulators -- which is also the only form whose evidence adds up across ranks.  The Trainer's capped form
    # (max_samples=500000: numpy-seeded reservoir, host-drawn indices) is mirrored and tested, and used by the parity block below.
    iou, ece = IoUEvaluator(NCLS), ECEAggregator(n_bins=15, mode="probs", ignore_index=0, max_samples=None)
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
    rccl_world = 1
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        rccl_world = dist.get_world_size()
        all_reduce_metrics(iou, ece, device=dev)                      # the evaluation's one exchange
    miou, _ = iou.compute([str(i) for i in range(NCLS)], test_mask=[0] + [1] * (NCLS - 1), ignore_gt=[0])
    (ece_v, _), _ = ece.compute()[:2]

    # ---- roofline of the dominant kernel: one extra step with HIP events around every conv launch ----
    ops.TIMING, ops.TIMING_TAGS = [], []
    step()
    torch.cuda.synchronize()
    per_kernel = {}       # name -> [launches, flops, layer-granular bytes, seconds, fused-minimum bytes]
    full_res = [0.0, 0.0, 0.0, 0.0]       # the 64x2048 (full-resolution) conv launches: flops, bytes, seconds, fused-minimum bytes
    for rec, tag in zip(ops.TIMING, ops.TIMING_TAGS):
        name, flops, nbytes, e0, e1 = rec[:5]
        min_bytes = rec[5] if len(rec) > 5 else nbytes
        sec = e0.elapsed_time(e1) * 1e-3
        k = per_kernel.setdefault(name, [0, 0.0, 0.0, 0.0, 0.0])
        k[0] += 1; k[1] += flops; k[2] += nbytes; k[3] += sec; k[4] += min_bytes
        if tag.endswith(f" {Hh}x{Ww}"):
            full_res[0] += flops; full_res[1] += nbytes; full_res[2] += sec; full_res[3] += min_bytes
    if args.breakdown and rank == 0:
        with open(args.breakdown, "w") as f:
            f.write("per kernel instantiation (one measured step): ms, launches, TFLOP/s, GB/s layer-granular (SURVEY 8(d)), GB/s of the kernel's own minimum bytes\n")
            for name, k in sorted(per_kernel.items(), key=lambda kv: -kv[1][3]):
                f.write(f"{k[3]*1e3:9.3f} ms  n={k[0]:3d}  {k[1]/k[3]/1e12:7.2f} TF/s  {k[2]/k[3]/1e9:8.1f} GB/s  {k[4]/k[3]/1e9:8.1f} GB/s(min)  {name}\n")
            f.write("\nper launch, in launch order (ms, TFLOP/s, GB/s layer-granular, GB/s fused minimum, shape tag, kernel)\n")
            for rec, tag in zip(ops.TIMING, ops.TIMING_TAGS):
                name, flops, nbytes, e0, e1 = rec[:5]
                mb = rec[5] if len(rec) > 5 else nbytes
                ms = e0.elapsed_time(e1)
                f.write(f"{ms:8.3f} ms {flops/ms/1e9:7.2f} TF/s {nbytes/ms/1e6:8.1f} GB/s {mb/ms/1e6:8.1f} GB/s(min)  {tag}  {name}\n")
    timing, timing_tags = ops.TIMING, ops.TIMING_TAGS
    ops.TIMING = None
    conv_s = sum(k[3] for k in per_kernel.values())
    conv_flops = sum(k[1] for k in per_kernel.values())
    conv_bytes = sum(k[2] for k in per_kernel.values())
    conv_min_bytes = sum(k[4] for k in per_kernel.values())
    ranked = sorted(per_kernel, key=lambda n: -per_kernel[n][3])
    pmc, pmc_note = (None, "traffic is only attached to the default f16 64x2048 workload")
    if args.precision == "f16" and (Hh, Ww) == (H, W) and not args.shared_prefix:
        pmc, pmc_note = load_pmc(args.scans * Tt)

    def kernel_traffic(name):
        """HBM bytes per launch from the PMC passes: 2 x FETCH_SIZE + WRITE_SIZE (KB; gfx950 correction of MI355X_MICROARCH.md)."""
        if pmc is None or name not in pmc or pmc[name].get("launches") != per_kernel[name][0]:
            return None
        return int((2.0 * pmc[name]["FETCH_SIZE"] + pmc[name]["WRITE_SIZE"]) * 1024)

    def launch_traffic():
        """PMC bytes of every timed conv launch, in launch order: the committed table lists the dispatches of one MC step in order
        (`_dispatches`); the i-th timed launch of a kernel name is the i-th dispatch of that name.  None when the table does not match."""
        if pmc is None or "_dispatches" not in pmc:
            return None
        by_name = {}
        for d in pmc["_dispatches"]:
            by_name.setdefault(d["name"], []).append(d)
        seen, out = {}, []
        for rec in timing:
            k = seen.get(rec[0], 0)
            seen[rec[0]] = k + 1
            lst = by_name.get(rec[0], [])
            if k >= len(lst):
                return None
            out.append((2.0 * lst[k]["FETCH_SIZE"] + lst[k]["WRITE_SIZE"]) * 1024)
        if any(seen[nm] != len(by_name.get(nm, [])) for nm in seen):
            return None
        return out

    def roof(name):
        """Which roof binds a kernel (algorithmic intensity of its launches vs the ridge of its MFMA path) and how close it gets.
        `frac` follows SURVEY 8(d)'s layer-granular bytes; `frac_fused_min` counts only what a fused kernel must move (its inputs,
        residual and outputs once); `frac_traffic` = measured PMC bytes / time."""
        n_l, k_fl, k_by, k_sec, k_min = per_kernel[name]
        peak = F16_MFMA_PEAK_TFLOPS if "h8" in name else (F16X3_MFMA_PEAK_TFLOPS if "f16x3" in name else FP32_MFMA_PEAK_TFLOPS)
        ridge = peak * 1e12 / (HBM_PEAK_GBS * 1e9)
        tr = kernel_traffic(name)
        common = {"kernel": name, "launches_per_step": n_l, "avg_launch_us": round(k_sec / n_l * 1e6, 1), "ms_per_step": round(k_sec * 1e3, 3),
                  "algorithmic_gflop_per_launch": round(k_fl / n_l / 1e9, 3), "algorithmic_mb_per_launch": round(k_by / n_l / 1e6, 2),
                  "fused_min_mb_per_launch": round(k_min / n_l / 1e6, 2),
                  "intensity_flop_per_byte": round(k_fl / k_by, 1), "ridge_flop_per_byte": round(ridge, 1),
                  "hbm_frac_layer_granular": round(k_by / k_sec / 1e9 / HBM_PEAK_GBS, 4),
                  "frac_fused_min": round(k_min / k_sec / 1e9 / HBM_PEAK_GBS, 4),
                  "frac_traffic": None if tr is None else round(tr * n_l / k_sec / 1e9 / HBM_PEAK_GBS, 4),
                  "traffic": tr,
                  "mfma_frac": round(k_fl / k_sec / 1e12 / peak, 4)}
        if k_by / k_sec / 1e9 > HBM_PEAK_GBS:
            common["note"] = ("layer-granular bytes / time exceeds the HBM peak: an accounting artefact of fusion (the kernel never moves the "
                              "intermediate tensor), not performance -- read frac_fused_min / frac_traffic")
        if k_fl / k_by >= ridge:
            a = k_fl / k_sec / 1e12
            return dict({"bound": "mfma", "achieved": round(a, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(a / peak, 4)}, **common)
        a = k_by / k_sec / 1e9
        return dict({"bound": "hbm", "achieved": round(a, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a / HBM_PEAK_GBS, 4)}, **common)

    roofline = roof(ranked[0])
    roofline["traffic_note"] = pmc_note
    roofline["all_convs"] = {"tflops": round(conv_flops / conv_s / 1e12, 2), "gbs": round(conv_bytes / conv_s / 1e9, 1),
                             "ms_per_step": round(conv_s * 1e3, 2), "hbm_frac": round(conv_bytes / conv_s / 1e9 / HBM_PEAK_GBS, 4),
                             "hbm_frac_fused_min": round(conv_min_bytes / conv_s / 1e9 / HBM_PEAK_GBS, 4)}
    if full_res[2] > 0:       # the north-star target's denominator: the full-resolution conv launches (context blocks, resBlock1, upBlock4, head)
        roofline[f"encoder_{Hh}x{Ww}"] = {"ms_per_step": round(full_res[2] * 1e3, 3), "tflops": round(full_res[0] / full_res[2] / 1e12, 2),
                                           "gbs": round(full_res[1] / full_res[2] / 1e9, 1),
                                           "hbm_frac": round(full_res[1] / full_res[2] / 1e9 / HBM_PEAK_GBS, 4),
                                           "hbm_frac_fused_min": round(full_res[3] / full_res[2] / 1e9 / HBM_PEAK_GBS, 4)}
    # what the step's conv launches would take if every one ran AT its binding roof: sum over launches of max(flops / MFMA peak,
    # fused-minimum bytes / 8 TB/s); frac = that bound / the measured time of the same launches (HIP events) -- the honest whole-step figure
    mfma_peak = {"f16": F16_MFMA_PEAK_TFLOPS, "f16x3": F16X3_MFMA_PEAK_TFLOPS, "fp32": FP32_MFMA_PEAK_TFLOPS}[args.precision] * 1e12
    bound_s = sum(max(rec[1] / mfma_peak, (rec[5] if len(rec) > 5 else rec[2]) / (HBM_PEAK_GBS * 1e9)) for rec in timing)
    roofline["step_roofline_bound"] = {"bound_ms": round(bound_s * 1e3, 3), "measured_conv_ms": round(conv_s * 1e3, 3),
                                       "frac": round(bound_s / conv_s, 4), "frac_of_step": round(bound_s / (dt / args.steps), 4),
                                       "definition": "sum over the step's conv launches of max(algorithmic flops / dense MFMA peak, fused-minimum bytes / 8 TB/s)"}
    lt = launch_traffic()
    if lt is not None:
        tot = sum(lt)
        roofline["all_convs"]["frac_traffic"] = round(tot / conv_s / 1e9 / HBM_PEAK_GBS, 4)
        fr = sum(t for t, tag in zip(lt, timing_tags) if tag.endswith(f" {Hh}x{Ww}"))
        if full_res[2] > 0:
            roofline[f"encoder_{Hh}x{Ww}"]["frac_traffic"] = round(fr / full_res[2] / 1e9 / HBM_PEAK_GBS, 4)
            roofline[f"encoder_{Hh}x{Ww}"]["traffic_gb_per_step"] = round(fr / 1e9, 3)
    roofline["kernels"] = [roof(n) for n in ranked[1:8]]
    if len(ranked) > 1:
        roofline["second"] = roofline["kernels"][0]
    roofline["over_unity_layer_granular"] = [n for n in ranked if per_kernel[n][2] / per_kernel[n][3] / 1e9 > HBM_PEAK_GBS]

    # the same workload under the opt-in MC schedule that computes the layers no active Dropout2d can reach once per scan (bit-identical
    # outputs, tests/test_gpu_model.py); reported beside the strict number, never as `value`
    shared = None
    if not args.shared_prefix and not args.no_shared_prefix and world == 1:
        for _ in range(2):
            step(True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step(True)
        torch.cuda.synchronize()
        dts = time.perf_counter() - t1
        shared = {"value": round(args.scans * args.steps / dts, 3), "unit": "scans/s", "ms_per_step": round(dts / args.steps * 1e3, 3),
                  "note": "shared deterministic prefix: 3 context blocks + resBlock1 + resBlock2's convs once per scan instead of T times; "
                          "outputs bit-identical to the strict schedule"}

    # B = 1 stream (inference_ouster.py processes one scan at a time): the same step on ONE scan (T stacked passes = 8 images per launch)
    latency = None
    if world == 1 and not args.no_shared_prefix and args.scans != 1:
        x1, l1 = x[:1].contiguous(), labels[:1].contiguous()

        def step1():
            p_bar, h_norm, mi_norm, preds = mc_predict(model, [x1], T=Tt, share_prefix=False)
            iou.update(preds, l1)
            ece.update(p_bar, l1)

        for _ in range(3):
            step1()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(max(args.steps, 10)):
            step1()
        torch.cuda.synchronize()
        lat = (time.perf_counter() - t1) / max(args.steps, 10)
        latency = {"latency_ms_per_scan": round(lat * 1e3, 3), "scans_per_s": round(1.0 / lat, 2),
                   "note": f"one scan per step (T = {Tt} stacked passes, every pass fully recomputed), same reduction and metric accumulation"}
        if args.precision == "f16" and not os.environ.get("SLU_BENCH_NO_B1_GRAPH"):      # the same stream with forward + head / MC reduction replayed as ONE HIP graph (graph_infer.py)
            try:
                from semanticlidarunc_amd.graph_infer import GraphedMCPredict
                stream = GraphedMCPredict(model, x1, T=Tt)

                def step1g():
                    p_bar, h_norm, mi_norm, preds = stream(x1)
                    iou.update(preds, l1)
                    ece.update(p_bar, l1)

                for _ in range(3):
                    step1g()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(max(args.steps, 10)):
                    step1g()
                torch.cuda.synchronize()
                latg = (time.perf_counter() - t1) / max(args.steps, 10)
                latency["hip_graph_latency_ms_per_scan"] = round(latg * 1e3, 3)
                del stream
            except Exception as e:      # an extra: never takes the line down
                latency["hip_graph_latency_ms_per_scan"] = f"error: {type(e).__name__}: {e}"

    # The Trainer's ECE configuration (trainer.py:215-222: max_samples = 500000 -> reservoir with host-drawn indices, one device sync per batch)
    # instead of the all-pixel bin accumulators of the headline loop: the same step, timed beside it
    capped = None
    if world == 1 and not args.no_shared_prefix:
        ece_cap = ECEAggregator(n_bins=15, mode="probs", ignore_index=0, max_samples=500000)

        def step_cap():
            p_bar, h_norm, mi_norm, preds = mc_predict(model, [x], T=Tt, share_prefix=args.shared_prefix)
            iou.update(preds, labels)
            ece_cap.update(p_bar, labels)

        for _ in range(2):
            step_cap()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step_cap()
        torch.cuda.synchronize()
        dtc = time.perf_counter() - t1
        capped = {"value": round(args.scans * args.steps / dtc, 3), "unit": "scans/s", "ms_per_step": round(dtc / args.steps * 1e3, 3),
                  "ece": "ECEAggregator(max_samples=500000): the reference Trainer's reservoir form (numpy-seeded draws on the host, boolean-mask "
                         "compaction, a device sync per batch)"}

    # BASELINE configs[1] (one rank) / configs[3] (data-parallel: batch 4 per GPU, the flat RCCL gradient all-reduce of SURVEY 8(e)) beside the
    # headline.  With several ranks EVERY rank takes part; the leg runs under a watchdog so that a stuck collective can never cost the headline line.
    train, train_hung = None, False
    if not args.no_train_step and (Hh, Ww) == (H, W):
        if world == 1 and not os.environ.get("SLU_BENCH_GUARDED_LEG"):      # (the variable: exercise the watchdog form on a one-GPU box)
            train = train_step_block(dev)
        else:
            train, train_hung = guarded_train_leg(dev, rank, world)

    if rank == 0:
        scans = args.scans * world * args.steps
        out = {
            "metric": f"range-image scans/sec ({Hh}x{Ww}, T={Tt} MC)", "value": round(scans / dt, 3), "unit": "scans/s",
            "n_gpus": world, "rccl_world_size": rccl_world, "steps": args.steps, "warmup": args.warmup,
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
                                      if args.shared_prefix else "every pass fully recomputed",
                       "ece_form": "ECEAggregator(max_samples=None): every valid pixel, exact per-bin accumulators on the device (no host sync); "
                                   "the Trainer's capped form is timed beside it as `ece_capped_500k`"},
            "metrics_of_the_timed_run": {"mIoU_random_labels": round(miou, 6), "ece_all_pixels": round(ece_v, 6),
                                         "note": "synthetic random labels: plumbing only; parity is the `parity` block"},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], ref = cpu_baseline_and_parity_reference(sd_cpu, x_cpu, Tt)
            out["parity"] = parity_block(model, x_cpu, ref, Tt, dev)
        if shared is not None:
            out["mc_shared_prefix"] = shared
        if latency is not None:
            out["b1_stream"] = latency
        if capped is not None:
            out["ece_capped_500k"] = capped
        if train is not None:
            out["train_step"] = train
        sys.stdout.flush()
        os.write(line_fd, (json.dumps(out) + "\n").encode())
    if train_hung:
        sys.stdout.flush()
        os._exit(0)              # a collective of the optional training leg is stuck: the line is out, do not wait in destroy_process_group
    if dist is not None:
        dist.destroy_process_group()


def train_step_block(dev, rank=0, world=1):
    """BASELINE configs[1] beside the headline: batch 4 of 64x2048 per GPU, fwd + NLL/Lovasz loss + bwd [+ flat RCCL gradient all-reduce] + AdamW,
    fp32 (configs[3] when world > 1: global batch 4 x world)."""
    try:
        from tools.train_bench import measure
    except Exception as e:      # the extra block must never take the headline line down
        return {"error": f"{type(e).__name__}: {e}"}
    try:
        res = measure(dev, batch=4, steps=5, warmup=2, rank=rank, world=world)
    except Exception as e:
        return {"error": f"{type(e).__name__}: {e}"}
    if world == 1:      # the same step replayed as ONE HIP graph, beside the eager number (GPU-bound step: the graph form is not faster here)
        try:
            g = measure(dev, batch=4, steps=5, warmup=2, graph=True)
            res["hip_graph_variant"] = {"ms_per_step": g["ms_per_step"], "scans_per_s": g["value"]}
        except Exception as e:
            res["hip_graph_variant"] = {"error": f"{type(e).__name__}: {e}"}
    return res


def guarded_train_leg(dev, rank, world, timeout_s=240.0):
    """(result, hung): the multi-rank training leg in a worker thread with a deadline."""
    import threading
    import torch
    box = {}

    def run():
        torch.cuda.set_device(dev)
        box["r"] = train_step_block(dev, rank, world)

    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(timeout_s)
    if th.is_alive():
        return {"error": f"the data-parallel training leg did not finish within {timeout_s:.0f} s on rank {rank} (skipped)"}, True
    return box.get("r"), False


if __name__ == "__main__":
    main()
