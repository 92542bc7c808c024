#!/usr/bin/env bash
# One GPU-box pass: bench line + per-launch breakdown, rocprofv3 kernel stats of the same command, PMC traffic passes.
# Usage (from the repo root on the GPU box):  bash tools/gpu_profile.sh [outdir]     -> files to copy into profiles/rNN/
set -eo pipefail
OUT=${1:-gpurun_out/prof}
mkdir -p "$OUT"
export TMPDIR=/tmp
sha256sum semanticlidarunc_amd/libslu_hip.so > "$OUT/libslu_hip.sha256"    # every number below comes from THIS build
echo "[1/6] bench.py (default)"; timeout -k 10 600 python bench.py --breakdown "$OUT/bench_f16_conv_breakdown_hipevents.txt" > "$OUT/bench_f16_line.json" 2> "$OUT/bench_f16.err"
tail -c 600 "$OUT/bench_f16_line.json"; echo
echo "[2/6] rocprofv3 --kernel-trace --stats"; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -o bench -- python3 bench.py --no-cpu-baseline --no-train-step --no-shared-prefix > "$OUT/bench_f16_line_under_rocprof.json" 2> "$OUT/rocprof_stats.err"
cp "$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1)" "$OUT/bench_f16_kernel_stats.csv"
echo "[3/6] PMC FETCH_SIZE"; SLU_CONV_PRECISION=f16 timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_rd" -- python3 tools/prof_forward.py 64 2 mc > "$OUT/pmc_rd.log" 2>&1
echo "[4/6] PMC WRITE_SIZE"; SLU_CONV_PRECISION=f16 timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/pmc_wr" -- python3 tools/prof_forward.py 64 2 mc > "$OUT/pmc_wr.log" 2>&1
echo "[5/6] tables"; python tools/pmc_table.py "$OUT/pmc_rd" "$OUT/pmc_wr" --json="$OUT/pmc_f16_traffic_N64.json" --meta-images=64 > "$OUT/pmc_f16_forward_N64_fetch_write.txt"
rm -rf "$OUT/stats" "$OUT/pmc_rd" "$OUT/pmc_wr"
echo "[6/6] training step: rocprofv3 kernel stats of tools/train_bench.py, weight-gradient kernels per shape"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/tstats" -o train -- python3 tools/train_bench.py > "$OUT/train_line_under_rocprof.json" 2> "$OUT/train_rocprof.err"
cp "$(find "$OUT/tstats" -name '*kernel_stats.csv' | head -1)" "$OUT/train_kernel_stats.csv"
rm -rf "$OUT/tstats"
timeout -k 10 300 python tools/wgrad_bench.py > "$OUT/train_wgrad_per_shape.txt" 2>&1
echo "[7] socket power / sclk while the two benches run (rocm-smi every 0.25 s)"
bash tools/power_trace.sh "$OUT/power_trace_bench.txt" timeout -k 10 300 python bench.py --no-cpu-baseline --no-train-step --no-shared-prefix --steps 200 --warmup 3 > "$OUT/power_trace_bench_line.json" 2> /dev/null
bash tools/power_trace.sh "$OUT/power_trace_train.txt" timeout -k 10 300 python tools/train_bench.py --batch 4 --steps 150 --warmup 3 > "$OUT/power_trace_train_line.json" 2> /dev/null
echo "[8] batch-size sweep of the inference step; fp32 conv kernel phase clocks (needs ab_libs/libslu_convprof.so)"
for s in 1 2 4 16; do timeout -k 10 200 python bench.py --scans $s --no-cpu-baseline --no-train-step --no-shared-prefix 2> /dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($s, 'scans/step:', d['value'], 'scans/s', d['ms_per_step'], 'ms/step')"; done > "$OUT/bench_f16_batch_sweep.txt"
if [ -f ab_libs/libslu_convprof.so ]; then timeout -k 10 300 python tools/conv_phase_prof.py ab_libs/libslu_convprof.so 4 > "$OUT/train_conv_phase_clocks.txt" 2>&1; fi
echo done
