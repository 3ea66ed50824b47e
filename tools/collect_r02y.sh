#!/bin/bash
# Run on the GPU box (via gpurun): the second half of round 2's measurements — the edge-group lookup kernel (bench line, phase
# statistics, ablation builds, HBM-side counters), LDS occupancy microbenchmark, altcorr backward (both paths, kernel traces, phase
# trace), Patchifier, the training step with the real module tree (json, per-iteration timing, kernel trace).
# Output: gpurun_out/<tag>/; copy the summaries into profiles/.
set -u
TAG=${1:-r02y}
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$TAG
rm -rf "$O"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-f16 --dtype f16 --steps 100 --warmup 10"
# ---- edge-group kernel
DEVO_CORR_GROUP=1 timeout 300 $B > "$O/group_f16_bench.json" 2> "$O/group_f16_bench.err"
timeout 300 $B > "$O/peredge_f16_bench.json" 2> "$O/peredge_f16_bench.err"
timeout 200 python "$R/tools/group_stats.py" cfg2 2>&1 | grep -a "stats\|plan\|order" > "$O/group_stats.txt"
for v in NOPAIRS NOSCATTER NOLOAD NOEPI; do
  if [ -f "$R/devo_amd/lib/libdevo_$v.so" ]; then
    echo -n "$v: " >> "$O/group_ablation.txt"
    DEVO_CORR_GROUP=1 DEVO_LIB=$R/devo_amd/lib/libdevo_$v.so timeout 200 python "$R/tools/bench_with_lib.py" --dtype f16 --no-f16 --steps 100 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'it/s,', d['roofline']['us_per_launch'], 'us per lookup launch')" >> "$O/group_ablation.txt"
  fi
done
DEVO_CORR_GROUP=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/group_trace" -o k -- $B > /dev/null 2> "$O/group_trace.log"
python "$R/tools/rocprof_summary.py" "$O/group_trace" > "$O/group_kernel_trace.txt" 2>&1
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  DEVO_CORR_GROUP=1 timeout 200 rocprofv3 --pmc $C --output-format csv -d "$O/group_pmc$i" -o p -- python "$R/tools/profile_corr.py" --reps 3 --dtype f16 > "$O/group_pmc$i.log" 2>&1
done
python "$R/tools/rocprof_summary.py" "$O" corr_fwd 2>&1 | grep -v "kernel_trace.csv" > "$O/group_pmc_corr_fwd.txt"
timeout 100 "$R/tools/ubench/lds_occupancy" > "$O/lds_occupancy.txt" 2>&1
# ---- altcorr backward
for m in atomic seg; do
  E=""; [ $m = seg ] && E="DEVO_CORR_BWD_SEG=1"
  env $E timeout 200 python "$R/tools/bench_corr_backward.py" 2>&1 | grep -a level >> "$O/corr_backward.txt"
  env $E timeout 200 python "$R/tools/bench_corr_backward.py" 1.0 2>&1 | grep -a level >> "$O/corr_backward.txt"
  env $E DEVO_CORR_BWD_TRACE=1 timeout 200 python "$R/tools/bench_corr_backward.py" 2>&1 | grep -a trace | sed -n "4p;30p" >> "$O/corr_backward.txt"
  env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/bwd_trace_$m" -o k -- python "$R/tools/bench_corr_backward.py" > /dev/null 2>&1
  python "$R/tools/rocprof_summary.py" "$O/bwd_trace_$m" 2>&1 | head -8 >> "$O/corr_backward_kernel_trace.txt"
done
# ---- Patchifier, training
timeout 300 python "$R/tools/bench_patchifier.py" 2>&1 | grep -v amdgpu.ids > "$O/patchifier.txt"
timeout 300 python "$R/tools/bench_training_step.py" > "$O/training_step.txt" 2>&1
timeout 300 python "$R/tools/bench_ba_train.py" > "$O/ba_train_step.txt" 2>&1
timeout 600 python "$R/bench.py" --mode train --steps 3 --warmup 1 > "$O/train_mode.json" 2> "$O/train_mode.err"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/train_trace" -o k -- python "$R/bench.py" --mode train --steps 2 --warmup 1 --train-iters 6 > /dev/null 2> "$O/train_trace.log"
python "$R/tools/rocprof_summary.py" "$O/train_trace" 2>&1 | head -45 > "$O/train_kernel_trace.txt"
ls "$O"
