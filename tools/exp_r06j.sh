#!/bin/bash
# bounded attempt (verdict r05 item 5): resident edges per CU against time and FETCH_SIZE of the stress lookup
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/r06j; rm -rf "$O"; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp
for pad in 0 20000 34000 47000 60000; do
  export DEVO_CORR_LDS_PAD=$pad
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/k_$pad" -o k -- python "$R/tools/profile_corr.py" --reps 6 --workload stress > "$O/k_$pad.log" 2>&1
  t=$(python "$R/tools/rocprof_summary.py" "$O/k_$pad" 2>&1 | grep "corr_fwd_mm" | grep -v "^#" | head -1 | awk "{print \$3}")
  timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/p_$pad" -o p -- python "$R/tools/profile_corr.py" --reps 3 --workload stress > "$O/p_$pad.log" 2>&1
  f=$(python - <<PY
import csv,glob
tot=0;n=0
for fn in glob.glob("$O/p_$pad/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "corr_fwd_mm" in r.get("Kernel_Name","") and r.get("Counter_Name")=="FETCH_SIZE":
            tot+=float(r["Counter_Value"]); n+=1
print(f"{tot/max(n,1)*2/1e6:.2f} GB x2-corrected ({n} launches)")
PY
)
  echo "pad=$pad  avg_us=$t  fetch=$f"
  rm -rf "$O/k_$pad" "$O/p_$pad"
done | tee "$O/stress_lds_pad.txt"
