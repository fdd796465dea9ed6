#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the evidence bundle of a round -- bench lines of every configuration (with roofline and
# cpu_baseline), rocprofv3 kernel statistics + gap analysis of the headline bench, PMC traffic.  Everything lands in gpurun_out/<tag>_*;
# copy what is to be judged into profiles/.
#   bash tools/collect_profiles.sh <tag> [quick]
tag=${1:-r03}; quick=$2
cd "$(dirname "$0")/.." && repo=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --steps 10 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -c 600 gpurun_out/${tag}_bench.json; echo
if [ -z "$quick" ]; then
  for cfg in unstructured portfolio lasso; do
    st=5; [ $cfg = lasso ] && st=2
    timeout 900 python bench.py --config $cfg --steps $st --warmup 1 --cpu-seconds 30 > gpurun_out/${tag}_bench_$cfg.json 2>> gpurun_out/${tag}_bench.err
    python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${tag}_bench_$cfg.json")); c = d["config"]; cb = d.get("cpu_baseline", {})
    print("$cfg: ms/step %.1f first cold %.1f ms (%s it) mean iters %.0f pcg/it %.2f pair frac %.3f | cpu: %s it/s, tts %s ms (%s)" % (d["ms_per_step"], c["first_cold_solve_ms"], c["first_cold_solve_admm_iters"], c["mean_admm_iters_per_step"], c["pcg_iters_per_admm_iter"], d["roofline"]["pcg_iteration"]["frac"], cb.get("value"), cb.get("time_to_solution_ms"), cb.get("sample", "")[:90]))
except Exception as e: print("$cfg ERR", e)
PY
  done
fi
# rocprofv3 kernel statistics + where the GPU idles
out=/tmp/prof_$tag; rm -rf $out; mkdir -p $out
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bench -- python $repo/bench.py --cpu-seconds 0 --steps 5 --warmup 1 > $out/bench_stdout.log 2>&1)
grep '^{' $out/bench_stdout.log | tail -1 > gpurun_out/${tag}_bench_under_rocprof.json
f=$(find $out -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${tag}_rocprofv3_kernel_stats.csv
t=$(find $out -name '*kernel_trace.csv' | head -1)
if [ -n "$t" ]; then python profiles/active_stats.py "$t" gpurun_out/${tag}_rocprofv3_active_stats.csv > /dev/null; python tools/gap_analysis.py "$t" > gpurun_out/${tag}_gap_analysis.txt; cat gpurun_out/${tag}_gap_analysis.txt; fi
head -8 gpurun_out/${tag}_rocprofv3_kernel_stats.csv
# PMC traffic (separate passes, counters only with --kernel-trace)
bash profiles/run_pmc.sh $tag banded_n100000 --steps 2 --warmup 1 2>&1 | grep -E "k_f1_probe|k_slot1|counter file" | head
