#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the evidence bundle of a round -- bench lines of every configuration (with roofline and
# cpu_baseline), rocprofv3 kernel statistics + gap analysis of the headline bench, PMC traffic (n = 100k and, with "big", n = 1M).
# Everything lands in gpurun_out/<tag>_*; copy what is to be judged into profiles/.  Every step has its own timeout and reads no stdin.
#   bash tools/collect_profiles.sh <tag> [quick|big]
tag=${1:-r04}; mode=$2
cd "$(dirname "$0")/.." && repo=$(pwd)
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python bench.py --steps 10 --warmup 2 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err < /dev/null; tail -c 400 gpurun_out/${tag}_bench.json; echo
summ() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2])); c = d["config"]; cb = d.get("cpu_baseline", {})
    print("%s: ms/step %.1f first cold %.1f ms (%s it) mean iters %.0f pcg/it %.2f launches/PCG-iteration %s frac %.3f precond %s | cpu: %s it/s, tts %s ms (%s)" % (sys.argv[1], d["ms_per_step"], c["first_cold_solve_ms"], c["first_cold_solve_admm_iters"], c["mean_admm_iters_per_step"], c["pcg_iters_per_admm_iter"], c["pcg_kernels_per_iteration"], d["roofline"]["pcg_iteration"]["frac"], c.get("preconditioner"), cb.get("value"), cb.get("time_to_solution_ms"), cb.get("sample", "")[:90]))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
}
summ banded gpurun_out/${tag}_bench.json
if [ "$mode" != quick ]; then
  for cfg in mixed shuffled unstructured portfolio lasso; do
    st=5; [ $cfg = lasso ] && st=2
    timeout 900 python bench.py --config $cfg --steps $st --warmup 1 --cpu-seconds 30 --batch 0 > gpurun_out/${tag}_bench_$cfg.json 2>> gpurun_out/${tag}_bench.err < /dev/null
    summ $cfg gpurun_out/${tag}_bench_$cfg.json
  done
fi
# rocprofv3 kernel statistics + where the GPU idles
out=/tmp/prof_$tag; rm -rf $out; mkdir -p $out
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bench -- python $repo/bench.py --cpu-seconds 0 --batch 0 --steps 5 --warmup 1 > $out/bench_stdout.log 2>&1 < /dev/null)
grep '^{' $out/bench_stdout.log | tail -1 > gpurun_out/${tag}_bench_under_rocprof.json
f=$(find $out -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${tag}_rocprofv3_kernel_stats.csv
t=$(find $out -name '*kernel_trace.csv' | head -1)
if [ -n "$t" ]; then python profiles/active_stats.py "$t" gpurun_out/${tag}_rocprofv3_active_stats.csv > /dev/null; python tools/gap_analysis.py "$t" > gpurun_out/${tag}_gap_analysis.txt; cat gpurun_out/${tag}_gap_analysis.txt; fi
[ -f gpurun_out/${tag}_rocprofv3_kernel_stats.csv ] && head -8 gpurun_out/${tag}_rocprofv3_kernel_stats.csv
# PMC traffic (separate passes, counters only with --kernel-trace)
timeout 900 bash profiles/run_pmc.sh $tag banded_n100000 --steps 2 --warmup 1 --batch 0 < /dev/null 2>&1 | grep -E "k_f1_probe|k_slot1|counter file" | head
if [ "$mode" = big ]; then
  timeout 900 python bench.py --n 1000000 --steps 3 --warmup 1 --cpu-seconds 0 --batch 0 > gpurun_out/${tag}_bench_n1M.json 2>> gpurun_out/${tag}_bench.err < /dev/null; summ n1M gpurun_out/${tag}_bench_n1M.json
  timeout 900 bash tools/prof_config.sh ${tag}_n1M --n 1000000 --steps 2 --warmup 1 < /dev/null | head -6
  timeout 900 bash profiles/run_pmc.sh $tag banded_n1000000 --n 1000000 --steps 1 --warmup 0 --batch 0 < /dev/null 2>&1 | grep -E "k_f1_probe|k_slot1" | head -4
  timeout 600 bash tools/prof_config.sh ${tag}_portfolio --config portfolio --steps 2 --warmup 1 < /dev/null | grep -E "k_wbx|k_wb_S"
  timeout 600 bash tools/prof_config.sh ${tag}_mixed --config mixed --steps 2 --warmup 1 < /dev/null | grep -E "k_slot1" | head -3
  # the batch path (configs[4]): bench line, where a problem's time goes, and the per-block mixing A/B + fuzz
  timeout 600 python bench_batch.py --steps 10 > gpurun_out/${tag}_bench_batch.json 2>> gpurun_out/${tag}_bench.err < /dev/null; tail -c 300 gpurun_out/${tag}_bench_batch.json; echo
  timeout 300 python tools/batch_trace.py 256 2>/dev/null | tail -4 > gpurun_out/${tag}_batch_trace.txt; timeout 300 python tools/batch_trace.py 4096 2>/dev/null | tail -4 >> gpurun_out/${tag}_batch_trace.txt
  timeout 600 python tools/mix_ab.py 0.02 2>/dev/null | grep iters > gpurun_out/${tag}_mix_ab.txt; cat gpurun_out/${tag}_mix_ab.txt
  timeout 900 python tools/mix_fuzz.py 15 2 2>/dev/null | grep -E "^ok|^BAD|cases" > gpurun_out/${tag}_mix_fuzz_seed2.txt; tail -1 gpurun_out/${tag}_mix_fuzz_seed2.txt
fi
