export TMPDIR=/tmp
out=/tmp/prof_lasso; rm -rf $out; mkdir -p $out
repo=$(pwd)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bench -- python $repo/bench.py --config lasso --cpu-seconds 0 --steps 1 --warmup 0 > $out/bench_stdout.log 2>&1)
grep '^{' $out/bench_stdout.log | tail -1 | cut -c1-400
f=$(find $out -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r03c_lasso_rocprofv3_kernel_stats.csv; head -16 "$f"
