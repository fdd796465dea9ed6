# kernel timeline of a few block steps of the lasso's Woodbury factorisation (dense_hip.hip dense_spd_inverse): run on the GPU box from the repo root
export TMPDIR=/tmp; R=$(pwd); cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/lp -o lasso -- python $R/tools/lasso_factor_prof.py 2>&1 | grep -v "^[EW]2" | tail -3
f=$(find /tmp/lp -name "*kernel_stats.csv" | head -1); cp $f $R/gpurun_out/${1:-r06}_lasso_kernel_stats.csv; head -8 $f | cut -c1-160
t=$(find /tmp/lp -name "*kernel_trace.csv" | head -1); python - <<PY
import csv
rows=list(csv.DictReader(open("$t")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_gj_pivot" in r["Kernel_Name"]]
i0=idx[100]
base=int(rows[i0-3]["Start_Timestamp"])
for r in rows[i0-3:i0+16]:
    print("%8.1f %8.1f  s%-3s %s"%((int(r["Start_Timestamp"])-base)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r.get("Stream_Id","?"),r["Kernel_Name"].split("::")[-1][:60]))
PY
