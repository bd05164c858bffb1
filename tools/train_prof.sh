set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
rm -rf $O/prof_train2
rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_train2 -o t -- python3 tools/train_profile.py ${1:-resnet18} > $O/train_prof.log 2>&1
python3 - <<'PY'
import csv, os
f = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/prof_train2/t_kernel_stats.csv"
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:34]:
    print(f'{r["Name"][:100]:100s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:9.1f} {float(r["Percentage"]):5.1f}%')
print("total ms", tot / 1e6)
PY
