# A/B builds of the library inside ONE gpurun call: stem kernel time (rocprofv3 kernel stats, 4 forwards of 1024 tiles).
# usage: bash tools/ab_stem.sh base exp ...  (expects deephisto_amd/libdeephisto_hip_<name>.so files)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd $R
for v in "$@"; do
  cp deephisto_amd/libdeephisto_hip_$v.so deephisto_amd/libdeephisto_hip.so
  rm -rf $O/abs_$v
  rocprofv3 --output-format csv --kernel-trace --stats -d $O/abs_$v -o s -- python3 tools/fwd_once.py 1024 4 > $O/abs_$v.log 2>&1
  V=$v python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/abs_" + os.environ["V"] + "/**/s_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "stem" in r["Name"]:
        print(f'{os.environ["V"]:10s} stem avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}')
PY
done
