# Stem kernel timing on the GPU box: parity test, phase stamps at 1024 tiles, rocprofv3 kernel stats of 3 forwards of 1024 tiles.
#   gpurun -- 'bash tools/stem_time.sh'
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python3 -m pytest tests/test_gpu_resnet.py -x -q -m gpu -k "stem_pool or bf16_logits or large_launch" 2>&1 | tail -3
python3 tools/conv_stamps.py 1024 2>/dev/null | tail -1
rm -rf $O/prof_stem
rocprofv3 --output-format csv --kernel-trace --stats -d $O/prof_stem -o s -- python3 tools/fwd_once.py 1024 4 > $O/prof_stem.log 2>&1
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/prof_stem/**/s_kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r["Percentage"]) > 3.0 or "stem" in r["Name"]:
        print(f'{r["Name"][:90]:90s} {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  min {float(r["MinNs"])/1e3:9.1f}  {float(r["Percentage"]):5.1f}%')
PY
