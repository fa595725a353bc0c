#!/bin/bash
# dev tool: rocprofv3 kernel stats of one python script (first argument; the rest are its arguments)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
rm -rf gpurun_out/prof_script
timeout -k 10 250 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_script -o s -- python3 "$@" > gpurun_out/prof_script.log 2>&1 || { tail -5 gpurun_out/prof_script.log; exit 1; }
tail -3 gpurun_out/prof_script.log
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_script/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:10]: print(r["Name"][:110], r["Calls"], r["AverageNs"], r["Percentage"])
PY
