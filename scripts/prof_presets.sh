#!/bin/bash
# dev tool: rocprofv3 kernel stats of scripts/preset_bench.py for the given presets
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_presets -o presets -- python3 scripts/preset_bench.py "$@" > gpurun_out/presets.log 2>&1 || { tail -5 gpurun_out/presets.log; exit 1; }
grep frames/s gpurun_out/presets.log
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_presets/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]: print(r["Name"][:100], r["Calls"], r["AverageNs"], r["Percentage"])
PY
