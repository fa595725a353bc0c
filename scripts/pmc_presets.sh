#!/bin/bash
# dev tool: one rocprofv3 --pmc pass per argument group over scripts/preset_bench.py $PRESET; summary for kernels matching $KEY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_presets/p$i -o pmc -- python3 ${SCRIPT:-scripts/preset_bench.py} $PRESET > gpurun_out/pmc_presets_$i.log 2>&1 || { tail -5 gpurun_out/pmc_presets_$i.log; exit 1; }
  f=$(find gpurun_out/pmc_presets/p$i -name '*counter_collection.csv' | head -1)
  python3 scripts/pmc_summary.py "$KEY" $f
done
