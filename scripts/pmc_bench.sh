#!/bin/bash
# dev tool: one rocprofv3 --pmc pass per argument group over bench.py (short run); summary for kernels matching $KEY
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $grp --output-format csv -d gpurun_out/pmc_bench/p$i -o pmc -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/pmc_bench_$i.log 2>&1 || { tail -5 gpurun_out/pmc_bench_$i.log; exit 1; }
  f=$(find gpurun_out/pmc_bench/p$i -name '*counter_collection.csv' | head -1)
  python3 scripts/pmc_summary.py "$KEY" $f
done
