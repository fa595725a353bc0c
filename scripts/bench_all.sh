#!/bin/bash
# one plain bench.py run of every configuration (JSON lines into gpurun_out/<tag>_bench_<cfg>.json)
# usage (GPU box): bash scripts/bench_all.sh r03
tag=$1
cd $GRAFT_REPO_ROOT || exit 1
for c in c2 c1 c3 c4 c5 pp pe; do
  timeout -k 10 400 python3 bench.py --config $c 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_$c.json || exit 1
  python3 -c "import json; d=json.load(open('gpurun_out/${tag}_bench_$c.json')); r=d['roofline']; print('$c', round(d['value']), 'frac %.4f' % r['frac'], 'kernel_ms %.4f' % r['kernel_ms_per_batch'], 'traffic', r['traffic'] and round(r['traffic']/1e9,3), 'checked', d['config']['checked'], 'cpu', round(d['cpu_baseline']['value'],1))"
done
BENCH_ALL_RANKS_ON_GPU0=1 timeout -k 10 400 python3 bench.py --gpus 2 --config c4 --streamed --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_c4_n2_streamed_rehearsal.json
timeout -k 10 400 python3 bench.py --config c4 --streamed --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${tag}_bench_c4_streamed.json
