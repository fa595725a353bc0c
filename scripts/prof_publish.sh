#!/bin/bash
# copies what scripts/prof_all.sh <tag> left under gpurun_out/ into profiles/ (tracked) and rebuilds the summary bench.py quotes
# usage (here, after the GPU calls): bash scripts/prof_publish.sh r03
tag=$1
for c in c1 c2 c3 c4 c5 pp pe; do
  [ -f gpurun_out/${tag}_$c/summary.json ] || continue
  cp gpurun_out/${tag}_$c/summary.json profiles/${tag}_${c}_counters.json
  f=$(find gpurun_out/${tag}_$c/stats -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" profiles/${tag}_${c}_kernel_stats.csv
done
for f in gpurun_out/${tag}_bench_*.json; do [ -s "$f" ] && cp "$f" profiles/; done
python3 scripts/pmc_merge.py
