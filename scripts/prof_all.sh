#!/bin/bash
# every bench configuration through scripts/prof_config.sh (kernel-trace stats + the PMC passes), one after the other
# usage (GPU box): bash scripts/prof_all.sh r03 "c1 c2 c3 c4 c5 pp pe"
tag=$1; cfgs=${2:-"c1 c2 c3 c4 c5 pp pe"}
for c in $cfgs; do
  case $c in c4) extra="--steps 2 --warmup 1";; c5) extra="--steps 12 --warmup 3";; c1) extra="--steps 60 --warmup 10";; *) extra="--steps 24 --warmup 4";; esac
  echo "== $c"; bash scripts/prof_config.sh $c ${tag}_$c $extra > gpurun_out/${tag}_$c.log 2>&1 || { tail -5 gpurun_out/${tag}_$c.log; exit 1; }
  tail -c 300 gpurun_out/${tag}_$c.log; echo
done
