#!/bin/bash
# one environment variable against frames/s: bash scripts/env_sweep.sh <config> <VAR> "<values>" [bench args]
set -e
cfg=$1; var=$2; vals=$3; shift 3
mkdir -p gpurun_out/es
for v in $vals; do
  env $var=$v timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-streamed "$@" > gpurun_out/es/${cfg}_${var}_$v.json
  python - <<PY
import json
d=json.load(open("gpurun_out/es/${cfg}_${var}_$v.json")); r=d["roofline"]
print("$cfg $var=$v", round(d["value"],1), "frac %.4f" % r["frac"], "k_eff %.4f" % r["kernel_ms_per_batch_effective"], d["config"]["checked"], flush=True)
PY
done
