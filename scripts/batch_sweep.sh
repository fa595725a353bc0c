#!/bin/bash
# frames per batch against frames/s (bench.py --batch), per configuration: "c3:64,128,256 c4:64,128"
set -e
mkdir -p gpurun_out/bs
for spec in "$@"; do
  cfg=${spec%%:*}
  for b in $(echo ${spec#*:} | tr , ' '); do
    timeout -k 10 300 python bench.py --config $cfg --batch $b --no-cpu-baseline --no-streamed > gpurun_out/bs/${cfg}_b$b.json
    python - <<PY
import json
d=json.load(open("gpurun_out/bs/${cfg}_b$b.json")); r=d["roofline"]
print("$cfg batch $b", round(d["value"],1), "frac %.4f" % r["frac"], "in flight %.2f" % r["batches_in_flight"], "k_eff %.4f" % r["kernel_ms_per_batch_effective"], d["config"]["checked"], flush=True)
PY
  done
done
