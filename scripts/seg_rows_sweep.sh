#!/bin/bash
# chain kernel segment height (IMGPROC_SEG_ROWS; 0 = the launcher's own choice) against frames/s, two batches in flight on two streams
# usage: bash scripts/seg_rows_sweep.sh "c2 c3" "0 150 200 300"
set -e
mkdir -p gpurun_out/sr
for cfg in $1; do for r in $2; do
  IMGPROC_SEG_ROWS=$r timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-streamed > gpurun_out/sr/${cfg}_r$r.json
  python - <<PY
import json
d=json.load(open("gpurun_out/sr/${cfg}_r$r.json")); r=d["roofline"]
print("$cfg seg_rows $r", round(d["value"],1), "frac %.4f" % r["frac"], "k_eff %.4f" % r["kernel_ms_per_batch_effective"], d["config"]["checked"], flush=True)
PY
done; done
