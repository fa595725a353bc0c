#!/bin/bash
# A/B: ticket 1 of the device-batch path on a stream of its own (IMGPROC_BATCH_STREAMS=2) against both tickets on one stream
set -e
mkdir -p gpurun_out/ts
for cfg in ${CFGS:-c2 c5 pp pe c4 c3 c1}; do
  for s in 1 2; do
    IMGPROC_BATCH_STREAMS=$s timeout -k 10 200 python bench.py --config $cfg --no-cpu-baseline --no-streamed > gpurun_out/ts/${cfg}_s$s.json
    python - <<PY
import json
d=json.load(open("gpurun_out/ts/${cfg}_s$s.json"))
print("$cfg streams=$s", round(d["value"],1), "ms/step", round(d["ms_per_step"],4), "kernel_ms", round(d["roofline"]["kernel_ms_per_batch"],4), "checked", d["config"]["checked"], flush=True)
PY
  done
done
