#!/bin/bash
# c5 against the segments per strip of the fused HSV + Gaussian + inRange kernel (IMGPROC_GAUSS_SEGS)
cd $GRAFT_REPO_ROOT || exit 1
for s in 0 8 12 16 23 34 46 68; do
  line=$(IMGPROC_GAUSS_SEGS=$s timeout -k 10 120 python3 bench.py --config c5 --no-cpu-baseline --no-check --steps 20 --warmup 4 2>/dev/null | tail -1) || exit 1
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('segs', sys.argv[2], 'kernel_ms %.4f' % d['roofline']['kernel_ms_per_batch'], 'fps %.0f' % d['value'], 'frac %.4f' % d['roofline']['frac'])" "$line" $s
done
