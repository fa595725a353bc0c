#!/bin/bash
# c2 chain kernel against the segment height (IMGPROC_SEG_ROWS), one bench run each; prints kernel ms and frames/s
cd $GRAFT_REPO_ROOT || exit 1
for r in 0 100 120 150 171 200 240 300 400 600; do
  line=$(IMGPROC_SEG_ROWS=$r timeout -k 10 120 python3 bench.py --config ${1:-c2} --no-cpu-baseline --no-check --steps 60 --warmup 10 2>/dev/null | tail -1) || exit 1
  python3 -c "import json,sys; d=json.loads(sys.argv[1]); print('seg_rows', sys.argv[2], 'kernel_ms %.4f' % d['roofline']['kernel_ms_per_batch'], 'fps %.0f' % d['value'], 'frac %.4f' % d['roofline']['frac'])" "$line" $r
done
