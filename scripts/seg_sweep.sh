#!/bin/bash
# dev tool: bench.py at several rows-per-wave segment lengths of the chain kernel (IMGPROC_SEG_ROWS)
for sr in "$@"; do
  IMGPROC_SEG_ROWS=$sr timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null > /tmp/seg_$sr.json
  python -c "import json; d=json.load(open('/tmp/seg_$sr.json')); print($sr, round(d['value']), d['roofline']['kernel_ms_per_launch'])"
done
