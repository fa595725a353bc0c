#!/bin/bash
# dev tool: the other benches at several IMGPROC_SEG_ROWS
for sr in "$@"; do
  echo "== seg rows $sr"
  IMGPROC_SEG_ROWS=$sr timeout -k 10 100 python scripts/bayer_bench.py 64 2>/dev/null | cut -c1-60
  IMGPROC_SEG_ROWS=$sr timeout -k 10 100 python scripts/preset_bench.py polygon_path pose_estimate baseline_8r 2>/dev/null | cut -c1-75
done
