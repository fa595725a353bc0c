#!/bin/bash
# dev tool: preset_bench / stress_bench with alternative builds of the library (image_processing_amd/lib/alt/*.so)
L=image_processing_amd/lib
cp $L/libimgproc_hip.so /tmp/orig.so
for so in "$@"; do
  if [ "$so" = orig ]; then cp /tmp/orig.so $L/libimgproc_hip.so; else cp $L/alt/$so $L/libimgproc_hip.so; fi
  echo "== $so"
  timeout -k 10 100 python scripts/preset_bench.py polygon_path pose_estimate_full 2>/dev/null | cut -c1-75
  timeout -k 10 100 python scripts/stress_bench.py 2>/dev/null | cut -c1-75
done
cp /tmp/orig.so $L/libimgproc_hip.so
