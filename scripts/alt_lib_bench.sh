#!/bin/bash
# dev tool: bench.py with alternative builds of the library (image_processing_amd/lib/alt/*.so), A/B in one GPU call
L=image_processing_amd/lib
cp $L/libimgproc_hip.so /tmp/orig.so
for so in "$@"; do
  if [ "$so" = orig ]; then cp /tmp/orig.so $L/libimgproc_hip.so; else cp $L/alt/$so $L/libimgproc_hip.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline 2>/dev/null > /tmp/alt.json
  python -c "import json; d=json.load(open('/tmp/alt.json')); print('$so', round(d['value']), d['roofline']['kernel_ms_per_launch'])"
done
cp /tmp/orig.so $L/libimgproc_hip.so
