#!/bin/bash
# dev tool: bench.py --config <cfgs> with alternative builds of the library (image_processing_amd/lib/alt/*.so), A/B in one GPU call
# usage: bash scripts/alt_lib_cfg.sh "c5 pe pp" orig pf4.so pf8.so
L=image_processing_amd/lib
cfgs=$1; shift
cp $L/libimgproc_hip.so /tmp/orig.so
for so in "$@"; do
  if [ "$so" = orig ]; then cp /tmp/orig.so $L/libimgproc_hip.so; else cp $L/alt/$so $L/libimgproc_hip.so; fi
  for c in $cfgs; do
    timeout -k 10 200 python bench.py --config $c --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/alt.json || { cp /tmp/orig.so $L/libimgproc_hip.so; exit 1; }
    python -c "import json; d=json.load(open('/tmp/alt.json')); print('$so', '$c', round(d['value']), 'kernel_ms %.4f' % d['roofline']['kernel_ms_per_batch'], 'checked', d['config']['checked'])"
  done
done
cp /tmp/orig.so $L/libimgproc_hip.so
