#!/bin/bash
# dev tool: latency / streaming throughput against the staging copy's thread count and chunk size
for cfg in "0 768" "3 768" "3 512" "3 1024" "2 768" "5 768" "3 384"; do
  set -- $cfg
  export IMGPROC_COPY_THREADS=$1 IMGPROC_COPY_CHUNK_KB=$2
  echo "threads=$1 chunk_kb=$2"
  timeout -k 10 100 python scripts/latency_bench.py 2>&1 | grep baseline || exit 1
  timeout -k 10 200 python scripts/stream_bench.py 2>&1 | tail -1 | cut -c1-60 || exit 1
done
