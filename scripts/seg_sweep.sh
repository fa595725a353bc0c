#!/bin/bash
# Kernel-trace of one bench configuration for several segment heights of the strip kernel.
# usage (on the GPU box): bash scripts/seg_sweep.sh <config> <seg> [<seg> ...]   (0 = the launch's own choice)
cfg=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
for seg in "$@"; do
  out=gpurun_out/seg_$cfg/$seg
  rm -rf $out && mkdir -p $out
  if [ "$seg" != 0 ]; then export IMGPROC_BITSTRIP_SEG_ROWS=$seg; else unset IMGPROC_BITSTRIP_SEG_ROWS; fi
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 bench.py --config $cfg --no-cpu-baseline --no-streamed > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
  echo "seg=$seg"; python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    print("  %-60s calls %6s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
