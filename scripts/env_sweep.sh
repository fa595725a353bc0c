#!/bin/bash
# Kernel-trace of one bench configuration for several values of an environment knob of the library.
# usage (on the GPU box): bash scripts/env_sweep.sh <config> <ENV_NAME> <kernel-name-substring> <value> [<value> ...]
cfg=$1; var=$2; pat=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
for v in "$@"; do
  out=gpurun_out/sweep_$cfg/$var/$v
  rm -rf $out && mkdir -p $out
  export $var=$v
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $out -o s -- python3 bench.py --config $cfg --no-cpu-baseline --no-streamed > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
  python3 - $out "$pat" "$var=$v" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if sys.argv[2] in r["Name"]:
        print("%-28s %-50s calls %6s avg %10.1f us" % (sys.argv[3], r["Name"][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
