#!/bin/bash
# One rocprofv3 --pmc pass over a bench configuration, per-kernel means printed (in millions).
# usage (on the GPU box): bash scripts/pmc_pass.sh <config> <counter> [<counter> ...]     (at most 8 SQ_ counters)
cfg=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
out=gpurun_out/pmc_pass_$cfg; rm -rf $out; mkdir -p $out
timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out -o pmc -- python3 bench.py --config $cfg --no-cpu-baseline --no-streamed --no-check > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].replace("void ", "")[:56]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "ip" in k:
        print(k)
        for c, v in sorted(cs.items()):
            print("    %-34s %12.3f M" % (c, sum(v) / len(v) / 1e6))
PY
