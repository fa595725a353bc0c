cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
out=gpurun_out/lds_pmc; rm -rf $out; mkdir -p $out
timeout -s KILL 150 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES --output-format csv -d $out -o pmc -- python3 bench.py --config c5 --no-cpu-baseline --no-streamed > $out/log.txt 2>&1 || { tail -5 $out/log.txt; exit 1; }
python3 - $out <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "ip" in k: print(k, {c: round(sum(v) / len(v) / 1e6, 3) for c, v in cs.items()})
PY
