"""Turns the passes of scripts/prof_config.sh into one JSON: per kernel of the library (ipf:: / ipm:: / ipk::) the
rocprofv3 kernel-trace average duration and the mean per dispatch of every counter collected.
usage: python3 scripts/pmc_collect.py gpurun_out/<tag> <config>"""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys

out, cfg = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def ours(name):
    return any(ns in name for ns in ("ipf::", "ipm::", "ipk::"))


def short(name):
    return re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "").split("(")[0]


kern = collections.OrderedDict()
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if ours(r["Name"]):
            kern[short(r["Name"])] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])}
# The two tickets of a device batch run on two streams, so launches overlap in the kernel trace: the sum of the launch
# durations over the union of their spans is the number of launches in flight while the chain kernels run.
spans = []
for f in glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if ours(r["Kernel_Name"]):
            spans.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
spans.sort()
bench = None
for ln in open(os.path.join(out, "stats.log")):
    if ln.startswith("{"):
        bench = json.loads(ln)


def overlap_stats(sp):
    union, end = 0, -1
    for t0, t1, _ in sp:
        if t1 > end:
            union += t1 - max(t0, end)
            end = t1
    total = sum(t1 - t0 for t0, t1, _ in sp)
    per = collections.defaultdict(list)
    for t0, t1, k in sp:
        per[k].append(t1 - t0)
    return {"launches": len(sp), "sum_ns": total, "union_ns": union, "launches_in_flight": total / union,
            "avg_ns": {k: sum(v) / len(v) for k, v in per.items()}}


trace = None
if spans:
    trace = {"all": overlap_stats(spans)}  # pre-roll and warmup included (what kernel_stats.csv averages over)
    if bench:  # the launches of the timed steps: the last steps x batches per step x launches per batch of the run
        cfgb, roof = bench["config"], bench["roofline"]
        n_timed = int(round(bench["steps"] * cfgb["frames_per_step"] / bench["n_gpus"] / cfgb["frames_per_batch"])) * int(roof["launches_per_batch"])
        if 0 < n_timed <= len(spans):
            trace["timed_steps"] = overlap_stats(spans[-n_timed:])
            trace["timed_steps"]["union_ms_per_batch"] = trace["timed_steps"]["union_ns"] / 1e6 / (n_timed / int(roof["launches_per_batch"]))
for pas in ("fetch", "write", "grbm", "sq1", "sq2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(out, pas, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if ours(r["Kernel_Name"]):
                acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if pas == "grbm":  # counter passes run one dispatch at a time: the duration of the launch on its own, same dispatch as the cycles
                    acc[short(r["Kernel_Name"])]["serial_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    for k, cs in acc.items():
        d = kern.setdefault(k, {})
        for c, v in cs.items():
            d[c] = sum(v) / len(v)
            d[c + "_n"] = len(v)
lib = os.path.join(ROOT, "image_processing_amd", "lib", "libimgproc_hip.so")
sha = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None
sys.path.insert(0, ROOT)
from bench import build_hashes  # the same two hashes bench.py compares at run time

assert build_hashes()["lib_sha16"] == sha
print(json.dumps({"config": cfg, "lib_sha16": sha, "src_sha16": build_hashes()["src_sha16"], "bench_line_under_rocprof": bench, "kernel_trace": trace, "kernels": kern}, indent=1))
