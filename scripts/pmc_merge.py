"""Rebuilds profiles/r03_pmc_summary.json (what bench.py quotes as offline traffic / VALU figures) from the per-configuration
counter files profiles/r03_<config>_counters.json that scripts/prof_config.sh + scripts/pmc_collect.py produce.
usage: python3 scripts/pmc_merge.py            (reads and writes under profiles/)"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")
WHAT = ("Per configuration of bench.py: HBM bytes and VALU instructions of one batch's chain kernel(s), from the rocprofv3 passes "
        "in profiles/r03_<config>_counters.json (scripts/prof_config.sh: kernel-trace stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE and "
        "two SQ groups, each a run of its own; merged by scripts/pmc_merge.py). avg_us is the kernel-trace average of a launch as "
        "bench.py runs it (launches of consecutive batches overlap on two streams: launches_in_flight), alone_us the same launch on its "
        "own (the GRBM counter pass runs one dispatch at a time; clock_ghz = GRBM_GUI_ACTIVE / 8 / that time). hbm_bytes = 2 x FETCH_SIZE + WRITE_SIZE for kernels "
        "whose loads the guide's gfx950 rule covers or that were calibrated here (k_fused_chain, k_bgr2hsv, k_gaussian_*, "
        "k_in_range_bits: 2 x FETCH_SIZE reproduces their known input bytes x halo); k_bitmorph's / k_bitstrip's dword loads of bit "
        "planes are counted as reported (FETCH_SIZE x 1: that already equals bit-plane bytes x tile overlap). bench.py quotes these "
        "as offline figures (traffic_source).")
FETCH_X1 = ("k_bitmorph", "k_bitstrip")  # FETCH_SIZE already in units of 1 KB of HBM reads for these


def main():
    out = {"_what": WHAT}
    classes = {}
    cpath = os.path.join(PROF, "r03_isa_classes.json")  # scripts/isa_count.py --json: cheap-class share of each chain kernel's main loop
    if os.path.exists(cpath):
        classes = json.load(open(cpath))
    for cfg in ("c1", "c2", "c3", "c4", "pp", "c5", "pe"):
        path = os.path.join(PROF, "r03_%s_counters.json" % cfg)
        if not os.path.exists(path):
            continue
        d = json.load(open(path))
        line = d["bench_line_under_rocprof"]
        frames = line["config"]["frames_per_batch"]
        steps = line["steps"] + line["warmup"]
        kernels, hbm, valu = {}, 0.0, 0.0
        total_calls = sum(k["calls"] for k in d["kernels"].values())
        launches = line["roofline"].get("launches_per_batch") or len(d["kernels"])
        for name, k in d["kernels"].items():
            if "FETCH_SIZE" not in k or "SQ_INSTS_VALU" not in k:
                continue
            per_batch = max(1, round(launches * k["calls"] / total_calls))  # this kernel's share of a batch's launches
            fx = 1.0 if any(s in name for s in FETCH_X1) else 2.0
            rd, wr = fx * k["FETCH_SIZE"] * 1024.0, k["WRITE_SIZE"] * 1024.0
            # effective clock of the dispatch: GRBM_GUI_ACTIVE is summed over the 8 XCDs (guide, DVFS give-back)
            # serial_ns: the launch on its own (the counter pass runs one dispatch at a time; avg_ns of the kernel trace is
            # longer because launches of consecutive batches overlap there)
            alone = k.get("serial_ns") or k["avg_ns"]
            ghz = k["GRBM_GUI_ACTIVE"] / 8.0 / alone if k.get("GRBM_GUI_ACTIVE") else None
            cyc = alone * (ghz or 2.4) * 1024.0 / k["SQ_INSTS_VALU"]  # SIMD cycles per VALU instruction at that clock
            kernels[name] = {"launches_per_batch": per_batch, "avg_us": k["avg_ns"] / 1e3, "alone_us": alone / 1e3, "read_bytes": rd, "write_bytes": wr,
                             "valu_insts": k["SQ_INSTS_VALU"], "salu_insts": k.get("SQ_INSTS_SALU"), "lds_insts": k.get("SQ_INSTS_LDS"),
                             "waves": k.get("SQ_WAVES"), "clock_ghz": ghz, "valu_cycles_per_inst": cyc,
                             "wait_inst_any_share": (k["SQ_WAIT_INST_ANY"] / k["SQ_WAVE_CYCLES"]) if k.get("SQ_WAVE_CYCLES") else None}
            hbm += per_batch * (rd + wr)
            valu += per_batch * k["SQ_INSTS_VALU"]
        ghzs = [(v["clock_ghz"], v["alone_us"] * v["launches_per_batch"]) for v in kernels.values() if v.get("clock_ghz")]
        out[cfg] = {"frames_per_launch": frames, "hbm_bytes_per_launch": hbm, "valu_insts_per_launch": valu,
                    "clock_ghz": (sum(g * t for g, t in ghzs) / sum(t for _, t in ghzs)) if ghzs else None,  # time-weighted
                    "cheap_class_share": classes.get(cfg, {}).get("cheap_class_share"),
                    "launches_in_flight": ((d.get("kernel_trace") or {}).get("timed_steps") or {}).get("launches_in_flight"),
                    "trace_union_ms_per_batch": ((d.get("kernel_trace") or {}).get("timed_steps") or {}).get("union_ms_per_batch"),
                    "lib_sha16": d.get("lib_sha16"), "src_sha16": d.get("src_sha16"), "kernels": kernels}
    json.dump(out, open(os.path.join(PROF, "r03_pmc_summary.json"), "w"), indent=1)
    for cfg, v in out.items():
        if cfg != "_what":
            print(cfg, "%.3f GB, %.1f M VALU per batch of %d" % (v["hbm_bytes_per_launch"] / 1e9, v["valu_insts_per_launch"] / 1e6, v["frames_per_launch"]))


if __name__ == "__main__":
    main()
