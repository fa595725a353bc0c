#!/bin/bash
# hipGraph replay of the per-frame work against direct launches (north_star names hipGraph; DESIGN §4 says it measured
# slower): scripts/stream_bench.py with IMGPROC_GRAPH=0 / 1, baseline chain and polygon_path, 1920x1200, 3 slots.
# Writes gpurun_out/graph_vs_direct.json (copied to profiles/r03_graph_vs_direct.json).
cd $GRAFT_REPO_ROOT || exit 1
out=gpurun_out/graph_vs_direct; mkdir -p $out
for preset in baseline polygon; do
  for g in 0 1; do
    IMGPROC_GRAPH=$g STREAM_PINNED=0 timeout -k 10 120 python3 scripts/stream_bench.py 256 3 1920 1200 $preset > $out/${preset}_g$g.log 2>&1 || { tail -3 $out/${preset}_g$g.log; exit 1; }
  done
done
python3 - $out <<'PY'
import json, re, sys, os
d = {"_what": "imgproc_process_batch (host frames -> pinned staging -> H2D -> chain -> moments) over 256 frames of 1920x1200, 3 slots; "
              "IMGPROC_GRAPH=0: direct launches per frame; IMGPROC_GRAPH=1: the per-frame work {H2D, memset, chain kernels, D2H} captured "
              "once per slot into a hipGraph and replayed with hipGraphLaunch.  scripts/graph_vs_direct.sh"}
for f in sorted(os.listdir(sys.argv[1])):
    if f.endswith(".log"):
        txt = open(os.path.join(sys.argv[1], f)).read()
        m = re.search(r"streamed \w+ \d+x\d+: (\d+) frames/s, ([\d.]+) GB/s", txt)
        d[f[:-4]] = {"frames_per_s": int(m.group(1)), "pcie_gbps": float(m.group(2))} if m else {"log": txt[-300:]}
json.dump(d, open(os.path.join(sys.argv[1], "..", "graph_vs_direct.json"), "w"), indent=1)
print(json.dumps(d, indent=1))
PY
