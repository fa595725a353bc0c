#!/bin/bash
# One configuration of bench.py under rocprofv3, each pass a run of its own:
#   kernel-trace stats, then --pmc FETCH_SIZE, --pmc WRITE_SIZE, --pmc GRBM_GUI_ACTIVE (clock) and two SQ counter groups.
# usage (on the GPU box): bash scripts/prof_config.sh <c2|c3|c4|c5> <tag> [bench args...]
# Output: gpurun_out/<tag>/{stats,fetch,write,grbm,sq1,sq2}/..., gpurun_out/<tag>/summary.json (scripts/pmc_collect.py)
cfg=$1; tag=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT || exit 1
out=gpurun_out/$tag
rm -rf $out && mkdir -p $out
BENCH="python3 bench.py --config $cfg --no-cpu-baseline --no-streamed --no-check --no-alone $*"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o s -- $BENCH > $out/stats.log 2>&1 || { tail -5 $out/stats.log; exit 1; }
for pass in "fetch FETCH_SIZE" "write WRITE_SIZE" "grbm GRBM_GUI_ACTIVE" \
            "sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" \
            "sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH"; do
  set -- $pass; name=$1; shift
  timeout -s KILL 150 rocprofv3 --pmc "$@" --output-format csv -d $out/$name -o pmc -- $BENCH > $out/$name.log 2>&1 || { tail -5 $out/$name.log; exit 1; }
done
python3 scripts/pmc_collect.py $out $cfg > $out/summary.json && cat $out/summary.json
