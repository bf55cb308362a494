#!/bin/bash
# Round-end evidence: default bench line, rocprofv3 kernel stats of the same command, HBM PMC pass -> gpurun_out/ (copy the
# summaries into profiles/).  usage (on the GPU box, from the repo root): bash tools/profile_round.sh r02 [config]
R=${1:-r02}
CFG=${2:-bair}
SUF=""; [ "$CFG" != "bair" ] && SUF="_$CFG"
ROOT=$(pwd)
OUT=$ROOT/gpurun_out
mkdir -p $OUT
python $ROOT/bench.py --config $CFG > $OUT/${R}_bench$SUF.json 2> $OUT/${R}_bench$SUF.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o p -- python $ROOT/bench.py --config $CFG --no-cpu-baseline --no-unshared > $OUT/prof_run.log 2>&1
cp /tmp/prof/p_kernel_stats.csv $OUT/${R}_kernel_stats$SUF.csv 2>/dev/null || find /tmp/prof -name "*kernel_stats.csv" -exec cp {} $OUT/${R}_kernel_stats$SUF.csv \;
if [ "$CFG" == "bair" ]; then
rm -rf /tmp/pmc && rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum \
    -d /tmp/pmc -o p --output-format csv -- python $ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-kernel-timing > $OUT/pmc_hbm.log 2>&1
f=$(find /tmp/pmc -name "*counter_collection.csv" | head -1)
python $ROOT/tools/hbm_traffic.py $f $OUT/${R}_hbm_traffic.json
fi
head -c 400 $OUT/${R}_bench$SUF.json; echo
