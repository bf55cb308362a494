#!/bin/bash
ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o p -- python $ROOT/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-kernel-timing > /tmp/tr.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
python $ROOT/tools/gpu_idle.py $f 2>&1 | tail -14
python $ROOT/tools/kernel_mix.py $f 2>&1 | head -30
