#!/bin/bash
# The whole evidence set of a round from ONE call on one box -> gpurun_out/<round>_* (copy what is to be judged into profiles/).
# usage (GPU box, repo root): bash tools/evidence_round.sh r05
R=${1:-r05}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
stats() {  # name, bench args...: rocprofv3 kernel stats of `bench.py <args>`
  n=$1; shift
  (cd /tmp && rm -rf /tmp/prof_$n && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$n -o p -- python $ROOT/bench.py "$@" > $OUT/prof_$n.log 2>&1;
   f=$(find /tmp/prof_$n -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${R}_kernel_stats$n.csv)
}
trace() {  # name, bench args...: per-queue timeline of one step
  n=$1; shift
  (cd /tmp && rm -rf /tmp/tr_$n && rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$n -o p -- python $ROOT/bench.py --no-cpu-baseline --no-kernel-timing --steps 4 --warmup 2 "$@" > $OUT/tr_$n.log 2>&1;
   python $ROOT/tools/step_timeline.py /tmp/tr_$n/p_kernel_trace.csv > $OUT/${R}_timeline_$n.txt 2>&1)
}
# headline: bench line, kernel stats of the same command, HBM PMC pass
bash tools/profile_round.sh $R
# the other configs: bench line + kernel stats
for c in kth human smmnist; do bash tools/profile_round.sh $R $c; done
# 24 sequences per GPU (config 4 as the reference splits it)
python bench.py --batch 24 --no-cpu-baseline > $OUT/${R}_bench_b24.json 2>/dev/null
python bench.py --batch 24 --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 10 > $OUT/${R}_bench_b24_untimed.json 2>/dev/null
SRVP_FORCE_COLLECTIVES=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --batch 24 --no-cpu-baseline --no-kernel-timing --steps 40 --warmup 10 > $OUT/${R}_bench_b24_forced.json 2> $OUT/forced.err
stats _b24 --batch 24 --no-cpu-baseline --no-unshared
python tools/count_collectives.py bair 24 2>/dev/null | grep "^{" | tail -1 > $OUT/${R}_collectives_b24.json
B=24 python tools/layer_times.py > $OUT/${R}_layer_times_b24.txt 2>/dev/null
python tools/layer_times.py > $OUT/${R}_layer_times_b192.txt 2>/dev/null
B=24 python tools/host_time.py > $OUT/${R}_host_time_b24.txt 2>/dev/null
trace b24 --batch 24
trace b192
B=24 python tools/side_timing.py > $OUT/${R}_side_timing_b24.txt 2>/dev/null
# unshared pass (second stream off for the whole process)
SRVP_OVERLAP_WGRAD=0 SRVP_ENC_WGRAD_SIDE_MAXN=0 SRVP_OVERLAP_SKIP=0 SRVP_OVERLAP_PACK=0 stats _unshared --no-cpu-baseline --no-kernel-timing --steps 10 --warmup 3
# config 5's test protocol
python bench.py --config human --mode rollout --no-cpu-baseline > $OUT/${R}_bench_rollout_human.json 2>/dev/null
stats _rollout_human --config human --mode rollout --no-cpu-baseline
# input through the prefetcher
python bench.py --h2d u8 --no-cpu-baseline --no-kernel-timing > $OUT/${R}_bench_h2d_u8.json 2>/dev/null
# persistent latent kernels in isolation, both forms
python tools/rollout_time.py 24 100 192 > $OUT/${R}_rollout_time.jsonl 2>/dev/null
(SRVP_RF_DEBUG=1 python tools/rollout_time.py 24 2>&1 | grep RF_DEBUG | tail -1) > $OUT/${R}_rollout_phase_times.txt
ls -la $OUT | grep ${R}_ | wc -l
for f in $OUT/${R}_bench*.json; do echo $f; python -c "import json,sys; d=json.loads(open('$f').read().strip().splitlines()[-1]); print(d['ms_per_step'], d.get('roofline',{}).get('frac'))"; done
