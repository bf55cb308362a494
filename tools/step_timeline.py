"""Timeline of ONE training step from a `rocprofv3 --kernel-trace` CSV (p_kernel_trace.csv): per-queue busy time, where the phases start,
idle gaps of the main queue and which queue finishes last -- the numbers behind DESIGN.md's statements about the two streams.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --output-format csv -d /tmp/prof -o p -- python bench.py --no-cpu-baseline --no-kernel-timing --steps 4 --warmup 2 [--batch 24]
    python tools/step_timeline.py /tmp/prof/p_kernel_trace.csv > profiles/r03_timeline_b192.txt

(Under the profiler every launch costs more host time: at 24 sequences per GPU the host becomes the limiter in places and gaps appear
that an untraced run does not have -- the step is 8.7-8.9 ms untraced, ~9.0 ms traced.)
"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ad = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
assert len(ad) >= 2, 'need at least two optimizer steps in the trace'
a, b = ad[-2], ad[-1]
step = rows[a:b + 1]
t0 = int(step[0]['Start_Timestamp'])
S = lambda r: (int(r['Start_Timestamp']) - t0) / 1e3
D = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print(f'step (adam -> adam): {S(step[-1]):.1f} us, {len(step)} launches')
busy, cnt = defaultdict(float), defaultdict(int)
for r in step:
    busy[r['Queue_Id']] += D(r)
    cnt[r['Queue_Id']] += 1
main = max(busy, key=busy.get)
for q in busy:
    print(f'  queue {q}{" (main)" if q == main else ""}: {cnt[q]} launches, {busy[q]:.1f} us of kernel time')
print('phase starts (us after adam):')
for name in ('conv_in_fwd', 'lstm_fused_fwd', 'rollout_fused_fwd', 'latent_to_z', 'nll_kernel', 'out_dpre', 'dz_split', 'rollout_fused_bwd',
             'rows_scatter', 'conv_in_wgrad'):
    f = next((r for r in step if name in r['Kernel_Name']), None)
    if f is not None:
        print(f'  {name:20s} {S(f):9.1f}')
prev, gaps = None, []
for r in step:
    if r['Queue_Id'] != main:
        continue
    if prev is not None and S(r) - prev > 20:
        gaps.append((S(r) - prev, S(r), r['Kernel_Name'][:60]))
    prev = S(r) + D(r)
print(f'idle gaps > 20 us on the main queue: {sum(g[0] for g in gaps):.1f} us')
for g in gaps:
    print(f'  {g[0]:8.1f} us before {g[1]:9.1f}  {g[2]}')
side = [r for r in step if r['Queue_Id'] != main]
if side:
    cur = None
    print('busy intervals of the second queue (gaps < 200 us merged):')
    for r in side:
        s, e = S(r), S(r) + D(r)
        if cur is None:
            cur = [s, e]
        elif s - cur[1] > 200:
            print(f'  {cur[0]:9.1f} .. {cur[1]:9.1f}')
            cur = [s, e]
        else:
            cur[1] = max(cur[1], e)
    print(f'  {cur[0]:9.1f} .. {cur[1]:9.1f}')
    last_main = max(S(r) + D(r) for r in step[:-1] if r['Queue_Id'] == main)
    print(f'last kernel before adam: main queue ends at {last_main:.1f}, second queue at {max(S(r) + D(r) for r in side):.1f}')
small = [D(r) for r in step if r['Queue_Id'] == main and D(r) < 10]
print(f'kernels < 10 us on the main queue: {len(small)} ({sum(small):.1f} us)')

if len(sys.argv) > 2 and sys.argv[2] == '--list':
    import re
    print('every launch of the step: queue, start us, duration us, kernel')
    for r in step:
        nm = re.sub(r'\(anonymous namespace\)::|void |at::native::', '', r['Kernel_Name'])[:90]
        print(f"  q{'M' if r['Queue_Id'] == main else 'S'} {S(r):9.1f} {D(r):8.1f}  {nm}")
