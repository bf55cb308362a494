"""Per-step launch count and busy time of every kernel in a rocprofv3 --kernel-trace CSV (steps = adam_kernel launches)."""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n_steps = sum(1 for r in rows if 'adam_kernel' in r['Kernel_Name'])
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    k = r['Kernel_Name']
    k = re.sub(r'\(anonymous namespace\)::', '', k)
    k = (k.split('(')[0] if 'at::native' not in k else 'torch: ' + k[:k.find('(', 60) if k.find('(', 60) > 0 else 150])[:150]
    agg[k][0] += 1
    agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
tot = sum(v[1] for v in agg.values())
print('steps', n_steps, 'kernel time per step %.2f ms' % (tot / n_steps / 1e6))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print('%7.1f launches/step %8.3f ms/step  %s' % (c / n_steps, t / n_steps / 1e6, k))
