"""How full is the GPU during a training step?  From a rocprofv3 --kernel-trace CSV: at every instant, the workgroups of all running
kernels (capped per kernel by its grid) are summed; prints, per step (delimited by adam_kernel), the time spent with fewer than 64 /
128 / 256 workgroups in flight and which kernels that time belongs to.  usage: python tools/gpu_fill.py <kernel_trace.csv>"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))


def wgs(r):
    g = int(r['Grid_Size_X']) * int(r.get('Grid_Size_Y', 1) or 1) * int(r.get('Grid_Size_Z', 1) or 1)
    w = int(r['Workgroup_Size_X']) * int(r.get('Workgroup_Size_Y', 1) or 1) * int(r.get('Workgroup_Size_Z', 1) or 1)
    return max(1, g // max(1, w))


ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'], wgs(r)) for r in rows)
adam = [i for i, e in enumerate(ev) if 'adam_kernel' in e[2]]
a, b = adam[-2], adam[-1]
seg = ev[a + 1:b + 1]
pts = []
for s, e, name, w in seg:
    pts.append((s, 1, name, w)); pts.append((e, -1, name, w))
pts.sort()
running = {}
low = defaultdict(float)
tot = {64: 0.0, 128: 0.0, 256: 0.0}
prev = pts[0][0]
for t, kind, name, w in pts:
    dt = t - prev
    if dt > 0:
        cur = sum(running.values())
        for th in tot:
            if cur < th:
                tot[th] += dt
        if cur < 128:
            key = ' + '.join(sorted(set(k[0].split('(')[0].replace('void (anonymous namespace)::', '')[:40] for k in running))) or '(idle)'
            low[key] += dt
    prev = t
    if kind == 1:
        running[(name, t, w)] = w
    else:
        for k in list(running):
            if k[0] == name and k[2] == w:
                del running[k]
                break
span = seg[-1][1] - seg[0][0]
print(f'step span {span/1e6:.2f} ms; time with < 64 / 128 / 256 workgroups in flight: ' + ' / '.join(f'{tot[k]/1e6:.2f}' for k in (64, 128, 256)) + ' ms')
for k, v in sorted(low.items(), key=lambda kv: -kv[1])[:25]:
    print(f'  {v/1e3:8.1f} us  {k}')
