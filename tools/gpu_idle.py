"""GPU idle time inside the training step from a rocprofv3 --kernel-trace CSV: union of the kernel intervals vs wall span, per
step (steps are delimited by the srvp adam_kernel launches).  usage: python tools/gpu_idle.py <kernel_trace.csv>"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows))
adam = [i for i, e in enumerate(ev) if 'adam_kernel' in e[2]]
print('kernels', len(ev), 'steps', len(adam))
for a, b in list(zip(adam[:-1], adam[1:]))[-4:]:
    seg = ev[a + 1:b + 1]
    t0, t1 = ev[a][1], ev[b][1]
    busy, cur_s, cur_e = 0, None, None
    gaps = []
    where = []
    last_name = 'adam'
    for s, e, name in sorted(seg):
        if cur_e is None:
            cur_s, cur_e = s, e
            gaps.append(s - t0)
            where.append((s - t0, last_name[:60], name[:60]))
        elif s <= cur_e:
            cur_e = max(cur_e, e)
        else:
            gaps.append(s - cur_e)
            where.append((s - cur_e, last_name[:60], name[:60]))
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        last_name = name
    busy += cur_e - cur_s
    gaps.sort(reverse=True)
    if (a, b) == list(zip(adam[:-1], adam[1:]))[-1]:
        for g, pn, nn in sorted(where, reverse=True)[:8]:
            print(f'   gap {g/1e3:.1f} us after [{pn}] before [{nn}]')
    span = t1 - t0
    print(f'step span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms  idle {(span-busy)/1e6:.2f} ms  launches {len(seg)}  '
          f'largest gaps us {[round(g/1e3,1) for g in gaps[:8]]}  gaps>5us: {sum(1 for g in gaps if g>5000)}  sum gaps<5us {sum(g for g in gaps if g<=5000)/1e6:.2f} ms')
