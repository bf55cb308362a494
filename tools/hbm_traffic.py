"""HBM traffic per kernel from a rocprofv3 PMC pass (own pass, --kernel-trace only):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum \
        -d /tmp/pmc -o p --output-format csv -- python bench.py --no-cpu-baseline --steps 2 --warmup 1 --no-kernel-timing
    python tools/hbm_traffic.py /tmp/pmc/p_counter_collection.csv profiles/r01_hbm_traffic.json

Bytes as /opt/skills/guides/MI355X_MICROARCH.md (HBM section) prescribes for gfx950: the L2's fabric-side requests, a
read request = 128 B unless counted in RDREQ_32B (rocprofv3's FETCH_SIZE tallies them at 64 B = half the bytes of a wide
streaming read -- "double it"), a write request = 64 B if counted in WRREQ_64B else 32 B.  Infinity-Cache hits are
included (the counters sit between L2 and the fabric).
"""
import collections
import csv
import json
import sys

rows = csv.DictReader(open(sys.argv[1]))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for r in rows:
    k = r['Kernel_Name']
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'TCC_EA0_RDREQ_sum':
        cnt[k] += 1
out = {}
for k, v in agg.items():
    rd, rd32 = v.get('TCC_EA0_RDREQ_sum', 0), v.get('TCC_EA0_RDREQ_32B_sum', 0)
    wr, wr64 = v.get('TCC_EA0_WRREQ_sum', 0), v.get('TCC_EA0_WRREQ_64B_sum', 0)
    out[k] = dict(launches=cnt[k], read_bytes=rd32 * 32 + (rd - rd32) * 128, write_bytes=wr64 * 64 + (wr - wr64) * 32)
# the launches behind srvp_conv_mfma / srvp_conv_mfma_multi: tile kernels + (round 4) the streaming kernels of the 64-channel 64x64 layers
conv = [v for k, v in out.items() if any(n in k for n in ('conv_halo_kernel', 'conv_mfma_kernel', 'conv_stream64_kernel', 'conv_stream_sub64_kernel'))]
summary = dict(
    source='rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum (own pass)',
    srvp_conv_mfma=dict(launches=sum(v['launches'] for v in conv),
                        bytes_per_launch=sum(v['read_bytes'] + v['write_bytes'] for v in conv) / max(1, sum(v['launches'] for v in conv))),
    kernels={k[:120]: v for k, v in sorted(out.items(), key=lambda kv: -(kv[1]['read_bytes'] + kv[1]['write_bytes']))[:25]})
json.dump(summary, open(sys.argv[2], 'w'), indent=1)
print(json.dumps(summary['srvp_conv_mfma']))
