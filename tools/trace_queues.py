"""Which hardware queue each kernel of the last training step ran on (rocprofv3 --kernel-trace CSV): per queue the number
of kernels, busy time and first/last timestamps relative to the step start -- shows whether the weight-gradient branch of the
captured graph really ran beside the main chain or behind it.  Usage: python tools/trace_queues.py <rocprofv3 output dir>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_pack_input' in r['Kernel_Name']]
seg = rows[idx[-1]:]
t0 = int(seg[0]['Start_Timestamp'])
qkey = 'Queue_Id' if 'Queue_Id' in seg[0] else [k for k in seg[0] if 'ueue' in k][0]
per = collections.OrderedDict()
for r in seg:
    q = r[qkey]
    d = per.setdefault(q, dict(n=0, busy=0, first=None, last=0, names=collections.Counter()))
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    d['n'] += 1; d['busy'] += e - s
    d['first'] = s if d['first'] is None else d['first']; d['last'] = max(d['last'], e)
    d['names'][r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('<')[0].split('(')[0]] += 1
print("columns:", [k for k in seg[0].keys()])
for q, d in per.items():
    print("queue %s: %d kernels, busy %.2f ms, first %.2f ms, last %.2f ms  %s" % (
        q, d['n'], d['busy'] / 1e6, d['first'] / 1e6, d['last'] / 1e6, dict(d['names'].most_common(6))))
side = lambda n: any(k in n for k in ('k_wgrad', 'k_reduce_slabs'))
ws = [r for r in seg if side(r['Kernel_Name'])]
print("weight-gradient kernels: first start %.2f ms, last end %.2f ms" % ((int(ws[0]['Start_Timestamp']) - t0) / 1e6,
                                                                          (int(ws[-1]['End_Timestamp']) - t0) / 1e6))
# time-binned occupancy of the side kernels: share of each ms of the step in which a weight-gradient kernel was running
end = max(int(r['End_Timestamp']) for r in seg) - t0
bins = [0.0] * (end // 1000000 + 1)
for r in ws:
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    b = s // 1000000
    while s < e:
        nxt = min(e, (b + 1) * 1000000)
        bins[b] += nxt - s
        s = nxt; b += 1
print("wgrad-resident share per ms of the step:", " ".join("%d" % round(100 * x / 1e6) for x in bins))
# dispatch order around the first weight-gradient kernels: id, queue, start, end (ms from step start)
first = seg.index(ws[0])
byid = sorted(seg, key=lambda r: int(r['Dispatch_Id']))
pos = byid.index(ws[0])
print("dispatch-id neighbourhood of the first weight-gradient kernel:")
for r in byid[max(0, pos - 4):pos + 8]:
    print("  id %s q%s  %8.3f -> %8.3f  %s" % (r['Dispatch_Id'], r[qkey], (int(r['Start_Timestamp']) - t0) / 1e6,
                                             (int(r['End_Timestamp']) - t0) / 1e6,
                                             r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:50]))
print("first 6 weight-gradient kernels by start time:")
for r in ws[:6]:
    print("  id %s q%s  %8.3f -> %8.3f  %s" % (r['Dispatch_Id'], r[qkey], (int(r['Start_Timestamp']) - t0) / 1e6,
                                             (int(r['End_Timestamp']) - t0) / 1e6,
                                             r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:50]))
hb = [r for r in seg if 'k_head_bwd' in r['Kernel_Name']]
if hb:
    print("k_head_bwd: id %s  %8.3f -> %8.3f" % (hb[0]['Dispatch_Id'], (int(hb[0]['Start_Timestamp']) - t0) / 1e6, (int(hb[0]['End_Timestamp']) - t0) / 1e6))
tw = int(ws[0]['Start_Timestamp'])
mains = [r for r in seg if not side(r['Kernel_Name'])]
print("main-chain kernels around the start of the first weight-gradient kernel (position in the main chain):")
for k, r in enumerate(mains):
    if abs(int(r['End_Timestamp']) - tw) < 300000 or abs(int(r['Start_Timestamp']) - tw) < 300000:
        print("  #%d id %s q%s  %8.3f -> %8.3f  %s" % (k, r['Dispatch_Id'], r[qkey], (int(r['Start_Timestamp']) - t0) / 1e6,
                                                    (int(r['End_Timestamp']) - t0) / 1e6,
                                                    r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:50]))
