"""Main-stream coverage of the last training step in a rocprofv3 kernel trace: time with no main-stream kernel resident
(weight-gradient kernels of the side stream excluded) and the kernel transitions those gaps sit at.
Usage: python tools/trace_gaps.py <rocprofv3 output dir>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_pack_input' in r['Kernel_Name']]
j = s0 = idx[-1]
while s0 > 0 and 'k_sgd' not in rows[s0 - 1]['Kernel_Name'] and j - s0 < 16:
    s0 -= 1                          # the step starts with the lr fill + weight re-layout launches, before k_pack_input
seg = rows[s0 if s0 > 0 and 'k_sgd' in rows[s0 - 1]['Kernel_Name'] else j:]
side = lambda n: any(k in n for k in ('k_wgrad', 'k_reduce_slabs', 'k_unpack_stem'))
short = lambda n: n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:30]
main = sorted([r for r in seg if not side(r['Kernel_Name'])], key=lambda r: int(r['Start_Timestamp']))
t0 = int(seg[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in seg)
tl = [int(r['Start_Timestamp']) for r in seg if 'k_main' in r['Kernel_Name']][0]
pairs = collections.Counter(); tot = collections.Counter(); gsum = 0
ce = int(main[0]['End_Timestamp']); prev = main[0]
for r in main[1:]:
    s = int(r['Start_Timestamp'])
    if s > ce:
        gsum += s - ce
        if s > ce + 3000:
            k = (short(prev['Kernel_Name']), short(r['Kernel_Name'])); pairs[k] += 1; tot[k] += s - ce
    if int(r['End_Timestamp']) > ce:
        ce = int(r['End_Timestamp']); prev = r
print("span %.2f ms (forward %.2f, backward+step %.2f); main-stream gaps %.2f ms" % ((t1 - t0) / 1e6, (tl - t0) / 1e6, (t1 - tl) / 1e6, gsum / 1e6))
for k, v in tot.most_common(8):
    print("  %-30s -> %-30s n=%3d total %.2f ms avg %.1f us" % (k[0], k[1], pairs[k], v / 1e6, v / pairs[k] / 1e3))
