"""Summarise a rocprofv3 --kernel-trace --stats CSV directory: per-kernel totals for the LAST training step."""
import csv, glob, sys, collections
d = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
f = glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_pack_input' in r['Kernel_Name']]
def step_start(j):
    """A step begins right after the previous step's last optimizer kernel: the learning-rate fill and the bf16
    weight re-layout launches precede k_pack_input."""
    s = j
    while s > 0 and 'k_sgd' not in rows[s - 1]['Kernel_Name'] and j - s < 16:
        s -= 1
    return s if s > 0 and 'k_sgd' in rows[s - 1]['Kernel_Name'] else j
seg = rows[step_start(idx[-steps]):]
def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    return n.split('(')[0][:60]
agg = collections.OrderedDict()
busy = 0
t0 = int(seg[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in seg)
for r in seg:
    k = short(r['Kernel_Name']); d_ = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    a = agg.setdefault(k, [0, 0]); a[0] += d_; a[1] += 1; busy += d_
print("last %d step(s): span %.2f ms, kernel busy %.2f ms, %d launches" % (steps, (t1 - t0) / 1e6, busy / 1e6, len(seg)))
for k, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("%7.3f ms %5.1f%%  calls %4d  avg %8.1f us  %s" % (t / 1e6 / steps, 100.0 * t / busy, c // steps, t / c / 1e3, k))
