"""Per-step view of a rocprofv3 kernel trace of the pipelined loop: busy time, gaps and per-kernel means on the queue that
runs the feature graphs (the one with adam_kernel).  usage: trace_steps.py <kernel_trace.csv> [n_steps]"""
import collections
import csv
import statistics
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
nst = int(sys.argv[2]) if len(sys.argv) > 2 else 15
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
adam = sorted([r for r in rows if r['Kernel_Name'].startswith('adam_kernel')], key=lambda r: r['s'])
q = adam[-1]['Queue_Id']
ends = [r['e'] for r in adam]
per = []
for a, b in zip(ends[-nst - 1:-1], ends[-nst:]):
    ks = sorted([r for r in rows if r['Queue_Id'] == q and r['s'] >= a and r['e'] <= b], key=lambda r: r['s'])
    busy = sum(r['e'] - r['s'] for r in ks)
    gaps = sum(max(0, ks[i + 1]['s'] - ks[i]['e']) for i in range(len(ks) - 1))
    per.append((b - a, busy, gaps, len(ks)))
print('mean step %.1f us  busy %.1f  gaps %.1f  kernels %.0f' % tuple(statistics.mean(x[i] for x in per) / (1000 if i < 3 else 1) for i in range(4)))
a, b = ends[-nst - 1], ends[-1]
ks = sorted([r for r in rows if r['Queue_Id'] == q and r['s'] >= a and r['e'] <= b], key=lambda r: r['s'])
d = collections.defaultdict(list)
for r in ks:
    d[r['Kernel_Name'][:90]].append(r['e'] - r['s'])
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1]))[:45]:
    print('%-92s n/step %4.1f  mean %6.1f us  per-step %6.1f us' % (k, len(v) / nst, statistics.mean(v) / 1000, sum(v) / nst / 1000))
gp = collections.defaultdict(list)
for i in range(len(ks) - 1):
    gp[(ks[i]['Kernel_Name'][:45], ks[i + 1]['Kernel_Name'][:45])].append(ks[i + 1]['s'] - ks[i]['e'])
for k, v in sorted(gp.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print('gap %.1f us/step' % (sum(v) / nst / 1000), k)
others = collections.defaultdict(list)
for r in rows:
    if r['Queue_Id'] != q and r['s'] >= a and r['e'] <= b:
        others[(r['Queue_Id'], r['Kernel_Name'][:50])].append(r['e'] - r['s'])
for k, v in sorted(others.items(), key=lambda kv: -sum(kv[1]))[:10]:
    print('side', k, 'n/step %.1f mean %.1f us' % (len(v) / nst, statistics.mean(v) / 1000))
if len(sys.argv) > 3 and sys.argv[3] == 'seq':          # the last step's launches in order: offset, duration, gap to the previous one
    a, b = ends[-2], ends[-1]
    ks = sorted([r for r in rows if r['Queue_Id'] == q and r['s'] >= a and r['e'] <= b], key=lambda r: r['s'])
    prev = a
    for i, r in enumerate(ks):
        print('%3d  +%7.1f us  %6.1f us  gap %5.1f  %s' % (i, (r['s'] - a) / 1000, (r['e'] - r['s']) / 1000, (r['s'] - prev) / 1000, r['Kernel_Name'][:100]))
        prev = r['e']
