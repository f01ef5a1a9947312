"""Main-stream timeline of the last training step in a rocprofv3 kernel trace (tools/prof_train.sh): start, gap before the kernel,
duration, workgroups, name - and the busy / gap totals of both streams.  usage: python tools/step_timeline.py <t_kernel_trace.csv> [-v] [--mark=<kernel name part>]
(the step = the launches after the second-to-last marker kernel up to the last one; default marker adam_kernel, inference: --mark=input_pack_kernel)"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
by = collections.defaultdict(list)
for r in rows:
    by[r['Queue_Id']].append(r)
dur = lambda r: int(r['End_Timestamp']) - int(r['Start_Timestamp'])
main = max(by.values(), key=lambda rs: sum(dur(r) for r in rs))
mark = ([a.split('=', 1)[1] for a in sys.argv if a.startswith('--mark=')] or ['adam_kernel'])[0]
idx = [i for i, r in enumerate(main) if mark in r['Kernel_Name']]
step = main[idx[-2] + 1:idx[-1] + 1]
t0, t1 = int(step[0]['Start_Timestamp']), int(step[-1]['End_Timestamp'])
short = lambda n: (n.split('(anonymous namespace)::')[1] if '(anonymous namespace)::' in n else n)[:48]
busy = gaps = 0.0
big = []
prev = None
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - prev) / 1e3 if prev else 0.0
    busy += (e - s) / 1e3
    gaps += max(0.0, gap)
    wg = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // (int(r['Workgroup_Size_X']) * int(r['Workgroup_Size_Y']))
    if '-v' in sys.argv:
        print('%8.1f  +%5.1f  %7.1f us  wgs %6d  %s' % ((s - t0) / 1e3, gap, (e - s) / 1e3, wg, short(r['Kernel_Name'])))
    if gap > 12:
        big.append((round((s - t0) / 1e3), round(gap, 1), short(r['Kernel_Name'])[:30]))
    prev = e
print('step %.1f us: main stream busy %.1f, gaps %.1f (%d launches)' % ((t1 - t0) / 1e3, busy, gaps, len(step)))
for q, rs in by.items():
    if rs is main:
        continue
    inside = [r for r in rs if t0 <= int(r['Start_Timestamp']) <= t1]
    if inside:
        print('other stream %s: busy %.1f us in %d launches, last ends %.1f us before the step does' %
              (q, sum(dur(r) for r in inside) / 1e3, len(inside), (t1 - max(int(r['End_Timestamp']) for r in inside)) / 1e3))
print('gaps > 12 us (start, gap, kernel):', big)
