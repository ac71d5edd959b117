"""Kernel list of ONE training step from a rocprofv3 --kernel-trace CSV: every dispatch between the last two AMSGrad
launches (name, duration, gap to the previous kernel), and the totals per kernel name.
  python scripts/step_trace.py <..._kernel_trace.csv> [min_us]"""
import collections
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 0.
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    idx = [i for i, r in enumerate(rows) if 'amsgrad' in r['Kernel_Name']]
    a, b = idx[-2], idx[-1]
    step = rows[a + 1:b + 1]
    t0 = prev_end = int(rows[a]['End_Timestamp'])
    per = collections.defaultdict(lambda: [0, 0.])
    busy = 0.
    for r in step:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        d = (e - s) / 1e3
        busy += d
        name = r['Kernel_Name'].split('(')[0][-70:]
        per[name][0] += 1
        per[name][1] += d
        if d >= min_us:
            print('%9.1f us  gap %6.1f  grid %8s  %s' % (d, (s - prev_end) / 1e3, r['Grid_Size_X'], name))
        prev_end = e
    print('step: %d kernels, span %.1f us, kernels busy %.1f us' % (len(step), (int(rows[b]['End_Timestamp']) - t0) / 1e3, busy))
    for name, (n, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:40]:
        print('  %8.1f us  x%-3d %s' % (t, n, name))


if __name__ == '__main__':
    main()
