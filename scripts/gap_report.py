"""Idle-time report from a rocprofv3 --kernel-trace CSV: where does the GPU wait between kernels?

usage: python scripts/gap_report.py <kernel_trace.csv> [steps] [last_ms]   (last_ms: only the final window of the trace)
Prints total busy / idle time, the largest idle gaps with the kernels either side, and idle time grouped by the
kernel that FOLLOWS the gap (= the launch the host was late with).
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(.*', '', name)
    name = re.sub(r'^void ', '', name)
    return name[-70:]


def main():
    path = sys.argv[1]
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    if len(sys.argv) > 3:
        lo = max(r[1] for r in rows) - int(float(sys.argv[3]) * 1e6)
        rows = [r for r in rows if r[0] >= lo]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy = 0
    cur_end = rows[0][0]
    gaps = []
    by_next = defaultdict(lambda: [0, 0])
    prev = None
    for s, e, n in rows:
        if s > cur_end:
            g = s - cur_end
            gaps.append((g, prev, n, s - t0))
            k = short(n)
            by_next[k][0] += g
            by_next[k][1] += 1
        busy += max(0, e - max(s, cur_end))
        if e > cur_end:
            cur_end = e
            prev = n
    span = t1 - t0
    print(f'kernels {len(rows)}  span {span/1e6:.2f} ms  busy {busy/1e6:.2f} ms  idle {(span-busy)/1e6:.2f} ms'
          f'  (per step: busy {busy/1e6/steps:.2f}, idle {(span-busy)/1e6/steps:.2f})')
    gaps.sort(reverse=True)
    print('\nlargest gaps:')
    for g, a, b, at in gaps[:25]:
        print(f'  {g/1e3:9.1f} us at {at/1e6:9.2f} ms  after {short(a or "")[:50]:50s} before {short(b)[:50]}')
    print('\nidle grouped by the kernel after the gap:')
    for k, (g, c) in sorted(by_next.items(), key=lambda kv: -kv[1][0])[:30]:
        print(f'  {g/1e6:8.3f} ms  {c:6d} gaps  avg {g/c/1e3:7.1f} us  {k}')
    hist = defaultdict(int)
    for g, *_ in gaps:
        b = 1
        while b < g / 1e3:
            b *= 2
        hist[b] += g
    print('\nidle by gap size (<= us): ' + '  '.join(f'{b}:{v/1e6:.2f}ms' for b, v in sorted(hist.items())))


if __name__ == '__main__':
    main()
