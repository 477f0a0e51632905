"""Sum rocprofv3 --pmc counter_collection CSVs per kernel name.  python tools/pmc_summary.py <csv> [steps]"""
import collections
import csv
import re
import sys


def main(path, steps=1):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.Counter()
    for r in rows:
        name = re.sub(r'\(anonymous namespace\)::', '', r['Kernel_Name'])[:70]
        agg[name][r['Counter_Name']] += float(r['Counter_Value'])
        calls[(name, r['Counter_Name'])] += 1
    tot = collections.defaultdict(float)
    print(f'# {path}: counter totals per kernel (sum over dispatches), {steps} timed+warm steps')
    for name, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
        for c, v in cs.items():
            tot[c] += v
            print(f'{name:72s} {c:14s} {v:16.0f}  n={calls[(name, c)]}')
    for c, v in tot.items():
        print(f'TOTAL {c}: {v:.0f} (raw units: KB for FETCH_SIZE/WRITE_SIZE) -> per step {v / steps * 1024 / 1e6:.1f} MB')


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
