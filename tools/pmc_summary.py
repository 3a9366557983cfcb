#!/usr/bin/env python3
"""usage: pmc_summary.py <dir_fetch> <dir_write> <out.txt>   (rocprofv3 --pmc csv outputs of tools/pmc_probe.py)"""
import csv
import glob
import sys
from collections import defaultdict


def load(d, counter):
    f = glob.glob(d + '/*counter_collection.csv')[0]
    per = defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter:
            per[r['Kernel_Name']].append(float(r['Counter_Value']))
    return per


def main():
    fetch = load(sys.argv[1], 'FETCH_SIZE')
    write = load(sys.argv[2], 'WRITE_SIZE')
    lines = ['# HBM traffic per launch from rocprofv3 PMC (separate passes: --pmc FETCH_SIZE ; --pmc WRITE_SIZE), workload tools/pmc_probe.py',
             '# raw counter units are KiB as reported; calibration = torch copy of 1 GiB (read 1 GiB + write 1 GiB)']
    cal_f = cal_w = None
    for k, v in fetch.items():
        if 'copy' in k.lower() or 'elementwise' in k.lower():
            big = max(v)
            if big > 1e5:
                cal_f = big
    for k, v in write.items():
        if 'copy' in k.lower() or 'elementwise' in k.lower():
            big = max(v)
            if big > 1e5:
                cal_w = big
    gib_kib = float(1 << 20)
    lines.append(f'calibration copy: FETCH_SIZE={cal_f} KiB (expected {gib_kib:.0f}) -> read factor {gib_kib / cal_f if cal_f else float("nan"):.3f}; '
                 f'WRITE_SIZE={cal_w} KiB -> write factor {gib_kib / cal_w if cal_w else float("nan"):.3f}')
    ff = gib_kib / cal_f if cal_f else 1.0
    fw = gib_kib / cal_w if cal_w else 1.0
    for name in sorted(set(fetch) | set(write)):
        if 'rdr::' not in name and 'kernel' not in name:
            continue
        f = fetch.get(name, [0]); w = write.get(name, [0])
        fa = sum(f) / len(f); wa = sum(w) / len(w)
        lines.append(f'{name[:110]}\n    launches={len(f)}  FETCH_SIZE avg {fa:.0f} KiB  WRITE_SIZE avg {wa:.0f} KiB  '
                     f'-> corrected read {fa * ff * 1024 / 1e9:.3f} GB, write {wa * fw * 1024 / 1e9:.3f} GB per launch')
    open(sys.argv[3], 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
