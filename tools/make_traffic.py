#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE passes of rocprofv3 -> profiles/traffic_<cfg>.json (see tools/traffic.sh).

Per dispatch the counter rows are summed; per kernel the FULL-SIZE launches (>= half of the largest
dispatch of that kernel: the 1-resample decomposition launches are left out) are averaged.  Units and
corrections as MI355X_MICROARCH.md prescribes: both counters are in KiB; FETCH_SIZE is doubled on gfx950
(128-B requests tallied at 64 B); WRITE_SIZE is taken as is (calibrated at factor 1.000 in round 2 on
the known byte count of k_xprod's R store, profiles/traffic_c4.json)."""
import collections
import csv
import json
import sys


def per_dispatch(path, counter):
    acc = collections.defaultdict(float)
    name = {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        acc[r['Dispatch_Id']] += float(r['Counter_Value'])
        name[r['Dispatch_Id']] = r['Kernel_Name'].split('(')[0].replace('void ', '')
    by_kernel = collections.defaultdict(list)
    for d, v in acc.items():
        by_kernel[name[d]].append(v * 1024.0)               # KiB -> bytes
    return by_kernel


def full_size_mean(vals):
    big = [v for v in vals if v >= 0.5 * max(vals)]
    return sum(big) / len(big), len(big)


def main():
    cfg, fetch_csv, write_csv, out = sys.argv[1:5]
    fe, wr = per_dispatch(fetch_csv, 'FETCH_SIZE'), per_dispatch(write_csv, 'WRITE_SIZE')
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        if not k.startswith('k_'):
            continue
        f, nf = full_size_mean(fe[k]) if k in fe else (0.0, 0)
        w, nw = full_size_mean(wr[k]) if k in wr else (0.0, 0)
        kernels[k] = {'full_size_launches': max(nf, nw), 'fetch_bytes_raw': f, 'fetch_bytes_corrected': 2.0 * f,
                      'write_bytes': w, 'hbm_bytes_per_launch': 2.0 * f + w,
                      'total_over_run': 2.0 * sum(fe.get(k, [])) + sum(wr.get(k, []))}
    xp = {k: v for k, v in kernels.items() if k.startswith('k_xprod')}
    dom = max(xp, key=lambda k: xp[k]['total_over_run']) if xp else max(kernels, key=lambda k: kernels[k]['total_over_run'])
    rec = {'config': cfg, 'kernel': dom, 'hbm_bytes_per_launch': kernels[dom]['hbm_bytes_per_launch'],
           'source': 'tools/traffic.sh {}: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of '
                     '`bench.py --config {} --cpu-sample 0 --steps 1 --warmup 1 --no-primal`; FETCH doubled (gfx950), '
                     'KiB -> bytes, full-size launches averaged'.format(cfg, cfg),
           'kernels': kernels}
    json.dump(rec, open(out, 'w'), indent=1)
    print(json.dumps({'config': cfg, 'kernel': dom, 'hbm_GB_per_launch': rec['hbm_bytes_per_launch'] / 1e9}))


if __name__ == '__main__':
    main()
