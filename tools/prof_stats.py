#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats run: python tools/prof_stats.py <kernel_stats.csv> [n]"""
import csv
import sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in rows[:n]:
    print('%-46s calls %5s total %9.3f ms avg %9.4f ms' % (r['Name'].split('(')[0].replace('void ', '')[:46], r['Calls'],
                                                          float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e6))
