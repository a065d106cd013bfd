#!/usr/bin/env python3
"""tools/kstats_rank.py <rocprofv3 kernel_stats.csv> [sets=20] -- per launch set of 8: microseconds and launches per kernel family,
convolutions and the rest apart (the table VERDICT's launch-diet item is judged on)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
sets = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
conv = ('k_sparse_conv', 'k_conv_rowlane', 'k_conv_row32', 'k_conv_tapc', 'k_bev_conv3x3', 'k_deconv_head', 'k_const_conv125')
acc = {}
for r in rows:
    n = r['Name']
    m = re.search(r'(k_\w+|wrapped_\w+|init_lookback\w+|__amd_rocclr_\w+)', n)
    key = m.group(1) if m else n[:40]
    if 'rocprim' in n:
        key = 'rocprim:' + key
    if key == 'k_bev_conv3x3_list': key = 'k_bev_conv3x3'
    a = acc.setdefault(key, [0, 0.0]); a[0] += int(r['Calls']); a[1] += float(r['TotalDurationNs'])
for title, sel in (("convolutions", True), ("everything else", False)):
    tot = 0; ln = 0
    print("== %s (per launch set of 8)" % title)
    for k, (c, t) in sorted(acc.items(), key=lambda x: -x[1][1]):
        if any(k.startswith(cv) for cv in conv) != sel: continue
        print("  %-46s launches %6.1f  us %8.1f" % (k, c / sets, t / 1e3 / sets)); tot += t; ln += c
    print("  TOTAL launches %.1f  us %.1f  (per window: %.1f launches, %.4f ms)" % (ln / sets, tot / 1e3 / sets, ln / sets / 8, tot / 1e6 / sets / 8))
