#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace csv of `ONLY=<case> REPS=3 python tools/time_whole_call.py` and prints the LAST call's
kernels in time order: start offset, duration, gap to the previous kernel's end (gaps = host round trips / syncs)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("oa::", "")) for r in csv.DictReader(open(f))]
rows.sort()
# the last call starts at the last k_pack_target (set_target's first kernel)
starts = [i for i, r in enumerate(rows) if "k_pack_target" in r[2]]
i0 = starts[-1]
# include what precedes it within 300 us (copies are not kernels)
t0 = rows[i0][0]
prev_end = t0
tot = 0
for s, e, n in rows[i0:]:
    print("%9.1f us  +%7.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, n[:90]))
    prev_end = max(prev_end, e); tot += e - s
print("kernels: %.1f us busy of %.1f us span" % (tot / 1e3, (prev_end - t0) / 1e3))
