#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace [--memory-copy-trace] csv of `ONLY=<case> REPS=3 python tools/time_whole_call.py` and prints
the LAST call's kernels AND copies in time order: start offset, gap to the previous operation's end (gaps = host round trips /
syncs), duration; for the copies the achieved GB/s when the sizes are given (argv[2]: "bytes,bytes,..." -- the uploads of one
call, largest first is fine: a copy is matched to the given size whose rate is physically plausible, else printed without)."""
import csv, glob, sys
d = sys.argv[1]
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 and sys.argv[2] else []
f = sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True))[-1]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("oa::", ""), "k")
        for r in csv.DictReader(open(f))]
cf = sorted(glob.glob(d + "/**/*_memory_copy_trace.csv", recursive=True))
if cf:
    rows += [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Direction"].replace("MEMORY_COPY_", "copy "), "c") for r in csv.DictReader(open(cf[-1]))]
rows.sort()
# the last call starts at the last k_pack_target (set_target's first kernel) -- or at the copy that feeds it
starts = [i for i, r in enumerate(rows) if "k_pack_target" in r[2]]
i0 = starts[-1]
while i0 > 0 and rows[i0 - 1][3] == "c" and rows[i0][0] - rows[i0 - 1][1] < 300_000:
    i0 -= 1
t0 = rows[i0][0]
prev_end = t0
busy = copy_ns = 0
left = sorted(sizes, reverse=True)
for s, e, n, kind in rows[i0:]:
    extra = ""
    if kind == "c":
        copy_ns += e - s
        dur = (e - s) * 1e-9
        # the large uploads: longest copies first take the largest sizes (a 12 MB copy cannot be the 4-byte state word)
        cand = [b for b in left if 1.0 < b / dur / 1e9 < 70.0]
        if cand:
            b = cand[0]; left.remove(b)
            extra = "   %.1f MB at %.1f GB/s" % (b / 1e6, b / dur / 1e9)
    print("%9.1f us  +%7.1f gap  %8.1f us  %s%s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, n[:90], extra))
    prev_end = max(prev_end, e); busy += e - s
print("kernels + copies: %.1f us busy (copies %.1f us) of %.1f us span" % (busy / 1e3, copy_ns / 1e3, (prev_end - t0) / 1e3))
