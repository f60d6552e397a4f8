#!/bin/bash
# GPU box: rocprofv3 --kernel-trace --memory-copy-trace of whole alignment calls (tools/time_whole_call.py), the last call of each case printed kernel
# by kernel -- start offset, gap to the previous kernel's end (host round trips, enqueue-bound stretches), duration
# (tools/trace_whole_call.py).  CASES="C1 2562|C2 100k|..." selects (substrings of the case names).
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
IFS='|' read -ra CS <<< "${CASES:-C1 2562|C2 100k|C3 1M|41k points|200k points}"
rm -f "$OUT/trace_calls.txt"
for c in "${CS[@]}"; do
  tag=$(echo "$c" | tr ' ' '_')
  rm -rf "$OUT/tr_$tag"
  ONLY="$c" REPS=5 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d "$OUT/tr_$tag" -- python $REPO/tools/time_whole_call.py > "$OUT/tr_$tag.log" 2>&1
  echo "== $c" >> "$OUT/trace_calls.txt"
  line=$(grep "uploads_bytes=" "$OUT/tr_$tag.log" | tail -1)
  echo "${line% uploads_bytes=*}  (under rocprofv3)" >> "$OUT/trace_calls.txt"
  python $REPO/tools/trace_whole_call.py "$OUT/tr_$tag" "${line##*uploads_bytes=}" >> "$OUT/trace_calls.txt" 2>&1
  rm -rf "$OUT/tr_$tag" "$OUT/tr_$tag.log"
done
