set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$REPO/gpurun_out"; mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
for c in "C2 100k" "C3 1M" "41k points" "200k points"; do
  tag=$(echo "$c" | tr ' ' '_')
  rm -rf "$OUT/tr_$tag"
  ONLY="$c" REPS=3 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr_$tag" -- python $REPO/tools/time_whole_call.py > "$OUT/tr_$tag.log" 2>&1
  echo "== $c" >> "$OUT/s5_trace_calls.txt"
  tail -1 "$OUT/tr_$tag.log" >> "$OUT/s5_trace_calls.txt"
  python $REPO/tools/trace_whole_call.py "$OUT/tr_$tag" >> "$OUT/s5_trace_calls.txt" 2>&1
  rm -rf "$OUT/tr_$tag"
done
