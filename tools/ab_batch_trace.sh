#!/bin/bash
# Kernel trace + per-stream timeline of a batch under environment variants: tools/ab_batch_trace.sh OUTDIR LANES "VAR=a" "VAR=b" ...
# ("-" = no variables). rocprofv3 gets the program itself (python3 ...), the variables are exported in this shell.
out=$1; shift
lanes=$1; shift
mkdir -p "$out"
cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - > /dev/null
i=0
for v in "$@"; do
  i=$((i+1))
  if [ "$v" = "-" ]; then v=""; fi
  echo "== batch trace $i: ${v:-default}" | tee -a "$out/trace.txt"
  ( for kv in $v; do export "$kv"; done
    timeout -k 10 400 rocprofv3 --kernel-trace --stats -d "$out/t$i" -o t --output-format csv -- python3 tools/batch_rate.py $lanes 700 500 > "$out/t$i.log" 2>&1 ) || { tail -20 "$out/t$i.log"; exit 1; }
  grep "^lanes" "$out/t$i.log" | tee -a "$out/trace.txt"
  f=$(find "$out/t$i" -name "*kernel_trace.csv" | head -1)
  python3 tools/batch_timeline.py "$f" 200 2>&1 | cut -c1-200 | tee -a "$out/trace.txt"
  find "$out/t$i" -name "*.csv" ! -name "*kernel_stats.csv" -delete
done
