#!/bin/bash
# A/B of environment variants on the single-stream rate: tools/ab_stream.sh OUTDIR "VAR=a VAR2=b" "VAR=c" ...
# ("-" = no variables). Each variant: tools/short_window.py (20-frame windows + a 3000-frame window) in a fresh process.
out=$1; shift
mkdir -p "$out"
i=0
for v in "$@"; do
  i=$((i+1))
  if [ "$v" = "-" ]; then v=""; fi
  echo "== variant $i: ${v:-default}" | tee -a "$out/ab.txt"
  env $v timeout -k 10 300 python tools/short_window.py 20 30 1000 > "$out/ab_$i.log" 2>&1 || { tail -20 "$out/ab_$i.log"; exit 1; }
  grep -v "^\[Gloo\]" "$out/ab_$i.log" | tail -7 | cut -c1-900 | tee -a "$out/ab.txt"
done
