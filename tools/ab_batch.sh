#!/bin/bash
# A/B of environment variants on a batch: tools/ab_batch.sh OUTDIR LANES "VAR=a" "VAR=b" ... ("-" = no variables)
out=$1; shift
lanes=$1; shift
mkdir -p "$out"
i=0
for v in "$@"; do
  i=$((i+1))
  if [ "$v" = "-" ]; then v=""; fi
  echo "== batch variant $i: ${v:-default}" | tee -a "$out/abb.txt"
  env $v timeout -k 10 300 python tools/batch_rate.py $lanes 1200 800 > "$out/abb_$i.log" 2>&1 || { tail -20 "$out/abb_$i.log"; exit 1; }
  grep "^lanes" "$out/abb_$i.log" | tee -a "$out/abb.txt"
done
