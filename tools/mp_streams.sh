#!/bin/bash
# N separate processes, one stream each, same GPU
N=$1; shift
for i in $(seq 1 $N); do
  ( "$@" python tools/multi_ctx.py 1 3000 800 2>&1 | grep "^B=" ) &
done
wait
