#!/bin/bash
# A/B of the scan kernels' chain forms under rocprofv3 (kernel stats of a short default bench run per variant).
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/scan_ab
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # name, env...
  name=$1; shift
  echo "variant $name"
  env "$@" true
  for kv in "$@"; do export "$kv"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name -- python3 bench.py --steps 600 --warmup 600 --no-cpu-baseline --lanes 0 --no-host-class > $OUT/$name.log 2>&1
  for kv in "$@"; do unset "${kv%%=*}"; done
  f=$(ls $OUT/$name/*/*_kernel_stats.csv | head -1)
  python3 tools/kstats.py $f > $OUT/$name.txt
  rm -rf $OUT/$name
  grep -h "rowscan\|colscan\|lm_chain" $OUT/$name.txt
  grep -o '"value": [0-9.]*' $OUT/$name.log | head -1
}
run default REBVIO_X=1
run col_w8 REBVIO_HIP_COLSCAN=w8
run col_lane REBVIO_HIP_COLSCAN=lane
run nosplit REBVIO_HIP_SCAN_SPLIT=0
run old REBVIO_HIP_SCAN_SPLIT=0 REBVIO_HIP_COLSCAN=lane REBVIO_HIP_ROWSCAN=lane
echo "--- unprofiled rates"
for v in "REBVIO_X=1" "REBVIO_HIP_COLSCAN=lane" "REBVIO_HIP_SCAN_SPLIT=0" "REBVIO_HIP_SCAN_SPLIT=0 REBVIO_HIP_COLSCAN=lane REBVIO_HIP_ROWSCAN=lane"; do
  echo "$v: $(env $v python3 bench.py --no-cpu-baseline --lanes 0 --no-host-class 2>/dev/null | grep -o '"value": [0-9.]*' | head -1)"
done
