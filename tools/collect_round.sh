#!/bin/bash
# One GPU call that produces everything profiles/rNN_* is made of: tools/collect_profiles.sh (kernel stats, HBM PMC passes),
# the counter groups of an 8-lane batch, the bench lines (default command, the driver's flags, config c3), LM stamps.
R=${1:-r03}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
bash tools/collect_profiles.sh $R > $OUT/collect.log 2>&1 || { tail -5 $OUT/collect.log; exit 1; }
bash tools/collect_pmc.sh prof_$R/pmc_b8 tools/batch_rate.py 8 150 500 > $OUT/pmc_b8.log 2>&1
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err && echo "bench default done" &&
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err && echo "bench driver flags done" &&
python3 bench.py --config c3 --no-cpu-baseline --lanes 0 --no-host-class --no-pcie > $OUT/bench_c3.json 2> $OUT/bench_c3.err &&
REBVIO_HIP_LM_STAMPS=1 python3 bench.py --no-cpu-baseline --lanes 0 --no-host-class --no-pcie --steps 600 > /dev/null 2> $OUT/lm_stamps.txt
tail -c 600 $OUT/bench.json
