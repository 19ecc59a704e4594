#!/bin/bash
# The GPU calls that produce everything profiles/rNN_* is made of (each stays inside one gpurun call of <= 20 minutes):
#   tools/collect_round.sh rNN profiles   kernel stats (c2, c3, 4- and 8-lane batches), track timeline, HBM PMC passes of the bench command
#   tools/collect_round.sh rNN bench      the bench lines (default command, the driver's flags, config c3), LM stamps + directedMatch wave stats
#   tools/collect_round.sh rNN pmc        the counter groups of an 8-lane batch
R=${1:-r04}
WHAT=${2:-profiles}
OUT=gpurun_out/prof_$R
mkdir -p $OUT
case $WHAT in
  profiles)
    bash tools/collect_profiles.sh $R > $OUT/collect.log 2>&1 || { tail -5 $OUT/collect.log; exit 1; }
    tail -16 $OUT/collect.log ;;
  bench)
    python3 bench.py > $OUT/bench.json 2> $OUT/bench.err && echo "bench default done" &&
    python3 bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench_driver_flags.err && echo "bench driver flags done" &&
    python3 bench.py --config c3 --no-cpu-baseline --lanes 0 --no-host-class --no-pcie > $OUT/bench_c3.json 2> $OUT/bench_c3.err && echo "bench c3 done" &&
    REBVIO_HIP_LM_STAMPS=1 REBVIO_HIP_DM_STATS=1 python3 bench.py --no-cpu-baseline --lanes 0 --no-host-class --no-pcie --steps 600 > /dev/null 2> $OUT/lm_stamps.txt
    grep "rebvio_hip" $OUT/lm_stamps.txt | cut -c1-400
    tail -c 400 $OUT/bench.json ;;
  pmc)
    bash tools/collect_pmc.sh prof_$R/pmc_b8 tools/batch_rate.py 8 150 500 > $OUT/pmc_b8.log 2>&1
    tail -5 $OUT/pmc_b8.log ;;
esac
