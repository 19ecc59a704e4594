#!/bin/bash
# Collects the round's profile records on the GPU box into gpurun_out/prof_rNN (copy the summaries into profiles/ afterwards):
#   kernel stats + track-stream timeline of the default bench command (steady state), kernel stats of config c3 and of 4- and
#   8-lane batches; two PMC passes (HBM bytes) of the bench command; the counter groups of tools/collect_pmc.sh for an 8-lane batch
set -e
R=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="bench.py --no-cpu-baseline --lanes 0 --no-host-class --no-pcie"
echo "kernel trace c2"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2 -- python3 $BENCH --steps 600 --warmup 1000 > $OUT/c2.log 2>&1
echo "kernel trace c3"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c3 -- python3 $BENCH --config c3 --steps 300 > $OUT/c3.log 2>&1
echo "kernel trace batch4"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b4 -- python3 tools/batch_rate.py 4 400 800 > $OUT/b4.log 2>&1
echo "kernel trace batch8"; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/b8 -- python3 tools/batch_rate.py 8 400 800 > $OUT/b8.log 2>&1
echo "pmc fetch"; rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- python3 $BENCH --steps 200 --warmup 400 > $OUT/pmc_fetch.log 2>&1
echo "pmc write"; rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write -- python3 $BENCH --steps 200 --warmup 400 > $OUT/pmc_write.log 2>&1
for d in c2 c3 b4 b8; do f=$(ls $OUT/$d/*/*_kernel_stats.csv | head -1); cp $f $OUT/${d}_kernel_stats.csv; python3 tools/kstats.py $f > $OUT/${d}_kernel_stats.txt; done
python3 tools/trace_gaps.py $(ls $OUT/c2/*/*_kernel_trace.csv | head -1) > $OUT/c2_track_timeline.txt 2>&1 || true
python3 tools/pmc_summary.py $(ls $OUT/pmc_fetch/*/*counter_collection.csv | head -1) $(ls $OUT/pmc_write/*/*counter_collection.csv | head -1) > $OUT/pmc_hbm.json
rm -rf $OUT/c2 $OUT/c3 $OUT/b4 $OUT/b8 $OUT/pmc_fetch $OUT/pmc_write
head -14 $OUT/c2_kernel_stats.txt
