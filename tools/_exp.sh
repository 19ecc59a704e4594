for rep in 1 2 3; do
for v in new old; do
if [ $v = old ]; then export REBVIO_HIP_BATCH_LANE_READY=1; else unset REBVIO_HIP_BATCH_LANE_READY; fi
echo -n "$v: "; timeout -k 10 120 python3 tools/batch_rate.py 8 1200 800 2>&1 | grep -h "lanes 8"
done; done
