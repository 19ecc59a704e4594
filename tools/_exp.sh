set -o pipefail
O=gpurun_out/s2f; mkdir -p $O
for w in 1 0 1 0; do
echo "== DETECT_WORKER=$w batches"
for L in 4 8; do REBVIO_HIP_DETECT_WORKER=$w REBVIO_HIP_DEBUG=1 timeout -k 10 120 python3 tools/batch_rate.py $L 1200 800 > $O/b${L}_$w.txt 2>&1 && grep -h "lanes\|host" $O/b${L}_$w.txt; done
done
