set -o pipefail
for rep in 1 2; do
for g in 1 0; do
echo "gyro_pre=$g"; REBVIO_HIP_GYRO_PRE=$g timeout -k 10 120 python3 tools/host_jitter.py 8 2000 --bind 2>/dev/null | tail -1
REBVIO_HIP_GYRO_PRE=$g timeout -k 10 120 python3 tools/short_window.py 20 40 2>/dev/null | tail -1
done; done
