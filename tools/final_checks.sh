set -o pipefail
O=gpurun_out/s2w; mkdir -p $O
timeout -k 10 300 python3 tools/fuzz_parity.py --trials 400 --seed 53 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.txt
timeout -k 10 200 python3 tools/stress_stream.py 100 > $O/soak.txt 2>&1; echo "soak rc=$?"; tail -3 $O/soak.txt
timeout -k 10 200 python3 tools/stress_stream.py 60 1 > $O/soak_jumps.txt 2>&1; echo "soak jumps rc=$?"; tail -3 $O/soak_jumps.txt
timeout -k 10 500 python3 tools/repeat_tests.py 8 "stream or failure or nan_path or flush or glue" > $O/repeat.txt 2>&1; tail -3 $O/repeat.txt
