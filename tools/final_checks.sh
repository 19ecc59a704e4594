# End-of-round checks on the GPU box beyond `pytest -m gpu` (one call each, no repeat loops: a record read before its pair wrote
# it is a hard error since round 4 - sequence stamps, status -12 - and the fresh-map fill race behind round 3's flaky records is
# fixed at its cause): the parity fuzzer over random sizes / budgets for the default directedMatch form and the
# one-lane form, a soak of the streaming driver on consecutive frames and on random jumps.
set -o pipefail
O=gpurun_out/final_${1:-r04}; mkdir -p $O
timeout -k 10 300 python3 tools/fuzz_parity.py --trials 300 --seed 53 > $O/fuzz.txt 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz.txt
REBVIO_HIP_DM_HEAD=compact1 timeout -k 10 300 python3 tools/fuzz_parity.py --trials 150 --seed 54 > $O/fuzz_compact1.txt 2>&1; echo "fuzz (compact1) rc=$?"; tail -2 $O/fuzz_compact1.txt
timeout -k 10 200 python3 tools/stress_stream.py 100 > $O/soak.txt 2>&1; echo "soak rc=$?"; tail -3 $O/soak.txt
timeout -k 10 200 python3 tools/stress_stream.py 60 1 > $O/soak_jumps.txt 2>&1; echo "soak jumps rc=$?"; tail -3 $O/soak_jumps.txt
