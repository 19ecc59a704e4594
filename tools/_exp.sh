O=gpurun_out/s2u; mkdir -p $O
for i in 1 2 3 4 5 6 7 8; do timeout -k 10 300 python -m pytest tests -m gpu -q > $O/gputests_$i.log 2>&1; echo "run $i rc=$? $(tail -1 $O/gputests_$i.log)"; done
