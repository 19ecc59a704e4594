for rep in 1 2 3; do
echo -n "b8: "; timeout -k 10 120 python3 tools/batch_rate.py 8 1200 800 2>&1 | grep "lanes 8"
echo -n "b4: "; timeout -k 10 120 python3 tools/batch_rate.py 4 1200 800 2>&1 | grep "lanes 4"
done
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "batch or lanes" 2>&1 | tail -2
