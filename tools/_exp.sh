echo -n "marker events (only form now): "; timeout -k 10 400 python3 tools/_subset.py 64 64 30 2>&1 | tail -1
