set -o pipefail
O=gpurun_out/s2p; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/gputests.log
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err && python3 -c "
import json; d=json.loads(open('$O/bench_driver.json').read().strip().splitlines()[-1]); print('driver flags', d['value'], d['config'].get('long_window',{}).get('value'), d['config']['pcie_inclusive_fps'], d['config']['host_class_fps'], [(x['lanes'],x['value']) for x in d.get('streams_per_gpu',[])], d['cpu_baseline']['value'])"
