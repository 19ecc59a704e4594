set -o pipefail
O=gpurun_out/s2l; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "fast_gaussian or scale_space or public_cpp or host_api" > $O/gputests.log 2>&1; echo "pytest rc=$?"; tail -5 $O/gputests.log
