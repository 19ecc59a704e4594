set -o pipefail
O=gpurun_out/s2n; mkdir -p $O
timeout -k 10 700 python -m pytest tests -m gpu -x -q > $O/gputests.log 2>&1; echo "pytest rc=$?"; tail -3 $O/gputests.log
for rep in 1 2; do for v in 1 0; do
echo "fuse_dog=$v"; REBVIO_HIP_FUSE_DOG=$v REBVIO_HIP_DEBUG=1 timeout -k 10 120 python3 tools/short_window.py 20 40 2>&1 | grep -v amdgpu | tail -3
done; done
