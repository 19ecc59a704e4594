for m in spec seq; do
echo "LM=$m"; REBVIO_HIP_LM=$m REBVIO_HIP_LM_STAMPS=1 timeout -k 10 120 python3 bench.py --no-cpu-baseline --lanes 0 --no-host-class --no-pcie --steps 600 2>&1 >/dev/null | grep "of which\|end of an LM"
done
