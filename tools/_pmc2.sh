OUT=$GRAFT_REPO_ROOT/gpurun_out/r3as
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 tools/batch_rate.py 8 150 400 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 tools/pmc_table.py $OUT > $OUT/pmc_summary.txt 2>&1 || true
rm -rf $OUT/p[0-9]*/
python3 - <<'PY'
import json,os
d=json.load(open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/r3as/pmc.json"))
def g(k,c): return d[k].get(c,{}).get("mean",float("nan"))
for k in sorted(d, key=lambda k:-g(k,"SQ_WAVE_CYCLES") if k.startswith("k_") else 0):
    if not k.startswith("k_"): continue
    w=g(k,"SQ_WAVES"); wc=g(k,"SQ_WAVE_CYCLES")
    print("%-28s waves %6.0f qc/wave %6.0f valu/w %5.0f salu/w %5.0f lds/w %4.0f vmem/w %3.0f | wait %4.1f iss %4.1f act %4.1f ldswait %4.1f bankconf/w %5.0f | us %5.1f"%(k[:28],w,wc/w,g(k,"SQ_INSTS_VALU")/w,g(k,"SQ_INSTS_SALU")/w,g(k,"SQ_INSTS_LDS")/w,g(k,"SQ_INSTS_VMEM")/w,100*g(k,"SQ_WAIT_ANY")/wc,100*g(k,"SQ_WAIT_INST_ANY")/wc,100*g(k,"SQ_ACTIVE_INST_ANY")/wc,100*g(k,"SQ_WAIT_INST_LDS")/wc,g(k,"SQ_LDS_BANK_CONFLICT")/w,g(k,"GRBM_GUI_ACTIVE")/8/2.1e3))
PY
