# PMC passes over the level-0 attention launch: bash tools/pmc_attn.sh <tag>   (from the repo root on the GPU box)
tag=$1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rm -rf /tmp/pmca_$tag$i
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmca_$tag$i -o p -- python /root/repo/tools/pmc_attn.py > /tmp/pmca_$tag$i.log 2>&1
  f=$(find /tmp/pmca_$tag$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python /root/repo/tools/pmc_agg.py $f /root/repo/gpurun_out/pmca_${tag}_$i.csv; else tail -5 /tmp/pmca_$tag$i.log > /root/repo/gpurun_out/pmca_${tag}_$i.err; fi
done
cd /root/repo
