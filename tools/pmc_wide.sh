# PMC passes over two launches of one GEMM tile: tools/pmc_wide.sh <tag> <tile> <M> <N> <K>   (run from the repo root on the GPU box)
tag=$1; tile=$2; M=$3; N=$4; K=$5
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$tag$i
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc_$tag$i -o p -- python /root/repo/tools/pmc_gemm.py $tile $M $N $K > /tmp/pmc_$tag$i.log 2>&1
  f=$(find /tmp/pmc_$tag$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python /root/repo/tools/pmc_agg.py $f /root/repo/gpurun_out/pmcw_${tag}_$i.csv; else tail -5 /tmp/pmc_$tag$i.log > /root/repo/gpurun_out/pmcw_${tag}_$i.err; fi
done
cd /root/repo
