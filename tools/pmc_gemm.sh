cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmc$i -o p -- python /root/repo/tools/pmc_gemm.py 4 > /tmp/pmc$i.log 2>&1
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python /root/repo/tools/pmc_agg.py $f /root/repo/gpurun_out/pmcg_$i.csv; else tail -5 /tmp/pmc$i.log > /root/repo/gpurun_out/pmcg_$i.err; fi
done
