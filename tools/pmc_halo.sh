# PMC passes over the level-0 halo conv (3 x 64 x 64, 320 -> 320, tile 12) and the batch-8 one: what the SIMDs wait for
cd /tmp && export TMPDIR=/tmp
for cfg in "12 1 3 64 320 320" "12 1 24 64 320 320"; do
tag=$(echo $cfg | tr ' ' '_')
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" "GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC" "SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES" "SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_WAIT_INST_LDS" "TCP_PENDING_STALL_CYCLES TCP_TCC_READ_REQ_sum TCP_GATE_EN1_sum TA_BUSY_avr"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmch${tag}_$i -o p -- python /root/repo/tools/pmc_halo.py $cfg > /tmp/pmch$i.log 2>&1
  f=$(find /tmp/pmch${tag}_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python /root/repo/tools/pmc_agg.py $f /root/repo/gpurun_out/pmch_${tag}_$i.csv; else tail -5 /tmp/pmch$i.log > /root/repo/gpurun_out/pmch_${tag}_$i.err; fi
done
done
