# MFMA-busy and LDS/wave activity of the implicit-GEMM kernels over eager UNet evaluations (separate --pmc pass, no tracing)
export DTP_TUNE_CACHE=/tmp/tc_pmc.txt
timeout 300 python tools/pmc_unet.py > gpurun_out/pmc_unet_warm.log 2>&1
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $grp --output-format csv -d /tmp/pmcm_$i -o p -- python /root/repo/tools/pmc_unet.py > /tmp/pmcm_$i.log 2>&1
  f=$(find /tmp/pmcm_$i -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python /root/repo/tools/pmc_agg.py $f /root/repo/gpurun_out/r02_pmc_unet_mfma_$i.csv; else tail -5 /tmp/pmcm_$i.log > /root/repo/gpurun_out/r02_pmc_unet_mfma_$i.err; fi
done
