#!/bin/bash
mkdir -p gpurun_out
timeout 900 python tools/bench_attn.py > gpurun_out/r05_bench_attn_variants.log 2>&1
cat gpurun_out/r05_bench_attn_variants.log
bash tools/pmc_attn.sh dma > /dev/null 2>&1
DTP_ATTN_DMA=0 bash tools/pmc_attn.sh old > /dev/null 2>&1
for f in gpurun_out/pmca_dma_*.csv gpurun_out/pmca_old_*.csv; do echo == $f; grep -i "attn\|attention\|Kernel" $f | head -5; done
