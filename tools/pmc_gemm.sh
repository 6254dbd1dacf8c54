#!/bin/bash
# Counter passes over the decode GEMM phase of the headline step (GPT-XL t2v, 32 rows, bf16): 48 eager decode steps of bench.py under
# rocprofv3 --pmc, one pass per counter group (TCC has 4 slots, FETCH_SIZE takes 3), no trace domain beside --kernel-trace.
# Run on the GPU box from the repo root: bash tools/pmc_gemm.sh  ->  gpurun_out/pmc_gemm/<pass>/..., gpurun_out/r04_pmc_gemm_raw.json
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_gemm
rm -rf $OUT; mkdir -p $OUT
rocprofv3 -L > $OUT/counters_available.txt 2>&1
CMD="python3 $PWD/bench.py --new-tokens 48 --no-vae --no-extras --no-cpu-baseline --no-roofline --no-graph --steps 1 --warmup 0"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum TCC_READ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  ( cd /tmp && rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- $CMD > $OUT/p$i.log 2>&1 ) || echo "pass $i ($grp) failed" >> $OUT/failed.txt
  echo "pass $i done: $grp"
done
python3 tools/pmc_summarize.py gpurun_out/r04_pmc_gemm_raw.json gemm_fused_kernel,attn_combine,attn_partial $OUT/p* > $OUT/summary.txt 2>&1
rm -rf $OUT/p[0-9]*/   # the raw per-dispatch CSVs are tens of MB: only the summary travels back
grep -A40 gemm_fused $OUT/summary.txt | head -230
