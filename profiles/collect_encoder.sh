#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: PMC passes (--pmc only, one counter group per run) for the fused
# encoder kernel on a 4096-frame batch; summaries go to gpurun_out/profiles/encoder_pmc_<group>.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/profiles
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for pmc in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_CYCLES_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_enc
  ENC_REPS=2 timeout 300 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_enc -o pmc -- python $R/profiles/encoder_microbench.py > /dev/null 2> /tmp/pmc_enc_err.log
  f=$(find /tmp/pmc_enc -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $R/profiles/summarize_pmc.py $f $OUT/encoder_pmc_$i.csv; grep encoder_fwd_k $OUT/encoder_pmc_$i.csv; else echo "group $i failed"; tail -3 /tmp/pmc_enc_err.log; fi
done
