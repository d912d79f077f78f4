#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: PMC passes (--pmc only, one counter group per run) for the layered
# encoder kernels on 512 frames of 224x224x3; summaries go to gpurun_out/profiles/encoder_general_pmc_<group>.csv
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
  rm -rf /tmp/pmc_eg
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_eg -o pmc -- python $R/profiles/probes/encoder_general_prof.py > /dev/null 2> /tmp/pmc_eg_err.log
  f=$(find /tmp/pmc_eg -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $R/profiles/summarize_pmc.py $f $OUT/encoder_general_pmc_$i.csv; grep enc_ $OUT/encoder_general_pmc_$i.csv | cut -c1-400; else echo "group $i failed"; tail -3 /tmp/pmc_eg_err.log; fi
done
