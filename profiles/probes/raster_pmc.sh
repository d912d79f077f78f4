# raster_pmc.sh — SQ counters of raster_k / raster_grip_k / encoder_fwd_k inside the kuka_pixels bench (run on the GPU box from the repo root)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/raster
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pmc in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_INSTS_SMEM"; do
  tag=$(echo $pmc | cut -d" " -f1)
  rm -rf /tmp/pmc_r
  timeout 300 rocprofv3 --pmc $pmc --output-format csv -d /tmp/pmc_r -o pmc -- python $R/bench.py --workload kuka_pixels --no-cpu-baseline --no-secondary --no-live-pmc --steps 2 --warmup 1 > /dev/null 2>&1
  python $R/profiles/summarize_pmc.py $(find /tmp/pmc_r -name "*counter_collection.csv" | head -1) $OUT/pixels_pmc_${tag}.csv
done
grep -h "raster_k\|raster_grip_k\|encoder_fwd_k" $OUT/pixels_pmc_*.csv
