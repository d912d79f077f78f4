mkdir -p gpurun_out/r04c
python -m pytest tests -m gpu -x -q -s > gpurun_out/r04c/gputests.log 2>&1; tail -4 gpurun_out/r04c/gputests.log
SRLHIP_LIB=$PWD/robotics-rl-srl_amd/csrc/build/libsrlhip_prof.so python profiles/probes/kuka_tree_phases.py 2048 > gpurun_out/r04c/phases.txt 2>&1; tail -30 gpurun_out/r04c/phases.txt
SRLHIP_SINGLE_DEVICE=1 SRLHIP_DIST_BACKEND=gloo timeout 600 python bench.py --workload kuka_pixels --gpus 8 > gpurun_out/r04c/bench_pixels_gpus8_single_device.json 2> gpurun_out/r04c/bench8.err; tail -c 400 gpurun_out/r04c/bench_pixels_gpus8_single_device.json
R=$PWD; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_ic
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_IFETCH --output-format csv -d /tmp/pmc_ic -o pmc -- python $R/bench.py --workload kuka --no-cpu-baseline --no-secondary --no-live-pmc --steps 3 --warmup 1 > /dev/null 2>/tmp/ic_err.log
f=$(find /tmp/pmc_ic -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then python $R/profiles/summarize_pmc.py $f $R/gpurun_out/r04c/kuka_pmc_icache.csv; grep kuka_tree_rollout_k $R/gpurun_out/r04c/kuka_pmc_icache.csv; else echo "icache pmc failed"; tail -3 /tmp/ic_err.log; fi
