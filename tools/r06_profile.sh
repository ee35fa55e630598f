#!/bin/bash
# Run ON THE GPU BOX: the round-6 evidence session — bench default + kernel trace + HBM / MFMA counters (gpu_profile.sh), the clip bench,
# the training bench at x4 and x2 + its kernel trace, the post-head and kernel-init traces, SQ counters of the fused pass.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
bash tools/gpu_profile.sh r06 > gpurun_out/r06_profile.log 2>&1
python bench.py --clip 8 --no-cpu-baseline > gpurun_out/r06_bench_clip8.json 2> gpurun_out/r06_bench_clip8.err
python bench.py --train --steps 30 --warmup 5 > gpurun_out/r06_bench_train_up4.json 2>/dev/null
python bench.py --train --train-up 2 --steps 30 --warmup 5 > gpurun_out/r06_bench_train_up2.json 2>/dev/null
bash tools/train_trace.sh r06 > gpurun_out/r06_train_trace.txt 2>&1
bash tools/trace_pan.sh > gpurun_out/r06_trace_pan.txt 2>&1
PAN_VARIANTS=1 python tools/pan_time.py >> gpurun_out/r06_trace_pan.txt 2>&1
bash tools/trace_init.sh > gpurun_out/r06_trace_init.txt 2>&1
bash tools/pmc_fused.sh 10 32 > gpurun_out/r06_pmc_fused.txt 2>&1
tail -5 gpurun_out/r06_profile.log
