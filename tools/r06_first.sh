set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
python -m pytest tests -m gpu -x -q 2>&1 | tail -30 > gpurun_out/r06a/gpu_tests.txt
python bench.py > gpurun_out/r06a/bench_default.json 2> gpurun_out/r06a/bench_default.err
python bench.py --train --steps 30 --warmup 5 > gpurun_out/r06a/bench_train_up4.json 2> gpurun_out/r06a/bench_train_up4.err
python bench.py --train --train-up 2 --steps 30 --warmup 5 > gpurun_out/r06a/bench_train_up2.json 2> gpurun_out/r06a/bench_train_up2.err
tail -5 gpurun_out/r06a/gpu_tests.txt
