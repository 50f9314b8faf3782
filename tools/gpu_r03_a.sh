#!/bin/bash
# round 3, first GPU session: the whole GPU suite, the judged bench line, the batched line, the torchrun leg
mkdir -p gpurun_out/r03
python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/r03/pytest_gpu.log; cat gpurun_out/r03/pytest_gpu.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r03/bench_20.json 2> gpurun_out/r03/bench_20.err; tail -c 1500 gpurun_out/r03/bench_20.json
python bench.py > gpurun_out/r03/bench_default.json 2>> gpurun_out/r03/bench_20.err
python bench.py --batched 16 --steps 100 --warmup 10 > gpurun_out/r03/bench_batched16.json 2>> gpurun_out/r03/bench_20.err; cut -c1-1800 gpurun_out/r03/bench_batched16.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --gather --steps 50 --warmup 5 --no-cpu > gpurun_out/r03/bench_torchrun_gather.json 2> gpurun_out/r03/torchrun.err; cut -c1-300 gpurun_out/r03/bench_torchrun_gather.json; tail -3 gpurun_out/r03/torchrun.err
python tools/requant_model_weights.py > gpurun_out/r03/requant_model_weights_batched.log 2>&1; cat gpurun_out/r03/requant_model_weights_batched.log
