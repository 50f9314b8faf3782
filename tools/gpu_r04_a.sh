#!/bin/bash
# round 4, first GPU trip: new tests, RCCL all-gather at world size 1, host overhead, LUT-vs-affine counters + staging ablation
mkdir -p gpurun_out/r04a; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04a
cd $R
timeout 900 python -m pytest tests/test_accelerate.py tests/test_holder_fast_call.py -m gpu -x -q > $O/pytest_new.log 2>&1; echo "new tests rc=$?" >> $O/pytest_new.log
tail -15 $O/pytest_new.log
timeout 300 python tools/host_overhead.py > $O/host_overhead.log 2>&1; cat $O/host_overhead.log
PORT=$(python -c "import socket; s=socket.socket(); s.bind(('127.0.0.1',0)); print(s.getsockname()[1])")
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus 1 --gather --steps 20 --warmup 5 --no-cpu --prewarm-seconds 0.3 --evidence-launches 0 > $O/bench_torchrun_gather.json 2> $O/bench_torchrun_gather.err; echo "gather rc=$?"
python - <<'PY'
import json,os
p=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out/r04a/bench_torchrun_gather.json")
for ln in open(p):
    if ln.startswith("{"):
        d=json.loads(ln); print(json.dumps(d.get("sharded_cfg5"),indent=1)); print("ranks_seen",d.get("ranks_seen"),"control",d["config"]["control_plane"])
PY
timeout 200 python tools/lut_vs_affine.py 100 > $O/lut_vs_affine.log 2>&1; cat $O/lut_vs_affine.log
MCTQ_HIP_LIB=$R/tools/ablate/libmctq_hip_STAGE.so MCTQ_BINDING=ctypes timeout 200 python tools/lut_vs_affine.py 100 > $O/lut_vs_affine_no_staging.log 2>&1; cat $O/lut_vs_affine_no_staging.log
MCTQ_BINDING=ctypes timeout 200 python tools/lut_vs_affine.py 100 > $O/lut_vs_affine_ctypes.log 2>&1; cat $O/lut_vs_affine_ctypes.log
bash tools/gpu_r04_lut_pmc.sh
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "all gpu tests rc=$?" >> $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
