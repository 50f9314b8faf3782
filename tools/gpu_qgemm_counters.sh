#!/bin/bash
# Counter passes for the dense int8 consumer kernel (run under gpurun from the repo root).
# usage: tools/gpu_qgemm_counters.sh <variant>   -> gpurun_out/qgemm_pmc_<variant>/
set -u
V=${1:-2588}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/qgemm_pmc_$V
mkdir -p $OUT
cat > /tmp/qg_one.py <<PY
import sys, torch
sys.path.insert(0, "$GRAFT_REPO_ROOT")
from mct_quantizers_amd.hip import native
lib = native.load(); dev = torch.device("cuda")
M = N = K = 8192
a = torch.randint(0, 256, (M, K), dtype=torch.uint8, device=dev)
w = torch.randint(-128, 128, (N, K), dtype=torch.int8, device=dev)
sc = torch.rand(N, device=dev) * 0.01; wsum = w.sum(1, dtype=torch.int32); bias = torch.randn(N, device=dev)
y = torch.empty(M, N, device=dev)
lib.mctq_set_tuning(b"ql_variant", $V)
for _ in range(30):
    assert lib.mctq_qlinear_i8(a.data_ptr(), native.CODE_U8, 3, 0.02, w.data_ptr(), sc.data_ptr(), wsum.data_ptr(), bias.data_ptr(), y.data_ptr(), M, N, K, torch.cuda.current_stream().cuda_stream) == 0
torch.cuda.synchronize()
PY
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_I8 SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d $OUT/$tag -o p -- python3 /tmp/qg_one.py > $OUT/$tag.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o p -- python3 /tmp/qg_one.py > $OUT/stats.log 2>&1
find $OUT -name "*.csv" | head -30
