#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_consumers.py -m gpu -q -x > gpurun_out/pytest_consumers.log 2>&1; echo "rc=$?" >> gpurun_out/pytest_consumers.log
timeout 900 python tools/qlinear_probe.py > gpurun_out/qlinear_probe.log 2>&1; echo "rc=$?" >> gpurun_out/qlinear_probe.log
tail -4 gpurun_out/pytest_consumers.log; grep -v "w4a8" gpurun_out/qlinear_probe.log | head -120
