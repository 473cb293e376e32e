#!/bin/bash
set -u
OUT=gpurun_out/r3_run6
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "native_frame" > $OUT/pytest_native.log 2>&1
tail -30 $OUT/pytest_native.log
