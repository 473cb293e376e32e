#!/bin/bash
set -u
OUT=gpurun_out/r3_run26
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -s -k "pixel_velocity or real_camera_pose or exact_rolling or pixvel" 2>&1 | grep -E "real pose|passed|failed|Error|assert" | tail -12
timeout 300 python tools/rs_forward_check.py 0.0667 1.5 120 160 0.0 2>&1 | tail -5
timeout 600 python tools/rs_e2e.py 2>&1 | tail -1
