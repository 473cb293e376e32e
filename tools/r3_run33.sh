#!/bin/bash
# --optimize-eval-cameras (train.py:180-183) end to end: evaluation poses carry 2 cm / 0.02 rad of noise; scored on the sharp frames
set -u
OUT=gpurun_out/r3_run33
mkdir -p $OUT
COMMON="--width 160 --height 120 --frames 16 --gaussians 4000 --iterations 600 --blur-samples 5 --pose-noise 0.02"
timeout 300 python tools/train_deblur.py --generate /tmp/ds_pn $COMMON --out $OUT/plain > $OUT/plain.log 2>&1; tail -1 $OUT/plain.log
timeout 300 python tools/train_deblur.py --data /tmp/ds_pn $COMMON --optimize-eval-cameras --out $OUT/opt > $OUT/opt.log 2>&1; tail -1 $OUT/opt.log
