#!/bin/bash
set -u
OUT=gpurun_out/r3_run25
mkdir -p $OUT
COMMON="--width 160 --height 120 --frames 16 --gaussians 4000 --iterations 600 --blur-samples 0 5"
timeout 300 python tools/train_deblur.py --generate /tmp/ds_a $COMMON --out $OUT/hip > $OUT/hip.log 2>&1; tail -1 $OUT/hip.log
GSD_TORCH_TRAIN=1 timeout 300 python tools/train_deblur.py --data /tmp/ds_a $COMMON --out $OUT/torch > $OUT/torch.log 2>&1; tail -1 $OUT/torch.log
