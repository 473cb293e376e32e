#!/bin/bash
# end-to-end: a dataset rendered WITH rolling shutter (readout 1/30 s) and motion blur; train without compensation,
# with 8 row bands (both motion models) and with the exact per-row mode (pixel-velocity model); sharp-frame PSNR / SSIM
set -u
OUT=gpurun_out/r3_run23
mkdir -p $OUT
DS=/tmp/ds_rs
COMMON="--width 160 --height 120 --frames 16 --gaussians 4000 --rolling-shutter-time 0.0333 --iterations 600 --blur-samples 5"
timeout 300 python tools/train_deblur.py --generate $DS $COMMON --rolling-shutter-mode off --out $OUT/off > $OUT/off.log 2>&1; tail -1 $OUT/off.log
timeout 300 python tools/train_deblur.py --data $DS $COMMON --rolling-shutter-mode bands --out $OUT/bands > $OUT/bands.log 2>&1; tail -1 $OUT/bands.log
timeout 300 python tools/train_deblur.py --data $DS $COMMON --rolling-shutter-mode bands --motion-model pixel_velocity --out $OUT/bands_pv > $OUT/bands_pv.log 2>&1; tail -1 $OUT/bands_pv.log
timeout 300 python tools/train_deblur.py --data $DS $COMMON --rolling-shutter-mode exact --motion-model pixel_velocity --out $OUT/exact_pv > $OUT/exact_pv.log 2>&1; tail -1 $OUT/exact_pv.log
