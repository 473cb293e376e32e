#!/bin/bash
set -u
OUT=gpurun_out/r3_run24
mkdir -p $OUT
COMMON="--width 160 --height 120 --frames 17 --gaussians 4000 --speed 1.5 --iterations 700 --blur-samples 0 5"
timeout 300 python tools/train_deblur.py --generate /tmp/ds_nors $COMMON --out $OUT/nors > $OUT/nors.log 2>&1; tail -1 $OUT/nors.log
timeout 300 python tools/train_deblur.py --generate /tmp/ds_rs $COMMON --rolling-shutter-time 0.0333 --rolling-shutter-mode bands --out $OUT/rs > $OUT/rs.log 2>&1; tail -1 $OUT/rs.log
python - <<'PY'
import sys
sys.path.insert(0, '.')
import torch, gsdeblur_amd as gs
for d in ('/tmp/ds_nors', '/tmp/ds_rs'):
    sc = gs.load_transforms(d)
    im = gs.data.load_scene_images(sc, torch.device('cuda', 0))
    print(d, 'frames', len(im), 'mean', [round(float(x.mean()), 4) for x in im[:4]], 'exposure', sc.exposure_time, 'rs', sc.rolling_shutter_time)
a = gs.data.load_scene_images(gs.load_transforms('/tmp/ds_nors'), torch.device('cuda', 0))
b = gs.data.load_scene_images(gs.load_transforms('/tmp/ds_rs'), torch.device('cuda', 0))
for i in (0, 1, 2, 8):
    print('frame', i, 'mean abs diff rs vs no-rs', round(float((a[i] - b[i]).abs().mean()), 5))
PY
