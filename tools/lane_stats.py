#!/usr/bin/env python
"""Lane utilisation of the compositor's walk ("make the SIMD waste a number").

GSD_LANE_STATS=1 python tools/lane_stats.py [--scene survey|trained|both]
One forward of bench.py's workload through gs_rasterize_fwd_slice_stats; prints the raw counters and
  * pixel utilisation            = blended pixels / (256 * entries walked)
  * 4x4 / 8x8 block utilisation  = blocks with a hit / (16 | 4) per entry
  * lock-step speed-up estimates = 64 / (mean steps per 64-entry chunk when every block walks only its own entries)
"""
import argparse
import json
import os
import sys
from pathlib import Path

os.environ["GSD_LANE_STATS"] = "1"
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402


def run(profile, N, W, H, S):
    import gsdeblur_amd as gs
    from gsdeblur_amd import ops
    sys.path.insert(0, str(ROOT / "tests"))
    import python_frame_path            # the counters are taken by the Python orchestration twin (GSD_LANE_STATS)
    python_frame_path.install()
    dev = torch.device("cuda", 0)
    wl = bench.Workload(gs, dev, 0, 1, N, W, H, S, 1, profile, "sparse")
    ops.lane_stats = None
    wl.step()
    torch.cuda.synchronize()
    c = [int(v) for v in ops.lane_stats.tolist()]
    ent, chunks = max(1, c[0]), max(1, c[7])
    out = {"scene": profile, "counters": c, "entries_walked": c[0], "slices": list(ops.last_slice_intersects),
           "pixel_utilisation_live": round(c[1] / (256.0 * ent), 4),
           "pixel_utilisation_geometric": round(c[2] / (256.0 * ent), 4),
           "blocks4x4_per_entry_geometric": round(c[3] / ent, 3), "quads8x8_per_entry_geometric": round(c[4] / ent, 3),
           "blocks4x4_per_entry_live": round(c[9] / ent, 3), "quads8x8_per_entry_live": round(c[10] / ent, 3),
           "entries_with_live_hit": round(c[8] / ent, 4),
           "entries_per_chunk": round(ent / chunks, 2),
           "lockstep_steps_per_chunk_4x4": round(c[5] / chunks, 2), "lockstep_steps_per_chunk_8x8": round(c[6] / chunks, 2),
           "lockstep_speedup_4x4": round(ent / max(1, c[5]), 3), "lockstep_speedup_8x8": round(ent / max(1, c[6]), 3),
           "lockstep_speedup_4x4_live": round(ent / max(1, c[11]), 3),
           "lockstep_speedup_8x8_live": round(ent / max(1, c[12]), 3)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="both")
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--subposes", type=int, default=5)
    a = ap.parse_args()
    for prof in (["survey", "trained"] if a.scene == "both" else [a.scene]):
        run(prof, a.gaussians, a.width, a.height, a.subposes)
