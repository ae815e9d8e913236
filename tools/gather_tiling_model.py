#!/usr/bin/env python3
"""Developer tool (CPU): slot efficiency (pairs inside the disc / evaluated slots) of candidate lane tilings of the backward
gather's window sweep (raster_backward.hip, render_backward_kernel, TPW = 4: 16 lanes per task), for a search radius of
rs pixels and uniformly random sub-pixel positions.  Round-3 result: at the 8.4 pixels of BASELINE configs 4-5 the current
16 lanes x 2 column slots evaluate 2.4 slots per pair; a flattened (row, column) slot index over the disc's chords ~1.0.
    python tools/gather_tiling_model.py [rs_px ...]"""
import sys

import numpy as np


def model(rs_px, n=4000, seed=0):
    rng = np.random.default_rng(seed)
    pairs = cur = t83 = sq = chord = 0
    for _ in range(n):
        cx, cy = rng.uniform(0, 1, 2)
        xs, ys = np.arange(-48, 49) - cx, np.arange(-48, 49) - cy
        ow, oh = int((np.abs(xs) <= rs_px).sum()), int((np.abs(ys) <= rs_px).sum())
        p = int(((xs[None, :] ** 2 + ys[:, None] ** 2) <= rs_px ** 2).sum())
        pairs += p
        cur += -(-ow // 32) * 32 * oh      # 16 lanes x 2 column slots per pass, one row per lane row (today)
        t83 += -(-ow // 24) * 24 * oh      # 8 lanes x 3 column slots
        sq += -(-(ow * oh) // 16) * 16     # flattened bounding square over 16 lanes
        chord += -(-p // 16) * 16          # flattened chords (only the disc) over 16 lanes
    return {"rs_px": rs_px, "pairs_per_point": round(pairs / n, 1), "today_16x2": round(pairs / cur, 3),
            "8x3": round(pairs / t83, 3), "flattened_square": round(pairs / sq, 3), "flattened_chords": round(pairs / chord, 3)}


if __name__ == "__main__":
    for r in [float(x) for x in sys.argv[1:]] or [6.0, 7.6, 8.4, 10.0, 12.8]:
        print(model(r))
