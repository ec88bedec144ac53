"""Variant rows per 16x16 tile that e4s_conv_region_bf16x3_f32 (csrc/conv_region.hip) needs on the bench's face-like label maps:
for every output pixel m of a tile and every 3x3 tap t, the halo pixel m + t needs a row scaled with region(m) whenever
region(m) != region(m + t) (its own row carries its own region's style); rows are counted per distinct (halo pixel, region) pair.
CPU only (numpy); prints the distribution per map size -- the numbers quoted in DESIGN.md 3.9 -- and the share of tiles that would
overflow the kernel's 256 rows and fall back to the region-select kernel.

  python tools/variant_rows_stats.py [batch] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e4s_amd import synth  # noqa: E402

VMAX, T = 256, 16


def nearest_down(lab, h):
    """F.interpolate(mode='nearest') as the kernels evaluate it: src = min(floor(dst * Hm / H), Hm - 1)."""
    hm = lab.shape[-1]
    idx = np.minimum(np.floor(np.arange(h) * (hm / h)).astype(int), hm - 1)
    return lab[:, idx][:, :, idx]


def variant_rows(lab):
    """lab [B,H,W] int -> variant rows of every tile."""
    b, h, w = lab.shape
    pad = np.full((b, h + 2, w + 2), -1)
    pad[:, 1:-1, 1:-1] = lab
    counts = []
    for n in range(b):
        for ty in range(0, h, T):
            for tx in range(0, w, T):
                need = set()
                for my in range(min(T, h - ty)):
                    for mx in range(min(T, w - tx)):
                        r = lab[n, ty + my, tx + mx]
                        win = pad[n, ty + my:ty + my + 3, tx + mx:tx + mx + 3]
                        for dy, dx in zip(*np.nonzero((win != -1) & (win != r))):
                            need.add((ty + my + dy, tx + mx + dx, r))
                counts.append(len(need))
    return np.array(counts)


if __name__ == "__main__":
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    lab512 = synth.synth_labels_face(batch, 512, seed=seed).numpy()[:, 0]
    for h in (16, 32, 64, 128, 256):
        c = variant_rows(nearest_down(lab512, h))
        print("%4d^2: tiles %5d  none %5.1f %%  mean %6.1f  p50 %5.0f  p90 %5.0f  max %4d  overflow (> %d) %5.1f %%"
              % (h, len(c), 100 * (c == 0).mean(), c.mean(), np.percentile(c, 50), np.percentile(c, 90), c.max(), VMAX,
                 100 * (c > VMAX).mean()))
