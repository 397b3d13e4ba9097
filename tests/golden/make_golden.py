"""Generates the golden fixtures tests/golden/<preset>.npz from the CPU oracle.

The oracle is itself pinned to the reference (tests/test_oracle_anchors.py: the sha256 of the whole
frame the unmodified reference writes, plus recorded per-pixel values); main.cpp of the reference
cannot be built in this image, so these vectors come from the oracle, not from the reference binary.

    python tests/golden/make_golden.py

Each file holds a ray set (camera rays of a 64x48 frame, rays leaving the hit surfaces, random rays,
rays aimed at mesh vertices and edges), the oracle's hit records and radiance for it, and two small
frames (reference 1-spp and 9-tap sampling) as linear radiance and quantised 8-bit images.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402
import parity  # noqa: E402
import ray_tracer_amd as rt  # noqa: E402

PRESETS = ["analytic", "teapot", "teapot_head", "shells", "textured"]
MAX_DEPTH = 8


def main():
    for preset in PRESETS:
        osc = O.Scene(preset)
        hs = rt.HostScene(preset)  # must outlive the description it hands out
        desc = hs.flatten().contents
        o, d = parity.ray_set(osc, desc, w=64, h=48, seed=11, n_random=800)
        hits = osc.find_closest_hit(o, d)
        rad, rays = osc.trace_rays(o, d, MAX_DEPTH)
        f1, f1q, _ = osc.render(64, 48, mode=0, max_depth=MAX_DEPTH)
        f9, f9q, _ = osc.render(64, 48, mode=1, max_depth=MAX_DEPTH)
        out = os.path.join(HERE, preset + ".npz")
        np.savez_compressed(out, o=o, d=d, hits=hits, radiance=rad, rays=rays.astype(np.uint32),
                            frame_1spp=f1, frame_1spp_rgb8=f1q, frame_aa9=f9, frame_aa9_rgb8=f9q,
                            max_depth=np.int32(MAX_DEPTH))
        print(preset, len(o), "rays", os.path.getsize(out) // 1024, "KiB")


if __name__ == "__main__":
    main()
