"""Generate tests/golden/head_frame.npz: the frame the reference writes at HEAD (main.cpp:1212-1288:
the `shells` scene, 1024x768, 1 spp), rendered by the CPU oracle.

The P3 file the oracle writes for this frame must hash to the sha256 recorded from the unmodified
reference binary (tests/golden/reference_anchors.json, "head_frame_sha256"); the script refuses to
write the fixture otherwise.  Stored: the 8-bit frame (what Image::writeImage prints) and, so that the
GPU test can also compare radiance, the linear float32 frame on a lattice (every 4th pixel of every
4th row).

usage: python tests/golden/make_head_frame.py        (about 30 s on 8 cores)
"""
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
import oracle_lib as O  # noqa: E402

W, H = 1024, 768


def main():
    want = json.load(open(os.path.join(HERE, "reference_anchors.json")))["head_frame_sha256"]
    rgb, rgb8, rays = O.Scene("shells").render(W, H, 90.0, mode=0, threads=os.cpu_count())
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "head.ppm")
        O.write_ppm(path, rgb8)
        sha = hashlib.sha256(open(path, "rb").read()).hexdigest()
    if sha != want:
        print(f"MISMATCH: oracle frame hashes to {sha}, the reference's to {want}")
        return 1
    np.savez_compressed(os.path.join(HERE, "head_frame.npz"), rgb8=rgb8, rgb_lattice=rgb[::4, ::4].copy(),
                        rays=np.int64(rays), sha256=np.bytes_(sha.encode()))
    print(f"head_frame.npz written: {rays} rays, P3 sha256 {sha} == the reference's")
    return 0


if __name__ == "__main__":
    sys.exit(main())
