"""Render the scene the reference builds at HEAD (1024x768, 1 spp) with the CPU oracle, write it
with the oracle's P3 writer and compare the file's sha256 with the hash recorded from the
unmodified reference binary.  Takes minutes (brute force over ~30k triangles per gated ray).

usage: python oracle/pin_head_frame.py [out.ppm]
"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import oracle_lib as O  # noqa: E402

REFERENCE_SHA256 = "0adcaa01f793f186f4ec0923c54004c94d9017c2d56b3fdd86e4457891a10baf"


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/oracle_head.ppm"
    t = time.time()
    scene = O.Scene("shells")
    rgb, rgb8, rays = scene.render(1024, 768, 90.0, mode=0, threads=os.cpu_count())
    O.write_ppm(out, rgb8)
    sha = hashlib.sha256(open(out, "rb").read()).hexdigest()
    print(f"rendered in {time.time() - t:.1f}s, rays={rays}, sha256={sha}")
    print("MATCH" if sha == REFERENCE_SHA256 else "MISMATCH vs " + REFERENCE_SHA256)
    return 0 if sha == REFERENCE_SHA256 else 1


if __name__ == "__main__":
    sys.exit(main())
