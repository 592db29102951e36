"""Build-container tool: extract the colour evidence the reference holds for `cv2.COLORMAP_JET`.

`/root/reference/reference/{1..4}.png` were written by `/root/reference/inference.py:114-120`
(`cv2.applyColorMap(uint8 disparity, cv2.COLORMAP_JET)` -> `cv2.imwrite`).  The disparities come from weights that were
never released, so the images are not numeric goldens for the network -- but every pixel IS one entry of OpenCV's
256-entry JET table, which makes them the only reference-held golden for the byte work of the output pipeline.

Writes `tests/golden/jet_reference_colours.npz` (data only):
  colours   [n,3] uint8  distinct RGB triples over the four images, lexicographic order (no ordering knowledge encoded)
  counts    [n]   int64  pixels of each colour over the four images
  patch     [4,16,64,3] uint8  one 16x64 crop per image (rows 176..191, columns 600..663): raw bytes for a
                               decode -> re-encode round trip
Run in the build container (the reference tree is not on the GPU box):  python tools/make_jet_fixture.py
"""
import os
import sys

import numpy as np
from PIL import Image

sys.dont_write_bytecode = True
REF = "/root/reference/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                   "jet_reference_colours.npz")


def main():
    tally = {}
    patches = []
    for i in range(1, 5):
        im = np.asarray(Image.open(os.path.join(REF, f"{i}.png")).convert("RGB"), dtype=np.uint8)
        assert im.shape == (368, 1232, 3), im.shape                      # inference.py:94 crop
        cols, cnt = np.unique(im.reshape(-1, 3), axis=0, return_counts=True)
        for c, n in zip(map(tuple, cols.tolist()), cnt.tolist()):
            tally[c] = tally.get(c, 0) + n
        patches.append(im[176:192, 600:664].copy())
    colours = np.array(sorted(tally), dtype=np.uint8)
    counts = np.array([tally[tuple(c)] for c in colours.tolist()], dtype=np.int64)
    np.savez_compressed(OUT, colours=colours, counts=counts, patch=np.stack(patches))
    print(f"{len(colours)} distinct colours over {counts.sum()} pixels -> {OUT}")


if __name__ == "__main__":
    main()
