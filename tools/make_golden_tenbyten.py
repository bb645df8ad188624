#!/usr/bin/env python3
"""Generates tests/golden/ten_by_ten.json from the reference's own test data file
costmap_2d/test/TenByTen.pgm + TenByTen.yaml (run in the build container only; the reference tree
does not exist on the GPU box).  The pixel -> occupancy arithmetic is map_server's
(map_server/src/image_loader.cpp:110-155, trinary mode): occ = (255 - p)/255; occ > occupied_thresh
-> 100, occ < free_thresh -> 0, else -1; image rows are flipped so cell (0,0) is bottom-left."""
import json, os, sys

REF = os.environ.get("NAV_REFERENCE", "/root/reference")
d = open(os.path.join(REF, "costmap_2d/test/TenByTen.pgm"), "rb").read()
i, toks = 0, []
while len(toks) < 4:
    while d[i:i + 1].isspace():
        i += 1
    if d[i:i + 1] == b"#":
        while d[i:i + 1] != b"\n":
            i += 1
        continue
    j = i
    while not d[j:j + 1].isspace():
        j += 1
    toks.append(d[i:j]); i = j
i += 1
assert toks[0] == b"P5"
w, h, mx = int(toks[1]), int(toks[2]), int(toks[3])
pix = d[i:i + w * h]
occ_th, free_th = 0.65, 0.196   # TenByTen.yaml
occ = [[0] * w for _ in range(h)]
for r in range(h):
    for c in range(w):
        o = (255 - pix[r * w + c]) / 255.0
        v = 100 if o > occ_th else (0 if o < free_th else -1)
        occ[h - r - 1][c] = v
out = {"source": "costmap_2d/test/TenByTen.pgm + TenByTen.yaml via map_server/src/image_loader.cpp:110-155",
       "resolution": 1.0, "origin": [0.0, 0.0], "width": w, "height": h, "occupancy_rows_y0_first": occ}
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "ten_by_ten.json")
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst)
for row in occ: print(row)
