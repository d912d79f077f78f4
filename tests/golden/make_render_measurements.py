"""Measure camera/scene facts on the reference's own rendered frames (imgs/kuka.gif, imgs/mobile_robot.gif, both
168x168 down-scales of the 224x224 TinyRenderer output) and store them as a small JSON fixture: the only pixels of
the real PyBullet renderer available without pybullet.  Run in the build container (needs /root/reference):
    python tests/golden/make_render_measurements.py
The test that consumes the fixture (tests/test_raster_reference_pin.py) renders the same scenes with the raster
oracle and checks the projected geometry (button centroid, table silhouette, checker phase/period, wall colours)."""
import json
import os

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/imgs"


def frame0(name):
    im = Image.open(os.path.join(REF, name))
    im.seek(0)
    return np.array(im.convert("RGB")).astype(int)


def yellow(a):
    return (a[..., 0] > 200) & (a[..., 1] > 200) & (a[..., 2] < 80)


def warm(a):
    return (a[..., 0] - a[..., 2] > 35) & (a[..., 0] > 120) & (a[..., 1] > 100)


def transitions(line, lo=200):
    """indices where a scan line switches between the white (>lo) and the blue checker squares"""
    white = line[:, 0] > lo
    return [int(i) for i in np.nonzero(white[1:] != white[:-1])[0] + 1]


def main():
    k = frame0("kuka.gif")
    ys, xs = np.nonzero(yellow(k))
    m = frame0("mobile_robot.gif")
    out = {
        "size": int(k.shape[0]),
        "kuka": {
            "button_cap_centroid_xy": [float(xs.mean()), float(ys.mean())],
            "button_cap_pixels": int(len(xs)),
            "table_edge_row_at_col": {str(c): int(np.argmax(warm(k[:, c]))) for c in (10, 40, 80, 120)},
        },
        "mobile": {
            "row": 120, "row_transitions": [t for t in transitions(m[120]) if 30 < t < 140],
            "col": 100, "col_transitions": [t for t in transitions(m[:, 100]) if 30 < t < 140],
            "blue_rgb": [int(v) for v in m[120, 30]], "white_rgb": [int(v) for v in m[120, 60]],
            "wall_left_rgb": [int(v) for v in m[84, 18]], "wall_right_rgb": [int(v) for v in m[84, 150]],
            "wall_top_rgb": [int(v) for v in m[17, 84]], "wall_bottom_rgb": [int(v) for v in m[150, 84]],
        },
    }
    with open(os.path.join(HERE, "render_reference_measurements.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
