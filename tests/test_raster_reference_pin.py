"""Pins the rasteriser's camera model and scene geometry to pixels of the real reference renderer: the fixture
tests/golden/render_reference_measurements.json holds measurements taken on the reference's own imgs/kuka.gif and
imgs/mobile_robot.gif (generator: tests/golden/make_render_measurements.py).  The raster oracle renders the same
scenes at the gifs' 168x168 size; the GPU rasteriser is bit-exact against that oracle (tests/test_gpu_raster.py)."""
import json
import os

import numpy as np

from oracle import kuka_clib, raster_clib

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "render_reference_measurements.json")) as f:
    REF = json.load(f)
S = REF["size"]


def _transitions(line, lo=200):
    white = line[:, 0].astype(int) > lo
    return [int(i) for i in np.nonzero(white[1:] != white[:-1])[0] + 1]


def test_kuka_camera_projects_button_and_table_where_pybullet_does():
    q = kuka_clib.settled()["q"]
    state = np.concatenate([q[:7], [0.0, 0.5, 0.0]])[None]          # default button at (0.5, 0)
    img = raster_clib.render(4, state, S, S)[0].astype(int)
    yellow = (img[..., 0] > 200) & (img[..., 1] > 200) & (img[..., 2] < 80)
    ys, xs = np.nonzero(yellow)
    rx, ry = REF["kuka"]["button_cap_centroid_xy"]
    assert abs(xs.mean() - rx) < 2.0 and abs(ys.mean() - ry) < 2.5   # pixels of 168 (1 px ~ 7 mm at the button)
    assert abs(len(xs) - REF["kuka"]["button_cap_pixels"]) < 0.1 * REF["kuka"]["button_cap_pixels"]
    warm = (img[..., 0] - img[..., 2] > 35) & (img[..., 0] > 120) & (img[..., 1] > 100)
    for col, row in REF["kuka"]["table_edge_row_at_col"].items():
        assert abs(int(np.argmax(warm[:, int(col)])) - row) <= 2, (col, row)


def test_mobile_camera_checker_phase_and_wall_colours_match_pybullet():
    m = REF["mobile"]
    img = raster_clib.render(0, np.array([[1.3, 2.9, 3.6, 1.0, 0, 0]]), S, S)[0]
    got = [t for t in _transitions(img[m["row"]]) if 30 < t < 140]
    assert len(got) == len(m["row_transitions"]) and np.abs(np.array(got) - m["row_transitions"]).max() <= 2
    got = [t for t in _transitions(img[:, m["col"]]) if 30 < t < 140]
    assert len(got) == len(m["col_transitions"]) and np.abs(np.array(got) - m["col_transitions"]).max() <= 2
    px = img.astype(int)
    for name, rc in (("blue_rgb", (120, 30)), ("white_rgb", (120, 60)), ("wall_left_rgb", (84, 18)),
                     ("wall_right_rgb", (84, 150)), ("wall_top_rgb", (17, 84)), ("wall_bottom_rgb", (150, 84))):
        assert np.abs(px[rc] - np.array(m[name])).max() <= 16, (name, px[rc], m[name])
