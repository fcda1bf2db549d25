"""Crowded mesh tiles run as two or four workgroups with a half or a quarter of each wave's lanes: whatever the
split, every pixel is traced exactly once and the image equals the exhaustive loop's and the oracle's.  The same
holds with and without the per-tile triangle lists that camera rays run down instead of walking the tree (none,
for every tile, only for tiles with at most 8 triangles), and of the lists of possible blockers that the shadow rays
run down.  A part of a tile split into four or more is one rectangle of the tile and has lists of its own (camera rays
and, if short, shadow rays): with and without them, with every part given a shadow list however long, and with a
refinement that splits nearly everything down to single pixels per wave.  From four parts on, whole waves render a
part (every pixel held by several lanes that share its lists out): with that switched off, on from eight parts, and
with the tree read from memory."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("parts,tiles,min_cost,lists", [(2, 100000, 1, {}), (4, 100000, 1, {}), (4, 7, 1, {}), (1, 0, 1, {}),
                                                        (2, 100000, 1, {"RT_HIP_TILE_LISTS": "0"}),
                                                        (4, 100000, 1, {"RT_HIP_LIST_MAX_WALK": "8"}),
                                                        (1, 0, 1, {"RT_HIP_LIST_MAX_WALK": "100000"}),
                                                        (2, 100000, 1, {"RT_HIP_SHADOW_LISTS": "0"}),
                                                        (1, 0, 1, {"RT_HIP_LIST_MAX_WALK": "24", "RT_HIP_LIST_LDS": "0"}),
                                                        (4, 100000, 1, {"RT_HIP_PART_LISTS": "0"}),
                                                        (4, 100000, 1, {"RT_HIP_PART_SHADOW_MAX": "0", "RT_HIP_SPLIT_MAX_LG": "4"}),
                                                        (4, 100000, 1, {"RT_HIP_PART_SHADOW_MAX": "100000", "RT_HIP_SPLIT_THRESHOLD": "0.2",
                                                                        "RT_HIP_LIST_MAX_WALK": "100000"}),
                                                        (2, 100000, 1, {"RT_HIP_SPLIT_THRESHOLD": "0.3", "RT_HIP_SPLIT_REFINE": "12",
                                                                        "RT_HIP_LIST_LDS": "0"}),
                                                        # whole waves on the parts of split tiles: off, from eight parts on with
                                                        # short shadow lists, and with the tree's nodes left in memory
                                                        (4, 100000, 1, {"RT_HIP_COOP": "0", "RT_HIP_SPLIT_THRESHOLD": "0.3"}),
                                                        (4, 100000, 1, {"RT_HIP_COOP": "3", "RT_HIP_COOP_SHADOW": "4", "RT_HIP_SPLIT_THRESHOLD": "0.2"}),
                                                        (4, 100000, 1, {"RT_HIP_LDS_NODES": "0", "RT_HIP_SPLIT_THRESHOLD": "0.3"})])
def test_split_mesh_tiles_change_nothing_in_the_image(parts, tiles, min_cost, lists):
    # the switches exist in the development build of the library only (lib/librt_hip_dev.so, -DRT_DEV_SWITCHES)
    env = dict(os.environ, RT_HIP_LIB="dev", RT_HIP_SPLIT_PARTS=str(parts), RT_HIP_SPLIT_TILES=str(tiles), RT_HIP_SPLIT_MIN_COST=str(min_cost), **lists)
    r = subprocess.run([sys.executable, os.path.join(HERE, "split_tiles_child.py")], env=env, capture_output=True, text=True,
                       timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    for name in ("plain", "reflective", "bands", "supersampled"):
        o = out[name]
        assert o["tree_equals_brute"] and o["tree_equals_oracle"], (name, o)
        assert o["rays"] == o["brute_rays"] == o["oracle_rays"], (name, o)
    assert out["plain_f64_equal"] and out["reflective_f64_equal"]
