"""CPU: the full-size verification of bench.verify on a stand-in stack.  The GPU run compares the
library's running state with the oracle fed CROPPED generator frames; that is only a proof if a crop reproduces the
full image's state away from its artificial borders (and up to the real image edges).  Here the "stack" is the oracle
itself run on the whole image, so the window / margin / alignment logic is checked without a GPU -- and a planted
error must be reported.  Reference: the state is what pyramid.py:48-55, :125-148 build per level."""
import types

import numpy as np
import pytest

import bench
from oracle import oracle as orc


class _Taps:
    TAP_INDEX, TAP_ENERGY, TAP_FUSED_LAP = 0, 1, 2


def _stand_in(so):
    class St:
        levels = so.levels

        def tap(self, kind, lv):
            return {0: so.best_idx, 1: so.best_e, 2: so.best_lap}[kind][lv]
    return St()


@pytest.mark.parametrize("arith", ["separable", "exact"])
def test_crops_reproduce_the_full_state_and_catch_a_planted_error(arith):
    H, W, N = 1048, 1240, 3
    so = orc.StreamingOracle(H, W, np.uint8, arith=arith)
    for f in range(N):
        so.push_frame(orc.synth_frame_u8(H, W, f, N))
    args = types.SimpleNamespace(height=H, width=W, dtype="u8", arith=arith)
    v = bench.verify(_Taps, _stand_in(so), args, N, 1)
    assert v["ok"] and v["crops_equal"] and len(v["crops"]) == 6, v
    names = [c["name"] for c in v["crops"]]
    assert names[:4] == ["top-left", "top-right", "bottom-left", "bottom-right"] and "super-block seam" in names
    for c in v["crops"]:
        assert sorted(c["levels"]) == ["0", "1", "2"] and all(x["equal"] and x["pixels"] > 5000 for x in c["levels"].values())
    # corner crops reach the image edges they touch: more compared pixels than an interior crop
    assert v["crops"][0]["levels"]["0"]["pixels"] > v["crops"][4]["levels"]["0"]["pixels"]
    # a single wrong arg-max at level 2, at the very last pixel of the image (bottom-right crop, on both real edges)
    so.best_idx[2][-1, -1] += 1
    v2 = bench.verify(_Taps, _stand_in(so), args, N, 1)
    assert not v2["ok"] and not v2["crops"][3]["levels"]["2"]["equal"] and v2["crops"][0]["equal"]
    so.best_idx[2][-1, -1] -= 1
    # ... and one wrong Laplacian value at level 0 next to the seam crop's centre
    y0, x0 = v["crops"][5]["origin"]
    so.best_lap[0][y0 + 256, x0 + 256, 1] += 1.0
    v3 = bench.verify(_Taps, _stand_in(so), args, N, 1)
    assert not v3["ok"] and not v3["crops"][5]["levels"]["0"]["equal"]
