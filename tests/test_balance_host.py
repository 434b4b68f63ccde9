"""CPU: the host half of BalanceFrames (shinestacker_amd/balance.py) -- corrections and look-up
tables -- against tests/golden/balance.npz, recorded from the reference's own correction classes
(balance.py:14-120) by oracle/gen_golden.py; and the oracle's NumPy restatement of the two device
steps (histogram, table apply) against the same recording."""
import json

import numpy as np
import pytest

from conftest import load_golden


@pytest.fixture(scope="module")
def gold():
    g = load_golden("balance")
    return g, json.loads(str(g["meta"]))


def _maps():
    from shinestacker_amd import balance as b
    return {"LINEAR": b.LinearMap, "GAMMA": b.GammaMap, "MATCH_HIST": b.MatchHist}


def test_tables_and_corrections_equal_the_reference(gold):
    g, meta = gold
    assert len(meta) >= 20
    for m in meta:
        t = m["tag"]
        dtype = np.dtype(m["dtype"])
        href = [h.astype(np.int64) for h in g[f"{t}_hist_ref"]]
        hmov = [h.astype(np.int64) for h in g[f"{t}_hist_mov"]]
        cm = _maps()[m["corr_map"]](dtype, href, m["opts"].get("intensity_interval"))
        corr = cm.correction(hmov)
        luts = np.stack([cm.table(corr[c], cm.reference[c]) for c in range(len(href))])
        assert luts.dtype == dtype
        assert np.array_equal(luts, g[f"{t}_luts"]), (m, int((luts != g[f"{t}_luts"]).sum()))
        size = np.asarray(cm.correction_size(corr), dtype=np.float64).ravel()
        assert np.array_equal(size, np.asarray(g[f"{t}_size"]).ravel()), m


def test_oracle_device_steps_equal_the_reference(gold, oracle):
    g, meta = gold
    for m in meta:
        t = m["tag"]
        mov = g["mov_" + m["dtype"]]
        o = m["opts"]
        kw = dict(subsample=o.get("subsample", 1), fast=o.get("fast_subsampling", False), mask_size=o.get("mask_size", 0))
        if m["channel"] in ("HSV", "HLS"):
            # balance.py:340-363: to the hue-based space, balance channels 1 and 2, back to BGR
            to, back = ((oracle.CVT_BGR2HSV, oracle.CVT_HSV2BGR) if m["channel"] == "HSV"
                        else (oracle.CVT_BGR2HLS, oracle.CVT_HLS2BGR))
            pre = oracle.cvt_color_u8(mov, to)
            assert np.array_equal(pre, g[f"{t}_pre"])
            assert np.array_equal(oracle.balance_hist(pre, False, **kw)[1:], g[f"{t}_hist_mov"])
            luts = np.concatenate([np.arange(256, dtype=np.uint8)[None], g[f"{t}_luts"]])
            assert np.array_equal(oracle.cvt_color_u8(oracle.apply_lut(pre, luts), back), g[f"{t}_out"])
            continue
        h = oracle.balance_hist(mov, m["channel"] == "LUMI", **kw)
        assert np.array_equal(h, g[f"{t}_hist_mov"])
        assert np.array_equal(oracle.apply_lut(mov, g[f"{t}_luts"]), g[f"{t}_out"])


def test_constructor_options():
    from shinestacker_amd.balance import BalanceFrames, LSCorrection, LumiCorrection, RGBCorrection, SVCorrection
    from shinestacker_amd.errors import InvalidOptionError
    assert isinstance(BalanceFrames().correction, LumiCorrection)
    b = BalanceFrames(channel="RGB", corr_map="MATCH_HIST", subsample=-1)
    assert isinstance(b.correction, RGBCorrection) and b.correction.subsample == 1   # balance.py:378-380
    assert BalanceFrames(subsample=-1).correction.subsample == 8
    assert isinstance(BalanceFrames(channel="HSV").correction, SVCorrection)     # balance.py:385-388
    assert isinstance(BalanceFrames(channel="HLS").correction, LSCorrection)
    with pytest.raises(InvalidOptionError):
        BalanceFrames(channel="XYZ")
    with pytest.raises(InvalidOptionError):      # cv2.cvtColor has no 16-bit HSV: the reference raises there as well
        SVCorrection()._need_u8(np.uint16)
    c = LumiCorrection(corr_map="NOPE")
    with pytest.raises((InvalidOptionError, Exception)):
        c.begin(np.zeros((8, 8, 3), np.uint8), 2, 0)
