"""debug: which tap of the pair path differs from the oracle, and where"""
import sys
import numpy as np
sys.path.insert(0, ".")
from shinestacker_amd import _lib as L
from oracle import oracle as orc
orc.build()

def run(h, w, n, dt, min_size, a, batch, pl=1):
    rng = np.random.default_rng(h * 7 + w)
    hi = 65536 if dt == np.uint16 else 256
    frames = [rng.integers(0, hi, (h, w, 3)).astype(dt) for _ in range(n)]
    so = orc.StreamingOracle(h, w, dt if dt != np.float32 else np.uint8, arith="separable", min_size=min_size, gen_kernel=a)
    gs = [so.push_frame(f) for f in frames]
    st = L.Stack(h, w, in_dtype=dt, out_dtype=np.uint16 if dt == np.uint16 else np.uint8, impl=2, arith="separable",
                 min_size=min_size, gen_kernel=a, batch_frames=batch, pair_levels=pl)
    for f in frames:
        st.push_frame(f)
    print(f"case {h}x{w} n={n} {np.dtype(dt).name} levels={st.levels} shapes={st.shapes}")
    def rep(name, got, want):
        bad = got != want
        if bad.ndim == 3:
            bad = bad.any(axis=2)
        if bad.any():
            ys, xs = np.nonzero(bad)
            print(f"  {name}: {bad.sum()} of {bad.size} differ; rows {ys.min()}..{ys.max()} cols {xs.min()}..{xs.max()}; first {ys[0]},{xs[0]} got {got[ys[0], xs[0]]} want {want[ys[0], xs[0]]}")
            if bad.shape[0] < 80 and bad.shape[1] < 140:
                for r in range(bad.shape[0]):
                    print("   ", "".join("X" if v else "." for v in bad[r]))
        else:
            print(f"  {name}: ok")
    for lv in range(1, st.levels + 1):
        rep(f"gauss {lv}", st.tap(L.TAP_GAUSS, lv), gs[-1][lv])
    for lv in range(st.levels):
        rep(f"energy {lv}", st.tap(L.TAP_ENERGY, lv), so.best_e[lv])
        rep(f"index {lv}", st.tap(L.TAP_INDEX, lv), so.best_idx[lv])
        rep(f"lap {lv}", st.tap(L.TAP_FUSED_LAP, lv), so.best_lap[lv])
    st.close()

if __name__ == "__main__":
    cases = [(133, 201, 4, np.uint8, 8, 0.4, 0), (112, 224, 4, np.uint8, 8, 0.4, 0), (300, 452, 5, np.uint8, 32, 0.4, 2),
             (96, 64, 3, np.float32, 8, 0.4, 0)]
    for c in cases:
        run(*c)
