"""distinct winners per 28 x 56 tile (levels 0 and 1 together, as the pair's tile payload pass sees them) of the bench stack"""
import sys
import numpy as np
sys.path.insert(0, ".")
from shinestacker_amd import _lib as L
H, W, N = 4000, 6000, int(sys.argv[1]) if len(sys.argv) > 1 else 256
dt = np.float32
per = H * W * 3 * 4
buf = L.DeviceBuffer(per * N)
L.synth_frames_device(buf.ptr, dt, H, W, 0, N, N)
st = L.Stack(H, W, in_dtype=dt, out_dtype=np.uint8, arith="separable", pair_levels=2)
st.push_frames_device(buf.ptr, N)
i0 = st.tap(L.TAP_INDEX, 0); i1 = st.tap(L.TAP_INDEX, 1)
cnt = []
for ty in range(0, H, 28):
    for tx in range(0, W, 56):
        a = np.unique(i0[ty:ty + 28, tx:tx + 56]); b = np.unique(i1[ty // 2:ty // 2 + 14, tx // 2:tx // 2 + 28])
        cnt.append(len(np.union1d(a, b)))
cnt = np.array(cnt)
print("tiles", len(cnt), "mean distinct", cnt.mean(), "median", np.median(cnt), "max", cnt.max(), "hist", np.bincount(np.minimum(cnt, 40))[:41])
# share of the tile's pixels (both levels) by winner rank, and what a "rare winner" threshold would leave to the per-quad kernels
import collections
ranks = np.zeros(16); rare_px = collections.Counter(); rare_fr = collections.Counter(); tot = 0
for ty in range(0, H, 28 * 4):          # every 4th tile row is plenty
    for tx in range(0, W, 56):
        v = np.concatenate([i0[ty:ty + 28, tx:tx + 56].ravel(), i1[ty // 2:ty // 2 + 14, tx // 2:tx // 2 + 28].ravel()])
        c = np.sort(np.bincount(v, minlength=N))[::-1]
        c = c[c > 0]
        ranks[:min(16, len(c))] += c[:16]
        tot += 1
        for thr in (8, 16, 32, 64, 128):
            rare_px[thr] += c[c < thr].sum(); rare_fr[thr] += (c < thr).sum()
print("mean pixels by winner rank:", np.round(ranks / tot, 1))
for thr in (8, 16, 32, 64, 128):
    print(f"threshold {thr}: {rare_fr[thr] / tot:.2f} winners / tile below it, holding {rare_px[thr] / tot:.1f} of 1960 pixels")
