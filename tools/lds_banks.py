#!/usr/bin/env python3
"""LDS bank-conflict model of the fused level kernel's access patterns (MI355X LDS rules from
/opt/skills/guides/MI355X_MICROARCH.md): searches row strides GS / NS / QS that minimise the
extra LDS cycles for a tile configuration."""
import itertools
import sys

G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
        list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32))]
G128 += [[l + 32 for l in g] for g in G128]
G32x2 = [list(range(0, 32)), list(range(32, 64))]
G16 = [list(range(i, i + 16)) for i in range(0, 64, 16)]


def cycles(addrs, width_dw, groups, nbanks):
    """addrs: dict lane -> dword address (or None). returns total LDS cycles (1 per group if clean)."""
    tot = 0
    for g in groups:
        banks = {}
        for l in g:
            a = addrs.get(l)
            if a is None:
                continue
            for k in range(width_dw):
                banks.setdefault((a + k) % nbanks, set()).add((a + k) // nbanks)
        tot += max((len(v) for v in banks.values()), default=0)
    return tot


def wave_addrs(tids, f):
    out = {}
    for lane, t in enumerate(tids):
        out[lane] = f(t)
    return out


def model(TH, TW, NT, GS, NS, QS, RU):
    GH, GW = TH + 12, TW + 12
    NH, NW = TH // 2 + 4, TW // 2 + 4
    BX = NW // 2
    total = {"reduce": 0, "lapq_sN": 0, "lapq_sG": 0, "energy": 0, "sQ_wr": 0, "sN_wr": 0}
    ideal = dict(total)
    for w0 in range(0, NT, 64):
        tids = list(range(w0, w0 + 64))
        # reduce: item it -> (rb, bx); loads: rows rr, 5 b128 at p0 + rr*GS + 4t, + b32 at +20
        nitems = (NH // RU) * BX

        def red(t, rr, tt):
            if t >= nitems:
                return None
            rb, bx = divmod(t, BX)
            return 2 * RU * rb * GS + 12 * bx + rr * GS + 4 * tt
        for tt in range(5):
            a = wave_addrs(tids, lambda t: red(t, 0, tt))
            if any(v is not None for v in a.values()):
                total["reduce"] += cycles(a, 4, G128, 64)
                ideal["reduce"] += sum(1 for g in G128 if any(a.get(l) is not None for l in g))
        # sN writes: 3 x b64 at ri*NS + rj*3 (+0,2,4), groups of 16 lanes, 32 banks

        def snw(t, u, k):
            if t >= nitems:
                return None
            rb, bx = divmod(t, BX)
            return (rb * RU + u) * NS + 6 * bx + 2 * k
        for k in range(3):
            a = wave_addrs(tids, lambda t: snw(t, 0, k))
            if any(v is not None for v in a.values()):
                total["sN_wr"] += cycles(a, 2, G16, 32)
                ideal["sN_wr"] += sum(1 for g in G16 if any(a.get(l) is not None for l in g))
        # lapq own quad: thread -> (oy, ox) = divmod(t, TW/2); qy = oy+1, qx = ox+1
        # sN b32 reads at (qy-1+ar)*NS... interior: re = qy+1 -> rows qy..qy+2, cols (qx)*3 + j
        for ar in range(3):
            for j in range(0, 9):
                def f(t):
                    oy, ox = divmod(t, TW // 2)
                    qy, qx = oy + 1, ox + 1
                    return (qy + ar) * NS + qx * 3 + j
                a = wave_addrs(tids, f)
                total["lapq_sN"] += cycles(a, 1, G32x2, 32)
                ideal["lapq_sN"] += 2
        # sG cells: rows 2qy+4 (+0,1), b64 x3 at (2qx+4)*3 + 2k
        for dr in range(2):
            for k in range(3):
                def f(t):
                    oy, ox = divmod(t, TW // 2)
                    qy, qx = oy + 1, ox + 1
                    return (2 * qy + 4 + dr) * GS + (2 * qx + 4) * 3 + 2 * k
                a = wave_addrs(tids, f)
                total["lapq_sG"] += cycles(a, 2, G32x2, 64)
                ideal["lapq_sG"] += 2
        # sQ writes b64 x2 (rows 2qy, 2qy+1 at 2qx)
        for dr in range(2):
            def f(t):
                oy, ox = divmod(t, TW // 2)
                return (2 * (oy + 1) + dr) * QS + 2 * (ox + 1)
            a = wave_addrs(tids, f)
            total["sQ_wr"] += cycles(a, 2, G16, 32)
            ideal["sQ_wr"] += 4
        # energy: 6 rows x 3 b64 at (2oy+rr)*QS + 2ox + 2k
        for rr in range(6):
            for k in range(3):
                def f(t):
                    oy, ox = divmod(t, TW // 2)
                    return (2 * oy + rr) * QS + 2 * ox + 2 * k
                a = wave_addrs(tids, f)
                total["energy"] += cycles(a, 2, G32x2, 64)
                ideal["energy"] += 2
    return total, ideal


if __name__ == "__main__":
    TH, TW, NT = (int(x) for x in (sys.argv[1:4] or (32, 64, 512)))
    GD, ND, QW = (TW + 12) * 3, (TW // 2 + 4) * 3, TW + 4
    best = None
    for RU in (1, 2):
        for GS in range(GD, GD + 68, 4):
            t, i = model(TH, TW, NT, GS, ND, QW, RU)
            key = t["reduce"] + t["lapq_sG"]
            if best is None or key < best[0]:
                best = (key, RU, GS)
            print(f"RU={RU} GS={GS}: reduce {t['reduce']}/{i['reduce']}  lapq_sG {t['lapq_sG']}/{i['lapq_sG']}")
    for NS in range(ND, ND + 40, 2):
        t, i = model(TH, TW, NT, GD, NS, QW, 1)
        print(f"NS={NS}: lapq_sN {t['lapq_sN']}/{i['lapq_sN']}  sN_wr {t['sN_wr']}/{i['sN_wr']}")
    for QS in range(QW, QW + 40, 2):
        t, i = model(TH, TW, NT, GD, ND, QS, 1)
        print(f"QS={QS}: energy {t['energy']}/{i['energy']}  sQ_wr {t['sQ_wr']}/{i['sQ_wr']}")


def ring_model(TH, TW, GS, NS, QS):
    """Second lapq pass: ring quads; lane it -> (qy, qx) as the kernel enumerates them."""
    QY, QX = TH // 2 + 2, TW // 2 + 2
    RING = QY * QX - (TH // 2) * (TW // 2)

    def coord(it):
        if it >= RING:
            return None
        if it < QX:
            return 0, it
        if it < 2 * QX:
            return QY - 1, it - QX
        s = it - 2 * QX
        return 1 + (s >> 1), (QX - 1 if s & 1 else 0)
    tot = {"sN": 0, "sG": 0, "sQw": 0}
    for w0 in range(0, RING, 64):
        tids = list(range(w0, w0 + 64))
        for ar in range(3):
            for j in range(9):
                def f(t):
                    c = coord(t)
                    return None if c is None else (c[0] + ar) * NS + c[1] * 3 + j
                tot["sN"] += cycles(wave_addrs(tids, f), 1, G32x2, 32)
        for dr in range(2):
            for k in range(3):
                def f(t):
                    c = coord(t)
                    return None if c is None else (2 * c[0] + 4 + dr) * GS + (2 * c[1] + 4) * 3 + 2 * k
                tot["sG"] += cycles(wave_addrs(tids, f), 2, G32x2, 64)
            def f(t):
                c = coord(t)
                return None if c is None else (2 * c[0] + dr) * QS + 2 * c[1]
            tot["sQw"] += cycles(wave_addrs(tids, f), 2, G16, 32)
    return tot


if __name__ == "__main__" and len(sys.argv) > 4 and sys.argv[4] == "ring":
    TH, TW = int(sys.argv[1]), int(sys.argv[2])
    GD, ND, QW = (TW + 12) * 3, (TW // 2 + 4) * 3, TW + 4
    print("ideal per full wave: sN 54, sG 12, sQw 8")
    for NS in range(ND, ND + 34, 2):
        print("NS", NS, ring_model(TH, TW, GD, NS, QW)["sN"])
    for QS in range(QW, QW + 34, 2):
        print("QS", QS, ring_model(TH, TW, GD, ND, QS)["sQw"])
    for GS in range(GD, GD + 68, 4):
        print("GS", GS, ring_model(TH, TW, GS, ND, QW)["sG"])
