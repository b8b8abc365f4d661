"""Where the recovery of the reference's risk index stands (run here, next to
the reference checkout; not part of the package or the tests).

    python scratch/cri_explore.py [/root/reference]

Static obstacles: `oracle/ref_cri.py` reproduces every record.  Moving
obstacles: prints what is left after the static rule is applied, split the way
that was most telling so far:
  * vessel at rest, obstacle opening, beyond the range term: risk =
    0.2 u_B(bearing) - delta, delta a smooth function of the vessel's angle off
    the target's stern (about 0.13 at 36 deg, 0.034 at 66 deg), the sum clipped
    at 0;
  * obstacles faster than the vessel gain up to +0.2 (a speed-ratio term).
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_cri  # noqa: E402


def load(ref):
    meta = [l.split() for l in open(os.path.join(ref, "data", "metadata_MovingObstaclesNoRules.csv"))]
    risk = [l.split() for l in open(os.path.join(ref, "data", "risk_MovingObstaclesNoRules.csv")) if l.strip()]
    rows = []
    for i, (m, r) in enumerate(zip(meta, risk)):
        m = np.array(m, float)
        r = np.array(r, float)
        for o in range((len(m) - 2) // 5):
            h, d, s, b, c = m[2 + 5 * o:7 + 5 * o]
            rows.append((i, o, m[0], m[1], h, d, s, b, c, r[o]))
    return np.array(rows)


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    A = load(ref)
    row, idx, ss, ho, ht, d, rs, b, c, r = [A[:, i] for i in range(10)]
    static = (ht == 0) & (rs == ss)
    model = ref_cri.cri_rows(ho, d, rs, b, c)
    e = r - model
    print("static : n=%d  max |reference - model| = %.2e" % (static.sum(), np.abs(e[static]).max()))
    dyn = ~static
    print("moving : n=%d  reference - static rule: min %.3f  max %.3f  exact %d  reference == 0: %d"
          % (dyn.sum(), e[dyn].min(), e[dyn].max(), (np.abs(e[dyn]) < 1e-9).sum(), (r[dyn] == 0).sum()))
    bw = b + ho
    px, py = d * np.cos(bw), d * np.sin(bw)
    along = px * np.cos(c) + py * np.sin(c)
    cross = px * np.sin(c) - py * np.cos(c)
    k = dyn & (ss < 0.02) & (along > 33) & (d > 46) & (r > 0)
    delta = 0.2 * ref_cri.bearing_membership(b[k]) - r[k]
    theta = np.degrees(np.arctan2(np.abs(cross[k]), along[k]))
    o = np.argsort(theta)
    print("vessel at rest, obstacle opening (angle off the target's stern, delta):")
    for t, dl in zip(theta[o][::3], delta[o][::3]):
        print("   %6.2f deg   %.5f" % (t, dl))


if __name__ == "__main__":
    main()
