"""How well `oracle/ref_cri.py` reproduces the reference's dataset, split by
obstacle kind (run here, next to the reference checkout; not part of the
package or the tests).

    python scratch/cri_explore.py [/root/reference]
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
    e = np.abs(r - ref_cri.cri_rows(ss, ho, ht, d, rs, b, c))
    print("static : n=%d  max |reference - oracle| = %.2e" % (static.sum(), e[static].max()))
    print("moving : n=%d  max |reference - oracle| = %.2e" % ((~static).sum(), e[~static].max()))
    raw = np.abs(ht - ho)[~static]
    print("moving records with |heading_ts - heading_OS| > pi (mirrored course): %d" % (raw > np.pi).sum())


if __name__ == "__main__":
    main()
