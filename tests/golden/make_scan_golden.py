"""Builds tests/golden/scan_golden.npz from the reference's dataset files.

    python tests/golden/make_scan_golden.py [/root/reference]

The LiDAR scans the reference logged (data/LiDAR_MovingObstaclesNoRules.csv)
are not in the checkout, but the scenes they were taken of can be rebuilt from
the metadata rows (utils/explore_dataset.py:8,40-53): every listed obstacle is
logged as (heading_ts, dist, speed, bearing, course) with
dist = centre distance - size - 3.255 m (make_track_golden.py,
make_ship_golden.py), so its centre lies at dist + size + 3.255 along `bearing`
in the vessel's frame once its size is known.

Sizes are recovered exactly, not estimated.  Needs track_golden.npz and
ship_golden.npz, whose obstacles have known offsets K = size + 3.255.  A static
obstacle with a known K seen in two consecutive rows is a landmark: it gives
the vessel's displacement between the rows,

    dp = (dist_a + K) e(bearing_a + psi_a) - (dist_b + K) e(bearing_b + psi_b),

and with dp every other static obstacle of the two rows follows from two
equations in its one unknown K (kept when met to 1e-6 m AND K - 3.255 is an
integer to 1e-5, which a wrong pairing never is), every moving one from two
equations in (K, speed).  New K's are new landmarks; the sweep runs forwards
and backwards to a fixed point.  Where the vessel does not move there is no
parallax: records that repeat (same dist and bearing to 1e-9) carry their K
over.

The fixture holds DATA only, per listed obstacle of every row whose obstacles
all have a recovered size: heading_ts, dist, bearing, integer size, class; per
row: its index in the file, heading_OS, and the label the reference trains
against (row-wise max of data/risk_*.csv, utils/dataloader.py:113-114); and the
mean and standard deviation of the test errors the reference publishes for its
three networks (results_RL/_mse_1conv.txt, _mse_3conv.txt, _mse_deep.txt); and
the bucket counts of the whole file as utils/explore_dataset.py:40-53 tallies
them (static / dynamic x distance, risk, relative speed; obstacles per row).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CLEARANCE = 3.255
LOG_EVERY = 5


def wrap(a):
    return (a + np.pi) % (2.0 * np.pi) - np.pi


def is_int(k):
    return abs((k - CLEARANCE) - round(k - CLEARANCE)) < 1e-5


class Log:
    def __init__(self, ref):
        self.rows = [np.array(line.split(), dtype=np.float64)
                     for line in open(os.path.join(ref, "data", "metadata_MovingObstaclesNoRules.csv"))]
        self.risk = [np.array(line.split(), dtype=np.float64)
                     for line in open(os.path.join(ref, "data", "risk_MovingObstaclesNoRules.csv")) if line.strip()]
        n = self.n = len(self.rows)
        self.speed = np.array([r[0] for r in self.rows])
        self.psi = np.array([r[1] for r in self.rows])
        starts = [0] + [i for i in range(1, n) if self.speed[i] == 0.0 and self.speed[i - 1] != 0.0]
        self.episode = np.zeros(n, dtype=np.int64)
        for e, s in enumerate(starts):
            self.episode[s:] = e
        self.recs = [r[2:].reshape(-1, 5) for r in self.rows]   # heading_ts, dist, speed, bearing, course
        # utils/explore_dataset.py:46
        self.static = [(R[:, 0] == 0) & (R[:, 2] == self.rows[i][0]) for i, R in enumerate(self.recs)]
        self.e = [np.stack([np.cos(R[:, 3] + self.psi[i]), np.sin(R[:, 3] + self.psi[i])], axis=1)
                  for i, R in enumerate(self.recs)]
        self.K = [np.full(len(R), np.nan) for R in self.recs]
        self.match = [np.full(len(R), -1) for R in self.recs]
        self.dp = {}
        # a ship is known by its episode and heading_ts; once solved, (K, speed) hold for all its records
        self.ships = {}
        self.ship_rows = {}
        for i, R in enumerate(self.recs):
            for o in np.where(~self.static[i])[0]:
                self.ship_rows.setdefault(self.ship_key(i, o), []).append((i, o))

    def ship_key(self, i, o):
        return (int(self.episode[i]), round(float(self.recs[i][o, 0]), 9))

    def learn_ship(self, i, o, k, s):
        key = self.ship_key(i, o)
        if key in self.ships:
            return 0
        self.ships[key] = (k, s)
        new = 0
        for r, j in self.ship_rows[key]:
            if not np.isfinite(self.K[r][j]):
                self.K[r][j] = k
                new += 1
        return new

    def same_episode(self, i):
        return i + 1 < self.n and self.episode[i + 1] == self.episode[i]

    def seed(self):
        t = np.load(os.path.join(HERE, "track_golden.npz"))
        for r, da, db, k in zip(t["row"], t["dist_a"], t["dist_b"], t["offset"]):
            self.K[r][np.where(self.recs[r][:, 1] == da)[0][0]] = k
            self.K[r + 1][np.where(self.recs[r + 1][:, 1] == db)[0][0]] = k
        s = np.load(os.path.join(HERE, "ship_golden.npz"))
        for j, r in enumerate(s["row"]):
            o = np.where(self.recs[r][:, 1] == s["records"][j][0][1])[0][0]
            self.learn_ship(r, o, float(s["offset"][j]), float(s["speed_ts"][j]))

    def known(self):
        return int(sum(np.isfinite(k).sum() for k in self.K))

    def solve_static(self, i, o, q, dp):
        ea, eb = self.e[i][o], self.e[i + 1][q]
        m = ea - eb
        if m @ m < 1e-12:
            return None
        rhs = dp + self.recs[i + 1][q, 1] * eb - self.recs[i][o, 1] * ea
        k = (m @ rhs) / (m @ m)
        return k if np.abs(k * m - rhs).max() < 1e-6 else None

    def find_displacement(self, i):
        """The vessel's displacement between rows i and i+1 from a landmark,
        sanity-checked against the logged speeds and headings; among several
        candidates the one most other static obstacles agree with."""
        A, B = self.recs[i], self.recs[i + 1]
        sa, sb = np.where(self.static[i])[0], np.where(self.static[i + 1])[0]
        best = None
        for o in sa:
            for q in sb:
                ka, kb = self.K[i][o], self.K[i + 1][q]
                k = ka if np.isfinite(ka) else kb
                if not np.isfinite(k) or (np.isfinite(ka) and np.isfinite(kb) and abs(ka - kb) > 1e-6):
                    continue
                if abs(B[q, 1] - A[o, 1]) > 7.0:
                    continue
                dp = (A[o, 1] + k) * self.e[i][o] - (B[q, 1] + k) * self.e[i + 1][q]
                length = np.hypot(*dp)
                if abs(length - 0.5 * LOG_EVERY * (self.speed[i] + self.speed[i + 1])) > 1.0:
                    continue
                if length > 1.0:
                    mid = self.psi[i] + 0.5 * wrap(self.psi[i + 1] - self.psi[i])
                    if abs(wrap(np.arctan2(dp[1], dp[0]) - mid)) > 0.6:
                        continue
                support = 0
                for o2 in sa:
                    for q2 in sb:
                        if o2 != o and q2 != q:
                            k2 = self.solve_static(i, o2, q2, dp)
                            support += k2 is not None and is_int(k2) and 10.0 < k2 < 50.0
                if best is None or support > best[0]:
                    best = (support, dp)
        if best is not None:
            return best[1]
        # no static landmark: a ship whose size and speed are known serves as one
        for o in np.where(~self.static[i])[0]:
            key = self.ship_key(i, o)
            if key not in self.ships:
                continue
            k, s = self.ships[key]
            for q in np.where(~self.static[i + 1])[0]:
                if self.ship_key(i + 1, q) != key:
                    continue
                h = A[o, 0]
                dp = ((A[o, 1] + k) * self.e[i][o] - (B[q, 1] + k) * self.e[i + 1][q]
                      + LOG_EVERY * s * np.array([np.cos(h), np.sin(h)]))
                if abs(np.hypot(*dp) - 0.5 * LOG_EVERY * (self.speed[i] + self.speed[i + 1])) <= 1.0:
                    return dp
        return None

    def sweep(self, order):
        new = 0
        for i in order:
            if not self.same_episode(i):
                continue
            if i not in self.dp:
                dp = self.find_displacement(i)
                if dp is None:
                    continue
                self.dp[i] = dp
            dp = self.dp[i]
            Ka, Kb = self.K[i], self.K[i + 1]

            def accept(o, q, k):
                nonlocal new
                if np.isfinite(Ka[o]) and abs(Ka[o] - k) > 1e-5:
                    return False
                if np.isfinite(Kb[q]) and abs(Kb[q] - k) > 1e-5:
                    return False
                self.match[i][o] = q
                for arr, j in ((Ka, o), (Kb, q)):
                    if not np.isfinite(arr[j]):
                        arr[j] = k
                        new += 1
                return True

            for o in np.where(self.static[i])[0]:
                if self.match[i][o] >= 0:
                    continue
                for q in np.where(self.static[i + 1])[0]:
                    if q in self.match[i]:
                        continue
                    k = self.solve_static(i, o, q, dp)
                    if k is not None and is_int(k) and 10.0 < k < 50.0 and accept(o, q, k):
                        break
            for o in np.where(~self.static[i])[0]:
                if self.match[i][o] >= 0:
                    continue
                for q in np.where(~self.static[i + 1])[0]:
                    if abs(self.recs[i + 1][q, 0] - self.recs[i][o, 0]) > 1e-9:
                        continue
                    # p_b + (d_b + K) e_b = p_a + (d_a + K) e_a + 5 s e(heading_ts)
                    h = self.recs[i][o, 0]
                    eh = np.array([np.cos(h), np.sin(h)])
                    ea, eb = self.e[i][o], self.e[i + 1][q]
                    M = np.array([[eb[0] - ea[0], -LOG_EVERY * eh[0]], [eb[1] - ea[1], -LOG_EVERY * eh[1]]])
                    rhs = -(dp + self.recs[i + 1][q, 1] * eb - self.recs[i][o, 1] * ea)
                    if abs(np.linalg.det(M)) < 1e-9:
                        continue
                    k, s = np.linalg.solve(M, rhs)
                    if is_int(k) and 0.9 < k - CLEARANCE < 30.0 and -0.01 <= s < 3.0 and accept(o, q, k):
                        new += self.learn_ship(i, o, k, s)
                        break
        return new

    def carry_repeats(self):
        """No parallax while the vessel rests: a record that repeats in the next
        row (same class, dist and bearing to 1e-9) is the same obstacle."""
        new = 0
        for i in range(self.n - 1):
            if not self.same_episode(i):
                continue
            A, B = self.recs[i], self.recs[i + 1]
            for o in range(len(A)):
                if self.match[i][o] >= 0:
                    continue
                for q in range(len(B)):
                    if (self.static[i][o] == self.static[i + 1][q] and abs(A[o, 1] - B[q, 1]) < 1e-9
                            and abs(wrap(A[o, 3] + self.psi[i] - B[q, 3] - self.psi[i + 1])) < 1e-9
                            and abs(A[o, 0] - B[q, 0]) < 1e-9 and q not in self.match[i]):
                        self.match[i][o] = q
                        break
        changed = True
        while changed:
            changed = False
            for i in range(self.n - 1):
                for o, q in enumerate(self.match[i]):
                    if q < 0:
                        continue
                    a, b = self.K[i][o], self.K[i + 1][q]
                    if np.isfinite(a) != np.isfinite(b):
                        if np.isfinite(a):
                            self.K[i + 1][q] = a
                        else:
                            self.K[i][o] = b
                        new += 1
                        changed = True
        return new

    def recover(self):
        self.seed()
        print("records", sum(len(R) for R in self.recs), "seeded", self.known())
        while True:
            new = self.sweep(range(self.n - 1)) + self.sweep(range(self.n - 2, -1, -1)) + self.carry_repeats()
            print("  sweep: +%d, known %d, row pairs with a displacement %d" % (new, self.known(), len(self.dp)))
            if new == 0:
                break


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    log = Log(ref)
    log.recover()
    total = sum(len(R) for R in log.recs)
    full = [i for i in range(log.n) if np.isfinite(log.K[i]).all()]
    n_test0 = int(0.7 * log.n)   # utils/dataloader.py:63-67: the test split is the last 30 % of the rows
    print("sizes recovered for %d of %d records; rows with every size known: %d of %d (test split: %d of %d)"
          % (log.known(), total, len(full), log.n, sum(i >= n_test0 for i in full), log.n - n_test0))
    # every chain of matched records must carry one size
    for i in range(log.n - 1):
        for o, q in enumerate(log.match[i]):
            assert q < 0 or not np.isfinite(log.K[i][o]) or abs(log.K[i][o] - log.K[i + 1][q]) < 1e-5
    row, psi, label, first, cols = [], [], [], [0], {k: [] for k in ("heading_ts", "dist", "bearing", "size", "static")}
    for i in full:
        R = log.recs[i]
        row.append(i)
        psi.append(log.psi[i])
        label.append(log.risk[i].max())
        cols["heading_ts"].extend(R[:, 0])
        cols["dist"].extend(R[:, 1])
        cols["bearing"].extend(R[:, 3])
        cols["size"].extend(np.rint(log.K[i] - CLEARANCE))
        cols["static"].extend(log.static[i])
        first.append(first[-1] + len(R))
    size = np.array(cols["size"])
    st = np.array(cols["static"], dtype=bool)
    print("static radii", int(size[st].min()), "..", int(size[st].max()), "; ship sizes", int(size[~st].min()), "..",
          int(size[~st].max()))
    # the test errors the reference publishes for its three networks (results_RL/_mse_*.txt, 100 values each)
    pub = [np.loadtxt(os.path.join(ref, "results_RL", f"_mse_{k}.txt")) for k in ("1conv", "3conv", "deep")]
    print("published test MSE (shallow, 2_deep, deep): mean", [float(p.mean()) for p in pub], "std", [float(p.std()) for p in pub])
    # the reference's own summary of its dataset (the buckets of utils/explore_dataset.py:40-53), over every record
    recs = np.concatenate(log.recs)
    own_speed = np.concatenate([np.full(len(R), log.speed[i]) for i, R in enumerate(log.recs)])
    risk_all = np.concatenate(log.risk)
    st_all = np.concatenate(log.static)
    buckets = {}
    for name, m in (("static", st_all), ("dynamic", ~st_all)):
        buckets[f"bucket_{name}_dist"] = np.histogram(recs[m, 1], bins=[-1e9, 50.0, 100.0, 1e9])[0]
        buckets[f"bucket_{name}_risk"] = np.histogram(risk_all[m], bins=[-1.0, 0.2, 0.4, 0.6, 0.8, 1.01])[0]
        buckets[f"bucket_{name}_speed"] = np.histogram(recs[m, 2], bins=[-1.0, 0.5, 1.0, 1.5, 1e9])[0]
    buckets["bucket_count"] = np.bincount([len(R) for R in log.recs], minlength=9)
    print("bucket table:", {k: v.tolist() for k, v in buckets.items()})
    dst = os.path.join(HERE, "scan_golden.npz")
    np.savez_compressed(dst, row=np.array(row, dtype=np.int32), heading_os=np.array(psi), label=np.array(label),
                        first=np.array(first, dtype=np.int32), heading_ts=np.array(cols["heading_ts"]),
                        dist=np.array(cols["dist"]), bearing=np.array(cols["bearing"]),
                        size=size.astype(np.int8), static=st, n_rows_file=np.int32(log.n),
                        published_mse_mean=np.array([p.mean() for p in pub]),
                        published_mse_std=np.array([p.std() for p in pub]), **buckets)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
