"""Builds tests/golden/contact_golden.npz from the reference's dataset file.

    python tests/golden/make_contact_golden.py [/root/reference]

The reference logs no contact flag, but its metadata file
(data/metadata_MovingObstaclesNoRules.csv; columns per utils/explore_dataset.py:8,40-53)
holds what a contact predicate can be checked against:

  * every record whose logged `dist` (centre distance - size - 3.255 m) is
    below 5 m -- 62 records of 12 close passes, 11 of them with dist < 0 --
    together with how many more rows its episode went on for: a pass the
    reference's vessel SURVIVED;
  * the 36 episodes of the file (a new one starts where speed_OS returns to
    exactly 0), each with the nearest logged obstacle at its last row.

The fixture holds DATA only.  Per near-contact record: row in the file
(counting from 0), the obstacle's place in the row, static / moving, size
(recovered exactly by make_scan_golden.py; needs scan_golden.npz),
heading_OS, speed_OS, the five logged columns, and rows left in the episode.
Per episode: first and last row, nearest `dist` at the last row and over the
whole episode.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NEAR = 5.0


def main(ref):
    rows = [np.array(line.split(), dtype=np.float64)
            for line in open(os.path.join(ref, "data", "metadata_MovingObstaclesNoRules.csv"))]
    n = len(rows)
    speed = np.array([r[0] for r in rows])
    starts = [0] + [i for i in range(1, n) if speed[i] == 0.0 and speed[i - 1] != 0.0]
    ends = [s - 1 for s in starts[1:]] + [n - 1]
    ep_of = np.zeros(n, dtype=np.int32)
    for k, s in enumerate(starts):
        ep_of[s:] = k
    g = np.load(os.path.join(HERE, "scan_golden.npz"))
    kept = {int(r): i for i, r in enumerate(g["row"])}
    out = {k: [] for k in ("row", "slot", "static", "size", "heading_os", "speed_os", "record", "rows_left", "episode")}
    for i, r in enumerate(rows):
        rec = r[2:].reshape(-1, 5)
        for j, q in enumerate(rec):
            if q[1] >= NEAR:
                continue
            if i not in kept:
                raise SystemExit(f"row {i}: size not recovered")
            o = int(g["first"][kept[i]]) + j
            assert g["dist"][o] == q[1] and g["bearing"][o] == q[3]
            out["row"].append(i)
            out["slot"].append(j)
            out["static"].append(bool(q[0] == 0.0 and q[2] == r[0]))      # utils/explore_dataset.py:46
            out["size"].append(int(g["size"][o]))
            out["heading_os"].append(r[1])
            out["speed_os"].append(r[0])
            out["record"].append(q)
            out["rows_left"].append(ends[ep_of[i]] - i)
            out["episode"].append(ep_of[i])
    ep = {"ep_first": np.array(starts, dtype=np.int32), "ep_last": np.array(ends, dtype=np.int32),
          "ep_last_row_nearest": np.array([rows[e][2:].reshape(-1, 5)[:, 1].min() for e in ends]),
          "ep_nearest": np.array([min(rows[i][2:].reshape(-1, 5)[:, 1].min() for i in range(s, e + 1))
                                  for s, e in zip(starts, ends)]),
          "ep_last_speed": speed[ends]}
    arr = {"row": np.array(out["row"], dtype=np.int32), "slot": np.array(out["slot"], dtype=np.int32),
           "static": np.array(out["static"]), "size": np.array(out["size"], dtype=np.int32),
           "heading_os": np.array(out["heading_os"]), "speed_os": np.array(out["speed_os"]),
           "record": np.array(out["record"]), "rows_left": np.array(out["rows_left"], dtype=np.int32),
           "episode": np.array(out["episode"], dtype=np.int32), "n_rows_file": np.int32(n)}
    np.savez_compressed(os.path.join(HERE, "contact_golden.npz"), **arr, **ep)
    print(len(arr["row"]), "near-contact records,", int((arr["record"][:, 1] < 0).sum()), "with dist < 0, fewest rows left",
          int(arr["rows_left"].min()), ";", len(starts), "episodes, nearest obstacle at an episode's last row:",
          float(ep["ep_last_row_nearest"].min()))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/root/reference")
