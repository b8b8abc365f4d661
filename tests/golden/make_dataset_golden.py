"""Generate the dataset-side fixtures from the reference's shipped data and its
own loader code.  Runs only where /root/reference is mounted.

  risk_rows.csv / metadata_rows.csv   the first 300 rows of
      data/risk_MovingObstaclesNoRules.csv and data/metadata_MovingObstaclesNoRules.csv
      (data files, copied verbatim: they pin the text format and the ragged layout)
  dataset_golden.npz                  what the reference's utils/dataloader.py computes on
      those rows: calculate_total_risk(mode) for max / sum / mean (:104-120) and the table pandas
      parses those rows into (the call of :108),
      prev_timesteps_feature_enginering (:123-134) on a small scan array, and the
      split sizes of load_LiDARDataset (:63-67) for the full 10,075-row set.

utils/dataloader.py:22 uses `np.float`, removed from numpy >= 1.24; the alias is
restored before the import (the module is otherwise imported unmodified).
"""
import os
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
N_ROWS = 300


def main():
    for name in ("risk", "metadata"):
        src = os.path.join(REF, "data", f"{name}_MovingObstaclesNoRules.csv")
        with open(src) as f:
            rows = [next(f) for _ in range(N_ROWS)]
        with open(os.path.join(HERE, f"{name}_rows.csv"), "w") as f:
            f.writelines(rows)
    np.float = float  # noqa: removed alias the reference module still names
    sys.path.insert(0, REF)
    from utils import dataloader as D

    out = {}
    risk_path = os.path.join(HERE, "risk_rows.csv")
    for mode in ("max", "sum", "mean"):
        out[f"label_{mode}"] = D.calculate_total_risk(risk_path, mode)
    # the table those labels were reduced from, as the reference's read_csv call (:108) parses the text: pandas'
    # default number parser is not correctly rounded (a quarter of the values are one unit in the last place
    # off), so the float64 labels above follow from THIS table, bit for bit, not from the text read exactly
    import pandas as pd

    out["risk_parsed"] = pd.read_csv(risk_path, delimiter=r"\s+", header=None, names=list(range(8))).to_numpy()
    # the rows of the whole shipped file that hold 4 or more risks (1,887 of 10,075; the first 300 rows hold at
    # most 3, where the order of addition cannot show): table as parsed, and the reference's labels of them
    full = os.path.join(REF, "data", "risk_MovingObstaclesNoRules.csv")
    table = pd.read_csv(full, delimiter=r"\s+", header=None, names=list(range(8))).to_numpy()
    many = np.flatnonzero((~np.isnan(table)).sum(axis=1) >= 4)
    out["many_rows"], out["many_parsed"] = many, table[many]
    for mode in ("max", "sum", "mean"):
        out[f"many_label_{mode}"] = D.calculate_total_risk(full, mode)[many]
    X = np.random.default_rng(0).uniform(0, 150, (7, D.N_SENSORS))
    out["prev_in"] = X
    out["prev_out"] = D.prev_timesteps_feature_enginering(X, 3)
    # split arithmetic of load_LiDARDataset for the shipped dataset size
    n = sum(1 for line in open(os.path.join(REF, "data", "risk_MovingObstaclesNoRules.csv")) if line.strip())
    train = int(0.7 * n)
    val = int(0.3 * train)
    out["split_full"] = np.array([n, train - val, val, n - train])
    np.savez_compressed(os.path.join(HERE, "dataset_golden.npz"), **out)
    print({k: np.asarray(v).shape for k, v in out.items()}, out["split_full"], out["label_max"][:3])


if __name__ == "__main__":
    main()
