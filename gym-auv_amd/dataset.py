"""The LiDAR-risk dataset: writing rows from a running simulation in the
reference's three text formats, and loading them back the way the
reference's trainer does.

Formats (numpy.savetxt defaults, "%.18e", space separated), one row per
logged sample, the three files row-aligned:
  LiDAR_<name>.csv     180 ranges in metres           (utils/dataloader.py:55-56)
  risk_<name>.csv      one risk per tracked obstacle  (utils/dataloader.py:108)
  metadata_<name>.csv  speed_OS heading_OS then (heading_ts dist speed bearing
                       course) per tracked obstacle   (utils/explore_dataset.py:8,40-53)
A sample is logged only when at least one obstacle is tracked (every shipped
row has one).  Which obstacles a row lists follows the rule the shipped rows
show (`update_tracked`): the list is rebuilt every 25 steps of an episode from
the obstacles whose logged distance is below the sensor range, and kept as it
is in between -- which is why the shipped rows hold obstacles out to 176 m.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib

N_SENSORS = 180       # utils/dataloader.py:13
SENSOR_RANGE = 150.0  # utils/dataloader.py:56


# ----------------------------------------------------------------------------
# writing
# ----------------------------------------------------------------------------
def format_rows(values: np.ndarray, row_len=None, view: bool = False):
    """Text of ragged rows: row i = values[i, :row_len[i]] (all columns when
    row_len is None), as bytes (or, with view=True, as a uint8 array without the
    copy).  Formatting runs in the native library, on the host threads."""
    lib = _lib.load()
    values = np.ascontiguousarray(values, dtype=np.float64)
    if values.ndim != 2:
        raise ValueError("values must be 2-D")
    n, stride = values.shape
    if row_len is None:
        row_len = np.full(n, stride, dtype=np.int32)
    row_len = np.ascontiguousarray(row_len, dtype=np.int32)
    if row_len.shape != (n,) or (row_len > stride).any() or (row_len < 0).any():
        raise ValueError("row_len")
    cap = int(row_len.sum()) * 27 + n + 64
    buf = np.empty(cap, dtype=np.uint8)
    got = lib.auv_format_rows(values.ctypes.data_as(C.c_void_p), row_len.ctypes.data_as(C.c_void_p), n, stride,
                              buf.ctypes.data_as(C.c_void_p), cap)
    if got < 0:
        raise _lib.AuvError(int(got), "auv_format_rows")
    return buf[:got] if view else buf[:got].tobytes()


class DatasetWriter:
    """Appends row-aligned samples to LiDAR_/risk_/metadata_<name>.csv.  Text
    goes out through positional writes; a large block is split over a few
    threads (the copy into the page cache is what a logged step of a large
    batch waits for)."""

    CHUNK = 8 << 20     # bytes per positional write of a split block
    THREADS = 8

    def __init__(self, directory: str, name: str = "MovingObstaclesNoRules"):
        os.makedirs(directory, exist_ok=True)
        self.paths = {k: os.path.join(directory, f"{k}_{name}.csv") for k in ("LiDAR", "risk", "metadata")}
        self._fd = {k: os.open(p, os.O_WRONLY | os.O_CREAT, 0o644) for k, p in self.paths.items()}
        self._end = {k: os.fstat(fd).st_size for k, fd in self._fd.items()}      # append behind what is there
        self._pool = None
        self.rows = 0

    @staticmethod
    def _put(fd, view, at):
        while len(view):
            n = os.pwrite(fd, view, at)
            view, at = view[n:], at + n

    def write_text(self, kind, text):
        """Append rows already formatted (bytes, or a uint8 array) to one of the three files."""
        self.write_texts({kind: text})

    def write_texts(self, texts):
        """Append a block of formatted rows to each named file; returns when all of it is written."""
        jobs = []
        for kind, text in texts.items():
            view = memoryview(text).cast("B")
            at, fd = self._end[kind], self._fd[kind]
            self._end[kind] += len(view)
            for o in range(0, len(view), self.CHUNK):
                jobs.append((fd, view[o:o + self.CHUNK], at + o))
        if len(jobs) == 1:
            self._put(*jobs[0])
        elif jobs:
            if self._pool is None:
                from concurrent.futures import ThreadPoolExecutor
                self._pool = ThreadPoolExecutor(self.THREADS)
            for j in [self._pool.submit(self._put, *job) for job in jobs]:
                j.result()

    def write(self, ranges, own, meta, risk, tracked):
        """ranges [n,R], own [n,2], meta [n,O,5], risk [n,O], tracked bool [n,O].
        Samples without a tracked obstacle are skipped.  Returns rows written."""
        tracked = np.asarray(tracked, dtype=bool)
        keep = tracked.any(axis=1)
        if not keep.any():
            return 0
        ranges, own = np.asarray(ranges)[keep], np.asarray(own)[keep]
        meta, risk, tracked = np.asarray(meta)[keep], np.asarray(risk)[keep], tracked[keep]
        n, O = tracked.shape
        cnt = tracked.sum(axis=1).astype(np.int32)
        # compact the tracked obstacles of each sample to the front, id order kept
        order = np.argsort(~tracked, axis=1, kind="stable")
        risk_c = np.take_along_axis(risk, order, axis=1)
        meta_c = np.take_along_axis(meta, order[:, :, None], axis=1).reshape(n, O * 5)
        self.write_texts({"LiDAR": format_rows(ranges, view=True), "risk": format_rows(risk_c, cnt, view=True),
                          "metadata": format_rows(np.concatenate([own, meta_c], axis=1), 2 + 5 * cnt, view=True)})
        self.rows += n
        return n

    def close(self):
        for fd in self._fd.values():
            os.close(fd)
        self._fd = {}
        if self._pool is not None:
            self._pool.shutdown()
            self._pool = None

    def __del__(self):
        try:
            self.close()        # raw descriptors are not closed by the garbage collector
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def update_tracked(tracked, t_step, present, dist, sensor_range=SENSOR_RANGE, refresh_every=25):
    """The obstacle list of every env after one more step.

    The reference's rows (data/metadata_MovingObstaclesNoRules.csv) change their
    obstacle lists almost only at every fifth logged row, i.e. every 25 steps:
    at those steps every obstacle whose logged `dist` is below the sensor range
    is listed (an obstacle never enters a list at 150 m or more: the largest
    `dist` in a row whose list has just changed is 149.93 m), and until the
    next rebuild the list stays as it is, so that obstacles sailing away remain
    listed out to 176 m.  tracked: bool [V,O] lists so far (None: build them
    now); t_step [V]: steps into the episode; present: bool [V,O] slot in use;
    dist [V,O]: the logged distance column."""
    fresh = present & (dist < sensor_range)
    if tracked is None:
        return fresh
    rebuild = (np.asarray(t_step) % int(refresh_every)) == 0
    return np.where(rebuild[:, None], fresh, tracked & present)


class DatasetLogger:
    """Logs the envs of an `AuvSim` whose episode is at a multiple of `every`
    steps (the shipped rows are 5 steps = 5 s of simulated time apart), with
    the obstacle lists kept by `update_tracked`."""

    def __init__(self, sim, writer: DatasetWriter, every: int = 5, refresh_every: int = 25):
        self.sim, self.writer, self.every, self.refresh_every = sim, writer, int(every), int(refresh_every)
        self._tracked = None

    def after_step(self, stream=None):
        sim = self.sim
        t = sim.read("t_step", stream, pinned=True)
        due = (t % self.every) == 0
        if not due.any():
            return 0
        # the lists only change at multiples of refresh_every, which `every` divides: looking at logging steps is enough
        meta = sim.read("meta", stream, pinned=True)
        present = np.isfinite(sim.read("obst_dist", stream, pinned=True))
        self._tracked = update_tracked(self._tracked, t, present, meta[:, :, 1], sim.cfg["sensor_range"],
                                       self.refresh_every)
        tracked = self._tracked & due[:, None]
        return self.writer.write(sim.read("ranges", stream, pinned=True), sim.read("own", stream, pinned=True),
                                 meta, sim.read("risk", stream, pinned=True), tracked)


class DeviceRowFormatter:
    """Text of ragged rows formed on the GPU (`auv_format_rows_device`): the
    rows leave the device as the bytes that go to the file.  One instance keeps
    its device text buffer, row offsets and page-locked host buffer between
    calls (grown when a call needs more)."""

    def __init__(self, device: int = 0):
        self._lib = _lib.load()
        self.device = int(device)
        self._text = self._off = self._status = self._host = None
        self.host_fallbacks = 0     # blocks that held a value outside the device's range

    def format(self, values, row_len=None, stream=None):
        """values: contiguous float64 CUDA tensor [n, stride]; row_len: int32 CUDA
        tensor [n] or None (whole rows).  Launches the formatting on `stream` (a
        torch stream; default: the current one) and returns a ticket for `fetch`."""
        import torch

        if not (values.is_cuda and values.dtype == torch.float64 and values.dim() == 2 and values.is_contiguous()):
            raise ValueError("values must be a contiguous 2-D float64 CUDA tensor")
        if values.device.index != self.device:
            raise ValueError(f"values on cuda:{values.device.index}, formatter on cuda:{self.device}")
        n, stride = values.shape
        if stride < 1:
            raise ValueError("rows need at least one column")
        if row_len is not None and not (row_len.is_cuda and row_len.dtype == torch.int32 and row_len.shape == (n,)
                                        and row_len.is_contiguous() and row_len.device == values.device):
            raise ValueError("row_len must be a contiguous int32 CUDA tensor [n] on the values' device")
        cap = 26 * n * stride + n + 64        # what the widest rows would take
        dev = values.device
        if self._text is None or self._text.numel() < cap:
            self._text = torch.empty(cap, dtype=torch.uint8, device=dev)
        if self._off is None or self._off.numel() < n + 3:
            self._off = torch.empty(n + 3, dtype=torch.int64, device=dev)   # row offsets, total, then the status words
        st = torch.cuda.current_stream(dev) if stream is None else stream
        off = self._off
        status_ptr = off.data_ptr() + 8 * (n + 1)
        rc = self._lib.auv_format_rows_device(self.device, C.c_void_p(values.data_ptr()),
                                              None if row_len is None else C.c_void_p(row_len.data_ptr()), n, stride,
                                              C.c_void_p(self._text.data_ptr()), self._text.numel(),
                                              C.c_void_p(off.data_ptr()), C.c_void_p(status_ptr),
                                              C.c_void_p(st.cuda_stream))
        if rc != 0:
            raise _lib.AuvError(rc, "auv_format_rows_device")
        return {"values": values, "row_len": row_len, "n": n, "stream": st}

    def tail(self, ticket):
        """The two words to read back before `fetch`: [bytes in all, status] (device tensor view, int64)."""
        return self._off[ticket["n"]:ticket["n"] + 2]

    def fetch(self, ticket, tail=None):
        """The text of a `format` call as a uint8 numpy array (a view of this object's page-locked buffer, valid
        until the next fetch).  `tail`: the two words of `tail(ticket)` already on the host (saves a round trip
        when several formatters are read together)."""
        import torch

        with torch.cuda.stream(ticket["stream"]):
            if tail is None:
                tail = self.tail(ticket).cpu()
            total = int(tail[0])
            bad, overflow = int(tail[1]) & 0xffffffff, int(tail[1]) >> 32
            if overflow:
                raise RuntimeError("auv_format_rows_device: text buffer too small")   # cannot happen: sized for the widest rows
            if bad:
                # infinities, NaN or magnitudes outside 1e-36 .. 1e18: the host formatter handles every double
                self.host_fallbacks += 1
                rl = None if ticket["row_len"] is None else ticket["row_len"].cpu().numpy()
                return format_rows(ticket["values"].cpu().numpy(), rl, view=True)
            if self._host is None or self._host.numel() < total:
                self._host = torch.empty(max(total, 1 << 20), dtype=torch.uint8, pin_memory=True)
            self._host[:total].copy_(self._text[:total], non_blocking=True)
            ticket["stream"].synchronize()
        return self._host[:total].numpy()


def format_rows_device(values, row_len=None):
    """One-off form of `DeviceRowFormatter`: bytes of the rows of a CUDA tensor."""
    f = DeviceRowFormatter(values.device.index)
    return f.fetch(f.format(values, row_len)).tobytes()


class DeviceDatasetLogger:
    """`DatasetLogger` with the rows built and formatted on the GPU: the
    obstacle lists (`update_tracked`), the choice and compaction of the rows
    (`auv_dataset_rows`) and the "%.18e" text (`auv_format_rows_device`) all
    stay on the device; the host receives the bytes of the three files and
    writes them.  Same files, byte for byte, as `DatasetLogger` writes from the
    same simulation.  `stream`: the torch stream the simulation is stepped on
    (default: the current one)."""

    def __init__(self, sim, writer: DatasetWriter | None, every: int = 5, refresh_every: int = 25,
                 collect: str | None = None):
        """writer None: no files.  collect "max" / "sum" / "mean": every logged row is also kept on the device
        as a training sample, the scan as the network sees it and the label `calculate_total_risk` would form
        from the row's risk record; `samples()` returns what `load_lidar_dataset(..., mode=collect)` would
        read back from the files, without the files."""
        import torch

        self.sim, self.writer, self.every, self.refresh_every = sim, writer, int(every), int(refresh_every)
        if writer is None and collect is None:
            raise ValueError("nothing to do: no writer and no samples to collect")
        if collect not in _lib.ROWS_LABEL:
            raise ValueError("collect must be None, 'max', 'sum' or 'mean'")
        if self.every < 1 or self.refresh_every < 1:
            raise ValueError("every and refresh_every must be >= 1")
        self.collect = collect
        self._lib = _lib.load()
        V, R = sim.V, sim.R
        O = int(sim.cfg["n_circles"]) + int(sim.cfg["n_polys"])
        self._V, self._R, self._O = V, R, O
        dev = torch.device("cuda", sim.device)
        new = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        # the lists, and room for a row of every env (device memory: torch is the allocator here, nothing more)
        self._tracked = torch.zeros((V, O), dtype=torch.uint8, device=dev)
        self._row_of, self._due = new((V + 1,), torch.int64), torch.zeros(1, dtype=torch.int32, device=dev)
        self._lidar, self._risk, self._meta = new((V, R), torch.float64), new((V, O), torch.float64), new((V, 2 + 5 * O), torch.float64)
        self._risk_len, self._meta_len = new((V,), torch.int32), new((V,), torch.int32)
        self._x = new((V, R), torch.float32) if collect else None
        self._label = new((V,), torch.float32) if collect else None
        self._first = True
        self._fmt = {k: DeviceRowFormatter(sim.device) for k in ("LiDAR", "risk", "metadata")}
        self._X, self._y = [], []
        self.rows = 0

    def samples(self):
        """(X float32 [n,1,R], y float32 [n,1]) of the rows logged so far, on the device, in file order."""
        import torch

        if not self.collect:
            raise ValueError("this logger was not asked to collect samples")
        if not self._X:
            dev = self._tracked.device
            return (torch.empty((0, 1, self._R), dtype=torch.float32, device=dev),
                    torch.empty((0, 1), dtype=torch.float32, device=dev))
        return torch.cat(self._X), torch.cat(self._y)

    def _rows(self, phase, st):
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        rc = self._lib.auv_dataset_rows(self.sim._h, phase, self.every, self.refresh_every, 1 if self._first else 0,
                                        _lib.ROWS_LABEL[self.collect], p(self._tracked), p(self._row_of), p(self._due),
                                        p(self._lidar), p(self._risk), p(self._meta), p(self._risk_len), p(self._meta_len),
                                        p(self._x), p(self._label), C.c_void_p(st.cuda_stream))
        if rc != 0:
            raise _lib.AuvError(rc, self._lib.auv_last_error(self.sim._h).decode())

    def after_step(self, stream=None):
        import torch

        st = torch.cuda.current_stream(self._tracked.device) if stream is None else stream
        with torch.cuda.stream(st):
            self._rows(0, st)
            if int(self._due.item()) == 0:      # no env at a logging step: the lists stay as they are
                return 0
            self._rows(1, st)
            self._first = False
            n = int(self._row_of[self._V].item())
            if n == 0:
                return 0
            if self.collect:
                self._X.append(self._x[:n].reshape(n, 1, self._R).clone())
                self._y.append(self._label[:n, None].clone())
            if self.writer is not None:
                rows = {"LiDAR": (self._lidar[:n], None), "risk": (self._risk[:n], self._risk_len[:n]),
                        "metadata": (self._meta[:n], self._meta_len[:n])}
                tickets = {k: self._fmt[k].format(v, rl, st) for k, (v, rl) in rows.items()}
                tails = torch.stack([self._fmt[k].tail(tickets[k]) for k in rows]).cpu()    # one read-back for the three
                texts = {k: self._fmt[k].fetch(tickets[k], tails[i]) for i, k in enumerate(rows)}
        if self.writer is not None:
            self.writer.write_texts(texts)
            self.writer.rows += n
        self.rows += n
        return n


def save_binary(path, ranges, own, meta, risk, tracked):
    """Compact sink for large regenerations; `export_csv` turns it into the text files."""
    np.savez_compressed(path, ranges=ranges, own=own, meta=meta, risk=risk, tracked=tracked)


def export_csv(npz_path, directory, name="MovingObstaclesNoRules"):
    d = np.load(npz_path)
    with DatasetWriter(directory, name) as w:
        return w.write(d["ranges"], d["own"], d["meta"], d["risk"], d["tracked"])


# ----------------------------------------------------------------------------
# reading (what the reference's trainer does with the files)
# ----------------------------------------------------------------------------
def read_ragged(path):
    """List of float64 arrays, one per non-empty line."""
    with open(path) as f:
        return [np.array(line.split(), dtype=np.float64) for line in f if line.strip()]


def parse_metadata_row(row):
    """(speed_OS, heading_OS, records [n,5]) of one metadata row
    (utils/explore_dataset.py:40-45)."""
    n = (len(row) - 2) // 5
    return row[0], row[1], np.asarray(row[2:2 + 5 * n]).reshape(n, 5)


def is_static(record, speed_os):
    """The reference's static-obstacle test (utils/explore_dataset.py:46)."""
    return record[0] == 0 and record[2] == speed_os


def reduce_risks(table, mode):
    """Labels of a table of per-obstacle risks [n, <= 8 columns], NaN = no such
    obstacle: 'max', 'sum', else the mean over the obstacles a row holds
    (utils/dataloader.py:110-117).  The reference forms the sum with pandas,
    which fills the absent columns with zeros and adds the 8 columns of a row
    as pairs of pairs; the same order here, so that the float64 labels are the
    reference's bit for bit (adding in file order differs in the last place
    for about a seventh of the rows that hold 4 to 7 risks)."""
    table = np.asarray(table, dtype=np.float64)
    have = ~np.isnan(table)
    if mode == "max":
        return np.where(have, table, -np.inf).max(axis=1)
    f = np.zeros((len(table), 8))
    f[:, :table.shape[1]] = np.where(have, table, 0.0)
    total = ((f[:, 0] + f[:, 1]) + (f[:, 2] + f[:, 3])) + ((f[:, 4] + f[:, 5]) + (f[:, 6] + f[:, 7]))
    return total if mode == "sum" else total / have.sum(axis=1)


def risk_table(rows):
    """Ragged risk rows as the 8-column table, NaN-filled, that the reference's
    `read_csv(..., names=range(8))` makes of them (utils/dataloader.py:108)."""
    out = np.full((len(rows), 8), np.nan)
    for i, r in enumerate(rows):
        out[i, :min(len(r), 8)] = r[:8]
    return out


def calculate_total_risk(path_y, mode):
    """Label per sample from its per-obstacle risks: 'max', 'sum', else the
    mean, over at most 8 columns, absent columns skipped
    (utils/dataloader.py:104-120).  The text is read correctly rounded; the
    reference's pandas parser is up to one unit in the last place off on a
    quarter of the values, so its float64 labels can differ from these in the
    last place (none of the shipped rows' labels differs as float32, which is
    what the networks are given)."""
    return reduce_risks(risk_table(read_ragged(path_y)), mode)


def prev_timesteps_feature_engineering(X, time_steps):
    """Stack the k previous scans behind each scan, [n,R] -> [n,R,k+1]; history
    before the first row is an empty scan of 150 m (utils/dataloader.py:123-134)."""
    X = np.asarray(X, dtype=np.float64)
    out = [X]
    prev = X
    for _ in range(time_steps):
        prev = np.concatenate([np.full((1, X.shape[1]), SENSOR_RANGE), prev[:-1]], axis=0)
        out.append(prev)
    return np.stack(out, axis=2)


def split_sizes(n, train_test_split=0.7, train_val_split=0.2):
    """(train, val, test) sizes of the sequential split (utils/dataloader.py:63-67)."""
    train = int(train_test_split * n)
    val = int(train_val_split * train)
    return train - val, val, n - train


def _binary_rows(npz_path, mode):
    """(ranges, labels) of the rows `export_csv` would write from a `save_binary` file, without the text:
    samples with a tracked obstacle, the label over at most 8 of the tracked obstacles' risks in id order."""
    d = np.load(npz_path)
    tracked = d["tracked"].astype(bool)
    keep = tracked.any(axis=1)
    ranges, risk, tracked = d["ranges"][keep], d["risk"][keep], tracked[keep]
    order = np.argsort(~tracked, axis=1, kind="stable")
    risk_c = np.take_along_axis(risk, order, axis=1)
    cnt = tracked.sum(axis=1)
    table = np.full((len(ranges), 8), np.nan)
    w = min(8, risk_c.shape[1])
    table[:, :w] = np.where(np.arange(w)[None, :] < cnt[:, None], risk_c[:, :w], np.nan)
    return ranges, reduce_risks(table, mode)


def load_lidar_dataset(path_x, path_y=None, mode=None, train_test_split=0.7, train_val_split=0.2):
    """Arrays of the three splits as the reference builds them
    (utils/dataloader.py:44-101): X = 1 - X/150 with a channel dimension,
    float32; y reduced by `mode` (None: the file holds one label per row).
    `path_x` may instead be a `save_binary` file (.npz; `path_y` unused, `mode`
    required): the same arrays as from the text files `export_csv` makes of it."""
    if str(path_x).endswith(".npz"):
        if mode is None:
            raise ValueError("a binary dataset holds per-obstacle risks: give mode 'max', 'sum' or 'mean'")
        ranges, y = _binary_rows(path_x, mode)
        X = 1.0 - ranges / SENSOR_RANGE
    else:
        X = 1.0 - np.loadtxt(path_x) / SENSOR_RANGE
        y = np.loadtxt(path_y) if mode is None else calculate_total_risk(path_y, mode)
    n_train, n_val, n_test = split_sizes(len(X), train_test_split, train_val_split)
    sl = {"train": slice(0, n_train), "val": slice(n_train, n_train + n_val), "test": slice(len(X) - n_test, len(X))}
    out = {k: (X[s].astype(np.float32)[:, None, :], y[s].astype(np.float32)[:, None]) for k, s in sl.items()}
    out["x_mean"], out["x_std"] = X[:n_train].mean(), X[:n_train].std()
    return out
