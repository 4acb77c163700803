"""Load the golden fixtures (tests/golden/*.npz) into oracle descriptors."""
import json
import os

import numpy as np

from oracle import ttsk_oracle as orc

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Cases:
    def __init__(self):
        self.z = np.load(os.path.join(GOLDEN, "sketch_cases.npz"))
        self.meta = json.loads(str(self.z["meta"]))
        self.keys = set(self.z.files)

    def names(self):
        return list(self.meta)

    def lst(self, prefix):
        out = []
        while f"{prefix}/{len(out)}" in self.keys:
            out.append(self.z[f"{prefix}/{len(out)}"])
        return out

    def tensor(self, name):
        """-> (kind, data) in the oracle's plain-array convention."""
        m = self.meta[name]
        items = []
        for i, kind in enumerate(m["kinds"]):
            p = f"{name}/tensor{i}"
            if kind in ("tt", "cp"):
                data = self.lst(p + "/cores")
            elif kind == "tucker":
                data = (self.lst(p + "/factors"), self.z[p + "/core"])
            elif kind == "dense":
                data = self.z[p + "/data"]
            elif kind == "sparse":
                data = (tuple(m["shape"]), self.z[p + "/indices"], self.z[p + "/entries"])
            items.append((kind, data))
        if len(items) == 1:
            return items[0]
        return ("sum", items)

    def drm(self, name, side):
        m = self.meta[name]
        p = f"{name}/{side}_drm"
        kind = m[f"{side}_drm"]
        shape = tuple(m["shape"])
        tr = bool(self.z[p + "/transpose"])
        rmin = tuple(int(x) for x in self.z[p + "/rank_min"])
        rmax = tuple(int(x) for x in self.z[p + "/rank_max"])
        true = tuple(int(x) for x in self.z[p + "/true_rank"])
        seed = int(self.z[p + "/seed"])
        if kind == "tt":
            return orc.TTDrm(self.lst(p + "/cores"), shape, tr, rmin, rmax)
        if kind == "dense":
            return orc.DenseDrm(self.lst(p + "/mats"), shape, tr)
        if kind == "hashgauss":
            return orc.HashGaussDrm(seed, shape, tr, rmin, rmax)
        if kind == "hashsign":
            nnz = tuple(int(x) for x in self.z[p + "/nnz"])
            return orc.HashSignDrm(seed, shape, tr, true, rmin, rmax, nnz)
        raise ValueError(kind)

    def out(self, name, what):
        return self.lst(f"{name}/out/{what}")


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.shape != b.shape:
        return np.inf
    nb = np.linalg.norm(b)
    return np.linalg.norm(a - b) / (nb if nb > 0 else 1.0)
