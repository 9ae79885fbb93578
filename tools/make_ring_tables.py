#!/usr/bin/env python3
"""Dump the graph-of-rings geometry tables of the REFERENCE into gaudi_amd/data/ring_tables.json.

Runs only in the build container.  The tables are data measured on the reference's datasets (quantiles of ring-ring
distances, 3-ring angles, 4-ring dihedrals; utils/helpers.py:11-196, data/aromatic_dataloader.py:31-35, data/ring.py:6-18)
and are needed verbatim for parity of the stability check; they are exported in array form, indexed by the ring-type
index the one-hot node features use:

    rings[dataset]            ring symbols, index = class index of the one-hot features
    dist_lo/hi[dataset][i][j] bonded-distance window of ring types (i, j), 0/0 when the pair never bonds
    a3[dataset][i]            list of (low, high) windows for the 3-ring angle centred on ring type i
    a4[dataset]               {"0": q, "180": q} dihedral thresholds;   n_nodes[dataset]: ring-count histogram
"""
import json
import os
import sys
from unittest.mock import MagicMock

sys.dont_write_bytecode = True
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, "/root/reference")
for m in ["rdkit", "rdkit.Chem", "rdkit.Chem.Draw", "rdkit.Chem.rdmolops", "rdkit.Chem.rdchem", "rdkit.Chem.AllChem",
          "imageio", "torch.utils.tensorboard"]:
    sys.modules[m] = MagicMock()

from data.aromatic_dataloader import RINGS_LIST  # noqa: E402
from utils import helpers as H  # noqa: E402

out = dict(rings={}, dist_lo={}, dist_hi={}, a3={}, a4={}, n_nodes={}, min_dist={})
for ds in ("cata", "hetro"):
    rings = list(RINGS_LIST[ds])
    R = len(rings)
    lo = [[0.0] * R for _ in range(R)]
    hi = [[0.0] * R for _ in range(R)]
    for i, si in enumerate(rings):
        for j, sj in enumerate(rings):
            # the lookup order of positions2adj (utils/helpers.py:178-182): "si-sj" first, then "sj-si"
            key = f"{si}-{sj}"
            if key not in H.ring_distances[ds]:
                key = f"{sj}-{si}"
            if key in H.ring_distances[ds]:
                lo[i][j], hi[i][j] = H.ring_distances[ds][key]
    out["rings"][ds] = rings
    out["dist_lo"][ds], out["dist_hi"][ds] = lo, hi
    out["min_dist"][ds] = min(r[0] for r in H.ring_distances[ds].values())
    out["a3"][ds] = [[list(w) for w in H.angels3_dict[ds][s].values()] if s in H.angels3_dict[ds] else [] for s in rings]
    out["a4"][ds] = H.angels4_dict[ds]
    out["n_nodes"][ds] = {str(k): v for k, v in H.analyzed_rings[ds]["n_nodes"].items()}
path = os.path.join(ROOT, "gaudi_amd", "data", "ring_tables.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print(path, os.path.getsize(path), "bytes")
