"""The stability oracle (oracle/stability_oracle.py) against the reference's own outputs (g11_stability.npz:
flags, distance matrices, adjacency, 3-ring / 4-ring angle multisets for 523 seeded molecules)."""
import numpy as np
import pytest

from oracle import stability_oracle as S


def _mol(g, ds, m):
    k = int(g[f"{ds}_n"][m])
    return g[f"{ds}_x"][m, :k], g[f"{ds}_types"][m, :k]


@pytest.mark.parametrize("ds", ["cata", "hetro"])
def test_flags_distances_adjacency_and_angles(golden, ds):
    g = golden("g11_stability")
    M = len(g[f"{ds}_n"])
    seen = set()
    for m in range(M):
        x, ty = _mol(g, ds, m)
        res, aux = S.check_stability(x, ty, dataset=ds, return_aux=True)
        flags = [res[k] for k in S.FLAG_NAMES]
        assert flags == g[f"{ds}_flags"][m].astype(bool).tolist(), (m, res)
        seen.add(tuple(flags))
        if not res["orientation_nodes"]:
            continue
        nr = len(x) if ds == "cata" else len(x) // 2
        np.testing.assert_allclose(aux["dist"], g[f"{ds}_dist"][m, :nr, :nr], rtol=2e-7, atol=1e-7)
        assert np.array_equal(aux["adj"], g[f"{ds}_adj"][m, :nr, :nr])
        if res["connected"]:
            # angle multisets per centre ring type (np.sort puts NaN -- acos of 1+ulp on exactly straight triplets -- last)
            n3, n4 = g[f"{ds}_counts"][m]
            wt, wa = g[f"{ds}_a3_type"][m, :n3], g[f"{ds}_a3"][m, :n3]
            gt = np.array([t for t, _ in aux["angels3"]], np.int64)
            ga = np.array([a for _, a in aux["angels3"]], np.float32)
            assert len(gt) == n3 and len(aux["angels4"]) == n4, m
            for t in set(gt.tolist()) | set(wt.tolist()):
                np.testing.assert_allclose(np.sort(ga[gt == t]), np.sort(wa[wt == t]), rtol=0, atol=1e-3, equal_nan=True)
            np.testing.assert_allclose(np.sort(np.array(aux["angels4"], np.float32)), np.sort(g[f"{ds}_a4"][m, :n4]),
                                       rtol=0, atol=2e-2, equal_nan=True)
    # the fixture exercises every exit of check_stability
    assert len(seen) >= 5, seen


def test_aggregate_matches_reference_counts(golden):
    g = golden("g11_stability")
    for ds in ("cata", "hetro"):
        mols = [_mol(g, ds, m) for m in range(len(g[f"{ds}_n"]))]
        d, stable = S.analyze_validity_for_molecules(mols, dataset=ds)
        fl = g[f"{ds}_flags"].astype(bool)
        assert d["mol_stable"] == fl.all(1).mean()
        for i, k in enumerate(S.FLAG_NAMES):
            assert d[k] == fl[:, i].mean()
        assert len(stable) == fl.all(1).sum()


def test_bfs_tree_follows_networkx_order():
    """bfs_edges: FIFO from node 0, neighbours ascending (what nx.bfs_edges yields for from_numpy_array graphs)."""
    adj = np.zeros((6, 6))
    for a, b in [(0, 3), (0, 1), (1, 2), (3, 2), (2, 4), (3, 5), (4, 5)]:
        adj[a, b] = adj[b, a] = 1
    edges, connected = S.bfs_edges(adj)
    assert edges == [(0, 1), (0, 3), (1, 2), (3, 5), (2, 4)] and connected
    adj[2, 4] = adj[4, 2] = adj[4, 5] = adj[5, 4] = 0
    assert S.bfs_edges(adj)[1] is False
