"""GPU parity of the graph-of-rings stability kernel (gaudi_check_stability) through the C ABI / gaudi_amd.analyze:
the reference's own flags, distances and adjacency (g11_stability.npz, 523 molecules), then the numpy oracle on fresh
seeded molecules and on molecules produced by the sampler.  Flags, adjacency and triplet counts are compared exactly,
distances bit for bit against the oracle (1 ulp against torch's sqrt), angle ranges to 1e-3 / 2e-2 degrees (acos / atan2 implementations differ by ulps)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mols(g, ds):
    return [(g[f"{ds}_x"][m, :k], g[f"{ds}_types"][m, :k]) for m, k in enumerate(g[f"{ds}_n"])]


@pytest.mark.parametrize("ds", ["cata", "hetro"])
def test_against_reference_outputs(golden, ds):
    from gaudi_amd import analyze
    g = golden("g11_stability")
    mols = _mols(g, ds)
    X, T, nn = analyze._pack(mols)
    flags, dist, adj, aux = analyze.check_stability_batch(X, T, nn, 0.1, ds, want_adj=True, want_aux=True)
    want = g[f"{ds}_flags"].astype(bool)
    assert np.array_equal(flags, want), np.argwhere(flags != want)[:10]
    ok = want[:, 0]
    NM = g[f"{ds}_dist"].shape[1]
    # distances: correctly rounded fp32 here (bit-identical to the oracle, next test); torch's vectorised CPU sqrt is off by
    # one ulp on ~1 % of near-tie inputs (e.g. sqrt(17.163665771484375f)), hence 1 ulp against the reference's numbers
    np.testing.assert_allclose(dist[ok][:, :NM, :NM], g[f"{ds}_dist"][ok], rtol=1.2e-7, atol=0)
    assert np.array_equal(adj[ok][:, :NM, :NM], g[f"{ds}_adj"][ok])
    conn = want[:, 2]
    assert np.array_equal(aux["n_triplets"][conn], g[f"{ds}_counts"][conn, 0])
    for m in np.nonzero(conn)[0]:
        n3, n4 = g[f"{ds}_counts"][m]
        if n3:
            a3 = g[f"{ds}_a3"][m, :n3]
            assert aux["n_nan_angles"][m] == np.isnan(a3).sum()
            if not np.isnan(a3).all():
                assert abs(aux["a3_min"][m] - np.nanmin(a3)) < 1e-3 and abs(aux["a3_max"][m] - np.nanmax(a3)) < 1e-3
        a4 = g[f"{ds}_a4"][m, :n4]
        if n4 and not np.isnan(a4).all():
            assert abs(aux["a4_min"][m] - np.nanmin(a4)) < 2e-2 and abs(aux["a4_max"][m] - np.nanmax(a4)) < 2e-2
    # reference-shaped entry points
    d, stable = analyze.analyze_validity_for_molecules(mols, dataset=ds)
    assert d["mol_stable"] == want.all(1).mean() and len(stable) == want.all(1).sum()
    for i, k in enumerate(analyze.FLAG_NAMES):
        assert d[k] == want[:, i].mean()
    assert d["molecule_stable_bool"] == want.all(1).tolist()
    for m in (0, 5, len(mols) - 1):
        res = analyze.check_stability(*mols[m], dataset=ds)
        assert [res[k] for k in analyze.FLAG_NAMES] == want[m].tolist()


def test_one_hot_ring_types_and_torch_inputs(golden):
    import torch
    from gaudi_amd import analyze
    g = golden("g11_stability")
    mols = _mols(g, "hetro")[:40]
    oh = [(torch.from_numpy(x), torch.nn.functional.one_hot(torch.from_numpy(t), 12).float()) for x, t in mols]
    d, _ = analyze.analyze_validity_for_molecules(oh, dataset="hetro")
    assert d["molecule_stable_bool"] == g["hetro_flags"][:40].astype(bool).all(1).tolist()


@pytest.mark.parametrize("ds", ["cata", "hetro"])
def test_random_molecules_vs_oracle(ds):
    """Fresh seeded molecules (random walks with bonded-length steps, 1..32 rings) against the numpy oracle."""
    from oracle import stability_oracle as S
    from gaudi_amd import analyze
    rng = np.random.default_rng(77 if ds == "cata" else 78)
    T = S.tables()
    R = len(T["rings"][ds]) - (1 if ds == "hetro" else 0)
    mols = []
    for i in range(300):
        n = int(rng.integers(1, 33 if i % 10 == 0 else 13))
        ty = rng.integers(0, R, n)
        pos = [np.zeros(3)]
        for k in range(1, n):
            p = int(rng.integers(k))
            d = rng.standard_normal(3)
            d[2] *= 0.15
            pos.append(pos[p] + d / np.linalg.norm(d) * rng.uniform(1.9, 2.7))
        x = np.array(pos, np.float32)
        if ds == "hetro":
            x = np.concatenate([x, x + 0.3], 0)
            ty = np.concatenate([ty, np.full(n, R)])
        mols.append((x, ty.astype(np.int64)))
    X, Ty, nn = analyze._pack(mols)
    flags, dist, adj, aux = analyze.check_stability_batch(X, Ty, nn, 0.1, ds, want_adj=True, want_aux=True)
    seen = set()
    for m, (x, ty) in enumerate(mols):
        res, a = S.check_stability(x, ty, dataset=ds, return_aux=True)
        want = [res[k] for k in S.FLAG_NAMES]
        assert flags[m].tolist() == want, (m, res, flags[m])
        seen.add(tuple(want))
        nr = len(a["dist"])
        assert np.array_equal(dist[m, :nr, :nr], a["dist"]) and np.array_equal(adj[m, :nr, :nr], a["adj"])
        if res["connected"]:
            assert aux["n_triplets"][m] == len(a["angels3"])
    assert len(seen) >= 3


def test_tolerance_argument_and_positions2adj(golden):
    from oracle import stability_oracle as S
    from gaudi_amd import analyze
    g = golden("g11_stability")
    mols = _mols(g, "cata")[:60]
    for tol in (0.0, 0.05, 0.3):
        X, T, nn = analyze._pack(mols)
        flags = analyze.check_stability_batch(X, T, nn, tol, "cata")
        for m, (x, ty) in enumerate(mols):
            res = S.check_stability(x, ty, tol=tol, dataset="cata")
            assert flags[m].tolist() == [res[k] for k in S.FLAG_NAMES], (tol, m)
    x = g["hetro_x"][:8, :6]
    ty = np.clip(g["hetro_types"][:8, :6], 0, 10)
    dist, adj = analyze.positions2adj(x, ty, dataset="hetro")
    for b in range(8):
        d0, a0 = S.positions2adj(x[b], ty[b], 0.1, "hetro")
        assert np.array_equal(dist[b], d0) and np.array_equal(adj[b], a0)


def test_error_paths():
    from gaudi_amd import analyze
    from gaudi_amd._lib import GaudiError
    x = np.zeros((1, 3), np.float32)
    with pytest.raises(GaudiError, match="ring type"):
        analyze.check_stability(x, np.array([3]), dataset="cata")
    with pytest.raises(GaudiError, match="null graph"):
        analyze.check_stability(x, np.array([11]), dataset="hetro")   # one orientation node, no ring
    with pytest.raises(GaudiError, match="32 rings"):
        analyze.check_stability(np.zeros((33, 3), np.float32), np.zeros(33, np.int64), dataset="cata")
    with pytest.raises(GaudiError, match="empty"):
        analyze.analyze_validity_for_molecules([], dataset="cata")
    assert analyze.check_stability(x, np.array([0]), dataset="cata") == dict(
        orientation_nodes=True, dist_stable=True, connected=True, angels3=True, angels4=True)


def test_sampler_output_feeds_the_check():
    """End of the eval_validity path: sample_pos_edm -> compaction by node_mask -> stability kernel, equal to the
    oracle on the same molecules."""
    import types
    from oracle import stability_oracle as S
    from gaudi_amd import analyze, sampling_edm, synth
    from gaudi_amd.models_edm import get_model
    eargs = synth.edm_args(diffusion_steps=20, nf=64, n_layers=2)
    model, _, _ = get_model(eargs, state_dict=synth.synth_edm_state_dict(eargs, 1, seed=3))
    args = types.SimpleNamespace(device="cuda", dataset="cata", max_nodes=11)
    x, one_hot, nm, em = sampling_edm.sample_pos_edm(args, model, [4, 11, 7, 9, 2, 11], std=0.7)
    mols = [(x[i][nm[i, :, 0].bool()], one_hot[i][nm[i, :, 0].bool()].argmax(1)) for i in range(x.shape[0])]
    d, stable = analyze.analyze_validity_for_molecules(mols, dataset="cata", engine=model.engine)
    want = [all(S.check_stability(p.numpy(), t.numpy(), dataset="cata").values()) for p, t in mols]
    assert d["molecule_stable_bool"] == want
    model.engine.close()


def test_eval_validity_analyze_and_save():
    """eval_validity.analyze_and_save: DistributionRings -> sample_pos_edm -> stability kernel; the drawn ring counts follow
    the seeded torch Categorical stream, the stability dict equals the oracle's on the same molecules."""
    import types
    import torch
    from oracle import stability_oracle as S
    from gaudi_amd import eval_validity, synth
    from gaudi_amd.models_edm import DistributionRings, get_model
    eargs = synth.edm_args(diffusion_steps=10, nf=32, n_layers=1)
    model, nodes_dist, _ = get_model(eargs, state_dict=synth.synth_edm_state_dict(eargs, 1, seed=5))
    assert isinstance(nodes_dist, DistributionRings)
    args = types.SimpleNamespace(device="cuda", dataset="cata", max_nodes=11, batch_size=16, exp_dir="synthetic")
    torch.manual_seed(3)
    d, mols, stable = eval_validity.analyze_and_save(args, model, nodes_dist, n_samples=20)
    assert len(mols) == 32  # rounded up to whole batches (eval_validity.py:29)
    torch.manual_seed(3)  # the torch stream of the run above: ring counts, one noise-key draw per sampling call, ring counts
    first = nodes_dist.sample(16)
    torch.randint(0, 2 ** 62, (1,), dtype=torch.int64)
    want_n = torch.cat([first, nodes_dist.sample(16)]).tolist()
    assert [len(x) for x, _ in mols] == want_n
    od, ostable = S.analyze_validity_for_molecules([(x.numpy(), t.numpy()) for x, t in mols], dataset="cata")
    assert d == od and len(stable) == len(ostable)
    model.engine.close()
