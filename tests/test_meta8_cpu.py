"""Host logic of the 8-wave kernels' graph metadata (gaudi_host_graph_meta8): invariants the device code relies on."""
import ctypes as C

import numpy as np

from gaudi_amd import _lib
from oracle import gaudi_oracle as O


def meta8(nm, em):
    lib = _lib.load_library()
    B, N = nm.shape[0], nm.shape[1]
    nm = np.ascontiguousarray(nm.reshape(B, N), np.float32)
    em = np.ascontiguousarray(em.reshape(B, N, N), np.float32)
    slots = C.c_int32()
    i32, u32, u16 = C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint16)
    assert lib.gaudi_host_graph_meta8(B, N, _lib.fptr(nm), _lib.fptr(em), C.byref(slots), None, None, None, None, None, None, None, 0,
                                      None) == 0
    S = slots.value
    order, ntiles, ncols = (np.zeros(B, np.int32) for _ in range(3))
    seg = np.zeros((B, N), np.uint32)
    edges = np.zeros((B, S), np.uint32)
    emask = np.zeros((B, S), np.float32)
    soff = np.zeros((B, N + 1), np.uint16)
    sidx = np.zeros((B, S), np.uint16)
    rc = lib.gaudi_host_graph_meta8(B, N, _lib.fptr(nm), _lib.fptr(em), C.byref(slots), order.ctypes.data_as(i32),
                                    ntiles.ctypes.data_as(i32), seg.ctypes.data_as(u32), edges.ctypes.data_as(u32), _lib.fptr(emask),
                                    soff.ctypes.data_as(u16), sidx.ctypes.data_as(u16), B * S, ncols.ctypes.data_as(i32))
    assert rc == 0
    return dict(S=S, order=order, ntiles=ntiles, ncols=ncols, seg=seg, edges=edges, emask=emask, soff=soff, sidx=sidx)


def check(nm, em):
    B, N = nm.shape[0], nm.shape[1]
    nm2, em3 = nm.reshape(B, N), em.reshape(B, N, N)
    M = meta8(nm, em)
    assert M["S"] % 16 == 0 and sorted(M["order"].tolist()) == list(range(B))
    assert all(M["ntiles"][M["order"][k]] >= M["ntiles"][M["order"][k + 1]] for k in range(B - 1))  # heaviest first
    for b in range(B):
        e, m = M["edges"][b], M["emask"][b]
        i, j = e & 255, (e >> 8) & 255
        rs, rend, part = (e >> 16) & 15, (e >> 20) & 1, (e >> 21) & 1
        ns = 16 * M["ntiles"][b]
        live = [(int(i[s]), int(j[s])) for s in range(ns) if m[s] != 0]
        want = [(a, c) for a in range(N) for c in range(N)
                if em3[b, a, c] != 0 and (nm2[b, a] != 0 or nm2[b, c] != 0)]
        assert live == want  # every live edge once, sorted by receiving then sending node
        assert np.all(m[ns:] == 0)
        for n in range(N):
            st, ln = int(M["seg"][b, n] >> 16), int(M["seg"][b, n] & 0xffff)
            assert ln == sum(1 for a, _ in want if a == n)
            assert all(i[st + k] == n and m[st + k] != 0 for k in range(ln))
            assert ln == 0 or (st + ln - 1) // 16 - st // 16 <= 1  # a run touches at most two tiles
            senders = M["sidx"][b, M["soff"][b, n]:M["soff"][b, n + 1]].tolist()
            assert senders == [s for s in range(ns) if m[s] != 0 and j[s] == n]
        for t0 in range(0, ns, 16):
            c = 0
            while c < 16:  # segments of equal receiving node: run_start / run_end / partial flags
                e2 = c
                while e2 + 1 < 16 and i[t0 + e2 + 1] == i[t0 + c]:
                    e2 += 1
                assert all(rs[t0 + k] == c for k in range(c, e2 + 1))
                assert [int(rend[t0 + k]) for k in range(c, e2 + 1)] == [0] * (e2 - c) + [1]
                node = int(i[t0 + c])
                st, ln = int(M["seg"][b, node] >> 16), int(M["seg"][b, node] & 0xffff)
                assert all(int(part[t0 + k]) == (1 if ln > 0 and st < t0 else 0) for k in range(c, e2 + 1))
                c = e2 + 1
        last = max([n for n in range(N) if nm2[b, n] != 0] + [max(a, c) for a, c in want] + [0])
        assert M["ncols"][b] == last + 1


def test_cata_and_hetero_batches():
    check(*O.build_masks([11, 4, 7, 1, 11], 11, False))
    check(*O.build_masks([3, 10, 6, 2, 9], 10, True))


def test_random_masks_including_long_runs_and_empty_molecules():
    rng = np.random.default_rng(0)
    for N in (5, 17, 20, 33):
        B = 6
        nm = (rng.random((B, N)) < 0.8).astype(np.float32)
        nm[0] = 0  # a molecule without live nodes
        em = (rng.random((B, N, N)) < 0.7).astype(np.float32) * rng.uniform(0.5, 2.0, (B, N, N)).astype(np.float32)
        em *= 1 - np.eye(N, dtype=np.float32)
        em[1] = 0  # a molecule without edges
        check(nm[:, :, None], em)


def test_pack_plan_groups_small_molecules():
    """gaudi_host_pack_plan: every molecule lands in exactly one group; a group holds at most 4 molecules, N node slots and 8
    edge tiles; big molecules stay alone; the groups' tile counts are the sums of their members' own tile counts (component
    starts are tile-aligned, so a molecule keeps the tiles it has on its own)."""
    import ctypes as C

    from gaudi_amd import _lib
    from gaudi_amd.sampling_edm import build_masks
    lib = _lib.load_library()
    i32 = C.POINTER(C.c_int32)
    rings = np.random.default_rng(1).integers(3, 11, size=64)
    nm3, em_flat, N = build_masks(rings, 10, True)
    B = len(rings)
    nm, em = np.ascontiguousarray(nm3.reshape(B, N)), np.ascontiguousarray(em_flat.reshape(B, N, N))
    G = C.c_int32()
    group_of, ntiles, ncols = np.full(B, -1, np.int32), np.zeros(B, np.int32), np.zeros(B, np.int32)
    assert lib.gaudi_host_pack_plan(B, N, _lib.fptr(nm), _lib.fptr(em), C.byref(G), group_of.ctypes.data_as(i32),
                                    ntiles.ctypes.data_as(i32), ncols.ctypes.data_as(i32)) == 0
    G = G.value
    assert 0 < G < B and (group_of >= 0).all() and group_of.max() == G - 1
    slots = C.c_int32()
    own_tiles = np.zeros(B, np.int32)
    assert lib.gaudi_host_graph_meta8(B, N, _lib.fptr(nm), _lib.fptr(em), C.byref(slots), None, own_tiles.ctypes.data_as(i32),
                                      None, None, None, None, None, 0, None) == 0
    live = (nm != 0).sum(1)
    for g in range(G):
        members = np.nonzero(group_of == g)[0]
        assert 1 <= len(members) <= 4
        assert ntiles[g] == own_tiles[members].sum() <= 8
        assert live[members].sum() <= N and ncols[g] <= N
    # cata 11-ring molecules (the C3 shape) fill N = 11 on their own: nothing packs
    nm3, em_flat, N = build_masks([11] * 8, 11, False)
    nm, em = np.ascontiguousarray(nm3.reshape(8, N)), np.ascontiguousarray(em_flat.reshape(8, N, N))
    G = C.c_int32()
    assert lib.gaudi_host_pack_plan(8, N, _lib.fptr(nm), _lib.fptr(em), C.byref(G), None, None, None) == 0 and G.value == 8


def test_wide_group_plan_pairs_cata_molecules():
    """gaudi_host_pack_plan_wide (device-free): with 2 N node slots and 16 edge tiles per group, 11-ring cata molecules (11
    nodes, 110 live edges = 7 tiles) pair up; a ragged batch packs by node slots and tiles, at most 4 molecules per group."""
    import ctypes as C
    from gaudi_amd import _lib
    from gaudi_amd.sampling_edm import build_masks
    lib = _lib.load_library()
    i32 = C.POINTER(C.c_int32)

    def plan(sizes, orientation, slots, tiles):
        nm3, em_flat, N = build_masks(sizes, max(sizes), orientation)
        B = len(sizes)
        nm, em = np.ascontiguousarray(nm3.reshape(B, N)), np.ascontiguousarray(em_flat.reshape(B, N, N))
        G = C.c_int32()
        group_of, ntiles, ncols = (np.zeros(B, np.int32) for _ in range(3))
        rc = lib.gaudi_host_pack_plan_wide(B, N, slots if slots else 2 * N, tiles, _lib.fptr(nm), _lib.fptr(em), C.byref(G),
                                           group_of.ctypes.data_as(i32), ntiles.ctypes.data_as(i32), ncols.ctypes.data_as(i32))
        assert rc == 0
        return G.value, group_of, ntiles[:G.value], ncols[:G.value], N

    G, group_of, ntiles, ncols, N = plan([11] * 6, False, 0, 16)
    assert G == 3 and sorted(np.bincount(group_of)) == [2, 2, 2] and list(ntiles) == [14] * 3 and list(ncols) == [22] * 3
    G, group_of, ntiles, ncols, N = plan([3, 10, 4, 3, 5, 7, 3, 6, 4, 9, 3, 4, 8, 5, 3, 3], True, 22, 16)
    assert G < 16 and np.bincount(group_of).max() <= 4 and ntiles.max() <= 16 and ncols.max() <= 22
    # node_slots < N is refused
    nm = np.ones((2, 5), np.float32)
    em = np.ascontiguousarray(np.broadcast_to(1.0 - np.eye(5, dtype=np.float32), (2, 5, 5)))
    Gc = C.c_int32()
    assert lib.gaudi_host_pack_plan_wide(2, 5, 4, 16, _lib.fptr(nm), _lib.fptr(em), C.byref(Gc), None, None, None) != 0
