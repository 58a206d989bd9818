"""CPU checks of the oracle's permutation test with unknowns (oracle/oracle.c orc_mica_permutation_test_masks): its
integer joint table against the independently written MI restatement (orc_mi_columns, SiteTools::jointEntropy with
resolveUnknowns = true, CoMap/Mica.cpp:93-95), and the properties the test must have whatever the shuffles are."""
import ctypes

import numpy as np
import pytest

import oracle


def _joint_table(A, masks, ci, cj):
    L = oracle.lib()
    V = ctypes.c_void_p
    L.orc_mica_joint_table.restype = ctypes.c_long
    L.orc_mica_joint_table.argtypes = [ctypes.c_int, V, ctypes.c_int, ctypes.c_int, V, V, V]
    tab = np.zeros((A, A), dtype=np.int64)
    ci, cj = np.ascontiguousarray(ci, dtype=np.uint8), np.ascontiguousarray(cj, dtype=np.uint8)
    lc = L.orc_mica_joint_table(A, None if masks is None else masks.ctypes.data, 0 if masks is None else len(masks), len(ci),
                                ci.ctypes.data, cj.ctypes.data, tab.ctypes.data)
    assert lc > 0
    return lc, tab


@pytest.mark.parametrize("A,ncodes", [(4, 12), (20, 4)])
def test_joint_table_gives_the_joint_entropy_of_the_mi_restatement(A, ncodes):
    rng = np.random.default_rng(A)
    T = 50
    masks = oracle.default_masks(A)[:A + ncodes].copy()
    for c in range(A, A + ncodes - 1):
        masks[c] = sum(1 << int(x) for x in rng.choice(A, size=int(rng.integers(2, 4)), replace=False))
    full = np.concatenate([masks, np.full(256 - len(masks), (1 << A) - 1, dtype=np.uint32)])
    for _ in range(5):
        aln = rng.integers(0, A, size=(T, 2)).astype(np.uint8)
        hit = rng.random(aln.shape) < 0.3
        aln[hit] = rng.integers(A, A + ncodes + 2, size=int(hit.sum()))     # codes past the table: unknowns
        lc, tab = _joint_table(A, masks, aln[:, 0], aln[:, 1])
        M = lc * lc * T
        assert tab.sum() == M                                               # every position carries weight L^2
        nz = tab[tab > 0].astype(np.float64)
        hj = np.log(M) - (nz * np.log(nz)).sum() / M
        ref = oracle.mi_columns(aln[:, :1], aln[:, 1:], A, masks=full)
        assert abs(hj - ref["hjoint"][0, 0]) < 1e-12


def test_permutation_test_with_unknowns_properties():
    rng = np.random.default_rng(8)
    A, T, n = 4, 60, 8
    aln = rng.integers(0, A, size=(T, n)).astype(np.uint8)
    aln[:, 1] = aln[:, 0]                                    # perfectly coupled ...
    gaps = rng.random(T) < 0.15
    aln[gaps, 1] = 4                                         # ... with gaps in one of the two
    aln[:, 6] = np.where(rng.random(T) < 0.5, 2, 4)          # constant besides gaps
    aln[:, 7] = 4                                            # gaps only
    pv, npm = oracle.mica_permutation_test(aln, A, 400, 5)
    iu = np.triu_indices(n, 1)
    const = np.isin(iu[0], (6, 7)) | np.isin(iu[1], (6, 7))
    assert np.all(pv[const] == 1.0) and np.all(npm[const] == 0)
    assert pv[0] == 1.0 / 401 and npm[0] == 400              # pair (0, 1): no shuffle reaches the observed MI
    # sharding the pairs changes nothing; a mask table that says the same as the default changes nothing
    p2, n2 = oracle.mica_permutation_test(aln, A, 400, 5, 5, 17)
    assert np.array_equal(p2, pv[5:17]) and np.array_equal(n2, npm[5:17])
    p3, n3 = oracle.mica_permutation_test(aln, A, 400, 5, masks=oracle.default_masks(A)[:6])
    assert np.array_equal(p3, pv) and np.array_equal(n3, npm)
    # resolved pairs do not notice unknowns elsewhere in the alignment
    clean = aln.copy()
    clean[:, 1] = aln[:, 0]
    pc, nc = oracle.mica_permutation_test(clean[:, :6], A, 400, 5)
    pg, ng = oracle.mica_permutation_test(aln[:, :6], A, 400, 5)
    keep = ~(np.isin(np.triu_indices(6, 1)[0], (1,)) | np.isin(np.triu_indices(6, 1)[1], (1,)))
    assert np.array_equal(pc[keep], pg[keep]) and np.array_equal(nc[keep], ng[keep])
    with pytest.raises(ValueError):
        bad = oracle.default_masks(A)[:8].copy()
        bad[5] = 0
        oracle.mica_permutation_test(aln, A, 10, 1, masks=bad)
