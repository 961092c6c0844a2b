"""GPU parity against the committed golden fixtures (tests/golden/): realistic Newton
systems captured from the restated Class 1 driver on the reference's bundled m=n=500
data, plus small synthetic cases.  No /root/reference and no oracle run is needed:
inputs and expected outputs come from the .npz files."""
import glob
import hashlib
import os

import numpy as np
import pytest
import scipy.sparse as sp

from tests.test_golden_oracle import GOLD, load, problem_from

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ipd():
    import codes_of_ipd_ssn_amg_method_amd as m
    return m


def digest(M):
    M = sp.csc_matrix(M)
    M.sort_indices()
    h = hashlib.sha256()
    for a in (np.asarray(M.shape, np.int64), M.indptr.astype(np.int64), M.indices.astype(np.int64),
              M.data.astype(np.float64)):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def _opts(cycle, pot=False):
    return dict(retol=1e-11, bigph=1, maxit=40 if pot else 30, theta=0.25, smoth=10 if pot else 5,
                cycle=cycle, isnsp=1, inter=1, guess=None, fnode=None)


def _check_solution(z, it, res, info, g, rng):
    assert np.array_equal(info, g["info"])
    noise = res <= 1e-10 and float(g["res"]) <= 1e-10
    assert abs(it - int(g["it"])) <= 1 or noise
    assert np.linalg.norm(z - g["zeta"]) <= 1e-5 * max(1.0, np.linalg.norm(g["zeta"]))
    assert rng.consumed == int(g["rand_consumed"])


@pytest.mark.parametrize("name", sorted(os.path.basename(f) for f in glob.glob(os.path.join(GOLD, "synth_*.npz"))))
def test_synthetic_golden(ipd, name):
    g = load(name)
    pd = problem_from(g)
    m, n = pd["m"], pd["n"]
    H0 = ipd.ASAt(pd["s"], pd["p"], pd["q"])
    assert np.array_equal(H0.indptr, g["H0_indptr"]) and np.array_equal(H0.indices, g["H0_indices"])
    assert np.array_equal(H0.data, g["H0_data"])                         # bit-exact
    assert np.array_equal(ipd.Aty(np.arange(m + n) * 0.1, pd["p"], pd["q"]), g["Aty"])
    assert np.allclose(ipd.Ax(np.arange(m * n) * 0.01, pd["p"], pd["q"]), g["Ax"], rtol=1e-13)
    pd["H0"] = H0
    pot = bool(int(g["pot"]))
    rng = ipd.MatlabRand()
    if pot:
        z, it, res, info = ipd.AMG4POT(pd, _opts(str(g["cycle"]), True), "amg", rng)
    else:
        z, it, res, info = ipd.Hybrid_AMG(pd, _opts(str(g["cycle"])), rng)
    _check_solution(z, it, res, info, g, rng)
    if int(g["ncalls"]) == 1 and not pot and int(g["info"][0]) == 1:
        # connected case: the whole hierarchy must equal the stored one bit for bit
        Ae = sp.csc_matrix((g["Ae_data"], g["Ae_indices"], g["Ae_indptr"]), shape=(m + n, m + n))
        o = _opts(str(g["cycle"]))
        o.update(fnode=int(g["c0_fnode"]), isnsp=int(g["c0_isnsp"]))
        rng2 = ipd.MatlabRand()
        rng2.rand(m + n)                                                  # the guess draw
        h = ipd.AMGHierarchy(Ae, o, rng2)
        assert h.level_sizes() == list(g["c0_levels"])
        for k in range(2, h.J + 1):
            assert digest(h.A(k)) == str(g["c0_Ahash%d" % k])
            assert digest(h.P(k)) == str(g["c0_Phash%d" % k])
            assert np.array_equal(np.packbits(h.cmask(k)), g["c0_cmask%d" % k])
        h.close()


@pytest.mark.parametrize("k", [1, 3, 8, 20, 40])
def test_realistic_golden(ipd, k):
    g = load("class1_500_k%02d.npz" % k)
    pd = problem_from(g)
    H0 = ipd.ASAt(pd["s"], pd["p"], pd["q"])
    assert H0.nnz == int(g["H0_nnz"]) and digest(H0) == str(g["H0_hash"])   # bit-exact
    pd["H0"] = H0
    rng = ipd.MatlabRand()
    z, it, res, info = ipd.Hybrid_AMG(pd, _opts("w"), rng)
    _check_solution(z, it, res, info, g, rng)
    M = pd["m"] + pd["n"]
    He = pd["bk1"] * sp.identity(M) + H0 / pd["tk"]
    assert np.linalg.norm(He @ z - pd["z"]) <= 1e-8 * np.linalg.norm(pd["z"])
