"""Host-side logic that needs no GPU: the MAT-v5 input reader of the drivers' workspaces."""
import os

import numpy as np
import pytest
import scipy.io


def pkg():
    try:
        import codes_of_ipd_ssn_amg_method_amd as ipd
    except ImportError as exc:      # the shared library is built by __graft_entry__.build()
        pytest.skip(str(exc))
    return ipd


def test_load_input_casts_and_shapes(tmp_path):
    ipd = pkg()
    m, n = 6, 4
    rs = np.random.RandomState(0)
    C = rs.random_sample((m, n))
    f1 = str(tmp_path / "data1.mat")       # Class 1 layout: integer-typed p, q, m, n; gama = Inf
    scipy.io.savemat(f1, dict(c=C.reshape(-1, 1, order="F"), r=rs.random_sample((n, 1)),
                              l=rs.random_sample((m, 1)), p=np.ones((m, 1), np.uint8),
                              q=np.ones((n, 1), np.uint8), gama=np.full((m * n, 1), np.inf),
                              m=np.uint16(m), n=np.uint16(n)))
    d = ipd.load_input(f1)
    assert d["cls"] == 1 and (d["m"], d["n"]) == (m, n)
    assert d["p"].dtype == np.float64 and d["q"].dtype == np.float64 and np.isinf(d["gama"])
    assert np.array_equal(d["c"], C.reshape(-1, order="F"))
    f2 = str(tmp_path / "data4.mat")       # Class 2 layout: C (matrix), mu, phi (uint8)
    scipy.io.savemat(f2, dict(C=C, c=C.reshape(-1, 1, order="F"), r=rs.random_sample((n, 1)),
                              l=rs.random_sample((m, 1)), p=np.ones((m, 1), np.uint8),
                              q=np.ones((n, 1), np.uint8), mu=1.25,
                              phi=np.ones((m * n, 1), np.uint8), m=np.uint16(m), n=np.uint16(n)))
    d = ipd.load_input(f2)
    assert d["cls"] == 2 and d["mu"] == 1.25 and d["phi"].dtype == np.float64
    assert np.array_equal(d["c"], C.reshape(-1, order="F"))


def test_load_input_reads_the_bundled_files_when_present():
    ipd = pkg()
    f = "/root/reference/Class1/InputData/data1-500.mat"
    if not os.path.exists(f):
        pytest.skip("reference not mounted (GPU box)")
    d = ipd.load_input(f)
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "data1_500.npz"))
    assert d["cls"] == 1 and d["m"] == d["n"] == 500 and np.isinf(d["gama"])
    assert np.array_equal(d["c"], g["c"]) and np.array_equal(d["r"], g["r"]) and np.array_equal(d["l"], g["l"])
