"""oracle/cpu_cycle.c (the C/OpenMP restatement bench.py times as `cpu_baseline`) against the
Python oracle on the same hierarchy: same residual after every cycle, V and W, with and without
the kernel-space correction, 1 and 2 threads.  CPU only."""
import numpy as np
import pytest

from oracle import ipd_oracle as O
from oracle.cpu_cycle import CpuCycle
from tests import problems as PR


def newton_system(m, n, s):
    pd = PR.make_prob(m, n, s)
    H0 = O.ASAt(s, pd["p"], pd["q"])
    Ae = O.build_Ae(H0, pd["T"], pd["p"], pd["q"], pd["bk1"], pd["tk"])[0]
    return Ae, np.concatenate([pd["q"], -pd["p"]]) * pd["z"]


def oracle_loop(h, A, f, x0, cycles, wc, isnsp):
    x = x0.copy()
    res = [np.linalg.norm(A @ x - f)]
    for _ in range(cycles):
        r = f - A @ x
        x = x + (O.MG_Wcycle(h, r, isnsp) if wc else O.MG_Vcycle(h, r, isnsp))
        res.append(np.linalg.norm(A @ x - f))
    return x, np.array(res)


@pytest.mark.parametrize("mask,isnsp,wc", [("tree", 1, False), ("tree", 1, True), ("dense", 1, False),
                                           ("bern", 0, True)])
def test_c_cycle_matches_python_oracle(mask, isnsp, wc):
    m, n = 60, 44
    s = {"tree": lambda: PR.mask_tree(m, n, seed=4), "dense": lambda: PR.mask_bernoulli(m, n, 1.0),
         "bern": lambda: PR.mask_bernoulli(m, n, 0.3, seed=9)}[mask]()
    Ae, f = newton_system(m, n, s)
    opts = O.amg_options_class1("w" if wc else "v")
    opts.update(fnode=n, isnsp=isnsp)
    h = O.amg_setup(Ae, opts, O.matlab_rng())
    A = h.Ack[1]
    x0 = 1e-4 * np.random.RandomState(1).random_sample(m + n)
    xo, reso = oracle_loop(h, A, f, x0, 4, wc, isnsp)
    cc = CpuCycle(h, isnsp)
    for threads in (1, 2):
        x, sec, res = cc.run(f, x0, 4, wc, threads)
        assert sec >= 0.0
        assert np.allclose(res, reso, rtol=1e-6, atol=1e-10 * reso[0]), (res, reso)
        assert np.linalg.norm(A @ (x - xo)) <= 1e-9 * np.linalg.norm(f)
