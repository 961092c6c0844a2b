#!/usr/bin/env python3
"""Generates the fixtures in tests/golden/ (run in the BUILD container only).

Inputs : the reference's bundled data file Class1/InputData/data1-500.mat (a data file,
         read with scipy.io.loadmat; nothing of the reference's code is executed -- it is
         MATLAB and cannot run here) and seeded synthetic masks.
Outputs: *derived* golden vectors -- produced by the build's own CPU oracle
         (oracle/ipd_oracle.py, oracle/drivers.py), NOT by the reference (SURVEY.md 4.2,
         8c: parity unpinned).  They pin the oracle against regressions and give the GPU
         tests realistic Newton systems without /root/reference at run time.

  python tests/golden/make_golden.py [/root/reference]
"""
import hashlib
import os
import sys

import numpy as np
import scipy.io
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import drivers as D          # noqa: E402
from oracle import ipd_oracle as O       # noqa: E402
from tests import problems as PR         # noqa: E402


def digest(M):
    M = sp.csc_matrix(M)
    M.sort_indices()
    h = hashlib.sha256()
    for a in (np.asarray(M.shape, np.int64), M.indptr.astype(np.int64), M.indices.astype(np.int64),
              M.data.astype(np.float64)):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def solve_record(pd, opts, pot=False):
    tr = []
    rng = O.matlab_rng()
    if pot:
        zeta, it, res, info = O.AMG4POT(pd, opts, rng, trace=tr)
    else:
        zeta, it, res, info = O.Hybrid_AMG(pd, opts, rng, trace=tr)
    used = sum(len(t["guess"]) + sum(len(i["mis"]["rand"]) for i in t["h"].info[2:]
                                     if i and i.get("mis")) for t in tr)
    rec = dict(zeta=zeta, it=it, res=res, info=np.asarray(info), ncalls=len(tr), rand_consumed=used)
    for c, t in enumerate(tr):
        h = t["h"]
        rec["c%d_levels" % c] = np.asarray(h.level_sizes())
        rec["c%d_nnz" % c] = np.asarray(h.level_nnz())
        rec["c%d_relresk" % c] = np.asarray(t["rel_resk"])
        rec["c%d_isnsp" % c] = t["isnsp"]
        rec["c%d_fnode" % c] = t["fnode"]
        for k in range(2, h.J + 1):
            rec["c%d_cmask%d" % (c, k)] = np.packbits(h.info[k]["isC"])
            rec["c%d_Ahash%d" % (c, k)] = digest(h.Ack[k])
            rec["c%d_Phash%d" % (c, k)] = digest(h.Prok[k])
    return rec


def realistic(ref_root):
    d = scipy.io.loadmat(os.path.join(ref_root, "Class1", "InputData", "data1-500.mat"))
    g = lambda k: np.asarray(d[k], dtype=np.float64).ravel()   # SURVEY A-12: cast to float64
    c, r, l, p, q, gama = g("c"), g("r"), g("l"), g("p"), g("q"), g("gama")
    m, n = len(p), len(q)
    want = [(1, 1), (3, 1), (8, 1), (20, 1), (40, 1)]
    run = D.apd_ssn_class1(c, r, l, p, q, gama, capture=want)
    assert run["converged"] and run["k"] == 58 and abs(run["fval"] - 1.126046) < 5e-7, run["k"]
    Es = [x["E"] for x in run["log"]]
    summary = dict(k=run["k"], fval=run["fval"], steps=len(Es), Emin=min(Es), Emed=float(np.median(Es)),
                   Emax=max(Es))
    np.savez_compressed(os.path.join(HERE, "class1_500_run.npz"), **summary)
    opts = O.amg_options_class1("w")            # the driver's options (APD_SsN_Class1.m:87-88)
    for cap in run["captured"]:
        s = np.unpackbits(cap["s"])[:cap["mn"]]
        H0 = O.ASAt(s, p, q)
        pd = dict(m=m, n=n, p=p, q=q, bk1=cap["bk1"], tk=cap["tk"], z=cap["z"], H0=H0,
                  T=sp.csr_matrix((m + n, m + n)))
        rec = solve_record(pd, opts)
        rec.update(m=m, n=n, s_packed=cap["s"], bk1=cap["bk1"], tk=cap["tk"], z=cap["z"], k=cap["k"],
                   ssn=cap["ssn"], E=cap["E"], H0_nnz=H0.nnz, H0_hash=digest(H0))
        np.savez_compressed(os.path.join(HERE, "class1_500_k%02d.npz" % cap["k"]), **rec)
        print("class1_500 k=%d E=%d it=%d info=%s levels=%s" % (cap["k"], cap["E"], rec["it"],
                                                                 rec["info"], rec.get("c0_levels")))


def synthetic():
    cases = [("tree24", 24, 24, PR.mask_tree(24, 24, seed=1), None, "v"),
             ("tree40x28", 40, 28, PR.mask_tree(40, 28, seed=2, connect=False), None, "w"),
             ("bern32", 32, 32, PR.mask_bernoulli(32, 32, 0.15, seed=3), None, "w"),
             ("class2_36", 36, 36, PR.mask_bernoulli(36, 36, 0.1, seed=4), 0.7, "w")]
    for name, m, n, s, tfrac, cyc in cases:
        t = None
        if tfrac is not None:
            t = (np.random.RandomState(9).random_sample(m + n) < tfrac).astype(float)
        pd = PR.make_prob(m, n, s, t=t, pq_random=True)
        pd["H0"] = O.ASAt(s, pd["p"], pd["q"])
        pot = tfrac is not None
        if pot:
            pd["z"] = np.random.RandomState(5).randn(m + n + 1)
            pd["phi"] = np.ones(m * n)
        opts = O.amg_options_class2(cyc) if pot else O.amg_options_class1(cyc)
        rec = solve_record(pd, opts, pot)
        H0 = sp.csc_matrix(pd["H0"])
        Ae = sp.csc_matrix(O.build_Ae(pd["H0"], pd["T"], pd["p"], pd["q"], pd["bk1"], pd["tk"])[0])
        rec.update(m=m, n=n, s=s, p=pd["p"], q=pd["q"], bk1=pd["bk1"], tk=pd["tk"], z=pd["z"],
                   t=(t if t is not None else np.zeros(m + n)), pot=int(pot), cycle=cyc,
                   H0_indptr=H0.indptr, H0_indices=H0.indices, H0_data=H0.data,
                   Ae_indptr=Ae.indptr, Ae_indices=Ae.indices, Ae_data=Ae.data,
                   Ax=O.Ax(np.arange(m * n) * 0.01, pd["p"], pd["q"]),
                   Aty=O.Aty(np.arange(m + n) * 0.1, pd["p"], pd["q"]))
        np.savez_compressed(os.path.join(HERE, "synth_%s.npz" % name), **rec)
        print("synth", name, "it", rec["it"], "info", rec["info"])


def bundled_driver_runs(ref_root):
    """The reference's two bundled problems (data files, read with scipy.io.loadmat) as npz
    inputs, and the histories the restated drivers (oracle/drivers.py, inner_solver = 4)
    produce on them -- the expected values of tests/test_gpu_driver_golden.py."""
    d = scipy.io.loadmat(os.path.join(ref_root, "Class1", "InputData", "data1-500.mat"))
    g = lambda k: np.asarray(d[k], dtype=np.float64).ravel()
    c, r, l, p, q, gama = g("c"), g("r"), g("l"), g("p"), g("q"), g("gama")
    assert np.all(p == 1) and np.all(q == 1) and np.all(np.isinf(gama))
    np.savez_compressed(os.path.join(HERE, "data1_500.npz"), c=c, r=r, l=l)
    x0, l0 = D.warmup_class1(c, r, l, p, q, gama, 100)
    run = D.apd_ssn_class1(c, r, l, p, q, gama, inner="amg", start=(x0, l0), rng=O.matlab_rng())
    assert run["converged"]
    np.savez_compressed(os.path.join(HERE, "class1_500_driver.npz"), k=run["k"], fval=run["fval"],
                        fxk=run["fxk"], KKT_xk=run["KKT_xk"], KKT_lk=run["KKT_lk"],
                        SsN_itnum=run["SsN_itnum"], warm_x_norm=np.linalg.norm(x0),
                        warm_x_nnz=int((x0 > 0).sum()), warm_l=l0,
                        E=[e["E"] for e in run["log"]], it=[e["it"] for e in run["log"]])
    print("class1 driver: k=%d f=%.6f steps=%d" % (run["k"], run["fval"], len(run["log"])))
    d = scipy.io.loadmat(os.path.join(ref_root, "Class2", "InputData", "data4-500.mat"))
    c, r, l, p, q, phi = g("c"), g("r"), g("l"), g("p"), g("q"), g("phi")
    mu = float(np.asarray(d["mu"]).ravel()[0])
    assert np.all(p == 1) and np.all(q == 1) and np.all(phi == 1)
    assert np.array_equal(np.asarray(d["C"], float).reshape(-1, order="F"), c)
    np.savez_compressed(os.path.join(HERE, "data4_500.npz"), c=c, r=r, l=l, mu=mu)
    u0, l0 = D.warmup_class2(c, r, l, p, q, mu, phi, 100)
    run = D.apd_ssn_class2(c, r, l, p, q, mu, phi, inner="amg", start=(u0, l0),
                           rng=O.matlab_rng())
    assert run["converged"]
    np.savez_compressed(os.path.join(HERE, "class2_500_driver.npz"), k=run["k"], fval=run["fval"],
                        fxk=run["fxk"], KKT_xk=run["KKT_xk"], KKT_lk=run["KKT_lk"],
                        KKT_yk=run["KKT_yk"], KKT_zk=run["KKT_zk"], SsN_itnum=run["SsN_itnum"],
                        warm_u_norm=np.linalg.norm(u0), warm_l=l0,
                        E=[e["E"] for e in run["log"]], it=[e["it"] for e in run["log"]])
    print("class2 driver: k=%d f=%.6f steps=%d" % (run["k"], run["fval"], len(run["log"])))


if __name__ == "__main__":
    root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
    if "--drivers-only" not in sys.argv:
        synthetic()
        realistic(root)
    bundled_driver_runs(root)
