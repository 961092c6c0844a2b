import os, sys, re, subprocess, collections
env = dict(os.environ, IPD_DEBUG_LEVELS="1", IPD_PROFILE="1")
for nomid in ("0",):
    env["IPD_NO_MID"] = nomid
    p = subprocess.run([sys.executable, "tools/bench_driver.py", "--sizes", "1024", "--classes", "1,2"], env=env, capture_output=True, text=True)
    for line in p.stdout.splitlines():
        if line.startswith("{"):
            import json
            d = json.loads(line)
            print("NO_MID", nomid, "cls", d["cls"], "apd_s %.3f" % d["apd_s"], d["profile_s_calls"])
    if nomid == "0":
        mids = collections.Counter(); rows = []
        for line in p.stderr.splitlines():
            m = re.search(r"J=(\d+) small=(\d) k_sub=(\d+) mid=(\d) resident=(\d).*levels: (.*)", line)
            if m:
                lv = [tuple(int(v) for v in t.split("/")) for t in m.group(6).split()]
                mids[(int(m.group(2)), int(m.group(4)), int(m.group(5)))] += 1
                rows.append((int(m.group(4)), lv))
        print("small/mid/resident counts:", dict(mids))
        import statistics
        no = [lv for mid, lv in rows if not mid and len(lv) >= 3]
        print("not mid examples:", no[:6])
        ye = [lv for mid, lv in rows if mid]
        print("mid examples:", ye[:4])
        import numpy as np
        big = [lv for mid, lv in rows if len(lv) >= 3 and lv[0][0] > 1024]
        a = np.array([[lv[0][0], lv[0][1] / lv[0][0], lv[1][0], lv[1][1] / lv[1][0], lv[2][0], lv[2][1] / lv[2][0], len(lv)] for lv in big])
        print("hierarchies with N1 > 1024 and >= 3 levels:", len(big))
        for q in (10, 25, 50, 75, 90, 100):
            print("  q%3d: N1 %5.0f nnz1/N1 %6.1f | N2 %5.0f nnz2/N2 %7.1f | N3 %4.0f nnz3/N3 %6.1f | J %d" % ((q,) + tuple(np.percentile(a, q, axis=0))))
        cond = (a[:, 1] <= 7) & (a[:, 3] <= 7) & (a[:, 4] <= 512)
        print("  would pass row-length conditions:", int(cond.sum()), " N2<=1024:", int((a[:,2] <= 1024).sum()))
        cond2 = (a[:, 1] <= 7) & (a[:, 3] <= 16) & (a[:, 4] <= 512)
        print("  with nnz2/N2 <= 16:", int(cond2.sum()))
