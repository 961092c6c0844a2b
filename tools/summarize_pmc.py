#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files into per-kernel means.

usage: summarize_pmc.py OUT.json NAME=counter_collection.csv [NAME=...] [RESIDENT_CYCLES=w,k]
                        [WORKLOAD=n1:1024,mask:bernoulli,rho:1,cycle:v[,newton_k:9]]
WORKLOAD: the bench.py arguments of the profiled command, stored under "_workload"; bench.py quotes
a summary's traffic only on a line of the same workload.
RESIDENT_CYCLES: the profiled command launched k_resident twice, with w (warm-up) and k (timed)
cycles; its traffic is then split into a fixed part (the one read of the matrices into registers)
and a per-cycle part (hand-off granules, transfer operators): "k_resident" entry.
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB per dispatch.  On gfx950
FETCH_SIZE counts 64 B per 128-B request for wide coalesced streaming reads
(MI355X_MICROARCH.md, section HBM): hbm_read_bytes = 2 * FETCH_SIZE * 1024.
"""
import collections
import csv
import json
import sys


def main():
    out = sys.argv[1]
    res = collections.defaultdict(dict)
    res_cycles = None
    workload = {"n1": 1024, "mask": "bernoulli", "rho": 1.0, "cycle": "v"}
    per_dispatch = collections.defaultdict(dict)
    for arg in sys.argv[2:]:   # (the dispatch split below needs it whatever the argument order)
        if arg.startswith("RESIDENT_CYCLES="):
            res_cycles = [int(v) for v in arg.split("=", 1)[1].split(",")]
    for arg in sys.argv[2:]:
        name, path = arg.split("=", 1)
        if name == "RESIDENT_CYCLES":
            continue
        if name == "WORKLOAD":
            kv = dict(t.split(":", 1) for t in path.split(","))
            workload = {"n1": int(kv.get("n1", 1024)), "mask": kv.get("mask", "bernoulli"),
                        "rho": float(kv.get("rho", 1.0)), "cycle": kv.get("cycle", "v")}
            if "newton_k" in kv:
                workload["newton_k"] = int(kv["newton_k"])
            continue
        agg = collections.defaultdict(list)
        resident = []   # (dispatch id, kernel, value) of every resident-kernel dispatch, in dispatch order
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == name:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
                if "k_resident" in r["Kernel_Name"]:
                    resident.append((int(r.get("Dispatch_Id", len(resident))), r["Kernel_Name"], float(r["Counter_Value"])))
        for k, v in agg.items():
            res[k][name + "_KiB_mean"] = sum(v) / len(v)
            res[k]["dispatches_" + name] = len(v)
        # the bench's warm-up and timed launches are the LAST dispatches of the process (a `--mask newton` line
        # runs the device driver first: hundreds of resident launches of other instantiations before them)
        resident.sort()
        if res_cycles and len(resident) >= len(res_cycles):
            last = resident[-len(res_cycles):]
            if len({k for _, k, _ in last}) == 1:
                per_dispatch[last[0][1]][name] = [v for _, _, v in last]
    for k, d in res.items():
        f = d.get("FETCH_SIZE_KiB_mean")
        w = d.get("WRITE_SIZE_KiB_mean")
        if f is not None:
            d["hbm_read_bytes_per_launch_corrected"] = 2.0 * f * 1024.0
        if w is not None:
            d["hbm_write_bytes_per_launch"] = w * 1024.0
        if f is not None and w is not None:
            d["hbm_traffic_bytes_per_launch"] = 2.0 * f * 1024.0 + w * 1024.0
    for k, d in per_dispatch.items():
        if res_cycles and all(len(d.get(n, [])) == len(res_cycles) for n in ("FETCH_SIZE", "WRITE_SIZE")):
            tr = [2.0 * f * 1024.0 + w * 1024.0 for f, w in zip(d["FETCH_SIZE"], d["WRITE_SIZE"])]
            (c0, c1), (t0, t1) = res_cycles[:2], tr[:2]
            per = (t1 - t0) / (c1 - c0)
            res["k_resident"] = {"kernel": k, "dispatch_cycles": res_cycles, "hbm_traffic_bytes": tr,
                                 "hbm_traffic_bytes_per_cycle": per,
                                 "hbm_traffic_bytes_fixed": t0 - per * c0}
    res["_workload"] = workload
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
