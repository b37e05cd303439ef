#!/usr/bin/env python
"""HBM traffic of one whole QUERY of a BASELINE config from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; summaries
written by scripts/rocpd_summary.py), merged into profiles/pmc_traffic.json under "_configs":
bytes = sum over kernels of calls x avg(FETCH_SIZE [KB]) x 2 x 1024 + calls x avg(WRITE_SIZE [KB]) x 1024, divided by the
number of query evaluations in the profiled run (the x 2 on FETCH_SIZE is the gfx950 correction of MI355X_MICROARCH.md's
HBM section, as for the per-kernel entries)."""
import json
import re
import sys


def parse(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+%s\s+(\d+)\s+([\d.]+)\s*$" % counter, line)
        if m:
            name = re.sub(r"<.*", "", m.group(1)).strip()
            calls, avg = int(m.group(2)), float(m.group(3))
            c0, s0 = out.get(name, (0, 0.0))
            out[name] = (c0 + calls, s0 + calls * avg)
    return out


def main(cfg, fetch_txt, write_txt, nqueries, rows, alg_bytes, json_path):
    f, w = parse(fetch_txt, "FETCH_SIZE"), parse(write_txt, "WRITE_SIZE")
    per = {}
    for k in sorted(set(f) | set(w)):
        if not k.endswith("_kernel"):          # libdthip's kernels only: the data generator's (torch) and the runtime's fills / copies are not the query
            continue
        b = (f.get(k, (0, 0.0))[1] * 2 * 1024 + w.get(k, (0, 0.0))[1] * 1024) / nqueries
        per[k] = {"launches_per_query": f.get(k, w.get(k))[0] / nqueries, "hbm_bytes_per_query": b}
    total = sum(v["hbm_bytes_per_query"] for v in per.values())
    try:
        doc = json.load(open(json_path))
    except Exception:
        doc = {}
    top = dict(sorted(per.items(), key=lambda kv: -kv[1]["hbm_bytes_per_query"])[:8])
    doc.setdefault("_configs", {})[cfg] = {"rows": rows, "alg_bytes": alg_bytes, "hbm_bytes_per_query": total,
                                           "amplification": total / alg_bytes, "queries_profiled": nqueries, "kernels": top}
    json.dump(doc, open(json_path, "w"), indent=1)
    print("%s: %.2f GB per query for %.2f GB algorithmic = %.2fx" % (cfg, total / 1e9, alg_bytes / 1e9, total / alg_bytes))
    for k, v in top.items():
        print("   %-34s %8.3f GB  x%g" % (k, v["hbm_bytes_per_query"] / 1e9, v["launches_per_query"]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), int(float(sys.argv[5])), int(float(sys.argv[6])), sys.argv[7])
