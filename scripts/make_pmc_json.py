#!/usr/bin/env python
"""profiles/pmc_traffic.json from the rocprofv3 PMC summaries (scripts/prof.sh):
HBM bytes per launch = FETCH_SIZE[KB] x 2 (gfx950 correction, MI355X_MICROARCH.md HBM section) x 1024
                     + WRITE_SIZE[KB] x 1024, per kernel, at the row count of the profiled run."""
import json
import re
import sys


def parse(path, counter):
    out = {}
    for line in open(path):
        m = re.match(r"^(\S.*?)\s+%s\s+(\d+)\s+([\d.]+)\s*$" % counter, line)
        if m:
            name = re.sub(r"<.*", "", m.group(1)).strip()
            out[name] = out.get(name, 0.0) + float(m.group(3)) * (1 if name not in out else 0) if False else float(m.group(3))
    return out


def main(fetch_txt, write_txt, rows, out_path):
    f, w = parse(fetch_txt, "FETCH_SIZE"), parse(write_txt, "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        if not k or k.startswith("__amd") or "kernel" not in k:
            continue
        b = f.get(k, 0.0) * 2 * 1024 + w.get(k, 0.0) * 1024
        res[k] = {"fetch_KB_raw": f.get(k), "write_KB_raw": w.get(k),
                  "hbm_bytes_per_launch_at_rows": {str(rows): b}}
    # the library build the counters were taken on (bench.py drops the record when it runs on another build)
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from datatable_amd import _lib
    res["_build_id"] = _lib.load().dthip_build_id().decode()
    json.dump(res, open(out_path, "w"), indent=1)
    for k, v in res.items():
        if not k.startswith("_"):
            print("%-32s %.3f GB / launch" % (k, v["hbm_bytes_per_launch_at_rows"][str(rows)] / 1e9))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4])
