#!/usr/bin/env python
"""Small .jay files written by THE UNMODIFIED REFERENCE (`Frame.to_jay`), with the column values it holds
stored next to them (tests/golden/jay/expected.json), for the Jay reader of datatable_amd/jay.py.
Run in the dev container: DT_REFERENCE_SRC=/tmp/dt_oracle/src python tests/golden/make_jay_golden.py"""
import json
import math
import os
import sys

SRC = os.environ.get("DT_REFERENCE_SRC", "/tmp/dt_oracle/src")
sys.path.insert(0, SRC)
import datatable as dt  # noqa: E402

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "jay")
inf = math.inf
expected = {}


def save(name, frame):
    path = os.path.join(HERE, name + ".jay")
    if os.path.exists(path):
        os.remove(path)
    frame.to_jay(path)
    back = dt.fread(path)
    cols = [[("inf" if x == inf else "-inf" if x == -inf else x) if isinstance(x, float) else x for x in c] for c in back.to_list()]
    expected[name] = {"names": list(back.names), "stypes": [s.value for s in back.stypes], "nkeys": len(back.key), "columns": cols}
    print(name, back.shape, back.stypes, back.key)


save("types", dt.Frame(b=[True, None, False, True], i8=[1, -2, None, 127], i16=[100, None, -300, 32767],
                       i32=[None, 2**31 - 1, -5, 0], i64=[2**40, -2**62, None, 7], f32=[1.5, None, -inf, 0.25],
                       f64=[None, 1e300, -0.0, inf],
                       stypes=dict(b=dt.bool8, i8=dt.int8, i16=dt.int16, i32=dt.int32, i64=dt.int64, f32=dt.float32, f64=dt.float64)))
K = dt.Frame(k=[5, 1, 9, 3], v=[0.5, 1.5, 2.5, 3.5], n=[10, 20, 30, 40])
K.key = "k"
save("keyed", K)
B = dt.Frame(k=[(i * 7919) % 1000 for i in range(5000)], v=[(i % 97) / 8.0 for i in range(5000)])
save("big", B)
save("view", B[::3, :])                       # a sliced view is materialised by to_jay
save("empty", dt.Frame(a=[], b=[], stypes=dict(a=dt.int32, b=dt.float64)))
save("with_string", dt.Frame(k=[1, 2, 3], s=["a", None, "ccc"], v=[1.0, 2.0, 3.0]))
json.dump(expected, open(os.path.join(HERE, "expected.json"), "w"))
print("wrote", len(expected), "files")
