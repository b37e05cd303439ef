#!/usr/bin/env python
"""Golden answers for the Frame-level API, produced by RUNNING THE UNMODIFIED REFERENCE in the dev
container (same recipe as make_golden.py: DT_REFERENCE_SRC=/tmp/dt_oracle/src).

Every entry of tests/golden/frame_queries.json holds the input columns, a query string that is valid
Python both for `datatable` and for `datatable_amd.frame` (names DT, f, by, sort, sum, mean, min, max,
count, first, last), and what the reference returned: names, stype codes, column values (None = NA).
tests/test_frame_golden.py evaluates the same strings against datatable_amd.frame on the GPU."""
import json
import math
import os
import sys
import warnings

SRC = os.environ.get("DT_REFERENCE_SRC", "/tmp/dt_oracle/src")
sys.path.insert(0, SRC)
import datatable as dt  # noqa: E402
from datatable import f, by, sort, sum, mean, min, max, count, first, last  # noqa: E402,A004
from datatable import sd, median, nunique, cov, corr, cumsum, cumprod, cummin, cummax, cumcount, ngroup, fillna  # noqa: E402

dt.options.progress.enabled = False
inf = math.inf

FRAMES = {
    "appendixB": {"k": ([3, None, 1, 3, 1, None, 2, 3], 4), "v": ([1.5, 2.0, None, 4.0, None, 8.0, 16.0, inf], 7),
                  "i": (list(range(8)), 4)},
    "two_keys": {"a": ([2, 1, 2, 1, 2, 1, None, 2], 4), "b": ([5, 5, 4, 5, 4, 3, 1, None], 4),
                 "v": ([0, 1, 2, 3, 4, 5, 6, 7], 5), "w": ([0.5, -1.0, None, 2.25, 1e300, -1e300, 0.0, 3.0], 7)},
    "types": {"g": ([1, 1, 2, 2, 2, 3], 2), "i8": ([1, -2, None, 4, 5, -6], 2), "i16": ([100, None, 300, -400, 500, 600], 3),
              "i64": ([2**40, -2**41, None, 7, 8, 9], 5), "f32": ([1.5, None, 2.5, -3.5, 4.5, 0.25], 6),
              "b": ([True, False, None, True, True, False], 1)},
    "floatkey": {"k": ([0.0, -0.0, 1.5, None, 1.5, -2.25, None, 0.0], 7), "v": ([1, 2, 3, 4, 5, 6, 7, 8], 4)},
    "big": {"k": ([(i * 7919) % 97 - 40 for i in range(5000)], 4), "v": ([((i * 31) % 1000) / 8.0 - 60 for i in range(5000)], 7),
            "n": ([None if i % 13 == 0 else (i * 17) % 500 - 250 for i in range(5000)], 4)},
}

QUERIES = [
    ("appendixB", "DT[:, [sum(f.v), mean(f.v), min(f.v), max(f.v), count(f.v), count()], by(f.k)]"),
    ("appendixB", "DT[:, f.i, by(f.k)]"),
    ("appendixB", "DT[:, :, by(f.k)]"),
    ("appendixB", "DT[:, [first(f.v), last(f.v), first(f.i), last(f.i)], by(f.k)]"),
    ("appendixB", "DT[:, sum(f.v), by(-f.k)]"),
    ("appendixB", "DT[:, :, by(f.k), sort(-f.v)]"),
    ("appendixB", "DT[:, :, sort(f.v)]"),
    ("appendixB", "DT[:, :, sort(f.k, reverse=True)]"),
    ("appendixB", "DT[:, :, sort(f.k, na_position='last')]"),
    ("appendixB", "DT[:, :, sort(f.k, na_position='remove')]"),
    ("appendixB", "DT[:, :, sort(f.k, f.v)]"),
    ("appendixB", "DT[:, [sum(f.v), count()]]"),
    ("appendixB", "DT[:, {'total': sum(f.v), 'n': count()}, by(f.k)]"),
    ("appendixB", "DT[:, [mean(f.v), f.i], by(f.k)]"),
    ("appendixB", "DT[:, 'k', by('k')]"),
    ("appendixB", "DT[f.v > 1.6, :]"),
    ("appendixB", "DT[f.v > 1.6, :][:, sum(f.v), by(f.k)]"),
    ("appendixB", "DT[f.i >= 2, :][f.v < 10, :][:, [count(), max(f.i)], by(f.k)]"),
    ("appendixB", "DT[f.k != 3, :]"),
    ("appendixB", "DT[f.k == 3, ['i', 'v']]"),
    ("two_keys", "DT[:, [count(), sum(f.v), sum(f.w)], by(f.a, f.b)]"),
    ("two_keys", "DT[:, [min(f.w), max(f.w), mean(f.w)], by(f.b, f.a)]"),
    ("two_keys", "DT[:, f.v, by(f.a, f.b)]"),
    ("two_keys", "DT[:, sum(f[:]), by(f.a)]"),
    ("two_keys", "DT[:, count(), by(-f.a, f.b)]"),
    ("two_keys", "DT[:, :, by(f.a), sort(f.w)]"),
    ("two_keys", "DT[:, :, sort(f.a, f.b)]"),
    ("two_keys", "DT[:, [f.a, f.b], by(f.a, f.b)]"),
    ("types", "DT[:, sum(f[:]), by(f.g)]"),
    ("types", "DT[:, mean(f[:]), by(f.g)]"),
    ("types", "DT[:, min(f[:]), by(f.g)]"),
    ("types", "DT[:, max(f[:]), by(f.g)]"),
    ("types", "DT[:, count(f[:]), by(f.g)]"),
    ("types", "DT[:, [first(f.f32), last(f.i8)], by(f.g)]"),
    ("types", "DT[:, [sum(f.i8), count()], by(f.b)]"),
    ("types", "DT[:, count(), by(f.i16)]"),
    ("types", "DT[:, count(), by(f.f32)]"),
    ("types", "DT[:, :, sort(f.f32, reverse=True)]"),
    ("types", "DT[:, :, sort(f.b, f.i8)]"),
    ("floatkey", "DT[:, [count(), sum(f.v)], by(f.k)]"),
    ("floatkey", "DT[:, f.v, by(f.k)]"),
    ("floatkey", "DT[:, :, sort(f.k)]"),
    ("floatkey", "DT[:, :, sort(-f.k)]"),
    ("big", "DT[:, [sum(f.v), mean(f.v), min(f.n), max(f.n), count(f.n), count()], by(f.k)]"),
    ("big", "DT[:, [sum(f.n), mean(f.n)], by(f.k)]"),
    ("big", "DT[f.v > 0, :][:, [sum(f.v), count()], by(f.k)]"),
    ("big", "DT[:, f.n, by(f.k)]"),
    ("big", "DT[:, :, sort(f.n, f.k)]"),
    ("big", "DT[:, [first(f.n), last(f.v)], by(f.k)]"),
    # the other group reducers and the cumulative operators
    ("appendixB", "DT[:, [sd(f.v), median(f.v), nunique(f.v)], by(f.k)]"),
    ("appendixB", "DT[:, [sd(f.i), median(f.i), nunique(f.i), sum(f.i)], by(f.k)]"),
    ("appendixB", "DT[:, [cov(f.v, f.i), corr(f.v, f.i)], by(f.k)]"),
    ("appendixB", "DT[:, [cumsum(f.v), cumsum(f.i), cummax(f.v), cumcount(), ngroup()], by(f.k)]"),
    ("appendixB", "DT[:, [cumsum(f.v, reverse=True), cummin(f.i, reverse=True), cumcount(reverse=True), ngroup(reverse=True)], by(f.k)]"),
    ("appendixB", "DT[:, [cumsum(f.i), cumprod(f.i), cummin(f.v), cummax(f.i)]]"),
    ("appendixB", "DT[:, [sd(f.v), median(f.i), nunique(f.k), cov(f.i, f.i), corr(f.i, f.v)]]"),
    ("appendixB", "DT[:, [f.i, cumsum(f.i), mean(f.i)], by(f.k)]"),
    ("appendixB", "DT[:, cumsum(f[:]), by(f.k)]"),
    ("appendixB", "DT[:, cumsum(f.k), by(f.k)]"),
    ("appendixB", "DT[:, {'s': sd(f.v), 'm': median(f.v)}, by(f.k)]"),
    ("appendixB", "DT[:, cumsum(f.i), by(f.k), sort(-f.i)]"),
    ("appendixB", "DT[f.v > 1.6, :][:, [median(f.v), cumsum(f.i)], by(f.k)]"),
    ("two_keys", "DT[:, [sd(f.w), median(f.v), nunique(f.w)], by(f.a, f.b)]"),
    ("two_keys", "DT[:, [cumsum(f.v), cumprod(f.v), cummin(f.w), cummax(f.w)], by(f.a)]"),
    ("two_keys", "DT[:, corr(f.v, f[:]), by(f.a)]"),
    ("two_keys", "DT[:, [cumcount(), ngroup()], by(f.a, f.b)]"),
    ("types", "DT[:, sd(f[:]), by(f.g)]"),
    ("types", "DT[:, median(f[:]), by(f.g)]"),
    ("types", "DT[:, nunique(f[:]), by(f.g)]"),
    ("types", "DT[:, cumsum(f[:]), by(f.g)]"),
    ("types", "DT[:, cumprod(f[:]), by(f.g)]"),
    ("types", "DT[:, cummin(f[:]), by(f.g)]"),
    ("types", "DT[:, cummax(f[:], reverse=True), by(f.g)]"),
    ("types", "DT[:, [cov(f.i8, f.i16), cov(f.f32, f.f32), corr(f.i64, f.f32)], by(f.g)]"),
    ("floatkey", "DT[:, [nunique(f.k), median(f.v)], by(f.k)]"),
    ("floatkey", "DT[:, [nunique(f.k), median(f.k), sd(f.k)]]"),
    ("big", "DT[:, [sd(f.v), median(f.v), nunique(f.v), median(f.n), nunique(f.n)], by(f.k)]"),
    ("big", "DT[:, [cov(f.v, f.n), corr(f.v, f.n)], by(f.k)]"),
    ("big", "DT[:, [cumsum(f.n), cummax(f.v), cummin(f.n, reverse=True), cumcount()], by(f.k)]"),
    ("big", "DT[:, [cumsum(f.v), cumsum(f.n), ngroup()]]"),
    # fillna(col, reverse) (fexpr_fillna.cc:85-117): the same segmented scan
    ("appendixB", "DT[:, [fillna(f.v), fillna(f.v, reverse=True), fillna(f.i)], by(f.k)]"),
    ("appendixB", "DT[:, fillna(f[:], reverse=True)]"),
    ("types", "DT[:, fillna(f[:]), by(f.g)]"),
    ("types", "DT[:, fillna(f[:], reverse=True), by(f.g)]"),
    ("big", "DT[:, [fillna(f.n), fillna(f.n, reverse=True), cumsum(f.n)], by(f.k)]"),
    ("big", "DT[f.v > 0, :][:, fillna(f.n), by(f.k)]"),
]

ST = {1: dt.bool8, 2: dt.int8, 3: dt.int16, 4: dt.int32, 5: dt.int64, 6: dt.float32, 7: dt.float64}


def build(spec):
    cols, names = [], []
    for nm, (vals, st) in spec.items():
        cols.append(dt.Frame([vals], stype=ST[st])[0] if False else dt.Frame([vals], stype=ST[st]))
        names.append(nm)
    fr = dt.cbind(*cols)
    fr.names = names
    return fr


def main():
    out = []
    for fname, q in QUERIES:
        DT = build(FRAMES[fname])
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            R = eval(q)
        cols = R.to_list()
        cols = [[(None if (isinstance(x, float) and x != x) else x) for x in c] for c in cols]
        out.append({"frame": fname, "query": q, "names": list(R.names), "stypes": [s.value for s in R.stypes],
                    "columns": cols})
        print("%-10s %-75s -> %s %s" % (fname, q, R.shape, R.names))
    frames = {k: {nm: {"values": v, "stype": st} for nm, (v, st) in spec.items()} for k, spec in FRAMES.items()}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "frame_queries.json")

    def enc(o):
        if isinstance(o, float) and math.isinf(o):
            return "inf" if o > 0 else "-inf"
        raise TypeError
    def walk(o):
        if isinstance(o, float) and math.isinf(o):
            return "inf" if o > 0 else "-inf"
        if isinstance(o, list):
            return [walk(x) for x in o]
        if isinstance(o, dict):
            return {k: walk(v) for k, v in o.items()}
        return o
    json.dump(walk({"frames": frames, "queries": out}), open(path, "w"))
    print("wrote %s (%d queries)" % (path, len(out)))


if __name__ == "__main__":
    main()
