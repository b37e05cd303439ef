#!/usr/bin/env python
"""Randomised Frame-level golden answers: seeded random frames (every fixed-width stype, NAs, +-inf, -0.0)
and random `DT[i, j, by/sort]` queries over them, evaluated by THE UNMODIFIED REFERENCE in the dev container
(DT_REFERENCE_SRC=/tmp/dt_oracle/src).  Output: tests/golden/frame_fuzz.json, same layout as
frame_queries.json; tests/test_frame_golden.py evaluates the same strings against datatable_amd.frame."""
import json
import math
import os
import random
import sys
import warnings

SRC = os.environ.get("DT_REFERENCE_SRC", "/tmp/dt_oracle/src")
sys.path.insert(0, SRC)
import datatable as dt  # noqa: E402
from datatable import f, by, sort, sum, mean, min, max, count, first, last  # noqa: E402,A004
from datatable import sd, median, nunique, cov, corr, cumsum, cumprod, cummin, cummax, cumcount, ngroup  # noqa: E402

dt.options.progress.enabled = False
ST = {1: dt.bool8, 2: dt.int8, 3: dt.int16, 4: dt.int32, 5: dt.int64, 6: dt.float32, 7: dt.float64}
rnd = random.Random(20251001)
NEG_ZERO = False      # batch 5 also puts -0.0 among the float VALUES (keys always had it)


def rand_col(st, n, role):
    vals = []
    for _ in range(n):
        if rnd.random() < (0.08 if role == "key" else 0.12):
            vals.append(None)
        elif st == 1:
            vals.append(rnd.random() < 0.5)
        elif st in (6, 7):
            if role == "key":
                vals.append(rnd.choice([0.0, -0.0, 1.5, -2.25, 3.0, math.inf]))
            else:
                r = rnd.random()
                vals.append(math.inf if r < 0.02 else -math.inf if r < 0.03 else -0.0 if (NEG_ZERO and r < 0.06) else rnd.randint(-400, 400) / 8.0)
        else:
            lim = {2: 5, 3: 8, 4: 12, 5: 6}[st] if role == "key" else {2: 100, 3: 3000, 4: 10**6, 5: 10**12}[st]
            vals.append(rnd.randint(-lim, lim))
    return vals


def make_frame(seed):
    rnd.seed(seed)
    n = rnd.choice([1, 2, 7, 40, 150] if seed < 100000 else [0, 1, 3, 20, 90])
    spec = {}
    for i in range(rnd.randint(1, 2)):
        spec["k%d" % i] = (rand_col(rnd.choice([1, 2, 3, 4, 5, 4, 5, 7]), n, "key"), None)
    for i in range(rnd.randint(2, 4)):
        spec["v%d" % i] = (rand_col(rnd.choice([1, 2, 3, 4, 5, 6, 7, 7, 4]), n, "val"), None)
    out = {}
    for nm, (vals, _) in spec.items():
        st = 1 if all(isinstance(x, bool) or x is None for x in vals) and any(isinstance(x, bool) for x in vals) else None
        out[nm] = vals
    return out


def stype_of(vals_hint):
    return vals_hint


RED = ["sum", "mean", "min", "max", "count", "first", "last", "sd", "median", "nunique"]
CUM = ["cumsum", "cumprod", "cummin", "cummax"]


def rand_query(names):
    keys = [nm for nm in names if nm.startswith("k")]
    vals = [nm for nm in names if nm.startswith("v")]
    kk = rnd.sample(keys, rnd.randint(1, len(keys)))
    bys = "by(%s)" % ", ".join(("-f.%s" if rnd.random() < 0.15 else "f.%s") % k for k in kk)
    kind = rnd.random()
    if kind < 0.45:
        items = []
        for _ in range(rnd.randint(1, 4)):
            r = rnd.random()
            if r < 0.1:
                items.append("count()")
            elif r < 0.2 and len(vals) >= 2:
                a, b = rnd.sample(vals, 2)
                items.append("%s(f.%s, f.%s)" % (rnd.choice(["cov", "corr"]), a, b))
            else:
                items.append("%s(f.%s)" % (rnd.choice(RED), rnd.choice(vals)))
        j = "[%s]" % ", ".join(items)
        return "DT[:, %s, %s]" % (j, bys) if rnd.random() < 0.85 else "DT[:, %s]" % j
    if kind < 0.65:
        items = []
        for _ in range(rnd.randint(1, 3)):
            r = rnd.random()
            if r < 0.15:
                items.append("%s(%s)" % (rnd.choice(["cumcount", "ngroup"]), rnd.choice(["", "reverse=True"])))
            else:
                items.append("%s(f.%s%s)" % (rnd.choice(CUM), rnd.choice(vals), rnd.choice(["", ", reverse=True"])))
        j = "[%s]" % ", ".join(items)
        return "DT[:, %s, %s]" % (j, bys) if rnd.random() < 0.8 else "DT[:, %s]" % j
    if kind < 0.75:
        cols = ", ".join("f.%s" % c for c in rnd.sample(names, rnd.randint(1, len(names))))
        opt = rnd.choice(["", ", reverse=True", ", na_position='last'"])
        return "DT[:, :, sort(%s%s)]" % (cols, opt)
    if kind < 0.85:
        return "DT[:, %s, %s]" % (rnd.choice([":", "f.%s" % rnd.choice(vals), "[f.%s, mean(f.%s)]" % (rnd.choice(vals), rnd.choice(vals))]), bys)
    if kind < 0.93:
        v = rnd.choice(vals)
        op = rnd.choice([">", ">=", "<", "<=", "==", "!="])
        return "DT[f.%s %s %s, :][:, [%s(f.%s), count()], %s]" % (v, op, rnd.choice(["0", "1", "-3", "2.5"]), rnd.choice(RED), rnd.choice(vals), bys)
    return "DT[:, :, %s, sort(f.%s)]" % (bys, rnd.choice(vals))


def rand_query2(names):
    keys = [nm for nm in names if nm.startswith("k")]
    vals = [nm for nm in names if nm.startswith("v")]
    kk = rnd.sample(keys, rnd.randint(1, len(keys)))
    style = rnd.random()
    if style < 0.3:
        bys = "by(%s)" % ", ".join("'%s'" % k for k in kk)
    else:
        bys = "by(%s)" % ", ".join("f.%s" % k for k in kk)
    kind = rnd.random()
    if kind < 0.2:
        return "DT[:, %s(f[:]), %s]" % (rnd.choice(RED + CUM), bys)
    if kind < 0.3:
        return "DT[:, %s(f[:])]" % rnd.choice(RED + CUM)
    if kind < 0.45:
        items = ", ".join("'%s': %s(f.%s)" % (nm, rnd.choice(RED), rnd.choice(vals)) for nm in rnd.sample(["a", "b", "v0", "k0", "zz"], rnd.randint(1, 3)))
        return "DT[:, {%s}, %s]" % (items, bys)
    if kind < 0.55:
        return "DT[:, [f.%s, %s(f.%s), count()], %s]" % (rnd.choice(vals), rnd.choice(RED), rnd.choice(vals), bys)
    if kind < 0.65:
        return "DT[:, [%s(f.%s), %s(f.%s)], %s, sort(%sf.%s)]" % (rnd.choice(["first", "last", "cumsum", "cummax"]), rnd.choice(vals),
                                                                    rnd.choice(["first", "last", "cumcount"]).replace("cumcount", "max"), rnd.choice(vals), bys,
                                                                    rnd.choice(["", "-"]), rnd.choice(vals))
    if kind < 0.75:
        v, w = rnd.choice(vals), rnd.choice(vals)
        return "DT[f.%s %s %s, :][f.%s %s %s, :][:, [count(), %s(f.%s)], %s]" % (
            v, rnd.choice([">", "<=", "!="]), rnd.choice(["0", "1", "-2"]), w, rnd.choice(["<", ">=", "=="]), rnd.choice(["0", "3", "1.5"]),
            rnd.choice(RED), rnd.choice(vals), bys)
    if kind < 0.82:
        return "DT[:, %s, %s]" % (rnd.choice(["f[:]", "':'".strip("'"), "[f.%s, f.%s]" % (rnd.choice(names), rnd.choice(names))]), bys)
    if kind < 0.9:
        return "DT[:, [f.%s, f.%s], sort(f.%s, f.%s, na_position='%s')]" % (rnd.choice(names), rnd.choice(names), rnd.choice(names), rnd.choice(names),
                                                                          rnd.choice(["first", "last", "remove"]))
    if kind < 0.95:
        return "DT[:, [%s(f.%s, f.%s), %s(f.%s, f[:])], %s]" % (rnd.choice(["cov", "corr"]), rnd.choice(vals), rnd.choice(vals),
                                                                 rnd.choice(["cov", "corr"]), rnd.choice(vals), bys)
    return "DT[:, [cumcount(), ngroup(), count()], %s]" % bys


def rand_query4(names):
    """chains of views: filters, sorts and column selections stacked before a final by() / sort()"""
    keys = [nm for nm in names if nm.startswith("k")]
    vals = [nm for nm in names if nm.startswith("v")]
    q = "DT"
    for _ in range(rnd.randint(1, 3)):
        r = rnd.random()
        if r < 0.5:
            q += "[f.%s %s %s, :]" % (rnd.choice(names), rnd.choice([">", ">=", "<", "<=", "==", "!="]), rnd.choice(["0", "1", "-1", "2", "0.5", "True"]))
        elif r < 0.8:
            q += "[:, :, sort(%s%s)]" % (", ".join("f.%s" % c for c in rnd.sample(names, rnd.randint(1, 2))), rnd.choice(["", ", reverse=True"]))
        else:
            q += "[:, :, by(f.%s)]" % rnd.choice(keys)
    r = rnd.random()
    k = rnd.choice(keys)
    if r < 0.35:
        q += "[:, [%s(f.%s), %s(f.%s), count()], by(f.%s)]" % (rnd.choice(RED), rnd.choice(vals), rnd.choice(RED), rnd.choice(vals), k)
    elif r < 0.5:
        q += "[:, [%s(f.%s), %s(f.%s%s)], by(f.%s)]" % (rnd.choice(CUM), rnd.choice(vals), rnd.choice(CUM), rnd.choice(vals), rnd.choice(["", ", reverse=True"]), k)
    elif r < 0.65:
        q += "[:, [%s]]" % ", ".join("'%s'" % c for c in rnd.sample(names, rnd.randint(1, len(names))))
    elif r < 0.8:
        q += "[:, [%s(f.%s), %s(f.%s)]]" % (rnd.choice(RED), rnd.choice(vals), rnd.choice(RED), rnd.choice(names))
    elif r < 0.9:
        q += "[:, f.%s, by(f.%s)]" % (rnd.choice(vals), k)
    return q


def rand_query5(names):
    """corners: reducers / cumulative operators ON key columns, reverse lists, na_position + reverse, dict j with
    cumulative operators, several descending by-columns, non-integral and negative filter scalars, -0.0 values"""
    keys = [nm for nm in names if nm.startswith("k")]
    vals = [nm for nm in names if nm.startswith("v")]
    kk = rnd.sample(keys, rnd.randint(1, len(keys)))
    bys = "by(%s)" % ", ".join(("-f.%s" if rnd.random() < 0.4 else "f.%s") % k for k in kk)
    kind = rnd.random()
    anyc = lambda: rnd.choice(names)
    if kind < 0.2:
        items = ["%s(f.%s)" % (rnd.choice(RED), anyc()) for _ in range(rnd.randint(1, 3))]
        return "DT[:, [%s], %s]" % (", ".join(items), bys)
    if kind < 0.35:
        items = ["%s(f.%s%s)" % (rnd.choice(CUM), anyc(), rnd.choice(["", ", reverse=True"])) for _ in range(rnd.randint(1, 3))]
        return "DT[:, [%s], %s]" % (", ".join(items), bys)
    if kind < 0.45:
        cols = rnd.sample(names, rnd.randint(2, min(3, len(names))))
        rev = "[%s]" % ", ".join(rnd.choice(["True", "False"]) for _ in cols)
        return "DT[:, :, sort(%s, reverse=%s%s)]" % (", ".join("f.%s" % c for c in cols), rev, rnd.choice(["", ", na_position='last'"]))
    if kind < 0.55:
        return "DT[:, :, sort(%sf.%s, na_position='%s', reverse=%s)]" % (rnd.choice(["", "-"]), anyc(), rnd.choice(["first", "last", "remove"]), rnd.choice(["True", "False"]))
    if kind < 0.65:
        items = ", ".join("'%s': %s(f.%s)" % (nm, rnd.choice(CUM + ["first", "last", "median"]), anyc()) for nm in rnd.sample(["a", "b", "c"], rnd.randint(1, 3)))
        return "DT[:, {%s}, %s]" % (items, bys)
    if kind < 0.8:
        v = anyc()
        return "DT[f.%s %s %s, :][:, [count(), %s(f.%s), %s(f.%s)], %s]" % (
            v, rnd.choice([">", ">=", "<", "<=", "==", "!="]), rnd.choice(["-1.5", "0.25", "-0.0", "2", "-3", "1e6", "-7.75"]),
            rnd.choice(RED), anyc(), rnd.choice(RED), anyc(), bys)
    if kind < 0.9:
        a, b = anyc(), anyc()
        return "DT[:, [%s(f.%s, f.%s), %s(f.%s, f.%s), nunique(f.%s), sd(f.%s)], %s]" % (
            rnd.choice(["cov", "corr"]), a, b, rnd.choice(["cov", "corr"]), b, b, anyc(), anyc(), bys)
    return "DT[:, [f.%s, %s(f.%s), cumcount(%s), %s(f.%s)], %s]" % (anyc(), rnd.choice(CUM), anyc(), rnd.choice(["", "reverse=True"]),
                                                                   rnd.choice(RED), anyc(), bys)


def rand_query6(names):
    """no-by mixtures (reducers broadcast next to plain columns), f[:] forms of every operator, duplicate sort
    columns, by()+sort() with na_position, dict j whose names collide with columns, 3 by-columns"""
    keys = [nm for nm in names if nm.startswith("k")]
    vals = [nm for nm in names if nm.startswith("v")]
    anyc = lambda: rnd.choice(names)
    kk = rnd.sample(names, rnd.randint(1, min(3, len(names))))
    bys = "by(%s)" % ", ".join("f.%s" % k for k in kk)
    kind = rnd.random()
    if kind < 0.15:
        return "DT[:, [f.%s, %s(f.%s), %s(f.%s)]]" % (anyc(), rnd.choice(RED), anyc(), rnd.choice(CUM), anyc())
    if kind < 0.3:
        return "DT[:, %s(f[:])%s]" % (rnd.choice(RED + CUM), rnd.choice(["", ", " + bys]))
    if kind < 0.4:
        c = anyc()
        return "DT[:, :, sort(f.%s, f.%s, f.%s%s)]" % (c, anyc(), c, rnd.choice(["", ", na_position='last'", ", reverse=True"]))
    if kind < 0.55:
        return "DT[:, [%s(f.%s), %s(f.%s), f.%s], %s, sort(f.%s, na_position='%s')]" % (
            rnd.choice(["first", "last", "cumsum", "cummin", "sum", "median"]), anyc(), rnd.choice(["first", "last", "max", "count"]), anyc(), anyc(),
            "by(f.%s)" % rnd.choice(keys), anyc(), rnd.choice(["first", "last"]))
    if kind < 0.7:
        nm = rnd.sample(names + ["count", "C0", "x"], rnd.randint(1, 3))
        items = ", ".join("'%s': %s(f.%s)" % (n_, rnd.choice(RED + CUM), anyc()) for n_ in nm)
        return "DT[:, {%s}, %s]" % (items, "by(f.%s)" % rnd.choice(keys))
    if kind < 0.85:
        return "DT[:, [%s(f.%s), %s(f.%s), count(), %s(f.%s)], %s]" % (rnd.choice(RED), anyc(), rnd.choice(RED), anyc(), rnd.choice(RED), anyc(), bys)
    return "DT[f.%s %s %s, :][:, :, sort(f.%s%s)][:, [%s(f.%s), f.%s], by(f.%s)]" % (
        anyc(), rnd.choice(["==", "!=", ">", "<"]), rnd.choice(["True", "False", "0", "1"]), anyc(), rnd.choice(["", ", reverse=True"]),
        rnd.choice(CUM + ["first", "last"]), anyc(), anyc(), rnd.choice(keys))


def keyed(F, *names):
    G = F.copy()
    G.key = names if len(names) > 1 else names[0]
    return G


def batch3():
    """set functions, unique, keys and natural joins over several frames"""
    from datatable import union, intersect, setdiff, symdiff, unique, join
    cases = []
    rnd.seed(3000)
    while len(cases) < 300:
        kind = rnd.random()
        frames = {}
        if kind < 0.5:
            # set functions over 1..4 single-column frames, stypes may differ (rbind up-casts)
            nsrc = rnd.randint(1, 4)
            base = rnd.choice([2, 3, 4, 5, 7, 4, 5])
            for i in range(nsrc):
                st = base if rnd.random() < 0.7 else rnd.choice([2, 3, 4, 5, 6, 7])
                n = rnd.choice([0, 1, 5, 30])
                frames["S%d" % i] = {"c%d" % i: (rand_col(st, n, "key"), st)}
            fn = rnd.choice(["union", "intersect", "setdiff", "symdiff"])
            q = "%s(%s)" % (fn, ", ".join(sorted(frames)))
        elif kind < 0.6:
            ncol = rnd.randint(1, 3)
            st = rnd.choice([2, 3, 4, 5, 7])
            n = rnd.choice([1, 6, 25])
            frames["DT"] = {"c%d" % i: (rand_col(st if rnd.random() < 0.7 else rnd.choice([4, 5, 7]), n, "key"), None) for i in range(ncol)}
            q = "unique(DT)"
        else:
            nk = rnd.choice([1, 1, 2])
            jst = [rnd.choice([2, 3, 4, 5, 4, 5, 7]) for _ in range(nk)]
            xst = [st if rnd.random() < 0.6 else rnd.choice([2, 3, 4, 5, 7]) for st in jst]
            nj, nx = rnd.choice([0, 1, 8, 30]), rnd.choice([0, 1, 10, 60])
            tuples = set()
            for _ in range(nj):
                tuples.add(tuple((None if rnd.random() < 0.05 else (rnd.randint(-6, 6) if st != 7 else rnd.randint(-12, 12) / 2.0)) for st in jst))
            tuples = list(tuples)
            rnd.shuffle(tuples)
            J = {"K%d" % k: ([t[k] for t in tuples], jst[k]) for k in range(nk)}
            J["W"] = ([rnd.randint(-99, 99) for _ in tuples], 4)
            J["Z"] = ([rnd.random() < 0.5 for _ in tuples], 1)
            X = {"K%d" % k: ([None if rnd.random() < 0.05 else (rnd.randint(-7, 7) if xst[k] != 7 else rnd.randint(-14, 14) / 2.0) for _ in range(nx)], xst[k]) for k in range(nk)}
            X["P"] = (list(range(nx)), 4)
            frames = {"X": X, "J": J}
            names = ", ".join("'K%d'" % k for k in range(nk))
            q = rnd.choice(["X[:, :, join(keyed(J, %s))]", "keyed(J, %s)", "X[:, ['P', 'W'], join(keyed(J, %s))]"]) % names
        # build reference frames
        env = {}
        ok = True
        for fname, spec in frames.items():
            cols, sts, nms = [], [], []
            for nm, (vals, st) in spec.items():
                nms.append(nm); cols.append(vals); sts.append(st)
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    F = dt.Frame(cols, names=nms, stypes=[ST[s] if s else None for s in sts]) if any(sts) else dt.Frame(cols, names=nms)
            except Exception:
                ok = False
                break
            if any(s.value not in ST for s in F.stypes):
                ok = False
                break
            env[fname] = F
        if not ok:
            continue
        ns = dict(env, union=union, intersect=intersect, setdiff=setdiff, symdiff=symdiff, unique=unique, join=join, keyed=keyed)
        rec = {"frames": {fname: {nm: {"values": clean([F[:, nm].to_list()[0]])[0], "stype": F.stypes[i].value} for i, nm in enumerate(F.names)}
                          for fname, F in env.items()}, "query": q}
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                R = eval(q, ns)
            if any(s.value not in ST for s in R.stypes):
                continue
            rec.update({"names": list(R.names), "stypes": [s.value for s in R.stypes], "columns": clean(R.to_list()), "nkeys": len(R.key)})
        except Exception as e:
            rec.update({"error": type(e).__name__, "message": str(e)[:200]})
        cases.append(rec)
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "frame_fuzz3.json")
    json.dump({"cases": cases}, open(path, "w"))
    from collections import Counter
    print("wrote %s: %d cases; errors: %s" % (path, len(cases), Counter(c.get("error") for c in cases if "error" in c)))


def clean(cols):
    out = []
    for c in cols:
        cc = []
        for x in c:
            if isinstance(x, float):
                if x != x:
                    x = None
                elif math.isinf(x):
                    x = "inf" if x > 0 else "-inf"
                elif x == 0 and math.copysign(1, x) < 0:
                    x = "-0.0"
            cc.append(x)
        out.append(cc)
    return out


def main(batch=1):
    global NEG_ZERO
    NEG_ZERO = batch >= 5
    frames, queries = {}, []
    seed = {1: 0, 2: 100000, 4: 200000, 5: 300000, 6: 400000}[batch]
    gen = {1: rand_query, 2: rand_query2, 4: rand_query4, 5: rand_query5, 6: rand_query6}[batch]
    while len(queries) < {1: 700, 2: 400, 4: 400, 5: 500, 6: 500}[batch]:
        seed += 1
        spec = make_frame(seed)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            DT = dt.Frame(spec)
        # stypes as the reference inferred them (all-None columns become void: skip such frames)
        if any(s.value not in ST for s in DT.stypes):
            continue
        fname = "fz%d" % seed
        frames[fname] = {nm: {"values": clean([DT[:, nm].to_list()[0]])[0], "stype": DT.stypes[i].value} for i, nm in enumerate(DT.names)}
        for _ in range(4):
            q = gen(list(DT.names))
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    R = eval(q)
                if any(s.value not in ST for s in R.stypes):
                    continue
                queries.append({"frame": fname, "query": q, "names": list(R.names), "stypes": [s.value for s in R.stypes],
                                "columns": clean(R.to_list())})
            except Exception as e:      # the reference refuses the query: record its exception type
                queries.append({"frame": fname, "query": q, "error": type(e).__name__})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), {1: "frame_fuzz.json", 2: "frame_fuzz2.json", 4: "frame_fuzz4.json", 5: "frame_fuzz5.json", 6: "frame_fuzz6.json"}[batch])
    json.dump({"frames": frames, "queries": queries}, open(path, "w"))
    nerr = len([q for q in queries if "error" in q])
    print("wrote %s: %d frames, %d queries (%d refused by the reference)" % (path, len(frames), len(queries), nerr))
    from collections import Counter
    print(Counter(q.get("error") for q in queries if "error" in q))


if __name__ == "__main__":
    b = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    if b == 3:
        batch3()
    else:
        main(b)
