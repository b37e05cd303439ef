"""Worker of tests/test_gpu_intree.py: evaluates a fixed list of `DT[i, j, by()]` statements on ONE build of the
reference -- the patched in-tree build (integration/_dt_hip, its group() and reducer columns on libdthip.so) or the
unmodified one (oracle/_ref) -- and stores every result column in an .npz file.  Two builds of a module called
`datatable` cannot live in one process, hence a process per build.

    python tests/intree_worker.py <package dir> <out.npz> <rows>

Test infrastructure: nothing of the product imports it."""
import sys

import numpy as np


def frames(n):
    rng = np.random.default_rng(20260930)
    k = rng.integers(0, max(2, n // 100), n).astype(np.int64)
    k[rng.random(n) < 0.01] = np.iinfo(np.int64).min                  # NA keys
    a = rng.integers(0, 50, n).astype(np.int32)
    v = rng.standard_normal(n)
    v[rng.random(n) < 0.02] = np.nan                                  # NA values
    w = rng.standard_normal(n).astype(np.float32)
    i = rng.integers(-1000, 1000, n).astype(np.int32)
    i[rng.random(n) < 0.02] = np.iinfo(np.int32).min
    x = rng.standard_normal(n)
    return {"k": k, "a": a, "v": v, "w": w, "i": i, "x": x}


def queries(dt):
    f, by = dt.f, dt.by
    red = [dt.sum, dt.mean, dt.min, dt.max, dt.count]
    return [
        # group() -> RowIndex -> reducers gathering through it (the view of every value column is peeled)
        ("by_k", lambda D: D[:, [r(f[c]) for r in red for c in ("v", "w", "i")], by(f.k)]),
        # two keys
        ("by_a_k", lambda D: D[:, [dt.sum(f.v), dt.count(f.i), dt.count()], by(f.a, f.k)]),
        # filter -> view -> groupby: the reducers' views are COMPOSED RowIndexes (not the one S-grp kept)
        ("filter_by_k", lambda D: D[f.x > 0, :][:, [dt.sum(f.v), dt.min(f.i), dt.max(f.w), dt.mean(f.i)], by(f.k)]),
        # no by(): one group over the stored column (nothing to peel)
        ("no_by", lambda D: D[:, [dt.sum(f.v), dt.min(f.v), dt.max(f.i), dt.count(f.w)]]),
        # sort() + by(): ordering by all columns, groups by the leading one
        ("by_a_sort_i", lambda D: D[:, [dt.sum(f.i), dt.max(f.v)], by(f.a), dt.sort(f.i)]),
    ]


def main():
    pkg, out, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    sys.path.insert(0, pkg)
    import datatable as dt
    assert dt.__file__.startswith(pkg), (dt.__file__, pkg)
    dt.options.progress.enabled = False
    cols = frames(n)
    D = dt.Frame(cols)
    res = {}
    for name, q in queries(dt):
        R = q(D)
        res[name + "/names"] = np.array(list(R.names))
        res[name + "/stypes"] = np.array([s.name for s in R.stypes])
        for c in range(R.ncols):
            a = R[:, c].to_numpy()
            if isinstance(a, np.ma.MaskedArray):                      # integer columns with NAs come out masked
                res["%s/%d.na" % (name, c)] = np.ma.getmaskarray(a).ravel()
                a = a.filled(0)
            res["%s/%d" % (name, c)] = np.asarray(a).ravel()
    np.savez(out, **res)


if __name__ == "__main__":
    main()
