"""oracle/ref.py -- the UNMODIFIED reference (h2oai/datatable) as built by oracle/build_ref.sh into
oracle/_ref/.  TEST INFRASTRUCTURE ONLY: imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__ -- never by datatable_amd/.

    from oracle import ref
    dt = ref.load()                  # the reference's `datatable` module, or None if oracle/_ref is absent
    ref.groupby_agg(...)             # DT[:, {reducers}, by(keys)] evaluated by the reference's own CPU path

What runs is the reference's own thread pool + MSD radix sort (src/core/parallel/parallel_for_static.h:113-165,
src/core/sort.cc:1128-1353) and its reducer columns (src/core/column/{sumprod,mean,minmax,count}.h): this is
the CPU path `north_star` asks to be timed beside the GPU path.
"""
import os
import sys
import time

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(_HERE, "_ref")

_dt = None


def available():
    lib = os.path.join(REF_DIR, "datatable", "lib")
    return os.path.isdir(lib) and any(n.startswith("_datatable") and n.endswith(".so") for n in os.listdir(lib))


def load():
    """import the reference build in oracle/_ref as `datatable` (None when it was never built)"""
    global _dt
    if _dt is not None:
        return _dt
    if not available():
        return None
    if REF_DIR not in sys.path:
        sys.path.insert(0, REF_DIR)
    import datatable as dt
    if not os.path.abspath(dt.__file__).startswith(REF_DIR):
        raise RuntimeError("`datatable` resolved to %s, not to oracle/_ref" % dt.__file__)
    dt.options.progress.enabled = False
    _dt = dt
    return dt


def provenance():
    p = os.path.join(REF_DIR, "PROVENANCE.txt")
    return open(p).read().strip().splitlines() if os.path.exists(p) else []


def set_threads(n):
    """Both thread counts of the path: dt.options.nthreads (the pool every parallel_for of the reducers uses) and
    sort.nthreads (group()'s own team size; its default is the hardware concurrency, src/core/sort.cc:264, and it
    does NOT follow options.nthreads).  The setter stores the value through a uint8_t cast (sort.cc:337), so 256
    becomes 0 and group() then divides by it (sort.cc:918-919: SIGFPE, observed on the 256-thread GPU box):
    sort.nthreads is clamped to 255 here."""
    dt = load()
    dt.options.nthreads = int(n)
    dt.options.sort.nthreads = max(1, min(int(n), 255))
    return dt.options.nthreads


def frame(columns):
    """{name: numpy array} -> dt.Frame, zero-copy (src/core/py_buffers.cc:54-113)"""
    return load().Frame(columns)


def groupby_agg(cols, keys, aggs, nthreads=None, reps=1):
    """The reference's DT[:, [op(f.col), ...], by(keys)].

    cols: dict name -> numpy array (zero-copy into the Frame); keys: list of names;
    aggs: list of (op, name | None) with op in sum/mean/min/max/count (None: count()).
    A fresh Frame per repetition so that cached column stats (min/max) are recomputed every time
    (SURVEY 8(d)).  Returns (result dt.Frame, best seconds)."""
    dt = load()
    from datatable import f, by
    if nthreads is not None:
        set_threads(nthreads)
    fn = {"sum": dt.sum, "mean": dt.mean, "min": dt.min, "max": dt.max, "count": dt.count}
    j = [fn[op](f[c]) if c is not None else dt.count() for op, c in aggs]
    best, res = None, None
    for _ in range(reps):
        DT = dt.Frame(cols)
        t0 = time.perf_counter()
        res = DT[:, j, by(*[f[k] for k in keys])]
        t = time.perf_counter() - t0
        best = t if best is None or t < best else best
        del DT
    return res, best


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def filter_group_rows(cols, xname, keyname, nthreads=None):
    """BASELINE config 5 in the only form the reference supports: V = DT[f.x > 0, :]; V[:, :, by(f.k)],
    the result materialised (the GPU path returns materialised columns; the reference's result columns are views until
    someone reads them).  Returns (result Frame, seconds)."""
    dt = load()
    from datatable import f, by
    if nthreads is not None:
        set_threads(nthreads)
    DT = dt.Frame(cols)
    t0 = time.perf_counter()
    V = DT[f[xname] > 0, :]
    R = V[:, :, by(f[keyname])]
    R.materialize()
    return R, time.perf_counter() - t0
