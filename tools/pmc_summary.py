"""Summarise a rocprofv3 rocpd sqlite database: per kernel, per counter (summed over dims), mean over dispatches."""
import sqlite3, sys, collections

def main():
    for f in sys.argv[1:]:
        db = sqlite3.connect(f)
        cur = db.cursor()
        rows = cur.execute("select dispatch_id, kernel_name, counter_name, sum(value), max(duration) from counters_collection "
                           "group by dispatch_id, kernel_name, counter_name").fetchall()
        agg = collections.defaultdict(list)
        dur = collections.defaultdict(list)
        for did, k, c, v, d in rows:
            agg[(k, c)].append(v)
            dur[k].append(d)
        print("==", f)
        kernels = sorted({k for k, _ in agg})
        for k in kernels:
            if "rocclr" in k or "at::" in k or "elementwise" in k:
                continue
            print(f"  {k[:90]}  n={len(set(dur[k]))} mean_dur_us={sum(dur[k])/len(dur[k])/1e3:.1f}")
            for (kk, c), vs in sorted(agg.items()):
                if kk == k:
                    print(f"      {c:28s} {sum(vs)/len(vs):16.0f}")


if __name__ == "__main__":
    main()
