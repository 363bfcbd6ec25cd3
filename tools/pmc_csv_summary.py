"""Per kernel (name contains any of the given substrings): mean over dispatches of every counter found in the
*_counter_collection.csv files under a directory tree, and the kernel-trace averages from *_kernel_stats.csv.
   python tools/pmc_csv_summary.py <dir> [substring ...]"""
import collections, csv, os, sys

def short(name):
    return name.replace("void ", "").replace("sdrhip::(anonymous namespace)::", "").split("(")[0]

def main():
    root = sys.argv[1]
    subs = sys.argv[2:] or ["k_"]
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    durs = collections.defaultdict(list)
    for d, _, fs in os.walk(root):
        for f in fs:
            p = os.path.join(d, f)
            if f.endswith("counter_collection.csv"):
                seen = {}
                for r in csv.DictReader(open(p)):
                    k = short(r["Kernel_Name"])
                    if not any(s in k for s in subs):
                        continue
                    per[(k, p, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
                    seen[(k, r["Dispatch_Id"])] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
                for (k, _), v in seen.items():
                    durs[(k, os.path.basename(d))].append(v)
            elif f.endswith("kernel_stats.csv"):
                for r in csv.DictReader(open(p)):
                    k = short(r["Name"])
                    if any(s in k for s in subs):
                        print(f"trace  {k[:80]:80s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}  max {float(r['MaxNs'])/1e3:9.1f}")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for (k, p, _), cs in per.items():
        for c, v in cs.items():
            agg[k][c].append(v)
    for k in sorted(agg):
        print("pmc   ", k)
        for (kk, sub), v in sorted(durs.items()):
            if kk == k:
                print(f"         dur[{sub}] {sum(v)/len(v)/1e3:.1f} us over {len(v)} dispatches")
        for c in sorted(agg[k]):
            v = agg[k][c]
            print(f"         {c:28s} {sum(v)/len(v):18.1f}")

if __name__ == "__main__":
    main()
