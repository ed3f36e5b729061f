#!/bin/bash
# GPU box: kernel trace + HBM counters of the Survive derivation alone (tools/sv_box_bench.py: whole record, box, box in batches of 16)
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/prof_${1:-r04}_survive
mkdir -p $OUT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 tools/sv_box_bench.py --n 160 > $OUT/kt.log 2>&1 || tail -5 $OUT/kt.log
for grp in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/pmc_$grp -- python3 tools/sv_box_bench.py --n 160 > $OUT/pmc_$grp.log 2>&1 || tail -5 $OUT/pmc_$grp.log
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out=sys.argv[1]
for f in glob.glob(out+"/kt/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "survive" in r["Name"]:
            print("kernel_stats", r["Name"][:60], "calls", r["Calls"], "avg_us %.1f min_us %.1f max_us %.1f" % (float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3, float(r["MaxNs"])/1e3))
# per-dispatch: group by grid size (whole / box / box batch) using the kernel trace
rows=[]
for f in glob.glob(out+"/kt/*/*kernel_trace.csv"):
    rows+= [r for r in csv.DictReader(open(f)) if "survive_kill9_rows" in r["Kernel_Name"]]
g=collections.defaultdict(list)
for r in rows:
    key=(r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))
    g[key].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(g.items(), key=lambda kv: -len(kv[1])):
    v=sorted(v); print("grid", k, "dispatches", len(v), "median_us %.1f" % v[len(v)//2])
for c in ("FETCH_SIZE","WRITE_SIZE"):
    gg=collections.defaultdict(list)
    for f in glob.glob(out+"/pmc_%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if "survive_kill9_rows" in r["Kernel_Name"]:
                gg[(r.get("Grid_Size_X") or r.get("Grid_Size"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))].append(float(r["Counter_Value"]))
    for k,v in gg.items():
        print(c, "grid", k, "dispatches", len(v), "mean_KiB %.0f" % (sum(v)/len(v)))
PY
