import csv, collections, glob, statistics as st, sys
d=sys.argv[1]
f=glob.glob(f"{d}/*/*counter_collection.csv")[0]
tr={r["Dispatch_Id"]:r for r in csv.DictReader(open(glob.glob(f"{d}/*/*kernel_trace.csv")[0]))}
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "fk::" not in r["Kernel_Name"]: continue
    name=r["Kernel_Name"].split("(")[0][-30:]+" g="+r["Grid_Size"]
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    t=tr.get(r["Dispatch_Id"])
    if t and r["Counter_Name"]=="SQ_WAVES": agg[name]["dur"].append(int(t["End_Timestamp"])-int(t["Start_Timestamp"]))
for n,c in sorted(agg.items()):
    m=lambda k: st.median(c[k]) if c[k] else 0
    w=max(m("SQ_WAVES"),1); wc=max(m("SQ_WAVE_CYCLES"),1)
    print(f"{n:44s} dur={m('dur')/1e3:6.1f}us waves={w:6.0f} " + " ".join(f"{k}={m(k):.3g}" for k in sorted(c) if k not in ("dur","SQ_WAVES")) + f" | cyc/wave={wc*4/w:.0f}")
