import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last pass: find the last k_layer_optics_lin kernel (start of a step) 
idx=[i for i,r in enumerate(rows) if "k_layer_optics_lin" in r["Kernel_Name"]]
print("optics_lin launches at", idx)
start=idx[-1]
sel=rows[start:]
t0=int(sel[0]["Start_Timestamp"])
end=max(int(r["End_Timestamp"]) for r in sel)
print("pass span ms", (end-t0)/1e6, "kernels", len(sel))
import collections
bins=collections.defaultdict(lambda: collections.defaultdict(float))
names=collections.defaultdict(lambda: collections.defaultdict(float))
for r in sel:
    s=int(r["Start_Timestamp"])-t0; e=int(r["End_Timestamp"])-t0; q=r["Queue_Id"]
    b=s//1000000
    bins[b][q]+=(e-s)/1e3
    key="dbl" if "k_dbl" in r["Kernel_Name"] else ("ia" if "k_ia128" in r["Kernel_Name"] else "other")
    names[b][key+q]+= (e-s)/1e3
for b in sorted(bins):
    print(b, {q:round(v) for q,v in sorted(bins[b].items())}, {k:round(v) for k,v in sorted(names[b].items())})
