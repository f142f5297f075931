import csv,sys
for c in ("FETCH_SIZE","WRITE_SIZE"):
    rows=list(csv.DictReader(open("gpurun_out/prof5/pmc_%s.csv"%c)))
    by={}
    for r in rows:
        if r["Counter_Name"]==c: by.setdefault(r["Dispatch_Id"],0.0); by[r["Dispatch_Id"]]+=float(r["Counter_Value"])
    vals=sorted(by.values())
    top=vals[-24:]
    print(c, len(vals), "mean of the 24 largest launches, KiB", sum(top)/len(top))
