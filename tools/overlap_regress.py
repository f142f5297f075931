#!/usr/bin/env python3
"""What does a kernel's duration under load depend on?  Reads the un-serialised kernel trace of the headline (tools/profile_round.sh part
`trace`: gpurun_out/prof<N>/kernel_trace.csv.gz), and for each named kernel regresses the duration of its launches in the timed region on
how much of each launch was overlapped by launches of OTHER queues of four classes -- the level-0 search, the coarse levels, the
in-loop filter sweeps, other streaming kernels -- and prints the mean duration by share of level-0 overlap.
usage: python3 tools/overlap_regress.py gpurun_out/prof6/kernel_trace.csv.gz "k_quant_level4<1>" "k_predict_w<1>" ..."""
import csv, gzip, io, collections, sys
import numpy as np
def short(n):
    return n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("dsv2::", "")
tr=[]
for r in csv.DictReader(io.TextIOWrapper(gzip.open(sys.argv[1]))):
    tr.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]))
tr.sort()
l0=[r for r in tr if r[2]=="k_hme_rows_l0"]
lo,hi=l0[len(l0)-24-384][0], l0[-25][1]
tr=[r for r in tr if r[0]>=lo and r[1]<=hi]
# classes
def cls(n):
    if n.startswith("k_hme_rows_l0"): return "l0"
    if n.startswith("k_hme_rows_lx"): return "lx"
    if "filter" in n: return "filt"
    return "stream"
classes=["l0","lx","filt","stream"]
# build per-class interval arrays
iv={c:[] for c in classes}
for s,e,n,q in tr: iv[cls(n)].append((s,e,q))
for c in classes: iv[c].sort()
starts={c:np.array([x[0] for x in iv[c]]) for c in classes}
ends={c:np.array([x[1] for x in iv[c]]) for c in classes}
qs={c:np.array([hash(x[2]) for x in iv[c]]) for c in classes}
def overlap(s,e,q,c):
    # total overlapped time of class c intervals (other queues) with [s,e], normalised by (e-s)
    S,E,Q=starts[c],ends[c],qs[c]
    m=(S<e)&(E>s)&(Q!=hash(q))
    return float(np.sum(np.minimum(E[m],e)-np.maximum(S[m],s)))/(e-s)
for K in sys.argv[2:]:
    rows=[(s,e,q) for s,e,n,q in tr if n==K]
    if not rows: continue
    X=[];Y=[]
    for s,e,q in rows[::max(1,len(rows)//400)]:
        X.append([1.0]+[overlap(s,e,q,c) for c in classes]); Y.append((e-s)/1e3)
    X=np.array(X);Y=np.array(Y)
    # model: duration*(1) ... fit rate: 1/duration? use duration = b0 + sum b_c f_c * duration -> duration*(1 - sum b f) = b0 ; fit log? simple linear on Y
    coef,res,rk,sv=np.linalg.lstsq(X,Y,rcond=None)
    print(K,"n",len(Y),"mean us %.0f"%Y.mean(),"min %.0f max %.0f"%(Y.min(),Y.max()))
    print("   mean overlap (x concurrent launches):",{c:round(float(X[:,i+1].mean()),2) for i,c in enumerate(classes)})
    print("   linear fit us: base %.0f "%coef[0]+" ".join("%s %+.0f"%(c,coef[i+1]) for i,c in enumerate(classes)))
    # bucket by l0 overlap
    for lo_,hi_ in ((0,0.05),(0.05,0.5),(0.5,0.95),(0.95,2)):
        m=(X[:,1]>=lo_)&(X[:,1]<hi_)
        if m.sum(): print("   l0 overlap %.2f-%.2f: n %d mean %.0f us  (filt %.2f stream %.2f lx %.2f)"%(lo_,hi_,m.sum(),Y[m].mean(),X[m,3].mean(),X[m,4].mean(),X[m,2].mean()))
