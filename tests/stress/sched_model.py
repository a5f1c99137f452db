"""Scheduling model of the k_match_pairs launch on BASELINE C1 (CPU only: python tests/stress/sched_model.py).
Work items = 64-row groups; cost model: an item takes 0.40 ms x (25 % fixed + 75 % proportional to the targets its
hull makes it walk), independent of how full its SIMD is (measured); greedy list scheduling onto 6144
wave slots in launch order.  Used to rank the options of DESIGN.md section 9.1 -- a model, not a measurement."""
import heapq, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import tests.test_culling_math as T
from line3dpp_amd.scene import make_config
sc = make_config("C1"); V={v.cam:v for v in sc.views}
pairs = sc.pair_tests()[1]
def items_for(order_fn, split=False):
    items=[]   # (pair, walked targets)
    for pi,(s,t) in enumerate(pairs):
        f=T._cull_forms(T._fundamental(V[s],V[t]),V[s].width,V[s].height,V[t].width,V[t].height)
        slo,shi,tlo,thi=T._bands(f,V[s].segs,V[t].segs)
        order=order_fn(slo,shi,V[t])
        for g0 in range(0,len(slo),64):
            idx=order[g0:g0+64]; lo=slo[idx].min(); hi=shi[idx].max()
            items.append((pi,int(((tlo<=hi)&(thi>=lo)).sum())))
    return items
def lo_order(slo,shi,vt): return np.argsort(slo,kind='stable')
def wide_first(slo,shi,vt):
    w=shi-slo; ref=(vt.width+vt.height)/2
    cls=np.where(w>ref/16,0,1)
    return np.lexsort((slo,cls))
def makespan(durs,P=6144):
    h=[0.0]*min(P,len(durs)); heapq.heapify(h); end=0
    for d in durs:
        t=heapq.heappop(h); heapq.heappush(h,t+d); end=max(end,t+d)
    return end
base=items_for(lo_order); w0=np.array([x[1] for x in base],float)
fixed=0.25   # fraction of an average item that does not scale with walked targets (prologue, epilogue, chunk tests)
unit=0.40/ (fixed*w0.mean()+ (1-fixed)*w0.mean())*1.0
dur=lambda w: 0.40*(fixed + (1-fixed)*w/w0.mean())
print("S0 current: items %d mean walked %.0f max %.0f  makespan %.3f ms  (sum/P %.3f)"%(len(w0),w0.mean(),w0.max(),makespan(dur(w0)),dur(w0).sum()/6144))
print("S0 + global LPT: makespan %.3f"%makespan(np.sort(dur(w0))[::-1]))
wf=items_for(wide_first); w1=np.array([x[1] for x in wf],float)
print("S1 wide-first classes: mean walked %.0f max %.0f makespan %.3f (sum/P %.3f)"%(w1.mean(),w1.max(),makespan(dur(w1)),dur(w1).sum()/6144))
# S2: split long items into k sub-items over target chunks, each with the fixed part again (x0.5: shared tables, own prologue)
narrow=np.median(w1)
d2=[]
for w in w1:
    k=max(1,int(round(w/narrow)))
    d2+= [0.40*(fixed*(1.0 if k==1 else 0.6) + (1-fixed)*(w/k)/w0.mean())]*k
d2=np.array(d2)
print("S2 wide-first + split long items: sub-items %d makespan %.3f (sum/P %.3f)"%(len(d2),makespan(d2),d2.sum()/6144))
print("S2 + LPT: %.3f"%makespan(np.sort(d2)[::-1]))
# S3: every item split in 2 (the current WPG=2), on current order
d3=np.repeat(0.40*(fixed*0.6+(1-fixed)*(w0/2)/w0.mean()),2)
print("S3 current order, all items in two halves: makespan %.3f (sum/P %.3f)"%(makespan(d3),d3.sum()/6144))
