import gzip, csv, re, collections, sys
rows=list(csv.DictReader(gzip.open('gpurun_out/trace/kernel_trace.csv.gz','rt')))
ks=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in rows)
starts=[k[0] for k in ks if 'rng_advance' in k[2]]
sel=[k for k in ks if starts[-3]<=k[0]<starts[-2]]
agg=collections.defaultdict(lambda:[0,0.0])
for s,e,n in sel:
    n=re.sub(r'^_Z\d+','',n)[:60]
    a=agg[n]; a[0]+=1; a[1]+=(e-s)/1e3
tot=sum(a[1] for a in agg.values())
print("step %.2f ms, %d kernels, sum of durations %.2f ms" % ((starts[-2]-starts[-3])/1e6, len(sel), tot/1e3))
for n,(c,t) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:28]:
    print("%4d x %7.1f us = %7.1f us  %s" % (c, t/c, t, n))
