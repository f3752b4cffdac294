import csv,sys,re,collections
def load(p):
    rows=list(csv.DictReader(open(p)))
    idx=[i for i,r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
    out=collections.Counter()
    n=0
    for s in range(len(idx)-3,len(idx)-1):   # two steady steps
        a,b=idx[s]+1,idx[s+1]+1
        for r in rows[a:b]:
            k=re.sub(r'\(.*','',r['Kernel_Name']).replace('void ','')[:70]+" g"+r['Grid_Size_X']
            out[k]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        n+=1
    return {k:v/n for k,v in out.items()}
a,b=load(sys.argv[1]),load(sys.argv[2])
keys=sorted(set(a)|set(b),key=lambda k:-abs(b.get(k,0)-a.get(k,0)))
print("total %.1f -> %.1f us"%(sum(a.values()),sum(b.values())))
for k in keys[:14]: print("%8.1f %8.1f %+8.1f  %s"%(a.get(k,0),b.get(k,0),b.get(k,0)-a.get(k,0),k))
