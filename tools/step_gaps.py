import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
idx=[i for i,r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
a,b=idx[-2]+1,idx[-1]+1
step=sorted(rows[a:b],key=lambda r:int(r['Start_Timestamp']))
t0=int(step[0]['Start_Timestamp']); end=max(int(r['End_Timestamp']) for r in step)
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in step)
print("span %.3f ms busy %.3f ms launches %d"%((end-t0)/1e6,busy/1e6,len(step)))
gaps=[]
for p,q in zip(step[:-1],step[1:]):
    g=int(q['Start_Timestamp'])-int(p['End_Timestamp'])
    gaps.append((g,(int(p['End_Timestamp'])-t0)/1e6,re.sub(r'\(.*','',p['Kernel_Name'])[-40:],re.sub(r'\(.*','',q['Kernel_Name'])[-40:]))
tot=sum(g for g,_,_,_ in gaps if g>0)
print("total gap %.3f ms; gaps>5us: %d"%(tot/1e6,sum(1 for g,_,_,_ in gaps if g>5000)))
for g,t,pn,qn in sorted(gaps,reverse=True)[:12]: print("%8.1f us at %.2f ms  %s -> %s"%(g/1e3,t,pn,qn))
