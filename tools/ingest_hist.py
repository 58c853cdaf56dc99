import sqlite3, sys, glob
f=glob.glob(sys.argv[1]+'/**/*.db', recursive=True)[0]
c=sqlite3.connect(f)
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
rows=list(c.execute(f"select s.kernel_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
import collections
dur=[(e-s)/1000 for n,s,e in rows if 'pool_ingest' in n]
dur.sort()
n=len(dur)
print('ingest n',n,'p10',dur[n//10],'p50',dur[n//2],'p90',dur[9*n//10],'p99',dur[99*n//100],'max',dur[-1],'mean',sum(dur)/n)
# gap analysis: time between publish end and next ingest end
pub=[(s,e) for n_,s,e in rows if 'pool_publish' in n_]
ing=[(s,e) for n_,s,e in rows if 'pool_ingest' in n_]
gaps=[]
j=0
for ps,pe in pub:
    while j < len(ing) and ing[j][0] < pe: j+=1
    if j < len(ing): gaps.append((ing[j][1]-pe)/1000)
gaps.sort(); m=len(gaps)
if m: print('publish->ingest done n',m,'p10',gaps[m//10],'p50',gaps[m//2],'p90',gaps[9*m//10],'mean',sum(gaps)/m)
# idle gaps between consecutive kernels
idle=[]
for (n1,s1,e1),(n2,s2,e2) in zip(rows,rows[1:]):
    idle.append(((s2-e1)/1000, n1[:40], n2[:40]))
tot=sum(x[0] for x in idle if x[0]>0)
print('total idle between kernels (us)', tot, 'of span', (rows[-1][2]-rows[0][1])/1000)
agg=collections.defaultdict(float)
for g,a,b in idle:
    if g>0: agg[(a,b)]+=g
for k,v in sorted(agg.items(), key=lambda x:-x[1])[:8]: print(round(v), k)
