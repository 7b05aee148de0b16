import csv,sys,collections
def load(f):
    rows=list(csv.DictReader(open(f)))
    d=collections.defaultdict(list)
    for r in rows: d[(r['tag'],r['M'],r['N'],r['K'])].append(float(r['ms'])*1e3)
    return rows,d
ra,da=load(sys.argv[1]); rb,db=load(sys.argv[2])
fam=collections.defaultdict(lambda:[0.0,0.0])
for r in ra: fam[r['tag'].split('.')[-1]][0]+=float(r['ms'])*1e3
for r in rb: fam[r['tag'].split('.')[-1]][1]+=float(r['ms'])*1e3
for k,v in sorted(fam.items(), key=lambda kv:-kv[1][0])[:14]: print(f"{k:12s} {v[0]:8.1f} {v[1]:8.1f}  {v[1]-v[0]:+7.1f}")
print("total", sum(v[0] for v in fam.values()), sum(v[1] for v in fam.values()))
