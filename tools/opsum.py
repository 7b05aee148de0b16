import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
prev=None
for i,r in enumerate(rows):
    t=r['tag']
    if t=='groupnorm' and prev is not None:
        print(f"{prev['tag']:14s} M={prev['M']:>5s} N={prev['N']:>5s} K={prev['K']:>6s} tile={prev['tile']:>2s} sk={prev['splitk']:>2s} {float(prev['ms'])*1e3:6.1f}us -> gn {float(r['ms'])*1e3:5.1f}us")
    prev=r
print("total", sum(float(r['ms']) for r in rows))
