import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
out=[]
for r in rows:
    if 'k_wgrad_h2<' in r['Name']:
        out.append(f"{r['Name'][17:29]}={float(r['AverageNs'])/1e3:.0f}")
print(sys.argv[2], ' '.join(sorted(out)))
