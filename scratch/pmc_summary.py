import sqlite3,glob,sys
from collections import defaultdict
db=sqlite3.connect(glob.glob(sys.argv[1]+"/*/*.db")[0]); cur=db.cursor()
rows=list(cur.execute("select kernel_name, counter_name, sum(value), count(*), sum(end-start)/1e6 from counters_collection group by kernel_name, counter_name"))
d=defaultdict(dict); t={}
for k,c,v,n,ms in rows: d[k][c]=v; t[k]=(n,ms)
for k,v in sorted(d.items(), key=lambda kv:-t[kv[0]][1])[:int(sys.argv[2]) if len(sys.argv)>2 else 8]:
    print(k.split("::")[-2 if "::" in k else 0][:70] if False else k[:90], "calls",t[k][0], "ms %.1f"%t[k][1])
    print("    ", {c: "%.3g"%x for c,x in v.items()})
