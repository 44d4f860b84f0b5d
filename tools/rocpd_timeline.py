import sqlite3,sys
c=sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'info_kernel_symbol' in t][0]
rows=list(c.execute(f"select s.kernel_name,d.start,d.end,d.queue_id,d.stream_id from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
st=[i for i,r in enumerate(rows) if 'k_bounds_init' in r[0]]
w=int(sys.argv[2]) if len(sys.argv)>2 else 1
j=st[-w]; k=st[-w+1] if w>1 else len(rows)
t0=rows[j][1]
thr=float(sys.argv[3]) if len(sys.argv)>3 else 0.03
for n,s,e,q,sid in rows[j:k]:
    if (e-s)>thr*1e6: print(f"{(s-t0)/1e6:8.3f} {(e-t0)/1e6:8.3f} dur {(e-s)/1e6:7.3f} q{q} s{sid} {n[:50]}")
