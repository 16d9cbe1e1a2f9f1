import sqlite3,sys
db=sqlite3.connect(sys.argv[1])
tabs=[r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd=[t for t in tabs if 'kernel_dispatch' in t][0]; ks=[t for t in tabs if 'kernel_symbol' in t][0]
q=f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e6 from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 4 desc limit %d" % int(sys.argv[2] if len(sys.argv)>2 else 10)
for r in db.execute(q): print("%-60s n=%d avg=%.1f us total=%.2f ms"%(r[0][:60],r[1],r[2],r[3]))
