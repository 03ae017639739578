"""Memory copies of a rocprofv3 --memory-copy-trace run (rocpd database) grouped by direction and size.

    python tools/memcpy_summary.py <dir>
"""
import glob, os, sqlite3, sys
d=sys.argv[1]
db=sorted(glob.glob(os.path.join(d,"**","*.db"),recursive=True),key=os.path.getmtime)[-1]
con=sqlite3.connect(db)
names=[r[0] for r in con.execute("select name from sqlite_master where type in ('table','view')")]
print([n for n in names if 'mem' in n.lower() or 'copy' in n.lower()])
for t in names:
    if 'memory_cop' in t.lower() and 'rocpd_memory_copy' not in t.lower():
        cols=[c[1] for c in con.execute(f"pragma table_info({t})")]
        print(t, cols)
        try:
            rows=con.execute(f"select name, size, count(*), sum(duration) from {t} group by name, size order by count(*) desc limit 40").fetchall()
            for r in rows: print(r)
        except Exception as e: print("ERR", e)
