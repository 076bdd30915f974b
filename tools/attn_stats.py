import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
rows = cur.execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start").fetchall()
agg = collections.defaultdict(list)
for n,s,e,gx,gy,gz,wx in rows:
    if 'attn_kernel' in n:
        agg[(n.split('(')[0][5:35], gx//wx, gy, gz, wx)].append((e-s)/1e3)
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1])):
    print(k, len(v), "avg %.2f min %.2f total_ms %.2f"%(sum(v)/len(v), min(v), sum(v)/1e3))
