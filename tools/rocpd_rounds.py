import sqlite3, sys
db=sqlite3.connect(sys.argv[1])
for r in db.execute("select name, (end-start)/1000.0 from kernels where name like '%raster_grid%' order by start"): print("   grid dispatch %.1f us" % r[1])
