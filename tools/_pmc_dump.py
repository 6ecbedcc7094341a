import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
q = "select kernel_name, counter_name, avg(value), count(*) from counters_collection where kernel_name like '%island%' group by kernel_name, counter_name"
for r in db.execute(q): print(r[1], round(r[2],1), r[3])
