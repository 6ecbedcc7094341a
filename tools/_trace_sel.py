"""dev tool: print blocks lo..hi of a tests/_trace.py pipe32 / c2pipe dump."""
import os as _os, sys as _sys; _R = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))); _sys.path[:0] = [_R, _os.path.join(_R, 'tests')]

import re, sys
lo, hi = int(sys.argv[2]), int(sys.argv[3])
for line in open(sys.argv[1]):
    if line.startswith('wave'): print(line.strip()); continue
    sel = [x for x in line.split() if (m := re.search(r'b(\d+)@', x)) and lo <= int(m.group(1)) <= hi]
    print('   ', ' '.join(sel))
