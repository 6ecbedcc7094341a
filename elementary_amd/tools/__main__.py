"""python -m elementary_amd.tools dump {c1|c2|c4:K} out.json — write a BASELINE graph's instruction batch as JSON."""
import sys

from .. import graphs
from ..reconciler import Delegate, batch_to_json, render_with_delegate


def main(argv):
    if len(argv) != 3 or argv[0] != "dump":
        raise SystemExit(__doc__)
    which, path = argv[1], argv[2]
    if which == "c1":
        roots = graphs.c1_graph()
    elif which == "c2":
        roots = graphs.c2_graph()
    elif which.startswith("c4:"):
        roots = [graphs.c4_instance(k) for k in range(int(which[3:]))]
    else:
        raise SystemExit(__doc__)
    d = Delegate()
    render_with_delegate(d, roots)
    open(path, "w").write(batch_to_json(d.packed()))


main(sys.argv[1:])
