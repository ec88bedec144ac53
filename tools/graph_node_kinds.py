"""Count the node kinds of a captured HIP graph from its DOT dump (E4S_GRAPH_DEBUG_DUMP=<path> makes optim.GraphedStep write one):
the captured steps of this library must consist of kernel nodes (+ the event-wait / record edges of the RCCL fork-join) only -- a MEMSET
node means an ATen global reduction or a library call slipped into the capture (kernels.sum_all says why that is fatal on this ROCm)."""
import collections
import json
import re
import sys


def kinds(path):
    txt = open(path, errors="replace").read()
    c = collections.Counter()
    for m in re.finditer(r'label\s*=\s*"([^"]*)"', txt):
        lab = m.group(1)
        up = lab.upper()
        if "MEMSET" in up:
            c["memset"] += 1
        elif "MEMCPY" in up or "MEMCOPY" in up:
            c["memcpy"] += 1
        elif "EVENT" in up:
            c["event"] += 1
        elif "KERNEL" in up or "(" in lab:
            c["kernel"] += 1
        else:
            c["other"] += 1
    return dict(c)


if __name__ == "__main__":
    print(json.dumps({p: kinds(p) for p in sys.argv[1:]}))
