#!/usr/bin/env python3
"""Per-kernel averages from a rocprofv3 rocpd database (`rocprofv3 --kernel-trace -d DIR -o NAME` writes
DIR/NAME_results.db on this image): count, mean / min microseconds and grid size per (kernel, grid), in order
of first launch.  Usage: rocpd_kernels.py results.db [name-filter]"""
import collections
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.OrderedDict()
for name, grid, start, end in db.execute("select name, grid_x, start, end from kernels order by start"):
    if flt in name:
        agg.setdefault((name, grid), []).append((end - start) / 1e3)
for (name, grid), v in agg.items():
    print(f"{len(v):5d} x {sum(v) / len(v):9.1f} us (min {min(v):9.1f})  grid {grid:10d}  {name[:150]}")
