#!/usr/bin/env python
"""Print per-kernel mean counter values of rocprofv3 --pmc passes (development tool).
usage: pmc_dump.py <kernel substring> <csv> [<csv> ...]"""
import sys

import pandas as pd

pat = sys.argv[1]
for path in sys.argv[2:]:
    df = pd.read_csv(path)
    df = df[df.Kernel_Name.str.contains(pat, regex=False)]
    t = df.pivot_table(index="Dispatch_Id", columns="Counter_Name", values="Counter_Value", aggfunc="sum")
    print(path.split("/")[-2], "launches", len(t))
    for c in t.columns:
        print(f"  {c:34s} {t[c].mean():16.1f}")
