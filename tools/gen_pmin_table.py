#!/usr/bin/env python3
"""Extract the empirical P(shared minimizer) table into a compact binary data file.

Source of the numbers: the text table embedded in the reference at
src/p_emp_prob_data.h:6-41886 (rows "k w p e1 e2"; read at src/p_emp_prob.cpp:22-47).
Only the *numbers* are kept; the output is a data file, not source:

  magic   8s   b"IOCPMIN1"
  ncombo  u32  number of (k, w) combinations, in the order they first appear in the table
  per combo:  k i32 | w i32 | p[15][15] f64   (cell [a][b] = error rates (a+1)/100, (b+1)/100;
                                              symmetric fill exactly as p_emp_prob.cpp:37-43 does;
                                              NaN = no row)

Run in the build container only (needs /root/reference):  python tools/gen_pmin_table.py
"""
import math
import re
import struct
import sys

SRC = "/root/reference/src/p_emp_prob_data.h"
DST = "isonclust2_amd/data/pmin_shared.bin"


def main():
    combos = {}
    order = []
    pat = re.compile(r'\s*"(\d+)\t(\d+)\t(\S+)\t(\S+)\t(\S+)\\n"')
    nrows = 0
    for line in open(SRC):
        m = pat.match(line)
        if not m:
            continue
        nrows += 1
        k, w = int(m[1]), int(m[2])
        p, e1, e2 = float(m[3]), float(m[4]), float(m[5])
        a, b = round(e1 * 100), round(e2 * 100)
        assert 1 <= a <= 15 and 1 <= b <= 15
        assert a / 100 == e1 and b / 100 == e2, (e1, e2)
        if (k, w) not in combos:
            combos[(k, w)] = [[math.nan] * 15 for _ in range(15)]
            order.append((k, w))
        t = combos[(k, w)]
        t[a - 1][b - 1] = p
        t[b - 1][a - 1] = p
    with open(DST, "wb") as f:
        f.write(b"IOCPMIN1")
        f.write(struct.pack("<I", len(order)))
        for k, w in order:
            f.write(struct.pack("<ii", k, w))
            for row in combos[(k, w)]:
                f.write(struct.pack("<15d", *row))
    print(f"{nrows} rows, {len(order)} (k,w) combos -> {DST}", file=sys.stderr)


if __name__ == "__main__":
    main()
