import re, sys
rows = {}
cur = None
for l in open(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/sk_bench.txt'):
    if l.startswith('=='):
        cur = l.split('==')[1].strip()
        continue
    m = re.match(r'(\S+)\s+(\S+)\s+([\d.]+) us\s+([\d.]+) TFLOP', l)
    if m:
        rows.setdefault((m.group(1), m.group(2)), {}).setdefault(cur, []).append((float(m.group(3)), float(m.group(4))))
cols = []
for v in rows.values():
    for c in v:
        if c not in cols:
            cols.append(c)
print("layer role  " + "   ".join("%-22s" % c for c in cols) + "  (TFLOP/s per run; last: total us of the best run)")
tot = {c: 0.0 for c in cols}
for k, v in rows.items():
    print("%-4s %-3s   " % k + "   ".join("%-22s" % " ".join("%6.1f" % x[1] for x in v.get(c, [])) for c in cols))
    for c in cols:
        if v.get(c):
            tot[c] += min(x[0] for x in v[c])
print("sum us     " + "   ".join("%-22.1f" % tot[c] for c in cols))
