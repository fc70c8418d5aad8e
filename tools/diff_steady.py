"""Development-only: per-kernel difference of two steady-state tables (tools/trace_steady.py output): which kernels one
configuration spends more time in than the other. usage: diff_steady.py a.txt b.txt"""
import re, sys
def load(p):
    d = {}
    for ln in open(p).read().splitlines()[1:]:
        m = re.match(r"(.*?)\s+([\d.]+)/step\s+([\d.]+) us/step", ln)
        if m:
            k = m.group(1).strip()[:90]
            c, u = d.get(k, (0.0, 0.0))
            d[k] = (c + float(m.group(2)), u + float(m.group(3)))
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
rows = [(b.get(k, (0, 0))[1] - a.get(k, (0, 0))[1], k) for k in set(a) | set(b)]
print("total a %.0f us, b %.0f us" % (sum(v[1] for v in a.values()), sum(v[1] for v in b.values())))
for d, k in sorted(rows, key=lambda r: -abs(r[0]))[:40]:
    print("%+9.1f us  a %5.1f x %7.1f | b %5.1f x %7.1f  %s" % (d, a.get(k, (0, 0))[0], a.get(k, (0, 0))[1], b.get(k, (0, 0))[0], b.get(k, (0, 0))[1], k))
