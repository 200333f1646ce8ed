"""Per-K-step times from a tools/h2_ablate.sh log:  python tools/h2_ablate_summary.py gpurun_out/h2/ablateN.log"""
import re
import sys
bits, rows = None, {}
for l in open(sys.argv[1]):
    m = re.match(r'ablate bits (\d+)', l)
    if m:
        bits = int(m.group(1))
        continue
    m = re.search(r'K=(\d+)\s+tile\s+(\d): .*\| h2\s+([\d.]+) us', l)
    if m:
        rows[(bits, int(m.group(2)), int(m.group(1)))] = float(m.group(3))
names = {16: 'noLDSst', 32: 'noGload', 64: 'noVALU', 4: 'noMFMA', 128: 'noB', 2: 'noStaging'}
for b in sorted(set(k[0] for k in rows)):
    d = ' '.join(n for v, n in names.items() if b & v) or 'full'
    for t in (3, 2):
        if (b, t, 4096) in rows:
            per = (rows[(b, t, 4096)] - rows[(b, t, 2048)]) / 64
            print(f"bits {b:3d} tile {t} {d:32s} K4096 {rows[(b, t, 4096)]:7.1f} us  per K-step {per:.3f} us")
