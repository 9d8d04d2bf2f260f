"""Per-kernel comparison of two rocprofv3 --stats runs (tools/ab_stats.sh): ms per step of every kernel family under A and B.
    python tools/ab_stats.py gpurun_out/ab_0_stats gpurun_out/ab_1_stats [steps_profiled]"""
import collections, csv, glob, os, re, sys


def load(d):
    f = glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True)[0]
    tab = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
        n = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', n)
        n = re.sub(r'\(.*', '', n)[:70]
        c, t = tab.get(n, (0, 0.0))
        tab[n] = (c + int(r['Calls']), t + float(r['TotalDurationNs']))
    return tab


a, b = load(sys.argv[1]), load(sys.argv[2])
steps = None
for k in a:
    if 'adam_kernel' in k:
        steps = a[k][0]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else steps
print('steps profiled: %d' % steps)
rows = []
for k in sorted(set(a) | set(b)):
    ca, ta = a.get(k, (0, 0.0))
    cb, tb = b.get(k, (0, 0.0))
    rows.append((tb - ta, k, ca, ta, cb, tb))
ta_all = sum(r[3] for r in rows) / steps / 1e6
tb_all = sum(r[5] for r in rows) / steps / 1e6
print('%-72s %6s %9s %9s | %6s %9s %9s | %8s' % ('kernel', 'n/step', 'ms/step', 'us/launch', 'n/step', 'ms/step', 'us/launch', 'delta ms'))
for d, k, ca, ta, cb, tb in sorted(rows, key=lambda r: -abs(r[0])):
    if abs(d) / steps / 1e6 < 0.005 and max(ta, tb) / steps / 1e6 < 0.1:
        continue
    print('%-72s %6.1f %9.3f %9.1f | %6.1f %9.3f %9.1f | %+8.3f' % (k, ca / steps, ta / steps / 1e6, ta / max(ca, 1) / 1e3, cb / steps,
                                                                      tb / steps / 1e6, tb / max(cb, 1) / 1e3, d / steps / 1e6))
print('sum of kernel time per step: %.3f -> %.3f ms' % (ta_all, tb_all))
