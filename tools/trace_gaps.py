"""Timeline of one training step from a rocprofv3 --kernel-trace CSV (tools/run_profiles.sh: gpurun_out/<tag>_stats): per
hardware queue the busy time and the idle gaps between consecutive kernels, the union over queues, and the kernels that sit in
front of the longest gaps.  A step = from one adam_kernel end to the next.
    python tools/trace_gaps.py <kernel_trace.csv> [steps to skip at the end]
(bench.py with the graph runs up to five EAGER steps with per-launch events after its timed replays: skip 6 to look at a replay)"""
import collections, csv, re, sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), int(r['Queue_Id']), r['Kernel_Name']))
rows.sort()
adam = [i for i, r in enumerate(rows) if 'adam_kernel' in r[3]]
# (the optimizer may be split into up to three launches around the last weight gradient: a step ends at the LAST of a group)
adam = [i for k, i in enumerate(adam) if k + 1 == len(adam) or rows[adam[k + 1]][0] - rows[i][1] > 2000000]
if len(adam) < 3:
    sys.exit('fewer than three optimizer launches in the trace')


def short(n):
    n = n.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.match(r'_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I', n)
    return (m.group(1) if m else re.sub(r'[<(].*', '', n))[:34]


skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
walls = [(rows[adam[i + 1]][1] - rows[adam[i]][1]) / 1e6 for i in range(len(adam) - 1)]
print('step walls (ms), oldest first:', ' '.join('%.2f' % w for w in walls[-16:]))
a, b = adam[-2 - skip], adam[-1 - skip]
step = rows[a + 1:b + 1]
t0, t1 = rows[a][1], rows[b][1]
print('step wall %.3f ms, %d kernels' % ((t1 - t0) / 1e6, len(step)))
byq = collections.defaultdict(list)
for r in step:
    byq[r[2]].append(r)
for q, rs in sorted(byq.items()):
    busy = sum(e - s for s, e, _, _ in rs)
    gaps = [(rs[i + 1][0] - rs[i][1], rs[i][3], rs[i + 1][3]) for i in range(len(rs) - 1)]
    pos = [g for g in gaps if g[0] > 0]
    print('queue %d: %4d kernels, busy %.3f ms, span %.3f ms, gaps>0: %d totalling %.3f ms (median %.2f us), overlapping pairs %d'
          % (q, len(rs), busy / 1e6, (rs[-1][1] - rs[0][0]) / 1e6, len(pos), sum(g[0] for g in pos) / 1e6,
             sorted(g[0] for g in pos)[len(pos) // 2] / 1e3 if pos else 0.0, sum(1 for g in gaps if g[0] <= 0)))
# union over queues
ev = sorted([(s, 1) for s, e, _, _ in step] + [(e, -1) for s, e, _, _ in step])
depth, last, idle, busy2 = 0, t0, 0, 0
for t, d in ev:
    if depth == 0:
        idle += t - last
    elif depth >= 2:
        busy2 += t - last
    last = t
    depth += d
print('union: idle %.3f ms of the step, two or more kernels in flight %.3f ms' % (idle / 1e6, busy2 / 1e6))
# idle intervals (nothing running) and what ran before / after them
ends = sorted(step, key=lambda r: r[1])
gl = []
depth, last = 0, t0
prev = None
for t, d, r in sorted([(s, 1, (s, e, q, n)) for s, e, q, n in step] + [(e, -1, (s, e, q, n)) for s, e, q, n in step], key=lambda x: (x[0], x[1])):
    if depth == 0 and d == 1 and prev is not None:
        gl.append((t - last, prev[3], r[3]))
    if d == -1:
        prev = r
    last = t
    depth += d
agg = collections.defaultdict(lambda: [0, 0])
for g, p, n in gl:
    k = short(p) + ' -> ' + short(n)
    agg[k][0] += 1
    agg[k][1] += g
print('idle time by (kernel before -> kernel after), top 25:')
for k, (c, g) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
    print('  %-72s n=%3d  %.1f us total, %.2f us each' % (k, c, g / 1e3, g / 1e3 / c))
# where the step's time goes: forward (up to the loss kernel), backward (up to the last kernel in front of the optimizer), tail;
# and whether the optimizer waited for the weight-gradient queue
qs = sorted(byq, key=lambda q: -len(byq[q]))
main_q = qs[0]
mainr = byq[main_q]
ce = [i for i, r in enumerate(mainr) if 'softmax_ce' in r[3]]
if ce:
    fwd_end = mainr[ce[-1]][1]
    first_adam = min(i for i, r in enumerate(mainr) if 'adam_kernel' in r[3] and r[0] > fwd_end)
    adam_start = mainr[first_adam][0]
    before_adam = mainr[first_adam - 1][1]
    print('main queue %d: forward %.3f ms, backward %.3f ms (to its last kernel), optimizer starts %.1f us after that kernel'
          % (main_q, (fwd_end - t0) / 1e6, (before_adam - fwd_end) / 1e6, (adam_start - before_adam) / 1e3))
    for q in qs[1:]:
        rs = byq[q]
        print('queue %d: first kernel %.3f ms into the step, last kernel ends %.1f us %s the main queue\'s last backward kernel'
              % (q, (rs[0][0] - t0) / 1e6, abs(rs[-1][1] - before_adam) / 1e3, 'AFTER' if rs[-1][1] > before_adam else 'before'))
        # how far the queue lags: for its kernels, time between becoming launchable (previous kernel on the same queue ended) -- not
        # known from the trace -- so report its idle share inside its own span instead
        busy = sum(e - s for s, e, _, _ in rs)
        print('          busy %.3f ms of a %.3f ms span (%.0f %%)' % (busy / 1e6, (rs[-1][1] - rs[0][0]) / 1e6,
                                                                   100.0 * busy / max(rs[-1][1] - rs[0][0], 1)))
    print('tail of the step (ms into it, queue, duration us):')
    for r in sorted(step, key=lambda r: r[0])[-8:]:
        print('   %.3f  q%d  %6.1f  %s' % ((r[0] - t0) / 1e6, r[2], (r[1] - r[0]) / 1e3, short(r[3])))
    print('head of the step:')
    for r in sorted(step, key=lambda r: r[0])[:8]:
        print('   %.3f  q%d  %6.1f  %s' % ((r[0] - t0) / 1e6, r[2], (r[1] - r[0]) / 1e3, short(r[3])))
    big = sorted(((mainr[i + 1][0] - mainr[i][1], short(mainr[i][3]), short(mainr[i + 1][3]), (mainr[i][1] - t0) / 1e6)
                  for i in range(len(mainr) - 1)), reverse=True)[:12]
    print('longest gaps on the main queue:')
    for g, p, n, at in big:
        print('  %7.1f us at %.2f ms  %s -> %s' % (g / 1e3, at, p, n))
    for q in qs[1:]:
        rs = byq[q]
        print('queue %d, its first 12 kernels (start ms, duration us, gap to the previous one us | main-queue kernel running at that start):' % q)
        for i, r in enumerate(rs[:12]):
            cur = [m for m in mainr if m[0] <= r[0] < m[1]]
            print('   %.3f  %6.1f  %7.1f  %-28s | %s' % ((r[0] - t0) / 1e6, (r[1] - r[0]) / 1e3, (r[0] - rs[i - 1][1]) / 1e3 if i else 0.0,
                                                    short(r[3]), short(cur[0][3]) if cur else '-'))
