"""Turns rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, CSV) plus a --kernel-trace
--stats run into profiles/pmc_traffic.json + a readable table.  Corrections per
/opt/skills/guides/MI355X_MICROARCH.md section HBM: counter unit is KiB; on gfx950 FETCH_SIZE reads
half the bytes of wide coalesced streaming loads, so it is doubled; WRITE_SIZE is exact.

    python tools/pmc_summary.py gpurun_out/pmc2_fetch gpurun_out/pmc2_write gpurun_out/prof4 <tag>
"""
import collections, csv, glob, json, os, re, sys


def short(name):
    name = name.replace('(anonymous namespace)::', '')
    m = re.match(r'_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I', name)
    if m:
        tail = name[m.end() - 1:]
        variant = re.sub(r'[^0-9A-Za-z]+', '_', tail[:28])
        return m.group(1) + '<' + variant.strip('_') + '>'
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\((?:anonymous namespace)::[A-Za-z]*Params\)|\(.*$', '', name).strip()[:60]


def load_counter(d, cname):
    f = glob.glob(os.path.join(d, '*', '*_counter_collection.csv'))[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == cname:
            k = short(r['Kernel_Name'])
            agg[k][0] += 1
            agg[k][1] += float(r['Counter_Value'])
    return agg


# bench.py's roofline objects are per libcnk ProfScope family (csrc/common.h CNK_KF_*): the launches of one family are the
# same population here (traffic and algorithmic bytes must describe the same launches)
FAMILY = {'gemm1x1_kernel': 'igemm_gather_kernel', 'fold3_kernel': 'igemm_gather_kernel',          # CNK_KF_GATHER
          'igemm_halo3x3_pf_kernel': 'igemm_halo3x3_kernel', 'pp3x3_kernel': 'igemm_halo3x3_kernel',      # CNK_KF_HALO
          'wgrad3x3_kernel': 'wgrad_mfma_kernel'}                                        # CNK_KF_WGRAD


def main(fetch_dir, write_dir, stats_dir, tag):
    fa, wa = load_counter(fetch_dir, 'FETCH_SIZE'), load_counter(write_dir, 'WRITE_SIZE')
    stats = {}
    f = glob.glob(os.path.join(stats_dir, '*', '*_kernel_stats.csv'))[0]
    for r in csv.DictReader(open(f)):
        stats[short(r['Name'])] = (int(r['Calls']), float(r['TotalDurationNs']), float(r['AverageNs']))
    rows, out, fam = [], {}, collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
    for k in fa:
        nf, fs = fa[k]
        nw, ws = wa.get(k, (1, 0.0))
        fetch = 2.0 * fs * 1024 / nf
        write = ws * 1024 / max(nw, 1)
        calls, tot, avg = stats.get(k, (0, 0.0, 0.0))
        rows.append((tot, k, nf, fetch, write, avg))
        f_ = fam[FAMILY.get(k.split('<')[0].strip(), k.split('<')[0].strip())]
        f_[0] += nf; f_[1] += fetch * nf; f_[2] += write * nf; f_[3] += avg * nf
        out[k] = {'unit': 'bytes/launch', 'fetch': round(fetch), 'write': round(write), 'total': round(fetch + write),
                  'avg_launch_us': round(avg / 1e3, 2),
                  'hbm_side_GBps': round((fetch + write) / avg, 1) if avg else None}
    rows.sort(reverse=True)
    lines = ['%-58s %7s %12s %12s %10s %10s' % ('kernel', 'launch', 'fetch MB/l', 'write MB/l', 'avg us', 'GB/s')]
    for tot, k, n, fe, wr, avg in rows[:24]:
        lines.append('%-58s %7d %12.1f %12.1f %10.1f %10.0f' % (k, n, fe / 1e6, wr / 1e6, avg / 1e3, (fe + wr) / avg if avg else 0))
    # whole-step traffic: every kernel's bytes per launch x its launches per step in the stats run (steps of that run = launches of
    # the once-per-step optimizer kernel)
    steps = next((c for k, (c, _, _) in stats.items() if k.startswith('adam_kernel')), 0)
    step_bytes = sum((out[k]['fetch'] + out[k]['write']) * stats[k][0] for k in out if k in stats) / steps if steps else None
    if step_bytes:
        lines.append('whole step: %.2f GB HBM-side over %d profiled steps (sum over kernels of bytes/launch x launches/step)' % (step_bytes / 1e9, steps))
    txt = '\n'.join(lines)
    print(txt)
    os.makedirs('profiles', exist_ok=True)
    json.dump({'tag': tag, 'note': 'FETCH_SIZE x2 (gfx950 wide-read correction), WRITE_SIZE exact; KiB->bytes; separate --pmc passes; '
                                   'families follow libcnk\'s ProfScope families: ' + ', '.join('%s -> %s' % kv for kv in sorted(FAMILY.items())),
               'kernels': out,
               'step': {'bytes': round(step_bytes) if step_bytes else None, 'steps_profiled': steps,
                        'note': 'HBM-side bytes of one training step: sum over kernels of (fetch + write per launch) x launches per step of the '
                                'kernel-stats run; steps = launches of adam_kernel'},
               'families': {k: {'unit': 'bytes/launch (launch-weighted mean over template variants)', 'launches_profiled': v[0],
                                'fetch': round(v[1] / v[0]), 'write': round(v[2] / v[0]), 'total': round((v[1] + v[2]) / v[0]),
                                'avg_launch_us': round(v[3] / v[0] / 1e3, 2)} for k, v in fam.items() if v[0]}},
              open('profiles/pmc_traffic.json', 'w'), indent=1)
    open('profiles/%s_pmc_traffic.txt' % tag, 'w').write(txt + '\n')


if __name__ == '__main__':
    main(*sys.argv[1:5])
