"""Summarises a rocprofv3 PMC pass with the matrix-core counters per kernel (north_star: "rocprof-reported ... MFMA utilisation"):
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace \
              --output-format csv -d gpurun_out/pmc_mfma -- python3 bench.py --no-cpu-baseline --steps 2 --warmup 2 --graph off
    python tools/pmc_mfma_summary.py gpurun_out/pmc_mfma > profiles/r02_pmc_mfma.txt

Per kernel (summed over its dispatches):
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (duration * 2.4 GHz * 1024 SIMDs)   share of SIMD-cycles (at the 2.4 GHz peak clock) with
               the matrix pipe busy -- the counter ticks in shader cycles, 32 per v_mfma_f32_32x32x16_bf16 (MI355X_MICROARCH.md)
  TFLOP/s    = SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512 flop / duration (ROCm's MfmaFlopsBF16 expression), i.e. the flops the matrix
               cores actually executed (padding and masked rows included), against the 2500 TFLOP/s dense bf16 peak
  clock      = GRBM_GUI_ACTIVE / 8 XCDs / duration: the effective shader clock while the kernel ran (DVFS)
Durations are End-Start timestamps of the same dispatches (profiled runs clock lower than unprofiled ones: ratios only)."""
import collections, csv, glob, os, re, sys


def short(name):
    name = name.replace('(anonymous namespace)::', '')
    m = re.match(r'_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)I', name)
    if m:
        return m.group(1) + '<' + re.sub(r'[^0-9A-Za-z]+', '_', name[m.end() - 1:][:28]).strip('_') + '>'
    name = re.sub(r'^void ', '', name)
    return re.sub(r'\((?:anonymous namespace)::[A-Za-z]*Params\)|\(.*$', '', name).strip()[:70]


def main(d):
    f = glob.glob(os.path.join(d, '*', '*_counter_collection.csv'))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    seen = set()
    for r in csv.DictReader(open(f)):
        k = short(r['Kernel_Name'])
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (r['Dispatch_Id'])
        if key not in seen:
            seen.add(key)
            agg[k]['ns'] += float(r['End_Timestamp']) - float(r['Start_Timestamp'])
            agg[k]['n'] += 1
    rows = []
    for k, a in agg.items():
        if a['ns'] <= 0:
            continue
        busy = a['SQ_VALU_MFMA_BUSY_CYCLES'] / (a['ns'] * 2.4 * 1024)
        tf = a['SQ_INSTS_VALU_MFMA_MOPS_BF16'] * 512 / a['ns'] / 1e3
        clk = a['GRBM_GUI_ACTIVE'] / 8 / a['ns']
        rows.append((a['ns'], k, int(a['n']), a['ns'] / a['n'] / 1e3, busy, tf, clk))
    rows.sort(reverse=True)
    tot = sum(r[0] for r in rows)
    print('%-74s %6s %9s %7s %10s %9s %7s' % ('kernel', 'calls', 'avg us', 'time %', 'mfma_busy', 'TFLOP/s', 'GHz'))
    for ns, k, n, us, busy, tf, clk in rows[:28]:
        print('%-74s %6d %9.1f %6.1f%% %9.1f%% %9.0f %7.2f' % (k, n, us, 100 * ns / tot, 100 * busy, tf, clk))


if __name__ == '__main__':
    main(sys.argv[1])
