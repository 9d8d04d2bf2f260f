"""Register / scratch / occupancy report of the kernels in one or more csrc/*.hip files (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_regs.py conv_igemm conv_wgrad [--filter gather]"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith('--')]
flt = sys.argv[sys.argv.index('--filter') + 1] if '--filter' in sys.argv else ''
args = [a for a in args if a != flt]
for f in args:
    src = os.path.join(ROOT, 'convnets_amd', 'csrc', f + '.hip')
    with tempfile.TemporaryDirectory() as td:
        r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-ffp-contract=off',
                            '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', os.path.join(td, 'o.o')],
                           capture_output=True, text=True)
    for b in re.split(r'remark: .*Function Name: ', r.stderr)[1:]:
        name = b.split('\n')[0].strip().split()[0]
        dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().replace('(anonymous namespace)::', '')
        dn = re.sub(r'^_ZN12_GLOBAL__N_1\d+', '', dn)
        if flt and flt not in dn:
            continue
        g = lambda k: (re.search(k + r': (\d+)', b) or [0, '?'])[1]
        print('%-100s V%-4s A%-4s scratch %-4s occ %s LDS %s' % (dn[:100], g('VGPRs'), g('AGPRs'), g(r'ScratchSize \[bytes/lane\]'),
                                                                   g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
