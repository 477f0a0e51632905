"""Per-kernel register / LDS / occupancy table of one translation unit, from hipcc's -Rpass-analysis=kernel-resource-usage (no GPU needed).

    python tools/kernel_resources.py gemm.hip [filter]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = os.path.join(ROOT, 'ezaudio_amd', 'csrc', sys.argv[1])
    flt = sys.argv[2] if len(sys.argv) > 2 else ''
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-c', src, '-o', os.path.join(d, 'x.o'),
                            '-Rpass-analysis=kernel-resource-usage'] + sys.argv[3:], capture_output=True, text=True)
    blocks = re.split(r'remark: Function Name: ', r.stderr)[1:]
    for b in blocks:
        name = b.split()[0]
        dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        dn = dn.replace('(anonymous namespace)::', '').replace('void ', '')
        if flt and flt not in dn:
            continue

        def g(k):
            m = re.search(k + r': (\d+)', b)
            return m.group(1) if m else '?'
        print('%-64s V=%s A=%s S=%s scratch=%s occ=%s lds=%s' % (dn[:64], g('VGPRs'), g('AGPRs'), g('TotalSGPRs'), g(r'ScratchSize \[bytes/lane\]'),
                                                                 g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))


if __name__ == '__main__':
    main()
