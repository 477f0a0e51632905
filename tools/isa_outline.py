"""Outline of one gfx950 kernel's ISA (no GPU needed): per basic block the counts of MFMA / LDS / global / accvgpr-move instructions and, in order, every
s_waitcnt / s_barrier -- enough to see whether a K loop is straight-line and where the epilogue waits.

    python tools/isa_outline.py gemm.hip 'k_gemm_ppILi128ELi144ELi4ELi1ELi4ELi3ELi2ELi64' [--seq]     (--seq: the memory / wait instructions of the epilogue in order)
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = os.path.join(ROOT, 'ezaudio_amd', 'csrc', sys.argv[1])
    key = sys.argv[2]
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, 'x.s')
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-S', '--cuda-device-only', src, '-o', out], check=True, capture_output=True)
        s = open(out).read()
    names = [m for m in re.findall(r'^(_Z\w+):', s, flags=re.M) if key in m]
    for name in names:
        i = s.index(name + ':')
        body = s[i:s.index('.Lfunc_end', i)].split('\n')
        print('==', name)
        lab, blocks, cur = 'entry', collections.OrderedDict(), []
        for ln in body:
            m = re.match(r'^(\.LBB\d+_\d+):', ln)
            if m:
                blocks[lab] = cur
                lab, cur = m.group(1), []
            else:
                cur.append(ln.strip())
        blocks[lab] = cur
        for k, v in blocks.items():
            c = lambda pat: sum(bool(re.match(pat, x)) for x in v)   # noqa: E731
            sync = [x.split(';')[0].strip().replace('s_waitcnt ', 'w:') for x in v if x.startswith(('s_waitcnt', 's_barrier'))]
            if c('v_mfma') or c('global_|buffer_') or c('ds_') or sync:
                print(f'{k:12s} n={len(v):4d} mfma={c("v_mfma"):3d} ds_r={c("ds_read"):3d} ds_w={c("ds_write"):3d} gld={c("global_load_dword|buffer_load"):3d} dma={c("global_load_lds"):3d} '
                      f'gst={c("global_store|buffer_store"):3d} accmov={c("v_accvgpr"):3d} scratch={c("scratch_"):2d} | {" ".join(sync)[:160]}')
        if '--seq' in sys.argv:
            last = max(k for k, l in enumerate(body) if 'v_mfma' in l)
            for ln in body[last:]:
                t = ln.strip().split(';')[0].strip()
                if re.match(r's_waitcnt|s_barrier|global_|buffer_|ds_|\.LBB|s_cbranch|s_branch|v_permlane', t):
                    print('   ', t[:100])


if __name__ == '__main__':
    main()
