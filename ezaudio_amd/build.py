"""Build libezaudio_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m ezaudio_amd.build [--force]

One translation unit per kernel family, compiled in parallel, linked into
ezaudio_amd/libezaudio_hip.so (git-ignored; it travels to the GPU box with the tree).
"""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(HERE, 'csrc', 'build')
LIB = os.path.join(HERE, 'libezaudio_hip.so')
SOURCES = ['gemm.hip', 'attn.hip', 'rowops.hip', 'api.hip', 'vae.hip']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']
if os.environ.get('EZAUDIO_ABLATE'):   # timing-only variants of the ping-pong K loop (VAR bits 8 / 16 / 32 of k_gemm_pp, tools/microbench/gemm_bench.cpp); never in the shipped build
    FLAGS.append('-DEZ_ABLATE')
if os.environ.get('EZAUDIO_DIAG'):      # diagnostic build: the `zfake` option (LayerNorm-algebra consumers on neutral tables); never in the shipped build
    FLAGS.append('-DEZ_DIAG')


def source_hash():
    """sha256 over the kernel sources + the ABI header: profiles record it so a stale measurement is never quoted for a newer tree."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.hip', '.h')))
    files.append(os.path.join(HERE, '..', 'include', 'ezdit.h'))
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, 'rb') as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith('.h')] + [os.path.join(HERE, '..', 'include', 'ezdit.h')]
    hipcc = _hipcc()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace('.hip', '.o'))
        if force or _stale(o, [s] + headers):
            jobs.append((s, o))

    def compile_one(job):
        s, o = job
        cmd = [hipcc] + FLAGS + ['-c', s, '-o', o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return job, r

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            for (s, o), r in ex.map(compile_one, jobs):
                if verbose and r.stderr.strip():
                    sys.stderr.write(r.stderr)
                if r.returncode != 0:
                    raise RuntimeError(f'hipcc failed on {s}:\n{r.stderr}')
    objs = [os.path.join(OBJ, s.replace('.hip', '.o')) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stderr}')
    return LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
