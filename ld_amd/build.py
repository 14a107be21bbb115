"""Build libldhip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

    python -m ld_amd.build            # incremental
    python -m ld_amd.build --force

Outputs ld_amd/_lib/libldhip.so (git-ignored, travels to the GPU box with the
gpurun snapshot).  hipcc cross-compiles for gfx950 without a GPU present.
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
OUT_DIR = os.path.join(HERE, '_lib')
LIB = os.path.join(OUT_DIR, 'libldhip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'

# -ffp-contract=off everywhere fp32 op order is part of the contract (target
# assignment must be bit-exact w.r.t. the reference's unfused fp32 ops).
COMMON = [
    '--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC',
    '-fhip-fp32-correctly-rounded-divide-sqrt', '-Wall', '-Wno-unused-function',
    '-Wno-unused-variable'
]
PER_FILE = {
    'targets.hip': ['-ffp-contract=off'],
    'loss.hip': ['-ffp-contract=off'],
    'rows.hip': ['-ffp-contract=off'],
    'infer.hip': ['-ffp-contract=off'],
    'quality.hip': ['-ffp-contract=off'],
    'pipeline.hip': ['-ffp-contract=off'],
}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))


def _deps():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    hdrs.append(os.path.join(os.path.dirname(HERE), 'include', 'ld_hip.h'))
    return hdrs


def _digest(path, flags):
    h = hashlib.sha1()
    for p in [path] + sorted(_deps()):
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(flags).encode())
    return h.hexdigest()


def build(force=False, verbose=True):
    os.makedirs(OUT_DIR, exist_ok=True)
    objs, jobs = [], []
    for src in _sources():
        path = os.path.join(CSRC, src)
        # LD_BUILD_DEFS: extra -D flags for instrumented debug builds
        flags = COMMON + PER_FILE.get(src, []) + os.environ.get('LD_BUILD_DEFS', '').split()
        obj = os.path.join(OUT_DIR, src + '.o')
        stamp = obj + '.sha1'
        dig = _digest(path, flags)
        fresh = (not force and os.path.exists(obj) and os.path.exists(stamp)
                 and open(stamp).read() == dig)
        if not fresh:
            jobs.append(([HIPCC] + flags + ['-c', path, '-o', obj], stamp, dig))
        objs.append(obj)

    def compile_one(job):
        cmd, stamp, dig = job
        if verbose:
            print('[ld_amd.build]', ' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
        with open(stamp, 'w') as f:
            f.write(dig)

    if jobs:  # translation units are independent: compile them side by side
        from concurrent.futures import ThreadPoolExecutor
        workers = max(1, min(len(jobs), int(os.environ.get('LD_BUILD_JOBS', '0'))
                             or (os.cpu_count() or 2) // 2))
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(compile_one, jobs))
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', LIB
               ] + objs
        if verbose:
            print('[ld_amd.build]', ' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
