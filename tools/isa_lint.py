"""ISA lint for the gfx950 kernels (no GPU needed): compile a .hip file to device
assembly and list, per kernel, the `s_waitcnt vmcnt(N)` values that sit inside
loop bodies.  A vmcnt(0) in a pipelined loop means the ring is not prefetching
(round 3: a branch around the ring loads of the C8 weight gradient made hipcc
merge the counters of both paths -- every step waited for the loads it had just
issued).  `--epilogue` also counts vmcnt(0) outside loops (a chain of dependent
loads: round 3's may-alias epilogue had 155 in one kernel).

    python tools/isa_lint.py ld_amd/csrc/conv_bf16.hip [--match wgrad] [--epilogue]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def device_asm(src):
    out = tempfile.mktemp(suffix='.s')
    cmd = [os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '-O3', '--offload-arch=gfx950',
           '-std=c++17', '-I' + os.path.join(REPO, 'include'),
           '-I' + os.path.join(REPO, 'ld_amd', 'csrc'), '--cuda-device-only', '-S', src,
           '-o', out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.remove(out)
    return text


def lint(text, match='', epilogue=False):
    lines = text.split('\n')
    starts = [(i, l.split(':')[0]) for i, l in enumerate(lines)
              if re.match(r'^_Z\w*kernel\w*:\s', l)]
    rows = []
    for n, (i, name) in enumerate(starts):
        if match and match not in name:
            continue
        end = starts[n + 1][0] if n + 1 < len(starts) else len(lines)
        inloop, hist, outside0 = False, {}, 0
        for l in lines[i:end]:
            if 'Loop Header' in l or 'in Loop' in l:
                inloop = True
            elif l.startswith('.LBB'):
                inloop = False
            m = re.search(r's_waitcnt vmcnt\((\d+)\)', l)
            if not m:
                continue
            if inloop:
                hist[int(m.group(1))] = hist.get(int(m.group(1)), 0) + 1
            elif int(m.group(1)) == 0:
                outside0 += 1
        short = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', name)
        rows.append((short, hist, outside0))
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('src')
    ap.add_argument('--match', default='')
    ap.add_argument('--epilogue', action='store_true')
    a = ap.parse_args()
    bad = 0
    for name, hist, out0 in lint(device_asm(a.src), a.match, a.epilogue):
        flag = ''
        if hist.get(0, 0) and len(hist) <= 3:
            flag = '   <-- vmcnt(0) dominates the loop'
            bad += 1
        tail = f'  vmcnt(0) outside loops: {out0}' if a.epilogue else ''
        if hist or a.epilogue:
            print(f'{name[:78]:78s} {dict(sorted(hist.items()))}{tail}{flag}')
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
