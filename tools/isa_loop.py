"""Print the skeleton (MFMA / LDS / VMEM / waits / barriers / branches) of one
kernel's main loop from a hipcc -S listing:
    python tools/isa_loop.py file.s <kernel-name-substring> [--all]"""
import sys

s = open(sys.argv[1]).read().split('\n')
sub = sys.argv[2]
start = next(i for i, l in enumerate(s) if sub in l and l.startswith("_Z") and l.split()[0].endswith(":"))
end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i])
body = s[start:end]
first = next(i for i, l in enumerate(body) if 'v_mfma' in l)
last = max(i for i, l in enumerate(body) if 'v_mfma' in l)
keys = ('v_mfma', 'ds_read', 'ds_write', 'buffer_load', 'buffer_store', 'global_', 's_waitcnt',
        's_barrier', 's_cbranch', '.LBB', 's_branch', 's_setprio', 's_nop')
other = 0
for l in body[max(0, first - 40):last + 3]:
    l = l.strip()
    if not l or l.startswith(';'):
        continue
    mn = l.split()[0]
    if mn.startswith(keys) or l.endswith(':'):
        if other:
            print(f'    ... {other} other')
            other = 0
        print(l[:110])
    else:
        other += 1
