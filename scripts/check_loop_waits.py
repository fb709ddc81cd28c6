"""List the compiler-inserted `s_waitcnt vmcnt(..)` (those OUTSIDE inline-asm blocks) inside the loops of the persistent
recurrence kernels.  A compiler-visible global load inside the step loop makes hipcc drain the whole VM queue (exchange
stores, saves, prefetch DMAs) every step -- the waits must be the hand-placed ones only.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iecog2txt_amd/csrc -S --cuda-device-only <src>.hip -o x.s
    python scripts/check_loop_waits.py x.s [kernel-name-regex]
"""
import re, sys, subprocess
lines = open(sys.argv[1]).read().split('\n')
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else 'persist|_big')
cur, in_asm, in_loop, out = None, False, False, {}
for ln in lines:
    m = re.match(r'^(_Z\w+):', ln)
    if m:
        cur = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
        in_loop = False
        continue
    if cur is None or not pat.search(cur):
        continue
    if 'ASMSTART' in ln: in_asm = True
    if 'ASMEND' in ln: in_asm = False
    if 'Loop Header' in ln or 'in Loop' in ln: in_loop = True
    if 's_endpgm' in ln: cur = None; continue
    if in_loop and not in_asm and 's_waitcnt' in ln and 'vmcnt' in ln:
        out.setdefault(cur, []).append(ln.strip())
for k, v in out.items():
    print('%-70s %d compiler vmcnt waits after the first loop header: %s' % (k[:70], len(v), ', '.join(sorted(set(v)))))
