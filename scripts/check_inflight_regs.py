"""Static check of hand-waited register loads in a gfx950 .s file: between an inline-asm `global_load_dword*` (destination
VGPRs) and the `s_waitcnt vmcnt(N)` that retires it, no other instruction may name those VGPRs -- hipcc treats an asm
load's destination as written at the end of the asm statement and may copy or reuse it while the data is still in flight.
Walks straight-line code per basic block sequence of ONE kernel (branches are followed textually, which is what the
unrolled attempt loops of the persistent recurrences look like).
    python scripts/check_inflight_regs.py x.s '<mangled kernel name substring>'
"""
import re, sys
src = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(src) if re.match(r'^_Z\w+:', l) and key in l)
end = next(i for i in range(start, len(src)) if 's_endpgm' in src[i])
vm = re.compile(r'^\s*(global_load|global_store|global_atomic|buffer_load|buffer_store|scratch_)')
rng = re.compile(r'v\[(\d+):(\d+)\]|v(\d+)\b')
def regs(text):
    out = set()
    for m in rng.finditer(text):
        if m.group(1): out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else: out.add(int(m.group(3)))
    return out
queue = []          # in issue order: (set of dest vgprs or empty, line no)
bad = 0
for i in range(start, end):
    l = src[i].split(';')[0]
    if not l.strip() or l.strip().startswith('.') : continue
    m = re.search(r's_waitcnt.*vmcnt\((\d+)\)', l)
    if m:
        n = int(m.group(1))
        queue = queue[len(queue) - n:] if n < len(queue) else queue
        if n == 0: queue = []
        continue
    if vm.match(l):
        ops = l.strip().split(None, 1)
        dest = set()
        if ops[0].startswith('global_load') and 'lds' not in ops[0]:
            dest = regs(ops[1].split(',')[0])
        # an address/data operand of this VMEM op that is still in flight is a hazard too
        used = regs(ops[1]) - dest
        for d, ln in queue:
            if d & used:
                print('line %d uses in-flight v%s (load at line %d): %s' % (i + 1, sorted(d & used)[:4], ln + 1, l.strip())); bad += 1
        queue.append((dest, i))
        if len(queue) > 64: queue = queue[-64:]
        continue
    used = regs(l)
    for d, ln in queue:
        if d & used:
            print('line %d touches in-flight v%s (load at line %d): %s' % (i + 1, sorted(d & used)[:4], ln + 1, l.strip())); bad += 1
print('%s: %d hazards' % (key, bad))
