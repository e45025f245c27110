#!/usr/bin/env python3
"""VGPR liveness over the ISA of one kernel (hipcc -S output): where is the register peak, and which
registers are live across a given loop?

    python tools/isa_liveness.py /tmp/pt_gpu.s k_wf_traceILb0ELb0ELb0

Builds the CFG from labels / s_branch / s_cbranch_*, runs a backward dataflow on v-registers (a write under
a partial exec mask is treated as a definition only at function level = conservative enough for a picture),
prints the peak and the live set at every block entry, and for every loop header the registers that are live
around the loop but never referenced inside it (candidates for parking in LDS)."""
import re, sys
path, kname = sys.argv[1], sys.argv[2]
txt = open(path).read()
funcs = re.split(r'\n(?=_Z[\w]+:\s)', txt)
body = next(f for f in funcs if f.split(':')[0].find(kname) >= 0)
lines = [l for l in body.split('\n')[1:]]
# instructions and labels
blocks, cur, order = {}, None, []
def new_block(name):
    global cur
    cur = name; blocks[name] = []; order.append(name)
new_block('entry')
for l in lines:
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.') and not re.match(r'^\.LBB\d+_\d+:', t):
        if t.startswith('.Lfunc_end'): break
        continue
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        new_block(m.group(1)); continue
    t = t.split(';')[0].strip()
    if t: blocks[cur].append(t)
    if t.startswith('s_endpgm'): break

def regs(tok):
    out = set()
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', tok):
        out |= set(range(int(a), int(b) + 1))
    for a in re.findall(r'\bv(\d+)\b', tok):
        out.add(int(a))
    return out

def defs_uses(ins):
    op, _, rest = ins.partition(' ')
    ops = [o.strip() for o in rest.split(',')] if rest else []
    d, u = set(), set()
    if not ops: return d, u
    stores = op.startswith(('global_store', 'scratch_store', 'ds_write', 'flat_store', 'buffer_store', 'global_atomic', 'ds_min', 'ds_add', 'ds_dec'))
    cmpx = op.startswith(('v_cmp', 's_', 'ds_write', 'v_cmpx'))
    if stores or op.startswith('v_cmp') or op.startswith('s_'):
        for o in ops: u |= regs(o)
        if op.startswith('v_readfirstlane') or op.startswith('v_readlane'): pass
        return d, u
    d |= regs(ops[0])
    for o in ops[1:]: u |= regs(o)
    if op.startswith(('v_fmac', 'v_mac', 'v_pk_fmac', 'v_dot2c')) or 'op_sel' in ins and False:
        u |= regs(ops[0])
    if op.startswith(('v_div_fmas',)): pass
    if op.startswith('v_cndmask') and False: pass
    return d, u

# successors
succ = {}
for i, b in enumerate(order):
    s = set()
    ins = blocks[b]
    fall = True
    for t in ins:
        m = re.match(r'^(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)', t)
        if m:
            s.add(m.group(2))
            if m.group(1) == 's_branch': fall = False
        if t.startswith('s_endpgm'): fall = False
    if fall and i + 1 < len(order): s.add(order[i + 1])
    succ[b] = s
live_in = {b: set() for b in order}
live_out = {b: set() for b in order}
changed = True
while changed:
    changed = False
    for b in reversed(order):
        out = set()
        for s in succ[b]: out |= live_in.get(s, set())
        live = set(out)
        for t in reversed(blocks[b]):
            d, u = defs_uses(t)
            # a VALU write under a divergent exec mask does not kill the old value: keep it live if the
            # block is inside divergent control flow; approximated by never killing in blocks that start
            # with an exec restore.  Good enough for a picture; errs on the high side.
            live -= d
            live |= u
        if live != live_in[b] or out != live_out[b]:
            live_in[b], live_out[b] = live, out; changed = True
# peak inside blocks
peak, where = 0, None
for b in order:
    live = set(live_out[b])
    for t in reversed(blocks[b]):
        d, u = defs_uses(t)
        n = len(live | d)
        if n > peak: peak, where = n, (b, t)
        live -= d; live |= u
print(f"kernel {kname}: {sum(len(v) for v in blocks.values())} instructions, {len(order)} blocks, peak live VGPRs ~{peak} at {where}")
# loops: back edges (successor earlier in order)
idx = {b: i for i, b in enumerate(order)}
for b in order:
    for s in succ[b]:
        if s in idx and idx[s] <= idx[b]:
            body_blocks = order[idx[s]:idx[b] + 1]
            n_ins = sum(len(blocks[x]) for x in body_blocks)
            ref = set()
            for x in body_blocks:
                for t in blocks[x]:
                    d, u = defs_uses(t); ref |= d | u
            through = live_in[s] - ref
            has_load = any('global_load_dwordx2' in t for x in body_blocks for t in blocks[x])
            print(f"loop {s}..{b}: {n_ins} instr, live-in {len(live_in[s])}, referenced {len(ref)}, live-through-untouched {len(through)}"
                  + ("  [node fetch]" if has_load else ""))
            if has_load and n_ins < 400:
                print("   untouched but live:", ' '.join(f'v{r}' for r in sorted(through)))

if len(sys.argv) > 3:
    print("per-block peaks (>= %s):" % sys.argv[3])
    for b in order:
        live = set(live_out[b]); pk = len(live); at = None
        for t in reversed(blocks[b]):
            d, u = defs_uses(t)
            if len(live | d) > pk: pk, at = len(live | d), t
            live -= d; live |= u
        if pk >= int(sys.argv[3]):
            loads = [t.split()[0] for t in blocks[b] if 'load' in t or 'ds_read' in t][:4]
            print(f"  {b:12s} n={len(blocks[b]):4d} peak={pk:3d} live_in={len(live_in[b]):3d} {' '.join(loads)}  @ {at}")
