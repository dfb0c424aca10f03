#!/usr/bin/env python3
"""Mechanical audit of the hand-counted `s_waitcnt` immediates and bare `s_barrier`s in the EMITTED gfx950 ISA.

Why ("What's weak" #1 and #3): the LDS-DMA pipelines of this library do not use `__syncthreads()` (its
release fence drains vmcnt(0)); they wait with hand-counted `s_waitcnt vmcnt(N)` where N = "vector-memory operations
this wave has issued AFTER the load the next barrier depends on" (loads and stores retire in order through ONE counter
on gfx9).  The count is a property of the instruction stream hipcc emits, not of the source: a spill, a split store, a
load sunk below the wait -- or, as in round 3, a code path that reaches the wait with a SHORTER history than the one
the count was derived for -- silently turns the wait into a no-op.  Round 3's attention backward shipped exactly that:
the peeled first item of `attn_bwd_dma_kernel<72>` reused the steady-state `vmcnt(10)` with 13 loads and no store
behind it, so waves crossed the barrier with up to ten loads in flight.

What this does: compiles the kernel sources to assembly with the product flags (`hipcc -S --cuda-device-only`, the same
compiler and options as the Makefile, so the text is what is inside libmaskdit_hip.so), builds the control-flow graph of
every watched kernel and runs a forward data-flow analysis over ALL paths (loop prologues, peeled iterations and steady
state alike) whose state is the queue of vector-memory operations that may still be in flight (oldest -> youngest; kinds:
D = LDS-DMA load, L = register load, S = store / atomic, X = scratch access) plus the LDS / scalar-memory queue (W =
LDS write, R = LDS read, M = scalar load).  `s_waitcnt vmcnt(N)` keeps the N youngest; `lgkmcnt(N)` likewise (with a
scalar load in flight only lgkmcnt(0) is meaningful: SMEM returns out of order).  Rules:

  barrier-lds   at every `s_barrier`, on every path, none of this wave's LDS WRITES is still in flight (otherwise another
                wave reads the location after the barrier before the write has landed);
  no-dma        at a source marker `; MDT_CHK no_dma` (placed in front of a barrier that hands an LDS-DMA'd buffer to the
                other waves) no LDS-DMA is in flight on any path -- attn_bwd_dma_kernel's buffer hand-over;
  vm-empty      at `; MDT_CHK vm_empty` nothing at all is in flight -- gemm_nt8's tile-top hand-over;
  spill         (kernels with HAND-counted waits) no scratch access inside a loop that contains a counted wait: a spill is
                one more vector-memory operation in the counted window, which hipcc's own waits account for and a
                hand-written immediate cannot; elsewhere spills only make the waits stricter and are listed;
  k-loop trace  gemm_nt8: the steady-state K loop (and the cross-tile pair that follows it) must show, phase by phase,
                exactly the issue sequence the counts were derived from -- c_issue(p) LDS-DMAs, then
                `vmcnt(wait_count(p))`, then the barrier -- and no other vector-memory operation
                (tests/test_host_cpu.py::test_gemm_nt8_wait_counts proves that MODEL safe; this proves the ISA IS the model);
  loader loop   gemm_nt8o (wave-specialised form): around the loaders' counted `vmcnt(NP / 2)` the loop issues nothing but
                LDS-DMAs, NP / 2 before the wait and NP / 2 after it, none inside a nested loop;
  ring phases   gemm_tn8: every phase of the steady loop issues 3 unconditional LDS-DMAs (+ 1 under the `mover2` branch
                for the 192-wide tile), waits with {3, 4} x (NSLOT - 2) in the two arms of the same branch, nothing else.

Usage:  python tools/check_waits.py [--define MACRO ...] [--verbose]      exit status 1 if any rule fails.
`--define MDT_REGRESS_R3_ATTN_WAIT` re-creates the round-3 wait: the tool must (and does) fail on it
(tests/test_host_cpu.py::test_isa_wait_audit)."""
from __future__ import annotations

import argparse
import hashlib
import os
import re
import subprocess
import sys
from collections import defaultdict
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'maskdit_amd', 'csrc')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result']  # = csrc/Makefile CXXFLAGS
CACHE = os.environ.get('MDT_ISA_CACHE', '/tmp/mdt_isa_cache')

# source file -> [(kernel-name regex, rule class)]
WATCH = {
    'attention.hip': [(r'attn_bwd_dma_kernel', 'attn'), (r'attn_(fwd|bwd)_sp_kernel', 'generic'),
                      (r'attn_bwd_(q|kv)_res_kernel', 'generic'), (r'attn_(fwd|bwd_dq|bwd_dkv)_kernel', 'generic')],
    'gemm_nt8_c0.hip': [(r'gemm_nt8_kernel', 'nt8')],
    'gemm_nt8_c1.hip': [(r'gemm_nt8_kernel', 'nt8')],
    'gemm_nt8_c2.hip': [(r'gemm_nt8_kernel', 'nt8')],
    'gemm_nt8_c3.hip': [(r'gemm_nt8_kernel', 'nt8')],
    'gemm_nt8_c4.hip': [(r'gemm_nt8_kernel', 'nt8')],
    'gemm_nt8_conv.hip': [(r'gemm_nt8_kernel', 'nt8')],
    'gemm_nt8o.hip': [(r'gemm_nt8o_kernel', 'nt8o')],
    'gemm_tn8.hip': [(r'gemm_tn8_kernel', 'tn8')],
    'norm.hip': [(r'ln_bwd_gate_split_kernel', 'generic'), (r'ln_modulate_(fwd|bwd)_kernel', 'generic')],
    'gemm.hip': [(r'gemm_(nt|tn)_kernel', 'generic')],
    # round 6: the fp32 path -- gemm_f32_dma_kernel moves its tiles by LDS-DMA (drained by the vmcnt(0) of __syncthreads) and
    # reads its fragments with inline-asm ds_read + hand-placed lgkmcnt waits
    'f32path.hip': [(r'gemm_f32_dma_kernel', 'generic'), (r'gemm_f32_kernel', 'generic'), (r'attn_f32_kernel', 'generic')],
}


# ------------------------------------------------------------------------------------------ compile
def source_digest(src: str, defines) -> str:
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith('.h') or f == src:
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), 'rb').read())
    h.update(open(os.path.join(ROOT, 'include', 'maskdit_hip.h'), 'rb').read())
    h.update(' '.join(FLAGS + sorted(defines)).encode())
    return h.hexdigest()[:20]


def compile_asm(src: str, defines=()) -> str:
    os.makedirs(CACHE, exist_ok=True)
    out = os.path.join(CACHE, f'{os.path.splitext(src)[0]}_{source_digest(src, defines)}.s')
    if not os.path.exists(out):
        cmd = [HIPCC] + FLAGS + [f'-D{d}' for d in defines] + ['--cuda-device-only', '-S', os.path.join(CSRC, src), '-o', out + '.tmp']
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc -S {src} failed:\n{r.stderr[-3000:]}')
        os.replace(out + '.tmp', out)
    return open(out).read()


# ------------------------------------------------------------------------------------------ parse
class Ins:
    __slots__ = ('line', 'op', 'args', 'kind', 'vm', 'lgkm', 'target', 'marker', 'hand')

    def __init__(self, line, op, args):
        self.line, self.op, self.args = line, op, args
        self.kind = None       # 'vm' | 'lds' | 'smem' | 'wait' | 'barrier' | 'branch' | 'cbranch' | 'end' | 'mfma' | 'marker'
        self.vm = None         # token for vm ops; for waits: vmcnt or None
        self.lgkm = None       # token for lds / smem ops; for waits: lgkmcnt or None
        self.target = None
        self.marker = None
        self.hand = False      # s_waitcnt written by hand (preceded by `; MDT_CHK hand_wait`)


def classify(line_no, text):
    t = text.strip()
    m = re.match(r';\s*MDT_CHK\s+(\S+)', t)
    if m:
        i = Ins(line_no, ';', t)
        i.kind, i.marker = 'marker', m.group(1)
        return i
    if not t or t[0] in ';.' or t.endswith(':'):
        return None
    t = t.split(';')[0].strip()
    if not t:
        return None
    parts = t.split(None, 1)
    op, args = parts[0], parts[1] if len(parts) > 1 else ''
    i = Ins(line_no, op, args)
    if op.startswith(('global_load_lds', 'buffer_load')) and (op.startswith('global_load_lds') or re.search(r'\blds\b', args)):
        i.kind, i.vm = 'vm', 'D'
    elif op.startswith(('global_load', 'buffer_load', 'flat_load')):
        i.kind, i.vm = 'vm', 'L'
    elif op.startswith(('scratch_load', 'scratch_store')):
        i.kind, i.vm = 'vm', 'X'
    elif op.startswith(('global_store', 'buffer_store', 'flat_store', 'global_atomic', 'buffer_atomic', 'flat_atomic')):
        i.kind, i.vm = 'vm', 'S'
    elif op.startswith('ds_'):
        i.kind = 'lds'
        i.lgkm = 'R' if (op.startswith('ds_read') or op.startswith('ds_bpermute') or op.startswith('ds_permute')
                         or op.startswith('ds_swizzle')) else 'W'
    elif op.startswith(('s_load_', 's_buffer_load', 's_scratch_load')):
        i.kind, i.lgkm = 'smem', 'M'
    elif op == 's_waitcnt':
        i.kind = 'wait'
        m = re.search(r'vmcnt\((\d+)\)', args)
        i.vm = int(m.group(1)) if m else None
        m = re.search(r'lgkmcnt\((\d+)\)', args)
        i.lgkm = int(m.group(1)) if m else None
        if not re.search(r'cnt\(', args):  # raw immediate: gfx9 encoding
            v = int(args, 0)
            i.vm = (v & 15) | ((v >> 14) & 3) << 4
            i.lgkm = (v >> 8) & 15
            if i.vm == 63:
                i.vm = None
            if i.lgkm == 15:
                i.lgkm = None
    elif op == 's_barrier':
        i.kind = 'barrier'
    elif op == 's_branch':
        i.kind, i.target = 'branch', args.strip()
    elif op.startswith('s_cbranch'):
        i.kind, i.target = 'cbranch', args.strip()
    elif op in ('s_endpgm', 's_setpc_b64', 's_trap'):
        i.kind = 'end'
    elif op.startswith('v_mfma'):
        i.kind = 'mfma'
    else:
        i.kind = 'other'
    return i


class Block:
    def __init__(self, name):
        self.name, self.ins, self.succ, self.pred = name, [], [], []


class Kernel:
    def __init__(self, name, lines, first_line):
        self.name, self.first_line = name, first_line
        self.blocks: dict = {}
        self.order: list = []
        cur = Block('entry')
        self.blocks['entry'] = cur
        self.order.append(cur)
        anon = 0
        pending_hand = False
        for k, text in enumerate(lines):
            lab = re.match(r'^(\.LBB[\w.]+):', text)
            if lab:
                nb = Block(lab.group(1))
                self.blocks[nb.name] = nb
                self.order.append(nb)
                if not cur.ins or cur.ins[-1].kind not in ('branch', 'end'):
                    cur.succ.append(nb.name)
                cur = nb
                continue
            ins = classify(first_line + k, text)
            if ins is None:
                continue
            if ins.kind == 'marker' and ins.marker == 'hand_wait':
                pending_hand = True
                continue
            if ins.kind == 'wait' and pending_hand:
                ins.hand, pending_hand = True, False
            cur.ins.append(ins)
            if ins.kind in ('branch', 'cbranch', 'end'):
                if ins.kind != 'end':
                    cur.succ.append(ins.target)
                anon += 1
                nb = Block(f'{cur.name}+{anon}')
                self.blocks[nb.name] = nb
                self.order.append(nb)
                if ins.kind == 'cbranch':
                    cur.succ.append(nb.name)
                cur = nb
        for b in self.order:
            b.succ = [s for s in b.succ if s in self.blocks]
            for s in b.succ:
                self.blocks[s].pred.append(b.name)

    # ---- loops as hipcc lays them out: a back edge i -> j (j <= i in layout order) closes the loop [j, i]
    def loops(self):
        """[(first index, last index)] of every natural loop, by layout range; `innermost` = contains no other loop."""
        pos = {b.name: k for k, b in enumerate(self.order)}
        rng = set()
        for b in self.order:
            for t in b.succ:
                if pos[t] <= pos[b.name]:
                    rng.add((pos[t], pos[b.name]))
        return sorted(rng)

    def loop_blocks(self, r):
        return [b for b in self.order[r[0]:r[1] + 1]]

    def innermost(self, pred=None):
        ls = [r for r in self.loops() if pred is None or pred(self.loop_blocks(r))]
        return [r for r in ls if not any(o != r and r[0] <= o[0] and o[1] <= r[1] for o in ls)]

    # ---- strongly connected components of the CFG
    def sccs(self):
        index, low, on, stack, out, counter = {}, {}, set(), [], [], [0]
        sys.setrecursionlimit(100000)

        def strong(v):
            index[v] = low[v] = counter[0]
            counter[0] += 1
            stack.append(v)
            on.add(v)
            for w in self.blocks[v].succ:
                if w not in index:
                    strong(w)
                    low[v] = min(low[v], low[w])
                elif w in on:
                    low[v] = min(low[v], index[w])
            if low[v] == index[v]:
                comp = []
                while True:
                    w = stack.pop()
                    on.discard(w)
                    comp.append(w)
                    if w == v:
                        break
                if len(comp) > 1 or v in self.blocks[v].succ:
                    out.append(set(comp))
        for b in self.order:
            if b.name not in index:
                strong(b.name)
        return out


def parse_kernels(asm: str):
    lines = asm.split('\n')
    out = []
    k = 0
    while k < len(lines):
        m = re.match(r'^(_Z\w+):\s*; @', lines[k])
        if m:
            name = m.group(1)
            e = k + 1
            while e < len(lines) and not lines[e].startswith('.Lfunc_end'):
                e += 1
            out.append(Kernel(name, lines[k + 1:e], k + 2))
            k = e
        k += 1
    return out


def demangle(names):
    try:
        r = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True)
        d = r.stdout.strip().split('\n')
        if len(d) == len(names):
            return dict(zip(names, d))
    except Exception:
        pass
    return {n: n for n in names}


# ------------------------------------------------------------------------------------------ data flow
CAP = 63  # vmcnt is 6 bits: the hardware never has more than 63 operations of a wave in flight behind the counter
LCAP = 15


# Both counters retire IN ORDER (vector memory always; LDS operations among themselves -- scalar loads do not, which is why
# only lgkmcnt(0) means anything while one is in flight).  "Some X is in flight" is therefore equivalent to "the YOUNGEST X
# is in flight", and an exact abstraction of what the rules ask is small:
#   n   vector-memory operations possibly in flight (0..63)
#   dD  operations issued after the youngest LDS-DMA (None: no DMA in flight)      -> a DMA is in flight iff dD < n
#   dW  LDS/SMEM operations issued after the youngest LDS write (None: none)       -> in flight until lgkmcnt(N <= dW)
#   sm  a scalar load may be in flight
# Each abstract state carries ONE example history string (oldest -> youngest) for the messages.
def step(state, ins):
    n, dD, dW, sm = state
    if ins.kind == 'vm':
        n = min(n + 1, CAP)
        dD = 0 if ins.vm == 'D' else (None if dD is None else min(dD + 1, CAP))
    elif ins.kind == 'lds':
        dW = 0 if ins.lgkm == 'W' else (None if dW is None else min(dW + 1, LCAP))
    elif ins.kind == 'smem':
        sm = True
        dW = None if dW is None else min(dW + 1, LCAP)
    elif ins.kind == 'wait':
        if ins.vm is not None:
            n = min(n, ins.vm)
            if dD is not None and dD >= n:
                dD = None
        if ins.lgkm is not None:
            if ins.lgkm == 0:
                dW, sm = None, False
            elif not sm and dW is not None and dW >= ins.lgkm:
                dW = None
    return (n, dD, dW, sm)


def step_example(ex, ins):
    vmq, lgq = ex
    if ins.kind == 'vm':
        vmq = (vmq + ins.vm)[-CAP:]
    elif ins.kind in ('lds', 'smem'):
        lgq = (lgq + ins.lgkm)[-LCAP:]
    elif ins.kind == 'wait':
        if ins.vm is not None:
            vmq = vmq[len(vmq) - ins.vm:] if ins.vm else ''
        if ins.lgkm is not None:
            if ins.lgkm == 0:
                lgq = ''
            elif 'M' not in lgq and ins.lgkm < len(lgq):
                lgq = lgq[len(lgq) - ins.lgkm:]
    return (vmq, lgq)


def analyse(kern: Kernel):
    """-> {block name: {abstract entry state: example (vm history, lgkm history)}} -- the fixed point over all paths.
    Every (block, state) pair is pushed through the block once; only the instructions that touch a counter are stepped."""
    live = {b.name: [i for i in b.ins if i.kind in ('vm', 'lds', 'smem', 'wait')] for b in kern.order}
    entry = {b.name: {} for b in kern.order}
    entry['entry'][(0, None, None, False)] = ('', '')
    work = [('entry', (0, None, None, False))]
    while work:
        name, st0 = work.pop()
        st, ex = st0, entry[name][st0]
        for ins in live[name]:
            st = step(st, ins)
            ex = step_example(ex, ins)
        for s in kern.blocks[name].succ:
            if st not in entry[s]:
                entry[s][st] = ex
                work.append((s, st))
    return entry


class Report:
    def __init__(self, verbose=False):
        self.errors, self.notes, self.verbose = [], [], verbose

    def err(self, kern, msg):
        self.errors.append(f'{kern}: {msg}')

    def note(self, kern, msg):
        self.notes.append(f'{kern}: {msg}')


def walk_states(kern, entry, visit):
    """visit(block, ins, {abstract state: example}) for every instruction of every reachable block."""
    for blk in kern.order:
        sts = entry[blk.name]
        if not sts:
            continue
        for ins in blk.ins:
            if ins.kind not in ('vm', 'lds', 'smem', 'wait', 'barrier', 'marker'):
                continue
            visit(blk, ins, sts)
            if ins.kind in ('barrier', 'marker'):
                continue
            nxt = {}
            for st, ex in sts.items():
                nxt.setdefault(step(st, ins), step_example(ex, ins))
            sts = nxt


def rule_generic(kern, pretty, entry, rep, klass):
    counted = []

    def visit(blk, ins, sts):
        if ins.kind == 'barrier':
            bad = [ex[1] for st, ex in sts.items() if st[2] is not None]
            if bad:
                rep.err(pretty, f'line {ins.line}: s_barrier reached with an LDS write possibly in flight (lgkm history {bad[:3]})')
        elif ins.kind == 'marker':
            if ins.marker == 'no_dma':
                bad = sorted((ex[0] for st, ex in sts.items() if st[1] is not None and st[1] < st[0]), key=len)
                if bad:
                    rep.err(pretty, f'line {ins.line}: MDT_CHK no_dma: an LDS-DMA may still be in flight; e.g. the in-flight queue '
                                    f'(oldest -> youngest; D = LDS-DMA, L = register load, S = store) "{bad[0]}"')
            elif ins.marker == 'vm_empty':
                bad = sorted((ex[0] for st, ex in sts.items() if st[0] > 0), key=len)
                if bad:
                    rep.err(pretty, f'line {ins.line}: MDT_CHK vm_empty: vector-memory operations may be in flight: "{bad[0]}"')
            elif ins.marker == 'nt8o_loader_wait':
                pass  # rule_nt8o
            else:
                rep.err(pretty, f'line {ins.line}: unknown marker {ins.marker}')
        elif ins.kind == 'wait' and ins.vm and ins.hand:
            counted.append((blk, ins, sorted({ex[0] for ex in sts.values()})))

    walk_states(kern, entry, visit)
    # spills inside the (innermost) loops that carry counted waits
    hot = set()
    if True:  # (hipcc's OWN counted waits account for the spills it inserts; the hand-written immediates cannot)
        for r in kern.innermost(lambda bl: any(i.kind == 'wait' and i.vm and i.hand for b in bl for i in b.ins)):
            hot |= {b.name for b in kern.loop_blocks(r)}
    n_scratch = 0
    for blk in kern.order:
        for ins in blk.ins:
            if ins.kind == 'vm' and ins.vm == 'X':
                n_scratch += 1
                if blk.name in hot:
                    rep.err(pretty, f'line {ins.line}: scratch access ({ins.op}) inside a loop with counted vmcnt waits')
    if n_scratch:
        rep.note(pretty, f'{n_scratch} scratch accesses outside counted loops (they only make counted waits stricter)')
    if rep.verbose:
        for blk, ins, queues in counted:
            tails = sorted({q[-ins.vm:] if len(q) >= ins.vm else '<' + q for q in queues})
            rep.note(pretty, f'line {ins.line}: vmcnt({ins.vm}) -- youngest {ins.vm} in flight, example histories: {tails[:6]}'
                             + (' ...' if len(tails) > 6 else ''))
    return counted


# ---- gemm_nt8: K-loop trace against the wait model of gemm_nt8_impl.h -------------------------------------------------
def nt8_c_issue(p, nf, rpp):
    return 1 + (rpp if p < nf else 0)


def nt8_wait_count(p, nf, rpp):
    w = rpp if ((p + 2) & 3) < nf else 0
    for d in range(5, -1, -1):
        w += nt8_c_issue((p - d) % 4, nf, rpp)
    if p == 2:
        w = min(w, (4 - nf) + nt8_c_issue(0, nf, rpp) + nt8_c_issue(1, nf, rpp) + nt8_c_issue(2, nf, rpp))
    return w


def linear_events(kern, names):
    ev = []
    for n in names:
        for ins in kern.blocks[n].ins:
            if ins.kind == 'vm':
                ev.append((ins.vm, ins.line))
            elif ins.kind == 'wait' and ins.vm is not None:
                ev.append((('W', ins.vm), ins.line))
            elif ins.kind == 'barrier':
                ev.append(('B', ins.line))
    return ev


def rule_nt8(kern, pretty, entry, rep):
    m = re.search(r'gemm_nt8_kernel<(\d+), (\d+), (\d+)', pretty)
    if not m:
        rep.err(pretty, 'cannot read the template arguments')
        return
    nf, wr = int(m.group(1)), int(m.group(2))
    rpp = 2 // wr
    expect = []
    for _half in range(2):
        for p in range(4):
            expect += ['D'] * nt8_c_issue(p, nf, rpp) + [('W', nt8_wait_count(p, nf, rpp)), 'B']
    # the K loop = the innermost loop with MFMAs and LDS-DMA
    inner = kern.innermost(lambda bl: any(i.kind == 'mfma' for b in bl for i in b.ins)
                           and any(i.kind == 'vm' and i.vm == 'D' for b in bl for i in b.ins))
    if len(inner) != 1:
        rep.err(pretty, f'expected ONE K loop with MFMA + LDS-DMA, found {len(inner)}')
        return
    names = [b.name for b in kern.loop_blocks(inner[0])]
    comp = set(names)
    if any(len([s for s in kern.blocks[n].succ if s in comp]) > 1 for n in names):
        rep.err(pretty, 'the K loop has internal control flow: the counted waits assume ONE straight-line issue sequence')
        return
    ev = linear_events(kern, names)
    got = [e for e, _ in ev]
    # hipcc may rotate the loop (the laid-out body then starts in the middle of a phase): compare as CYCLIC sequences
    rot = next((r for r in range(len(expect)) if got == expect[r:] + expect[:r]), None) if len(got) == len(expect) else None
    if rot is None:
        k = next((i for i, (a, b) in enumerate(zip(got, expect)) if a != b), min(len(got), len(expect)))
        rep.err(pretty, f'K-loop issue/wait trace is not (a rotation of) the model; first difference at event {k} '
                        f'(line {ev[k][1] if k < len(ev) else "?"}): ISA {got[max(0, k - 3):k + 4]} vs model {expect[max(0, k - 3):k + 4]}')
        return
    expect = expect[rot:] + expect[:rot]
    # the cross-tile pair right after the loop (PAIR_BODY(1)): same sequence before anything else touches vector memory
    # (a rotated loop can leave from a block in the middle of its laid-out body: the continuation then starts at that
    # block's position in the cyclic sequence)
    exits, seen_ev = [], 0
    for n in names:
        seen_ev += len(linear_events(kern, [n]))
        exits += [(t, seen_ev % len(expect)) for t in kern.blocks[n].succ if t not in comp]
    if not exits:
        rep.err(pretty, 'K loop has no exit')
        return
    for first, off in exits:
        want = expect[off:] + expect[:off]
        # every path from the loop exit (the `more tiles?` diamond re-points the operand bases on one arm) through one
        # more K-tile pair
        paths, stack = [], [(first, [])]
        while stack and len(paths) < 64:
            cur, ev2 = stack.pop()
            ev2 = ev2 + linear_events(kern, [cur])
            if len(ev2) >= len(want) or not kern.blocks[cur].succ:
                paths.append(ev2)
                continue
            for nx in kern.blocks[cur].succ:
                stack.append((nx, ev2))
        if not paths:
            rep.err(pretty, 'no path from the K loop to the cross-tile pair')
            return
        for ev2 in paths:
            got2 = [e for e, _ in ev2][:len(want)]
            if got2 != want:
                k = next((i for i, (x, y) in enumerate(zip(got2, want)) if x != y), min(len(got2), len(want)))
                rep.err(pretty, f'cross-tile pair after the K loop differs from the model at event {k} '
                                f'(line {ev2[k][1] if k < len(ev2) else "?"}): ISA {got2[max(0, k - 3):k + 4]} vs model {want[max(0, k - 3):k + 4]}')
                return
    rep.note(pretty, f'K loop + cross-tile pair match the wait model (NF {nf}, {4 * wr} waves): waits '
                     f'{[nt8_wait_count(p, nf, rpp) for p in range(4)]}')


# ---- gemm_tn8: ring phases -------------------------------------------------------------------------------------------
def rule_tn8(kern, pretty, entry, rep):
    m = re.search(r'gemm_tn8_kernel<(true|false), (\d+), (true|false), (true|false)>', pretty)
    if not m:
        rep.err(pretty, 'cannot read the template arguments')
        return
    yf, fine = int(m.group(2)), m.group(3) == 'true'
    nslot = 5 if yf == 6 else 6
    late_ok = not fine  # the round-2 staggered form: waves 4-7 wait one slot less (kept for A/B runs)
    loops = kern.innermost(lambda bl: any(i.kind == 'mfma' for b in bl for i in b.ins)
                           and any(i.kind == 'vm' and i.vm == 'D' for b in bl for i in b.ins))
    if not loops:
        rep.err(pretty, 'no ring loop found')
        return
    allowed = {3 * (nslot - 2)} | ({4 * (nslot - 2)} if yf == 6 else set()) | ({3 * (nslot - 3)} if late_ok else set()) | {0}
    for r in loops:
        names = [b.name for b in kern.loop_blocks(r)]
        comp = set(names)
        # phases = stretches between barriers in layout order
        uncond = cond = 0
        waits = set()
        nphase = 0
        steady = not any(i.kind == 'wait' and i.vm == 0 for n in names for i in kern.blocks[n].ins)
        for n in names:
            blk = kern.blocks[n]
            # a block is "conditional" if some predecessor inside the loop branches AROUND it (has another successor)
            is_cond = any(len(kern.blocks[p].succ) > 1 for p in blk.pred if p in comp) and len(blk.ins) < 40
            for ins in blk.ins:
                if ins.kind == 'vm':
                    if ins.vm != 'D':
                        rep.err(pretty, f'line {ins.line}: {ins.op} inside the ring loop (only LDS-DMA refills are counted)')
                    elif is_cond:
                        cond += 1
                    else:
                        uncond += 1
                elif ins.kind == 'wait' and ins.vm is not None:
                    waits.add(ins.vm)
                elif ins.kind == 'barrier':
                    if fine and steady:
                        want_c = 1 if yf == 6 else 0
                        if (uncond, cond) != (3, want_c):
                            rep.err(pretty, f'line {ins.line}: ring phase issues {uncond} unconditional + {cond} conditional LDS-DMAs, '
                                            f'expected 3 + {want_c}')
                        want_w = {3 * (nslot - 2)} | ({4 * (nslot - 2)} if yf == 6 else set())
                        if {w for w in waits if w != 63} != want_w:
                            rep.err(pretty, f'line {ins.line}: ring phase waits {sorted(waits)}, expected {sorted(want_w)}')
                        nphase += 1
                    bad = {w for w in waits if w != 63} - allowed
                    if bad:
                        rep.err(pretty, f'line {ins.line}: unexpected vmcnt operands {sorted(bad)} in the ring loop (allowed {sorted(allowed)})')
                    uncond = cond = 0
                    waits = set()
        if fine and steady:
            rep.note(pretty, f'steady ring loop: {nphase} phases of {"3 (+1 conditional)" if yf == 6 else "3"} LDS-DMAs, waits {sorted(w for w in allowed if w)}')


# ------------------------------------------------------------------------------------------ driver
# ---- gemm_nt8o: the loader waves' counted wait -------------------------------------------------------------------------
def rule_nt8o(kern, pretty, entry, rep):
    """gemm_nt8o.hip: a loader wave publishes K-tile g - 1 behind `vmcnt(NP / 2)` issued after the first NP / 2 LDS-DMA
    pieces of K-tile g.  That is right iff the wave's queue holds NOTHING but LDS-DMAs there and exactly NP / 2 of them
    belong to K-tile g: in the smallest loop around every `; MDT_CHK nt8o_loader_wait` the only vector-memory
    instructions must be LDS-DMAs, N of them in layout order before the wait and N after it (N = the wait's immediate),
    and none may hide in a nested loop (the counter polls are LDS-only spins)."""
    pos = {b.name: k for k, b in enumerate(kern.order)}
    loops = kern.loops()
    found = 0
    for blk in kern.order:
        for k, ins in enumerate(blk.ins):
            if ins.kind != 'marker' or ins.marker != 'nt8o_loader_wait':
                continue
            found += 1
            wait = next((i for i in blk.ins[k + 1:] if i.kind == 'wait'), None)
            if wait is None or not wait.hand or not wait.vm:
                rep.err(pretty, f'line {ins.line}: nt8o_loader_wait marker without a counted hand wait behind it')
                continue
            N = wait.vm
            around = [r for r in loops if r[0] <= pos[blk.name] <= r[1]]
            if not around:
                rep.err(pretty, f'line {ins.line}: counted loader wait outside any loop')
                continue
            lo, hi = min(around, key=lambda r: r[1] - r[0])
            inner = [r for r in loops if lo <= r[0] and r[1] <= hi and (r[0], r[1]) != (lo, hi)]
            before = after = 0
            for b in kern.order[lo:hi + 1]:
                nested = any(r[0] <= pos[b.name] <= r[1] for r in inner)
                for i in b.ins:
                    if i.kind != 'vm':
                        continue
                    if i.vm != 'D' or nested:
                        rep.err(pretty, f'line {i.line}: {i.op} ({"nested loop" if nested else "kind " + i.vm}) inside the loader loop of the '
                                        f'counted wait at line {wait.line}: its queue must hold LDS-DMAs only')
                    elif i.line < wait.line:
                        before += 1
                    else:
                        after += 1
            if before != N or after != N:
                rep.err(pretty, f'line {wait.line}: loader loop issues {before} LDS-DMAs before and {after} after vmcnt({N}); both must be {N}')
    if found == 0:
        rep.err(pretty, 'no counted loader wait found (expected one per loader walk)')


def check_file(src, defines, rep):
    asm = compile_asm(src, defines)
    kernels = parse_kernels(asm)
    names = demangle([k.name for k in kernels])
    done = 0
    for kern in kernels:
        pretty = names[kern.name]
        short = re.sub(r'\(.*$', '', pretty.replace('void ', ''))
        klass = next((c for pat, c in WATCH[src] if re.search(pat, pretty)), None)
        if klass is None:
            continue
        entry = analyse(kern)
        rule_generic(kern, f'{src}: {short}', entry, rep, klass)
        if klass == 'nt8':
            rule_nt8(kern, f'{src}: {short}', entry, rep)
        elif klass == 'tn8':
            rule_tn8(kern, f'{src}: {short}', entry, rep)
        elif klass == 'nt8o':
            rule_nt8o(kern, f'{src}: {short}', entry, rep)
        done += 1
    return done


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--define', action='append', default=[], help='extra -D macro (e.g. MDT_REGRESS_R3_ATTN_WAIT)')
    ap.add_argument('--files', nargs='*', default=sorted(WATCH))
    ap.add_argument('--verbose', action='store_true', help='list the in-flight tail at every counted wait')
    ap.add_argument('--jobs', type=int, default=min(8, os.cpu_count() or 1))
    a = ap.parse_args(argv)
    rep = Report(a.verbose)
    with ThreadPoolExecutor(a.jobs) as ex:  # compile in parallel (hipcc is the cost), analyse serially
        list(ex.map(lambda f: compile_asm(f, a.define), a.files))
    total = 0
    for f in a.files:
        total += check_file(f, a.define, rep)
    for n in rep.notes:
        print('note ', n)
    for e in rep.errors:
        print('ERROR', e)
    print(f'check_waits: {total} kernels audited, {len(rep.errors)} errors')
    return 1 if rep.errors else 0


if __name__ == '__main__':
    sys.exit(main())
