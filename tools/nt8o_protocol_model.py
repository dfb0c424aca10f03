"""Exhaustive model check of gemm_nt8o's LDS-counter protocol (maskdit_amd/csrc/gemm_nt8o.hip) -- CPU, no GPU.

The kernel's waves synchronise through monotonic counters instead of s_barrier:
    full[s]  += 1 by each LOADER when its share of a fill of ring stage s has landed   (fill n ready   <=> full[s]  >= NL n)
    empty[s] += 1 by each MMA wave after its last read of a fill of stage s            (fill n drained <=> empty[s] >= W n)
    ydone[w] += 1 by MMA wave w once its y stores of a tile are acknowledged          (tile T in L2   <=> every ydone[w] >= T + 1)
(`summed_ydone=True` re-creates the first form of the kernel -- ONE ydone word, threshold W (T + 1) -- on which this check fails:
two waves that have finished the last tile lift the sum over the previous tile's threshold before the others acknowledged theirs.)
This file restates each role's walk (same order of waits / adds / reads / LDS-DMA issues as the kernel, with the use
counters and thresholds computed the same way) as a small-step transition system and explores EVERY interleaving of the
roles and of the asynchronous LDS-DMA completions (in order per loader, like vmcnt) for small parameters, checking:
    * no deadlock: every reachable state with an unfinished role has an enabled step, and every run terminates;
    * a fragment read of (stage, fill) only happens after ALL loaders' pieces of that fill have landed, and never after
      any loader has started overwriting the stage with the next fill;
    * a loader never issues into a stage whose previous fill some MMA wave has not finished reading;
    * an epilogue wave picks tile T up only after every MMA wave's y stores of tile T were acknowledged.
The model abstracts the data path (a "read" is one event per phase group) but not the counting: stage rotation over the
3-stage ring across tile boundaries, K-tile counts that are not multiples of 3, the half-way publication of the previous
K-tile behind vmcnt(NP / 2), the deferred ydone publication at K-tile `sig_kt` of the next tile.
    python tools/nt8o_protocol_model.py            # prints the explored state counts
tests/test_host_cpu.py::test_nt8o_counter_protocol_model runs it."""
import sys
from collections import deque

NS = 3


def mma_program(w, W, NL, nk, tiles, plain, summed):
    """Mirror of the MMA role (gemm_nt8o.hip, `if (wave < 4)`): yields (op, args)."""
    st, use = 0, 1
    ops = [('wait', ('full', 0), NL)]
    ops.append(('read', 0, 1))  # first fragments of K-tile 0
    sig_kt = 4 if nk > 4 else nk - 1
    for T in range(tiles):
        for kt in range(nk):
            stn = 0 if st == NS - 1 else st + 1
            usen = use + (1 if stn == 0 else 0)
            has_next = (kt + 1 < nk) or (T + 1 < tiles)
            if not plain and T > 0 and kt == sig_kt:
                ops.append(('acked', T - 1))        # s_waitcnt vmcnt(0): tile T - 1's y stores acknowledged
                ops.append(('add', ('ydone',) if summed else ('ydone', w)))
            ops.append(('read', st, use))           # phases 0-2: A fragments of the current stage
            if has_next:
                ops.append(('wait', ('full', stn), NL * usen))
            ops.append(('done', st, use))           # the last read of this fill retired (LDS executes in order)
            ops.append(('add', ('empty', st)))
            if has_next:
                ops.append(('read', stn, usen))     # phase 3: the next K-tile's first fragments
            st, use = stn, usen
        ops.append(('ystore', T))
    if not plain:
        ops.append(('acked', tiles - 1))
        ops.append(('add', ('ydone',) if summed else ('ydone', w)))
    return ops


def loader_program(l, W, NL, nk, tiles):
    """Mirror of the loader role: per K-tile [wait empty] issue half, vmcnt(NP/2) + publish the previous K-tile, issue half."""
    st, st_prev, use, first = 0, 0, 1, True
    prev = None
    ops = []
    for T in range(tiles):
        for kt in range(nk):
            if use > 1:
                ops.append(('wait', ('empty', st), W * (use - 1)))
            ops.append(('dma', st, use, 0))
            if not first:
                ops.append(('landed_before', (st, use, 0)))  # vmcnt(NP / 2): everything older than this half has landed
                ops.append(('add', ('full', st_prev)))
            ops.append(('dma', st, use, 1))
            first = False
            st_prev = st
            st = 0 if st == NS - 1 else st + 1
            use += 1 if st == 0 else 0
    ops.append(('landed_before', None))  # vmcnt(0)
    ops.append(('add', ('full', st_prev)))
    return ops


def epilogue_program(e, W, tiles, summed):
    ops = []
    for T in range(tiles):
        if summed:
            ops.append(('wait', ('ydone',), W * (T + 1)))
        else:
            for w in range(W):
                ops.append(('wait', ('ydone', w), T + 1))
        ops.append(('pickup', T))
    return ops


def check(W=2, NL=2, NE=1, nk=4, tiles=2, summed_ydone=False, max_states=2_000_000):
    plain = NE == 0
    progs = [mma_program(w, W, NL, nk, tiles, plain, summed_ydone) for w in range(W)] + [loader_program(l, W, NL, nk, tiles) for l in range(NL)] + \
            [epilogue_program(e, W, tiles, summed_ydone) for e in range(NE)]
    kinds = ['mma'] * W + ['ld'] * NL + ['ep'] * NE
    n = len(progs)
    # state: (pcs, landed) -- counters and everything else are functions of the pcs (all adds are unconditional), landed[l] =
    # number of this loader's issued DMA halves that have completed (in order)
    def derived(pcs):
        cnt = {}
        issued = [[] for _ in range(NL)]       # per loader: list of (stage, fill, half) in issue order
        done_reads = set()                     # (wave, stage, fill)
        acked = set()
        for a in range(n):
            for op in progs[a][:pcs[a]]:
                if op[0] == 'add':
                    cnt[op[1]] = cnt.get(op[1], 0) + 1
                elif op[0] == 'dma':
                    issued[a - W].append(op[1:])
                elif op[0] == 'done':
                    done_reads.add((a, op[1], op[2]))
                elif op[0] == 'acked':
                    acked.add((a, op[1]))
        return cnt, issued, done_reads, acked

    start = (tuple([0] * n), tuple([0] * NL))
    seen = {start}
    q = deque([start])
    explored = 0
    while q:
        pcs, landed = q.popleft()
        explored += 1
        if explored > max_states:
            raise RuntimeError('state space larger than expected')
        cnt, issued, done_reads, acked = derived(pcs)
        succ = []
        # asynchronous completions: the oldest outstanding DMA half of any loader may land
        for l in range(NL):
            if landed[l] < len(issued[l]):
                succ.append((pcs, landed[:l] + (landed[l] + 1,) + landed[l + 1:]))
        for a in range(n):
            if pcs[a] >= len(progs[a]):
                continue
            op = progs[a][pcs[a]]
            ok = True
            if op[0] == 'wait':
                ok = cnt.get(op[1], 0) >= op[2]
            elif op[0] == 'landed_before':
                l = a - W
                if op[1] is None:
                    ok = landed[l] == len(issued[l])
                else:
                    ok = landed[l] >= issued[l].index(op[1])  # every half issued BEFORE this one has landed
            elif op[0] == 'read':
                _, s_, f_ = op
                for l in range(NL):
                    halves = [k for k, d in enumerate(issued[l]) if d[0] == s_ and d[1] == f_]
                    assert len(halves) == 2 and all(k < landed[l] for k in halves), \
                        f'MMA wave {a} reads stage {s_} fill {f_} before loader {l} has landed it (pcs {pcs})'
                    assert not any(d[0] == s_ and d[1] == f_ + 1 for d in issued[l]), \
                        f'MMA wave {a} reads stage {s_} fill {f_} while loader {l} is overwriting it (pcs {pcs})'
            elif op[0] == 'dma':
                _, s_, f_, _h = op
                if f_ > 1:
                    for w in range(W):
                        assert (w, s_, f_ - 1) in done_reads, f'loader {a - W} refills stage {s_} (fill {f_}) before MMA wave {w} finished fill {f_ - 1}'
            elif op[0] == 'pickup':
                for w in range(W):
                    assert (w, op[1]) in acked, f'epilogue picks tile {op[1]} up before MMA wave {w} acknowledged its y stores'
            if ok:
                succ.append((pcs[:a] + (pcs[a] + 1,) + pcs[a + 1:], landed))
        finished = all(pcs[a] >= len(progs[a]) for a in range(n))
        if not succ:
            assert finished, f'DEADLOCK at pcs {pcs}: ' + ', '.join(f'{kinds[a]}:{progs[a][pcs[a]] if pcs[a] < len(progs[a]) else "end"}' for a in range(n))
        for s in succ:
            if s not in seen:
                seen.add(s)
                q.append(s)
    return explored


if __name__ == '__main__':
    for cfg in (dict(W=2, NL=2, NE=1, nk=4, tiles=2), dict(W=2, NL=2, NE=1, nk=5, tiles=2), dict(W=2, NL=1, NE=0, nk=7, tiles=2),
                dict(W=1, NL=3, NE=1, nk=4, tiles=3)):
        print(cfg, '->', check(**cfg), 'states, no deadlock, no hazard')
    try:
        check(W=2, NL=2, NE=1, nk=4, tiles=2, summed_ydone=True)
        print('summed ydone: NOT caught')
        sys.exit(1)
    except AssertionError as e:
        print('summed ydone (first form of the kernel) is caught:', e)
    sys.exit(0)
