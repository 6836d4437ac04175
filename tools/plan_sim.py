#!/usr/bin/env python3
"""plan_sim.py -- discrete-event model of the one-launch tree build (csrc/p2mt_mmr.hip, k_tree_plan).

A plan is a list of work items in TICKET order; a workgroup that becomes resident takes the next ticket, waits for the
items its inputs come from, runs, and publishes.  The chip model: 256 CUs x 4 resident 256-lane workgroups; a workgroup
is one wavefront per SIMD; a SIMD with k busy wavefronts gives each min(1, 2/k) of the speed of a lone one (measured:
a lone wavefront issues at half the rate two reach together, tools/ubench_valu.hip).  Durations are "lone" microseconds:

  S<lv>  per-lane subtree of 2^lv leaves        (2^lv - 1) x T_HASH
  U      one two_to_one per lane, 256 nodes     T_HASH
  Q      four lanes per node, 64 nodes          T_QUAD
  W      one wavefront per node, 4 nodes        T_WAVE

The tool prints the makespan of a plan and where the chip idles; it exists to choose the ticket ORDER (the kernel is
correct under any topological order).
"""
import argparse
import heapq
from collections import defaultdict

T_HASH = 36.0   # us, one lane-per-hash two_to_one by a lone wavefront (10.1 k VALU instr at ~half issue)
T_QUAD = 13.0
T_WAVE = 7.0
HOP = 2.0       # us: flag publish -> seen by a polling consumer + the sc1 loads of the children


class Item:
    __slots__ = ("kind", "h", "j0", "n", "work", "deps", "id", "start", "end", "ready")

    def __init__(self, kind, h, j0, n, work):
        self.kind, self.h, self.j0, self.n, self.work = kind, h, j0, n, work
        self.deps = []
        self.start = self.end = self.ready = None


def build_items(log_n, lv_of_block, upper_kind):
    """lv_of_block(b) -> subtree levels of stage-1 block group; upper_kind(h, n_nodes, j) -> 'U' | 'Q' | 'W'."""
    raise NotImplementedError


def simulate(items, slots_per_cu=4, n_cu=256, verbose=False):
    """items in ticket order, deps = list of item objects.  Returns makespan and fills start / end."""
    n_slots = slots_per_cu * n_cu
    # CU state: list of running (item, remaining lone-us); processor sharing with rate min(1, 2/k)
    cu_items = [dict() for _ in range(n_cu)]   # item id -> remaining
    cu_last = [0.0] * n_cu
    cu_waiting = [0] * n_cu                    # resident but polling
    t = 0.0
    nxt = 0
    events = []  # (time, seq, kind, payload)
    seq = 0
    busy_area = 0.0

    def rate(k):
        return 1.0 if k <= 2 else 2.0 / k

    def advance(cu, now):
        k = len(cu_items[cu])
        if k and now > cu_last[cu]:
            r = rate(k) * (now - cu_last[cu])
            for i in cu_items[cu]:
                cu_items[cu][i] -= r
        cu_last[cu] = now

    def next_finish(cu):
        k = len(cu_items[cu])
        if not k:
            return None
        i = min(cu_items[cu], key=cu_items[cu].get)
        return cu_last[cu] + max(cu_items[cu][i], 0.0) / rate(k), i

    # simple loop: global time stepping over events = {block finishes, dep-ready times}
    resident = [0] * n_cu
    by_id = {}
    for k, it in enumerate(items):
        it.id = k
        by_id[k] = it
    pending_start = []   # (ready_time, id, cu) resident blocks waiting for deps
    finished = 0
    n = len(items)
    now = 0.0
    gen = [0] * n_cu

    def place(now):
        nonlocal nxt
        while nxt < n:
            cu = min(range(n_cu), key=lambda c: resident[c])
            if resident[cu] >= slots_per_cu:
                return
            it = items[nxt]
            nxt += 1
            resident[cu] += 1
            # deps all have tickets < this one, so they are at least resident; their end may be unknown yet
            pending_start.append((it, cu))

    def try_start(now):
        still = []
        changed = False
        for it, cu in pending_start:
            if all(d.end is not None and d.end + HOP <= now + 1e-9 for d in it.deps):
                advance(cu, now)
                cu_items[cu][it.id] = it.work
                it.start = now
                gen[cu] += 1
                changed = True
            else:
                still.append((it, cu))
        pending_start[:] = still
        return changed

    place(0.0)
    try_start(0.0)
    while finished < n:
        # next event: earliest finish over CUs, or earliest dep-ready of a pending block
        best = None
        for cu in range(n_cu):
            nf = next_finish(cu)
            if nf and (best is None or nf[0] < best[0]):
                best = (nf[0], cu, nf[1])
        tready = None
        for it, cu in pending_start:
            if all(d.end is not None for d in it.deps):
                tr = max([d.end + HOP for d in it.deps] + [now])
                if tready is None or tr < tready:
                    tready = tr
        if best is None and tready is None:
            raise RuntimeError("deadlock in plan at t=%.1f (ticket %d)" % (now, nxt))
        if tready is not None and (best is None or tready <= best[0]):
            now = tready
            try_start(now)
            continue
        now, cu, iid = best
        advance(cu, now)
        del cu_items[cu][iid]
        by_id[iid].end = now
        resident[cu] -= 1
        finished += 1
        place(now)
        try_start(now)
    return now


def mk_plan(log_n, order="lag", lv=4, tail_q_max=1 << 13, tail_w_max=1 << 10, lag_s=1100, end_lv=None, end_frac_log=None,
            u_first=True):
    """Items of a 2^log_n-leaf build and a ticket order.
    Stage 1: blocks of 256 lanes x 2^lv leaves (the last 2^end_frac_log leaves with 2^end_lv-leaf subtrees when given).
    Level h above: U items (256 nodes) while the level has more than tail_q_max nodes, Q items (64 nodes) down to tail_w_max,
    W items (4 nodes) below."""
    N = 1 << log_n
    items = []
    node_owner = {}  # (h, chunk64) -> item that produces it
    # stage 1
    split = N - (1 << end_frac_log) if end_frac_log is not None else N
    s_items = []
    leaf = 0
    while leaf < N:
        l = lv if leaf < split else end_lv
        span = 256 << l
        it = Item("S%d" % l, l, leaf >> l, 256, ((1 << l) - 1) * T_HASH)
        s_items.append(it)
        for c in range(4):
            node_owner[(l, (it.j0 >> 6) + c)] = it
        leaf += span
    min_lv = min(lv, end_lv) if end_lv is not None else lv
    upper = defaultdict(list)
    for h in range(min_lv + 1, log_n + 1):
        n_nodes = N >> h
        j = 0
        while j < n_nodes:
            # nodes whose children exist at level h-1 as S roots of the same level or as upper nodes
            first_leaf = j << h
            child_lv_is_s = None
            # a node at level h is produced by an S item if h <= that region's lv
            region_lv = lv if first_leaf < split else end_lv
            if h <= region_lv:
                # skip the whole region's span at this level
                if first_leaf < split:
                    j = split >> h
                else:
                    j = n_nodes
                continue
            if n_nodes - 0 > tail_q_max:
                kind, cnt, work = "U", 256, T_HASH
            elif n_nodes > tail_w_max:
                kind, cnt, work = "Q", 64, T_QUAD
            else:
                kind, cnt, work = "W", 4, T_WAVE
            cnt = min(cnt, n_nodes - j)
            it = Item(kind, h, j, cnt, work)
            # deps: producers of children chunks [2j, 2j + 2cnt) at level h-1
            seen = set()
            for c in range((2 * j) >> 6, ((2 * j + 2 * cnt - 1) >> 6) + 1):
                d = node_owner.get((h - 1, c))
                if d is None:
                    raise RuntimeError("no producer for level %d chunk %d" % (h - 1, c))
                if id(d) not in seen:
                    seen.add(id(d))
                    it.deps.append(d)
            for c in range(j >> 6, ((j + cnt - 1) >> 6) + 1):
                # several small items may share a chunk: owner = list
                prev = node_owner.get((h, c))
                if prev is None:
                    node_owner[(h, c)] = it
                else:
                    # chain: represent a multi-producer chunk by a pseudo list
                    if not isinstance(prev, list):
                        prev = [prev]
                    prev.append(it)
                    node_owner[(h, c)] = prev
            upper[h].append(it)
            j += cnt
    # flatten multi-producer owners in deps
    for h in upper:
        for it in upper[h]:
            flat = []
            for d in it.deps:
                if isinstance(d, list):
                    flat.extend(d)
                else:
                    flat.append(d)
            it.deps = flat
    all_upper = [it for h in sorted(upper) for it in upper[h]]
    if order == "levels":      # stage 1, then level by level (today's launches, as one grid)
        return s_items + all_upper
    if order == "sim":         # greedy list schedule by a coarse model: upper items as soon as their inputs are `lag` old
        return list_schedule(s_items, all_upper, u_first)
    if order == "lag":
        # emit S items in order; an upper item is emitted once all its deps were emitted >= lag tickets ago (S deps: lag_s, others: lag_u)
        out = []
        pos = {}
        pend = list(all_upper)
        si = 0
        lag_u = 140
        while si < len(s_items) or pend:
            if si < len(s_items):
                it = s_items[si]
                si += 1
                pos[id(it)] = len(out)
                out.append(it)
            progressed = True
            while progressed:
                progressed = False
                rest = []
                for it in pend:
                    ok = True
                    for d in it.deps:
                        p = pos.get(id(d))
                        if p is None:
                            ok = False
                            break
                        need = lag_s if d.kind[0] == "S" else lag_u
                        if si < len(s_items) and len(out) - p < need:
                            ok = False
                            break
                    if ok:
                        pos[id(it)] = len(out)
                        out.append(it)
                        progressed = True
                    else:
                        rest.append(it)
                pend = rest
                if si < len(s_items):
                    break
        return out
    raise ValueError(order)


def list_schedule(s_items, upper, u_first):
    """Coarse greedy: 1024 slots, fixed durations at full occupancy; ready upper items go first (highest level first)."""
    n_slots = 1024
    dur = lambda it: it.work * 2.0  # at 4 per CU
    free = [(0.0, k) for k in range(n_slots)]
    heapq.heapify(free)
    end = {}
    out = []
    si = 0
    pend = list(upper)
    waiting = []  # heap of (ready_time, -h, seq, item)
    seq = 0
    unresolved = {id(it): len(it.deps) for it in upper}
    children = defaultdict(list)
    for it in upper:
        for d in it.deps:
            children[id(d)].append(it)
    ready_t = {}

    def finish(it, t):
        nonlocal seq
        end[id(it)] = t
        for c in children[id(it)]:
            unresolved[id(c)] -= 1
            if unresolved[id(c)] == 0:
                rt = max(end[id(d)] for d in c.deps) + HOP
                heapq.heappush(waiting, (rt, -c.h, seq, c))
                seq += 1

    n_total = len(s_items) + len(upper)
    while len(out) < n_total:
        t, slot = heapq.heappop(free)
        it = None
        if waiting and waiting[0][0] <= t and (u_first or si >= len(s_items)):
            it = heapq.heappop(waiting)[3]
        elif si < len(s_items):
            it = s_items[si]
            si += 1
        elif waiting:
            rt, _, _, it = heapq.heappop(waiting)
            t = max(t, rt)
        else:
            raise RuntimeError("list_schedule: nothing to run")
        out.append(it)
        e = t + dur(it)
        finish(it, e)
        heapq.heappush(free, (e, slot))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=24)
    ap.add_argument("--lv", type=int, default=4)
    ap.add_argument("--order", default="sim")
    ap.add_argument("--tail-q-max-log", type=int, default=13)
    ap.add_argument("--tail-w-max-log", type=int, default=10)
    ap.add_argument("--end-lv", type=int, default=None)
    ap.add_argument("--end-frac-log", type=int, default=None)
    ap.add_argument("--lag-s", type=int, default=1100)
    ap.add_argument("--s-first", action="store_true")
    a = ap.parse_args()
    plan = mk_plan(a.log_n, a.order, a.lv, 1 << a.tail_q_max_log, 1 << a.tail_w_max_log, a.lag_s, a.end_lv, a.end_frac_log,
                   not a.s_first)
    ms = simulate(plan)
    n_s = sum(1 for it in plan if it.kind[0] == "S")
    last_s = max(it.end for it in plan if it.kind[0] == "S")
    work = sum(it.work for it in plan)
    print("items %d (stage-1 %d)  makespan %.1f us  last stage-1 end %.1f us  tail %.1f us  ideal %.1f us" % (
        len(plan), n_s, ms, last_s, ms - last_s, work * 2 / 1024))
    wait = sum((it.start - max([d.end for d in it.deps] + [0])) for it in plan if it.deps)
    print("kinds:", {k: sum(1 for it in plan if it.kind == k) for k in sorted(set(it.kind for it in plan))})


if __name__ == "__main__":
    main()


# ---------------------------------------------------------------- explicit "rounds" orders (python -c "import plan_sim; ...")
def items_for(log_n, regions, tail_q_max, tail_w_max, kind_fn=None):
    """regions: list of (n_leaves, lv) covering 2^log_n leaves left to right (each n_leaves a multiple of 256 << lv)."""
    N = 1 << log_n
    owner = {}
    s_items = []
    leaf = 0
    lv_at = []  # (leaf_lo, leaf_hi, lv)
    for n_leaves, l in regions:
        lv_at.append((leaf, leaf + n_leaves, l))
        for b in range(n_leaves // (256 << l)):
            it = Item("S%d" % l, l, (leaf >> l) + 256 * b, 256, ((1 << l) - 1) * T_HASH)
            s_items.append(it)
            for c in range(4):
                owner[(l, (it.j0 >> 6) + c)] = [it]
        leaf += n_leaves
    assert leaf == N

    def region_lv(first_leaf):
        for lo, hi, l in lv_at:
            if lo <= first_leaf < hi:
                return l
        raise KeyError

    upper = []
    min_lv = min(l for _, l in regions)
    for h in range(min_lv + 1, log_n + 1):
        n_nodes = N >> h
        j = 0
        while j < n_nodes:
            if h <= region_lv(j << h):
                j += 1 if False else max(1, 64)  # S-owned; skip a chunk (regions are chunk-aligned at every level <= lv)
                continue
            if kind_fn:
                kind = kind_fn(h, j, n_nodes)
            else:
                kind = "U" if n_nodes > tail_q_max else ("Q" if n_nodes > tail_w_max else "W")
            cnt, work = {"U": (256, T_HASH), "Q": (64, T_QUAD), "W": (4, T_WAVE)}[kind]
            # an item never crosses into an S-owned stretch
            cnt = min(cnt, n_nodes - j)
            k = 0
            while k < cnt and h > region_lv((j + k) << h):
                k += 64 if cnt >= 64 else cnt
            cnt = min(cnt, k)
            it = Item(kind, h, j, cnt, work)
            seen = set()
            for c in range((2 * j) >> 6, ((2 * j + 2 * cnt - 1) >> 6) + 1):
                for d in owner[(h - 1, c)]:
                    if id(d) not in seen:
                        seen.add(id(d))
                        it.deps.append(d)
            for c in range(j >> 6, ((j + cnt - 1) >> 6) + 1):
                owner.setdefault((h, c), []).append(it)
            upper.append(it)
            j += cnt
    return s_items, upper


def order_rounds(s_items, upper, round_sizes):
    """round r: round_sizes[r] stage-1 items, then every upper item whose stage-1 inputs all lie in rounds < r (level order)."""
    out = []
    emitted = set()
    s_round = {}
    k = 0
    for r, n in enumerate(round_sizes):
        for it in s_items[k:k + n]:
            s_round[id(it)] = r
        k += n
    assert k == len(s_items), (k, len(s_items))
    # round in which an upper item's inputs are complete = max over deps
    memo = {}

    def rnd(it):
        if it.kind[0] == "S":
            return s_round[id(it)]
        v = memo.get(id(it))
        if v is None:
            v = max(rnd(d) for d in it.deps)
            memo[id(it)] = v
        return v
    by_round = defaultdict(list)
    for it in upper:
        by_round[rnd(it)].append(it)
    k = 0
    for r, n in enumerate(round_sizes):
        out.extend(s_items[k:k + n])
        k += n
        if r >= 1:
            out.extend(by_round[r - 1])
    out.extend(by_round[len(round_sizes) - 1])
    return out


def report(plan, label=""):
    ms = simulate(plan)
    last_s = max(it.end for it in plan if it.kind[0] == "S")
    work = sum(it.work for it in plan)
    print("%-40s items %6d  makespan %7.1f us  last stage-1 end %7.1f  tail %6.1f  ideal %7.1f" % (
        label, len(plan), ms, last_s, ms - last_s, work * 2 / 1024))
    return ms
