#!/usr/bin/env python3
"""CPU: a time-stepped model of the persistent dataflow Cholesky (csrc/persist.hip) -- the chain workgroup, the dedicated
near-tile owners and the far-tile workers with their static ownership, pick rule and progress words -- used to price changes
of the ownership / pick policy BEFORE they are built.  Costs are the measured ones of
profiles/r5b_persist_probe.txt / r6a_persist_probe_baseline.txt (microseconds):

    chain: factor 15.9, solve 12.8, update 11.3, dcnt published +3 after the factor, row j+1 published +4 into the update
    worker: K = 128 c columns in one pass 7 + 13.5 c, solve of a finished tile against L_kk 17, hand-over / poll latency ~1.5

    python tools/persist_sim.py [nt ...]        prints the modelled time of the shipped policy and of the variants

State of the model (end of round 6): it prices the far half of the reserve rule (it predicted -3 %, measured -1.5 %) and the far
order; it does NOT model the near owners' reserve (measured -7 % at nt = 32: the model lets a near owner serve its most urgent
tile first, which the kernel only does since that rule), the second sub-diagonal in 64-row halves on two CUs ("near tasks x0.55"
is its stand-in: the model says -0.1 %, measured -4.4 %, because in the kernel the halves also take a third of the near owners'
load away) or the slower factorisation of the first steps (18-21 us against 15.2 while the whole chip is busy).
"""
import sys

F, S, U = 15.9, 12.8, 11.3
DCNT_LAG, ROW_LAG = 3.0, 4.0
POLL = 1.5


def hdiv_default(nt):
    return 2 if nt <= 20 else (4 if nt <= 27 else 6)


class Sim(object):
    def __init__(self, nt, cus=255, hdiv=None, kcap=2, pass_fix=7.0, pass_col=13.5, trsm=17.0, near_scale=1.0, far_scale=1.0,
                 order="row", chain=(F, S, U), lazy=0, reserve=None):
        self.nt, self.kcap = nt, kcap
        self.pass_fix, self.pass_col, self.trsm = pass_fix, pass_col, trsm
        self.near_scale, self.far_scale, self.order, self.lazy = near_scale, far_scale, order, lazy
        self.reserve = reserve
        self.F, self.S, self.U = chain
        D = 2
        ntl = nt * (nt + 1) // 2
        grid = min(cus, ntl + 1)
        nw = grid - 1
        near = []
        for i in range(nt):
            for k in range(max(0, i - D), i + 1):
                near.append((i, k))
        far = [(i, k) for i in range(D + 1, nt) for k in range(0, i - D)]
        # enumeration order of persist.hip: near by rows; far: f = r (r + 1) / 2 + k, i = r + D + 1 (row-major)
        hd = hdiv if hdiv else hdiv_default(nt)
        H = max(1, nw // hd)
        H = min(H, len(near))
        if not far:
            H = min(nw, len(near))
        W = nw - H
        self.own = [[] for _ in range(nw)]
        for e, t in enumerate(near):
            self.own[e % H].append(t)
        for f, t in enumerate(far):
            self.own[H + f % W].append(t)
        self.H = H
        self.nw = nw

    def run(self):
        nt = self.nt
        INF = 1e18
        cnt = [0] * nt          # cnt[i]: L(i, 0..cnt-1) final   (time-stamped publication below)
        cnt_t = [[INF] * (nt + 1) for _ in range(nt)]     # cnt_t[i][c]: when cnt[i] >= c became visible
        for i in range(nt):
            cnt_t[i][0] = 0.0
        dcnt_t = [INF] * (nt + 1)
        dcnt_t[0] = 0.0
        sub_t = [INF] * nt      # tile (i, i-1) handed to the chain
        dia_t = [INF] * nt
        pre_t = [INF] * nt
        dia_t[0] = 0.0
        # worker state
        prog = {}               # tile -> columns applied
        state = {}              # tile -> 0 working, 1 waiting for L_kk, 2 done, 3 (near (i,i-2)) pending last column of (i,i-1)
        for w in range(self.nw):
            for t in self.own[w]:
                prog[t] = 0
                state[t] = 0
        if (0, 0) in state:
            state[(0, 0)] = 2
        free_at = [0.0] * self.nw
        # chain as a generator of events evaluated lazily: we step time
        t = 0.0
        dt = 0.5
        chain_j, chain_phase, chain_t = 0, "factor", 0.0     # phase ends at chain_t
        chain_t = self.F
        done = False
        pending = []            # (time, fn) publications
        total_wait = 0.0

        def vis(tt, x):
            return x <= tt

        end_time = None
        while not done and t < 1e5:
            # --- publications whose time has come are already encoded as timestamps (cnt_t, dcnt_t, ...)
            # --- chain
            while True:
                if chain_phase == "factor" and t >= chain_t:
                    j = chain_j
                    dcnt_t[j + 1] = min(dcnt_t[j + 1], chain_t + DCNT_LAG)
                    if j == nt - 1:
                        done = True
                        end_time = chain_t + 2.0
                        break
                    chain_phase = "wait_sub"
                elif chain_phase == "wait_sub":
                    j = chain_j
                    if sub_t[j + 1] <= t:
                        chain_phase, chain_t = "solve", max(t, chain_t) + self.S
                    else:
                        total_wait += dt
                        break
                elif chain_phase == "solve" and t >= chain_t:
                    j = chain_j
                    cnt_t[j + 1][j + 1] = chain_t + ROW_LAG
                    chain_phase = "wait_dia"
                elif chain_phase == "wait_dia":
                    j = chain_j
                    if dia_t[j + 1] <= t:
                        chain_phase, chain_t = "update", max(t, chain_t) + self.U
                    else:
                        total_wait += dt
                        break
                elif chain_phase == "update" and t >= chain_t:
                    chain_j += 1
                    chain_phase, chain_t = "factor", chain_t + 1.0 + self.F
                else:
                    break
            if done:
                break
            # --- workers
            for w in range(self.nw):
                if free_at[w] > t:
                    continue
                tiles = self.own[w]
                if self.order == "col" and w >= self.H:
                    tiles = sorted(tiles, key=lambda x: (x[1], x[0]))
                pick = None
                hot = False
                if self.reserve is not None and w >= self.H:
                    # visible dcnt
                    dc = 0
                    while dc < nt and dcnt_t[dc + 1] <= t:
                        dc += 1
                    hot = any(state[x] != 2 and x[1] <= dc + self.reserve for x in tiles)
                for tl in tiles:
                    if hot and tl[1] > dc + self.reserve:
                        continue
                    st = state[tl]
                    if st == 2:
                        continue
                    i, k = tl
                    p = prog[tl]
                    if st == 3:
                        # last column (i-2) of tile (i, i-1): needs row i-1 final through column i-2 and PRE(i)
                        if cnt_t[i - 1][i - 1] <= t and pre_t[i] <= t:
                            pick = (tl, "fin")
                            break
                        continue
                    sub = (i == k + 1)
                    limit = i - 1 if i == k else ((k - 1) if (sub and i >= 2) else k)
                    if st == 1:
                        if dcnt_t[k + 1] <= t:
                            pick = (tl, "trsm")
                            break
                        continue
                    # columns available: min(cnt[i], cnt[k]) visible now
                    jm = 0
                    ci, ck = cnt_t[i], cnt_t[k]
                    while jm < limit and ci[jm + 1] <= t and (i == k or ck[jm + 1] <= t):
                        jm += 1
                    if jm > p or p >= limit:
                        if self.lazy and w >= self.H and p < limit and jm < limit and (jm - p) < self.lazy and k - chain_j > 3:
                            continue                         # far tile, deadline far away, only a shallow pass available: leave it
                        q = min(jm, p + self.kcap) if jm > p else p
                        pick = (tl, "pass", p, q, limit)
                        break
                if pick is None:
                    free_at[w] = t + POLL
                    continue
                tl = pick[0]
                i, k = tl
                near = w < self.H
                sc = self.near_scale if near else self.far_scale
                if pick[1] == "pass":
                    _, _, p, q, limit = pick
                    dur = (self.pass_fix + self.pass_col * (q - p)) * sc if q > p else 0.5
                    fin = t + dur
                    prog[tl] = q
                    if q >= limit:
                        if i == k:
                            dia_t[i] = fin + 0.5
                            state[tl] = 2
                        elif i == k + 1 and i >= 2:
                            pre_t[i] = fin + 0.5
                            state[tl] = 2
                        elif i == k + 1:          # tile (1, 0): handed over directly
                            sub_t[i] = fin + 0.5
                            state[tl] = 2
                        else:
                            state[tl] = 1
                    free_at[w] = fin
                elif pick[1] == "trsm":
                    fin = t + self.trsm * sc
                    cnt_t[i][k + 1] = fin + 0.5
                    state[tl] = 3 if i == k + 2 else 2
                    free_at[w] = fin
                else:
                    fin = t + (self.pass_fix + self.pass_col) * sc
                    sub_t[i] = fin + 0.5
                    state[tl] = 2
                    free_at[w] = fin
            t += dt
        return end_time, total_wait


def main():
    sizes = [int(v) for v in sys.argv[1:]] or [16, 32]
    variants = [
        ("shipped policy", {}),
        ("near tiles in 64-row halves on two CUs (near tasks x0.55)", dict(near_scale=0.55)),
        ("far workers with two tile pipelines (far passes x0.7)", dict(far_scale=0.7)),
        ("both", dict(near_scale=0.55, far_scale=0.7)),
        ("far tiles picked by column (deadline) instead of by row", dict(order="col")),
        ("far tiles: lazy deep passes (>= 4 columns unless urgent), kcap 8", dict(lazy=4, kcap=8)),
        ("kcap 4", dict(kcap=4)),
        ("by column + a worker with a tile of column <= dcnt + 0 works on nothing else", dict(order="col", reserve=0)),
        ("by column + ... column <= dcnt + 1", dict(order="col", reserve=1)),
        ("by column + ... column <= dcnt + 2", dict(order="col", reserve=2)),
        ("by column, kcap 1", dict(order="col", kcap=1)),
        ("by column + reserve 1, kcap 1", dict(order="col", reserve=1, kcap=1)),
        ("half of the workers near", dict(hdiv=2)),
        ("chain 30 us per step (solve pipelined under the factor) + near halves", dict(chain=(F, 2.0, U), near_scale=0.55)),
        ("chain 30 us + near halves + two pipelines", dict(chain=(F, 2.0, U), near_scale=0.55, far_scale=0.7)),
    ]
    for nt in sizes:
        print("nt = %d (N = %d)" % (nt, 128 * nt))
        for name, kw in variants:
            s = Sim(nt, **kw)
            end, wait = s.run()
            print("   %-72s %7.1f us  (%.1f us per step, chain waited %.0f us; %d near owners, %d far workers)" % (
                name, end, end / nt, wait, s.H, s.nw - s.H))


if __name__ == "__main__":
    main()
