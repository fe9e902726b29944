"""The symbolic phase of the reduced-system Cholesky, executed on the CPU.

The library's tile schedule (csrc/cholesky.hip::build_chol_plan: symbolic fill at column-pair granularity, TRSM row lists,
thin / look-ahead / bulk / cross-part update lists, backward-solve lists, elimination-tree parts) is a set of index lists
that do not depend on the tile size.  Here the real library builds them for real graphs -- in a child process under
tools/hipstub, host code only -- and a numpy interpreter executes them, step for step what the kernels do, on a random
SPD matrix with exactly the tile pattern the graph produces (8x8 tiles instead of 128x128):

    panel(k): L_kk = chol(A_kk); L_Ik = A_Ik L_kk^-T for I in the TRSM rows (the rhs row included: forward solve for free)
    thin update (second column of a pair), then per pair the K = 2 tiles updates  A_IJ -= L_I,k:k+2 L_J,k:k+2^T
    backward solve by block rows with the stored column lists

The result must equal a dense Cholesky / solve: every tile the factor fills in is in the schedule (no missing fill, no
missing update, also across nested-dissection parts), and nothing outside the stored tiles is touched.
"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools import host_profile as HP  # noqa: E402

_CHILD = r'''
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from tools import host_profile as HP
from gtsam_amd import lib as L
problem, _ = HP.problem_for(%(workload)r)
g = L.DeviceGraph(problem)
pl = g.cholesky_plan()
out = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in pl.items()}
out["dense_fraction_flops"] = g.cholesky_flops()
df = g.df_plan()
out["df"] = {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in df.items()}
print("RESULT " + json.dumps(out))
'''


def _plan(workload, nd_depth=0):
    env = {} if nd_depth is None else {"GTG_ND_DEPTH": str(nd_depth)}      # None: the library's own choice; 0: nested dissection off
    d = HP.run_snippet(_CHILD % {"root": ROOT, "workload": workload}, env_extra=env)
    for k in ("rows", "pairs", "bcols", "pair_part", "part_parent"):
        d[k] = np.array(d[k], np.int64)
    for k in ("stored", "exch"):
        d[k] = np.array(d[k], np.int64).reshape(-1, 2)
    d["per_tile"] = np.array(d["per_tile"], np.int64).reshape(-1, 4); d["per_pair"] = np.array(d["per_pair"], np.int64).reshape(-1, 8)
    return d


def _execute(pl, t=8, seed=0):
    """Run the schedule on a random SPD matrix with the graph's tile pattern; returns the worst deviations from dense."""
    rng = np.random.default_rng(seed)
    nt = int(pl["nt"]); n = nt * t
    # the matrix before the factorisation: something in every tile of the exchange list (= structurally non-zero), rhs row incl.
    A = np.zeros((n, n)); g = np.zeros(n)
    for I, J in pl["exch"]:
        if I == nt:
            g[J * t:(J + 1) * t] = rng.normal(size=t)
        elif I != J:
            A[I * t:(I + 1) * t, J * t:(J + 1) * t] = rng.normal(size=(t, t)) * 0.3
    A = A + A.T
    A += np.diag(np.abs(A).sum(1) + 1.0 + rng.uniform(0, 1, n))                    # strictly diagonally dominant: SPD
    Ld = np.linalg.cholesky(A); yd = np.linalg.solve(Ld, g); xd = np.linalg.solve(Ld.T, yd)
    # tile store: only the stored tiles exist; touching anything else is a schedule error
    stored = {(int(I), int(J)) for I, J in pl["stored"]}
    tile = {}
    for (I, J) in stored:
        tile[(I, J)] = (g[J * t:(J + 1) * t][None, :].copy() if I == nt else A[I * t:(I + 1) * t, J * t:(J + 1) * t].copy())
    for I in range(nt):
        for J in range(I + 1):
            if (I, J) not in stored:
                assert not A[I * t:(I + 1) * t, J * t:(J + 1) * t].any(), "a non-zero tile of the graph is not stored"

    def T(I, J):
        if (I, J) not in tile:
            raise AssertionError(f"the schedule reads / writes tile ({I},{J}) which is not stored")
        return tile[(I, J)]
    rows, pairs = pl["rows"], pl["pairs"].reshape(-1, 2)
    pt, pp = pl["per_tile"], pl["per_pair"]

    def panel(k):
        Lkk = np.linalg.cholesky(np.tril(T(k, k)) + np.tril(T(k, k), -1).T)
        tile[(k, k)] = Lkk
        for I in rows[pt[k, 0]:pt[k, 0] + pt[k, 1]]:
            tile[(int(I), k)] = np.linalg.solve(Lkk, T(int(I), k).T).T

    def update(k, off, cnt, width):
        for I, J in pairs[off:off + cnt]:
            I, J = int(I), int(J)
            acc = T(I, J)
            for kk in range(k, k + width):
                acc = acc - T(I, kk) @ T(J, kk).T
            tile[(I, J)] = acc
    npairs = pp.shape[0]
    for p in range(npairs):
        k = 2 * p
        panel(k)
        if k + 1 < nt:
            update(k, pp[p, 0], pp[p, 1], 1)          # thin update of column k+1
            panel(k + 1)
            for off, cnt in ((pp[p, 2], pp[p, 3]), (pp[p, 4], pp[p, 5]), (pp[p, 6], pp[p, 7])):   # look-ahead, bulk, cross-part
                update(k, off, cnt, 2)
    # compare the factor: stored tiles equal dense L, dense L is zero elsewhere
    worst = 0.0
    for I in range(nt):
        for J in range(I + 1):
            blk = Ld[I * t:(I + 1) * t, J * t:(J + 1) * t]
            if (I, J) in stored:
                got = np.tril(tile[(I, J)]) if I == J else tile[(I, J)]
                worst = max(worst, float(np.abs(got - blk).max()))
            else:
                assert np.abs(blk).max() <= 1e-13, f"fill-in at tile ({I},{J}) is missing from the schedule"
    y = np.concatenate([T(nt, J)[0] for J in range(nt)])
    worst_y = float(np.abs(y - yd).max())
    # backward solve by block rows: x_k = L_kk^-T y_k, then y_c -= L_kc^T x_k for the stored column tiles c of row k
    y = y.copy(); x = np.zeros(n)
    bc = pl["bcols"]
    for k in range(nt - 1, -1, -1):
        xk = np.linalg.solve(tile[(k, k)].T, y[k * t:(k + 1) * t]); x[k * t:(k + 1) * t] = xk
        for c in bc[pt[k, 2]:pt[k, 2] + pt[k, 3]]:
            c = int(c)
            y[c * t:(c + 1) * t] -= T(k, c).T @ xk
    return worst, worst_y, float(np.abs(x - xd).max() / max(np.abs(xd).max(), 1e-300)), len(stored)


def _execute_df(pl, df, t=8, seed=0):
    """The dataflow schedule (csrc/chol_dataflow.hip) in numpy: the tasks in ticket order, exactly what a workgroup does with
    one -- PD(J): A_JJ -= sum_k L_Jk L_Jk^T over its k list (then the chain kernel applies L_J,J-1 and factors the tile); (I, J): R = A_IJ -
    sum_k L_Ik L_Jk^T, L_IJ = R L_JJ^-T; I == nt is the rhs row.  Every operand must be FINAL when it is read in this order
    (that is the no-deadlock argument: a task only waits for tasks before it), every k list must be complete (dense result).
    A long contraction comes in pieces (task fields r of R) that accumulate in place and are queued ahead of the tile's column."""
    rng = np.random.default_rng(seed)
    nt = int(pl["nt"]); n = nt * t
    assert int(df["nt"]) == nt
    A = np.zeros((n, n)); g = np.zeros(n)
    for I, J in pl["exch"]:
        if I == nt:
            g[J * t:(J + 1) * t] = rng.normal(size=t)
        elif I != J:
            A[I * t:(I + 1) * t, J * t:(J + 1) * t] = rng.normal(size=(t, t)) * 0.3
    A = A + A.T
    A += np.diag(np.abs(A).sum(1) + 1.0 + rng.uniform(0, 1, n))
    Ld = np.linalg.cholesky(A); yd = np.linalg.solve(Ld, g)
    tasks = np.array(df["tasks"], np.int64).reshape(-1, 6); klist = np.array(df["klist"], np.int64)
    # the order in which the block columns are queued (interleaves the independent parts of a nested-dissection ordering) and the
    # diagonal chains: every diagonal tile in exactly one chain workgroup's list, every list in queueing order -- a chain workgroup
    # that waits for its next tile then waits for the tile whose inputs are queued first (the no-deadlock argument of build_df_plan)
    seq = np.array(df["seq"], np.int64); coff = np.array(df["chain_off"], np.int64); ctiles = np.array(df["chain_tiles"], np.int64)
    assert sorted(seq.tolist()) == list(range(nt)) and sorted(ctiles.tolist()) == list(range(nt))
    pos = np.empty(nt, np.int64); pos[seq] = np.arange(nt)
    assert len(coff) >= 2 and coff[0] == 0 and coff[-1] == nt
    for w in range(len(coff) - 1):
        mine = ctiles[coff[w]:coff[w + 1]]
        assert (np.diff(pos[mine]) > 0).all(), f"chain workgroup {w} does not take its tiles in queueing order"
    owned = {(int(I), int(J)) for I, J in tasks[:, :2]}
    assert len(owned) == int((tasks[:, 4] == tasks[:, 5] - 1).sum()), "a tile must have exactly one finishing piece"
    stored_old = {(int(I), int(J)) for I, J in pl["stored"]}
    assert owned <= stored_old, "the dataflow schedule touches a tile the zeroing / backward lists do not know"
    for I in range(nt):
        for J in range(I + 1):
            if (I, J) not in owned:
                assert not A[I * t:(I + 1) * t, J * t:(J + 1) * t].any(), "a non-zero tile of the graph has no task"
    tile = {(I, J): (np.vstack([g[J * t:(J + 1) * t][None, :], np.zeros((t - 1, t))]) if I == nt else A[I * t:(I + 1) * t, J * t:(J + 1) * t].copy())
            for (I, J) in owned}
    final, diag_done, pieces_done, seen_k = set(), set(), {}, {}

    def L(I, k):
        assert (I, k) in final, f"tile ({I},{k}) is read before the task that produces it in ticket order"
        return tile[(I, k)]
    for I, J, off, cnt, r, R in tasks:
        I, J, off, cnt, r, R = (int(v) for v in (I, J, off, cnt, r, R))
        ks = [int(k) for k in klist[off:off + cnt]]
        assert [pos[k] for k in ks] == sorted(pos[k] for k in ks) and all(k < J for k in ks)     # steps in the order their operands appear
        # pieces of a tile's contraction accumulate in place, in order (piece r waits for piece r - 1), oldest steps first
        assert pieces_done.get((I, J), 0) == r and 0 <= r < R, f"piece {r} of tile ({I},{J}) out of order"
        assert not seen_k.get((I, J)) or (ks and pos[ks[0]] > pos[seen_k[(I, J)][-1]]) or not ks
        seen_k.setdefault((I, J), []).extend(ks)
        pieces_done[(I, J)] = r + 1
        acc = tile[(I, J)]
        for k in ks:
            acc = acc - L(I, k) @ L(J, k).T
        tile[(I, J)] = acc
        if r + 1 < R:
            continue
        if I == J:
            # k_df_chain: the update of block column J-1 is its own (streamed behind the substitution of tile (J, J-1), which
            # must therefore precede PD(J) in ticket order -- it is the first task of column J-1's group), then the factorisation
            # (round 4: with one chain also block column J-2's, applied in front of it; recognisable by its absence from PD's list)
            if (J, J - 2) in owned and J - 2 not in seen_k[(J, J)]:
                assert (J, J - 1) in owned and J - 1 not in seen_k[(J, J)]
                acc = acc - L(J, J - 2) @ L(J, J - 2).T
            if (J, J - 1) in owned:
                assert J - 1 not in seen_k[(J, J)], "PD(J) and the chain kernel would both apply block column J-1"
                acc = acc - L(J, J - 1) @ L(J, J - 1).T
            tile[(J, J)] = np.linalg.cholesky(np.tril(acc) + np.tril(acc, -1).T); diag_done.add(J)
        else:
            assert J in diag_done, "a tile's substitution precedes the accumulation of its diagonal tile in ticket order"
            tile[(I, J)] = np.linalg.solve(tile[(J, J)], acc.T).T
            final.add((I, J))
    worst = 0.0
    for I in range(nt):
        for J in range(I + 1):
            blk = Ld[I * t:(I + 1) * t, J * t:(J + 1) * t]
            if (I, J) in owned:
                worst = max(worst, float(np.abs(tile[(I, J)] - blk).max()))
            else:
                assert np.abs(blk).max() <= 1e-13, f"fill-in at tile ({I},{J}) has no task"
    y = np.concatenate([tile[(nt, J)][0] for J in range(nt)])
    return worst, float(np.abs(y - yd).max()), len(owned)


def _simulate_df(pl, df, n_bulk):
    """Deadlock freedom of the dataflow schedule, simulated with BLOCKING workers: n_bulk bulk workgroups take tickets in order and
    then hold their task until everything it reads exists (a piece waits for the piece before it, a contraction for its operand tiles,
    a substitution for the diagonal tile of its column); the chain workgroups walk their tile lists in order and hold a tile until its
    accumulated updates (PD) and the tile to its left are in.  Nobody ever lets go of a task.  The claim of build_df_plan: this
    finishes for ANY number of resident bulk workgroups, also one -- a task only waits for smaller tickets and for diagonal tiles
    whose own inputs have smaller tickets, and a chain workgroup waits for its tiles in queueing order."""
    nt = int(df["nt"])
    tasks = np.array(df["tasks"], np.int64).reshape(-1, 6); klist = np.array(df["klist"], np.int64)
    coff = np.array(df["chain_off"], np.int64); ctiles = np.array(df["chain_tiles"], np.int64)
    owned = {(int(I), int(J)) for I, J in tasks[:, :2]}
    final, pd_done, diag_done, pieces = set(), set(), set(), {}
    chain_at = [int(coff[w]) for w in range(len(coff) - 1)]
    holding = [None] * n_bulk           # ticket a bulk workgroup holds
    next_ticket = 0
    done_tasks = 0

    def ready(t):
        I, J, off, cnt, r, R = (int(v) for v in tasks[t])
        if pieces.get((I, J), 0) != r:
            return False
        for k in klist[off:off + cnt]:
            k = int(k)
            if (I, k) not in final and not (I == J and False):
                return False
            if (J, k) not in final:
                return False
        if r + 1 == R and I != J and J not in diag_done:
            return False
        return True

    def finish(t):
        I, J, off, cnt, r, R = (int(v) for v in tasks[t])
        pieces[(I, J)] = r + 1
        if r + 1 == R:
            if I == J:
                pd_done.add(J)
            else:
                final.add((I, J))
    progress = True
    while progress:
        progress = False
        for w in range(n_bulk):
            if holding[w] is None and next_ticket < len(tasks):
                holding[w] = next_ticket; next_ticket += 1; progress = True
            if holding[w] is not None and ready(holding[w]):
                finish(holding[w]); holding[w] = None; done_tasks += 1; progress = True
        for w in range(len(chain_at)):
            if chain_at[w] < coff[w + 1]:
                J = int(ctiles[chain_at[w]])
                if J in pd_done and ((J, J - 1) not in owned or (J, J - 1) in final) and ((J, J - 2) not in owned or (J, J - 2) in final):
                    diag_done.add(J); chain_at[w] += 1; progress = True
    assert done_tasks == len(tasks) and len(diag_done) == nt, \
        f"deadlock with {n_bulk} bulk workgroups: {done_tasks} of {len(tasks)} tasks, {len(diag_done)} of {nt} diagonal tiles"


@pytest.fixture(scope="module")
def stub():
    if not os.path.exists(os.path.join(ROOT, "gtsam_amd", "lib", "libgtsam_amd.so")):
        pytest.skip("libgtsam_amd.so not built")
    return HP.build_stub()


@pytest.mark.parametrize("workload,nd", [("bal:60:6000:7", 0), ("dubrovnik16", 0), ("sphere2500", 0), ("sphere2500", 2),
                                         ("sphere2500", 3), ("bal:300:20000:3", 0), ("bal:300:20000:3", 2),
                                         ("ladybug1723", 0), ("ladybug1723", 2), ("w20000", 0), ("w20000", 3),
                                         ("sphere2500", None), ("w20000", None), ("ladybug1723", None), ("dubrovnik16", None)])
def test_schedule_reproduces_a_dense_cholesky(stub, workload, nd):
    pl = _plan(workload, nd)
    if nd is None:     # the library's own choice: several chains for the sparse pose graphs, one band for the camera systems
        nd = 3 if workload in ("sphere2500", "w20000") else 0
    if nd:
        assert len(pl["part_parent"]) > 1, "nested dissection was requested but the plan has a single part"
        # children precede their parent, a pair of columns belongs to exactly one part
        assert all(pl["part_parent"][x] == -1 or pl["part_parent"][x] > x for x in range(len(pl["part_parent"])))
        assert len(pl["pair_part"]) == pl["per_pair"].shape[0] and (np.diff(pl["pair_part"]) >= 0).all()
    else:
        assert len(pl["part_parent"]) == 0 and not pl["per_pair"][:, 7].any()
    worst, worst_y, worst_x, n_stored = _execute(pl)
    assert worst <= 1e-10 and worst_y <= 1e-10 and worst_x <= 1e-10, (worst, worst_y, worst_x)
    # the dataflow schedule (the default, with or without nested dissection): same result, never more tiles
    assert pl["df"]["active"]
    w, wy, n_df = _execute_df(pl, pl["df"])
    assert w <= 1e-10 and wy <= 1e-10, (w, wy)
    assert n_df <= n_stored
    n_chain_wg = len(pl["df"]["chain_off"]) - 1
    nt = int(pl["nt"])
    if nd:    # independent subtrees run as several diagonal chains (<= 4 slots of two workgroups)
        assert 2 < n_chain_wg <= 32 and n_chain_wg % 2 == 0, n_chain_wg
        longest = max(np.diff(np.array(pl["df"]["chain_off"])[::2]))        # diagonal tiles of the longest slot
        assert longest < nt, (longest, nt)
    else:
        assert n_chain_wg == (2 if nt > 1 else 1)
    nt = int(pl["nt"])
    assert n_stored <= (nt + 1) * (nt + 2) // 2
    # the exchange list (structure before the factorisation) is a subset of the stored tiles
    stored = {(int(i), int(j)) for i, j in pl["stored"]}
    assert all((int(i), int(j)) in stored for i, j in pl["exch"])


def test_forced_dense_schedule(stub):
    d = HP.run_snippet(_CHILD % {"root": ROOT, "workload": "bal:60:6000:7"}, env_extra={"GTG_DENSE_PLAN": "1", "GTG_ORDERING": "natural"})
    pl = dict(d)
    for k in ("rows", "pairs", "bcols", "pair_part", "part_parent"):
        pl[k] = np.array(d[k], np.int64)
    for k in ("stored", "exch"):
        pl[k] = np.array(d[k], np.int64).reshape(-1, 2)
    pl["per_tile"] = np.array(d["per_tile"], np.int64).reshape(-1, 4); pl["per_pair"] = np.array(d["per_pair"], np.int64).reshape(-1, 8)
    nt = int(pl["nt"])
    assert len(pl["stored"]) == nt * (nt + 1) // 2 + nt          # every lower tile + the rhs row
    worst, worst_y, worst_x, _ = _execute(pl)
    assert max(worst, worst_y, worst_x) <= 1e-10
    w, wy, n_df = _execute_df(pl, pl["df"])
    assert max(w, wy) <= 1e-10 and n_df == nt * (nt + 1) // 2 + nt


@pytest.mark.parametrize("workload,nd", [("sphere2500", None), ("sphere2500", 2), ("w20000", None), ("ladybug1723", None), ("bal:300:20000:3", 2)])
def test_dataflow_schedule_cannot_deadlock(stub, workload, nd):
    """One, three and 248 resident bulk workgroups, blocking semantics (see _simulate_df): single-chain and multi-chain plans."""
    pl = _plan(workload, nd)
    for n_bulk in (1, 3, 248):
        _simulate_df(pl, pl["df"], n_bulk)


_TABLES_CHILD = r'''
import json, sys
import numpy as np
sys.path.insert(0, %(root)r)
from tools import host_profile as HP
from gtsam_amd import lib as L
problem, _ = HP.problem_for(%(workload)r)
g = L.DeviceGraph(problem)
pl = g.df_plan()
t12, s6, chain = g.df_device_tables()
sizes = np.zeros(8, np.int64)
L._check(g.lib.gtg_debug_plan_sizes(g.h, sizes.ctypes.data), "sizes")
out = dict(nt=pl["nt"], n_stored=int(sizes[4]), tasks=pl["tasks"].tolist(), klist=pl["klist"].tolist(), t12=t12.tolist(), s6=s6.tolist(), chain=chain.tolist(),
           flops=g.cholesky_flops(), executed=g.cholesky_flops_executed())
g.close()
print("RESULT " + json.dumps(out))
'''


@pytest.mark.parametrize("workload", ["bal:60:6000:7", "sphere2500"])
def test_dataflow_device_tables_resolve_the_plan(workload):
    """The dataflow plan as the kernels read it (gtg_debug_df_device_tables; built by the host loop here -- the dry-run runtime runs no
    kernel --, by k_df_resolve on a GPU, where tests/test_gpu_device_analysis.py compares the two word for word): every task carries its six
    plan words, the slots of its tile and of its diagonal tile and the tile's own sub-tile mask; every contraction step the slots of its two
    operand tiles, distinct stored tiles, and their masks; a step's operand masks are never empty (an empty operand tile would not be in
    the list), and the executed flop count is the stored-tile count minus what the masks skip."""
    d = HP.run_snippet(_TABLES_CHILD % {"root": ROOT, "workload": workload})
    nt, n_stored = d["nt"], d["n_stored"]
    tasks, t12, s6 = np.array(d["tasks"]), np.array(d["t12"]), np.array(d["s6"])
    assert t12.shape == (len(tasks), 12) and np.array_equal(t12[:, :6], tasks)
    assert s6.shape == (len(d["klist"]), 6)
    slot_of = {}
    for row in t12:
        I, J, q, qd = int(row[0]), int(row[1]), int(row[6]), int(row[7])
        assert 0 <= q < n_stored and 0 <= qd < n_stored
        assert slot_of.setdefault((I, J), q) == q and slot_of.setdefault((J, J), qd) == qd
        mask = (int(row[10]) & 0xFFFFFFFF) | ((int(row[11]) & 0xFFFFFFFF) << 32)
        assert mask != 0
        if I == J: assert mask == 2 ** 64 - 1
        if I == nt: assert mask == 0xFF
        G, first = int(row[8]), int(row[9])
        assert (G == 1 and first == -1) or (2 <= G <= 8 and first >= n_stored)
        for e in range(int(row[2]), int(row[2]) + int(row[3])):
            k = d["klist"][e]
            a, b = int(s6[e][0]), int(s6[e][1])
            assert 0 <= a < n_stored and 0 <= b < n_stored
            assert slot_of.setdefault((I, k), a) == a and slot_of.setdefault((J, k), b) == b
            ma = (int(s6[e][2]) & 0xFFFFFFFF) | ((int(s6[e][3]) & 0xFFFFFFFF) << 32)
            mb = (int(s6[e][4]) & 0xFFFFFFFF) | ((int(s6[e][5]) & 0xFFFFFFFF) << 32)
            assert ma != 0 and mb != 0
    assert len(set(slot_of.values())) == len(slot_of)                 # one slot per tile
    assert 0 < d["executed"] <= d["flops"]
