"""Host-side plan of a LAS scoring pass (pika_amd.model.las.scoring_plan: rows, spans, prefix sharing, forks, the (step, row)
pairs) against the per-hypothesis form it replaced: dict tries and Python loops, spelled out below."""
import numpy as np
import pytest


def _plan_loop_form(flat, own_h, sos, eos, pad, share):
    n = len(flat)
    ntok = np.array([len(h) + 1 for h in flat], dtype=np.int64)
    L = int(ntok.max())
    rep = None
    act = None
    if share:
        rep, act = [None] * n, np.full(n, L + 1, np.int64)
        roots = {}
        for i, h in enumerate(flat):                    # a trie node: {token: child, -1: the entry that created it}
            b = int(own_h[i])
            node = roots.get(b)
            if node is None:
                node = roots[b] = {-1: i}
            r = [node[-1]]
            a = 0 if node[-1] == i else None
            for t, tokv in enumerate(h, 1):
                nxt = node.get(tokv)
                if nxt is None:
                    nxt = node[tokv] = {-1: i}
                node = nxt
                w = node[-1]
                r.append(w)
                if a is None and w == i:
                    a = t
            rep[i] = r
            if a is not None:
                act[i] = a
    perm = np.argsort(-ntok, kind="stable")
    first = act[perm] if share else np.zeros(n, np.int64)
    end = ntok[perm]
    col_of = np.empty(n, np.int64)
    col_of[perm] = np.arange(n)
    tok = np.full((L, n), pad, dtype=np.int64)
    for col, i in enumerate(perm):
        h = flat[i]
        tok[0, col] = sos
        tok[1:len(h) + 1, col] = h
    forks = None
    if share:
        f_t, f_dst, f_src = [], [], []
        for i in range(n):
            if 0 < act[i] <= L:
                f_t.append(int(act[i]))
                f_dst.append(int(col_of[i]))
                f_src.append(int(col_of[rep[i][act[i] - 1]]))
        order = np.argsort(np.asarray(f_t, np.int64), kind="stable")
        f_t = np.asarray(f_t, np.int64)[order]
        fork_off = np.searchsorted(f_t, np.arange(L + 1)).astype(np.int32)
        forks = (fork_off, np.asarray(f_dst, np.int32)[order], np.asarray(f_src, np.int32)[order])
    tt = np.concatenate([np.arange(k) for k in ntok])
    ii = np.repeat(np.arange(n), ntok)
    rr = col_of[np.concatenate([np.asarray(r, np.int64) for r in rep])] if share else col_of[ii]
    tgt = np.concatenate([np.asarray(list(h) + [eos], np.int64) for h in flat])
    return {"L": L, "ntok": ntok, "perm": perm, "tok": tok, "first": first, "end": end, "forks": forks,
            "row_steps": int(np.maximum(end - first, 0).sum()), "pair_step": tt, "pair_row": rr, "pair_target": tgt}


def _nbest_lists(seed, B, nb, U, V):
    """n-best-like lists: per utterance a transcript and entries that differ from it by a few edits; duplicates, entries
    that are prefixes of earlier ones, and an empty hypothesis are in."""
    r = np.random.default_rng(seed)
    flat, own = [], []
    for b in range(B):
        base = r.integers(1, V, max(1, U + int(r.integers(-3, 4)))).tolist()
        for j in range(nb):
            h = list(base)
            for _ in range(int(r.integers(0, 3))):
                k, pos = int(r.integers(0, 3)), int(r.integers(0, max(len(h), 1)))
                if k == 0 and len(h) > 1:
                    del h[pos]
                elif k == 1 and h:
                    h[pos] = int(r.integers(1, V))
                else:
                    h.insert(pos, int(r.integers(1, V)))
            if j == 3:
                h = list(flat[-1])                      # an exact duplicate of the entry before
            if j == 4:
                h = list(base[:max(len(base) // 2, 0)])     # a prefix of an earlier entry
            flat.append(h)
            own.append(b)
    if len(flat) > 1:
        flat[1] = []
    return flat, np.asarray(own, np.int64)


@pytest.mark.parametrize("share", [True, False])
@pytest.mark.parametrize("B,nb,U,V", [(1, 1, 3, 10), (3, 6, 7, 12), (8, 16, 25, 300), (2, 5, 1, 4)])
def test_scoring_plan_equals_the_per_hypothesis_form(share, B, nb, U, V):
    from pika_amd.model.las import scoring_plan
    for seed in range(4):
        flat, own = _nbest_lists(seed, B, nb, U, V)
        want = _plan_loop_form(flat, own, V + 1, V + 2, V + 3, share)
        got = scoring_plan(flat, own, V + 1, V + 2, V + 3, share)
        for k in ("L", "row_steps"):
            assert want[k] == got[k], k
        for k in ("ntok", "perm", "tok", "first", "end", "pair_step", "pair_row", "pair_target"):
            assert got[k].dtype == want[k].dtype and np.array_equal(want[k], got[k]), k
        if share:
            for w, g in zip(want["forks"], got["forks"]):
                assert g.dtype == w.dtype and np.array_equal(w, g)
        else:
            assert got["forks"] is None


@pytest.mark.parametrize("share", [True, False])
def test_distinct_step_row_cells_are_what_np_unique_finds(share):
    """Net._score_stages takes the distinct (step, row) pairs of a plan -- what the tail projects onto the vocabulary -- from
    the rows' spans instead of sorting the pairs: same keys in the same order, same inverse."""
    from pika_amd.model.las import scoring_plan
    for seed in range(6):
        flat, own = _nbest_lists(seed, 5, 7, 9, 20)
        n = len(flat)
        p = scoring_plan(flat, own, 21, 22, 23, share)
        key, inv = np.unique(p["pair_step"] * n + p["pair_row"], return_inverse=True)
        t_col = np.arange(p["L"])[:, None]
        cells = (t_col >= p["first"][None, :]) & (t_col < p["end"][None, :])
        assert np.array_equal(key, np.flatnonzero(cells.ravel()))
        assert np.array_equal(inv.reshape(-1), (np.cumsum(cells.ravel()) - 1)[p["pair_step"] * n + p["pair_row"]])
