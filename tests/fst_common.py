"""Synthetic back-off bigram LM as arcs, shared by the golden generator and the tests.
Labels: word w (1..V-1) has ilabel w+1 (the decoder adds 1, beam_transducer.py:139); back-off
label = V+1, one disambiguation label = V+2."""
import numpy as np


def bigram_arcs(V, seed=123, n_succ=6):
    rng = np.random.default_rng(seed)
    backoff_id, disambig = V + 1, V + 2
    n_states = 1 + V               # 0 = unigram/back-off state (also start), 1+w = history w
    arcs, finals = [], {0: float(rng.uniform(1, 3))}
    for w in range(1, V):
        arcs.append((0, w + 1, float(rng.uniform(2, 7)), 1 + w))          # unigram
    for w in range(1, V):
        h = 1 + w
        for v in rng.choice(np.arange(1, V), size=n_succ, replace=False):
            arcs.append((h, int(v) + 1, float(rng.uniform(0.5, 4)), 1 + int(v)))   # bigram
        arcs.append((h, backoff_id, float(rng.uniform(0.2, 2)), 0))       # back-off
        if w % 3 == 0:
            finals[h] = float(rng.uniform(0.5, 2))
        if w % 5 == 0:
            arcs.append((h, disambig, float(rng.uniform(0.1, 1)), 1 + (w % (V - 1)) + 0))
    # OpenFST's TropicalWeight is a float32: keep every weight float32-representable
    arcs = [(a, b, float(np.float32(c)), d) for a, b, c, d in arcs]
    finals = {k: float(np.float32(v)) for k, v in finals.items()}
    params = dict(max_num_arcs=V + 4, max_id=V + 3, backoff_id=backoff_id, disambig_ids=[disambig])
    return n_states, arcs, finals, params


class DuckFst(object):
    """What the REFERENCE SortedMatcher asks of an OpenFST object (`arcs(state)` -> iterator with seek/done/value,
    `final(state).value`; decoder/sorted_matcher.py:28-47,97) over a plain arc list -- golden generators only."""

    def __init__(self, n_states, arcs, finals):
        from types import SimpleNamespace
        self._ns = SimpleNamespace
        self.by = [[] for _ in range(n_states)]
        for s, i, w, d in arcs:
            self.by[s].append(SimpleNamespace(ilabel=i, weight=SimpleNamespace(value=w), nextstate=d))
        for lst in self.by:
            lst.sort(key=lambda a: a.ilabel)
        self.finals = finals

    def arcs(self, state):
        lst = self.by[state]

        class It(object):
            pos = 0
            def seek(s, p): s.pos = p
            def done(s): return s.pos >= len(lst)
            def value(s): return lst[s.pos]
        return It()

    def final(self, state):
        return self._ns(value=self.finals.get(state, float("inf")))
