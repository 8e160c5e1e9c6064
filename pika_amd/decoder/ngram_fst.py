"""N-gram language-model FST for shallow fusion, without OpenFST/PyKaldi.

`NgramFst` holds an ilabel-sorted CSR arc table (what `kaldi.fstext.StdVectorFst` + arc iterators
provide to decoder/sorted_matcher.py) and `SortedMatcher` answers the three queries the beam search
makes (decoder/sorted_matcher.py:24-111): arc lookup by binary search, back-off chain scoring,
optionally through disambiguation arcs, and final cost.
"""
import math
import struct

import numpy as np


class NgramFst(object):
    def __init__(self, offsets, ilabel, weight, nextstate, final, start=0):
        self.offsets = np.asarray(offsets, np.int64)       # (S+1,)
        self.ilabel = np.asarray(ilabel, np.int32)
        self.weight = np.asarray(weight, np.float32)
        self.nextstate = np.asarray(nextstate, np.int32)
        self.final = np.asarray(final, np.float32)         # +inf = not final
        self.start = start
        for s in range(len(self.offsets) - 1):
            seg = self.ilabel[self.offsets[s]:self.offsets[s + 1]]
            if np.any(seg[1:] < seg[:-1]):
                raise ValueError("arcs of state %d are not ilabel-sorted" % s)

    @property
    def num_states(self):
        return len(self.offsets) - 1

    @classmethod
    def from_arcs(cls, num_states, arcs, finals, start=0):
        """arcs: iterable of (src, ilabel, weight, dst); finals: {state: weight}."""
        by_state = [[] for _ in range(num_states)]
        for s, i, w, d in arcs:
            by_state[s].append((i, w, d))
        off, il, wt, ns = [0], [], [], []
        for lst in by_state:
            lst.sort(key=lambda a: a[0])
            for i, w, d in lst:
                il.append(i); wt.append(w); ns.append(d)
            off.append(len(il))
        fin = np.full(num_states, np.inf, np.float32)
        for s, w in finals.items():
            fin[s] = w
        return cls(off, il, wt, ns, fin, start)

    @classmethod
    def read_text(cls, path):
        """AT&T text format: `src dst ilabel olabel [weight]` / `final [weight]` lines."""
        arcs, finals, n = [], {}, 0
        with open(path) as f:
            for line in f:
                p = line.split()
                if not p:
                    continue
                if len(p) <= 2:
                    finals[int(p[0])] = float(p[1]) if len(p) == 2 else 0.0
                    n = max(n, int(p[0]) + 1)
                else:
                    s, d, i = int(p[0]), int(p[1]), int(p[2])
                    arcs.append((s, i, float(p[4]) if len(p) > 4 else 0.0, d))
                    n = max(n, s + 1, d + 1)
        return cls.from_arcs(n, arcs, finals)

    isymbols = osymbols = None      # {key: symbol} of the embedded tables of a binary file that carries them

    @classmethod
    def read_binary(cls, path):
        """OpenFST binary `vector` / `standard` FST, with or without embedded symbol tables (what
        `fst.StdVectorFst.read` loads at decode_transducer.py:83; `fstcompile --isymbols=... --keep_isymbols` or an
        arpa2fst output written with its tables carries them: header flags bit 0 / bit 1).  A table on disk
        (OpenFST 1.6 / 1.7 `SymbolTableImpl::Write`): int32 magic 2125658996, string name, int64 available key, int64
        size, then `size` x (string symbol, int64 key); strings are int32 length + bytes.  The search only ever uses
        integer labels (sorted_matcher.py:24-111): the tables are parsed, kept as `isymbols` / `osymbols` and otherwise
        ignored.  Files written with --align (flag bit 2) are refused."""
        with open(path, "rb") as f:
            data = f.read()
        pos = [0]

        def take(fmt):
            v = struct.unpack_from("<" + fmt, data, pos[0])
            pos[0] += struct.calcsize("<" + fmt)
            return v[0]

        def string():
            n = take("i")
            s = data[pos[0]:pos[0] + n].decode()
            pos[0] += n
            return s
        if take("i") != 2125659606:
            raise ValueError("not an OpenFST binary file")
        fsttype, arctype = string(), string()
        take("i")
        flags = take("i")
        take("Q")
        start, nstates, _ = take("q"), take("q"), take("q")
        if fsttype != "vector" or arctype != "standard":
            raise NotImplementedError("need a vector / standard FST, got %s / %s" % (fsttype, arctype))
        if flags & 4:
            raise NotImplementedError("aligned OpenFST files (fstconvert --align) are not supported")

        def symbol_table():
            if take("i") != 2125658996:
                raise ValueError("corrupt embedded symbol table in %s" % path)
            string()                    # table name
            take("q")                   # available key
            table = {}
            for _ in range(take("q")):
                sym = string()
                table[take("q")] = sym
            return table
        isyms = symbol_table() if flags & 1 else None
        osyms = symbol_table() if flags & 2 else None
        off, il, wt, ns, fin = [0], [], [], [], []
        for _ in range(nstates):
            fin.append(take("f"))
            for _ in range(take("q")):
                il.append(take("i")); take("i"); wt.append(take("f")); ns.append(take("i"))
            off.append(len(il))
        fst = cls(off, il, wt, ns, fin, start)
        fst.isymbols, fst.osymbols = isyms, osyms
        return fst


class SortedMatcher(object):
    """Same constructor and return conventions as decoder/sorted_matcher.py:15-111."""

    def __init__(self, vector_fst, max_num_arcs, max_id, backoff_id, disambig_ids):
        self.fst = vector_fst
        self.max_num_arcs = max_num_arcs
        self.max_id = max_id
        self.backoff_id = backoff_id
        self.disambig_ids = disambig_ids

    def search(self, state_id, ilabel):
        """Index of the arc `ilabel` out of `state_id`, or -1.  The reference bisects a window of
        `max_num_arcs` slots whose tail beyond the real arcs reads as `max_id` (:30-48): a lower
        bound over the first min(narcs, max_num_arcs) arcs, clamped to the window's last slot."""
        f = self.fst
        lo, hi = int(f.offsets[state_id]), int(f.offsets[state_id + 1])
        n = hi - lo
        w = min(n, self.max_num_arcs)
        idx = int(np.searchsorted(f.ilabel[lo:lo + w], ilabel, side="left"))
        if idx >= w:
            # every real arc in the window is smaller: the virtual tail (value max_id) decides
            idx = w if (w < self.max_num_arcs and self.max_id >= ilabel) else self.max_num_arcs - 1
        if idx >= n or f.ilabel[lo + idx] != ilabel:
            return -1
        return lo + idx

    def get_scores_wodisambig(self, state_id, ilabel, init_score=0.0):
        f, scores, states = self.fst, [], []
        bf, cur = init_score, state_id
        while True:
            a = self.search(cur, ilabel)
            if a >= 0:
                scores.append(bf + float(f.weight[a]))
                states.append(int(f.nextstate[a]))
            b = self.search(cur, self.backoff_id)
            if b < 0:
                return scores, states
            bf += float(f.weight[b])
            cur = int(f.nextstate[b])

    def get_scores(self, state_id, ilabel):
        init_scores, init_states = [0.0], [state_id]
        for label in self.disambig_ids:
            a = self.search(state_id, label)
            if a >= 0:
                init_scores.append(float(self.fst.weight[a]))
                init_states.append(int(self.fst.nextstate[a]))
        scores, states = [], []
        for s0, st in zip(init_scores, init_states):
            sc, ns = self.get_scores_wodisambig(st, ilabel, s0)
            scores.extend(sc)
            states.extend(ns)
        return scores, states

    def final_score(self, state_id):
        f = self.fst
        fs, st = [0.0], [state_id]
        for label in self.disambig_ids:
            a = self.search(state_id, label)
            if a >= 0:
                fs.append(float(f.weight[a]))
                st.append(int(f.nextstate[a]))
        for i in range(len(fs)):
            score, cur = fs[i], st[i]
            while True:
                fw = float(f.final[cur])
                if math.isinf(fw):
                    b = self.search(cur, self.backoff_id)
                    if b < 0:
                        score, cur = float("inf"), None
                        break
                    score += float(f.weight[b])
                    cur = int(f.nextstate[b])
                else:
                    score += fw
                    break
            fs[i], st[i] = score, cur
        return fs, st
