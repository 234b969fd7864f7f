"""TEST-ONLY reader and interpreter of the gorp_amd table blob.

Lets the CPU test-suite (-m "not gpu") check what the table compiler emits
against the oracle without a GPU: it decodes the blob gx_blob_copy returns and
walks the tables exactly as gx_kernels.hip's generic kernel does.  It is an
executable statement of the kernel contract, deliberately slow and simple, and
is not importable from the product package.
"""
import struct

import numpy as np

SRC_POS = 0xFFFF
SRC_NIL = 0xFFFE


class _Reader:
    def __init__(self, buf):
        self.b = memoryview(bytes(buf))
        self.at = 0

    def pod(self, fmt):
        v = struct.unpack_from("<" + fmt, self.b, self.at)[0]
        self.at += struct.calcsize("<" + fmt)
        return v

    def vec(self, dtype):
        n = self.pod("Q")
        dt = np.dtype(dtype)
        a = np.frombuffer(self.b, dtype=dt, count=n, offset=self.at).copy()
        self.at += (n * dt.itemsize + 7) & ~7
        return a


class Blob:
    def __init__(self, buf):
        r = _Reader(buf)
        assert r.pod("I") == 0x31425847, "bad magic"
        version = r.pod("I")
        assert version in (2, 3), "bad version"
        self.n_rules = r.pod("i")
        self.ncls = r.pod("i")
        self.max_groups = r.pod("i")
        self.has_capture = r.pod("i") != 0
        self.m_states = r.pod("i")
        self.m_dead = r.pod("i")
        self.cls256 = r.vec(np.uint8)
        self.hi_lo = r.vec(np.uint16)
        self.hi_cls = r.vec(np.uint16)
        self.m_next = r.vec(np.uint32).reshape(self.m_states, self.ncls)
        self.m_accept_first = r.vec(np.int32)
        self.m_accept_off = r.vec(np.uint32)
        self.m_accept_list = r.vec(np.int32)
        self.rules = []
        for _ in range(r.pod("Q")):
            rt = {"n_groups": r.pod("i"), "n_states": r.pod("i"), "n_regs": r.pod("i"), "dead": r.pod("i")}
            rt["trans"] = r.vec(np.uint32).reshape(rt["n_states"], self.ncls)
            rt["fin"] = r.vec(np.int32)
            self.rules.append(rt)
        self.ops_off = r.vec(np.uint32)
        self.ops = r.vec(np.uint16)
        self.fin_tags = r.vec(np.uint16)
        self.union_ok = r.pod("i") != 0
        r.pod("i")  # padding
        self.uni = None
        if self.union_ok:
            u = {"n_groups": r.pod("i"), "n_states": r.pod("i"), "n_regs": r.pod("i"), "dead": r.pod("i")}
            u["trans"] = r.vec(np.uint32).reshape(u["n_states"], self.ncls)
            u["fin"] = r.vec(np.int32)
            self.uni = u
        # version 3: programs of the extractions whose capture automaton is not built ahead of time
        self.pike_off = self.pike_code = self.pike_sets = None
        if version >= 3:
            self.pike_off, self.pike_code, self.pike_sets = r.vec(np.uint32), r.vec(np.uint32), r.vec(np.uint32)
            if len(self.pike_code) == 0:
                self.pike_off = None

    def is_pike(self, k):
        return self.pike_off is not None and int(self.pike_off[k + 1]) != int(self.pike_off[k])

    def pike_capture(self, k, cls):
        """gx_kernels.hip: pike_capture -- extraction k's program run as it is: thread lists in priority order."""
        base, n_inst = int(self.pike_off[k]), int(self.pike_off[k + 1]) - int(self.pike_off[k])
        code = [(int(self.pike_code[2 * (base + q)]) & 0xFF, int(self.pike_code[2 * (base + q)]) >> 8, int(self.pike_code[2 * (base + q) + 1])) for q in range(n_inst)]
        ng = self.rules[k]["n_groups"]

        def add_thread(lst, seen, pc0, caps, pos):
            stack = [(pc0, None, None)]
            while stack:
                pc, slot, val = stack.pop()
                if pc < 0:
                    caps[slot] = val
                    continue
                if pc in seen:
                    continue
                seen.add(pc)
                op, x, y = code[pc]
                if op == 1:
                    stack.append((y, None, None)); stack.append((x, None, None))
                elif op == 2:
                    stack.append((x, None, None))
                elif op == 3:
                    stack.append((-1, x, caps[x]))
                    caps[x] = pos
                    stack.append((pc + 1, None, None))
                elif op in (0, 4):
                    lst.append((pc, list(caps)))

        clist = []
        add_thread(clist, set(), 0, [-1] * (2 * ng), 0)
        for p, c in enumerate(cls):
            nlist, seen = [], set()
            for pc, caps in clist:
                op, x, _ = code[pc]
                if op != 0 or not (int(self.pike_sets[8 * x + (c >> 5)]) >> (c & 31)) & 1:
                    continue
                add_thread(nlist, seen, pc + 1, list(caps), p + 1)
            clist = nlist
            if not clist:
                break
        for pc, caps in clist:
            if code[pc][0] == 4:
                return caps
        return None

    # -- kernel contract ---------------------------------------------------
    def class_of(self, c):
        if c < 256:
            return int(self.cls256[c])
        i = int(np.searchsorted(self.hi_lo, c, side="right")) - 1
        return int(self.hi_cls[i])

    def classes(self, units):
        return [self.class_of(int(c)) for c in units]

    def match_state(self, units):
        st = 0
        for c in self.classes(units):
            st = int(self.m_next[st, c])
            if st == self.m_dead:
                break
        return st

    def match(self, units):
        st = self.match_state(units)
        return self.m_accept_list[self.m_accept_off[st]:self.m_accept_off[st + 1]].tolist()

    def extract(self, units):
        """(match_id, [(b,e)|None]*groups) with the C-ABI's encoding of outcomes."""
        cls = self.classes(units)
        st = 0
        for c in cls:
            st = int(self.m_next[st, c])
            if st == self.m_dead:
                break
        k = int(self.m_accept_first[st])
        if k < 0:
            return -1, []
        rt = self.rules[k]
        if self.is_pike(k):
            got = self.pike_capture(k, cls)
            if got is None:
                return -2 - k, []
            return k, [None if got[2 * g] < 0 or got[2 * g + 1] < 0 else (got[2 * g], got[2 * g + 1]) for g in range(rt["n_groups"])]
        regs = {}
        ts = 0
        for p, c in enumerate(cls):
            w = int(rt["trans"][ts, c])
            ts = w & 0xFFFF
            op = w >> 16
            if op:
                for j in range(int(self.ops_off[op]), int(self.ops_off[op + 1])):
                    dst, src = int(self.ops[2 * j]), int(self.ops[2 * j + 1])
                    regs[dst] = p if src == SRC_POS else regs[src]
        f = int(rt["fin"][ts])
        if f < 0:
            return -2 - k, []
        caps = []
        n = len(cls)
        for g in range(rt["n_groups"]):
            vb, ve = int(self.fin_tags[f + 2 * g]), int(self.fin_tags[f + 2 * g + 1])
            pb = n if vb == SRC_POS else (-1 if vb == SRC_NIL else regs[vb])
            pe = n if ve == SRC_POS else (-1 if ve == SRC_NIL else regs[ve])
            caps.append(None if pb < 0 or pe < 0 else (pb, pe))
        return k, caps


def _extract_union(self, units):
    """Single pass over the fused automaton (match automaton x joined capture automata): same result encoding."""
    assert self.union_ok
    u = self.uni
    cls = self.classes(units)
    regs = {}
    ts = 0
    for p, c in enumerate(cls):
        w = int(u["trans"][ts, c])
        ts = w & 0xFFFF
        op = w >> 16
        if op:
            for j in range(int(self.ops_off[op]), int(self.ops_off[op + 1])):
                dst, src = int(self.ops[2 * j]), int(self.ops[2 * j + 1])
                regs[dst] = p if src == SRC_POS else regs[src]
    f = int(u["fin"][ts])
    if f < 0:
        return f, []
    k = int(self.fin_tags[f])
    caps = []
    n = len(cls)
    for g in range(self.rules[k]["n_groups"]):
        vb, ve = int(self.fin_tags[f + 1 + 2 * g]), int(self.fin_tags[f + 2 + 2 * g])
        pb = n if vb == SRC_POS else (-1 if vb == SRC_NIL else regs[vb])
        pe = n if ve == SRC_POS else (-1 if ve == SRC_NIL else regs[ve])
        caps.append(None if pb < 0 or pe < 0 else (pb, pe))
    return k, caps


Blob.extract_union = _extract_union


def units_of(s):
    if isinstance(s, (bytes, bytearray)):
        return list(s)
    raw = s.encode("utf-16-le", "surrogatepass")
    return list(np.frombuffer(raw, dtype=np.uint16)) if raw else []
