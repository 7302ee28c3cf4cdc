#!/usr/bin/env python3
"""tools/models/multilane_model.py -- 2^NL lanes per code word (NL = 1, 2): the index algebra of vit_four_lanes.hpp on the CPU.

Generalises twolane_model.py: NL rotating LANE bits, place L_i(t) = (a_i + t) mod 6 with a = (3,) or (3, 5), beside the pair bit tau = t mod 4; a lane holds
the 2^(5 - NL) registers of the states whose bits at the lane places equal its lane id.  Registers are numbered by COMPACTION: the state's bits in place
order with the pair bit and the lane bits taken out.  A step is local unless one lane bit sits at place 5; then the lanes that differ in that bit exchange
(each computes the successors whose new bit 0 equals its id bit, from its own and its partner's registers of the same index).
New against the two-lane kernel: NO per-lane metric tables.  A lane's branch codes are code(2 j) ^ g with g = XOR of code(2 << L_i) over its set lane bits
below place 5, and the metric of code c ^ g for the received nibble v is the metric of code c for v ^ cw(g) (cw = the 4-bit code word, linear in c): the
lane reads the ONE table at the row of (v ^ cw(g)) & mask.  At an exchange over bit i the lanes with that bit set own the HIGH predecessor: they read the
tagged and untagged parts swapped.  Asserts equality with a plain 64-state add-compare-select (with tags) after every step, the re-pairing inside a lane, and
the byte positions of the survivor record that the chain-back uses.  Usage: multilane_model.py [steps=480]"""
import random
import sys


def parity(x):
    return bin(x).count("1") & 1


def code3(i):
    return parity(i & 0x6d) | (parity(i & 0x4f) << 1) | (parity(i & 0x53) << 2)


def cw4(c):                         # the 4-bit code word of 3-bit code c (bit 3 = bit 0): linear in c
    return c | ((c & 1) << 3)


def compact(k, removed):
    """k's bits in place order with the places in `removed` taken out"""
    out, pos = 0, 0
    for b in range(6):
        if b in removed:
            continue
        out |= ((k >> b) & 1) << pos
        pos += 1
    return out


def expand(P, removed):
    """inverse of compact with zeros at the removed places"""
    out, pos = 0, 0
    for b in range(6):
        if b in removed:
            continue
        out |= ((P >> pos) & 1) << b
        pos += 1
    return out


def places(starts, t):
    return tuple((a + t) % 6 for a in starts)


def lane_of(k, Ls):
    return sum(((k >> L) & 1) << i for i, L in enumerate(Ls))


def phys_of(k, tau, Ls):
    return compact(k, set(Ls) | {tau})


def state_of(P, tau, Ls):
    return expand(P, set(Ls) | {tau})


def reference_step(M, bm, tag):
    new = [0] * 64
    for j in range(32):
        c = code3(2 * j)
        x, y = M[j], M[j + 32]
        new[2 * j] = max(x + bm[c] + tag, y + bm[c ^ 7])
        new[2 * j + 1] = max(x + bm[c ^ 7] + tag, y + bm[c])
    return new


def metrics(v, n):
    m = (1 << n) - 1
    bm = [0] * 8
    for c in range(4):
        bm[c] = 256 * bin(~(v ^ cw4(c)) & m & 15).count("1")
        bm[c ^ 7] = 256 * n - bm[c]
    return bm


def table_row(v, n, tau, tag):
    """the ONE table's row for received value v (n bits): A (tagged: low predecessor) and B words for codes 0..7, as (lo, hi) of the pair (c, c ^ gamma)"""
    bm = metrics(v, n)
    gamma = code3(2 << tau)
    A = [(bm[c] + tag, bm[c ^ gamma] + tag) for c in range(8)]
    B = [(bm[c], bm[c ^ gamma]) for c in range(8)]
    return A, B


def add(a, b):
    return (a[0] + b[0], a[1] + b[1])


def vmax(a, b):
    return (max(a[0], b[0]), max(a[1], b[1]))


def run(starts, steps, seed):
    NL = len(starts)
    nlanes, nreg = 1 << NL, 1 << (5 - NL)
    for t in range(48):                                    # the schedule's conditions
        Ls = places(starts, t)
        assert t % 4 not in Ls and len(set(Ls)) == NL
        if t % 4 == 0:
            assert not (set(Ls) & {0, 4})
    rnd = random.Random(seed)
    M = [0] * 64
    M[0] = 64 * 256
    regs = [[(0, 0)] * nreg for _ in range(nlanes)]
    regs[0][0] = (M[0], 0)
    tau = 0
    for t in range(steps):
        Ls = places(starts, t)
        for k in range(64):
            assert regs[lane_of(k, Ls)][phys_of(k, tau, Ls)][(k >> tau) & 1] == M[k], (t, k)
        n = rnd.randrange(5)
        v = rnd.randrange(16) & ((1 << n) - 1)
        tag = 1 << (t % 8)
        Mn = reference_step(M, metrics(v, n), tag)
        Ln = places(starts, t + 1)
        at5 = [i for i, L in enumerate(Ls) if L == 5]
        new = [[None] * nreg for _ in range(nlanes)]
        for lane in range(nlanes):
            g = 0
            for i, L in enumerate(Ls):
                if L < 5 and (lane >> i) & 1:
                    g ^= code3(2 << L)
            A, B = table_row((v ^ cw4(g)) & ((1 << n) - 1), n, tau, tag)      # the lane's row of the one table
            p = regs[lane]
            if at5:
                i5 = at5[0]
                if (lane >> i5) & 1:
                    # this lane owns the high predecessor: parts swapped (its own operand takes the untagged words)
                    A, B = [(a[0] - tag, a[1] - tag) for a in A], [(b[0] + tag, b[1] + tag) for b in B]
                partner = regs[lane ^ (1 << i5)]
                for r in range(nreg):
                    j = state_of(r, tau, Ls)               # canonical: all lane bits 0 (side 0)
                    c = code3(2 * j)
                    out = vmax(add(p[r], A[c]), add(partner[r], B[c ^ 7]))
                    P = phys_of(2 * j, tau + 1, Ln)
                    assert new[lane][P] is None
                    new[lane][P] = out
            else:
                for q in range(nreg // 2):
                    j0 = state_of(q, tau, Ls)              # side 0 (bit 5 is the top remaining bit), lane bits 0
                    assert j0 < 32 and state_of(q + nreg // 2, tau, Ls) == j0 + 32
                    c = code3(2 * j0)
                    x, y = p[q], p[q + nreg // 2]
                    e = vmax(add(x, A[c]), add(y, B[c ^ 7]))
                    o = vmax(add(x, A[c ^ 7]), add(y, B[c]))
                    Pe, Po = phys_of(2 * j0, tau + 1, Ln), phys_of(2 * j0 + 1, tau + 1, Ln)
                    assert new[lane][Pe] is None and new[lane][Po] is None and Pe != Po
                    new[lane][Pe], new[lane][Po] = e, o
        regs, M, tau = new, Mn, tau + 1
        if t % 8 == 7:                                     # the record: register P -> word P >> 1, byte 2 (P & 1) + half, per lane
            rec = [[0] * (nreg // 2) for _ in range(nlanes)]
            for lane in range(nlanes):
                for P in range(nreg):
                    for half in range(2):
                        rec[lane][P >> 1] |= (regs[lane][P][half] & 255) << (8 * (2 * (P & 1) + half))
            for k in range(64):                            # the chain-back's address of state k
                lane, P, half = lane_of(k, Ln), compact(k, set(Ln) | {4}), (k >> 4) & 1
                assert (rec[lane][P >> 1] >> (8 * (2 * (P & 1) + half))) & 255 == M[k] & 255, (t, k)
            M = [x & ~255 for x in M]
            regs = [[(a & ~255, b & ~255) for a, b in lane] for lane in regs]
        if tau == 4:                                       # re-pair inside each lane
            new = [[None] * nreg for _ in range(nlanes)]
            for lane in range(nlanes):
                for P in range(nreg):
                    k = state_of(P, 0, Ln)
                    a, b, half = phys_of(k, 4, Ln), phys_of(k + 1, 4, Ln), (k >> 4) & 1
                    new[lane][P] = (regs[lane][a][half], regs[lane][b][half])
            regs, tau = new, 0
        if (t + 1) % 32 == 0:
            base = M[0] - 64 * 256
            M = [x - base for x in M]
            regs = [[(a - base, b - base) for a, b in lane] for lane in regs]
    return True


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 480
    for starts in ((3,), (3, 5), (1, 3), (1, 5)):
        run(starts, steps, 11 + len(starts))
        print("%d lanes per code word, lane bits start at %s: %d steps equal to the 64-state reference" % (1 << len(starts), starts, steps))


if __name__ == "__main__":
    main()
