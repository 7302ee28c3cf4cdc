#!/usr/bin/env python3
"""tools/models/twolane_model.py -- the two-lanes-per-code-word decoder's index algebra, checked on the CPU before any GPU time.

A code word's 64 path metrics live in 2 lanes x 16 registers of two packed 16-bit states.  The PAIR bit (which two states share a register) walks through
places 0..3 as in the lane form (k_decode.hip: layouts L(tau), re-paired every fourth step); the LANE bit (which lane holds a state) walks through places
3, 4, 5, 0, 1, 2, ... one place per trellis step, because a state bit moves up one place per step: with the lane bit below place 5 a butterfly's two
predecessors share it (the step is local to the lane); at place 5 every butterfly straddles the lanes and lane 0 computes all even successors, lane 1 all odd
ones, each from ITS registers and its partner's registers OF THE SAME INDEX (a DPP operand).  This model runs that schedule with the formulas of the kernel
(phys_of / state_of, the per-lane metric tables, the re-pairing, the survivor record's byte positions) against a plain 64-state add-compare-select with
tags, on random branch metrics, and asserts equality after every step.  Usage: twolane_model.py [steps=480]"""
import random
import sys


def parity(x):
    return bin(x).count("1") & 1


def code3(i):                      # k_decode.hip: branch_code3
    return parity(i & 0x6d) | (parity(i & 0x4f) << 1) | (parity(i & 0x53) << 2)


def expand_bit(r, tau):
    return ((r >> tau) << (tau + 1)) | (r & ((1 << tau) - 1))


def compress_bit(k, tau):
    return ((k >> (tau + 1)) << tau) | (k & ((1 << tau) - 1))


def remove_bit(r, b):
    return ((r >> (b + 1)) << b) | (r & ((1 << b) - 1))


def insert_bit(q, b, v):
    return ((q >> b) << (b + 1)) | (v << b) | (q & ((1 << b) - 1))


def lane_of(k, L):
    return (k >> L) & 1


def phys_of(k, tau, L):
    """register of state k (either member of its pair) inside its lane, layout (pair bit tau = 0..4, lane bit L != tau)"""
    side, r = k >> 5, compress_bit(k & 31, tau)
    if L == 5:
        return r
    lr = L if L < tau else L - 1
    return side * 8 + remove_bit(r, lr)


def state_of(P, tau, L):
    """low member (pair bit clear) of the pair in register P of LANE 0"""
    if L == 5:
        return expand_bit(P, tau)
    lr = L if L < tau else L - 1
    return 32 * (P >> 3) + expand_bit(insert_bit(P & 7, lr, 0), tau)


def lane_bit(t):
    return (3 + t) % 6


def check_layouts():
    for tau in range(5):
        for L in range(6):
            if L == tau:
                continue
            seen = set()
            for k in range(64):
                seen.add((lane_of(k, L), phys_of(k, tau, L), (k >> tau) & 1))
            assert len(seen) == 64, (tau, L)
            for P in range(16):
                k = state_of(P, tau, L)
                assert lane_of(k, L) == 0 and ((k >> tau) & 1) == 0 and phys_of(k, tau, L) == P, (tau, L, P, k)
    for t in range(48):
        assert lane_bit(t) != t % 4                      # never on the pair bit
        if t % 4 == 0:
            assert lane_bit(t) not in (0, 4)              # the re-pairing joins states that differ in bits 0 and 4: both in one lane


def reference_step(M, bm, tag):
    new = [0] * 64
    for j in range(32):
        c = code3(2 * j)
        x, y = M[j], M[j + 32]
        new[2 * j] = max(x + bm[c] + tag, y + bm[c ^ 7])
        new[2 * j + 1] = max(x + bm[c ^ 7] + tag, y + bm[c])
    return new


def tables(bm, tag, tau, L, lane):
    """the 8 + 8 packed metric words (A: added to the lane's own low-index operand, B: to the other) as (lo, hi) tuples, by the kernel's rule"""
    gamma = code3(2 << tau)
    def word(c, tagged):
        return (bm[c] + (tag if tagged else 0), bm[c ^ gamma] + (tag if tagged else 0))
    if lane == 0:
        return [word(i, True) for i in range(8)], [word(i, False) for i in range(8)]
    if L == 5:                                            # exchange step: lane 1's own operand is the HIGH predecessor
        return [word(i, False) for i in range(8)], [word(i, True) for i in range(8)]
    g = code3(2 << L)
    return [word(i ^ g, True) for i in range(8)], [word(i ^ g, False) for i in range(8)]


def add(a, b):
    return (a[0] + b[0], a[1] + b[1])


def vmax(a, b):
    return (max(a[0], b[0]), max(a[1], b[1]))


def read_state(regs, k, tau, L):
    v = regs[lane_of(k, L)][phys_of(k, tau, L)]
    return v[(k >> tau) & 1]


def main():
    check_layouts()
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 480
    rnd = random.Random(7)
    M = [0] * 64
    M[0] = 64 * 256
    regs = [[(0, 0)] * 16 for _ in range(2)]
    regs[0][0] = (M[0], 0)
    tau = 0
    records = []
    for t in range(steps):
        L = lane_bit(t)
        assert tau == t % 4
        for k in range(64):
            assert read_state(regs, k, tau, L) == M[k], (t, k)
        n = rnd.randrange(5)
        v, m = rnd.randrange(16), (1 << n) - 1
        bm = [0] * 8
        for c in range(4):
            cw = c | ((c & 1) << 3)
            bm[c] = 256 * bin(~(v ^ cw) & m & 15).count("1")
            bm[c ^ 7] = 256 * n - bm[c]
        tag = 1 << (t % 8)
        Mn = reference_step(M, bm, tag)
        new = [[None] * 16 for _ in range(2)]
        for lane in range(2):
            A, B = tables(bm, tag, tau, L, lane)
            p = regs[lane]
            if L == 5:
                for r in range(16):
                    c = code3(2 * expand_bit(r, tau))
                    own, other = p[r], regs[lane ^ 1][r]               # the partner's register OF THE SAME INDEX
                    out = vmax(add(own, A[c]), add(other, B[c ^ 7]))
                    P = phys_of(2 * expand_bit(r, tau), tau + 1, 0)
                    assert new[lane][P] is None
                    new[lane][P] = out
            else:
                lr = L if L < tau else L - 1
                for q in range(8):
                    j0 = expand_bit(insert_bit(q, lr, 0), tau)
                    c = code3(2 * j0)
                    x, y = p[q], p[8 + q]
                    e = vmax(add(x, A[c]), add(y, B[c ^ 7]))
                    o = vmax(add(x, A[c ^ 7]), add(y, B[c]))
                    Pe, Po = phys_of(2 * j0, tau + 1, L + 1), phys_of(2 * j0 + 1, tau + 1, L + 1)
                    assert new[lane][Pe] is None and new[lane][Po] is None and Pe != Po
                    new[lane][Pe], new[lane][Po] = e, o
        regs, M, tau = new, Mn, tau + 1
        Ln = lane_bit(t + 1)
        if t % 8 == 7:                                    # survivor record: byte 2 (P & 1) + half of word P >> 1, per lane; then the tags go
            rec = [[0] * 8 for _ in range(2)]
            for lane in range(2):
                for P in range(16):
                    for half in range(2):
                        rec[lane][P >> 1] |= (regs[lane][P][half] & 255) << (8 * (2 * (P & 1) + half))
            records.append((rec, Ln))
            for k in range(64):                           # the chain-back's address of state k: lane, word, byte
                side, r = k >> 5, k & 15
                lane = (k >> 5) if Ln == 5 else (k >> Ln) & 1
                P = r if Ln == 5 else side * 8 + remove_bit(r, Ln)
                half = (k >> 4) & 1
                assert (rec[lane][P >> 1] >> (8 * (2 * (P & 1) + half))) & 255 == M[k] & 255, (t, k)
            M = [x & ~255 for x in M]
            regs = [[(a & ~255, b & ~255) for a, b in lane] for lane in regs]
        if tau == 4:                                      # re-pair (k, k ^ 16) -> (k, k ^ 1): inside each lane
            new = [[None] * 16 for _ in range(2)]
            for lane in range(2):
                for P in range(16):
                    k = state_of(P, 0, Ln)                # lane 0's state; lane 1's differs in the lane bit only: same registers
                    a, b, half = phys_of(k, 4, Ln), phys_of(k + 1, 4, Ln), (k >> 4) & 1
                    new[lane][P] = (regs[lane][a][half], regs[lane][b][half])
            regs, tau = new, 0
        if (t + 1) % 256 == 0:
            base = M[0] - 64 * 256
            M = [x - base for x in M]
            regs = [[(a - base, b - base) for a, b in lane] for lane in regs]
    lanes = {}
    for b, (_, Ln) in enumerate(records):
        lanes.setdefault(b % 3, set()).add(Ln)
    assert lanes == {0: {5}, 1: {1}, 2: {3}}, lanes
    print("two lanes per code word: %d steps equal to the 64-state reference; layouts bijective; lane bit at the end of block b: 5, 1, 3 for b mod 3 = 0, 1, 2" % steps)


if __name__ == "__main__":
    main()
