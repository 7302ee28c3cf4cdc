"""ETI(NI) frame validator built from the layout the reference writes (misc.c:153-314)
and reads back (eti2mpa.c:40-66).  Test infrastructure."""
import numpy as np


def crc16_ccitt(data, crc=0xFFFF):
    for b in data:
        crc ^= int(b) << 8
        for _ in range(8):
            crc = ((crc << 1) ^ 0x1021) & 0xFFFF if crc & 0x8000 else (crc << 1) & 0xFFFF
    return crc


def parse(frame):
    """-> dict(fct, nst, fl, stc=[(scid, sad, tpl, stl)], fic, subch=[bytes]) ; raises AssertionError when malformed."""
    e = np.asarray(frame, dtype=np.uint8)
    assert e.size == 6144 and e[0] == 0xFF
    fsync = bytes(e[1:4])
    fct = int(e[4])
    assert fsync == (b"\xf8\xc5\x49" if fct & 1 else b"\x07\x3a\xb6"), "FSYNC does not match FCT parity"
    assert e[5] & 0x80, "FICF must be set"
    nst = int(e[5] & 0x7F)
    fp, mid = int(e[6]) >> 5, (int(e[6]) >> 3) & 3
    fl = ((int(e[6]) & 7) << 8) | int(e[7])
    assert mid == 1
    stc, pos = [], 8
    for _ in range(nst):
        scid, sad = int(e[pos]) >> 2, ((int(e[pos]) & 3) << 8) | int(e[pos + 1])
        tpl, stl = int(e[pos + 2]) >> 2, ((int(e[pos + 2]) & 3) << 8) | int(e[pos + 3])
        stc.append((scid, sad, tpl, stl))
        pos += 4
    assert bytes(e[pos:pos + 2]) == b"\xff\xff"                      # MNSC
    hcrc = (int(e[pos + 2]) << 8) | int(e[pos + 3])
    assert hcrc == (~crc16_ccitt(e[4:pos + 2])) & 0xFFFF, "HCRC"
    pos += 4
    assert fl == nst + 1 + 24 + sum(2 * s[3] for s in stc), "FL"
    mst0 = pos
    fic = e[pos:pos + 96].copy()
    pos += 96
    sub = []
    for s in stc:
        sub.append(e[pos:pos + 8 * s[3]].copy())
        pos += 8 * s[3]
    eof = (int(e[pos]) << 8) | int(e[pos + 1])
    assert eof == (~crc16_ccitt(e[mst0:pos])) & 0xFFFF, "EOF CRC"
    assert bytes(e[pos + 2:pos + 8]) == b"\xff" * 6
    assert (e[pos + 8:] == 0x55).all(), "padding"
    return {"fct": fct, "fp": fp, "nst": nst, "fl": fl, "stc": stc, "fic": fic, "subch": sub}


def check_sequence(frames):
    """consecutive frames of one locked run: FCT counts modulo 250, FP = count modulo 8 (as the reference guesses it)"""
    prev = None
    for f in frames:
        p = parse(f)
        if prev is not None:
            assert p["fct"] == (prev + 1) % 250
        prev = p["fct"]
    return len(frames)
