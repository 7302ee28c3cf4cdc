#!/usr/bin/env python3
"""Fixture: what the REAL reference prints on stderr (dab.c:51,57,78-82 -> misc.c:316-328) while it processes the golden back-end run
(backend_e2e.npz: 34 TF at 9 dB, one TF with a destroyed FIC = a lock loss and a re-lock) -> backend_e2e_stderr.txt.
Run in the build container (needs /root/reference through oracle/_ref/libdabref.so); the text is data: the reference's output on the committed input."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as ol  # noqa: E402


def reference_stderr(tf_bits):
    R = ol.ref()
    Hd = R.refh_new()
    libc = C.CDLL(None)
    with tempfile.TemporaryFile() as tmp:
        saved = os.dup(2)
        libc.fflush(None)
        os.dup2(tmp.fileno(), 2)
        try:
            for row in tf_bits:
                bits = np.ascontiguousarray(np.unpackbits(row))
                C.memmove(R.refh_tf_fic(Hd), ol._ptr(bits[:9216]), 9216)
                C.memmove(R.refh_tf_msc(Hd), ol._ptr(bits[9216:9216 + 221184]), 221184)
                R.refh_process(Hd)
            libc.fflush(None)
        finally:
            os.dup2(saved, 2)
            os.close(saved)
        tmp.seek(0)
        return tmp.read().decode("ascii"), R.refh_neti(Hd)


if __name__ == "__main__":
    g = np.load(os.path.join(HERE, "backend_e2e.npz"))
    text, n = reference_stderr(g["tf_bits"])
    assert n == len(g["eti"]), (n, len(g["eti"]))
    with open(os.path.join(HERE, "backend_e2e_stderr.txt"), "w") as f:
        f.write(text)
    sys.stdout.write(text)
