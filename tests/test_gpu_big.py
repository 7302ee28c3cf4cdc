"""Maximum sizes: one stream longer than 4 GiB (byte offsets beyond 2^32, a session segment beyond 2^31 bytes, 44,740 ETI frames, the CIF counter
wrapping at 5000 eight times), decoded in one call and by a session in odd segments: the same bytes, dab2eti's 4 (T - 15) frames, and the CPU
oracle's bytes on the first and on the last 150 transmission frames (the latter start 4.3 GB into the stream).  tools/big_stream_check.py."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_one_stream_beyond_four_gib():
    import big_stream_check
    out = big_stream_check.run(tfs=11200, oracle_tfs=150)
    assert out["ok"], out
    assert out["one_shot"]["eti_frames"] == 4 * (11200 - 15) and out["session"]["equal_to_one_shot"]
    assert out["session"]["largest_segment_bytes"] > 2 ** 31
    assert out["session_segment_just_below_4gib"]["equal_to_one_shot"] and 2 ** 32 - (4 << 20) < out["session_segment_just_below_4gib"]["segment_bytes"] < 2 ** 32
    assert out["oracle_tail"]["first_byte_offset"] > 2 ** 32
