"""The decoder with two and with four lanes per code word (vit_two_lanes.hpp, vit_four_lanes.hpp: mid-size batches, hard decisions) against the lane form and
the oracle: the same bytes.  (The whole suite also runs with every lane-form decode forced through them: tools/gpu/two_lanes.sh, four_lanes.sh STAGE=suite.)"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _engine(dab, form):
    """form: 0 = one lane per code word, 1 = two lanes (vit_two_lanes.hpp), 2 = two lanes without tables, 4 = four lanes (vit_four_lanes.hpp)"""
    old = {k: os.environ.get(k) for k in ("DABHIP_VIT_TWO_LANES", "DABHIP_VIT_FOUR_LANES", "DABHIP_VIT_LANES_PLAIN", "DABHIP_VIT_WAVE_MAX")}
    os.environ["DABHIP_VIT_TWO_LANES"] = "1" if form in (1, 2) else "0"
    os.environ["DABHIP_VIT_LANES_PLAIN"] = "1" if form == 2 else "0"
    os.environ["DABHIP_VIT_FOUR_LANES"] = "1" if form == 4 else "0"
    os.environ["DABHIP_VIT_WAVE_MAX"] = "0"                  # no wave-per-code-word form: the batch below is the lane form's or the two-lane form's
    try:
        return dab.Engine(0)                                  # (the knobs are read when an engine is made)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("preset,streams,tfs", [(0, 10, 26), (1, 7, 21)])
def test_two_lanes_per_code_word_decode_what_the_lane_form_and_the_oracle_decode(preset, streams, tfs):
    import dabtools_amd as dab
    import oracle_lib as ol
    # noisy enough that the decoders correct errors all the time and metrics tie (9 .. 13 dB, hard decisions), every stream its own payload and offset
    caps = [dab.synth_generate(dab.synth_preset(preset, seed=7100 + 13 * b + preset, cif_count0=37 * b, skip_samples=1000 * b, snr_db=9.0 + (b % 5)), tfs) for b in range(streams)]
    out = {}
    for mode in (0, 1, 2, 4):
        eng = _engine(dab, mode)
        n = eng.decode(caps)
        out[mode] = [eng.eti(b) for b in range(streams)]
        assert n == sum(x.shape[0] for x in out[mode]) and n > 0
        eng.close()
    for mode in (1, 2, 4):
        for b in range(streams):
            assert out[0][b].shape == out[mode][b].shape and np.array_equal(out[0][b], out[mode][b]), (mode, b)
    for b in (0, streams - 1):
        want, _ = ol.or_replay(caps[b], cap_frames=4 * tfs)
        assert want.shape[0] > 0 and np.array_equal(out[4][b], want) and np.array_equal(out[1][b], want), b


def test_the_default_rule_takes_two_lanes_for_a_mid_size_batch_and_the_bytes_do_not_depend_on_it():
    import dabtools_amd as dab
    # 16 streams x 24 TF of the 12-sub-channel multiplex = 6,912 code words in 9 x 4 x 16 frames: above the wave form's range only with the knob; the default
    # engine takes whatever its rules say -- the bytes must be those of the forced forms
    caps = [dab.synth_generate(dab.synth_preset(0, seed=7300 + b, snr_db=11.0), 24) for b in range(16)]
    ref = None
    for mode in (None, 0, 1, 4):
        eng = dab.Engine(0) if mode is None else _engine(dab, mode)
        eng.decode(caps)
        got = np.concatenate([eng.eti(b) for b in range(16)])
        eng.close()
        if ref is None:
            ref = got
        assert got.shape == ref.shape and np.array_equal(got, ref), mode
