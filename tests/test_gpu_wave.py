"""The two forms of the channel decoder on the same input (hard decisions = reference semantics, viterbi.c:352-451):
one wave per code word (k_vitwave.hip, small batches: the single live ensemble of dab2eti.c:60-115) and one lane per code word
(viterbi_fused_kernel, the batch form) must produce identical ETI bytes -- and the oracle's."""
import numpy as np
import pytest

import dabtools_amd as dab
import oracle_lib as ol

pytestmark = pytest.mark.gpu


def _streams():
    out = []
    for seed, skip, snr, preset in ((501, 0, 1000.0, 0), (502, 41000, 9.0, 0), (503, 0, 7.5, 1), (504, 150000, 1000.0, 1)):
        cfg = dab.synth_preset(preset, seed=seed, cif_count0=4980 + seed % 7, skip_samples=skip, snr_db=snr)
        out.append(dab.synth_generate(cfg, 21))
    return out


def test_wave_form_equals_lane_form_equals_oracle(monkeypatch):
    streams = _streams()
    want = [ol.or_replay(iq)[0] for iq in streams]
    got = {}
    for form, env in (("wave", None), ("lane", "0")):
        if env is None:
            monkeypatch.delenv("DABHIP_VIT_WAVE_MAX", raising=False)
        else:
            monkeypatch.setenv("DABHIP_VIT_WAVE_MAX", env)
        eng = dab.Engine(0)
        assert eng.decode(streams) == sum(w.shape[0] for w in want)
        got[form] = [eng.eti(b) for b in range(len(streams))]
        eng.close()
    for b in range(len(streams)):
        assert want[b].shape[0] > 0
        assert np.array_equal(got["wave"][b], want[b]), "stream %d: wave form differs from the oracle" % b
        assert np.array_equal(got["lane"][b], want[b]), "stream %d: lane form differs from the oracle" % b


def test_wave_form_on_every_code_word_length_against_the_real_reference(monkeypatch):
    """All 64 UEP + 24 EEP shapes, random MSC bits (what comes out is the decoder's tie rule and metric, nothing else), through the S3 seam
    with the wave form forced for every size (the 384 kbit/s code word, 9222 steps, runs two waves to a workgroup)."""
    monkeypatch.setenv("DABHIP_VIT_WAVE_MAX", "1000000")
    import test_gpu_parity_r2 as r2
    r2.test_all_uep_and_eep_profiles_through_process_frame()
