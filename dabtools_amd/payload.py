"""What a decoded ETI stream carries against what the synthetic modulator sent (bench.py's noisy configurations, the CPU baselines'
BER): frames whose FIC names a CIF of the ensemble are compared sub-channel by sub-channel with dabhip_synth_payload."""
import numpy as np


def bench_cfg(dab, global_stream, snr_db=1000.0):
    """The ensemble bench.py gives global stream g (seed = 1000 * config + stream, config 2)."""
    from . import shard
    return dab.synth_preset(0, seed=shard.stream_seed(2, global_stream), cif_count0=(97 * global_stream) % 5000, snr_db=snr_db)


class PayloadCheck:
    def __init__(self):
        self.frames = self.good = self.bit_err = self.bits = self.streams = self.expected = self.compared = 0

    def add_stream(self, dab, cfg, ntf, eti_frames):
        """eti_frames: iterable of 6144-byte uint8 arrays, the frames one decoder produced for this stream."""
        fib_index = {dab.synth_fibs(cfg, c).tobytes(): c for c in range(4 * ntf)}
        self.streams += 1
        self.expected += 4 * (ntf - 15)
        for e in eti_frames:
            self.frames += 1
            nst = int(e[5]) & 0x7f
            pos = 12 + 4 * nst
            cif = fib_index.get(e[pos:pos + 96].tobytes())
            if cif is None or nst != cfg.nsub:
                continue                                   # (counted: frames_unmatched in result())
            self.compared += 1
            pos += 96
            wrong = 0
            for k in range(nst):
                want = dab.synth_payload(cfg, cif, k)
                wrong += int(np.unpackbits(np.bitwise_xor(e[pos:pos + want.size], want)).sum())
                self.bits += 8 * want.size
                pos += want.size
            self.bit_err += wrong
            self.good += int(wrong == 0)

    def result(self):
        # The payload of a frame can only be looked up when its 96 FIC bytes name a CIF of the ensemble exactly: frames with a damaged FIC (or a
        # different NST) are NOT in payload_ber -- at 5 dB that is most frames of a hard-decision decoder, so the figure flatters it; frames_compared
        # and frames_unmatched say how many frames the BER stands on, payload_ber_unmatched_as_half counts every unmatched frame as coin tosses
        unmatched = self.frames - self.compared
        per_frame = (self.bits / self.compared) if self.compared else 0.0
        return {"streams_checked": self.streams, "frames_expected_if_locked": self.expected, "frames_out": self.frames,
                "frames_compared": self.compared, "frames_unmatched": unmatched,
                "error_free_frames": self.good, "payload_ber": (self.bit_err / self.bits) if self.bits else None,
                "payload_ber_is": "bit errors / payload bits over the frames_compared FIC-matched frames",
                "payload_ber_unmatched_as_half": ((self.bit_err + 0.5 * per_frame * unmatched) / (self.bits + per_frame * unmatched)) if self.bits else None}
