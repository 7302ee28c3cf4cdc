"""What a decoded ETI stream carries against what the synthetic modulator sent (bench.py's noisy configurations, the CPU baselines'
BER): frames whose FIC names a CIF of the ensemble are compared sub-channel by sub-channel with dabhip_synth_payload."""
import numpy as np


def bench_cfg(dab, global_stream, snr_db=1000.0):
    """The ensemble bench.py gives global stream g (seed = 1000 * config + stream, config 2)."""
    from . import shard
    return dab.synth_preset(0, seed=shard.stream_seed(2, global_stream), cif_count0=(97 * global_stream) % 5000, snr_db=snr_db)


class PayloadCheck:
    def __init__(self):
        self.frames = self.good = self.bit_err = self.bits = self.streams = self.expected = 0

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
                continue
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
        return {"streams_checked": self.streams, "frames_expected_if_locked": self.expected, "frames_out": self.frames,
                "error_free_frames": self.good, "payload_ber": (self.bit_err / self.bits) if self.bits else None}
