"""GPU tests of the multi-device batch entry (dabhip_multi_*; BASELINE configs[3] = 2048 streams as
256 per GPU x 8).  The boxes of this pool have ONE GPU, so the eight slices of a node are all mapped onto device 0 (a device
may be listed more than once; every entry is a slice with its own engine, host thread and HIP streams): what is proven here
is the sharding rule, the concurrency of the slices' host sides and byte-equality with the single-engine decode and the
oracle -- not a scaling figure."""
import os
import subprocess

import numpy as np
import pytest

import dabtools_amd as dab
import oracle_lib as ol
from dabtools_amd import shard

pytestmark = pytest.mark.gpu


def _captures(n, ntf=19):
    caps = []
    for i in range(n):
        cfg = dab.synth_preset(i % 2, seed=3100 + i, cif_count0=(611 * i) % 5000, skip_samples=(0, 77001, 0, 1234)[i % 4],
                               snr_db=(1000.0, 1000.0, 12.0)[i % 3])
        caps.append(dab.synth_generate(cfg, ntf + i % 3))
    return caps


def test_eight_slices_on_one_gpu_equal_single_engine_and_oracle():
    """19 streams dealt to 8 slices (3,3,3,2,2,2,2,2 -- the remainder goes to the low slices, shard.shard_streams' rule), all
    slices decoding concurrently on GPU 0: every stream's ETI bytes and per-call trace equal the single-engine decode of the same
    batch; six streams are also held against the oracle; the drain delivers the frames in global stream order."""
    caps = _captures(19)
    single = dab.Engine(0)
    total_single = single.decode(caps)
    multi = dab.Multi([0] * 8)
    total = multi.decode(caps)
    assert total == total_single > 19 * 8
    for rank in range(8):
        for s in shard.shard_streams(len(caps), 8, rank):
            assert multi.slice_of(s) == (rank, 0)
    for b, iq in enumerate(caps):
        want = single.eti(b)
        got = multi.eti(b)
        assert got.shape == want.shape and np.array_equal(got, want), b
        ncalls = iq.size // dab.CHUNK_BYTES
        ti, tf = single.trace(b, ncalls)
        mi, mf = multi.trace(b, ncalls)
        assert np.array_equal(ti, mi) and np.array_equal(tf, mf), b
    for b in (0, 3, 7, 8, 13, 18):
        assert np.array_equal(multi.eti(b), ol.or_replay(caps[b])[0]), b
    drained = multi.drain()
    assert [b for b, _ in drained] == [b for b in range(len(caps)) for _ in range(multi.eti_count(b))]
    assert b"".join(f for _, f in drained) == b"".join(single.eti(b).tobytes() for b in range(len(caps)))
    # the slices really ran side by side: the call took less than the sum of the slices' own wall clocks
    walls = [multi.wall_ms(i) for i in range(8)]
    assert all(w > 0 for w in walls) and multi.wall_ms() < sum(walls)
    # a second decode with fewer streams than slices: the empty slices stay idle
    assert multi.decode(caps[:3]) == sum(single.eti_count(b) for b in range(3))
    assert [multi.slice_of(b)[0] for b in range(3)] == [0, 1, 2]
    for b in range(3):
        assert np.array_equal(multi.eti(b), single.eti(b))
    # settings reach every slice
    multi.set_subchannels([2])
    single.set_subchannels([2])
    multi.decode(caps[:9])
    single.decode(caps[:9])
    for b in range(9):
        assert np.array_equal(multi.eti(b), single.eti(b)), b
    multi.close()
    single.close()


def test_multi_rejects_bad_devices():
    with pytest.raises(dab.DabhipError):
        dab.Multi([0, 99])
    with pytest.raises(dab.DabhipError):
        dab.Multi([])


def test_cli_devices_flag_equals_single_device_output(tmp_path):
    """`dab2eti-hip --devices 0,0,0 f0 .. f4` (three slices on GPU 0) writes the bytes of `dab2eti-hip f0 .. f4`."""
    exe = os.path.join(os.path.dirname(dab.LIB_PATH), "dab2eti-hip")
    names = []
    for i, iq in enumerate(_captures(5, ntf=18)):
        p = tmp_path / ("cap%d.cu8" % i)
        iq.tofile(p)
        names.append(str(p))
    one = subprocess.run([exe] + names, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    three = subprocess.run([exe, "--devices", "0,0,0"] + names, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    assert len(one.stdout) > 5 * 8 * dab.ETI_BYTES and one.stdout == three.stdout
    assert b"(device 0)" in three.stderr
    rng = subprocess.run([exe, "--devices", "0-0"] + names[:2], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
    assert rng.stdout == subprocess.run([exe] + names[:2], stdout=subprocess.PIPE, check=True).stdout
    bad = subprocess.run([exe, "--devices", "0,x"] + names[:1], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert bad.returncode == 1


def test_eti_consumer_pipe_against_the_reference_eti2mpa(tmp_path):
    """`dab2eti-hip capture.cu8 | eti2mpa N` with the REFERENCE's own eti2mpa.c (oracle/_ref/eti2mpa_ref, compiled unmodified) as the
    consumer of GPU-made ETI: its output equals our eti2mpa's and the payload the modulator sent, for three SubChIds."""
    ref_exe = os.path.join(os.path.dirname(os.path.dirname(dab.LIB_PATH)), "oracle", "_ref", "eti2mpa_ref")
    assert os.path.exists(ref_exe), "oracle/_ref/eti2mpa_ref missing: __graft_entry__.build() makes it where the reference is present"
    here = os.path.dirname(dab.LIB_PATH)
    cfg = dab.synth_preset(0, seed=77, cif_count0=4990)            # the 12 sub-channel mix; the CIF counter wraps inside the capture
    cap = tmp_path / "cap.cu8"
    dab.synth_generate(cfg, 22).tofile(cap)
    eti = subprocess.run([os.path.join(here, "dab2eti-hip"), str(cap)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True).stdout
    assert len(eti) == 4 * (22 - 15) * dab.ETI_BYTES
    frames = np.frombuffer(eti, np.uint8).reshape(-1, dab.ETI_BYTES)
    eti_file = tmp_path / "out.eti"             # a regular file: the reference's single read() per frame (eti2mpa.c:32) comes back short on a pipe
    eti_file.write_bytes(eti)
    fib_index = {dab.synth_fibs(cfg, c).tobytes(): c for c in range(4 * 22)}
    for slot, scid in ((0, 1), (5, 6), (10, 11)):                  # UEP 128k, UEP 192k, the EEP 2-A 8 kbit/s special case
        with open(eti_file, "rb") as f:
            ref = subprocess.run([ref_exe, str(scid)], stdin=f, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        ours = subprocess.run([os.path.join(here, "eti2mpa"), str(scid)], input=eti, stdout=subprocess.PIPE, stderr=subprocess.PIPE, check=True)
        assert ref.returncode == 1 and b"Extracting channel: SCID=%d" % scid in ref.stderr      # eti2mpa.c:33-36,49
        assert ref.stdout == ours.stdout and len(ref.stdout) > 0
        sent = b"".join(dab.synth_payload(cfg, fib_index[f[12 + 4 * (f[5] & 0x7f):][:96].tobytes()], slot).tobytes() for f in frames)
        # STL counts 64-bit words: the 8 kbit/s sub-channel's 24 payload bytes sit in 3 words exactly, the others likewise
        assert ref.stdout == sent, scid


def test_bench_in_process_mode_and_config3_tool_at_reduced_size():
    """`bench.py --gpus N --in-process` (dabhip_multi over the node's devices from one process) and tools/config3_one_gpu.py, both with all
    slices on GPU 0 and a small batch: one JSON line each, every slice its frames, sampled streams equal to the oracle."""
    import json
    import sys
    root = os.path.dirname(os.path.dirname(dab.LIB_PATH))
    env = dict(os.environ, DABHIP_BENCH_ONE_DEVICE="1")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--in-process", "--streams", "8", "--tfs", "20", "--steps", "2", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 3 and d["config"]["eti_frames_per_step"] == 3 * 8 * 4 * (20 - 15)
    assert [s["eti_frames"] for s in d["slices"]] == [8 * 20] * 3 and d["value"] > 0
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "config3_one_gpu.py"), "--slices", "4", "--streams-per-slice", "6", "--tfs", "19", "--steps", "1",
                        "--oracle-streams", "3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, (r.stdout[-500:], r.stderr[-2000:])
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["eti_frames"] == 24 * 16 and d["all_streams_full_count"] and d["oracle_byte_equal"].startswith("3 of 3") and d["sharding_rule_ok"]


def test_plan_before_the_decode_uneven_batch_with_device_pointers():
    """the dealing rule must be usable BEFORE the first decode (an on_device caller has to put stream b's samples on the device
    of its slice).  10 streams on 4 slices = 3, 3, 2, 2: dabhip_multi_plan says so up front, the samples go where it says (device buffers),
    the decode agrees (slice_of) and the bytes are the single engine's."""
    caps = _captures(10)
    multi = dab.Multi([0] * 4)
    plan = [multi.plan(len(caps), b) for b in range(len(caps))]
    assert [p[0] for p in plan] == [0, 0, 0, 1, 1, 1, 2, 2, 3, 3]
    assert [multi.plan(2048, s)[0] for s in (0, 511, 512, 2047)] == [0, 0, 1, 3] and multi.plan(3, 2)[0] == 2
    for rank in range(4):
        assert [b for b in range(10) if plan[b][0] == rank] == list(shard.shard_streams(10, 4, rank))
    bufs = []
    for b, iq in enumerate(caps):
        buf = dab.DeviceBuffer(iq.size, device=plan[b][1])       # on the device the plan names
        buf.upload(iq)
        bufs.append(buf)
    total = multi.decode_device([x.ptr for x in bufs], [iq.size for iq in caps])
    single = dab.Engine(0)
    assert total == single.decode(caps)
    for b in range(len(caps)):
        assert multi.slice_of(b) == plan[b]
        assert np.array_equal(multi.eti(b), single.eti(b))
    # host placement: reported per slice; on a one-socket box nothing is bound, on a two-socket one the chunks are disjoint and on the GPU's node
    seen = set()
    for sl in range(4):
        cpus, node = multi.slice_cpus(sl)
        assert not (seen & set(cpus)), "slices share CPUs"
        seen |= set(cpus)
    for x in bufs:
        x.free()
    single.close()
    multi.close()


def test_two_distinct_devices_get_their_own_kernel_attributes():
    """The > 64 KB dynamic-LDS attribute of the scan / guard kernels is per device: a second device must work too."""
    if dab.lib().dabhip_device_count() < 2:
        pytest.skip("one GPU on this box")
    caps = _captures(4)
    multi = dab.Multi([0, 1])
    single = dab.Engine(0)
    assert multi.decode(caps) == single.decode(caps)
    for b in range(4):
        assert np.array_equal(multi.eti(b), single.eti(b))
    single.close()
    multi.close()
