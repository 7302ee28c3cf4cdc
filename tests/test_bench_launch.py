"""bench.py's multi-rank launch, sharding and book-keeping without a GPU (`--dry-run`): `--gpus N` by itself must start N
ranks (the driver's N = 1 command is plain `python bench.py --gpus 1 ...`; SCALE runs may or may not come through
torch.distributed.run), each with its own 256 streams, and rank 0 prints ONE JSON line with n_gpus = N."""
import json
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _one_json_line(out):
    lines = [l for l in out.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


def _check(res, n, steps):
    assert res["n_gpus"] == n and res["steps"] == steps and res["scaling"] == "weak" and res["dry_run"] is True
    ranks = res["ranks"]
    assert [r["rank"] for r in ranks] == list(range(n))
    for r in ranks:                                              # stream s on rank s // 256, seeds 2000 + s
        assert (r["first_stream"], r["last_stream"], r["streams"]) == (256 * r["rank"], 256 * r["rank"] + 255, 256)
        assert r["first_seed"] == 2000 + r["first_stream"]
        assert r["eti_frames_per_step"] == 256 * 4 * (64 - 15)
    assert res["config"]["eti_frames_per_step"] == n * 256 * 196
    slowest = max(r["elapsed_s"] for r in ranks)                 # MAX over ranks, whole-job frames
    assert abs(res["value"] - n * 256 * 196 * steps / slowest) < 1e-6 * res["value"]
    assert abs(res["ms_per_step"] - 1e3 * slowest / steps) < 1e-6
    # every rank records the device it decoded on; the line says how many distinct ones there were
    assert [r["device"]["pci_bus_id"] for r in ranks] == res["devices"]["pci_bus_ids"]
    assert res["devices"]["distinct_pci_bus_ids"] == len(set(res["devices"]["pci_bus_ids"]))
    assert res["devices"]["distinct_pci_bus_ids"] == n or res["devices"]["rehearsal_on_one_device"]
    if n > 1:
        assert ranks[-1]["elapsed_s"] > ranks[0]["elapsed_s"]   # the dry run makes higher ranks slower on purpose
        assert all(r["host_threads"] != "auto" for r in ranks)  # per-rank host pool capped (cores / ranks)


def test_self_launch_two_and_four_ranks():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    for n in (2, 4):
        out = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--dry-run", "--steps", "3"], env=env, stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True, timeout=120, check=True).stdout
        _check(_one_json_line(out), n, 3)


def test_self_launch_eight_ranks_each_sees_its_own_gpu_and_its_share_of_the_cpus():
    """configs[3]'s launch (8 ranks x 256 streams) without a GPU: every rank is started with ROCR_VISIBLE_DEVICES = its GPU before it exists
    (and uses device 0 of that view), and with a host pool that is its share of the CPUs the container really has -- affinity mask and CFS
    quota, not os.cpu_count() (the GPU boxes report 256 hardware threads and grant 16 CPUs)."""
    sys.path.insert(0, ROOT)
    from dabtools_amd import shard
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES",
                                                             "CUDA_VISIBLE_DEVICES", "DABHIP_HOST_THREADS", "DABHIP_CPUS", "DABHIP_BENCH_ONE_DEVICE")}

    def run(extra):
        out = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-run", "--steps", "2"], env=dict(base, **extra), stdout=subprocess.PIPE,
                             stderr=subprocess.PIPE, text=True, timeout=180, check=True).stdout
        res = _one_json_line(out)
        _check(res, 8, 2)
        return res["ranks"]

    ranks = run({})
    assert [r["visible_devices"] for r in ranks] == [str(i) for i in range(8)]
    assert all(r["host_threads"] == str(shard.host_threads_per_rank(8)) for r in ranks)
    ranks = run({"DABHIP_CPUS": "16"})                               # the GPU boxes' grant: 16 CPUs for 8 ranks -> the floor of 2 threads each
    assert all(r["host_threads"] == "2" for r in ranks)
    ranks = run({"DABHIP_CPUS": "256"})
    assert all(r["host_threads"] == "16" for r in ranks)
    ranks = run({"ROCR_VISIBLE_DEVICES": "4,5,6,7,0,1,2,3"})         # the launcher was itself given a list: rank r takes its r-th entry
    assert [r["visible_devices"] for r in ranks] == ["4", "5", "6", "7", "0", "1", "2", "3"]
    ranks = run({"HIP_VISIBLE_DEVICES": "0,1,2,3,4,5,6,7"})          # the caller selects through HIP's own list: not combined with ROCR's
    assert all(r["visible_devices"] is None for r in ranks)
    a, q = shard.cpu_budget()
    assert 1 <= shard.usable_cpus() <= a and (q is None or q >= 1)
    if hasattr(os, "sched_setaffinity"):                             # a launcher's taskset shrinks the budget
        code = "import os,sys; sys.path.insert(0, %r); os.sched_setaffinity(0, {sorted(os.sched_getaffinity(0))[0]}); from dabtools_amd import shard; print(shard.usable_cpus())" % ROOT
        assert subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, text=True, check=True, env=base).stdout.strip() == "1"


def test_ranks_that_share_a_device_are_refused_unless_declared_a_rehearsal():
    """An N-rank line is an N-GPU figure only if the N ranks sat on N distinct GPUs: rank 0 compares the PCI bus ids the ranks recorded and prints NO line
    (exit code non-zero) when they do not differ -- except under DABHIP_BENCH_ONE_DEVICE=1, the declared one-GPU rehearsal, whose line then says so
    itself.  tools/scale_sweep.py builds the 1/2/4/8 table from such lines and labels every row."""
    base = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "DABHIP_BENCH_ONE_DEVICE", "DABHIP_BENCH_DRY_BUS_ID")}
    cmd = [sys.executable, BENCH, "--gpus", "4", "--dry-run", "--steps", "2"]
    p = subprocess.run(cmd, env=dict(base, DABHIP_BENCH_DRY_BUS_ID="0000:c1:00.0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode != 0 and "4 ranks decoded on 1 distinct devices" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{")], p.stdout
    p = subprocess.run(cmd, env=dict(base, DABHIP_BENCH_ONE_DEVICE="1"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120, check=True)
    res = _one_json_line(p.stdout)
    assert res["devices"]["distinct_pci_bus_ids"] == 1 and res["devices"]["rehearsal_on_one_device"] is True and "NOT a 4-GPU figure" in res["devices"]["note"]
    # the sweep: 1, 2 and 8 ranks, dry; then the declared rehearsal
    sweep = [sys.executable, os.path.join(ROOT, "tools", "scale_sweep.py"), "--dry-run", "--gpus", "1,2,8", "--steps", "2", "--streams", "4", "--tfs", "20"]
    p = subprocess.run(sweep, env=base, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, check=True)
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert [r["n_gpus"] for r in d["rows"]] == [1, 2, 8] and all(r["kind"] == "measured" and r["distinct_devices"] == r["n_gpus"] for r in d["rows"])
    assert not d["any_rehearsal"] and d["rows"][0]["efficiency_vs_n1"] == 1.0 and d["rows"][2]["efficiency_vs_n1"] is not None
    assert d["last_driver_bench"] is None or d["last_driver_bench"]["file"].startswith("BENCH_r")
    p = subprocess.run(sweep + ["--one-device"], env=base, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, check=True)
    d = json.loads(p.stdout.strip().splitlines()[-1])
    assert d["any_rehearsal"] and [r["kind"] for r in d["rows"]] == ["measured", "REHEARSAL (one GPU)", "REHEARSAL (one GPU)"]
    assert d["rows"][2]["efficiency_vs_n1"] is None and "REHEARSAL" in p.stderr


def test_a_dying_rank_ends_the_launch_with_an_error():
    """One of four ranks exits before the first barrier (test knob DABHIP_BENCH_DIE_RANK): the launcher must not hang at that barrier -- it ends the
    other ranks and returns non-zero, without a result line."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["DABHIP_BENCH_DIE_RANK"] = "2"
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--dry-run", "--steps", "3"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=120)
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.strip().startswith("{")], p.stdout


def test_single_rank_unchanged():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--dry-run", "--steps", "2", "--warmup", "0"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=120, check=True).stdout
    _check(_one_json_line(out), 1, 2)


def test_torchrun_launch_two_ranks_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), BENCH, "--gpus", "2", "--dry-run", "--steps", "2"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=300, check=True).stdout
    _check(_one_json_line(out), 2, 2)


def test_cpu_baseline_tool_small_sample(tmp_path):
    """tools/cpu_baseline.py (the child bench.py starts for its cpu_baseline object) on a small capture."""
    sys.path.insert(0, ROOT)
    import dabtools_amd as dab
    iq = dab.synth_generate(dab.synth_preset(1, seed=5), 18)
    path = tmp_path / "s0.cu8"
    iq.tofile(path)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--tfs", "18", "--cores", "2", "--iq", str(path)],
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300, check=True).stdout
    res = json.loads(out.strip().splitlines()[-1])
    assert res["kind"] == "port" and res["cores"] == 1 and res["value"] > 0 and res["oracle_dft2048_us"] > 0
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libdabref.so")):
        for key in ("reference_backend_scalar", "reference_backend_sse"):
            assert res[key]["kind"] == "reference" and res[key]["value"] > 0 and res[key]["all_cores"]["cores"] == 2
        assert res["reference_backend_sse"]["value"] > res["reference_backend_scalar"]["value"]


def test_roofline_objects_are_computable_from_the_tracked_profile():
    """bench.py takes every roofline input it cannot measure live (VALU instructions per trellis step, effective clocks, HBM bytes from the PMC
    passes, LDS conflict rate) from the tracked profile of the round -- and fails loudly without it.  The file is there, every cell the
    rooflines read exists, and each `frac` follows from the named cells."""
    import argparse
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert os.path.exists(bench.PROFILE_PMC), "profiles/rNN_pmc_summary.csv is not tracked: run tools/refresh_profiles.sh on a GPU box"
    prof = bench.load_profile()
    args = argparse.Namespace(streams=256, tfs=64, subchannels="", soft=False, two_kernel_ofdm=False, no_parity_guard=False)
    frames, ntf = 256 * 196, 256 * 62
    stage = {"viterbi": 5.0, "fft": 4.4}
    out = bench.rooflines(prof, args, 1, frames, ntf, stage, (30, 30 * 3968, 35.0), {"fill": 6000.0, "copy": 5200.0, "k2_mix": 5500.0})
    assert set(out) == {"roofline", "roofline_viterbi", "roofline_ofdm_fused"}
    r = out["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12 and 0.95 < r["traffic"] / (1556480 * 3968) < 1.1
    v = out["roofline_viterbi"]
    insts = prof.cell("clk", "viterbi_fused_kernel<1>", "SQ_INSTS_VALU") / (prof.meta("full_decodes") * prof.meta("viterbi_wave_steps_per_decode")
                                                                               + prof.meta("viterbi_wave_steps_setup"))
    clock = prof.cell("clk", "viterbi_fused_kernel<1>", "GRBM_GUI_ACTIVE") / 8 / prof.cell("clk", "viterbi_fused_kernel<1>", "DURATION_NS")
    assert 100 < insts < 140 and 1.5 < clock < 2.5
    cycles = 2 * 64 + 4 * (insts - 64)
    assert abs(v["frac"] - cycles * bench.wave_steps(frames, 0) / 5.0e-3 / (1024 * clock * 1e9)) < 1e-9
    assert v["survivor_traffic"]["x_stage_io"] > 10
    assert v["frac_if_every_instruction_took_2_cycles"] < v["frac"] and 0.3 < v["frac_from_counters"]["value"] < 1.0
    inside = out["roofline"]["in_timed_step"]               # the kernels the timed step runs, inside the object the driver's record keeps
    assert inside["ofdm_demap_kernel"]["frac"] == out["roofline_ofdm_fused"]["frac"] and inside["viterbi_fused_kernel"]["frac"] == v["frac"]
    assert inside["ofdm_demap_kernel"]["bound"] == inside["viterbi_fused_kernel"]["bound"] == "valu issue"
    f = out["roofline_ofdm_fused"]
    assert f["bound"] == "valu issue" and 5 < f["lds"]["bank_conflict_pct"] < 40 and f["hbm"]["frac"] < 0.3
    # ONE number: the instruction mix of the symbol loop priced in issue cycles -- no bracket -- and the counters' own busy
    # fraction beside it; the two are independent and have to tell the same story
    import json
    assert "frac_if_all" not in json.dumps(f)
    mix = json.load(open(bench.PROFILE_FUSED_MIX))
    insts = f["valu_issue"]["insts_per_wave_per_transform"]
    cyc = mix["issue_cycles_per_unit"] + max(0.0, insts - mix["valu_per_unit"]) * mix["remainder_cycles_per_valu"]
    assert abs(f["valu_issue"]["issue_cycles_per_wave_per_transform"] - cyc) < 1e-6 and 250 < mix["valu_per_unit"] < insts < 400
    assert 0.4 < f["frac_from_counters"]["value"] < 1.0
    # (stage time of this call is made up: compare the model with the counters on the profiled run's own time instead)
    t_prof = prof.cell("clk", "ofdm_demap_kernel<false>", "DURATION_NS") * 1e-9
    tr_prof = prof.meta("full_decodes") * prof.meta("ofdm_transforms_per_decode") + prof.meta("ofdm_transforms_setup")
    clock = prof.cell("clk", "ofdm_demap_kernel<false>", "GRBM_GUI_ACTIVE") / 8 / prof.cell("clk", "ofdm_demap_kernel<false>", "DURATION_NS")
    model = 4.0 * cyc * tr_prof / t_prof / 1e9 / (1024 * clock)
    assert abs(model - f["frac_from_counters"]["value"]) < 0.12, (model, f["frac_from_counters"]["value"])
    # the mix file belongs to THIS tree's kernel sources
    import hashlib
    src = b"".join(open(os.path.join(ROOT, "dabtools_amd", "csrc", n), "rb").read() for n in ("k_fused.hip", "fft_core.hpp", "device_types.hpp"))
    assert mix["source_sha256"] == hashlib.sha256(src).hexdigest(), "profiles/r06_fused_isa_mix.json is stale: run tools/fused_isa_mix.sh"
    # a missing profile, or a missing cell, is an error that says what to do
    import pytest
    with pytest.raises(SystemExit, match="refresh_profiles"):
        prof.cell("clk", "no_such_kernel", "SQ_INSTS_VALU")


def test_decision_audit_profile_belongs_to_this_trees_kernels():
    """profiles/r06_decision_audit.json -- the ground both guard levels stand on -- names the kernel the default
    decode runs, was measured on THIS tree's sources (tools/decision_audit.py hashes them; re-run it on a GPU box after touching any), covers >= 10^10
    decisions with no disagreement outside EITHER level's band and none with the guard on at either level; the worst measured errors lie below the proven
    bound (tools/fft_error_bound.py), the proven level's constants are >= that bound, the measured level keeps a margin of >= 4 over the worst errors, and
    the shipping build left the audit build's bits and list counts in every case."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import decision_audit
    import fft_error_bound
    d = json.load(open(os.path.join(ROOT, "profiles", "r06_decision_audit.json")))
    assert "ofdm_demap_kernel" in d["kernel"] and d["sources"] == list(decision_audit.AUDITED_SOURCES)
    assert d["source_sha256"] == decision_audit.source_sha(), "profiles/r06_decision_audit.json is stale: run tools/decision_audit.py --channels 300 on a GPU box"
    assert d["total_decisions"] >= 1e10 and d["total_disagree_outside_band"] == 0 and d["total_disagree_guard_on"] == 0
    assert d["proven_level"]["total_disagree_outside_band"] == 0 and d["proven_level"]["total_disagree_guard_on"] == 0
    assert d["margin_kGuardC_over_worst"] >= 4 and d["margin_kGuardProd_over_worst"] >= 4
    assert d["shipping_kernel_same_bits_in_every_case"] and d["shipping_kernel_same_list_count_in_every_case"]
    bound = fft_error_bound.constants()
    assert d["proven_bound"]["bin_err_over_l2"] == bound["bin_bound"] and d["proven_bound"]["product_rounding_over_l2l2"] == bound["prod_bound"]
    assert d["worst_measured_below_proven_bound"] and d["worst_max_bin_err_over_sqrt_energy"] <= bound["bin_bound"] and d["worst_max_residual_product_err"] <= bound["prod_bound"]
    assert d["guard_constants"]["proven"]["kGuardCProven"] >= bound["bin_bound"] and d["guard_constants"]["proven"]["kGuardProdProven"] >= bound["prod_bound"]
    assert any(c["channel"] != "ideal" for c in d["cases"])


def test_proven_guard_constants_cover_the_rigorous_bound():
    """device_types.hpp's kGuardCProven / kGuardProdProven (read through the library: no GPU needed) are >= the forward-error bound tools/fft_error_bound.py
    derives for the transform and product as the kernels write them, and that script's fp32 model of the same operation order stays below the bound on
    adversarial inputs (single tones, clipped tones, cancelling halves, OFDM symbols) -- while exceeding the MEASURED level's constant on some of them,
    which is why the proven level exists."""
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import dabtools_amd as dab
    import fft_error_bound
    c = fft_error_bound.constants()
    assert 6.4e-5 < c["bin_bound"] < 6.6e-5 and 1.19e-7 < c["prod_bound"] < 1.2e-7
    assert dab.guard_constants(2)[0] >= c["bin_bound"] and dab.guard_constants(2)[1] >= c["prod_bound"]
    assert dab.guard_constants(1)[0] < c["bin_bound"]                       # the measured level does NOT claim the bound
    # the proven level lists PER BIN: constant x guard_bin_scale(k) covers bin k's own bound (its stage terms by its index digits), for every bin
    bounds = np.array([fft_error_bound.bin_bound(k, c) for k in range(2048)])
    scale = np.array([dab.guard_bin_scale(k) for k in range(2048)])
    assert (dab.guard_constants(2)[0] * scale >= bounds).all() and scale.max() <= 1.00001 and 0.33 < scale.min() < 0.34
    inband = list(range(1, 769)) + list(range(1280, 2048))
    assert 0.78 < scale[inband].mean() < 0.80
    assert dab.guard_default_level() in (1, 2)
    rng = np.random.default_rng(6)
    worst = 0.0
    for name, x in fft_error_bound.adversarial_inputs(rng):
        norm = float(np.sqrt(np.sum(np.abs(x) ** 2)))
        if norm == 0:
            continue
        per_bin = np.abs(fft_error_bound.fft2048_model(x) - np.fft.fft(x)) / norm
        err = float(per_bin.max())
        assert err <= c["bin_bound"] and (per_bin <= bounds).all(), (name, err)        # every bin below ITS OWN bound
        worst = max(worst, err)
    assert worst > dab.guard_constants(1)[0]                                # a clipped tone: 5.3e-6 |x|_2 > 5e-6
