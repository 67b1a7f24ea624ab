"""bench.py pieces that do not need a GPU: the synthetic video is a pure function of (seed, frame) -- what lets every rank
build only its shard -- and the chunk plan of the multi-GPU schedule covers the video."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_synth_video_is_a_function_of_seed_and_frame():
    from bench import synth_video
    whole = synth_video(0, 9, seed=0, h=72, w=128)
    part = synth_video(4, 9, seed=0, h=72, w=128)
    assert whole.dtype == torch.uint8 and whole.shape == (9, 3, 72, 128)
    assert torch.equal(whole[4:], part)
    assert not torch.equal(whole[0], whole[1])                    # objects move
    other = synth_video(0, 2, seed=1, h=72, w=128)
    assert not torch.equal(other[0], whole[0])
    # spatial structure (not i.i.d. pixels): neighbouring pixels are strongly correlated
    f = whole[0].float()
    a, b = f[:, :, :-1].flatten(), f[:, :, 1:].flatten()
    assert float(torch.corrcoef(torch.stack([a, b]))[0, 1]) > 0.8


def test_bench_shards_reassemble_the_video():
    from bench import synth_video
    from mdqe_cvpr2023_amd import sharding
    L, T, stride, win = 50, 4, 1, 10
    plan = sharding.chunk_plan(L, T, stride, win)
    video = synth_video(0, L, seed=0, h=40, w=64)
    seen = set()
    for world in (1, 2, 3):
        for rank in range(world):
            for g in sharding.owned_chunks(plan, world, rank):
                clips, f0, f1 = plan[g]
                assert torch.equal(synth_video(f0, f1, seed=0, h=40, w=64), video[f0:f1])
                if world == 3:
                    seen.update(c[0] for c in clips)
    assert seen == {c[0] for c in __import__("mdqe_cvpr2023_amd.meta_arch", fromlist=["MDQE"]).MDQE.clip_schedule(L, T, stride)}


def _bench(args, env_extra, timeout=300):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)


def test_bench_gpus_n_starts_n_ranks_itself():
    """`python bench.py --gpus 2` (no torchrun around it) must run TWO ranks: the launcher spawns them as a child process tree before any
    GPU call, the ranks count each other by all-reduce and rank 0 reports the count (MDQE_BENCH_RANK_PROBE=1 stops there: no GPU here).
    Reference: train_net.py:264-271 (`launch(main, args.num_gpus, ...)`)."""
    import json
    for n in (2, 3):
        r = _bench(["--gpus", str(n)], {"MDQE_BENCH_RANK_PROBE": "1"})
        assert r.returncode == 0, r.stderr[-2000:]
        lines = [l for l in r.stdout.splitlines() if l.strip()]
        assert len(lines) == 1 and lines[0].startswith("{"), r.stdout      # ONE line on stdout: gloo's / RCCL's own chatter goes to stderr
        d = json.loads(lines[0])
        assert d["n_gpus"] == n and d["ranks_seen"] == n


def test_bench_never_reports_the_wrong_world():
    """Loud failures instead of a line for the wrong N: fewer visible GPUs than --gpus (no rehearsal hook), and a wrapper whose world
    size differs from --gpus."""
    r = _bench(["--gpus", "2"], {})
    assert r.returncode != 0 and "{" not in r.stdout and "visible" in r.stderr
    r = _bench(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "MDQE_BENCH_RANK_PROBE": "1"})
    assert r.returncode != 0 and "{" not in r.stdout and "refusing" in r.stderr
    r = _bench(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "MDQE_BENCH_RANK_PROBE": "1"})
    assert r.returncode != 0 and "{" not in r.stdout


def test_host_cpu_reports_model_and_physical_cores():
    from bench import host_cpu
    model, phys, logical = host_cpu()
    assert isinstance(model, str) and model and 1 <= phys <= logical


def test_cpu_baseline_runs_and_reports_cores_and_model():
    """bench.cpu_baseline on configs[0] (4 synthetic 360p frames through the oracle, both schedules, two thread counts): the leg the
    driver's bench line carries -- it must not be able to break unnoticed (it runs last in bench.py, after minutes of GPU work)."""
    from bench import cpu_baseline, synth_video
    from mdqe_cvpr2023_amd.config import PRESETS
    from mdqe_cvpr2023_amd.params import random_state
    cfg = PRESETS["R50_ovis_360"]
    sd = random_state(cfg, seed=0, remove_zero_init_trap=True)
    r = cpu_baseline(cfg, sd, synth_video(0, 4, seed=0))
    assert r["kind"] == "port" and r["unit"] == "frames/s" and r["value"] > 0 and 1 <= r["cores"] <= r["physical_cores"] <= r["logical_cpus"]
    assert r["cpu_model"] and len(r["threads_tried"]) >= 1 and abs(max(r["threads_tried"].values()) - r["value"]) < 1e-3
    assert 0 < r["as_reference_value"] < r["value"]              # the reference's window recompute costs three more frame passes
    assert len(r["threads_tried"]) == 1 and len(r["sample"]) <= 140      # a bounded sweep; the line quotes the sample as it is


def test_a_failing_rank_ends_the_whole_run_quickly_and_loudly():
    """gloo rehearsal of the failure path of the first real multi-GPU run (no GPU needed: MDQE_BENCH_RANK_PROBE): rank 1 raises before
    its first collective -> the parent exits non-zero within seconds (torch.distributed.run ends the other ranks), with the failing
    rank's message on stderr and no JSON line on stdout."""
    import time
    t0 = time.time()
    r = _bench(["--gpus", "2"], {"MDQE_BENCH_RANK_PROBE": "1", "MDQE_BENCH_FAIL_RANK": "1", "MDQE_BENCH_COLLECTIVE_TIMEOUT_S": "30",
                                 "MDQE_COLLECTIVE_TIMEOUT_S": "30"}, timeout=240)
    assert r.returncode != 0 and "{" not in r.stdout
    assert "rank 1 fails on purpose" in r.stderr
    assert time.time() - t0 < 120


def test_a_hung_rank_times_the_collective_out():
    """A rank that never arrives (sleeps before its first collective): the others raise after the process group's timeout instead of
    waiting for the driver's kill; the parent exits non-zero."""
    import time
    t0 = time.time()
    r = _bench(["--gpus", "2"], {"MDQE_BENCH_RANK_PROBE": "1", "MDQE_BENCH_HANG_RANK": "1", "MDQE_COLLECTIVE_TIMEOUT_S": "8"}, timeout=240)
    assert r.returncode != 0 and "{" not in r.stdout
    assert "hangs on purpose" in r.stderr
    assert time.time() - t0 < 150


def test_the_launching_parent_never_touches_the_gpu_runtime():
    """`python bench.py --gpus N` is a parent that only starts a child process tree: no torch.cuda call of any kind before that
    (let the ranks refuse)."""
    import ast
    import inspect
    import bench
    src = inspect.getsource(bench.spawn_ranks)
    assert "torch.cuda" not in src and "device_count" not in src.replace("not even a device count", "")
    main_src = inspect.getsource(bench.main)
    head = main_src[:main_src.index("sys.exit(spawn_ranks(")]
    assert "torch.cuda" not in "\n".join(l for l in head.splitlines() if not l.strip().startswith("#"))
    ast.parse(main_src.lstrip())


def test_same_output_compares_bit_for_bit():
    from bench import same_output
    a = {"pred_labels": [1, 2], "pred_scores": [0.5, 0.25], "pred_masks": [torch.zeros(2, 4, 4, dtype=torch.bool), torch.ones(2, 4, 4, dtype=torch.bool)]}
    b = {k: ([m.clone() for m in v] if k == "pred_masks" else list(v)) for k, v in a.items()}
    assert same_output(a, b) == (True, "")
    b["pred_masks"][1][0, 0, 0] = False
    assert same_output(a, b)[0] is False and same_output(a, None)[0] is False
    b = dict(a, pred_scores=[0.5, 0.2500001])
    assert same_output(a, b)[0] is False


def test_soft_deadline_prints_what_it_has_and_leaves_with_exit_code_zero():
    """bench.Deadline guards the optional halo-exchange A/B of a multi-GPU run: if the guarded region does not finish in time, the
    callback runs (rank 0 prints the line it already has) and the process leaves with exit code 0 -- the headline is never lost to a
    hang in an extra; a region that finishes in time cancels the timer."""
    import subprocess
    import textwrap
    script = textwrap.dedent('''
        import sys, time
        sys.path.insert(0, %r)
        from bench import Deadline
        with Deadline(30.0, lambda: print("NEVER")):
            pass                                            # finishes in time: timer cancelled
        print("FIRST", flush=True)
        with Deadline(0.3, lambda: print('{"value": 1.0, "halo_exchange": {"error": "did not finish"}}', flush=True)):
            time.sleep(60)                                  # a collective that never returns
        print("UNREACHABLE")
    ''' % ROOT)
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout.splitlines()
    assert out[0] == "FIRST" and out[1].startswith('{"value": 1.0') and "UNREACHABLE" not in r.stdout and "NEVER" not in r.stdout


def _fat_full_line():
    """A full result object with every leg on and paragraphs in every free-text field (what round 5's 25.7 KB line looked like)."""
    prose = "x" * 3000
    roof = {"bound": "mfma", "kernel": "gemm_nt_f32_k16_kernel " + prose, "achieved": 93.7238391576, "peak": 157.3, "unit": "TFLOP/s", "frac": 0.5958286024,
            "traffic": None, "traffic_ref": "profiles/r05_pmc_gemm_summary.txt " + prose, "launches": 312, "launches_timed": 312, "launches_total": 1560,
            "avg_launch_us": 558.5852541710035, "note": prose}
    msda = {"bound": "hbm", "kernel": prose, "achieved": 1.1229757790084047, "peak": 8.0, "unit": "TB/s", "frac": 0.1403719723760506, "frac_isolated": 0.157,
            "avg_launch_us": 314.3, "algorithmic_MB_per_launch": 352.95, "traffic": None, "traffic_ref": "profiles/x.txt", "bytes": prose,
            "decoder_box": {"frac": 0.034, "frac_isolated": 0.205, "bytes": prose}, "decoder_temporal": {"frac": 0.056, "frac_isolated": 0.139, "bytes": prose}}
    side = {"value": 306.9, "ms_per_step": 195.5, "frames_per_step": 60, "steps": 4, "workload": prose, "roofline": dict(roof), "roofline_msda": dict(msda)}
    rl = {"world": 8, "ms_per_step": 164.4, "root_ms_per_step": 164.4, "other_rank_ms_per_step": 157.0, "predicted_efficiency": 0.884, "verified": True,
          "what": prose, "chunk_frames_per_round": [69, 34, [0, 20, 20, 20, 19, 19, 19, 19]], "tracker_native_ms_per_step": {"updates_per_step": 1197}}
    full = {"metric": "frames/sec (eval-only) R50 OVIS 360p 4-frame clip", "value": 826.123456789, "value_median": 830.8, "unit": "frames/s", "n_gpus": 1,
            "steps": 20, "warmup": 5, "ms_per_step": 145.26, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic", "value_is": prose,
            "config": {"workload": "R50_ovis_360 eval-only, H2D included: " + prose, "frames_per_gpu": 120, "clips_per_step": 118, "instances_out": 10,
                       "tracked_instances": 7, "parallelism": "single GPU", "output": prose},
            "roofline": roof, "roofline_isolated": dict(roof), "roofline_msda": msda,
            "clip_stage": {"frac": 0.385, "tflops": 60.6, "ms_per_step": 24.2, "clips": 117, "what": prose},
            "cpu_baseline": {"value": 0.5763, "unit": "frames/s", "cores": 32, "kind": "port", "sample": prose, "cpu_model": "AMD EPYC 9575F 64-Core Processor",
                             "physical_cores": 128, "as_reference_value": 0.34, "threads_tried": {"16": 0.53, "32": 0.57}},
            "config_R50_ovis_720": side, "config_swinl_ovis": dict(side), "root_load": rl, "root_load_halo": dict(rl),
            "scaling_breakdown": {"per_rank_ms": {k: [1.0] * 8 for k in ("compute", "pack", "gather_wait")}, "replay_exposed_ms": 3.0, "gather_ms": 1.0,
                                  "halo_frac": 0.07, "rounds": 3, "what": prose},
            "halo_exchange": {"value": 800.0, "ms_per_step": 150.0, "verified": True, "halo_frac": 0.0, "what": prose},
            "bench_wall_s": 101.2}
    for k in ("fast_mode", "autocast_f16", "reference_precision_map", "stream_mode", "frames_resident", "late_masks", "init_reference"):
        full[k] = {"value": 1072.9996920898907, "unit": "frames/s", "ms_per_step": 111.8, "what": prose}
    return full


def test_compact_line_is_bounded_whatever_the_legs_say():
    """the 25.7 KB line came back `parsed: null`.  The printed line is <= 4096 bytes with EVERY leg on and paragraphs in every
    free-text field, keeps the driver's contract keys, `roofline` and `cpu_baseline`, and carries the extras as numbers only."""
    import json
    from bench import LINE_LIMIT, compact_line
    full = _fat_full_line()
    text = compact_line(full, "gpurun_out/bench_extras.json")
    assert LINE_LIMIT == 4096 and len(text) <= LINE_LIMIT and "\n" not in text
    d = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline", "cpu_baseline", "roofline_msda", "extras"):
        assert k in d, k
    assert d["value"] == 826.123 and len(d["config"]["workload"]) <= 300 and "model" not in d["config"]
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"} and len(d["roofline"]["kernel"]) <= 60
    assert set(d["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"} and d["cpu_baseline"]["kind"] == "port"
    assert d["roofline_msda"]["decoder_box_frac"] == 0.034 and d["root_load"]["predicted_efficiency"] == 0.884
    assert d["config_R50_ovis_720"] == {"value": 306.9, "value_median": None, "ms_per_step": 195.5, "frames_per_step": 60, "steps": 4,
                                        "roofline_frac": 0.5958, "msda_frac": 0.1404} or d["config_R50_ovis_720"]["roofline_frac"] == 0.5958
    assert d["fast_mode"] == 1073.0 and not any(isinstance(v, str) and len(v) > 300 for v in d.values())
    # no string anywhere in the line is a paragraph
    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(v) for v in strings(d)) <= 300
    # an object that cannot fit drops optional keys from the end, never the contract
    full["scaling_breakdown"]["per_rank_ms"]["compute"] = [123.456] * 3000
    d2 = json.loads(compact_line(full, None))
    assert "roofline" in d2 and "cpu_baseline" in d2 and "scaling_breakdown" not in d2 and len(json.dumps(d2)) <= LINE_LIMIT


def test_orchestrator_runs_the_headline_first_and_skips_the_rehearsal_when_the_budget_is_spent(monkeypatch, capsys, tmp_path):
    """The default invocation: the parent starts the `main` leg FIRST, then the N = 8 root-load leg in a second child only while the wall
    budget has room; it prints one compact line and writes the full objects to the extras file.  (run_leg is replaced: no GPU here.)"""
    import json
    import types
    import bench
    calls = []
    full = _fat_full_line()
    del full["root_load"], full["root_load_halo"]

    def fake_leg(leg, env, argv, budget):
        calls.append((leg, dict(env), list(argv)))
        if leg == "main":
            return dict(full), 0
        if leg == "cpu":                                      # the CPU baseline is its own leg, alone on the box between the two GPU legs
            return {"cpu_baseline": dict(full["cpu_baseline"], value=0.5)}, 0
        return {"root_load": {"world": 8, "ms_per_step": 160.0, "verified": True}, "root_load_halo": {"world": 8, "ms_per_step": 150.0, "verified": True}}, 0
    monkeypatch.setattr(bench, "run_leg", fake_leg)
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    args = types.SimpleNamespace(no_fast_mode=False, no_cpu_baseline=False, config="R50_ovis_360", precision="f32", frames=120)
    monkeypatch.delenv("MDQE_BENCH_ROOT_LOAD_LEG", raising=False)
    assert bench.orchestrate(args, ["--steps", "20"]) == 0
    out = [l for l in capsys.readouterr().out.splitlines() if l.strip()]
    assert len(out) == 1 and len(out[0]) <= bench.LINE_LIMIT
    d = json.loads(out[0])
    assert [c[0] for c in calls] == ["main", "cpu", "root_load"] and calls[2][1]["MDQE_BENCH_ROOT_LOAD"] == "8"
    assert "--no-cpu-baseline" in calls[0][2] and d["cpu_baseline"]["value"] == 0.5       # the main leg is not asked for it; the CPU leg's result rides in the line
    assert abs(d["root_load"]["predicted_efficiency"] - 145.26 / 160.0) < 1e-3 and abs(d["root_load_halo"]["predicted_efficiency"] - 145.26 / 150.0) < 1e-3
    ex = json.load(open(os.path.join(str(tmp_path), d["extras"])))
    assert len(ex["config"]["workload"]) > 3000 and ex["root_load"]["single_gpu_ms_per_step"] == 145.26      # the full objects are in the file
    # budget spent: the rehearsal is skipped, the headline still goes out
    calls.clear()
    monkeypatch.setenv("MDQE_BENCH_BUDGET_S", "1")
    assert bench.orchestrate(args, []) == 0
    d = json.loads(capsys.readouterr().out.strip())
    assert [c[0] for c in calls] == ["main", "cpu"] and "skipped" in d["root_load"] and d["value"] == 826.123
    # a failing main leg: no line, its exit code
    monkeypatch.setattr(bench, "run_leg", lambda *a: ({"error": "boom"}, 3))
    assert bench.orchestrate(args, []) == 3 and capsys.readouterr().out.strip() == ""
