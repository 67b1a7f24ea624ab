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
    assert 0 < r["as_reference"]["value"] < r["value"]           # the reference's window recompute costs three more frame passes


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
