"""GPU: the bench's one JSON line keeps the driver's contract (keys, types, internal consistency) -- a short run of the real script."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line_and_full(stdout):
    """(the ONE printed line as a dict, its text, the full objects of the extras file it names)."""
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), stdout[:2000]          # nothing but the JSON line on stdout
    assert len(lines[0]) <= 4096, len(lines[0])                               # a 25.7 KB line came back `parsed: null`
    d = json.loads(lines[0])
    path = d["extras"] if os.path.isabs(d["extras"]) else os.path.join(ROOT, d["extras"])
    return d, lines[0], json.load(open(path))


def _contract(d, frames, steps, warmup):
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict)):
        assert isinstance(d[k], t), (k, d[k])
    assert d["vs_baseline"] is None and d["steps"] == steps and d["warmup"] == warmup and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["unit"] == "frames/s"
    assert "workload" in d["config"] and "model" not in d["config"] and "H2D included" in d["config"]["workload"] and len(d["config"]["workload"]) <= 300
    assert abs(d["value"] - frames * 1e3 / d["ms_per_step"]) < 2e-3 * d["value"]          # value = frames per step / time per step (rounded to 3 places)
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s") and rf["traffic"] is None and rf["traffic_ref"].startswith("profiles/")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.0 < rf["frac"] < 1.0 and rf["peak"] == 157.3 and len(rf["kernel"]) <= 60


def test_bench_line_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--frames", "24", "--no-cpu-baseline", "--no-fast-mode"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    c, text, d = _line_and_full(r.stdout)
    _contract(c, 24, 2, 1)
    assert c["n_gpus"] == 1 and c["config"]["clips_per_step"] == 22 and c["config"]["instances_out"] >= 1
    assert 0.0 < c["roofline_msda"]["frac"] < 1.0 and 0.0 < c["roofline_msda"]["decoder_box_frac"] < 1.0 and 0.0 < c["roofline_msda"]["decoder_temporal_frac"] < 1.0
    assert "cpu_baseline" not in c and "root_load" not in c                  # (--no-cpu-baseline --no-fast-mode)
    # ... and the full objects of the same run (the extras file)
    assert abs(d["value"] - 24 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"] and abs(d["value"] - c["value"]) < 1e-3
    rf = d["roofline"]
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert d["config"]["clips_per_step"] == 24 - 2 and d["config"]["instances_out"] >= 1
    assert rf["launches_timed"] <= rf["launches_total"] and rf["launches_total"] >= 1
    # the deformable gather's HBM figure is in the same line (north star: "rocprof HBM GB/s on the deformable gather")
    rm = d["roofline_msda"]
    assert rm["bound"] == "hbm" and rm["peak"] == 8.0 and rm["unit"] == "TB/s" and 0.0 < rm["frac"] < 1.0 and 0.0 < rm["frac_isolated"] < 1.0
    assert abs(rm["frac"] - rm["achieved"] / rm["peak"]) < 1e-9 and rm["traffic"] is None and "profiles/" in rm["traffic_ref"]
    # 24 frames = one pass: 18.3 MB of algorithmic bytes per frame and layer at 360p
    assert abs(rm["algorithmic_MB_per_launch"] / 24 - 18.28) < 0.1
    for k in ("decoder_box", "decoder_temporal"):
        # on UNIQUE value bytes (a batch of stride-1 clips shares a frame's map T ways): nothing may exceed the HBM roof, in the pipeline or alone
        assert 0.0 < rm[k]["frac"] < 1.0 and 0.0 < rm[k]["frac_isolated"] < 1.0 and rm[k]["launches_timed"] >= 1
        pc = rm[k]["bytes_per_clip_convention"]
        assert pc["MB_per_launch"] >= rm[k]["algorithmic_MB_per_launch"] and "frac" not in pc
    # mean and median of the per-step times, and which one `value` is
    assert d["value_median"] > 0 and abs(d["value_median"] / d["value"] - 1) < 0.25
    assert d["bench_wall_s"] > 0 and "degraded" not in d


def test_bench_default_invocation_every_leg_on_fits_the_line():
    """The DEFAULT invocation (every leg on: extra modes, CPU baseline, configs[2] / configs[3], the N = 8 root-load rehearsal in both forms) at
    reduced size: ONE line of at most 4096 bytes that keeps the driver's contract and carries the extras as numbers; the headline is timed
    FIRST (the rehearsal's child starts after the main leg has left); the full objects are in the extras file."""
    env = dict(os.environ, MDQE_BENCH_SIDE_FRAMES="12", MDQE_BENCH_SIDE_STEPS="1", MDQE_BENCH_SIDE_S="300", MDQE_BENCH_ROOT_LOAD_LEG="8",
               MDQE_BENCH_ROOT_LOAD_S="400")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--frames", "36"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    c, text, d = _line_and_full(r.stdout)
    _contract(c, 36, 2, 1)
    cb = c["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1 and cb["unit"] == "frames/s" and 0 < cb["as_reference_value"] < cb["value"] and cb["sample"]
    for k in ("fast_mode", "autocast_f16", "reference_precision_map", "stream_mode", "frames_resident", "late_masks", "init_reference"):
        assert isinstance(c[k], float) and c[k] > 0, k                        # numbers only
    for k in ("config_R50_ovis_720", "config_swinl_ovis"):
        assert c[k]["value"] > 0 and 0.0 < c[k]["roofline_frac"] < 1.0 and 0.0 < c[k]["msda_frac"] < 1.0
    for k in ("root_load", "root_load_halo"):
        assert c[k]["world"] == 8 and c[k]["ms_per_step"] > 0 and 0.0 < c[k]["predicted_efficiency"] <= 1.1 and c[k]["verified"] is True
    assert 0.0 < c["clip_stage"]["frac"] < 1.0 and 0.0 < c["roofline_isolated"]["frac"] < 1.0
    assert "degraded" not in d and abs(d["value"] - 36 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert d["root_load"]["wall_s"] > 0 and d["bench_wall_s"] > d["root_load"]["wall_s"]
    for key, cfgname in (("config_R50_ovis_720", "R50_ovis_720"), ("config_swinl_ovis", "swinl_ovis")):
        e = d[key]
        assert "error" not in e, e
        assert e["value"] > 0 and e["frames_per_step"] == 12 and e["steps"] == 1 and cfgname in e["workload"] and e["dtype"] == "f32"
        assert abs(e["value"] - 12 * 1e3 / e["ms_per_step"]) < 1e-6 * e["value"]
        assert 0.0 < e["roofline"]["frac"] < 1.0 and e["roofline"]["peak"] == 157.3 and 0.0 < e["roofline_msda"]["frac"] < 1.0
        assert e["instances_out"] >= 1
    rl = d["root_load"]
    assert "error" not in rl, rl
    assert rl["world"] == 8 and rl["frames_virtual"] == 8 * 36 and rl["ms_per_step"] > 0 and 0.0 < rl["predicted_efficiency"] <= 1.1
    assert abs(rl["single_gpu_ms_per_step"] - d["ms_per_step"]) < 1e-9      # against THIS run's headline
    assert rl["verified"] is True and rl["replay_total_ms"] > 0 and rl["compute"] > 0 and rl["tracked_instances"] >= 1
    # rank 0 rests in the last round: the job's step is the slower of rank 0's and another rank's (a second child plays rank 1)
    assert isinstance(rl["chunk_frames_per_round"][-1], list) and rl["chunk_frames_per_round"][-1][0] == 0
    assert rl["ms_per_step"] == max(rl["root_ms_per_step"], rl["other_rank_ms_per_step"]) and rl["other_rank_frames_per_step"] > rl["root_frames_per_step"]
    assert rl["root_ms_per_step"] >= rl["last_gather_not_before_ms"]               # rank 0 was held at the last gather until rank 1 would have delivered
    assert abs(rl["predicted_efficiency"] - rl["single_gpu_ms_per_step"] / rl["ms_per_step"]) < 1e-9
    assert rl["tracker_native_ms_per_step"]["updates_per_step"] >= 8 * 30       # the replay really carried eight ranks' clips
    # ... and the halo-exchange form of the same rehearsal: no frame twice, rank 0 rests in the last round and takes a smaller chunk before
    rh = d["root_load_halo"]
    assert "error" not in rh, rh
    assert rh["world"] == 8 and rh["verified"] is True and rh["halo_frac"] == 0.0 and rh["ms_per_step"] > 0 and 0.0 < rh["predicted_efficiency"] <= 1.1
    first, last = rh["chunk_frames_per_round"][0], rh["chunk_frames_per_round"][-1]
    assert isinstance(first, list) and first[0] < first[1] and sum(first) == 8 * rl["chunk_frames_per_round"][0] and last[0] == 0
    assert rh["tracker_native_ms_per_step"]["updates_per_step"] >= 8 * 30


def test_bench_gpus_2_runs_two_ranks_without_torchrun():
    """`python bench.py --gpus 2` spawns its own ranks (rehearsal hooks: both on this box's one GPU, gloo instead of RCCL) and prints a
    line for TWO ranks."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    # (--halo-exchange: the rehearsal also takes the neighbour send/recv of the partitioned chunks through bench.py itself, on its own
    # process group; the recompute form of the same schedule is held to the single-GPU result in tests/test_sharded_gpu.py)
    # MDQE_BENCH_HALO_AB=1: the halo-exchange form runs as the extra key `halo_exchange` of the same line (its own process group, its own
    # bit-exact verification); the recompute form is the headline
    args = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--frames", "24", "--no-cpu-baseline", "--no-fast-mode"]
    env["MDQE_BENCH_HALO_AB"] = "1"
    if torch.cuda.device_count() < 2:
        # (that the same command WITHOUT the rehearsal hooks fails loudly on a box with fewer GPUs than ranks is held by
        # tests/test_bench_cpu.py::test_bench_never_reports_the_wrong_world -- the ranks' own device-count check, no GPU needed)
        env.update(MDQE_BENCH_BACKEND="gloo", MDQE_BENCH_ONE_DEVICE="1")
    r = subprocess.run(args, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    c, text, d = _line_and_full(r.stdout)                                    # (gloo / RCCL banners -> stderr)
    assert c["n_gpus"] == 2 and c["config"]["ranks_seen"] == 2 and c["verified"] is True and c["scaling"] == "weak"
    assert abs(c["value"] - 48 * 1e3 / c["ms_per_step"]) < 2e-3 * c["value"] and len(c["scaling_breakdown"]["compute_ms"]) == 2
    assert c["halo_exchange"]["verified"] is True and c["halo_exchange"]["value"] > 0
    # rank 0's chunk share came from this run's own measured per-rank times (sharding.measured_root_share), the same on every rank
    mp = d["measured_plan"]
    assert mp["share_before"] == 1.0 and 0.5 <= mp["share"] <= 1.25 and len(mp["busy_ms"]) == 2 and min(mp["busy_ms"]) > 0 and len(mp["frames"]) == 2
    assert c["measured_plan"]["share"] == mp["share"] and mp["chunk_frames_per_round"]
    assert d["n_gpus"] == 2 and d["config"]["ranks_seen"] == 2 and d["config"]["frames_per_gpu"] == 24 and d["scaling"] == "weak"
    assert abs(d["value"] - 48 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]              # whole-job frames / max-over-ranks time
    assert d["config"]["instances_out"] >= 1
    # self-verification + the per-rank breakdown of the sharded schedule (what makes the first real 8-GPU run diagnosable)
    assert d["verified"] is True and d["verification"]["ok"] is True and d["verification"]["frames"] >= 60
    sb = d["scaling_breakdown"]
    for k in ("compute", "pack", "gather_wait", "gather_payload", "feed", "replay_exposed"):
        assert len(sb["per_rank_ms"][k]) == 2 and all(v >= 0 for v in sb["per_rank_ms"][k])
    assert sb["per_rank_ms"]["compute"][0] > 0 and sb["replay_exposed_ms"] >= 0 and sb["gather_ms"] >= 0 and 0 <= sb["halo_frac"] < 0.5
    he = d["halo_exchange"]
    assert he["verified"] is True and he["value"] > 0 and he["halo_frac"] == 0.0 and len(he["per_rank_ms"]["compute"]) == 2
