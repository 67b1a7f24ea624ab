"""GPU: the sharded path end to end with 2 ranks must reproduce the single-process result.  With two or more GPUs visible
the ranks take one GPU each and talk over RCCL (backend "nccl": the production path of bench.py --gpus N); on a 1-GPU box
both ranks share cuda:0 and rendezvous over gloo (RCCL needs one GPU per rank).  MDQE_TEST_BACKEND=gloo|nccl overrides."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _cfg():
    from mdqe_cvpr2023_amd.config import MDQEConfig
    return MDQEConfig(enc_layers=1, dec_layers=1, n_frames=3, n_frames_test=3, n_frames_window_test=4, num_classes=5,
                      num_queries=16, query_embed_dim=16, n_max_inst=40, apply_cls_thres=0.12)


def _video(L=11, seed=2):
    g = torch.Generator().manual_seed(seed)
    base = torch.randint(0, 256, (3, 64, 96), generator=g, dtype=torch.uint8).float()
    fr = torch.randint(0, 256, (L, 3, 64, 96), generator=g, dtype=torch.uint8).float()
    return (0.8 * base[None] + 0.2 * fr).round().to(torch.uint8)


STREAM = ((11, 2), (5, 3), (14, 4))        # (frames, seed) of the videos of the stream test: 3, 1 and 4 chunks of 4 frames


def worker(rank, world, port, outdir):
    import torch.distributed as dist
    from mdqe_cvpr2023_amd import sharding
    from mdqe_cvpr2023_amd.meta_arch import MDQE
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    backend = os.environ.get("MDQE_TEST_BACKEND") or ("nccl" if torch.cuda.device_count() >= world else "gloo")
    dev = rank if backend == "nccl" else 0
    torch.cuda.set_device(dev)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    import dataclasses
    cfg = dataclasses.replace(_cfg(), device="cuda:%d" % dev)
    model = MDQE(cfg, seed=5).eval()
    video = _video()
    L = video.shape[0]
    with torch.no_grad():
        if os.environ.get("MDQE_TEST_SHARDING", "").startswith("stream"):
            root_only = os.environ["MDQE_TEST_SHARDING"] == "stream_root_only"
            jobs = []
            halo = os.environ["MDQE_TEST_SHARDING"] == "stream_halo_exchange"
            for Lv, seed in STREAM:
                v = _video(Lv, seed)
                chunk = 8 if os.environ["MDQE_TEST_SHARDING"] == "stream_two_window_chunks" else 4     # tracker window = 4 frames
                plan = sharding.chunk_plan(Lv, cfg.n_frames_test, cfg.clip_stride, chunk, halo_exchange=halo)
                jobs.append(({g: v[plan[g][1]:plan[g][2]].cuda() for g in sharding.owned_chunks(plan, world, rank)}, plan, v[:0].cuda()))
            out = list(sharding.run_round_robin_stream(model, jobs, rank, world, dist, (64, 96), root_only=root_only, halo_exchange=halo))
            assert len(out) == len(STREAM) and (not root_only or all((o is None) == (rank != 0) for o in out))
        elif os.environ.get("MDQE_TEST_SHARDING") == "contiguous":
            f0, f1 = sharding.frame_range(L, world, rank, cfg.n_frames_test)
            out = sharding.run_sharded(model, video[f0:f1].cuda(), f0, L, rank, world, dist, (64, 96))
        else:
            halo = os.environ.get("MDQE_TEST_SHARDING") in ("halo_exchange", "decreasing_halo_exchange")
            sizes = [4, 3] if os.environ.get("MDQE_TEST_SHARDING", "").startswith("decreasing") else 4     # decreasing rounds: 4 4 | 3
            plan = sharding.chunk_plan(L, cfg.n_frames_test, cfg.clip_stride, sizes, halo_exchange=halo, world=world)   # 4-frame chunks -> 3 chunks, 2 rounds
            frames = {g: video[plan[g][1]:plan[g][2]].cuda() for g in sharding.owned_chunks(plan, world, rank)}
            out = sharding.run_round_robin(model, frames, plan, rank, world, dist, (64, 96),
                                           root_only=os.environ.get("MDQE_TEST_SHARDING") == "root_only", halo_exchange=halo)
            if os.environ.get("MDQE_TEST_SHARDING") == "root_only":
                assert (out is None) == (rank != 0)
    if os.environ.get("MDQE_TEST_P2P_SELF") == "1":
        # sharding._Halo's grouped send/recv through the real backend, addressed to this rank itself (the only peer a 1-GPU box has)
        h = sharding._Halo(dist, rank, rank, (2, 7, 16, 4, 6, 8), torch.device("cuda", dev))
        g = torch.Generator().manual_seed(1)
        enc, mf = torch.randn(2, 7, 16, generator=g).cuda(), torch.randn(2, 4, 6, 8, generator=g).cuda()
        h.on_tail(enc, mf)
        e2, m2 = h.head()
        torch.cuda.synchronize()
        assert torch.equal(e2, enc) and torch.equal(m2, mf)
    torch.save(out, os.path.join(outdir, f"rank{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["stream", "stream_root_only", "stream_two_window_chunks", "stream_halo_exchange"])
def test_two_rank_stream_of_videos_equals_single_gpu(tmp_path, mode):
    """run_round_robin_stream over three videos of 3, 1 and 4 chunks (a rank sits out a last round, or a whole video): every video's
    result equals the single-process one, in order, on both ranks (all-ranks form) or on rank 0 (root-only form)."""
    from mdqe_cvpr2023_amd.meta_arch import MDQE
    os.environ["MDQE_TEST_SHARDING"] = mode
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    model = MDQE(_cfg(), seed=5).eval()
    with torch.no_grad():
        refs = [model([{"image": _video(Lv, seed), "height": 64, "width": 96}]) for Lv, seed in STREAM]
    for r in range(1 if mode == "stream_root_only" else 2):
        outs = torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"), weights_only=False)
        for out, ref in zip(outs, refs):
            assert out["pred_labels"] == ref["pred_labels"]
            assert torch.allclose(torch.tensor(out["pred_scores"]), torch.tensor(ref["pred_scores"]), atol=1e-6)
            assert all(torch.equal(a, b) for a, b in zip(out["pred_masks"], ref["pred_masks"]))


@pytest.mark.parametrize("mode", ["round_robin", "root_only", "contiguous", "halo_exchange", "decreasing", "decreasing_halo_exchange"])
def test_two_rank_sharded_video_equals_single_gpu(tmp_path, mode):
    from mdqe_cvpr2023_amd.meta_arch import MDQE
    os.environ["MDQE_TEST_SHARDING"] = mode
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=worker, args=(r, 2, port, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    model = MDQE(_cfg(), seed=5).eval()
    with torch.no_grad():
        ref = model([{"image": _video(), "height": 64, "width": 96}])
    for r in range(1 if mode == "root_only" else 2):
        out = torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"), weights_only=False)
        assert out["pred_labels"] == ref["pred_labels"]
        assert torch.allclose(torch.tensor(out["pred_scores"]), torch.tensor(ref["pred_scores"]), atol=1e-6)
        assert all(torch.equal(a, b) for a, b in zip(out["pred_masks"], ref["pred_masks"]))


@pytest.mark.parametrize("mode", ["round_robin", "root_only", "stream_root_only", "halo_exchange", "stream_halo_exchange"])
def test_one_rank_rccl_communicator(tmp_path, mode):
    """The sharded schedule over the REAL backend of bench.py --gpus N (`nccl` = RCCL) with a one-rank communicator: all a 1-GPU box can
    offer, but it puts every collective call of the path (all_gather of sizes, gather / all_gather of int64 + fp32 payloads, the
    grouped isend/irecv of the halo exchange addressed to self, barrier-free teardown) through RCCL's argument checks, stream
    handling and kernels beside the replay thread."""
    from mdqe_cvpr2023_amd.meta_arch import MDQE
    os.environ["MDQE_TEST_SHARDING"] = mode
    os.environ["MDQE_TEST_BACKEND"] = "nccl"
    os.environ["MDQE_TEST_P2P_SELF"] = "1"
    try:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ctx = mp.get_context("spawn")
        p = ctx.Process(target=worker, args=(0, 1, port, str(tmp_path)))
        p.start()
        p.join(timeout=300)
        assert p.exitcode == 0
    finally:
        del os.environ["MDQE_TEST_BACKEND"], os.environ["MDQE_TEST_P2P_SELF"]
    model = MDQE(_cfg(), seed=5).eval()
    with torch.no_grad():
        vids = STREAM if mode.startswith("stream") else ((11, 2),)
        refs = [model([{"image": _video(Lv, seed), "height": 64, "width": 96}]) for Lv, seed in vids]
    outs = torch.load(os.path.join(str(tmp_path), "rank0.pt"), weights_only=False)
    outs = outs if mode.startswith("stream") else [outs]
    for out, ref in zip(outs, refs):
        assert out["pred_labels"] == ref["pred_labels"]
        assert torch.allclose(torch.tensor(out["pred_scores"]), torch.tensor(ref["pred_scores"]), atol=1e-6)
        assert all(torch.equal(a, b) for a, b in zip(out["pred_masks"], ref["pred_masks"]))
