"""world_size-2 tests of the multi-GPU path on CPU (gloo): shard bounds, the single fused all-gather of the scorer's
exchange step, and replicated results.  The predictors are replaced by CPU stubs with the same ``predict`` signatures
(the real ones need the HIP library and a GPU by design); the collective / sharding / ordering logic under test is the
code that runs on RCCL."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from foundationpose_amd import dist as fpd


def test_shard_bounds():
    assert fpd.shard_bounds(252, 8) == [(0, 32), (32, 64), (64, 96), (96, 128), (128, 160), (160, 192), (192, 224), (224, 252)]
    assert fpd.shard_bounds(7, 2) == [(0, 4), (4, 7)]
    assert fpd.shard_bounds(3, 4) == [(0, 1), (1, 2), (2, 3), (3, 3)]
    assert fpd.shard_bounds(0, 2) == [(0, 0), (0, 0)]
    for n in (1, 5, 64, 252):
        for w in (1, 2, 3, 8):
            b = fpd.shard_bounds(n, w)
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(w - 1))


class _StubRefiner:
    """pose -> pose, row-wise deterministic (stands in for PoseRefinePredictor.predict)"""

    def predict(self, rgb, depth, K, ob_in_cams, xyz_map, mesh=None, mesh_tensors=None, mesh_diameter=None, iteration=5,
                shared_translation=None):
        P = torch.as_tensor(ob_in_cams, dtype=torch.float32).clone()
        P[:, :3, 3] += 0.01 * iteration * torch.sin(P[:, :3, 3] * 37.0)
        return P, None


class _StubScorer:
    """features = fixed random projection of the pose (row-wise), then the REAL cross-hypothesis head of ScorePlan"""

    def __init__(self):
        g = torch.Generator().manual_seed(11)
        self.proj = torch.randn((16, 512), generator=g)
        from foundationpose_amd.engine import _TorchMHA
        sd = {"att_cross.in_proj_weight": torch.randn((1536, 512), generator=g) * 0.05,
              "att_cross.in_proj_bias": torch.randn((1536,), generator=g) * 0.05,
              "att_cross.out_proj.weight": torch.randn((512, 512), generator=g) * 0.05,
              "att_cross.out_proj.bias": torch.randn((512,), generator=g) * 0.05}
        self.att_cross = _TorchMHA(sd, "att_cross", torch.float32)
        self.lin = torch.randn((1, 512), generator=g) * 0.05

    def predict(self, rgb, depth, K, ob_in_cams, mesh=None, mesh_tensors=None, mesh_diameter=None, feature_exchange=None):
        feats = torch.tanh(torch.as_tensor(ob_in_cams, dtype=torch.float32).reshape(-1, 16) @ self.proj)
        if feature_exchange is not None:
            feats = feature_exchange(feats)
        x = self.att_cross(feats[None])
        return (x @ self.lin.t()).reshape(-1) + 100, None


def _poses(n):
    g = torch.Generator().manual_seed(n)
    P = torch.eye(4).repeat(n, 1, 1)
    P[:, :3, :] = torch.randn((n, 3, 4), generator=g)
    return P


def _worker(rank, world, port, n, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.set_num_threads(1)
        P = _poses(n)
        poses, scores, order = fpd.register_hypothesis_parallel(_StubRefiner(), _StubScorer(), None, None, None, P, None,
                                                                iteration=3)
        b, e = fpd.shard_bounds(n, world)[rank]
        rows = fpd.all_gather_rows(torch.arange(b, e, dtype=torch.float32)[:, None] * torch.ones(1, 3), n)
        rec = fpd.gather_object_records(scores + rank, poses)
        q.put((rank, poses.numpy(), scores.numpy(), order.numpy(), rows.numpy(), rec.numpy()))
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("n", [8, 7, 3])
def test_hypothesis_parallel_world2_matches_single_process(n):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=60) for _ in range(world)], key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference (world size 1 path of the same functions)
    P = _poses(n)
    poses, scores, order = fpd.register_hypothesis_parallel(_StubRefiner(), _StubScorer(), None, None, None, P, None, iteration=3)
    for rank, p, s, o, rows, rec in res:
        assert np.array_equal(o, order.numpy()), "ranking differs from the single-process result"
        np.testing.assert_allclose(p, poses.numpy(), rtol=0, atol=0)
        np.testing.assert_allclose(s, scores.numpy(), rtol=0, atol=1e-5)
        assert np.array_equal(rows[:, 0], np.arange(n, dtype=np.float32))  # padded shards are stripped, order kept
        assert rec.shape == (world, n, 17)
        np.testing.assert_allclose(rec[1, :, 0] - rec[0, :, 0], 1.0, atol=1e-5)  # record of rank r carries rank r's scores
    assert np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])  # replicated
