"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the
CPU tests).  The reference has no distributed code at all; this is the design of SURVEY.md 8(e):

  * object-parallel (BASELINE configs[3]): rank g owns object g -- mesh, frame, all its hypotheses.  No data-path
    collective until the result: ONE all-gather of the fused per-object record [score | refined pose] per register().
  * hypothesis-parallel (one object, N hypotheses): contiguous shards of ceil(N/G) hypotheses.  Refinement
    (estimater.py:215) is embarrassingly parallel.  Scoring has exactly one exchange step: ScoreNetMultiPair's
    cross-hypothesis attention (score_network.py:84-88) mixes all L hypotheses, so the pooled 512-d feature of every
    hypothesis is all-gathered ONCE -- with the refined pose riding in the same buffer -- and every rank then runs the
    two tiny cross layers redundantly, which leaves scores and poses replicated without a second collective.

Both collectives move tens to hundreds of KB: latency-bound on xGMI, so they are single fused all-gathers enqueued on
the compute stream, never rings of small messages.  "On the compute stream": a synchronous torch.distributed collective
(async_op=False; distributed_c10d sets `AllgatherOptions.asyncOp = False` and does not call work.wait() -- "the backend
has sync'ed at CPP level") is issued by ProcessGroupNCCL on the CURRENT stream in the PyTorch of this image (2.10), not on
the group's side stream, so the exchange is ordered like any other kernel of the step and is captured into a hipGraph with
it (tests/test_gpu_parity.py::test_rccl_all_gather_is_captured_with_the_compute_stream).

A group on the "gloo" backend with device tensors (several ranks sharing one GPU -- RCCL refuses two ranks on one device --
or a host without RCCL) takes the same two functions: `_all_gather` stages that one buffer through host memory.  Only the
collective changes; it is how tests/test_gpu_parity.py runs two REAL ranks (real predictors, real rendezvous, a real
inter-process all-gather) on the single GPU of the test box.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, world):
    """contiguous shards of ceil(n/world) items (the last ones may be short or empty) -> [(begin, end)] * world"""
    chunk = -(-n // world) if n > 0 else 0
    return [(min(r * chunk, n), min((r + 1) * chunk, n)) for r in range(world)]


def _world(group):
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def _all_gather(out, send, group=None):
    """the all-gather of this module: RCCL on the current stream; on a gloo group with device tensors, through host memory"""
    if send.is_cuda and dist.get_backend(group) == "gloo":
        host = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(host, send.cpu(), group=group)
        out.copy_(host)
        return
    dist.all_gather_into_tensor(out, send, group=group)


def all_gather_rows(local, n_total, group=None, collective=None, world_rank=None):
    """local: this rank's shard (n_local, C) of a row-sharded (n_total, C) tensor (sharding = shard_bounds).
    Returns the full tensor on every rank with ONE all-gather (short shards are padded to the common chunk size).
    collective(out, send): the all-gather itself (default torch.distributed.all_gather_into_tensor on `group`);
    world_rank=(world, rank) overrides the process group's -- both exist so that tests can run the G ranks of a node one
    after the other on one GPU with everything else (padding, stripping, ordering) being the code that runs on RCCL."""
    world, rank = world_rank if world_rank is not None else _world(group)
    if world == 1:
        if local.shape[0] != n_total:
            raise ValueError(f"all_gather_rows: single process holds {local.shape[0]} rows, expected {n_total}")
        return local
    bounds = shard_bounds(n_total, world)
    b, e = bounds[rank]
    if local.shape[0] != e - b:
        raise ValueError(f"all_gather_rows: rank {rank} holds {local.shape[0]} rows, its shard is [{b},{e})")
    chunk = bounds[0][1] - bounds[0][0]
    C = local.shape[1]
    send = local.contiguous()
    if e - b < chunk:  # pad with copies of the last valid row (or zeros for an empty shard); stripped below
        pad = send[-1:].expand(chunk - (e - b), C) if e > b else torch.zeros((chunk, C), dtype=local.dtype, device=local.device)
        send = torch.cat([send, pad], dim=0)
    out = torch.empty((world * chunk, C), dtype=local.dtype, device=local.device)
    if collective is not None:
        collective(out, send)
    else:
        _all_gather(out, send, group)
    if world * chunk == n_total:
        return out
    return torch.cat([out[r * chunk:r * chunk + (be - bb)] for r, (bb, be) in enumerate(bounds)], dim=0)


class FeaturePoseExchange:
    """The scorer's single exchange step.  Called by ScorePredictor.predict(feature_exchange=...) with the pooled
    features of the local shard; all-gathers [feature (512) | refined pose (16)] records and keeps the gathered
    poses in ``poses_all``.  Records travel as float32 (528 floats/hypothesis, 532 KB at N=252)."""

    def __init__(self, poses_local, n_total, group=None, collective=None, world_rank=None):
        self.poses_local = poses_local.reshape(-1, 16)
        self.n_total = n_total
        self.group = group
        self.collective, self.world_rank = collective, world_rank      # see all_gather_rows
        self.poses_all = None

    def __call__(self, feats_local):
        rec = torch.cat([feats_local.float(), self.poses_local.float()], dim=1)
        full = all_gather_rows(rec, self.n_total, self.group, self.collective, self.world_rank)
        D = feats_local.shape[1]
        self.poses_all = full[:, D:].reshape(-1, 4, 4).contiguous()
        return full[:, :D].to(feats_local.dtype).contiguous()


def register_hypothesis_parallel(refiner, scorer, rgb, depth, K, poses_all, xyz_map, mesh=None, mesh_tensors=None,
                                 mesh_diameter=None, iteration=5, group=None, collective=None, world_rank=None,
                                 shared_translation=None):
    """estimater.py:214-229 (refine all hypotheses, score them, sort) with the hypotheses sharded over the ranks.
    ``poses_all`` (N,4,4) is the same on every rank.  Returns (poses sorted by score (N,4,4), scores sorted (N,),
    order) -- replicated on every rank.  collective / world_rank: see all_gather_rows; shared_translation: see
    PoseRefinePredictor.predict."""
    world, rank = world_rank if world_rank is not None else _world(group)
    poses_all = torch.as_tensor(poses_all)
    N = poses_all.shape[0]
    b, e = shard_bounds(N, world)[rank]
    # a shard must return the bits of the single batch whatever its size: with more than one rank no call takes the small-call
    # kernels (split-K convolutions / two-stream heads, engine.py), whose summation order is another one.  (A single batch of <= 12
    # hypotheses does take them; sharding such a call over ranks is not something the bit-equality contract covers.)
    saved = getattr(refiner, "small_calls", True), getattr(scorer, "small_calls", True)
    if world > 1:
        refiner.small_calls = scorer.small_calls = False
    try:
        local, _ = refiner.predict(rgb, depth, K, poses_all[b:e], xyz_map, mesh=mesh, mesh_tensors=mesh_tensors,
                                   mesh_diameter=mesh_diameter, iteration=iteration, shared_translation=shared_translation)
        ex = FeaturePoseExchange(local, N, group, collective, world_rank)
        scores, _ = scorer.predict(rgb, depth, K, local, mesh=mesh, mesh_tensors=mesh_tensors,
                                   mesh_diameter=mesh_diameter, feature_exchange=ex)
    finally:
        refiner.small_calls, scorer.small_calls = saved
    order = scores.argsort(descending=True)
    return ex.poses_all[order], scores[order], order


def gather_object_records(scores, poses, group=None):
    """object-parallel result exchange: every rank contributes its object's [score | pose] (N,17) record;
    returns (world, N, 17) on every rank with ONE all-gather."""
    world, _ = _world(group)
    rec = torch.cat([scores.reshape(-1, 1).float(), poses.reshape(-1, 16).float()], dim=1).contiguous()
    if world == 1 and not (dist.is_available() and dist.is_initialized()):
        return rec[None]
    out = torch.empty((world * rec.shape[0], rec.shape[1]), dtype=rec.dtype, device=rec.device)
    _all_gather(out, rec, group)  # concatenated along dim 0 (same form on RCCL and gloo)
    return out.reshape(world, rec.shape[0], rec.shape[1])
