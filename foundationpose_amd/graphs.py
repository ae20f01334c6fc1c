"""hipGraph capture of the per-frame tracking path (BASELINE configs[4]).

`FoundationPose.track_one` (estimater.py:250-268) is, per frame: depth erosion -> bilateral filter -> back-projection
-> `iteration` x (crop windows -> rasterise -> observed crop -> RefineNet -> pose update).  Shapes are static (frame
size, number of hypotheses, iteration count), only values change, and every entry point of libfp_amd.so enqueues on
the caller's stream without allocating or synchronising, so the whole frame is captured ONCE into hipGraphs (through
torch.cuda.CUDAGraph, which also owns the private memory pool of the intermediate tensors) and replayed per frame:
about 100 kernel launches per refine iteration collapse into a few graph launches, and the pose stays on the device
between frames (the reference does a .cpu()/.cuda() round trip and an empty_cache() per frame,
estimater.py:263, predict_pose_refine.py:237).
"""
import numpy as np
import torch

from . import ops
from .Utils import get_mesh_handle


class PartGraphs:
    """k LINEAR hipGraphs, one per hypothesis sub-batch of a predictor (overlap.py), replayed on the sub-batch streams:
    forked from / joined into the caller's stream, so the overlap between sub-batches is ordinary stream semantics and every
    graph stays a simple chain.  `body(h)` issues part h's whole launch sequence on the CURRENT stream; everything it
    addresses by raw pointer has to outlive the graphs (static inputs / outputs owned by the caller; tensors allocated inside
    the capture live in the graph's private pool)."""

    def __init__(self, sub, dev, n_parts, body, warmup=2):
        self.sub, self.dev, self.n = sub, dev, n_parts
        # eager warm-up runs on a side stream (lazy one-time initialisation inside the library and in PyTorch), then the
        # captures, one after the other
        s = torch.cuda.Stream(device=dev)
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            for _ in range(warmup):
                for h in range(n_parts):
                    body(h)
        torch.cuda.current_stream(dev).wait_stream(s)
        torch.cuda.synchronize(dev)
        self.graphs = []
        for h in range(n_parts):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                body(h)
            self.graphs.append(g)

    def replay(self):
        streams = self.sub.streams(self.dev, self.n)
        self.sub.fork(streams)
        for h, g in enumerate(self.graphs):
            with torch.cuda.stream(streams[h]):
                g.replay()
        self.sub.join(streams)


class GraphCache:
    """what `graph="auto"` of the predictors means: a (shape, intrinsics, mesh, ...) key is captured the SECOND time it is
    seen -- a single call (a test, one register()) stays eager, a loop over frames or objects of one shape replays graphs from
    its third pass on -- and at most `cap` captured keys are kept (least recently used first out; a capture owns a private
    memory pool).

    Guard rails (round 5, the advisor's finding): a capture costs two eager warm-ups, a device synchronisation, the captures and a
    private pool -- about four eager calls, for a replay that saves a few per cent of one -- so a key set larger than `cap` visited
    round-robin (five tracked objects; register and track keys mixed) must not turn into capture-evict-capture.  (i) Under 'auto' a
    full cache evicts only an entry that has not been replayed for `stale_after` calls of get(); otherwise the new key runs eagerly
    and the resident captures keep replaying.  (`True` is an explicit request and evicts the least recently used entry at once.)
    (ii) An evicted key forgets its sightings: it needs two fresh ones before it is captured again.  (iii) Under 'auto' a failing
    build() (out of memory in a private pool, a capture error) is logged once per key, the key is left eager for good, and the call
    runs eagerly; `True` still raises."""

    def __init__(self, cap=4, stale_after=64):
        self.cap, self.seen, self.items = cap, {}, {}
        self.stale_after, self.clock, self.last_use, self.failed = stale_after, 0, {}, set()
        self.stats = dict(captures=0, replays=0, evictions=0, eager_by_guard=0, build_failures=0)

    def get(self, key, mode, build):
        """mode True: capture now; 'auto': on the second sighting; -> the captured object or None (= run eagerly)"""
        self.clock += 1
        it = self.items.pop(key, None)
        if it is not None:
            self.items[key] = it          # most recently used last
            self.last_use[key] = self.clock
            self.stats["replays"] += 1
            return it
        if mode is False or mode is None:
            return None
        n = self.seen[key] = self.seen.get(key, 0) + 1
        if len(self.seen) > 64:
            self.seen = {key: n}
        if mode == "auto":
            if n < 2 or key in self.failed:
                return None
            if len(self.items) >= self.cap and self.clock - self.last_use[next(iter(self.items))] <= self.stale_after:
                self.stats["eager_by_guard"] += 1      # (i): the residents are in use; this key stays eager
                return None
        try:
            it = build()
        except Exception as e:           # noqa: BLE001 -- whatever the capture raised, 'auto' must not take the call down
            if mode is True:
                raise
            import logging
            logging.getLogger(__name__).warning("graph capture failed for %r (%s: %s); this key stays on eager launches",
                                                key, type(e).__name__, e)
            self.failed.add(key)
            if len(self.failed) > 64:
                self.failed = {key}
            self.stats["build_failures"] += 1
            return None
        self.stats["captures"] += 1
        self.items[key] = it
        self.last_use[key] = self.clock
        while len(self.items) > self.cap:
            old = next(iter(self.items))
            self.items.pop(old)
            self.last_use.pop(old, None)
            self.seen.pop(old, None)      # (ii): two fresh sightings before it is captured again
            self.stats["evictions"] += 1
        return it


class GraphedTracker:
    """Static-shape tracker: `step(rgb, depth[, poses])` -> refined poses (N,4,4) on the device (a static buffer, valid
    until the next call).  With `poses=None` the previous output is the next input (tracking).

    The frame is captured as 1 + k LINEAR graphs: the depth pre-processing, and the whole refine loop of each of the k
    hypothesis sub-batches of the refiner (overlap.py).  A replay launches the first on the caller's stream and the k part
    graphs on k streams forked from / joined into it, so the overlap between sub-batches is ordinary stream semantics and
    every graph stays a simple chain."""

    def __init__(self, refiner, mesh_tensors, mesh_diameter, K, H, W, n_hyp=1, iteration=2, device=None):
        self.refiner = refiner
        self.handle = get_mesh_handle(mesh_tensors)
        self.dev = torch.device(device) if device is not None else self.handle.device
        self.K = np.asarray(K, dtype=np.float64).copy()
        self.H, self.W, self.N, self.R = int(H), int(W), int(n_hyp), int(iteration)
        self.diameter = float(mesh_diameter)
        self.rgb = torch.zeros((H, W, 3), dtype=torch.float32, device=self.dev)
        self.depth = torch.zeros((H, W), dtype=torch.float32, device=self.dev)
        self.poses_in = torch.eye(4, device=self.dev).repeat(self.N, 1, 1).contiguous()
        # everything the captured kernels address by raw pointer and that is not allocated inside a capture belongs to
        # the tracker: the outputs and one rasteriser scratch per part here, the encoder's activation sets by
        # (batch, H, W, slot) in the plan (never dropped)
        oh, ow = int(refiner.cfg["input_resize"][0]), int(refiner.cfg["input_resize"][1])
        self.parts = refiner.sub.parts(self.N, self.dev)
        self.workspace = [torch.empty(max(16, ops.workspace_bytes(b - a, self.handle.V, self.handle.T, oh, ow)),
                                      dtype=torch.uint8, device=self.dev) for a, b in self.parts]
        self.outs = refiner.alloc_outputs(self.N, self.dev) + (self.R,)
        self.poses_out = self.outs[0]
        self.xyz = None
        self.g_pre, self.g_part = None, []
        self._have_output = False

    def _pre(self):
        d = ops.bilateral_filter_depth(ops.erode_depth(self.depth, radius=2), radius=2)
        return ops.depth_to_xyz(d, self.K, zfar=float("inf"), f64_internal=False)     # depth2xyzmap_batch variant

    def _part(self, h, xyz):
        self.refiner.refine_part(h, self.parts[h], self.rgb, xyz, self.poses_in, self.K, self.H, self.W, self.handle,
                                 self.diameter, range(self.R), self.outs, self.workspace[h])

    def _body(self):
        """the frame without graphs: same launches, same streams"""
        xyz = self._pre()
        streams = self.refiner.sub.streams(self.dev, len(self.parts))
        self.refiner.sub.fork(streams)
        for h in range(len(self.parts)):
            with torch.cuda.stream(streams[h]):
                self._part(h, xyz)
        self.refiner.sub.join(streams)
        if self.R <= 0:
            self.poses_out.copy_(self.poses_in)
        return self.poses_out

    @torch.inference_mode()
    def capture(self):
        """two eager warm-up runs on a side stream (lazy one-time initialisation inside the library and in PyTorch),
        then the captures, one after the other"""
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            for _ in range(2):
                self._body()
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        self.g_pre = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_pre):
            self.xyz = self._pre()
        self.g_part = []
        for h in range(len(self.parts)):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._part(h, self.xyz)
            self.g_part.append(g)
        return self

    def replay(self):
        """one frame from the static inputs (rgb, depth, poses_in) into poses_out"""
        self.g_pre.replay()
        streams = self.refiner.sub.streams(self.dev, len(self.parts))
        self.refiner.sub.fork(streams)
        for h, g in enumerate(self.g_part):
            with torch.cuda.stream(streams[h]):
                g.replay()
        self.refiner.sub.join(streams)
        if self.R <= 0:
            self.poses_out.copy_(self.poses_in)
        return self.poses_out

    @torch.inference_mode()
    def step(self, rgb, depth, poses=None):
        if self.g_pre is None:
            self.capture()
        self.rgb.copy_(torch.as_tensor(rgb, device=self.dev), non_blocking=True)
        self.depth.copy_(torch.as_tensor(depth, device=self.dev), non_blocking=True)
        if poses is not None:
            self.poses_in.copy_(torch.as_tensor(poses, device=self.dev, dtype=torch.float32).reshape(self.N, 4, 4))
        elif self._have_output:
            self.poses_in.copy_(self.poses_out)
        else:
            raise RuntimeError("GraphedTracker.step: no previous output to track from, pass poses")
        self._have_output = True
        return self.replay()

    @torch.inference_mode()
    def step_eager(self, rgb, depth, poses):
        """the same frame without the graphs (for A/B timing and the equality test)"""
        self.rgb.copy_(torch.as_tensor(rgb, device=self.dev))
        self.depth.copy_(torch.as_tensor(depth, device=self.dev))
        self.poses_in.copy_(torch.as_tensor(poses, device=self.dev, dtype=torch.float32).reshape(self.N, 4, 4))
        return self._body()
