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

    def _body(self, xyz=None):
        """the frame without graphs: same launches, same streams (xyz given: the refine loop alone, on that map)"""
        xyz = self._pre() if xyz is None else xyz
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
        return self.replay_parts()

    def replay_parts(self):
        """the refine loop alone, from the static inputs (rgb, xyz, poses_in) into poses_out (FramePipeline fills xyz itself)"""
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


class FramePipeline:
    """BASELINE configs[4] as its own workload (round 6): the frame loop of run_demo.py:57-66 / estimater.py:250-268 with the per-frame
    upload and depth pre-processing of frame f + 1 running UNDER the refine loop of frame f.

    The tracker's captured graphs address ONE set of static inputs (rgb, xyz_map, poses_in).  Two staging slots are filled on a
    dedicated ingest stream -- pinned host -> device copies of the uint8 colour image, the float depth map and (optionally) the
    hypotheses, the u8 -> f32 conversion, and the three ingest launches (erode, bilateral, back-projection: the tracker's own `_pre`
    arithmetic on the slot's depth) -- and `run(slot)` moves a finished slot into the static inputs with device-to-device copies on the
    compute stream (7.4 MB, a few microseconds) before it replays the part graphs.  Slots are handed over with events (`ready`:
    ingest -> compute, `free`: compute -> ingest); nothing synchronises the host.  Per frame the same kernels see the same bytes as in
    GraphedTracker.step, so the poses are bit-identical to the unpipelined loop (tests/test_gpu_parity.py).  Refinement of frame f + 1
    never starts before frame f is complete (one compute stream order): a tracker's hypotheses depend on the previous pose."""

    def __init__(self, tracker, slots=2):
        if tracker.g_pre is None:
            tracker.capture()
        t = self.trk = tracker
        dev = t.dev
        self.n = int(slots)
        self.ingest, self.ingest_overlaps = self._pick_stream(dev, tracker.refiner.sub.n_streams)
        z = lambda shape, dt: [torch.zeros(shape, dtype=dt, device=dev) for _ in range(self.n)]
        self.rgb_u8, self.rgb = z((t.H, t.W, 3), torch.uint8), z((t.H, t.W, 3), torch.float32)
        self.depth, self.xyz = z((t.H, t.W), torch.float32), z((t.H, t.W, 3), torch.float32)
        self.poses = z((t.N, 4, 4), torch.float32)
        self.has_poses = [False] * self.n
        self.ready = [torch.cuda.Event() for _ in range(self.n)]
        self.free = [torch.cuda.Event() for _ in range(self.n)]
        self.used = [False] * self.n

    @torch.inference_mode()
    def submit(self, slot, rgb_u8_host, depth_host, poses_host=None):
        """enqueue upload + ingest of one frame into `slot` on the ingest stream (host tensors should be pinned).  poses_host None:
        run(slot) tracks from the previous output"""
        t = self.trk
        if self.used[slot]:
            self.ingest.wait_event(self.free[slot])          # the compute stream has copied the slot's previous frame out
        with torch.cuda.stream(self.ingest):
            self.rgb_u8[slot].copy_(rgb_u8_host, non_blocking=True)
            self.depth[slot].copy_(depth_host, non_blocking=True)
            self.has_poses[slot] = poses_host is not None
            if poses_host is not None:
                self.poses[slot].copy_(poses_host, non_blocking=True)
            self.rgb[slot].copy_(self.rgb_u8[slot])
            d = ops.bilateral_filter_depth(ops.erode_depth(self.depth[slot], radius=2), radius=2)
            self.xyz[slot].copy_(ops.depth_to_xyz(d, t.K, zfar=float("inf"), f64_internal=False))
            self.ready[slot].record(self.ingest)
        self.used[slot] = True

    @staticmethod
    def _pick_stream(dev, n_streams):
        """a stream for the ingest that runs BESIDE the compute streams: ROCm binds the streams of a process to a handful of hardware
        queues, and two streams on one queue serialise (overlap.py) -- correct, but then nothing is hidden.  Up to six candidates are
        probed with the spin-kernel measurement of overlap.py against the main stream and against the sub-batch side stream; the
        first that overlaps with both is kept (the last candidate otherwise).  -> (stream, overlaps)"""
        from . import overlap
        if torch.cuda.is_current_stream_capturing():
            return torch.cuda.Stream(device=dev), None
        side = overlap.reserve_streams(dev, max(1, n_streams - 1))
        st = None
        for _ in range(6):
            st = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(st):
                torch.zeros(1, device=dev)                   # first use binds the stream to its hardware queue
            ok = overlap._overlaps_with_current(st, dev)
            if ok and n_streams > 1:
                with torch.cuda.stream(side[0]):
                    ok = overlap._overlaps_with_current(st, dev)
            if ok:
                return st, True
        return st, False

    @torch.inference_mode()
    def run(self, slot, graph=True):
        """refine the frame staged in `slot` on the current stream -> the tracker's static output buffer (valid until the next run).
        graph=False: the same launches issued eagerly (A/B; the same bits)"""
        t = self.trk
        cur = torch.cuda.current_stream(t.dev)
        cur.wait_event(self.ready[slot])
        t.rgb.copy_(self.rgb[slot])
        t.xyz.copy_(self.xyz[slot])
        if self.has_poses[slot]:
            t.poses_in.copy_(self.poses[slot])
        elif t._have_output:
            t.poses_in.copy_(t.poses_out)
        else:
            raise RuntimeError("FramePipeline.run: no previous output to track from, submit poses with the first frame")
        self.free[slot].record(cur)
        t._have_output = True
        return t.replay_parts() if graph else t._body(t.xyz)
