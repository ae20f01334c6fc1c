"""Hypothesis sub-batches on concurrent HIP streams.

Every hypothesis goes through the refiner (and through the scorer's encoder) independently of the others, so a batch
of N can run as k sub-batches whose launch sequences never touch each other's data.  Issued on k streams, the
hardware queues overlap them: the large-tile GEMM kernels hold one workgroup per CU, so a layer of U tiles leaves the
chip partly idle in its last round (788 tiles on 256 CUs: 3.08 rounds of work in 4) and between two dependent launches
(~10 us each, ~130 launches per step); with a second, independent launch sequence in flight those holes are filled with
the other sub-batch's workgroups.  Nothing changes per element: each hypothesis sees the same kernels, the same tile
shapes and the same summation order as in one batch (results are bit-identical, tests/test_gpu_parity.py).

The fork / join edges are stream waits on events (torch.cuda.Stream.wait_stream), i.e. parallel branches under hipGraph
capture.  Every buffer a sub-batch touches is its own: the encoder's activation sets are keyed by `slot`, the rasteriser
scratch by stream (ops._workspace) or passed per part by the caller.
"""
import logging

import numpy as np
import torch

log = logging.getLogger(__name__)


# side streams are shared by every predictor of the process (one list per device): ROCm maps HIP streams onto a handful
# of hardware queues, and streams that share a queue serialise -- with one side stream each for the refiner and the scorer
# plus RCCL's stream, the RCCL bench ran SLOWER than one stream (46 ms against 41 ms per step)
_SIDE = {}
# Experiment / test switches (round 6: module constants, not environment variables -- the release package reads none).  FORCE_OVERLAP:
# None = probe the side stream and run the exactness canary (the product), True / False = skip both and force sub-batch overlap on /
# off for side streams created from now on; SIDE_PRIORITY: HIP priority of the side streams (0 = default).
FORCE_OVERLAP = None
SIDE_PRIORITY = 0


def reserve_streams(device, k=1):
    """create -- and use once -- the k shared side streams of `device` now.  ROCm maps the streams of a process onto a few
    hardware queues (4 by default), binding a stream at its first use; streams that share a queue serialise.  A process
    that also initialises RCCL (which brings streams of its own) should call this BEFORE
    torch.distributed.init_process_group: measured on one MI355X with RCCL initialised, 37.7 ms per bench step with the
    side stream bound first, 43.5-45.9 ms when it was bound afterwards (slower than one stream: 41.0 ms)"""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    side = _SIDE.setdefault(idx, [])
    while len(side) < k:
        st = torch.cuda.Stream(device=device, priority=int(SIDE_PRIORITY)) if SIDE_PRIORITY else torch.cuda.Stream(device=device)
        with torch.cuda.stream(st):                  # first use: the runtime binds a stream to its hardware queue lazily
            torch.zeros(1, device=device)
        side.append(st)
        if FORCE_OVERLAP is not None:                              # True / False: skip the probes and force the decision
            ok, why = bool(FORCE_OVERLAP), "forced by overlap.FORCE_OVERLAP"
        else:
            ok = _overlaps_with_current(st, device)
            why = "side stream runs beside the main stream" if ok else "side stream shares the main stream's hardware queue"
            if ok and not _exact_next_to_gemm(st, device):
                ok, why = False, "CANARY FAILED: the rasteriser is not bit-exact next to a GEMM on another stream"
        log.info("overlap: side stream %d of cuda:%d %s (%s)", len(side) - 1, idx, "enabled" if ok else "disabled", why)
        _OVERLAPS[(idx, len(side) - 1)] = ok
    return side[:k]


_OVERLAPS = {}


def _overlaps_with_current(st, device, cycles=400_000, repeats=3):
    """does work on `st` run beside work on the current stream?  Two spin kernels, one per stream, against one alone
    (about 1 ms each; the median of `repeats` measurements, once per side stream).  A stream that landed on the main
    stream's hardware queue serialises with it."""
    if torch.cuda.is_current_stream_capturing():
        return True                                  # cannot measure inside a capture; the eager warm-up has done it
    main = torch.cuda.current_stream(device)
    torch.cuda._sleep(cycles)                        # warm the spin kernel
    ratios = []
    for _ in range(repeats):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        torch.cuda.synchronize(device)
        e[0].record(main)
        torch.cuda._sleep(cycles)
        e[1].record(main)
        st.wait_stream(main)
        e[2].record(main)
        with torch.cuda.stream(st):
            torch.cuda._sleep(cycles)
        torch.cuda._sleep(cycles)
        main.wait_stream(st)
        e[3].record(main)
        torch.cuda.synchronize(device)
        ratios.append(e[2].elapsed_time(e[3]) / max(e[0].elapsed_time(e[1]), 1e-6))
    return sorted(ratios)[len(ratios) // 2] < 1.5


def _exact_next_to_gemm(st, device, launches=24):
    """Canary for the one way concurrent streams were ever seen to change a result (on MI355X / ROCm 7.2 a
    build of the rasteriser WITH packed-fp32 VALU instructions returned wrong lanes while an MFMA kernel of another stream
    shared the chip; the library is built without them and csrc/Makefile checks the objects).  Before sub-batches are
    allowed to overlap, the rasteriser runs `launches` times beside a GEMM on the side stream and every output must be
    bit-identical to the launch that ran alone (~10 ms, once per process and device).  A failure disables the overlap --
    results stay right, the step gets slower -- and is logged as an error."""
    if torch.cuda.is_current_stream_capturing():
        return True
    from . import ops, synthetic as syn
    from .Utils import get_mesh_handle, make_mesh_tensors
    from .mesh import make_can_mesh
    mesh = make_can_mesh()
    gm = make_mesh_tensors(mesh, device=device)      # keeps the device tensors of the handle alive for the duration
    h = get_mesh_handle(gm)
    n = 38
    T = syn.gt_pose(0).astype(np.float32)
    P = torch.as_tensor(syn.perturbed_poses(T, n, seed=3, max_trans=0.01, max_rot_deg=170.0).astype(np.float32), device=device)
    diam = float(np.linalg.norm(mesh.vertices.max(0) - mesh.vertices.min(0)))
    _, bb = ops.crop_windows(P, syn.YCBV_K, diam, 1.2, (160, 160))
    A = torch.zeros((n, 6, 160, 160), dtype=torch.float16, device=device)
    ws = torch.empty(max(16, ops.workspace_bytes(n, h.V, h.T, 160, 160)), dtype=torch.uint8, device=device)
    x = torch.randn((14800, 512), device=device, dtype=torch.float16)
    w = torch.randn((512, 512), device=device, dtype=torch.float16)
    y = torch.empty((14800, 512), device=device, dtype=torch.float16)

    def render():
        return ops.render_crops(h, P, bb, syn.YCBV_K, syn.H, syn.W, out_hw=(160, 160), mesh_diameter=diam, xyz_thr=0.001,
                                normalize_xyz=True, A_out=A, workspace=ws, want=("A",))["A"]
    base = render().clone()
    torch.matmul(x, w.t(), out=y)                     # warm rocBLAS outside the measured overlap
    torch.cuda.synchronize(device)
    main = torch.cuda.current_stream(device)
    bad = 0
    for _ in range(launches):
        st.wait_stream(main)
        r = render()
        with torch.cuda.stream(st):
            torch.matmul(x, w.t(), out=y)
        main.wait_stream(st)
        bad += int(not torch.equal(r, base))
    torch.cuda.synchronize(device)
    if bad:
        log.error("overlap canary: %d of %d rasteriser launches next to a GEMM differ from the launch that ran alone; "
                  "sub-batches will NOT overlap on this device", bad, launches)
    return bad == 0


def reserved(device, k=1):
    """have the first k side streams of `device` been created and probed already? (no side effect: safe inside a stream capture)"""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return len(_SIDE.get(idx, ())) >= k


def side_streams_overlap(device, k=1):
    """True if the first k side streams of `device` were measured to run beside the main stream"""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    reserve_streams(device, k)
    return all(_OVERLAPS.get((idx, i), True) for i in range(k))


class SubBatches:
    def __init__(self, n_streams=2, min_rows=32):
        self.n_streams = max(1, int(n_streams))
        self.min_rows = int(min_rows)
        self.serial = False      # True: the same parts, all issued on the current stream (isolated per-kernel timing)

    def parts(self, N, device=None):
        """[(first, last+1)] row ranges: up to n_streams near-equal contiguous parts of at least `min_rows` rows.  With a
        CUDA `device` given, one part if its side streams do not overlap with the main stream (sub-batches issued one
        after the other are slower than one launch sequence over the whole batch)"""
        k = min(self.n_streams, max(1, N // max(1, self.min_rows)))
        if k > 1 and device is not None and torch.device(device).type == "cuda" and not self.serial \
                and not side_streams_overlap(device, k - 1):
            k = 1
        if k <= 1:
            return [(0, N)]
        base, extra = divmod(N, k)
        out, a = [], 0
        for i in range(k):
            b = a + base + (1 if i < extra else 0)
            out.append((a, b))
            a = b
        return out

    def streams(self, device, k):
        """the current stream (part 0) + k-1 side streams of `device` (created once and kept)"""
        device = torch.device(device)
        if device.type != "cuda":
            return [None] * k        # torch.cuda.stream(None) is a no-op context
        if self.serial:
            return [torch.cuda.current_stream(device)] * k
        idx = device.index if device.index is not None else torch.cuda.current_device()
        return [torch.cuda.current_stream(device)] + reserve_streams(device, k - 1)

    @staticmethod
    def fork(streams):
        for s in streams[1:]:
            if s is not None and s != streams[0]:
                s.wait_stream(streams[0])

    @staticmethod
    def join(streams):
        for s in streams[1:]:
            if s is not None and s != streams[0]:
                streams[0].wait_stream(s)
