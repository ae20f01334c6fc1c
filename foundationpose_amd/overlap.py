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
import torch


# side streams are shared by every predictor of the process (one list per device): ROCm maps HIP streams onto a handful
# of hardware queues, and streams that share a queue serialise -- with one side stream each for the refiner and the scorer
# plus RCCL's stream, the RCCL bench ran SLOWER than one stream (46 ms against 41 ms per step)
_SIDE = {}


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
        st = torch.cuda.Stream(device=device)
        with torch.cuda.stream(st):                  # first use: the runtime binds a stream to its hardware queue lazily
            torch.zeros(1, device=device)
        side.append(st)
        _OVERLAPS[(idx, len(side) - 1)] = _overlaps_with_current(st, device)
    return side[:k]


_OVERLAPS = {}


def _overlaps_with_current(st, device, cycles=400_000):
    """does work on `st` run beside work on the current stream?  Two spin kernels, one per stream, against one alone
    (about 1 ms, once per side stream).  A stream that landed on the main stream's hardware queue serialises with it."""
    if torch.cuda.is_current_stream_capturing():
        return True                                  # cannot measure inside a capture; the eager warm-up has done it
    main = torch.cuda.current_stream(device)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    torch.cuda._sleep(cycles)                        # warm the spin kernel
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
    alone, pair = e[0].elapsed_time(e[1]), e[2].elapsed_time(e[3])
    return pair < 1.5 * alone


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
