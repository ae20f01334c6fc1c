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


class SubBatches:
    def __init__(self, n_streams=2, min_rows=32):
        self.n_streams = max(1, int(n_streams))
        self.min_rows = int(min_rows)
        self.serial = False      # True: the same parts, all issued on the current stream (isolated per-kernel timing)
        self._side = {}

    def parts(self, N):
        """[(first, last+1)] row ranges: up to n_streams near-equal contiguous parts of at least `min_rows` rows"""
        k = min(self.n_streams, max(1, N // max(1, self.min_rows)))
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
        side = self._side.setdefault(idx, [])
        while len(side) < k - 1:
            side.append(torch.cuda.Stream(device=device))
        return [torch.cuda.current_stream(device)] + side[: k - 1]

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
