"""host logic of the hypothesis sub-batches (foundationpose_amd/overlap.py): partition rules; CPU tensors take the
single-sequence path (no streams)"""
import torch

from foundationpose_amd.overlap import SubBatches


def test_parts_cover_the_batch_contiguously():
    for ns in (1, 2, 3, 4):
        sb = SubBatches(ns)
        for n in (0, 1, 2, 31, 63, 64, 65, 75, 126, 252, 253, 1000):
            parts = sb.parts(n)
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            assert len(parts) <= ns
            if len(parts) > 1:
                sizes = [b - a for a, b in parts]
                assert min(sizes) >= sb.min_rows and max(sizes) - min(sizes) <= 1
    assert SubBatches(2).parts(252) == [(0, 126), (126, 252)]
    assert SubBatches(2).parts(63) == [(0, 63)]          # below 2 x min_rows: one launch sequence
    assert SubBatches(2).parts(2) == [(0, 2)]            # the two-pose broadcasting quirk is never split


def test_cpu_device_has_no_streams():
    sb = SubBatches(2)
    st = sb.streams(torch.device("cpu"), 2)
    assert st == [None, None]
    sb.fork(st)
    sb.join(st)
    with torch.cuda.stream(st[1]):                       # a no-op context
        pass
