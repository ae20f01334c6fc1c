"""CPU: numpy / plain-Python model of the operand paths of the row-owning kernels (foundationpose_amd/csrc/linear_ln.hip: k_rows512,
k_linear512): the LDS-DMA lane -> (row, 16-byte chunk) map of an A block with its XOR swizzle against the fragment-read address of
(lane, row tile, k-substep); the bank behaviour of those reads; the fragment-packed weight addressing; and the counted vmcnt waits of
the first loop (ll_after_w), checked against a simulation of the order in which a wave issues its requests.  The constants are read
from the source, so the model follows the kernel.  (The kernels themselves are tested on the GPU: tests/test_gpu_parity.py.)"""
import os
import re

import numpy as np

SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "foundationpose_amd", "csrc", "linear_ln.hip")


def _const(name):
    m = re.search(r"\b" + name + r" = (\d+)", open(SRC).read())
    assert m, name
    return int(m.group(1))


LA, LW = _const("LL_LA"), _const("LL_LW")
BM, K, BK, NW = 128, 512, 32, 8
NK = K // BK


def swz(row):
    return (row >> 2) & 3


def test_a_block_dma_and_fragment_addresses_agree():
    """block ks of the A tile: wave w carries rows [16 w, +16), lane l row l / 4 and the LOGICAL chunk that belongs in physical chunk
    l % 4; a fragment read of (row tile t, k-substep kk) by lane (frow, fhalf) must return x[m0 + 32 t + frow][32 ks + 16 kk + 8 fhalf ..]"""
    rng = np.random.default_rng(0)
    M, m0 = 300, 256                                     # a ragged last tile: rows past the end are clamped
    x = rng.integers(1, 2 ** 30, size=(M, K), dtype=np.int64)
    for ks in (0, 5, NK - 1):
        lds = np.zeros(BM * 32, dtype=np.int64)          # one 8 KiB block: 128 rows x 32 halves
        for wid in range(NW):
            for lane in range(64):
                row = wid * 16 + lane // 4
                c = (lane % 4) ^ swz(row)
                m = min(m0 + row, M - 1)
                dst = (wid * 1024 + lane * 16) // 2
                lds[dst:dst + 8] = x[m, ks * BK + c * 8: ks * BK + c * 8 + 8]
        for t in range(BM // 32):
            for lane in range(64):
                frow, fhalf = lane & 31, lane >> 5
                for kk in range(2):
                    addr = frow * 64 + (((2 * kk + fhalf) ^ swz(frow)) << 4) + t * 32 * 64     # a_ptr + row tile offset
                    m = min(m0 + 32 * t + frow, M - 1)
                    k0 = ks * BK + 16 * kk + 8 * fhalf
                    assert np.array_equal(lds[addr // 2: addr // 2 + 8], x[m, k0:k0 + 8]), (ks, t, lane, kk)


def test_fragment_reads_are_bank_conflict_free():
    """ds_read_b128 is served in four groups of 16 lanes; a group is conflict-free when its 16 addresses fall into 16 different
    16-byte slots of the 256-byte bank row (MI355X_MICROARCH.md, LDS)"""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    for kk in range(2):
        for fhalf in range(2):
            for g in groups:
                slots = {((frow * 64 + (((2 * kk + fhalf) ^ swz(frow)) << 4)) % 256) // 16 for frow in g}
                assert len(slots) == 16, (kk, fhalf, g)


def test_packed_weight_block_addresses():
    """request_w: the wave's fragment of k-step ks, k-substep kk, channel tile i starts at halves
    ((w * (K / 16) + ks * 2) * 1024 + (kk * 2 + i) * 512 + lane * 8 of the packed matrix = block ((w * 32 + q) * 2 + i) of
    fp_pack_linear512_f16's layout with q = 2 ks + kk"""
    for w in (0, 3, 7):
        for ks in (0, 9, 15):
            for kk in range(2):
                for i in range(2):
                    off = (w * (K // 16) + ks * 2) * 1024 + (kk * 2 + i) * 512
                    q = 2 * ks + kk
                    assert off == (((w * 32 + q) * 2 + i) * 64) * 8


def ll_after_w(k1):
    n = 0
    for j in range(k1 + 1, min(NK, k1 + LW)):
        n += 4
    for j in range(k1 + 1, k1 + LW):
        n += 1 if j + LA - LW < NK else 0
    return n


def test_counted_waits_of_the_first_loop_match_the_issue_order():
    """ll_after_w(k1) = vector-memory operations a wave has issued AFTER weight fragment k1 when it waits for it.  Simulated issue order:
    prologue [bias?], A blocks 0 .. LA - LW - 1, then pairs [A block j + LA - LW, fragment j] for j < LW; k-step t sends
    [A block t + LA, fragment t + LW] (where they exist); the wait for fragment k1 sits in the middle of k-step k1 - 1 (k1 = 0: at
    the end of the prologue).  An A block is one instruction, a fragment four.  Waiting for `n` outstanding must leave exactly the
    younger operations in flight -- and every A block up to k1 must be older than fragment k1 (it is read right after the barrier)."""
    assert LA > LW and NK > LA
    order = []                                            # (kind, index) per instruction
    for j in range(LA - LW):
        order.append(("A", j))
    for j in range(LW):
        order.append(("A", j + LA - LW))
        order += [("W", j)] * 4
    waits = {0: len(order)}                               # k1 -> number of instructions issued when the wait for fragment k1 executes
    for t in range(NK):
        if t + LA < NK:
            order.append(("A", t + LA))
        if t + LW < NK:
            order += [("W", t + LW)] * 4
        if t + 1 < NK:
            waits[t + 1] = len(order)
    for k1, issued in waits.items():
        last_w = max(i for i in range(issued) if order[i] == ("W", k1))
        assert issued - 1 - last_w == ll_after_w(k1), (k1, issued - 1 - last_w, ll_after_w(k1))
        assert all(order.index(("A", a)) < last_w for a in range(k1 + 1)), k1        # blocks 0 .. k1 are older than fragment k1
    assert ll_after_w(NK - 1) == 0
    # every value the kernel can ask for has a case in ll_wait_vm's switch
    cases = {int(v) for v in re.findall(r"case (\d+): asm volatile\(\"s_waitcnt vmcnt", open(SRC).read())}
    assert {ll_after_w(k) for k in range(NK)} <= cases
