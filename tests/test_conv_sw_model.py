"""CPU: numpy model of the addressing of k_conv_sw (foundationpose_amd/csrc/conv_sw.hip), the shifted-window 3x3
convolution kernel: the LDS-DMA lane -> (padded pixel, 16-byte chunk) map of the patch staging with its XOR swizzle, and
the fragment-read address of (lane, pixel tile t, tap, k-substep).  Together they must deliver, for GEMM row m and
k = (tap, ci), the element x[b, oy+ky, ox+kx, ci] of the zero-bordered NHWC input -- including tiles that cross image
rows and images, and the clamped rows of the last tile -- and the 16-lane groups of ds_read_b128 must be bank-conflict
free for consecutive patch rows at any offset.  (The kernel itself is tested on the GPU: tests/test_gpu_amp.py.)"""
import numpy as np
import pytest


def swz(row): return (row >> 2) & 3
def run(B, Ho, Wo, Cin, m0, TM, NWN, seed=0, BM=256):
    Hp, Wp = Ho+2, Wo+2
    HoWo = Ho*Wo
    M = B*HoWo
    rng = np.random.default_rng(seed)
    cstride = Cin
    x = rng.integers(1, 2**30, size=(B*Hp*Wp*cstride,), dtype=np.int64)   # unique-ish element ids
    def q_of(m):
        b = m // HoWo; r = m - b*HoWo; oy = r // Wo; ox = r - oy*Wo
        return (b*Hp + oy)*Wp + ox
    q0 = q_of(m0); qmax = B*Hp*Wp - 1
    PROWS = 512 if BM == 256 else 768
    PI = PROWS // 16 // 8
    for cc in range(Cin//32):
        # ---- patch image in LDS (bytes -> we store element ids per half element)
        lds = np.zeros((PROWS*32,), dtype=np.int64)   # 512 rows x 32 halves
        for wid in range(8):
            for j in range(PI):
                for lane in range(64):
                    row = (wid*PI + j)*16 + (lane >> 2)
                    c = (lane & 3) ^ swz(row)
                    q = min(q0 + row, qmax)
                    src = q*cstride + c*8 + cc*32          # element offset
                    dst_byte = (wid*PI + j)*1024 + lane*16
                    lds[dst_byte//2: dst_byte//2 + 8] = x[src:src+8]
        # ---- fragment reads
        assert (BM // (32*TM)) * NWN == 8
        for wm in range(BM // (32*TM)):
            for t in range(TM):
                for lane in range(64):
                    frow, fhalf = lane & 31, lane >> 5
                    m = m0 + wm*(32*TM) + t*32 + frow
                    mc = min(m, M-1)
                    arow = q_of(mc) - q0
                    for tap in range(9):
                        ky, kx = tap//3, tap%3
                        pr = arow + ky*Wp + kx
                        assert 0 <= pr < PROWS, (pr, m0, m)
                        a0 = (pr << 6) + ((fhalf ^ swz(pr)) << 4)
                        for kk, addr in enumerate((a0, a0 ^ 32)):
                            got = lds[addr//2: addr//2 + 8]
                            b = mc // HoWo; r = mc - b*HoWo; oy = r//Wo; ox = r - oy*Wo
                            src = ((b*Hp + oy+ky)*Wp + ox+kx)*cstride + cc*32 + 16*kk + 8*fhalf
                            assert np.array_equal(got, x[src:src+8]), (m, tap, kk, lane)
    return True
# bank-conflict check of the fragment reads: 16-lane groups of ds_read_b128 must hit 16 distinct 16-B slots of the 256-B bank row
def conflicts(arows, shift):
    groups = [[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
    worst = 1
    for g in groups:
        for fhalf in (0,1):
            slots = {}
            for l in g:
                pr = arows[l] + shift
                a0 = (pr << 6) + ((fhalf ^ swz(pr)) << 4)
                slot = (a0 % 256)//16
                slots[slot] = slots.get(slot,0)+1
            worst = max(worst, max(slots.values()))
    return worst

# the two tile shapes of the kernel: 256 x 256 (TM 4, 4 wave columns) and 512 x 128 (TM 4, 2 wave columns, 768 patch rows)
@pytest.mark.parametrize("B,Ho,Cin,m0,TM,NWN,BM", [(3, 40, 64, 0, 4, 4, 256), (3, 40, 64, 1536, 4, 4, 256), (3, 40, 64, 4608, 4, 4, 256),
                                                   (5, 20, 64, 256, 4, 4, 256), (5, 20, 64, 1792, 4, 4, 256),
                                                   (3, 40, 64, 0, 4, 2, 512), (3, 40, 64, 1536, 4, 2, 512), (3, 40, 64, 4608, 4, 2, 512),
                                                   (3, 24, 64, 1536, 4, 2, 512), (6, 20, 64, 1024, 4, 2, 512)])
def test_conv_sw_patch_and_fragment_addressing(B, Ho, Cin, m0, TM, NWN, BM):
    assert run(B, Ho, Ho, Cin, m0, TM, NWN, BM=BM)


def test_conv_sw_fragment_reads_are_bank_conflict_free():
    shifts = (0, 1, 2, 42, 43, 44, 84, 85, 86, 22, 23, 24, 45, 46)
    assert max(conflicts([off + l for l in range(32)], s) for off in range(64) for s in shifts) == 1
    # one image-row crossing inside the 32 rows of a fragment (a gap of 2 patch rows): at most 2-way
    assert max(conflicts([l if l < c else l + 2 for l in range(32)], s) for c in range(1, 32) for s in shifts) <= 2
