"""host-side logic of the predictors that needs neither a GPU nor the HIP library"""
import numpy as np
import pytest


def test_resolve_shared_translation():
    """host side of PoseRefinePredictor.predict(shared_translation=...): no GPU involved"""
    import torch
    from foundationpose_amd.predict_pose_refine import resolve_shared_translation as rs
    P = np.tile(np.eye(4), (5, 1, 1))
    P[:, :3, 3] = [0.1, -0.2, 0.7]
    assert rs(P, None) is True and rs(P, True) is True and rs(P, False) is False
    assert rs(torch.as_tensor(P), None) is True and rs(P.tolist(), None) is True
    assert rs(P[:1], None) is False and rs(P[:1], True) is False            # one hypothesis: nothing to share
    Q = P.copy()
    Q[3, 1, 3] += 1e-6
    assert rs(Q, None) is False and rs(Q, False) is False
    with pytest.raises(ValueError):
        rs(Q, True)
    R = P.copy()
    R[2, 0, 3] += 1e-12                                                     # equal as the float32 values that are uploaded
    assert rs(R, None) is True
    assert rs(np.zeros((0, 4, 4)), None) is False
