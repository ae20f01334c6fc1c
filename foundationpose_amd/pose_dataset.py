"""BatchPoseData attribute bag (reference: learning/datasets/pose_dataset.py:66-134)."""
import torch

_FIELDS = ("rgbAs", "rgbBs", "depthAs", "depthBs", "normalAs", "normalBs", "poseA", "poseB", "maskAs", "maskBs",
           "xyz_mapAs", "xyz_mapBs", "tf_to_crops", "crop_masks", "Ks", "model_pts", "mesh_diameters", "labels")


class BatchPoseData:
    def __init__(self, **kw):
        unknown = set(kw) - set(_FIELDS)
        if unknown:
            raise TypeError(f"unknown BatchPoseData fields: {sorted(unknown)}")
        for f in _FIELDS:
            setattr(self, f, kw.get(f))

    def _map(self, fn):
        for f in _FIELDS:
            v = getattr(self, f)
            if torch.is_tensor(v):
                try:
                    setattr(self, f, fn(v))
                except Exception:
                    pass
        return self

    def pin_memory(self):
        return self._map(lambda t: t.pin_memory())

    def cuda(self):
        return self._map(lambda t: t.cuda())

    def select_by_indices(self, ids):
        out = BatchPoseData()
        for f in _FIELDS:
            v = getattr(self, f)
            if v is not None:
                setattr(out, f, v[ids.to(v.device)])
        return out
