"""Stand-ins for the `dataset` objects the predictors expose (reference: learning/datasets/h5_dataset.py).
Only ``transform_batch`` is on the hot path there (:79-127 refine, :137-179 score); in this implementation that
arithmetic (rgb/255, xyz - t, 1/radius, invalid masks) is fused into fp_render_crops / fp_warp_crops, so the
batch handed to ``transform_batch`` is already network-ready (make_crop_data_batch marks it with ``batch.AB``, the
fused network-input buffer) and is returned unchanged.  A batch that did NOT come from this package's
make_crop_data_batch -- raw 0..255 colours, metric xyz -- would need the un-fused arithmetic, which does not exist
here: it is refused instead of being passed through silently.  H5 file loading is training-only and out of scope."""


class _FusedTransformDataset:
    mode = "test"

    def __init__(self, cfg, h5_file=None, mode="test", max_num_key=None, cache_data=None):
        self.cfg = cfg
        self.mode = mode

    def __len__(self):
        return 1

    def transform_batch(self, batch, H_ori=None, W_ori=None, bound=1):
        if getattr(batch, "AB", None) is None:
            raise RuntimeError(
                "transform_batch: this batch was not produced by foundationpose_amd's make_crop_data_batch (no fused "
                "network-input buffer `batch.AB`); the normalisation of h5_dataset.py:79-170 is fused into "
                "fp_render_crops / fp_warp_crops and cannot be applied to an un-fused batch")
        return batch


class PoseRefinePairH5Dataset(_FusedTransformDataset):
    xyz_invalid_thr = 0.001


class ScoreMultiPairH5Dataset(_FusedTransformDataset):
    xyz_invalid_thr = 0.1
