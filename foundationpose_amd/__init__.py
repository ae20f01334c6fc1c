"""MI355X-native render-and-compare 6D pose refinement (drop-in for the FoundationPose hot path).

Public surface mirrors the reference (NVlabs/FoundationPose):
  FoundationPose            (estimater.py:18-268)            -> foundationpose_amd.estimater
  PoseRefinePredictor       (learning/training/predict_pose_refine.py:93-295) -> foundationpose_amd.predict_pose_refine
  ScorePredictor            (learning/training/predict_score.py:117-226)      -> foundationpose_amd.predict_score
  nvdiffrast_render & co.   (Utils.py)                       -> foundationpose_amd.Utils
  dr.RasterizeCudaContext   (nvdiffrast)                     -> foundationpose_amd.dr

The compute path is libfp_amd.so (hand-written HIP for gfx950, C ABI in include/fp_amd.h);
there is no CPU fallback: using an op without the built library / without a GPU raises.
"""
__version__ = "0.1.0"
