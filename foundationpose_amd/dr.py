"""nvdiffrast look-alike so that reference drivers (`import nvdiffrast.torch as dr`; run_demo.py:40,
estimater.py:102,168) keep working: the context is an opaque token, rasterisation happens inside
fp_render_crops (hand-written HIP, no nvdiffrast)."""
import torch


class RasterizeCudaContext:
    def __init__(self, device=None):
        self.device = torch.device(device) if device is not None else torch.device("cuda")


class RasterizeGLContext(RasterizeCudaContext):
    pass
