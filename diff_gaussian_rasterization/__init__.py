"""Drop-in module name for the reference's import at gaussian_renderer/__init__.py:8:

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

With this repository on PYTHONPATH the reference's render_predicted() runs unchanged on MI355X.
"""
from unipre3d_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                                     rasterize_gaussians, rasterize_gaussians_batched)
