"""MI355X-native PointNet2 hot path of IGNF/StrataNet2-Vegetation-Coverage-Maps.

Mirror of the reference's `model/` package for that path:
    from stratanet2_vegetation_coverage_maps_amd import PointNet2, project_to_plotwise_coverages, project_to_2d_rasters
The HIP library (csrc/libstrata_hip.so) is loaded on first use and is mandatory: there is no CPU fallback.
"""
from .point_net2 import PointNet2  # noqa: F401
from .project_to_2d import (project_batch_to_2d_rasters, project_to_2d_rasters,  # noqa: F401
                            project_to_plotwise_coverages)

__all__ = ["PointNet2", "project_to_plotwise_coverages", "project_to_2d_rasters", "project_batch_to_2d_rasters"]
