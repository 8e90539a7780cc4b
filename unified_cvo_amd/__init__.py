"""unified_cvo_amd -- MI355X (gfx950) backend for unified_cvo's pairwise SE(3) kernel-correlation path.

Only what the path needs: csrc/ (HIP kernels + the C-ABI of include/cvo_hip.h) and the host-side
mirror of the reference interface (CvoGPU / CvoPointCloud / CvoParams).
"""
from .params import CvoParams, read_cvo_params_yaml, parse_cvo_yaml_text  # noqa: F401
from .api import (CvoGPU, CvoPointCloud, DeviceCloud, DeviceCloudAoS, CvoError, AlignResult, CvoFrameGPU,  # noqa: F401
                  BinaryStateGPU, CVO_POINT_DTYPE, cvo_points_from_pointcloud)
