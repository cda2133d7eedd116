"""coslam_amd -- MI355X-native implementation of CoSLAM's per-frame hot path.

Only what the path needs: csrc/ (hand-written HIP for gfx950 + the C-ABI of include/coslam_hip.h),
the ctypes mirror of the reference's tracker / pose / BA interface, and the synthetic multi-camera
sequence generator used by tests and bench.  There is no CPU fallback in this package.
"""
from ._lib import CoslamHipError, lib  # noqa: F401
from .klt import (  # noqa: F401
    KLT_SequenceTracker,
    KLT_SequenceTrackerConfig,
    KLT_TrackedFeature,
    KLT_TrackerGroup,
    coslam_config,
)

__version__ = "0.1.0"


def debug_set(key, value):
    """cs_debug_set: a test / diagnostic switch of the whole process ("ba_syrk", "ba_packed", "ba_graphs", "merge_print"; -1 = default).
    The library reads nothing from the environment."""
    from ._lib import check

    check(lib().cs_debug_set(key.encode(), int(value)), "cs_debug_set")


from .pose import IntraCamPoseOption, intraCamEstimate  # noqa: F401,E402
from .ba import BAInterCam, BAOutput, BAStats, BAWindow, BAWorkspace, bundleAdjustRobust  # noqa: F401,E402
from .handback import HandbackCam, handback_cams, handback_dev  # noqa: F401,E402
from .register import (RegisterCam, RegisterPass, register_cams, register_passes, register_search, register_search_dev,  # noqa: F401,E402
                       register_search_passes_dev)
from .poseupdate import (MAP_DYNAMIC, MAP_FALSE, MAP_UNCERTAIN, PoseUpdateCam, TrackHistory, pose_update3d_dev,  # noqa: F401,E402
                         poseupdate_cams)
from .ncc import (ncc_blocks_dev, ncc_epi_mat_dev, ncc_get_blocks_dev, ncc_match_between, ncc_match_between_full,  # noqa: F401,E402
                  ncc_scaled_dims)
from .posegraph import PoseGraphs, after_ba_function, after_ba_record, posegraph_set_poses_dev  # noqa: F401,E402
from .results import ExportCam, export_results_v1  # noqa: F401,E402
