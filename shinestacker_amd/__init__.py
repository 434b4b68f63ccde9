"""shinestacker_amd -- MI355X-native focus-stacking hot path behind shinestacker's
FocusStack / StackJob action API."""
from .errors import (FocusStackError, InvalidOptionError, ImageLoadError, ImageSaveError,  # noqa: F401
                     AlignmentError, BitDepthError, ShapeError, RunStopException, DeviceError)
from .defaults import constants  # noqa: F401
from .pyramid import BaseStackAlgo, PyramidStack  # noqa: F401
from .depth_map import DepthMapStack  # noqa: F401
from .actions import (StackJob, FocusStack, FocusStackBunch, CombinedActions, SubAction,  # noqa: F401
                      get_bunches)

from .align import AlignFrames, align_images  # noqa: F401,E402
from .balance import BalanceFrames  # noqa: F401,E402

__all__ = ["AlignFrames", "BalanceFrames", "align_images", "PyramidStack", "DepthMapStack", "BaseStackAlgo", "StackJob", "FocusStack", "FocusStackBunch",
           "CombinedActions", "SubAction", "get_bunches", "constants"]
