"""Error taxonomy of the drop-in boundary.

Same class names, constructor arguments and message texts as the reference's
core/exceptions.py:2-52, so callers that catch or print them behave the same.
If the real `shinestacker` package is importable its classes are re-used, which
makes `except shinestacker.core.exceptions.ShapeError` work unchanged.
"""
try:  # pragma: no cover - only where the reference package is installed
    from shinestacker.core.exceptions import (  # type: ignore
        FocusStackError, InvalidOptionError, ImageLoadError, ImageSaveError, AlignmentError,
        BitDepthError, ShapeError, RunStopException)
except Exception:  # noqa: BLE001 - any import failure means "not installed"

    class FocusStackError(Exception):
        """Root of every error raised by the stacking pipeline."""

    class InvalidOptionError(FocusStackError):
        def __init__(self, option, value, details=""):
            self.option, self.value, self.details = option, value, details
            tail = f": {details}" if details != "" else ""
            super().__init__(f"Invalid option {option} = {value}{tail}")

    class ImageLoadError(FocusStackError):
        def __init__(self, path, details=""):
            self.path, self.details = path, details
            tail = f": {details}" if details != "" else ""
            super().__init__(f"Failed to load {path}{tail}")

    class ImageSaveError(FocusStackError):
        def __init__(self, path, details=""):
            self.path, self.details = path, details
            tail = f": {details}" if details != "" else ""
            super().__init__(f"Failed to save {path}{tail}")

    class AlignmentError(FocusStackError):
        def __init__(self, index, details):
            self.index, self.details = index, details
            super().__init__(f"Alignment failed for image {index}: {details}")

    class BitDepthError(FocusStackError):
        def __init__(self, dtype_ref, dtype):
            super().__init__(f"Image has type {dtype}, expected {dtype_ref}.")

    class ShapeError(FocusStackError):
        def __init__(self, shape_ref, shape):
            super().__init__(
                f"\nImage has shape ({shape[1]}x{shape[0]}), while it was expected "
                f"({shape_ref[1]}x{shape_ref[0]}).\n")

    class RunStopException(FocusStackError):
        def __init__(self, name):
            label = f"{name} " if name != "" else ""
            super().__init__(f"Job {label}stopped")


class DeviceError(RuntimeError):
    """The HIP library is missing, failed to load, or a device call failed."""
