"""MI355X-native OpenProvence inference path."""

from .config import DEFAULT_PROCESS_THRESHOLD, EncoderDims, OpenProvenceConfig, UnsupportedModelError

__all__ = ["DEFAULT_PROCESS_THRESHOLD", "EncoderDims", "OpenProvenceConfig", "UnsupportedModelError"]
