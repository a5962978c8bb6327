"""MI355X-native speech-enhancement hot path (STFT -> denoiser -> iSTFT, mixing) behind the
reference's module interfaces.  Kernels live in libnoise_hip.so (hand-written HIP for gfx950)."""
from . import _lib  # noqa: F401  (fails loudly when the kernel library is missing)
from .stft import STFT  # noqa: F401
from .mixer import segmental_snr_mixer, activitydetector  # noqa: F401
from .models import VADSpectralV2, VADMobileNetV2  # noqa: F401
from .optim import FusedAdamW  # noqa: F401
from .data import VADNoiseDataModule  # noqa: F401
from .streaming import StreamingDenoiser  # noqa: F401
from .augment import AugmentPools, VADTrainAugment, add_pyreverb, impulse_response  # noqa: F401
